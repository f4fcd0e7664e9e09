"""kvfe_stereo_params.ssd_tie_policy: which minimum searchRightKeypointEpipolar takes when the
SSDs of two offsets are closer than float32 can tell apart.  cv::matchTemplate's result matrix is CV_32F
(StereoMatcher.cpp:388-392); KVFE_SSD_TIE_EXACT (default) takes the first minimum of the exact integers,
KVFE_SSD_TIE_F32 the first minimum of the float32-rounded values.  profiles/r4_ssd_tie_exposure.md: on the reference's
EuRoC frames the two agree on all 16 209 matches, so the scene here is built to be as hostile as possible -- two nearly
flat images 180 grey levels apart, every SSD ~3.6e7 (float32 spacing 4) and many offsets within 2 of each other."""
import copy
import os

import numpy as np
import pytest

import oracle_lib as O
from kimera_vio_amd import _abi as abi
from kimera_vio_amd import params as P

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _scene(seed=5, w=752, h=480):
    """left: 240 with a sparse lattice of 241; right: 60 with 0.5 % random 61.  Every SSD is ~3.6e7 (float32 spacing 4),
    and the SSDs of different offsets differ by sums of +-359 and +-361: differences of 2 are common."""
    rng = np.random.RandomState(seed)
    left = np.full((h, w), 240, np.uint8)
    left[::7, ::5] = 241
    right = np.full((h, w), 60, np.uint8)
    right[rng.rand(h, w) < 0.005] = 61
    ys, xs = np.meshgrid(np.arange(40, h - 40, 37), np.arange(160, w - 60, 53), indexing="ij")
    kps = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.float32)
    return left, right, kps


def _params(policy):
    p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=0)
    p.stereo.ssd_tie_policy = policy
    return p


def _numpy_argmins(left, right, kps, sp, stripe_cols):
    """(exact, float32-rounded) first-minimum x of the right keypoint per left keypoint, straight from the definition"""
    tc, tr = sp.templ_cols, sp.templ_rows
    h, w = left.shape
    out = []
    for x, y in kps:
        rx, ry = int(round(float(x))), int(round(float(y)))
        ty, tx = ry - (tr - 1) // 2, rx - (tc - 1) // 2
        sx = rx + (tc - 1) // 2 - stripe_cols
        assert tx >= 0 and tx + tc <= w - 1 and sx >= 0 and sx + stripe_cols <= w - 1 and ty >= 0 and ty + tr <= h - 1
        T = left[ty:ty + tr, tx:tx + tc].astype(np.int64)
        S = right[ty:ty + tr, sx:sx + stripe_cols].astype(np.int64)
        ssd = np.array([int(((S[:, o:o + tc] - T) ** 2).sum()) for o in range(stripe_cols - tc + 1)], np.int64)
        assert ssd.min() > 2 ** 24
        out.append((sx + int(np.argmin(ssd)) + (tc - 1) // 2, sx + int(np.argmin(ssd.astype(np.float32))) + (tc - 1) // 2))
    return np.array(out)


def _stripe_cols(cam, sp):
    c = int(round(cam.rect.P1[0] * cam.rect.baseline / sp.min_point_dist)) + sp.templ_cols + 4
    return c + 1 - c % 2


def test_oracle_ssd_tie_policies_follow_their_definitions():
    L = P.load_camera_params(os.path.join(G, "sensorLeft.yaml"))
    R = P.load_camera_params(os.path.join(G, "sensorRight.yaml"))
    cam = O.Camera(L, R)
    left, right, kps = _scene()
    st = np.zeros(len(kps), np.uint8)
    exp = _numpy_argmins(left, right, kps, _params(0).stereo, _stripe_cols(cam, _params(0).stereo))
    assert (exp[:, 0] != exp[:, 1]).sum() >= 5, "the scene must separate the two policies"
    for policy in (abi.SSD_TIE_EXACT, abi.SSD_TIE_F32):
        xy, rst, _ = cam.get_right_keypoints_rectified(left, right, kps, st, _params(policy).stereo)
        assert np.all(rst == abi.KP_VALID)
        assert np.array_equal(xy[:, 0].astype(np.int64), exp[:, policy]), policy
        assert np.array_equal(xy[:, 1], np.round(kps[:, 1]))


@pytest.mark.gpu
@pytest.mark.parametrize("ssd_impl", [0, 1])
def test_gpu_ssd_tie_policies_match_the_oracle(ssd_impl):
    """both search implementations (matrix cores / v_dot4) under both policies, bit for bit against the oracle"""
    from kimera_vio_amd import frontend as F
    L = P.load_camera_params(os.path.join(G, "sensorLeft.yaml"))
    R = P.load_camera_params(os.path.join(G, "sensorRight.yaml"))
    cam = O.Camera(L, R)
    left, right, kps = _scene()
    st = np.zeros(len(kps), np.uint8)
    res = {}
    for policy in (abi.SSD_TIE_EXACT, abi.SSD_TIE_F32):
        p = _params(policy)
        exy, est, esc = cam.get_right_keypoints_rectified(left, right, kps, st, p.stereo)
        c = F.Context(L, R, p, ssd_impl=ssd_impl)
        try:
            rxy, rst, sc = c.get_right_keypoints_rectified(left, right, kps, st)
        finally:
            c.close()
        assert np.array_equal(rst, est) and np.array_equal(rxy, exy) and np.array_equal(sc, esc), policy
        res[policy] = rxy
    assert (res[0][:, 0] != res[1][:, 0]).sum() >= 5
