"""Round 3: the SSD search of searchRightKeypointEpipolar (StereoMatcher.cpp:196-423) on the matrix cores
(`ssd_search_mfma`, v_mfma_i32_16x16x64_i8 on the images shifted to signed bytes) against the oracle and against the
v_dot4 search it replaces (kvfe_config.ssd_impl = 1), tolerance 0: right keypoints, statuses and scores.  Template / stripe
geometries cover both K-step variants (compile-time 2 = the shipped 101-column template, run-time 1 and 3), stripes
higher than the template (several offset rows), template widths with one and three bytes in the last dword, and keypoints whose template or stripe is clamped at either image border."""
import os

import numpy as np
import pytest
from PIL import Image

import oracle_lib as O
from kimera_vio_amd import frontend as F
from kimera_vio_amd import params as P

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _gray(name):
    return np.array(Image.open(os.path.join(G, name)).convert("L"))


def _cams():
    return (P.load_camera_params(os.path.join(G, "sensorLeft.yaml")),
            P.load_camera_params(os.path.join(G, "sensorRight.yaml")))


def _params(tc, tr, extra, min_dist):
    p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=0)
    p.stereo.templ_cols, p.stereo.templ_rows, p.stereo.stripe_extra_rows = tc, tr, extra
    p.stereo.min_point_dist = min_dist
    return p


def _keypoints(left):
    kps, _ = O.good_features_to_track(left, 150, 0.001, 10, 3)
    h, w = left.shape
    ys = np.linspace(20, h - 20, 9, dtype=np.float32)
    border = [[x, y] for y in ys for x in (0.0, 3.4, 17.0, 49.6, 51.0, w - 52.0, w - 50.4, w - 18.0, w - 2.0)]
    rows = [[w / 2, y] for y in (0.0, 4.6, 5.4, 7.0, h - 8.0, h - 6.5, h - 5.0, h - 1.0)]
    return np.concatenate([kps, np.array(border + rows, np.float32)]).astype(np.float32)


@pytest.mark.parametrize("tc,tr,extra,min_dist", [   # (template sizes are odd, extra rows even: the reference's CHECKs)
    (101, 11, 0, 0.5),    # shipped (EuRoC): 2 K steps, 101 offsets, odd template height (one phantom row)
    (101, 11, 2, 0.5),    # three offset rows
    (103, 11, 0, 0.5),    # three bytes in the last template dword
    (99, 9, 0, 0.5),      # 2 K steps, 25 template dwords
    (41, 7, 2, 0.4),      # one K step
    (121, 5, 0, 0.6),     # three K steps
    (49, 3, 4, 0.3),      # template + 15 = 64: exactly one K step; 160-odd offsets, five offset rows
])
def test_ssd_search_matrix_cores_vs_oracle_and_dot4(tc, tr, extra, min_dist):
    L, R = _cams()
    p = _params(tc, tr, extra, min_dist)
    ocam = O.Camera(L, R)
    left, right = _gray("left_img_0.png"), _gray("right_img_0.png")
    kps = _keypoints(left)
    st = np.zeros(len(kps), np.uint8)
    exy, est, esc = ocam.get_right_keypoints_rectified(left, right, kps, st, p.stereo)
    for impl in (0, 1):   # kvfe_config.ssd_impl: 0 = matrix cores where the geometry fits, 1 = v_dot4 everywhere
        c = F.Context(L, R, p, ssd_impl=impl)
        try:
            rxy, rst, sc = c.get_right_keypoints_rectified(left, right, kps, st)
            assert np.array_equal(rst, est), impl
            assert np.array_equal(rxy, exy), impl
            assert np.array_equal(sc, esc), impl
        finally:
            c.close()
    assert (est == 0).sum() > 60


def test_ssd_search_matrix_cores_shifted_image():
    """the reference's own construction (tests/testStereoMatcher.cpp:272-388): the right image is the left one shifted,
    so every interior match is exact (SSD 0) -- with the EuRoC stripe width, which takes the matrix-core path"""
    L, R = _cams()
    p = _params(101, 11, 0, 0.5)
    ocam = O.Camera(L, R)
    left = _gray("left_img_0.png")
    kps, _ = O.good_features_to_track(left, 100, 0.01, 10, 3)
    cols = left.shape[1]
    c = F.Context(L, R, p)
    try:
        for offset in (-20, -5):
            right = np.zeros_like(left)
            right[:, : cols + offset] = left[:, -offset:]
            st = np.zeros(len(kps), np.uint8)
            rxy, rst, sc = c.get_right_keypoints_rectified(left, right, kps, st)
            exy, est, esc = ocam.get_right_keypoints_rectified(left, right, kps, st, p.stereo)
            assert np.array_equal(rxy, exy) and np.array_equal(rst, est) and np.array_equal(sc, esc)
            ok = (rst == 0) & (kps[:, 0] + offset >= 60) & (kps[:, 0] + 60 < cols)
            assert ok.sum() > 30
            assert np.all(np.abs(rxy[ok, 0] - (np.round(kps[ok, 0]) + offset)) < 0.5)
    finally:
        c.close()
