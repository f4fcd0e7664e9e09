"""Known-answer test of the CPU oracle's PnP tracking (oracle/opengv_epnp.inl, kimera::pnp), restating the reference's
own test of this call site:

  tests/testTracker.cpp:1613-1800  TEST_F(TestTracker, PnPTracking)

The scene is the reference's: the stereo rig of tests/data/ForStereoFrame with the left camera at (I, [0, 0, -2]) and
the right one at (I, [1, 0, -2]), 22 landmarks (the unit cube + 14 more) projected into the rectified left camera,
bearing vectors from the float32 pixel through P1 (getBearingVectorFromUndistortedKeypoint), 3 outliers (wrong
landmark / pixel pairs), EPNP, ransac_threshold_pnp 0.5 px, min_pnp_inliers 10.  Assertions as upstream: pnp() returns
true, 22 inliers, the pose equals the rectified left camera pose within 1e-5 (translation and quaternion).
"""
import os

import numpy as np
import pytest

import oracle_lib as O
from kimera_vio_amd import _abi as abi
from kimera_vio_amd import params as P

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

INLIER_LMKS = np.array([
    (0.0, 0.0, 0.0), (0.0, 0.0, 1.0), (0.0, 1.0, 0.0), (0.0, 1.0, 1.0),
    (1.0, 0.0, 0.0), (1.0, 0.0, 1.0), (1.0, 1.0, 0.0), (1.0, 1.0, 1.0),
    (0.3, 0.2, 0.2), (0.2, 0.1, 0.8), (0.4, 0.2, 0.2), (0.4, 0.3, 0.9), (-0.3, 0.3, 0.9), (0.8, -0.1, 0.3),
    (-0.8, -0.7, 0.3), (-0.2, 0.1, 0.3), (0.1, -0.2, 0.3), (-0.4, 0.3, 0.3), (0.3, -0.1, 0.3), (-0.3, -0.3, -0.2),
    (-0.6, -0.2, -0.3), (-0.7, -0.0, -0.1)], np.float64)
OUTLIER_LMKS = np.array([(1.0, 2.3, 0.4), (0.3, 0.3, 0.4), (1.5, 1.3, 0.4)], np.float64)
OUTLIER_KPTS = np.array([(100, 23), (234, 223), (400, 543)], np.float32)


def pose(t):
    m = np.eye(4)
    m[:3, 3] = t
    return m


def quat(R):
    w = 0.5 * np.sqrt(max(1.0 + np.trace(R), 0.0))
    return np.array([w, (R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w)])


def pnp_scene():
    """bearings, world points, expected pose (3x4), mean focal length of the original left camera"""
    L = P.load_camera_params(os.path.join(G, "sensorLeft.yaml"))
    R = P.load_camera_params(os.path.join(G, "sensorRight.yaml"))
    L.body_pose_cam[:] = pose([0.0, 0.0, -2.0]).reshape(-1)
    R.body_pose_cam[:] = pose([1.0, 0.0, -2.0]).reshape(-1)
    cam = O.Camera(L, R)
    P1 = np.array(cam.rect.P1, np.float64).reshape(3, 4)
    R1 = np.array(cam.rect.R1, np.float64).reshape(3, 3)
    K = P1[:, :3]
    # body_Pose_left_cam_rect = body_Pose_cam * (R1^T, 0) (StereoCamera.cpp); R1 = I for this rig
    assert np.allclose(R1, np.eye(3), atol=1e-12)
    W_T_cam = pose([0.0, 0.0, -2.0])
    # StereoCamera::project -> gtsam::StereoCamera(body_Pose_left_cam_rect, K): u = fx X/Z + cx (float32 keypoints)
    pc = INLIER_LMKS - W_T_cam[:3, 3]
    uv = (K @ (pc / pc[:, 2:3]).T).T[:, :2].astype(np.float32)

    def bearing(kp):  # UndistorterRectifier::getBearingVectorFromUndistortedKeypoint (K^-1 [u v 1], normalised)
        h = np.concatenate([kp.astype(np.float64), np.ones((len(kp), 1))], 1)
        v = (np.linalg.inv(K) @ h.T).T
        return v / np.linalg.norm(v, axis=1, keepdims=True)

    f = np.concatenate([bearing(uv), bearing(OUTLIER_KPTS)])
    pw = np.concatenate([INLIER_LMKS, OUTLIER_LMKS])
    return f, pw, W_T_cam[:3, :], 0.5 * (L.intrinsics[0] + L.intrinsics[1])


def test_pnp_tracking_reference_scene():
    f, pw, expected, focal = pnp_scene()
    tp = P.default_frontend_params().tracker
    tp.ransac_randomize = 0
    pp = abi.pnp_params_default()
    pp.min_pnp_inliers = 10
    pp.ransac_threshold_pnp = 0.5
    for policy in (0, 1):   # both std::uniform_int_distribution implementations (ransac_rng_policy)
        tp.ransac_rng_policy = policy
        r = O.pnp(f, pw, focal, tp, pp)
        assert r["success"] and r["status"] == abi.TRACKING_VALID
        assert r["n_inliers"] == len(INLIER_LMKS) and list(r["inliers"]) == list(range(len(INLIER_LMKS)))
        tol = 0.00001
        assert np.all(np.abs(r["pose"][:, 3] - expected[:, 3]) < tol), r["pose"]
        assert np.all(np.abs(quat(r["pose"][:, :3]) - quat(expected[:, :3])) < tol)


def test_epnp_exact_data_all_points_and_samples():
    """absolute_pose::epnp on exact correspondences: the pose is recovered to 1e-9 from all 22 points and from 6-point
    samples, for rotated / translated cameras and for points behind the sign convention (bearing z < 0)."""
    rng = np.random.default_rng(5)
    for trial in range(20):
        w = rng.normal(size=3) * 0.4
        th = np.linalg.norm(w)
        k = w / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        Rwc = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        t = np.array([0.2, -0.1, -3.0]) + rng.normal(size=3) * 0.2
        pw = INLIER_LMKS + rng.normal(size=INLIER_LMKS.shape) * 0.05
        pc = (Rwc.T @ (pw - t).T).T
        f = pc / np.linalg.norm(pc, axis=1, keepdims=True)
        for idx in (np.arange(len(pw)), rng.permutation(len(pw))[:6], rng.permutation(len(pw))[:6]):
            m = O.epnp(f, pw, idx)
            assert np.allclose(m[:, :3], Rwc, atol=1e-7) and np.allclose(m[:, 3], t, atol=1e-7), (trial, idx)


def test_pnp_degenerate_inputs():
    tp = P.default_frontend_params().tracker
    tp.ransac_randomize = 0
    pp = abi.pnp_params_default()
    r = O.pnp(np.zeros((0, 3)), np.zeros((0, 3)), 458.0, tp, pp)     # "No 2D-3D correspondences found"
    assert not r["success"] and r["status"] == abi.TRACKING_FEW_MATCHES and r["n_inliers"] == 0
    assert np.array_equal(r["pose"], np.eye(4)[:3])
    f, pw, _, focal = pnp_scene()
    r = O.pnp(f[:5], pw[:5], focal, tp, pp)                            # fewer than the sample size of 6
    assert not r["success"] and r["n_inliers"] == 0
    pp.min_pnp_inliers = 30                                            # more than the scene holds
    r = O.pnp(f, pw, focal, tp, pp)
    assert r["success"] and r["status"] == abi.TRACKING_FEW_MATCHES and r["n_inliers"] == 22


# ---------------------------------------------------------------------------------------------------------------
# pnp_algorithm: 1 (Kneip's P3P, params/KinectAzure)
# ---------------------------------------------------------------------------------------------------------------
def test_quartic_roots_closed_form():
    """math::o4_roots (Ferrari) with the deterministic complex roots: quartics with four real roots, two, none
    (real parts of the complex pairs), against numpy.roots"""
    rng = np.random.default_rng(3)
    for trial in range(200):
        kind = trial % 3
        if kind == 0:
            r = np.sort(rng.uniform(-1, 1, 4))
            coef = np.poly(r) * rng.uniform(0.5, 3.0)
        elif kind == 1:
            a, b = rng.uniform(-1, 1, 2)
            c = complex(rng.uniform(-1, 1), rng.uniform(0.1, 1))
            coef = np.real(np.poly([a, b, c, np.conj(c)])) * rng.uniform(0.5, 3.0)
        else:
            c1 = complex(rng.uniform(-1, 1), rng.uniform(0.1, 1))
            c2 = complex(rng.uniform(-1, 1), rng.uniform(0.1, 1))
            coef = np.real(np.poly([c1, np.conj(c1), c2, np.conj(c2)])) * rng.uniform(0.5, 3.0)
        got = np.sort(O.quartic_roots(coef))
        exp = np.sort(np.real(np.roots(coef)))
        assert np.allclose(got, exp, atol=2e-6), (trial, got, exp)


def test_p3p_kneip_contains_the_true_pose():
    """absolute_pose::p3p_kneip on exact correspondences: one of the (up to four) solutions is the true camera pose,
    and AbsolutePoseSacProblem's fourth point picks it"""
    rng = np.random.default_rng(8)
    tp = P.default_frontend_params().tracker
    tp.ransac_randomize = 0
    hits = 0
    for trial in range(60):
        w = rng.normal(size=3) * 0.5
        th = np.linalg.norm(w)
        k = w / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        Rwc = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        t = rng.normal(size=3)
        pc = np.stack([rng.uniform(-2, 2, 30), rng.uniform(-1.5, 1.5, 30), rng.uniform(1.5, 8, 30)], 1)
        pw = (Rwc @ pc.T).T + t
        f = pc / np.linalg.norm(pc, axis=1, keepdims=True)
        sols = O.p3p_kneip(f, pw, [0, 1, 2])
        assert len(sols) == 4
        err = [np.abs(s[:, :3] - Rwc).max() + np.abs(s[:, 3] - t).max() for s in sols if np.all(np.isfinite(s))]
        assert min(err) < 1e-6, (trial, err)
        hits += 1
        pp = abi.pnp_params_default()
        pp.pnp_algorithm = abi.PNP_KNEIP_P3P
        r = O.pnp(f, pw, 458.0, tp, pp)
        assert r["success"] and r["n_inliers"] == 30
        assert np.allclose(r["pose"][:, :3], Rwc, atol=1e-6) and np.allclose(r["pose"][:, 3], t, atol=1e-6)
    assert hits == 60


def test_pnp_tracking_reference_scene_kneip():
    """the PnPTracking scene with pnp_algorithm 1: same assertions (22 inliers of 25, pose within 1e-5)"""
    f, pw, expected, focal = pnp_scene()
    tp = P.default_frontend_params().tracker
    tp.ransac_randomize = 0
    pp = abi.pnp_params_default()
    pp.pnp_algorithm = abi.PNP_KNEIP_P3P
    pp.min_pnp_inliers = 10
    pp.ransac_threshold_pnp = 0.5
    r = O.pnp(f, pw, focal, tp, pp)
    assert r["success"] and r["status"] == abi.TRACKING_VALID and r["n_inliers"] == len(INLIER_LMKS)
    assert np.all(np.abs(r["pose"][:, 3] - expected[:, 3]) < 1e-5)
    assert np.all(np.abs(quat(r["pose"][:, :3]) - quat(expected[:, :3])) < 1e-5)
