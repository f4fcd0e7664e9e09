"""Tracker::pnp on the device (kvfe_pnp, k_pnp.inl) against the CPU oracle: the reference's PnPTracking scene
(tests/testTracker.cpp:1613-1800) with upstream's assertions, and randomised scenes compared field by field --
status, success flag, iteration count, inlier list and the pose BIT-EXACT (same float64 operations in the same order)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from kimera_vio_amd import _abi as abi
from kimera_vio_amd import frontend as F
from kimera_vio_amd import params as P
from test_oracle_pnp import INLIER_LMKS, pnp_scene, quat

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _ctx(**tracker):
    L = P.load_camera_params(os.path.join(G, "sensorLeft.yaml"))
    R = P.load_camera_params(os.path.join(G, "sensorRight.yaml"))
    p = P.default_frontend_params()
    p.tracker.ransac_randomize = 0
    for k, v in tracker.items():
        setattr(p.tracker, k, v)
    return F.Context(L, R, p), p, 0.5 * (L.intrinsics[0] + L.intrinsics[1])


def _same(a, b):
    assert a["success"] == b["success"] and a["status"] == b["status"], (a, b)
    assert a["iterations"] == b["iterations"] and a["n_inliers"] == b["n_inliers"], (a["iterations"], b["iterations"])
    assert np.array_equal(a["inliers"], b["inliers"])
    assert np.array_equal(a["pose"], b["pose"]), np.abs(a["pose"] - b["pose"]).max()


def test_pnp_tracking_reference_scene():
    f, pw, expected, focal = pnp_scene()
    for policy in (0, 1):
        c, p, cf = _ctx(ransac_rng_policy=policy)
        try:
            assert cf == focal
            pp = abi.pnp_params_default()
            pp.min_pnp_inliers = 10
            pp.ransac_threshold_pnp = 0.5
            got = c.pnp(f, pw, pp)
            assert got["success"] and got["status"] == abi.TRACKING_VALID and got["n_inliers"] == len(INLIER_LMKS)
            tol = 0.00001
            assert np.all(np.abs(got["pose"][:, 3] - expected[:, 3]) < tol)
            assert np.all(np.abs(quat(got["pose"][:, :3]) - quat(expected[:, :3])) < tol)
            _same(got, O.pnp(f, pw, focal, p.tracker, pp))
        finally:
            c.close()


def test_pnp_random_scenes_bit_exact():
    """40 scenes: 20-400 correspondences, 0-60 % outliers, pixel noise; thresholds 0.5-3 px; both sampler policies;
    few iterations allowed (ransac_max_iterations 3) and many."""
    rng = np.random.default_rng(11)
    for trial in range(40):
        n = int(rng.integers(20, 400))
        frac_out = float(rng.choice([0.0, 0.1, 0.3, 0.6]))
        w = rng.normal(size=3) * 0.3
        th = np.linalg.norm(w)
        k = w / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        Rwc = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        t = rng.normal(size=3)
        pc = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(1.5, 8, n)], 1)
        pw = (Rwc @ pc.T).T + t
        noise = rng.normal(size=(n, 2)) * float(rng.choice([0.0, 0.2, 0.5])) / 458.0
        fb = np.concatenate([pc[:, :2] / pc[:, 2:3] + noise, np.ones((n, 1))], 1)
        fb /= np.linalg.norm(fb, axis=1, keepdims=True)
        bad = rng.random(n) < frac_out
        pw[bad] += rng.normal(size=(int(bad.sum()), 3))
        c, p, focal = _ctx(ransac_rng_policy=int(trial % 2), ransac_max_iterations=int(rng.choice([3, 100, 500])),
                           ransac_probability=float(rng.choice([0.95, 0.995])))
        try:
            pp = abi.pnp_params_default()
            pp.ransac_threshold_pnp = float(rng.choice([0.5, 1.0, 3.0]))
            pp.min_pnp_inliers = int(rng.choice([5, 20, 200]))
            got = c.pnp(fb, pw, pp)
            exp = O.pnp(fb, pw, focal, p.tracker, pp)
            _same(got, exp)
            if frac_out <= 0.3 and p.tracker.ransac_max_iterations >= 100:
                assert got["success"] and got["n_inliers"] >= 0.5 * (~bad).sum()
        finally:
            c.close()


def test_pnp_kneip_p3p_bit_exact():
    """pnp_algorithm 1 (KneipP3P, params/KinectAzure): the reference scene and 30 random scenes, field by field and
    the pose bit-exact (the closed-form quartic uses only +, -, *, / and sqrt on both sides)"""
    f, pw, expected, focal = pnp_scene()
    c, p, _ = _ctx()
    try:
        pp = abi.pnp_params_default()
        pp.pnp_algorithm = abi.PNP_KNEIP_P3P
        pp.min_pnp_inliers = 10
        pp.ransac_threshold_pnp = 0.5
        got = c.pnp(f, pw, pp)
        assert got["success"] and got["n_inliers"] == len(INLIER_LMKS)
        assert np.all(np.abs(got["pose"][:, 3] - expected[:, 3]) < 1e-5)
        _same(got, O.pnp(f, pw, focal, p.tracker, pp))
    finally:
        c.close()
    rng = np.random.default_rng(21)
    for trial in range(30):
        n = int(rng.integers(8, 300))
        w = rng.normal(size=3) * 0.3
        th = np.linalg.norm(w)
        k = w / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        Rwc = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        t = rng.normal(size=3)
        pc = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(1.5, 8, n)], 1)
        pw = (Rwc @ pc.T).T + t
        noise = rng.normal(size=(n, 2)) * float(rng.choice([0.0, 0.3])) / 458.0
        fb = np.concatenate([pc[:, :2] / pc[:, 2:3] + noise, np.ones((n, 1))], 1)
        fb /= np.linalg.norm(fb, axis=1, keepdims=True)
        bad = rng.random(n) < float(rng.choice([0.0, 0.2, 0.5]))
        pw[bad] += rng.normal(size=(int(bad.sum()), 3))
        if trial % 7 == 0:
            pw[:3] = pw[0] + np.outer([0.0, 1.0, 2.0], [0.1, 0.2, 0.3])   # collinear triples: computeModel fails
        c, p, focal = _ctx(ransac_rng_policy=int(trial % 2), ransac_max_iterations=int(rng.choice([5, 100])))
        try:
            pp = abi.pnp_params_default()
            pp.pnp_algorithm = abi.PNP_KNEIP_P3P
            pp.ransac_threshold_pnp = float(rng.choice([0.5, 2.0]))
            _same(c.pnp(fb, pw, pp), O.pnp(fb, pw, focal, p.tracker, pp))
        finally:
            c.close()


def test_pnp_degenerate_and_unsupported():
    c, p, focal = _ctx()
    try:
        pp = abi.pnp_params_default()
        r = c.pnp(np.zeros((0, 3)), np.zeros((0, 3)), pp)
        assert not r["success"] and r["status"] == abi.TRACKING_FEW_MATCHES and np.array_equal(r["pose"], np.eye(4)[:3])
        f, pw, _, _ = pnp_scene()
        _same(c.pnp(f[:5], pw[:5], pp), O.pnp(f[:5], pw[:5], focal, p.tracker, pp))   # below the sample size
        _same(c.pnp(f[:6], pw[:6], pp), O.pnp(f[:6], pw[:6], focal, p.tracker, pp))   # exactly the sample size
        for alg in (abi.PNP_KNEIP_P2P, abi.PNP_GAO_P3P, abi.PNP_UPNP, abi.PNP_UP3P,
                    abi.PNP_NONLINEAR, abi.PNP_MLPNP):
            pp.pnp_algorithm = alg
            with pytest.raises(F.KvfeError) as e:
                c.pnp(f, pw, pp)
            assert e.value.status == abi.KVFE_ERR_UNSUPPORTED
        pp = abi.pnp_params_default()
        pp.optimize_2d3d_pose_from_inliers = 1
        with pytest.raises(F.KvfeError) as e:
            c.pnp(f, pw, pp)
        assert e.value.status == abi.KVFE_ERR_UNSUPPORTED
    finally:
        c.close()
