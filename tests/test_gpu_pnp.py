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


# ---------------------------------------------------------------------------------------------------------------
# use_pnp_tracking inside the step: Tracker::updateMap + outlierRejectionPnP on keyframes
# (StereoVisionImuFrontend.cpp:389-399, RgbdVisionImuFrontend.cpp:328-341)
# ---------------------------------------------------------------------------------------------------------------
def _seq():
    z = np.load(os.path.join(G, "micro_euroc_f10_18.npz"))
    return dict(lefts=z["lefts"], rights=z["rights"], ts=z["timestamps"], body_R=z["body_R"])


def _cam_rotations(L, R, body_R):
    cam = O.Camera(L, R)
    TL = np.array(L.body_pose_cam).reshape(4, 4)
    b_R_c = TL[:3, :3] @ np.array(cam.rect.R1).reshape(3, 3).T
    return [b_R_c.T @ Rb @ b_R_c for Rb in body_R]


PNP_KEYS = ("tracking_status_pnp", "nr_pnp_inliers")


@pytest.mark.parametrize("alg,streams", [(abi.PNP_EPNP, 1), (abi.PNP_KNEIP_P3P, 2)])
def test_frontend_step_with_pnp_tracking(alg, streams):
    """the shipped Euroc parameters with use_pnp_tracking: 1; the back-end's map is played by the 3-D points of the
    first keyframe (world = its camera frame), handed over after frame 0 and replaced after frame 3 (a map with
    duplicated and unsorted ids); every later keyframe's kfTracking_status_pnp_ / W_T_k_pnp_ / inlier count equals
    the oracle front-end's, bit for bit, next to the unchanged rest of the frame."""
    seq = _seq()
    L, R = P.load_camera_params(os.path.join(G, "sensorLeft.yaml")), P.load_camera_params(os.path.join(G, "sensorRight.yaml"))
    p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=None)
    p.use_pnp_tracking = 1
    p.pnp.pnp_algorithm = alg
    p.pnp.min_pnp_inliers = 10
    camR = _cam_rotations(L, R, seq["body_R"])
    fe = [O.Frontend(L, R, p) for _ in range(streams)]
    c = F.Context(L, R, p, batch=streams)
    valid_seen = 0
    try:
        kf = [0] * streams
        for i in range(8):
            idx = [i if s == 0 else 8 - i for s in range(streams)]
            Rs = [camR[kf[s]].T @ camR[idx[s]] for s in range(streams)]
            ts = [int(seq["ts"][i])] * streams
            hl = np.stack([seq["lefts"][j] for j in idx])
            hr = np.stack([seq["rights"][j] for j in idx])
            c.step_host(hl, hr, c.make_inputs(ts, Rs, [1] * streams))   # every frame a keyframe
            for s in range(streams):
                exp = fe[s].process(seq["lefts"][idx[s]], seq["rights"][idx[s]], ts[s], Rs[s], True)
                got = c.get_output(s)
                for k in ("n_keypoints", "is_keyframe", "n_tracked", "tracking_status_mono", "tracking_status_stereo") + PNP_KEYS:
                    assert got[k] == exp[k], (i, s, k, got[k], exp[k])
                assert np.array_equal(got["W_T_k_pnp"], exp["W_T_k_pnp"]), (i, s)
                assert np.array_equal(got["landmarks"], exp["landmarks"]) and np.array_equal(got["keypoints"], exp["keypoints"])
                if i >= 1:
                    valid_seen += int(got["tracking_status_pnp"] == abi.TRACKING_VALID)
                if exp["is_keyframe"]:
                    kf[s] = idx[s]
                if i in (0, 3):   # Tracker::updateMap from the "back-end"
                    ok = (exp["right_status"] == abi.KP_VALID) & (exp["landmarks"] != -1)
                    ids, pts = exp["landmarks"][ok], exp["keypoints_3d"][ok]
                    if i == 3:   # unsorted, with duplicated ids (the later entry wins, as in std::unordered_map::operator[])
                        ids = np.concatenate([ids[::-1], ids[:5]])
                        pts = np.concatenate([pts[::-1], pts[:5] + 0.001])
                    fe[s].update_map(ids, pts)
                    c.update_map(s, ids, pts)
        assert valid_seen >= 2 * streams   # the map really produces poses (not just FEW_MATCHES)
    finally:
        c.close()


@pytest.mark.parametrize("alg", [abi.PNP_KNEIP_P3P, abi.PNP_EPNP])
def test_rgbd_frontend_with_pnp_tracking_kinect_azure_style(alg):
    """params/KinectAzure ships use_pnp_tracking: 1 with pnp_algorithm: 1 (KneipP3P) on the RGBD front-end:
    RgbdVisionImuFrontend::handleKeyframe's outlierRejectionPnP (:328-341) against the oracle; useRANSAC: 0 reports
    DISABLED (:342-346).  As upstream, the "bearing vectors" handed to OpenGV are keypoints_3d_ (3-D points, not unit
    vectors: Tracker.cpp:1100): Kneip's P3P then has no real solution once the depths exceed 1 m (cos beta > 1) and the
    keyframe reports FEW_MATCHES with no inliers, EPnP (scale free) still recovers a pose -- both reproduced."""
    from test_gpu_parity import _synthetic_depth
    seq = _seq()
    L = P.load_camera_params(os.path.join(G, "sensorLeft.yaml"))
    h, w = seq["lefts"][0].shape
    TL = np.array(L.body_pose_cam).reshape(4, 4)
    camR = [TL[:3, :3].T @ Rb @ TL[:3, :3] for Rb in seq["body_R"]]
    for use_ransac in (1, 0):
        p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=use_ransac)
        p.detector.max_features_per_frame = 200
        p.use_pnp_tracking = 1
        p.pnp.pnp_algorithm = alg
        p.pnp.min_pnp_inliers = 20
        dp = abi.depth_params_default(abi.DEPTH_U16)
        dp.virtual_baseline, dp.min_depth, dp.depth_to_meters = 0.05, 0.3, 0.001
        fe = O.Frontend(L, L, p, depth=dp)
        c = F.Context(L, L, p, batch=1, frontend_type=abi.FRONTEND_RGBD, depth=dp)
        try:
            kf0 = 0
            statuses = []
            for i in range(6):
                depth = _synthetic_depth(h, w, i, abi.DEPTH_U16, seed=i)
                Rk = camR[kf0].T @ camR[i]
                ts = int(seq["ts"][i])
                c.step_host(seq["lefts"][i][None], depth[None], c.make_inputs([ts], [Rk], [1]))
                exp = fe.process(seq["lefts"][i], depth, ts, Rk, True)
                got = c.get_output(0)
                for k in ("n_keypoints", "is_keyframe", "tracking_status_stereo") + PNP_KEYS:
                    assert got[k] == exp[k], (use_ransac, i, k, got[k], exp[k])
                assert np.array_equal(got["W_T_k_pnp"], exp["W_T_k_pnp"]), (use_ransac, i)
                statuses.append(got["tracking_status_pnp"])
                if exp["is_keyframe"]:
                    kf0 = i
                if i == 0:
                    ok = (exp["right_status"] == abi.KP_VALID) & (exp["landmarks"] != -1)
                    fe.update_map(exp["landmarks"][ok], exp["keypoints_3d"][ok])
                    c.update_map(0, exp["landmarks"][ok], exp["keypoints_3d"][ok])
            if use_ransac and alg == abi.PNP_EPNP:
                assert abi.TRACKING_VALID in statuses[1:]
            elif use_ransac:
                assert all(st == abi.TRACKING_FEW_MATCHES for st in statuses[1:])
            else:
                assert all(st == abi.TRACKING_DISABLED for st in statuses[1:])
        finally:
            c.close()
