"""GPU parity tests: the HIP path (through the C ABI of libkvfe.so) against the CPU oracle on the
same inputs.  Integer / index / status outputs must be bit-exact; float outputs are produced by
the same IEEE operations in the same order on both sides, so they are asserted bit-exact too
(tolerance 0) except where a tolerance is written explicitly in the test.

Run on the MI355X box:  python -m pytest tests -m gpu -x -q
"""
import os

import numpy as np
import pytest
from PIL import Image

import oracle_lib as O
from kimera_vio_amd import _abi as abi
from kimera_vio_amd import frontend as F
from kimera_vio_amd import params as P
from kimera_vio_amd import synth

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gray(name):
    return np.array(Image.open(os.path.join(G, name)).convert("L"))


def euroc_cams():
    return (P.load_camera_params(os.path.join(G, "sensorLeft.yaml")),
            P.load_camera_params(os.path.join(G, "sensorRight.yaml")))


def euroc_params(**det):
    p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=0)
    for k, v in det.items():
        setattr(p.detector, k, v)
    return p


@pytest.fixture(scope="module")
def seq():
    z = np.load(os.path.join(G, "micro_euroc_f10_18.npz"))
    return dict(lefts=z["lefts"], rights=z["rights"], ts=z["timestamps"], body_R=z["body_R"])


@pytest.fixture(scope="module")
def ctx():
    L, R = euroc_cams()
    c = F.Context(L, R, euroc_params())
    yield c
    c.close()


@pytest.fixture(scope="module")
def ocam():
    L, R = euroc_cams()
    return O.Camera(L, R)


# ---------------------------------------------------------------------------------------------
def test_native_library_is_loaded():
    from kimera_vio_amd import lib
    assert os.path.exists(lib.SO_PATH)
    assert lib.load().kvfe_version().decode().startswith("libkvfe")


def test_rectify_bit_exact(ctx, ocam, seq):
    """K1 vs cv::remap restatement: every pixel identical, both cameras."""
    for cam, img in ((0, gray("left_img_0.png")), (1, gray("right_img_0.png")),
                     (0, seq["lefts"][3]), (1, seq["rights"][3])):
        got = ctx.undistort_rectify_image(cam, img)
        exp = ocam.rectify_image(cam, img)
        assert np.array_equal(got, exp)
    rng = np.random.RandomState(0)
    noise = rng.randint(0, 256, size=(480, 752)).astype(np.uint8)
    assert np.array_equal(ctx.undistort_rectify_image(0, noise), ocam.rectify_image(0, noise))


def test_undistort_keypoints_and_versors_bit_exact(ctx, ocam):
    rng = np.random.RandomState(1)
    pts = np.stack([rng.uniform(-5, 757, 500), rng.uniform(-5, 485, 500)], 1).astype(np.float32)
    for cam in (0, 1):
        for useR, useP in ((1, 1), (1, 0), (0, 0)):
            got = ctx.undistort_rectify_keypoints(cam, pts, useR, useP)
            exp = ocam.undistort_keypoints(cam, pts, useR, useP)
            assert np.array_equal(got, exp)
        assert np.array_equal(ctx.get_bearing_vectors(cam, pts), ocam.bearing_vectors(cam, pts))


def test_predict_sparse_flow_bit_exact(ctx):
    L, _ = euroc_cams()
    rng = np.random.RandomState(2)
    pts = np.stack([rng.uniform(0, 752, 300), rng.uniform(0, 480, 300)], 1).astype(np.float32)
    for ang in (0.0, 1e-6, 0.002, 0.02, 0.3):
        R = synth.rot_from_axis_angle([0.3, -0.8, 0.5], ang)
        got = ctx.predict_sparse_flow(pts, R)
        exp = O.predict_sparse_flow(abi.FLOW_ROTATIONAL, L, pts, R)
        assert np.array_equal(got, exp)


def test_raw_gftt_bit_exact_and_kat_393():
    """K2: cv::GFTTDetector::detect restatement; includes the reference KAT (393 corners,
    tests/testFeatureDetector.cpp:25-50)."""
    img = gray("left_fisheye_img_0.png")
    L, R = euroc_cams()
    d = P.load_detector_params(os.path.join(G, "ForFeatureDetector", "frontendParams-noNMS.yaml"))
    p = euroc_params()
    p.detector = d
    c = F.Context(L, R, p)
    try:
        got = c.raw_feature_detection(img)
        exp, _ = O.good_features_to_track(img, d.max_nr_keypoints_before_anms, d.quality_level,
                                          d.min_distance, 3)
        assert len(got) == 393
        assert np.array_equal(got, exp)
        rng = np.random.RandomState(3)
        mask = (rng.uniform(size=img.shape) > 0.3).astype(np.uint8) * 255
        mask[100:200, 300:500] = 0
        got = c.raw_feature_detection(img, mask)
        exp, _ = O.good_features_to_track(img, d.max_nr_keypoints_before_anms, d.quality_level,
                                          d.min_distance, 3, mask=mask)
        assert np.array_equal(got, exp)
        blank = np.full(img.shape, 200, np.uint8)
        assert len(c.raw_feature_detection(blank)) == 0
    finally:
        c.close()


@pytest.mark.parametrize("yaml_name,overrides,expected", [
    ("frontendParams-noNMS.yaml", {}, 393),
    ("frontendParams-noNMS.yaml", {"quality_level": 1e-10}, 400),
    ("frontendParams-NMS-TopN.yaml", {}, 300),
    ("frontendParams-NMS-Binning.yaml", {}, 20),
    ("frontendParams-NMS-Binning.yaml", {"max_features_per_frame": 200, "quality_level": 1e-10,
                                         "enable_subpixel_corner_refinement": 0}, 200),
    ("frontendParams-NMS-Binning2.yaml", {"quality_level": 1e-10,
                                          "enable_subpixel_corner_refinement": 0}, 140),
    ("frontendParams-NMS-Binning.yaml", {"sortidx_policy": abi.SORTIDX_STABLE}, 20),
])
def test_feature_detection_kats_and_parity(yaml_name, overrides, expected):
    """FeatureDetector::featureDetection on the reference's own KAT configurations
    (tests/testFeatureDetector.cpp): counts match the reference, every keypoint matches the oracle."""
    img = gray("left_fisheye_img_0.png")
    L, R = euroc_cams()
    d = P.load_detector_params(os.path.join(G, "ForFeatureDetector", yaml_name))
    for k, v in overrides.items():
        setattr(d, k, v)
    p = euroc_params()
    p.detector = d
    c = F.Context(L, R, p)
    try:
        none = np.zeros((0, 2), np.float32)
        got = c.feature_detection(img, none, d.max_features_per_frame)
        exp, _ = O.feature_detection(img, none, d.max_features_per_frame, d)
        assert len(got) == expected
        assert np.array_equal(got, exp)
    finally:
        c.close()


def test_feature_detection_with_tracked_mask(ctx, seq):
    """mask = 255 minus cv::circle discs around tracked keypoints (FeatureDetector.cpp:185-203)."""
    img = seq["lefts"][0]
    d = ctx.params.detector
    first, _ = O.feature_detection(img, np.zeros((0, 2), np.float32), 300, d)
    tracked = first[::2] + np.float32(0.37)  # non-integer centres exercise cvRound
    need = 300 - len(tracked)
    got = ctx.feature_detection(img, tracked, need)
    exp, _ = O.feature_detection(img, tracked, need, d)
    assert len(got) > 0
    assert np.array_equal(got, exp)
    # keypoints at the image border exercise the clipped discs
    border = np.array([[0.2, 0.4], [751.4, 479.2], [3.5, 470.5], [748.5, 2.5]], np.float32)
    got = ctx.feature_detection(img, border, 120)
    exp, _ = O.feature_detection(img, border, 120, d)
    assert np.array_equal(got, exp)


def test_corner_subpix_bit_exact(ctx, seq):
    img = seq["lefts"][1]
    pts, _ = O.good_features_to_track(img, 400, 0.001, 10, 3)
    edge = np.array([[2.0, 3.0], [749.0, 477.0], [5.0, 240.0], [375.0, 4.0], [11.5, 11.5]], np.float32)
    pts = np.concatenate([pts, edge])
    got = ctx.corner_subpix(img, pts, 10, -1, 40, 0.001)
    exp = O.corner_subpix(img, pts, 10, -1, 40, 0.001)
    assert np.array_equal(got, exp)
    got = ctx.corner_subpix(img, pts[:50], 5, 1, 10, 0.01)
    exp = O.corner_subpix(img, pts[:50], 5, 1, 10, 0.01)
    assert np.array_equal(got, exp)


def test_lk_bit_exact(ctx, seq):
    """K4: cv::calcOpticalFlowPyrLK restatement (SSE2 accumulation order): positions, status and
    error identical."""
    t = ctx.params.tracker
    for a, b in ((0, 1), (2, 5), (0, 8)):
        prev, cur = seq["lefts"][a], seq["lefts"][b]
        pts, _ = O.good_features_to_track(prev, 300, 0.001, 20, 3)
        pts = O.corner_subpix(prev, pts)
        extra = np.array([[1.0, 1.0], [750.5, 478.5], [10.0, 470.0], [745.0, 8.0], [376.0, 240.0]], np.float32)
        pts = np.concatenate([pts, extra])
        init = pts + np.float32(0.75)
        got, gst, gerr = ctx.calc_optical_flow_pyr_lk(prev, cur, pts, init)
        exp, est, eerr, lvl = O.calc_optical_flow_pyr_lk(prev, cur, pts, init, t.klt_win_size,
                                                         t.klt_max_level, t.klt_max_iter, t.klt_eps)
        assert lvl == 4
        assert np.array_equal(gst, est)
        assert np.array_equal(got, exp)
        assert np.array_equal(gerr, eerr)
        assert est.sum() > 100


@pytest.mark.parametrize("win,max_level,max_iter,eps", [(16, 2, 30, 0.1), (32, 3, 10, 0.01),
                                                        (21, 2, 30, 0.1), (9, 1, 5, 0.03),
                                                        (24, 0, 30, 0.001)])
def test_lk_other_windows_bit_exact(seq, win, max_level, max_iter, eps):
    """The systolic kernel covers windows 16/24/32, every other size takes the generic kernel;
    both must reproduce the oracle exactly, also for large initial errors and border points."""
    L, R = euroc_cams()
    p = euroc_params()
    p.tracker.klt_win_size, p.tracker.klt_max_level = win, max_level
    p.tracker.klt_max_iter, p.tracker.klt_eps = max_iter, eps
    c = F.Context(L, R, p)
    try:
        prev, cur = seq["lefts"][1], seq["lefts"][4]
        pts, _ = O.good_features_to_track(prev, 200, 0.001, 15, 3)
        rng = np.random.default_rng(win)
        extra = np.stack([rng.uniform(-3, 755, 60), rng.uniform(-3, 483, 60)], 1).astype(np.float32)
        pts = np.concatenate([pts, extra])
        init = pts + rng.uniform(-6, 6, pts.shape).astype(np.float32)
        got, gst, gerr = c.calc_optical_flow_pyr_lk(prev, cur, pts, init)
        exp, est, eerr, _ = O.calc_optical_flow_pyr_lk(prev, cur, pts, init, win, max_level, max_iter, eps)
        assert np.array_equal(gst, est)
        assert np.array_equal(got, exp)
        assert np.array_equal(gerr, eerr)
        assert est.sum() > 50
    finally:
        c.close()


def test_stereo_match_849_of_900_and_parity(ocam):
    """tests/testStereoMatcher.cpp:272-388 on the GPU path (default StereoMatchingParams)."""
    L, R = euroc_cams()
    p = euroc_params()
    p.stereo = P.default_frontend_params().stereo
    left = gray("left_img_0.png")
    kps, _ = O.good_features_to_track(left, 100, 0.01, 10, 3)
    rows, cols = left.shape
    count_valid = total = 0
    # the reference test passes fx = 458.654; build a context whose rectified fx is what the
    # matcher sees is not possible through the API, so compare against the oracle at the context's
    # own fx and count validity with the reference's rule.
    c = F.Context(L, R, p)
    try:
        for offset in (-20, -10, -5):
            right = np.zeros_like(left)
            right[:, : cols + offset] = left[:, -offset:]
            acc = np.zeros((0, 2), np.float32)
            for t in range(2):
                add = kps if t == 0 else np.round(kps)
                acc = np.concatenate([acc, add.astype(np.float32)])
                st = np.zeros(len(acc), np.uint8)
                rxy, rst, sc = c.get_right_keypoints_rectified(left, right, acc, st)
                exy, est, esc = ocam.get_right_keypoints_rectified(left, right, acc, st, p.stereo)
                assert np.array_equal(rxy, exy) and np.array_equal(rst, est) and np.array_equal(sc, esc)
                for i in range(len(acc)):
                    total += 1
                    y_left = float(acc[i, 1])
                    x_exp = float(acc[i, 0]) + offset
                    if y_left <= 7 or y_left + 7 >= rows:
                        assert rst[i] == abi.KP_NO_RIGHT_RECT
                    elif x_exp >= 50 and x_exp + 50 < cols:
                        assert rst[i] == abi.KP_VALID
                        assert abs(x_exp - float(rxy[i, 0])) < 0.5
                        assert abs(float(acc[i, 1]) - float(rxy[i, 1])) < 0.5
                        count_valid += 1
    finally:
        c.close()
    assert total == 900 and count_valid == 849


@pytest.mark.parametrize("subpix", [0, 1])
def test_sparse_stereo_bit_exact(ocam, subpix):
    L, R = euroc_cams()
    p = euroc_params()
    p.stereo.subpixel_refinement = subpix
    left, right = gray("left_img_0.png"), gray("right_img_0.png")
    kps, _ = O.good_features_to_track(left, 300, 0.001, 10, 3)
    kps = np.concatenate([O.corner_subpix(left, kps),
                          np.array([[3.0, 3.0], [748.0, 476.0], [400.0, 2.0], [-4.0, 100.0]], np.float32)])
    c = F.Context(L, R, p)
    try:
        got = c.sparse_stereo_reconstruction(left, right, kps, want_images=True)
        exp = ocam.sparse_stereo(left, right, kps, p.stereo, want_images=True)
        for k in exp:
            assert np.array_equal(got[k], exp[k]), k
        assert (got["right_status"] == abi.KP_VALID).sum() > 100
    finally:
        c.close()


# ---------------------------------------------------------------------------------------------
def _kf_rotations(body_R, cam: O.Camera):
    """camLrect_R_body used by nominalSpinStereo (StereoVisionImuFrontend.cpp:143-150)."""
    TL = np.array(cam.left.body_pose_cam).reshape(4, 4)
    R1 = np.array(cam.rect.R1).reshape(3, 3)
    body_R_cam = TL[:3, :3] @ R1.T
    return [body_R_cam.T @ R @ body_R_cam for R in body_R]


def _run_sequence(oracle_fe, gpu_ctx, seq, stream_of=lambda s, i: i, force_kf=False, n=None):
    """drives oracle and GPU front-ends in lock-step and asserts identical outputs per frame"""
    B = gpu_ctx.batch
    n = n or len(seq["ts"])
    kf_index = [0] * B
    camR = seq["camR"]
    kinds = []
    for i in range(n):
        idx = [stream_of(s, i) for s in range(B)]
        Rs = [camR[kf_index[s]].T @ camR[idx[s]] for s in range(B)]
        ts = [int(seq["ts"][i]) for _ in range(B)]
        lefts = np.stack([seq["lefts"][j] for j in idx])
        rights = np.stack([seq["rights"][j] for j in idx])
        inputs = gpu_ctx.make_inputs(ts, Rs, [int(force_kf)] * B)
        gpu_ctx.step_host(lefts, rights, inputs)
        for s in range(B):
            exp = oracle_fe[s].process(lefts[s], rights[s], ts[s], Rs[s], force_kf)
            got = gpu_ctx.get_output(s)
            for k in ("n_keypoints", "is_keyframe", "n_tracked", "n_detected", "n_measurements", "frame_id"):
                assert got[k] == exp[k], (i, s, k, got[k], exp[k])
            keys = ["landmarks", "landmarks_age", "keypoints", "versors"]
            if exp["has_stereo"] and exp["is_keyframe"]:
                keys += ["left_rect_xy", "left_status", "right_rect_xy", "right_status", "depth",
                         "right_xy", "keypoints_3d", "meas_landmark"]
            for k in keys:
                assert np.array_equal(got[k], exp[k]), (i, s, k)
            if exp["is_keyframe"]:
                assert np.array_equal(got["meas_uL_uR_v"], exp["meas_uL_uR_v"], equal_nan=True), (i, s)
                kf_index[s] = idx[s]
            # TrackerStatusSummary / DebugTrackerInfo (geometric outlier rejection)
            for k in ("tracking_status_mono", "tracking_status_stereo", "nr_mono_putatives",
                      "nr_mono_inliers", "nr_stereo_putatives", "nr_stereo_inliers"):
                assert got[k] == exp[k], (i, s, k, got[k], exp[k])
            for k in ("lkf_T_k_mono", "lkf_T_k_stereo", "info_mat_stereo_translation"):
                assert np.array_equal(got[k], exp[k]), (i, s, k, got[k], exp[k])
            kinds.append((i, s, exp["is_keyframe"], exp["n_tracked"], exp["n_detected"]))
    return kinds


def test_frontend_sequence_single_stream(seq, ocam):
    """C2: one EuRoC stream through processFirstStereoFrame / processStereoFrame (useRANSAC = 0);
    landmark ids, ages, keypoints, versors, stereo statuses, depths and measurements identical to
    the oracle on every frame; keyframes every 0.2 s as in the reference."""
    seq = dict(seq)
    seq["camR"] = _kf_rotations(seq["body_R"], ocam)
    L, R = euroc_cams()
    p = euroc_params()
    fe = [O.Frontend(L, R, p)]
    c = F.Context(L, R, p, batch=1)
    try:
        kinds = _run_sequence(fe, c, seq)
    finally:
        c.close()
    kfs = [k[2] for k in kinds]
    assert kfs == [1, 0, 0, 0, 1, 0, 0, 0, 1]
    assert all(k[3] > 100 for k in kinds[1:])


@pytest.mark.parametrize("B,groups", [(4, 1), (4, 2), (5, 3)])
def test_frontend_sequence_batched_streams(seq, ocam, B, groups):
    """C3-style: independent streams in one context (different frame offsets / directions), each
    with its own landmark-id counter starting at 0, all identical to per-stream oracles -- with the
    batch on one HIP stream and split into stream groups (uneven split included)."""
    seq = dict(seq)
    seq["camR"] = _kf_rotations(seq["body_R"], ocam)
    L, R = euroc_cams()
    p = euroc_params(max_features_per_frame=200)
    fe = [O.Frontend(L, R, p) for _ in range(B)]
    c = F.Context(L, R, p, batch=B, stream_groups=groups)

    def stream_of(s, i):
        return [i, 8 - i, min(i + 2, 8), (3 * i) % 9, (5 * i + 1) % 9][s]

    try:
        _run_sequence(fe, c, seq, stream_of=stream_of, n=7)
    finally:
        c.close()


def test_frontend_force_keyframe_every_frame(seq, ocam):
    """KF-every-frame mode used by the benchmark (Frame::isKeyframe_ forced by the caller)."""
    seq = dict(seq)
    seq["camR"] = _kf_rotations(seq["body_R"], ocam)
    L, R = euroc_cams()
    p = euroc_params()
    fe = [O.Frontend(L, R, p)]
    c = F.Context(L, R, p, batch=1)
    try:
        kinds = _run_sequence(fe, c, seq, force_kf=True, n=5)
    finally:
        c.close()
    assert all(k[2] == 1 for k in kinds)


def test_frontend_lost_tracks_redetects(ocam):
    """StereoVisionImuFrontend.cpp:313-323: when LK loses every point the frame re-detects and is
    not a keyframe (tests/testStereoVisionImuFrontend.cpp testLostFeatureTrack)."""
    L, R = euroc_cams()
    p = euroc_params()
    rng = np.random.RandomState(5)
    a = gray("left_img_0.png")
    b = gray("right_img_0.png")
    flat = np.full_like(a, 127)
    fe = O.Frontend(L, R, p)
    c = F.Context(L, R, p, batch=1)
    try:
        frames = [(a, b), (flat, flat), (a, b), (a, b)]
        for i, (l, r) in enumerate(frames):
            ts = 1000 + i * 50_000_000
            c.step_host(l[None], r[None], c.make_inputs([ts]))
            exp = fe.process(l, r, ts)
            got = c.get_output(0)
            for k in ("n_keypoints", "is_keyframe", "n_tracked", "n_detected"):
                assert got[k] == exp[k], (i, k, got[k], exp[k])
            for k in ("landmarks", "landmarks_age", "keypoints"):
                assert np.array_equal(got[k], exp[k]), (i, k)
        assert exp["n_keypoints"] > 0
    finally:
        c.close()


def test_synthetic_1280x720_properties_and_parity():
    """C5-sized synthetic frames (1280x720, 1000 features, 4-level LK): parity with the oracle on
    a short run plus size-independent properties (ids unique and increasing, ages, statuses)."""
    L, R = euroc_cams()
    for cam in (L, R):
        cam.width, cam.height = 1280, 720
        cam.intrinsics[0] *= 1280 / 752.0
        cam.intrinsics[1] *= 1280 / 752.0
        cam.intrinsics[2] = 640.0 + (cam.intrinsics[2] - 376.0)
        cam.intrinsics[3] = 360.0 + (cam.intrinsics[3] - 240.0)
    p = euroc_params(max_features_per_frame=1000)
    p.tracker.klt_max_level = 3
    st = synth.SyntheticStream(L, seed=7)
    fe = O.Frontend(L, R, p)
    c = F.Context(L, R, p, batch=1)
    try:
        kf = 0
        seen = set()
        for t in range(4):
            l, r = st.frame(t)
            Rk = synth.keyframe_R_cur(st, kf, t)
            ts = t * 50_000_000
            c.step_host(l[None], r[None], c.make_inputs([ts], [Rk], [1]))
            exp = fe.process(l, r, ts, Rk, True)
            got = c.get_output(0)
            assert got["n_keypoints"] == exp["n_keypoints"] > 500
            for k in ("landmarks", "landmarks_age", "keypoints", "left_status", "right_status",
                      "right_rect_xy", "depth"):
                assert np.array_equal(got[k], exp[k]), (t, k)
            ids = got["landmarks"]
            assert len(set(ids.tolist())) == len(ids)
            new = ids[got["n_tracked"]:]
            assert np.all(np.diff(new) == 1) if len(new) > 1 else True
            assert not (set(new.tolist()) & seen)
            seen |= set(ids.tolist())
            assert np.all(got["landmarks_age"][got["n_tracked"]:] == 1)
            kf = t
    finally:
        c.close()


# ---------------------------------------------------------------------------------------------
# geometric outlier rejection (FrontendParams::useRANSAC_)
# ---------------------------------------------------------------------------------------------
import test_oracle_ransac as TR  # scene generators shared with the oracle's known-answer tests


@pytest.fixture(scope="module")
def rctx():
    L, R = euroc_cams()
    p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=None)
    assert p.use_ransac == 1 and p.tracker.ransac_use_2point_mono == 1 and p.tracker.ransac_use_1point_stereo == 1
    c = F.Context(L, R, p)
    yield c
    c.close()


def _same_ransac(got, exp):
    assert got["status"] == exp["status"]
    assert list(got["inliers"]) == list(exp["inliers"])
    assert got["iterations"] == exp["iterations"]
    assert np.array_equal(got["pose"], exp["pose"]), (got["pose"], exp["pose"])
    assert np.array_equal(got["info"], exp["info"])


@pytest.mark.parametrize("planar,n_in,n_out,seed", [(False, 80, 0, 1), (False, 80, 20, 2), (True, 80, 20, 3),
                                                    (False, 300, 250, 4), (False, 6, 0, 5), (False, 1, 0, 6)])
def test_outlier_rejection_2d2d_given_rotation(rctx, ocam, planar, n_in, n_out, seed):
    """2-point RANSAC (opengv TranslationOnlySacProblem): same sample stream, same hypotheses, same
    inlier sets, iteration count and pose as the CPU path; scenes of tests/testTracker.cpp:804-895."""
    rng = np.random.default_rng(seed)
    R = TR.expmap([0.02, -0.01, 0.03]) if seed % 2 else np.eye(3)
    T = np.array([1.0, 0.0, 0.0]) if seed % 2 == 0 else np.array([0.3, 0.1, -0.05])
    f_ref, f_cur = TR.mono_scene(ocam, rng, R, T, planar, n_in, n_out)
    exp = O.outlier_rejection_2d2d_given_rotation(f_ref, f_cur, R, rctx.params.tracker)
    got = rctx.outlier_rejection_2d2d_given_rotation(f_ref, f_cur, R)
    _same_ransac(got, exp)
    if n_in >= 10:  # (a random outlier pair can lie on an epipolar line: allow a few extra inliers)
        assert exp["status"] == abi.TRACKING_VALID
        assert set(range(n_in)) <= set(exp["inliers"]) and len(exp["inliers"]) <= n_in + 3


@pytest.mark.parametrize("case,planar,n_in,n_out,noise", [(0, False, 3, 0, 0.0), (1, False, 40, 0, 0.0),
                                                          (2, False, 80, 40, 0.0), (3, True, 80, 40, 0.01),
                                                          (4, False, 500, 100, 0.0), (5, False, 1, 0, 0.0),
                                                          (6, False, 0, 0, 0.0)])
def test_outlier_rejection_3d3d_given_rotation(rctx, ocam, case, planar, n_in, n_out, noise):
    """1-point voting: float32 Mahalanobis coherence, first largest coherent set, float64 information-
    weighted translation summed in inlier order; scenes of tests/testTracker.cpp:1042-1185."""
    rng = np.random.default_rng(300 + case)
    R = TR.expmap([0.1, 0.1, 0.1])
    T = np.array([ocam.rect.baseline, 0.0, 0.0])
    rl, rr, p_ref, cl, cr, p_cur = TR.stereo_scene(ocam, rng, R, T, planar, n_in, n_out, noise)
    exp = O.outlier_rejection_3d3d_given_rotation(ocam, rl, rr, p_ref, cl, cr, p_cur, R, rctx.params.tracker)
    got = rctx.outlier_rejection_3d3d_given_rotation(rl, rr, p_ref, cl, cr, p_cur, R)
    exp["iterations"] = got["iterations"] = 1
    _same_ransac(got, exp)
    if n_in >= 5:
        assert exp["status"] == abi.TRACKING_VALID and list(exp["inliers"]) == list(range(n_in))


@pytest.mark.parametrize("case,n_in,n_out,noise", [(0, 3, 0, 0.0), (1, 40, 0, 0.0), (2, 80, 40, 0.0),
                                                   (3, 400, 300, 0.002), (4, 2, 0, 0.0), (5, 0, 0, 0.0)])
def test_outlier_rejection_3d3d_arun(ocam, case, n_in, n_out, noise):
    """3-point Arun RANSAC (opengv PointCloudSacProblem, Tracker.cpp:667-742): same sample stream, the same
    Jacobi SVD operation for operation on both sides -> identical hypotheses, inlier sets, iteration counts
    and poses; scenes of tests/testTracker.cpp:898-1039."""
    L, R_ = euroc_cams()
    p = _euroc_ransac_params()
    p.tracker.ransac_use_1point_stereo = 0
    p.tracker.ransac_threshold_stereo = 1e-3 if noise == 0 else 0.05
    c = F.Context(L, R_, p)
    try:
        rng = np.random.default_rng(400 + case)
        R = TR.expmap([0.1, 0.1, 0.1])
        T = np.array([ocam.rect.baseline, 0.0, 0.0])
        _, _, p_ref, _, _, p_cur = TR.stereo_scene(ocam, rng, R, T, False, n_in, n_out, noise)
        exp = O.outlier_rejection_3d3d(p_ref, p_cur, p.tracker)
        got = c.outlier_rejection_3d3d(p_ref, p_cur)
        _same_ransac(got, exp)
        if n_in >= 3 and noise == 0:
            assert exp["status"] in (abi.TRACKING_VALID, abi.TRACKING_FEW_MATCHES)
            assert list(exp["inliers"]) == list(range(n_in))
            assert np.allclose(exp["pose"][:, :3], R, atol=1e-6) and np.allclose(exp["pose"][:, 3], T, atol=1e-6)
        if n_in < 3:
            assert exp["status"] == abi.TRACKING_INVALID
    finally:
        c.close()


@pytest.mark.parametrize("planar,n_in,n_out,seed", [(False, 82, 0, 1), (False, 80, 40, 2), (True, 80, 40, 3),
                                                    (False, 300, 200, 4), (False, 8, 0, 5), (False, 7, 0, 6)])
def test_outlier_rejection_2d2d_five_point(ocam, planar, n_in, n_out, seed):
    """5-point Nister RANSAC (opengv CentralRelativePoseSacProblem, ransac_use_2point_mono: 0): the device
    solves 16 hypotheses per round in block-cooperative stages and replays them in order; same sample stream,
    same solver operation for operation -> identical inlier sets, iteration counts and poses as the CPU path; scenes of
    tests/testTracker.cpp:704-802."""
    L, R_ = euroc_cams()
    p = _euroc_ransac_params()
    p.tracker.ransac_use_2point_mono = 0
    p.tracker.ransac_max_iterations = 1000
    c = F.Context(L, R_, p)
    try:
        rng = np.random.default_rng(500 + seed)
        R = TR.expmap([0.01, 0.01, 0.01])
        T = np.array([1.0, 0.0, 0.0])
        f_ref, f_cur = TR.mono_scene(ocam, rng, R, T, planar, n_in, n_out)
        exp = O.outlier_rejection_2d2d(f_ref, f_cur, p.tracker)
        got = c.outlier_rejection_2d2d(f_ref, f_cur)
        _same_ransac(got, exp)
        if n_in >= 80:
            assert exp["status"] == abi.TRACKING_VALID
            assert set(range(n_in)) <= set(exp["inliers"]) and len(exp["inliers"]) <= n_in + 3
            assert np.allclose(exp["pose"][:, :3], R, atol=1e-8)
        if n_in < 8:
            assert exp["status"] == abi.TRACKING_INVALID   # fewer matches than the 5 + 3 sample
    finally:
        c.close()


def test_frontend_sequence_d455_style_ransac(seq, ocam):
    """The shipped params/D455/FrontendParams.yaml as it is (ransac_use_2point_mono: 0, ransac_use_1point_stereo:
    0, 500 iterations, no optical-flow predictor) on the EuRoC camera pair: keyframe outlier rejection through
    the 5-point and the 3-point problems, two streams, identical to the oracle."""
    seq = dict(seq)
    seq["camR"] = _kf_rotations(seq["body_R"], ocam)
    L, R = euroc_cams()
    p = P.load_frontend_params(os.path.join(G, "params_d455", "FrontendParams.yaml"), use_ransac=None)
    assert p.use_ransac == 1 and p.tracker.ransac_use_2point_mono == 0 and p.tracker.ransac_use_1point_stereo == 0
    assert p.tracker.ransac_max_iterations == 500 and p.tracker.pose_2d2d_algorithm == 1
    p.detector.max_features_per_frame = 200
    fe = [O.Frontend(L, R, p) for _ in range(2)]
    c = F.Context(L, R, p, batch=2)
    try:
        _run_sequence(fe, c, seq, stream_of=lambda s, i: i if s == 0 else 8 - i, force_kf=True, n=6)
        last = [c.get_output(s) for s in range(2)]
    finally:
        c.close()
    for o in last:
        assert o["tracking_status_mono"] in (abi.TRACKING_VALID, abi.TRACKING_LOW_DISPARITY, abi.TRACKING_FEW_MATCHES)
        assert o["nr_mono_putatives"] > 50


@pytest.mark.parametrize("one_point", [0, 1])
def test_frontend_sequence_three_point_stereo(seq, ocam, one_point):
    """outlierRejectionStereo's 3-point branch (VisionImuFrontend.cpp:137-142): with
    ransac_use_1point_stereo: 0 (params/D455, KinectAzure) on every keyframe, and with the shipped Euroc
    setting on keyframes whose gyro rotation is exactly identity (streams alternate here)."""
    seq = dict(seq)
    seq["camR"] = _kf_rotations(seq["body_R"], ocam)
    if one_point:
        seq["camR"] = [np.eye(3) for _ in seq["camR"]]   # no usable rotation: mono INVALID, stereo Arun
    L, R = euroc_cams()
    p = _euroc_ransac_params(max_features_per_frame=200)
    p.tracker.ransac_use_1point_stereo = one_point
    fe = [O.Frontend(L, R, p) for _ in range(2)]
    c = F.Context(L, R, p, batch=2)
    try:
        _run_sequence(fe, c, seq, stream_of=lambda s, i: i if s == 0 else 8 - i, force_kf=True, n=6)
        last = [c.get_output(s) for s in range(2)]
    finally:
        c.close()
    for o in last:
        assert o["tracking_status_stereo"] in (abi.TRACKING_VALID, abi.TRACKING_FEW_MATCHES)
        assert np.all(o["info_mat_stereo_translation"] == 0)
        assert o["nr_stereo_putatives"] > 20 and o["nr_stereo_inliers"] > 10


def _euroc_ransac_params(**det):
    p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=None)
    for k, v in det.items():
        setattr(p.detector, k, v)
    return p


def test_frontend_sequence_with_outlier_rejection(seq, ocam):
    """The shipped params/Euroc/FrontendParams.yaml as it is (useRANSAC: 1, 2-point mono + 1-point
    stereo): landmarks rejected by the mono RANSAC become -1 and are re-detected, measurements skip
    them, TrackerStatusSummary (statuses, lkf_T_k, information matrix) identical to the oracle."""
    seq = dict(seq)
    seq["camR"] = _kf_rotations(seq["body_R"], ocam)
    L, R = euroc_cams()
    p = _euroc_ransac_params()
    fe = [O.Frontend(L, R, p)]
    c = F.Context(L, R, p, batch=1)
    try:
        kinds = _run_sequence(fe, c, seq)
        last = c.get_output(0)
    finally:
        c.close()
    # the clip is almost static: the keyframe at 0.2 s is LOW_DISPARITY, which (VisionImuFrontend.cpp:
    # 194-197) suppresses the "disparity low for the first time" keyframe at 0.4 s
    assert [k[2] for k in kinds] == [1, 0, 0, 0, 1, 0, 0, 0, 0]
    assert last["tracking_status_mono"] == abi.TRACKING_LOW_DISPARITY
    assert last["tracking_status_stereo"] == abi.TRACKING_VALID
    assert last["nr_mono_putatives"] > 250 and last["nr_mono_inliers"] > 30 and last["nr_stereo_inliers"] > 10


def test_frontend_outlier_rejection_batched_and_forced_keyframes(seq, ocam):
    """Outlier rejection on every frame (forced keyframes) for 3 streams with different frame orders:
    exercises landmark -1 entries in frames k-1 / lkf (tracking gather, keyframe matching)."""
    seq = dict(seq)
    seq["camR"] = _kf_rotations(seq["body_R"], ocam)
    L, R = euroc_cams()
    p = _euroc_ransac_params(max_features_per_frame=200)
    p.tracker.ransac_threshold_mono = 2e-7   # tight: real outliers on most keyframes
    B = 3
    fe = [O.Frontend(L, R, p) for _ in range(B)]
    c = F.Context(L, R, p, batch=B)

    def stream_of(s, i):
        return [i, 8 - i, (2 * i) % 9][s]

    try:
        _run_sequence(fe, c, seq, stream_of=stream_of, force_kf=True, n=8)
        outs = [c.get_output(s) for s in range(B)]
    finally:
        c.close()
    assert any((o["landmarks"] == -1).any() for o in outs)


# ---------------------------------------------------------------------------------------------
# input side: equalizeHist and the staged (pinned, copy-stream) hand-off
# ---------------------------------------------------------------------------------------------
def test_equalize_hist_bit_exact(ctx):
    for name in ("left_img_0.png", "right_img_0.png", "left_fisheye_img_0.png"):
        img = gray(name)
        assert np.array_equal(ctx.equalize_hist(img), O.equalize_hist(img)), name
    const = np.full((480, 752), 200, np.uint8)
    assert np.array_equal(ctx.equalize_hist(const), const)


@pytest.mark.parametrize("equalize", [0, 1])
def test_frontend_staged_input_matches_host_input(seq, ocam, equalize):
    """kvfe_frontend_step_staged (pinned slots, upload on the copy stream overlapping the previous
    step) == kvfe_frontend_step_host == oracle, with and without equalizeImage, 2 streams."""
    seq = dict(seq)
    seq["camR"] = _kf_rotations(seq["body_R"], ocam)
    L, R = euroc_cams()
    p = _euroc_ransac_params(max_features_per_frame=150)
    p.stereo.equalize_image = equalize
    B = 2
    fe = [O.Frontend(L, R, p) for _ in range(B)]
    c = F.Context(L, R, p, batch=B)
    try:
        kf = [0] * B
        for i in range(9):
            idx = [i, 8 - i]
            Rs = [seq["camR"][kf[s]].T @ seq["camR"][idx[s]] for s in range(B)]
            ts = [int(seq["ts"][i])] * B
            slot = i % 3
            c.staging_wait(slot)
            sl, sr = c.staging_buffers(slot)
            for s in range(B):
                sl[s] = seq["lefts"][idx[s]]
                sr[s] = seq["rights"][idx[s]]
            c.step_staged(slot, c.make_inputs(ts, Rs, [0] * B))   # enqueue only: the next slot is
            if i + 1 < 9:                                           # filled while this step runs
                c.staging_wait((i + 1) % 3)
            for s in range(B):
                exp = fe[s].process(seq["lefts"][idx[s]], seq["rights"][idx[s]], ts[s], Rs[s], False)
                got = c.get_output(s)
                for k in ("n_keypoints", "is_keyframe", "n_tracked", "n_detected", "n_measurements",
                          "tracking_status_mono", "tracking_status_stereo"):
                    assert got[k] == exp[k], (i, s, k)
                for k in ("landmarks", "keypoints", "versors"):
                    assert np.array_equal(got[k], exp[k]), (i, s, k)
                if exp["is_keyframe"]:
                    assert np.array_equal(got["meas_uL_uR_v"], exp["meas_uL_uR_v"], equal_nan=True)
                    kf[s] = idx[s]
    finally:
        c.close()


def test_frontend_stream_groups_equalize_device_input(seq, ocam):
    """stream_groups > 1 + equalizeImage + kvfe_frontend_step_device used to skip cv::equalizeHist
    (the group path called do_step directly).  3 streams in 2 groups, device-resident raw images, every frame
    equal to the oracle (which equalises, UtilsOpenCV.cpp:398-401)."""
    import torch
    from parity_util import assert_step_equal
    dev = torch.device("cuda", 0)
    seq = dict(seq)
    seq["camR"] = _kf_rotations(seq["body_R"], ocam)
    L, R = euroc_cams()
    p = _euroc_ransac_params(max_features_per_frame=150)
    p.stereo.equalize_image = 1
    B = 3
    fe = [O.Frontend(L, R, p) for _ in range(B)]
    c = F.Context(L, R, p, batch=B, stream_groups=2)
    keep = []
    try:
        kf = [0] * B
        for i in range(6):
            idx = [i, 8 - i, (2 * i) % 9]
            Rs = [seq["camR"][kf[s]].T @ seq["camR"][idx[s]] for s in range(B)]
            ts = [int(seq["ts"][i])] * B
            dl = torch.from_numpy(np.stack([seq["lefts"][j] for j in idx])).to(dev)
            dr = torch.from_numpy(np.stack([seq["rights"][j] for j in idx])).to(dev)
            keep = keep[-2:] + [(dl, dr)]
            c.step_device(dl.data_ptr(), dr.data_ptr(), c.make_inputs(ts, Rs, [0] * B))
            for s in range(B):
                exp = fe[s].process(seq["lefts"][idx[s]], seq["rights"][idx[s]], ts[s], Rs[s], False)
                assert_step_equal(c.get_output(s), exp, (i, s))
                if exp["is_keyframe"]:
                    kf[s] = idx[s]
    finally:
        c.close()


def test_c_abi_argument_validation(ctx):
    """the C ABI never aborts: negative capacities / sizes are KVFE_ERR_INVALID_ARG"""
    import ctypes as C
    from kimera_vio_amd.lib import KvfeError
    out = abi.FrameOutput()
    out.capacity = -1
    assert ctx.lib.kvfe_frontend_get_output(ctx._h, 0, C.byref(out)) == abi.KVFE_ERR_INVALID_ARG
    L, R = euroc_cams()
    p = euroc_params(max_nr_keypoints_before_anms=-5)
    with pytest.raises(KvfeError) as e:
        F.Context(L, R, p)
    assert e.value.status == abi.KVFE_ERR_INVALID_ARG


# ---------------------------------------------------------------------------------------------
# ANMS: the radius-search variants of anms/anms.cpp (RangeTree is the class default)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("anms_type", [abi.ANMS_BROWN, abi.ANMS_SDC, abi.ANMS_KDTREE, abi.ANMS_RANGETREE, abi.ANMS_SSC])
def test_feature_detection_anms_variants(anms_type, seq):
    """featureDetection with non_max_suppression_type 1..5.  BrownANMS (1): the radii and the libstdc++ std::sort
    order over their many ties (kvfe_stdsort.inl on the device, the real std::sort in the oracle).  2..5: the
    binary search on the suppression radius and every greedy sweep equal the CPU path (same corners, same order, then cornerSubPix), with and
    without tracked keypoints, for several `need`; need < 2 returns nothing (closed form divides by 0)."""
    L, R = euroc_cams()
    p = euroc_params()
    p.detector.non_max_suppression_type = anms_type
    c = F.Context(L, R, p)
    try:
        none = np.zeros((0, 2), np.float32)
        for img in (gray("left_fisheye_img_0.png"), seq["lefts"][2]):
            for need in (300, 120, 40, 1500):
                got = c.feature_detection(img, none, need)
                exp, _ = O.feature_detection(img, none, need, p.detector)
                assert np.array_equal(got, exp), (anms_type, need, len(got), len(exp))
            assert len(exp) > 0
            first = exp
            tracked = first[::3]
            got = c.feature_detection(img, tracked, 200)
            exp, _ = O.feature_detection(img, tracked, 200, p.detector)
            assert len(exp) > 20 and np.array_equal(got, exp)
        # numRetPoints 0 / 1 (the frame already holds maxFeaturesPerFrame keypoints): Sdc is defined upstream and
        # still returns a corner or two; KdTree / RangeTree / Ssc divide by zero there -> no new corners on
        # both sides (found by tools/fuzz_frontend.py)
        for need in (0, 1, 2, 3):
            got = c.feature_detection(img, none, need)
            exp, _ = O.feature_detection(img, none, need, p.detector)
            assert np.array_equal(got, exp), (anms_type, need, len(got), len(exp))
            if anms_type not in (abi.ANMS_SDC, abi.ANMS_BROWN) and need < 2:
                assert len(got) == 0
        if anms_type == abi.ANMS_SDC:
            assert len(c.feature_detection(img, none, 0)) >= 1
    finally:
        c.close()


def test_feature_detection_small_min_distance(seq):
    """min_distance 5 on 752x480: more grid cells of side minDistance than the LDS work area holds; the
    acceleration grid then uses larger cells (found by tools/fuzz_frontend.py)"""
    L, R = euroc_cams()
    for anms in (abi.ANMS_RANGETREE, abi.ANMS_BINNING):
        p = euroc_params(min_distance=5, quality_level=0.001, non_max_suppression_type=anms,
                         max_features_per_frame=200)
        c = F.Context(L, R, p)
        try:
            none = np.zeros((0, 2), np.float32)
            img = seq["lefts"][4]
            got = c.feature_detection(img, none, 200)
            exp, _ = O.feature_detection(img, none, 200, p.detector)
            assert len(exp) > 100 and np.array_equal(got, exp)
            raw = c.raw_feature_detection(img)
            assert len(raw) == p.detector.max_nr_keypoints_before_anms   # dense texture: the cap is hit
        finally:
            c.close()


def test_feature_detection_more_candidates_than_the_lds_list():
    """band-limited noise at 1280x720 and 752x480: 10-25 k local maxima above the quality threshold, more than the
    8192 keys the select kernel ranks in LDS -> its two-pass path (top bins first, then the survivors of the bitmap);
    raw GFTT output (order included), the ANMS selection and the refined corners equal the CPU path.  With
    min_distance 3 the second pass still accepts corners; with a smooth image the one-pass path runs as control."""
    from scipy import ndimage as ndi
    from kimera_vio_amd import workloads as WL
    rng = np.random.default_rng(77)
    none = np.zeros((0, 2), np.float32)
    for (w, h, sigma, md) in ((1280, 720, 1.2, 20), (1280, 720, 1.0, 3), (752, 480, 0.9, 8), (1280, 720, 3.0, 20)):
        img = ndi.gaussian_filter(rng.normal(size=(h, w)), sigma)
        img = np.clip(128 + img / img.std() * 45, 0, 255).astype(np.uint8)
        L, R = WL.make_cameras(w, h)
        p = euroc_params(min_distance=md, quality_level=0.0005, max_features_per_frame=800)
        p.detector.max_nr_keypoints_before_anms = 6000
        c = F.Context(L, R, p)
        try:
            raw = c.raw_feature_detection(img)
            exp_raw, _ = O.good_features_to_track(img, 6000, 0.0005, md, 3)
            assert len(exp_raw) > 300 and np.array_equal(raw, exp_raw), (w, h, sigma, md, len(raw), len(exp_raw))
            got = c.feature_detection(img, none, 800)
            exp, _ = O.feature_detection(img, none, 800, p.detector)
            assert np.array_equal(got, exp), (w, h, sigma, md)
        finally:
            c.close()


def test_large_keypoint_capacity_lds_budget(seq, ocam):
    """max_nr_keypoints_before_anms 6000: the per-stream bookkeeping / outlier-rejection kernels need 120 KB of dynamic
    LDS (above the 64 KB a launch gets without the opt-in: found by tools/fuzz_frontend.py as an aborted HSA queue).
    The sequence still equals the oracle; a capacity no kernel can hold is refused at kvfe_create with
    KVFE_ERR_UNSUPPORTED (5-point mono RANSAC: its solver slots leave less LDS)."""
    seq = dict(seq)
    seq["camR"] = _kf_rotations(seq["body_R"], ocam)
    L, R = euroc_cams()
    p = _euroc_ransac_params(max_features_per_frame=300)
    p.detector.max_nr_keypoints_before_anms = 6000
    fe = [O.Frontend(L, R, p)]
    c = F.Context(L, R, p, batch=1)
    try:
        _run_sequence(fe, c, seq, force_kf=True, n=4)
    finally:
        c.close()
    q = _euroc_ransac_params(max_features_per_frame=1000)
    q.detector.max_nr_keypoints_before_anms = 8192
    q.tracker.ransac_use_2point_mono = 0
    with pytest.raises(F.KvfeError) as e:
        F.Context(L, R, q, batch=1)
    assert e.value.status == abi.KVFE_ERR_UNSUPPORTED


def test_frontend_sequence_class_default_anms(seq, ocam):
    """FeatureDetectorParams' class default is RangeTree (FeatureDetectorParams.h): the front-end with
    it, outlier rejection on, identical to the oracle over the clip."""
    seq = dict(seq)
    seq["camR"] = _kf_rotations(seq["body_R"], ocam)
    L, R = euroc_cams()
    p = _euroc_ransac_params(non_max_suppression_type=abi.ANMS_RANGETREE, max_features_per_frame=250)
    fe = [O.Frontend(L, R, p)]
    c = F.Context(L, R, p, batch=1)
    try:
        _run_sequence(fe, c, seq, force_kf=True, n=6)
    finally:
        c.close()


# ---------------------------------------------------------------------------------------------
# monocular front-end (MonoVisionImuFrontend): the same kernels without the right camera
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("use_ransac,force_kf", [(1, False), (1, True), (0, False)])
def test_mono_frontend_sequence(seq, use_ransac, force_kf):
    """MonoVisionImuFrontend::processFirstFrame / processFrame: tracking, keyframe decision, 2-point
    RANSAC, detection, Camera::undistortKeypoints (R = I, P = K) and the mono measurements
    (uL, NaN, v) identical to the oracle; bearing vectors are in the unrectified camera frame."""
    L, _ = euroc_cams()
    p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=use_ransac)
    p.detector.max_features_per_frame = 200
    TL = np.array(L.body_pose_cam).reshape(4, 4)
    camR = [TL[:3, :3].T @ Rb @ TL[:3, :3] for Rb in seq["body_R"]]  # rotations of the (unrectified) camera
    fe = O.Frontend(L, L, p, mono=True)
    c = F.Context(L, L, p, batch=1, frontend_type=abi.FRONTEND_MONO)
    try:
        kf = 0
        n_kf = 0
        for i in range(9):
            Rk = camR[kf].T @ camR[i]
            ts = int(seq["ts"][i])
            c.step_host(seq["lefts"][i][None], None, c.make_inputs([ts], [Rk], [int(force_kf)]))
            exp = fe.process(seq["lefts"][i], seq["lefts"][i], ts, Rk, force_kf)
            got = c.get_output(0)
            for k in ("n_keypoints", "is_keyframe", "n_tracked", "n_detected", "n_measurements",
                      "tracking_status_mono", "tracking_status_stereo", "nr_mono_putatives", "nr_mono_inliers"):
                assert got[k] == exp[k], (i, k, got[k], exp[k])
            for k in ("landmarks", "landmarks_age", "keypoints", "versors"):
                assert np.array_equal(got[k], exp[k]), (i, k)
            assert np.array_equal(got["lkf_T_k_mono"], exp["lkf_T_k_mono"])
            if exp["is_keyframe"]:
                assert np.array_equal(got["left_rect_xy"], exp["left_rect_xy"])
                assert np.array_equal(got["left_status"], exp["left_status"])
                assert np.array_equal(got["meas_landmark"], exp["meas_landmark"])
                assert np.array_equal(got["meas_uL_uR_v"], exp["meas_uL_uR_v"], equal_nan=True)
                assert np.isnan(got["meas_uL_uR_v"][:, 1]).all() or len(got["meas_uL_uR_v"]) == 0
                kf = i
                n_kf += 1
            assert got["tracking_status_stereo"] == (abi.TRACKING_DISABLED if i > 0 else abi.TRACKING_INVALID)
        assert n_kf >= 2
    finally:
        c.close()


# ---------------------------------------------------------------------------------------------
# RgbdVisionImuFrontend: depth image instead of a right camera
# ---------------------------------------------------------------------------------------------
def _synthetic_depth(h, w, t, depth_type, seed=0):
    """smooth scene depth 1.2-6 m that drifts with the frame index, with holes (0 / NaN), a band beyond
    max_depth and a band below min_depth; uint16 millimetres or float32 metres"""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    d = 3.5 + 2.2 * np.sin(xx / 97.0 + 0.05 * t) * np.cos(yy / 71.0) + 0.002 * yy
    d[60:90, 100:260] = 14.0            # beyond max_depth 10 m: masked out of detection, still a valid depth
    d[300:330, 400:520] = 0.05          # below min_depth
    holes = rng.randint(0, 100, (h, w)) < 4
    if depth_type == abi.DEPTH_F32:
        d = d.astype(np.float32)
        d[holes] = np.nan
        d[200:205, 600:640] = np.inf
        return d
    d = np.clip(np.rint(d * 1000.0), 0, 65535).astype(np.uint16)
    d[holes] = 0
    return d


@pytest.mark.parametrize("depth_type,use_ransac,one_point", [(abi.DEPTH_U16, 1, 1), (abi.DEPTH_F32, 1, 0),
                                                              (abi.DEPTH_U16, 0, 1)])
def test_rgbd_frontend_sequence(seq, depth_type, use_ransac, one_point):
    """RgbdVisionImuFrontend::processFirstFrame / processFrame / handleKeyframe: detection under the
    depth-range mask, tracking, undistortion with R = I / P = K, RgbdFrame::fillStereoFrame (hallucinated
    right keypoints, depths, 3-D points, distorted right pixels), mono + stereo outlier rejection with the
    fake stereo camera, fillSmartStereoMeasurements — all identical to the oracle, two streams."""
    L, _ = euroc_cams()
    p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=use_ransac)
    p.detector.max_features_per_frame = 200
    p.tracker.ransac_use_1point_stereo = one_point
    p.tracker.ransac_threshold_stereo = 0.5
    dp = abi.depth_params_default(depth_type)
    dp.virtual_baseline = 0.05
    dp.min_depth = 0.3
    if depth_type == abi.DEPTH_U16:
        dp.depth_to_meters = 0.001
    TL = np.array(L.body_pose_cam).reshape(4, 4)
    camR = [TL[:3, :3].T @ Rb @ TL[:3, :3] for Rb in seq["body_R"]]
    B = 2
    fe = [O.Frontend(L, L, p, depth=dp) for _ in range(B)]
    c = F.Context(L, L, p, batch=B, frontend_type=abi.FRONTEND_RGBD, depth=dp)
    h, w = seq["lefts"][0].shape
    try:
        kf = [0] * B
        n_kf = 0
        for i in range(9):
            idx = [i, 8 - i]
            Rs = [camR[kf[s]].T @ camR[idx[s]] for s in range(B)]
            ts = [int(seq["ts"][i])] * B
            lefts = np.stack([seq["lefts"][j] for j in idx])
            depths = np.stack([_synthetic_depth(h, w, j, depth_type, seed=j) for j in idx])
            c.step_host(lefts, depths, c.make_inputs(ts, Rs, [0] * B))
            for s in range(B):
                exp = fe[s].process(lefts[s], depths[s], ts[s], Rs[s], False)
                got = c.get_output(s)
                for k in ("n_keypoints", "is_keyframe", "n_tracked", "n_detected", "n_measurements",
                          "tracking_status_mono", "tracking_status_stereo", "nr_mono_putatives", "nr_mono_inliers",
                          "nr_stereo_putatives", "nr_stereo_inliers"):
                    assert got[k] == exp[k], (i, s, k, got[k], exp[k])
                for k in ("landmarks", "landmarks_age", "keypoints", "versors", "lkf_T_k_mono", "lkf_T_k_stereo",
                          "info_mat_stereo_translation"):
                    assert np.array_equal(got[k], exp[k]), (i, s, k)
                if exp["is_keyframe"]:
                    for k in ("left_rect_xy", "left_status", "right_rect_xy", "right_status", "depth", "right_xy",
                              "keypoints_3d", "meas_landmark"):
                        assert np.array_equal(got[k], exp[k]), (i, s, k)
                    assert np.array_equal(got["meas_uL_uR_v"], exp["meas_uL_uR_v"], equal_nan=True)
                    kf[s] = idx[s]
                    n_kf += 1
                    if i == 8:
                        st = exp["right_status"]
                        assert (st == 0).sum() > 50 and (st == 3).sum() > 0     # VALID and NO_DEPTH both occur
                        m = exp["meas_uL_uR_v"]
                        assert np.isnan(m[:, 1]).any() and (~np.isnan(m[:, 1])).any()
                if not use_ransac and exp["is_keyframe"] and i > 0:
                    assert got["tracking_status_mono"] == got["tracking_status_stereo"] == abi.TRACKING_DISABLED
        assert n_kf >= 4
    finally:
        c.close()


def test_rgbd_unsupported_configurations():
    L, _ = euroc_cams()
    p = euroc_params()
    dp = abi.depth_params_default()
    dp.is_registered = 0            # needs cv::rgbd::registerDepth
    with pytest.raises(F.KvfeError) as e:
        F.Context(L, L, p, frontend_type=abi.FRONTEND_RGBD, depth=dp)
    assert e.value.status == abi.KVFE_ERR_UNSUPPORTED
    dp = abi.depth_params_default()
    dp.virtual_baseline = 0.0       # CHECK_GT(virtual_baseline_, 0.0) (CameraParams.cpp:344)
    with pytest.raises(F.KvfeError) as e:
        F.Context(L, L, p, frontend_type=abi.FRONTEND_RGBD, depth=dp)
    assert e.value.status == abi.KVFE_ERR_INVALID_ARG


# ---------------------------------------------------------------------------------------------
# equidistant distortion model (cv::fisheye; params/RealSenseIR, tests/data/ForStereoFrame/*_fisheye.yaml)
# ---------------------------------------------------------------------------------------------
def _fisheye_cams():
    return (P.load_camera_params(os.path.join(G, "left_sensor_fisheye.yaml")),
            P.load_camera_params(os.path.join(G, "right_sensor_fisheye.yaml")))


def test_fisheye_rectify_and_keypoints(seq):
    """distortion_model: equidistant.  Rectified images bit-exact (the maps are host float64 math on both
    sides); undistorted keypoints / bearing vectors go through tan() on the device, whose last bit may
    differ from glibc's: tolerance 1e-4 px / 1e-7 (north_star allows 0.5 px)."""
    L, R = _fisheye_cams()
    cam = O.Camera(L, R)
    c = F.Context(L, R, euroc_params())
    try:
        for camid, img in ((0, gray("left_fisheye_img_0.png")), (1, seq["rights"][0])):
            assert np.array_equal(c.undistort_rectify_image(camid, img), cam.rectify_image(camid, img))
        rng = np.random.default_rng(3)
        px = np.stack([rng.uniform(0, 751, 500), rng.uniform(0, 479, 500)], 1).astype(np.float32)
        for useR, useP in ((True, True), (False, False), (True, False)):
            got = c.undistort_rectify_keypoints(0, px, useR, useP)
            exp = cam.undistort_keypoints(0, px, useR, useP)
            assert np.allclose(got, exp, rtol=0, atol=1e-4), (useR, useP, np.abs(got - exp).max())
        assert np.allclose(c.get_bearing_vectors(0, px), cam.bearing_vectors(0, px), rtol=0, atol=1e-7)
    finally:
        c.close()


def test_fisheye_frontend_sequence(seq):
    """the whole front-end on an equidistant pair (outlier rejection on): discrete outputs identical,
    float outputs that depend on tan() within 1e-4 px / 1e-6 relative."""
    L, R = _fisheye_cams()
    cam = O.Camera(L, R)
    p = _euroc_ransac_params(max_features_per_frame=200)
    TL = np.array(L.body_pose_cam).reshape(4, 4)
    body_R_cam = TL[:3, :3] @ np.array(cam.rect.R1).reshape(3, 3).T
    camR = [body_R_cam.T @ Rb @ body_R_cam for Rb in seq["body_R"]]
    fe = O.Frontend(L, R, p)
    c = F.Context(L, R, p, batch=1)
    try:
        kf = 0
        for i in range(6):
            Rk = camR[kf].T @ camR[i]
            ts = int(seq["ts"][i])
            c.step_host(seq["lefts"][i][None], seq["rights"][i][None], c.make_inputs([ts], [Rk], [1]))
            exp = fe.process(seq["lefts"][i], seq["rights"][i], ts, Rk, True)
            got = c.get_output(0)
            for k in ("n_keypoints", "is_keyframe", "n_tracked", "n_detected", "n_measurements",
                      "tracking_status_mono", "tracking_status_stereo", "nr_mono_putatives",
                      "nr_mono_inliers", "nr_stereo_putatives", "nr_stereo_inliers"):
                assert got[k] == exp[k], (i, k, got[k], exp[k])
            for k in ("landmarks", "landmarks_age", "keypoints", "left_status", "right_status", "meas_landmark"):
                assert np.array_equal(got[k], exp[k]), (i, k)
            assert np.allclose(got["versors"], exp["versors"], rtol=0, atol=1e-7)
            assert np.allclose(got["left_rect_xy"], exp["left_rect_xy"], rtol=0, atol=1e-4)
            assert np.allclose(got["right_rect_xy"], exp["right_rect_xy"], rtol=0, atol=1e-4)
            assert np.allclose(got["depth"], exp["depth"], rtol=1e-5, atol=0)
            assert np.allclose(got["keypoints_3d"], exp["keypoints_3d"], rtol=1e-5, atol=1e-9)
            assert np.allclose(got["lkf_T_k_stereo"], exp["lkf_T_k_stereo"], rtol=1e-5, atol=1e-7)
            kf = i
        assert (got["right_status"] == abi.KP_VALID).sum() > 20
    finally:
        c.close()


# ---------------------------------------------------------------------------------------------
# dense stereo: cv::StereoSGBM MODE_HH + medianBlur + filterSpeckles
# ---------------------------------------------------------------------------------------------
def _dense_pairs(ctx, ocam, seq):
    pairs = [(ocam.rectify_image(0, gray("left_img_0.png")), ocam.rectify_image(1, gray("right_img_0.png")))]
    for i in (2, 6):
        pairs.append((ocam.rectify_image(0, seq["lefts"][i]), ocam.rectify_image(1, seq["rights"][i])))
    return pairs


def test_dense_stereo_default_params_bit_exact(ctx, ocam, seq):
    """StereoMatcher::denseStereoReconstruction with the reference's DenseStereoParams (SGBM MODE_HH,
    block 11, 64 disparities from 1, P1/P2 120/240, speckle 500/3) on rectified EuRoC pairs: every
    int16 disparity identical to the oracle; batched call == single calls."""
    dp = abi.dense_stereo_params_default()
    pairs = _dense_pairs(ctx, ocam, seq)
    exp = [O.dense_stereo_reconstruction(l, r, dp) for l, r in pairs]
    got = ctx.dense_stereo_reconstruction([p[0] for p in pairs], [p[1] for p in pairs], dp)
    for g, e in zip(got, exp):
        assert np.array_equal(g, e), (np.count_nonzero(g != e), g.shape)
    single = ctx.dense_stereo_reconstruction(pairs[1][0], pairs[1][1], dp)
    assert np.array_equal(single, exp[1])
    valid = exp[0] != (dp.min_disparity - 1) * 16
    assert valid.mean() > 0.3        # the scene is matchable: not a trivially empty comparison
    assert exp[0][valid].min() >= dp.min_disparity * 16
    assert exp[0][valid].max() <= (dp.min_disparity + dp.num_disparities - 1) * 16


@pytest.mark.parametrize("kw", [
    dict(num_disparities=32, sad_window_size=5, p1=40, p2=160),
    dict(num_disparities=48, min_disparity=0, sad_window_size=7, uniqueness_ratio=10),
    dict(num_disparities=16, min_disparity=4, sad_window_size=3, disp_12_max_diff=2, speckle_window_size=0),
    dict(sad_window_size=9, uniqueness_ratio=5, speckle_window_size=100, speckle_range=1,
         median_blur_disparity=1, pre_filter_cap=63),
    dict(use_mode_hh=0),                                                      # cv::StereoSGBM::MODE_SGBM
    dict(use_mode_hh=0, num_disparities=32, min_disparity=3, sad_window_size=5, uniqueness_ratio=15),
])
def test_dense_stereo_parameter_variants_bit_exact(ctx, ocam, seq, kw):
    dp = abi.dense_stereo_params_default()
    for k, v in kw.items():
        setattr(dp, k, v)
    l, r = _dense_pairs(ctx, ocam, seq)[0]
    exp = O.dense_stereo_reconstruction(l, r, dp)
    got = ctx.dense_stereo_reconstruction(l, r, dp)
    assert np.array_equal(got, exp), (kw, np.count_nonzero(got != exp))


@pytest.mark.parametrize("kw", [
    dict(),                                                                    # reference defaults, use_sgbm_ = false
    dict(uniqueness_ratio=15, texture_threshold=10, speckle_window_size=100, speckle_range=4),
    dict(num_disparities=32, min_disparity=0, sad_window_size=21, pre_filter_cap=20, median_blur_disparity=1),
    dict(num_disparities=16, min_disparity=5, sad_window_size=5, speckle_window_size=0),
])
def test_dense_stereo_block_matching_bit_exact(ctx, ocam, seq, kw):
    """use_sgbm_ = false: cv::StereoBM (XSOBEL prefilter, SAD block matching inside the valid-disparity
    rectangle of the rectification's ROIs, texture / uniqueness tests, sub-pixel fit, filterSpeckles with the
    unscaled range) — every int16 identical to the oracle, on two pairs in one call."""
    dp = abi.dense_stereo_params_default()
    dp.use_sgbm = 0
    for k, v in kw.items():
        setattr(dp, k, v)
    pairs = _dense_pairs(ctx, ocam, seq)[:2]
    roi1, roi2 = list(ctx.rect.roi1), list(ctx.rect.roi2)
    exp = [O.dense_stereo_reconstruction(l, r, dp, roi1, roi2) for l, r in pairs]
    got = ctx.dense_stereo_reconstruction([p[0] for p in pairs], [p[1] for p in pairs], dp)
    for g, e in zip(got, exp):
        assert np.array_equal(g, e), (kw, np.count_nonzero(g != e))
    inv = (dp.min_disparity - 1) * 16
    assert (exp[0] != inv).mean() > 0.05


def test_dense_stereo_synthetic_shift_and_noise(ctx):
    """known disparity: right(x) = left(x + d) -> most valid pixels within 1 px of d, and bit-exact
    against the oracle also on pure noise (worst case for ties / saturation of the summed costs)."""
    dp = abi.dense_stereo_params_default()
    h, w = 480, 752
    tex = synth.base_texture(w, h, 5)
    base = np.clip(np.rint(tex[96:96 + h, 40:40 + w + 64]), 0, 255).astype(np.uint8)
    left = np.ascontiguousarray(base[:, :w])
    d = 23
    right = np.ascontiguousarray(base[:, d:d + w])
    exp = O.dense_stereo_reconstruction(left, right, dp)
    got = ctx.dense_stereo_reconstruction(left, right, dp)
    assert np.array_equal(got, exp)
    valid = got != (dp.min_disparity - 1) * 16
    assert valid.mean() > 0.5
    assert np.mean(np.abs(got[valid] / 16.0 - d) <= 1.0) > 0.95
    rng = np.random.RandomState(3)
    nl = rng.randint(0, 256, (h, w)).astype(np.uint8)
    nr = rng.randint(0, 256, (h, w)).astype(np.uint8)
    assert np.array_equal(ctx.dense_stereo_reconstruction(nl, nr, dp), O.dense_stereo_reconstruction(nl, nr, dp))
    flat = np.full((h, w), 77, np.uint8)
    assert np.array_equal(ctx.dense_stereo_reconstruction(flat, flat, dp),
                          O.dense_stereo_reconstruction(flat, flat, dp))


def test_dense_stereo_unsupported_configurations(ctx):
    l = np.zeros((480, 752), np.uint8)
    for kw in (dict(num_disparities=128), dict(sad_window_size=21), dict(use_sgbm=0, pre_filter_type=0),
               dict(use_sgbm=0, disp_12_max_diff=1)):
        dp = abi.dense_stereo_params_default()
        for k, v in kw.items():
            setattr(dp, k, v)
        with pytest.raises(F.KvfeError) as e:
            ctx.dense_stereo_reconstruction(l, l, dp)
        assert e.value.status == abi.KVFE_ERR_UNSUPPORTED, kw


def test_backproject_disparity_to_3d_bit_exact(ctx, ocam, seq):
    """StereoCamera::backProjectDisparityTo3D == cv::reprojectImageTo3D(handleMissingValues=true)"""
    dp = abi.dense_stereo_params_default()
    l, r = _dense_pairs(ctx, ocam, seq)[0]
    disp = ctx.dense_stereo_reconstruction(l, r, dp).astype(np.float32) / 16.0
    Q = np.array(ctx.rect.Q, np.float64).reshape(4, 4)
    exp = O.reproject_image_to_3d(disp, Q, True)
    got = ctx.backproject_disparity_to_3d(disp)
    # invalid pixels (disparity 0 -> w = 0) give inf / nan in x, y on both sides; z = 10000 there
    assert np.array_equal(got, exp, equal_nan=True)
    # tests/testStereoCamera.cpp:264-362: points reproject onto the pixels that generated them
    valid = (disp > 0) & (got[..., 2] < 5.0)
    fx, cx, cy = Q[2, 3], -Q[0, 3], -Q[1, 3]
    v, u = np.nonzero(valid)
    p = got[v, u]
    assert np.allclose(fx * p[:, 0] / p[:, 2] + cx, u, atol=1e-2)
    assert np.allclose(fx * p[:, 1] / p[:, 2] + cy, v, atol=1e-2)


@pytest.mark.parametrize("w,h", [(323, 241), (1280, 720)])
def test_dense_stereo_other_image_sizes(w, h):
    """odd sizes (tiles / chunks not multiples of 64 / 32) and BASELINE's C5 size: SGBM MODE_HH, MODE_SGBM and
    StereoBM identical to the oracle on a synthetic pair with a known shift"""
    from kimera_vio_amd import workloads
    L, R = workloads.make_cameras(w, h)
    c = F.Context(L, R, euroc_params())
    try:
        tex = synth.base_texture(w + 80, h, 11)
        base = np.clip(np.rint(tex[96:96 + h, 60:60 + w + 70]), 0, 255).astype(np.uint8)
        left, right = np.ascontiguousarray(base[:, :w]), np.ascontiguousarray(base[:, 13:13 + w])
        roi1, roi2 = list(c.rect.roi1), list(c.rect.roi2)
        for kw in (dict(), dict(use_mode_hh=0), dict(use_sgbm=0)):
            if (w, h) == (1280, 720) and kw:
                continue   # (the oracle takes ~1 s per SGBM call at this size: the default configuration only)
            dp = abi.dense_stereo_params_default()
            for k, v in kw.items():
                setattr(dp, k, v)
            exp = O.dense_stereo_reconstruction(left, right, dp, roi1, roi2)
            got = c.dense_stereo_reconstruction(left, right, dp)
            assert np.array_equal(got, exp), (w, h, kw, np.count_nonzero(got != exp))
            valid = got != (dp.min_disparity - 1) * 16
            if dp.use_sgbm:
                assert valid.mean() > 0.4 and np.mean(np.abs(got[valid] / 16.0 - 13) <= 1.0) > 0.9
    finally:
        c.close()


def test_frontend_device_input_path_matches_oracle(seq, ocam):
    """kvfe_frontend_step_device (inputs already in HBM: the path bench.py times) gives the same per-frame
    outputs as the oracle: stereo with the shipped Euroc parameters on two streams with a padded row stride,
    and the RGBD front-end with a device-resident uint16 depth image."""
    import torch
    dev = torch.device("cuda", 0)
    seq = dict(seq)
    seq["camR"] = _kf_rotations(seq["body_R"], ocam)
    L, R = euroc_cams()
    p = _euroc_ransac_params(max_features_per_frame=200)
    B, h, w = 2, 480, 752
    pitch = w + 16
    fe = [O.Frontend(L, R, p) for _ in range(B)]
    c = F.Context(L, R, p, batch=B)
    # the device buffers are read by their own step only (kvfe.h): ONE pair of buffers is reused for every frame and
    # scribbled over as soon as the step's output has been fetched (no pointer retained past a call)
    dl = torch.zeros((B, h, pitch), dtype=torch.uint8, device=dev)
    dr = torch.zeros((B, h, pitch), dtype=torch.uint8, device=dev)
    try:
        kf = [0] * B
        for i in range(7):
            idx = [i, 8 - i]
            Rs = [seq["camR"][kf[s]].T @ seq["camR"][idx[s]] for s in range(B)]
            ts = [int(seq["ts"][i])] * B
            hl = np.zeros((B, h, pitch), np.uint8)
            hr = np.zeros((B, h, pitch), np.uint8)
            for s in range(B):
                hl[s, :, :w] = seq["lefts"][idx[s]]
                hr[s, :, :w] = seq["rights"][idx[s]]
            dl.copy_(torch.from_numpy(hl))
            dr.copy_(torch.from_numpy(hr))
            torch.cuda.synchronize()
            c.step_device(dl.data_ptr(), dr.data_ptr(), c.make_inputs(ts, Rs, [0] * B), row_stride=pitch,
                          image_stride=h * pitch)
            c.synchronize()
            dl.fill_(0x5a)
            dr.fill_(0xa5)
            torch.cuda.synchronize()
            for s in range(B):
                exp = fe[s].process(seq["lefts"][idx[s]], seq["rights"][idx[s]], ts[s], Rs[s], False)
                got = c.get_output(s)
                for k in ("n_keypoints", "is_keyframe", "n_tracked", "n_detected", "n_measurements",
                          "tracking_status_mono", "tracking_status_stereo"):
                    assert got[k] == exp[k], (i, s, k)
                for k in ("landmarks", "keypoints", "versors", "lkf_T_k_stereo"):
                    assert np.array_equal(got[k], exp[k]), (i, s, k)
                if exp["is_keyframe"]:
                    for k in ("right_rect_xy", "right_status", "depth", "keypoints_3d", "meas_landmark"):
                        assert np.array_equal(got[k], exp[k]), (i, s, k)
                    kf[s] = idx[s]
    finally:
        c.close()
    # RGBD: intensity + uint16 depth on the device
    dp = abi.depth_params_default(abi.DEPTH_U16)
    dp.virtual_baseline, dp.min_depth, dp.depth_to_meters = 0.05, 0.3, 0.001
    TL = np.array(L.body_pose_cam).reshape(4, 4)
    camR = [TL[:3, :3].T @ Rb @ TL[:3, :3] for Rb in seq["body_R"]]
    fe = O.Frontend(L, L, p, depth=dp)
    c = F.Context(L, L, p, batch=1, frontend_type=abi.FRONTEND_RGBD, depth=dp)
    try:
        kf0 = 0
        for i in range(6):
            depth = _synthetic_depth(h, w, i, abi.DEPTH_U16, seed=i)
            dl = torch.from_numpy(np.ascontiguousarray(seq["lefts"][i])[None]).to(dev)
            dd = torch.from_numpy(depth.view(np.int16)[None].copy()).to(dev)
            Rk = camR[kf0].T @ camR[i]
            ts = int(seq["ts"][i])
            c.step_device(dl.data_ptr(), dd.data_ptr(), c.make_inputs([ts], [Rk], [0]))
            c.synchronize()
            dl.fill_(0x33)     # (the buffers belong to the caller again once the step has completed)
            dd.fill_(0)
            torch.cuda.synchronize()
            exp = fe.process(seq["lefts"][i], depth, ts, Rk, False)
            got = c.get_output(0)
            for k in ("n_keypoints", "is_keyframe", "n_measurements", "tracking_status_stereo"):
                assert got[k] == exp[k], (i, k)
            assert np.array_equal(got["keypoints"], exp["keypoints"])
            if exp["is_keyframe"]:
                for k in ("right_rect_xy", "right_status", "depth", "keypoints_3d"):
                    assert np.array_equal(got[k], exp[k]), (i, k)
                kf0 = i
    finally:
        c.close()


# ---------------------------------------------------------------------------------------------
# the C++ side of the drop-in boundary: a plain g++ host program over include/kvfe_adapter.hpp
# ---------------------------------------------------------------------------------------------
def _read_records(path):
    recs = []
    with open(path, "rb") as f:
        buf = f.read()
    o = 0
    while o < len(buf):
        tag = buf[o:o + 16].split(b"\0")[0].decode()
        nb = int(np.frombuffer(buf, np.int64, 1, o + 16)[0])
        recs.append((tag, buf[o + 24:o + 24 + nb]))
        o += 24 + nb
    return recs


def test_cpp_adapter_program_matches_oracle(seq, ocam, tmp_path):
    """tests/cpp/adapter_sequence.cpp uses libkvfe only through the reference's class and method names
    (kvfe_adapter.hpp: StereoCamera, UndistorterRectifier, FeatureDetector, Tracker, StereoMatcher,
    StereoVisionImuFrontend::spinOnce), compiled by g++ without HIP.  Component calls equal the ctypes
    path and the oracle; the six-frame front-end sequence (shipped Euroc parameters, useRANSAC: 1) equals
    the oracle front-end frame by frame, bit for bit; a contract violation surfaces as kvfe::Error."""
    import ctypes
    import subprocess
    cpp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp")
    exe = os.path.join(cpp, "adapter_sequence")
    subprocess.run(["make", "-C", cpp], check=True, capture_output=True)
    L, R = euroc_cams()
    p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=None)
    assert p.use_ransac == 1
    cfg = abi.Config()
    cfg.left, cfg.right, cfg.params, cfg.batch, cfg.device = L, R, p, 1, 0
    n = 6
    camR = _kf_rotations(seq["body_R"], ocam)
    fe = O.Frontend(L, R, p)
    H, W = seq["lefts"][0].shape
    exp_frames, inputs = [], []
    kf = 0
    c = F.Context(L, R, p, batch=1)
    try:
        for i in range(n):
            Rk = camR[kf].T @ camR[i]
            inputs.append(c.make_inputs([int(seq["ts"][i])], [Rk], [0])[0])
            e = fe.process(seq["lefts"][i], seq["rights"][i], int(seq["ts"][i]), Rk, False)
            exp_frames.append(e)
            if e["is_keyframe"]:
                kf = i
        with open(tmp_path / "in.bin", "wb") as f:
            f.write(bytes(cfg))
            f.write(np.array([n, W, H], np.int32).tobytes())
            for i in range(n):
                f.write(bytes(inputs[i]))
                f.write(np.ascontiguousarray(seq["lefts"][i]).tobytes())
                f.write(np.ascontiguousarray(seq["rights"][i]).tobytes())
        r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True,
                           text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        recs = _read_records(tmp_path / "out.bin")
        one = {t: b for t, b in recs if not t.startswith("f_")}

        # StereoCamera / UndistorterRectifier
        assert abs(np.frombuffer(one["baseline"], np.float64)[0] - 0.110078) < 1e-5  # testStereoMatcher.cpp:148
        assert np.array_equal(np.frombuffer(one["P1"], np.float64), np.array(ocam.rect.P1))
        assert np.array_equal(np.frombuffer(one["left_rect"], np.uint8).reshape(H, W),
                              ocam.rectify_image(0, seq["lefts"][0]))
        # FeatureDetector::featureDetection on an empty frame = GFTT + ANMS + cornerSubPix
        corners = np.frombuffer(one["corners"], np.float32).reshape(-1, 2)
        assert np.array_equal(corners, c.feature_detection(seq["lefts"][0], np.zeros((0, 2), np.float32),
                                                           p.detector.max_features_per_frame))
        assert len(corners) > 200
        assert np.array_equal(np.frombuffer(one["versors"], np.float64).reshape(-1, 3),
                              c.get_bearing_vectors(0, corners))
        # Tracker::featureTracking = predictor + calcOpticalFlowPyrLK against the oracle
        t = p.tracker
        Rk1 = np.array(inputs[1].keyframe_R_cur_frame).reshape(3, 3)
        init = c.predict_sparse_flow(corners, Rk1)
        exp, est, eerr, _ = O.calc_optical_flow_pyr_lk(seq["lefts"][0], seq["lefts"][1], corners, init,
                                                       t.klt_win_size, t.klt_max_level, t.klt_max_iter, t.klt_eps)
        assert np.array_equal(np.frombuffer(one["lk_px"], np.float32).reshape(-1, 2), exp)
        assert np.array_equal(np.frombuffer(one["lk_status"], np.uint8), est)
        assert np.array_equal(np.frombuffer(one["lk_err"], np.float32), eerr)
        # StereoMatcher::sparseStereoReconstruction
        sr = c.sparse_stereo_reconstruction(seq["lefts"][0], seq["rights"][0], corners)
        for tag, key, dt in (("st_lstat", "left_status", np.uint8), ("st_rstat", "right_status", np.uint8),
                             ("st_lrect", "left_rect_xy", np.float32), ("st_rrect", "right_rect_xy", np.float32),
                             ("st_depth", "depth", np.float64), ("st_3d", "keypoints_3d", np.float64)):
            assert np.array_equal(np.frombuffer(one[tag], dt), np.asarray(sr[key]).reshape(-1)), tag
        assert (np.frombuffer(one["st_rstat"], np.uint8) == 0).sum() > 100
        assert np.frombuffer(one["err_status"], np.int32)[0] == abi.KVFE_ERR_INVALID_ARG
    finally:
        c.close()

    # ---- round-2 methods called by the program -------------------------------------------------------------
    f32 = lambda tag: np.frombuffer(one[tag], np.float32).reshape(-1, 2)   # noqa: E731
    u8 = lambda tag: np.frombuffer(one[tag], np.uint8)                     # noqa: E731
    exy, est = ocam.undistort_rectify_left(corners)
    assert np.array_equal(u8("r2_url_st"), est) and np.array_equal(f32("r2_url_xy"), exy)
    cxy, cst = O.check_undistorted_rectified(ocam, 0, corners, ocam.undistort_keypoints(0, corners, True, True), 0.5)
    assert np.array_equal(u8("r2_chk_st"), cst) and np.array_equal(f32("r2_chk_xy"), cxy)
    lrect, rrect = ocam.rectify_image(0, seq["lefts"][0]), ocam.rectify_image(1, seq["rights"][0])
    assert np.array_equal(u8("r2_lrect").reshape(H, W), lrect) and np.array_equal(u8("r2_rrect").reshape(H, W), rrect)
    esr = ocam.sparse_stereo(seq["lefts"][0], seq["rights"][0], corners, p.stereo)
    assert np.array_equal(u8("r2_dep_st"), esr["right_status"]) and np.array_equal(f32("r2_dep_xy"), esr["right_rect_xy"])
    assert np.array_equal(np.frombuffer(one["r2_depth"], np.float64), esr["depth"])
    assert np.array_equal(f32("r2_rkps"), esr["right_xy"])

    def frame_of(pre):
        return dict(keypoints=f32(pre + "_kp"), landmarks=np.frombuffer(one[pre + "_lmk"], np.int64),
                    landmarks_age=np.frombuffer(one[pre + "_age"], np.int32),
                    versors=np.frombuffer(one[pre + "_ver"], np.float64).reshape(-1, 3))

    def same_frame(a, b):
        for k in ("keypoints", "landmarks", "landmarks_age", "versors"):
            assert np.array_equal(a[k], b[k]), k
    e0, ectr = O.feature_detection_frame(L, R, p, seq["lefts"][0], None, 0)
    same_frame(frame_of("r2_f0"), e0)
    Rk1 = np.array(inputs[1].keyframe_R_cur_frame).reshape(3, 3)
    eref, e1 = O.feature_tracking_frame(L, R, p, seq["lefts"][0], seq["lefts"][1], e0, Rk1)
    assert np.array_equal(np.frombuffer(one["r2_ref_lmk"], np.int64), eref)
    same_frame(frame_of("r2_f1t"), e1)
    e1d, ectr2 = O.feature_detection_frame(L, R, p, seq["lefts"][1], e1, ectr)
    same_frame(frame_of("r2_f1d"), e1d)
    assert np.frombuffer(one["r2_counter"], np.int64)[0] == ectr2
    # MonoVisionImuFrontend through the adapter
    mfe = O.Frontend(L, L, p, mono=True)
    mono_frames, curm = [], None
    for tag, b in recs:
        if tag == "m_head":
            curm = {}
            mono_frames.append(curm)
        if tag.startswith("m_"):
            curm[tag] = b
    assert len(mono_frames) == 3
    for i, g in enumerate(mono_frames):
        Rm = np.array(inputs[i].keyframe_R_cur_frame).reshape(3, 3)
        e = mfe.process(seq["lefts"][i], seq["lefts"][i], int(seq["ts"][i]), Rm, False)
        assert list(np.frombuffer(g["m_head"], np.int32)) == [e["n_keypoints"], e["is_keyframe"], e["n_tracked"],
                                                              e["n_measurements"]], i
        assert np.array_equal(np.frombuffer(g["m_lmk"], np.int64), e["landmarks"]), i
        assert np.array_equal(np.frombuffer(g["m_kp"], np.float32).reshape(-1, 2), e["keypoints"]), i
        if e["is_keyframe"]:
            assert np.array_equal(np.frombuffer(g["m_meas"], np.float64).reshape(-1, 3), e["meas_uL_uR_v"],
                                  equal_nan=True), i

    # Tracker::updateMap / pnp / outlierRejectionPnP through the adapter on every keyframe after the first: the map
    # is frame 0's 3-D points, the correspondences are gathered as Tracker::pnp(const StereoFrame&) does
    e0f = exp_frames[0]
    lmk_map = {int(l): e0f["keypoints_3d"][q] for q, l in enumerate(e0f["landmarks"])
               if e0f["right_status"][q] == abi.KP_VALID and l != -1}
    pp = abi.pnp_params_default()
    focal = 0.5 * (L.intrinsics[0] + L.intrinsics[1])
    pnp_recs = [(t, b) for t, b in recs if t in ("p_status", "p_pose")]
    kf_after = [e for e in exp_frames[1:] if e["is_keyframe"]]
    assert len(pnp_recs) == 2 * len(kf_after) and len(kf_after) >= 1
    for j, e in enumerate(kf_after):
        sel = [q for q, l in enumerate(e["landmarks"])
               if e["left_status"][q] == abi.KP_VALID and l != -1 and int(l) in lmk_map]
        fb = e["keypoints_3d"][sel]
        pw = np.array([lmk_map[int(e["landmarks"][q])] for q in sel]).reshape(-1, 3)
        exp_pnp = O.pnp(fb, pw, focal, p.tracker, pp)
        st = np.frombuffer(pnp_recs[2 * j][1], np.int32)
        assert st[0] == exp_pnp["status"] and st[1] == len(lmk_map), (j, st, exp_pnp["status"], len(sel))
        assert np.array_equal(np.frombuffer(pnp_recs[2 * j + 1][1], np.float64).reshape(3, 4), exp_pnp["pose"]), j

    # StereoVisionImuFrontend::spinOnce, frame by frame against the oracle front-end
    frames, cur = [], None
    for tag, b in recs:
        if tag == "f_head":
            cur = {}
            frames.append(cur)
        if tag.startswith("f_"):
            cur[tag] = b
    assert len(frames) == n
    for i, (g, e) in enumerate(zip(frames, exp_frames)):
        head = np.frombuffer(g["f_head"], np.int32)
        assert list(head) == [e["n_keypoints"], e["is_keyframe"], e["n_tracked"], e["n_detected"],
                              e["n_measurements"], e["frame_id"]], (i, head)
        assert np.array_equal(np.frombuffer(g["f_lmk"], np.int64), e["landmarks"]), i
        assert np.array_equal(np.frombuffer(g["f_age"], np.int32), e["landmarks_age"]), i
        assert np.array_equal(np.frombuffer(g["f_kp"], np.float32).reshape(-1, 2), e["keypoints"]), i
        assert np.array_equal(np.frombuffer(g["f_versors"], np.float64).reshape(-1, 3), e["versors"]), i
        if e["is_keyframe"]:
            assert np.array_equal(np.frombuffer(g["f_lrect"], np.float32).reshape(-1, 2), e["left_rect_xy"]), i
            assert np.array_equal(np.frombuffer(g["f_lstat"], np.uint8), e["left_status"]), i
            assert np.array_equal(np.frombuffer(g["f_rrect"], np.float32).reshape(-1, 2), e["right_rect_xy"]), i
            assert np.array_equal(np.frombuffer(g["f_rstat"], np.uint8), e["right_status"]), i
            assert np.array_equal(np.frombuffer(g["f_depth"], np.float64), e["depth"]), i
            assert np.array_equal(np.frombuffer(g["f_3d"], np.float64).reshape(-1, 3), e["keypoints_3d"]), i
            assert np.array_equal(np.frombuffer(g["f_mlmk"], np.int64), e["meas_landmark"]), i
            assert np.array_equal(np.frombuffer(g["f_meas"], np.float64).reshape(-1, 3), e["meas_uL_uR_v"],
                                  equal_nan=True), i
        trk = np.frombuffer(g["f_trk"], np.int32)
        assert list(trk) == [e[k] for k in ("tracking_status_mono", "tracking_status_stereo", "nr_mono_putatives",
                                            "nr_mono_inliers", "nr_stereo_putatives", "nr_stereo_inliers")], (i, trk)
        assert np.array_equal(np.frombuffer(g["f_Tmono"], np.float64).reshape(3, 4), e["lkf_T_k_mono"]), i
        assert np.array_equal(np.frombuffer(g["f_Tstereo"], np.float64).reshape(3, 4), e["lkf_T_k_stereo"]), i
    assert [e["is_keyframe"] for e in exp_frames] == [1, 0, 0, 0, 1, 0]
