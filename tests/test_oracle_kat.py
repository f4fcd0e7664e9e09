"""Pins the CPU oracle against the reference's own known-answer tests.

Each test cites the reference test it restates (paths relative to /root/reference).
These run on CPU (no GPU needed) and gate every GPU parity claim: the HIP path is
compared against this oracle, and the oracle is compared against the reference's KATs here.
"""
import os

import numpy as np
import pytest
from PIL import Image

import oracle_lib as O
from kimera_vio_amd import _abi as abi
from kimera_vio_amd import params as P

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gray(name):
    return np.array(Image.open(os.path.join(G, name)).convert("L"))


@pytest.fixture(scope="module")
def fisheye_img():
    return gray("left_fisheye_img_0.png")


def detect(img, yaml_name, **overrides):
    d = P.load_detector_params(os.path.join(G, "ForFeatureDetector", yaml_name))
    for k, v in overrides.items():
        setattr(d, k, v)
    out, raw = O.feature_detection(img, np.zeros((0, 2), np.float32), d.max_features_per_frame, d)
    return out, raw, d


def bin_counts(kps, d, shape):
    rows, cols = shape
    brs = np.float32(rows) / np.float32(d.nr_vertical_bins)
    bcs = np.float32(cols) / np.float32(d.nr_horizontal_bins)
    cnt = np.zeros((d.nr_vertical_bins, d.nr_horizontal_bins), int)
    for x, y in kps:
        cnt[int(np.float32(y) / brs), int(np.float32(x) / bcs)] += 1
    return cnt


# ---- tests/testFeatureDetector.cpp -------------------------------------------------------------
def test_detector_no_nms_393(fisheye_img):
    """tests/testFeatureDetector.cpp:25-50 FeatureDetectorNoNonMaxSuppression -> 393 keypoints."""
    out, raw, _ = detect(fisheye_img, "frontendParams-noNMS.yaml")
    assert len(out) == 393


def test_detector_no_nms_quality_400(fisheye_img):
    """tests/testFeatureDetector.cpp:53-80 (quality 1e-10 -> max_nr_keypoints_before_anms = 400)."""
    out, _, _ = detect(fisheye_img, "frontendParams-noNMS.yaml", quality_level=1e-10)
    assert len(out) == 400


def test_detector_anms_topn_300(fisheye_img):
    """tests/testFeatureDetector.cpp:83-107 FeatureDetector_ANMS_TopN -> 300."""
    out, _, _ = detect(fisheye_img, "frontendParams-NMS-TopN.yaml")
    assert len(out) == 300


@pytest.mark.parametrize("policy", [abi.SORTIDX_LIBSTDCXX, abi.SORTIDX_STABLE])
def test_detector_anms_binning_20(fisheye_img, policy):
    """tests/testFeatureDetector.cpp:110-149: 20 keypoints, exactly one per 5x4 bin.
    NB: subpixel refinement is ON in this test, as in the reference."""
    out, _, d = detect(fisheye_img, "frontendParams-NMS-Binning.yaml", sortidx_policy=policy)
    assert len(out) == 20
    assert np.array_equal(bin_counts(out, d, fisheye_img.shape), np.ones((5, 4), int))


@pytest.mark.parametrize("policy", [abi.SORTIDX_LIBSTDCXX, abi.SORTIDX_STABLE])
def test_detector_anms_binning_200(fisheye_img, policy):
    """tests/testFeatureDetector.cpp:152-199: 200 keypoints, 10 per bin."""
    out, _, d = detect(fisheye_img, "frontendParams-NMS-Binning.yaml", max_features_per_frame=200,
                       quality_level=1e-10, enable_subpixel_corner_refinement=0,
                       sortidx_policy=policy)
    assert len(out) == 200
    assert np.array_equal(bin_counts(out, d, fisheye_img.shape), 10 * np.ones((5, 4), int))


def test_detector_anms_binning_masked_140(fisheye_img):
    """tests/testFeatureDetector.cpp:202-258: 140 keypoints, 10 per active bin, 0 in masked bins."""
    out, _, d = detect(fisheye_img, "frontendParams-NMS-Binning2.yaml", quality_level=1e-10,
                       enable_subpixel_corner_refinement=0)
    assert len(out) == 140
    expected = 10 * np.ones((5, 4), int)
    expected[0, :] = 0
    expected[3, 0] = 0
    expected[3, 2] = 0
    assert np.array_equal(bin_counts(out, d, fisheye_img.shape), expected)


# ---- tests/testFrame.cpp / tests/testUtilsOpenCV.cpp -------------------------------------------
def test_extract_corners_chessboard_63():
    """tests/testFrame.cpp:56-79, tests/testUtilsOpenCV.cpp:386-427: default ExtractCorners
    (100, 0.01, 10, 3) on chessboard.png -> exactly 63 corners; blank image -> 0."""
    ch = gray("chessboard.png")
    xy, _ = O.good_features_to_track(ch, 100, 0.01, 10, 3)
    assert len(xy) == 63
    blank = np.full((300, 240), 255, np.uint8)
    xy, _ = O.good_features_to_track(blank, 100, 0.01, 10, 3)
    assert len(xy) == 0


# ---- cv::sortIdx permutation ---------------------------------------------------------------------
@pytest.mark.parametrize("n", [0, 1, 2, 15, 16, 17, 18, 33, 100, 393, 400, 777, 1999, 2000, 5000])
def test_sortidx_permutation_matches_libstdcxx(n):
    """NonMaximumSuppression.cpp:50-60: the explicit introsort restatement equals what
    std::sort(always-false comparator)+reverse produces with this toolchain's libstdc++."""
    a = O.sortidx_permutation(n)
    b = O.sortidx_permutation_stdsort(n)
    assert np.array_equal(a, b)
    assert sorted(a.tolist()) == list(range(n))
    if n > 1:
        assert not np.array_equal(a, np.arange(n))  # it is NOT the identity (quality order is lost)


# ---- tests/testStereoMatcher.cpp / tests/testStereoCamera.cpp ------------------------------------
@pytest.fixture(scope="module")
def euroc_cam():
    L = P.load_camera_params(os.path.join(G, "sensorLeft.yaml"))
    R = P.load_camera_params(os.path.join(G, "sensorRight.yaml"))
    return O.Camera(L, R)


def test_stereo_baseline(euroc_cam):
    """tests/testStereoMatcher.cpp:148: baseline 0.110078 +- 1e-5."""
    assert abs(euroc_cam.rect.baseline - 0.110078) < 1e-5


def test_rectification_structure(euroc_cam):
    """tests/testStereoCamera.cpp:374-440: P1[:3,:3] == P2[:3,:3]; P2[:,3] = -K [b,0,0];
    rectified relative pose is a pure x-translation."""
    r = euroc_cam.rect
    P1 = np.array(r.P1).reshape(3, 4)
    P2 = np.array(r.P2).reshape(3, 4)
    assert np.allclose(P1[:, :3], P2[:, :3], atol=1e-12)
    assert np.allclose(P2[:, 3], -P1[:, :3] @ np.array([r.baseline, 0, 0]), atol=1e-9)
    R1 = np.array(r.R1).reshape(3, 3)
    R2 = np.array(r.R2).reshape(3, 3)
    assert np.allclose(R1 @ R1.T, np.eye(3), atol=1e-9)
    TL = np.array(euroc_cam.left.body_pose_cam).reshape(4, 4)
    TR = np.array(euroc_cam.right.body_pose_cam).reshape(4, 4)
    rel = np.linalg.inv(TL) @ TR  # camL_Pose_camR
    # rectified: camLrect_Pose_camRrect = R1 * rel * R2^T
    Rrel = R1 @ rel[:3, :3] @ R2.T
    trel = R1 @ rel[:3, 3]
    assert np.allclose(Rrel, np.eye(3), atol=1e-6)
    assert abs(trel[0] - r.baseline) < 1e-6 and abs(trel[1]) < 1e-6 and abs(trel[2]) < 1e-6


def test_rectification_rotation_matches_published_euroc(euroc_cam):
    """Third-party pin: the EuRoC LEFT.R rectification rotation published in ORB-SLAM2's
    Examples/Stereo/EuRoC.yaml (computed with cv::stereoRectify from the same sensor.yaml
    extrinsics).  Focal/principal point are OpenCV-version dependent and are not compared."""
    published = np.array([0.999966347530033, -0.001422739138722922, 0.008079580483432283,
                          0.001365741834644127, 0.9999741760894847, 0.007055629199258132,
                          -0.008089410156878961, -0.007044357138835809, 0.9999424675829176])
    assert np.allclose(np.array(euroc_cam.rect.R1), published, atol=2e-10)


def _translate_x_nearest(img, dist):
    """cv::warpPerspective(img, [[1,0,dist],[0,1,0],[0,0,1]], INTER_NEAREST, BORDER_CONSTANT 0)."""
    out = np.zeros_like(img)
    d = int(dist)
    if d < 0:
        out[:, : img.shape[1] + d] = img[:, -d:]
    elif d > 0:
        out[:, d:] = img[:, : img.shape[1] - d]
    else:
        out[:] = img
    return out


def test_get_right_keypoints_rectified_849_of_900(euroc_cam):
    """tests/testStereoMatcher.cpp:272-388: right = left shifted by {-20,-10,-5} px;
    849 of the 900 tested keypoints are VALID inside-frame matches within 0.5 px."""
    left = gray("left_img_0.png")
    kps, _ = O.good_features_to_track(left, 100, 0.01, 10, 3)
    assert len(kps) == 100
    sp = P.default_frontend_params().stereo
    rows, cols = left.shape
    count_valid = total = 0
    for offset in (-20, -10, -5):
        right = _translate_x_nearest(left, offset)
        acc = np.zeros((0, 2), np.float32)
        for t in range(2):
            add = kps if t == 0 else np.stack([np.round(kps[:, 0]), np.round(kps[:, 1])], 1)
            acc = np.concatenate([acc, add.astype(np.float32)])  # not cleared between t, as in the ref
            st = np.zeros(len(acc), np.uint8)
            cam = euroc_cam
            fx_saved = cam.rect.P1[0]
            cam.rect.P1[0] = 458.654  # the test passes fx = 458.654 explicitly
            try:
                rxy, rst, _ = cam.get_right_keypoints_rectified(left, right, acc, st, sp)
            finally:
                cam.rect.P1[0] = fx_saved
            for i in range(len(acc)):
                total += 1
                y_left = float(acc[i, 1])
                x_exp = float(acc[i, 0]) + offset
                x_act = float(rxy[i, 0])
                stripe_rows = 11 + 4
                if y_left <= (stripe_rows - 1) // 2 or y_left + (stripe_rows - 1) // 2 >= rows:
                    assert rst[i] == abi.KP_NO_RIGHT_RECT
                elif x_exp >= 50 and x_exp + 50 < cols:
                    assert rst[i] == abi.KP_VALID
                    assert abs(x_exp - x_act) < 0.5
                    assert abs(float(acc[i, 1]) - float(rxy[i, 1])) < 0.5
                    count_valid += 1
    assert total == 900
    assert count_valid == 849


def test_sparse_stereo_real_pair(euroc_cam):
    """tests/testStereoMatcher.cpp:135-266 (sparseStereoReconstruction on left/right_img_0):
    more than 68 of 100 corners VALID; 3D point = versor*depth/versor.z; reprojection <= 1 px."""
    left, right = gray("left_img_0.png"), gray("right_img_0.png")
    kps, _ = O.good_features_to_track(left, 100, 0.01, 10, 3)
    sp = P.default_frontend_params().stereo
    res = euroc_cam.sparse_stereo(left, right, kps, sp)
    valid = res["right_status"] == abi.KP_VALID
    assert valid.sum() > 68
    versors = euroc_cam.bearing_vectors(0, kps)
    P1 = np.array(euroc_cam.rect.P1).reshape(3, 4)
    for i in np.nonzero(valid)[0]:
        v = versors[i]
        assert abs(np.linalg.norm(v) - 1) < 1e-9
        p3 = res["keypoints_3d"][i]
        assert np.allclose(p3, v * res["depth"][i] / v[2], rtol=1e-12)
        assert abs(p3[2] - res["depth"][i]) < 1e-9
        px = P1 @ np.append(p3, 1.0)
        px = px[:2] / px[2]
        assert np.max(np.abs(px - res["left_rect_xy"][i])) <= 1.0
        # depth = fx*b / disparity
        disp = float(res["left_rect_xy"][i, 0]) - float(res["right_rect_xy"][i, 0])
        assert abs(res["depth"][i] - P1[0, 0] * euroc_cam.rect.baseline / disp) < 1e-9


# ---- tests/testUndistortRectifier.cpp ----------------------------------------------------------
def test_undistort_rectify_roundtrip(euroc_cam):
    """tests/testUndistortRectifier.cpp:87-145: an 8x10 grid of rectified pixels mapped through
    the maps (distort) and back through undistortRectifyKeypoints lands within 1 px."""
    mx, my = euroc_cam.maps(0)
    h, w = mx.shape
    pts = [(x, y) for y in np.linspace(40, h - 40, 8) for x in np.linspace(40, w - 40, 10)]
    rect_px = np.array(pts, np.float32)
    dist_px = np.array([[mx[int(round(y)), int(round(x))], my[int(round(y)), int(round(x))]]
                        for x, y in pts], np.float32)
    back = euroc_cam.undistort_keypoints(0, dist_px, True, True)
    assert np.max(np.abs(back - np.round(rect_px))) < 1.0
    xy, st = euroc_cam.undistort_rectify_left(dist_px)
    assert np.all(st == abi.KP_VALID)


def test_distort_matches_radtan_model(euroc_cam):
    """tests/testUndistortRectifier.cpp:152-221: map lookup == gtsam::Cal3DS2::uncalibrate of
    R1^T P1^-1 [u v 1] to <= 1e-3 px (forward radtan model evaluated in float64 here)."""
    mx, my = euroc_cam.maps(0)
    c = euroc_cam.left
    fx, fy, cx, cy = list(c.intrinsics)
    k1, k2, p1, p2 = list(c.distortion)[:4]
    R1 = np.array(euroc_cam.rect.R1).reshape(3, 3)
    P1 = np.array(euroc_cam.rect.P1).reshape(3, 4)[:, :3]
    rng = np.random.RandomState(0)
    for _ in range(200):
        u, v = rng.randint(0, c.width), rng.randint(0, c.height)
        xn = R1.T @ np.linalg.inv(P1) @ np.array([u, v, 1.0])
        x, y = xn[0] / xn[2], xn[1] / xn[2]
        r2 = x * x + y * y
        kr = 1 + k1 * r2 + k2 * r2 * r2
        xd = x * kr + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        yd = y * kr + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        assert abs(fx * xd + cx - mx[v, u]) < 1e-3
        assert abs(fy * yd + cy - my[v, u]) < 1e-3


# ---- tests/testStereoVisionImuFrontend.cpp (bootstrap on the synthetic pair) --------------------
def _load_corners(name):
    vals = np.loadtxt(os.path.join(G, "ForStereoTracker", name))
    return vals.reshape(-1, 2) if vals.ndim == 1 else vals


def _load_pts(name):
    vals = np.loadtxt(os.path.join(G, "ForStereoTracker", name), skiprows=1)
    return vals


def test_process_first_frame_synthetic_pair():
    """tests/testStereoVisionImuFrontend.cpp:454-672 processFirstFrame on img_distort_{left,right}.png
    with the test's parameter overrides (default FrontendParams; min_distance = (int)0.05 = 0,
    quality 0.1, max_point_dist 500, templ_cols 9, subpixel stereo refinement on; ANMS = the class
    default RangeTree): every detected keypoint is within 3 px (inf-norm) of a golden corner, left
    rectified keypoints within 2 px of undistortPoints(golden corner), all stereo statuses VALID,
    right keypoints within 2 px of the golden right corners, depth within +-4 of the golden depth."""
    L = P.load_camera_params(os.path.join(G, "ForStereoTracker", "camLeft.yaml"))
    R = P.load_camera_params(os.path.join(G, "ForStereoTracker", "camRight.yaml"))
    left = gray("ForStereoTracker/img_distort_left.png")
    right = gray("ForStereoTracker/img_distort_right.png")
    p = P.default_frontend_params()
    p.detector.min_distance = int(0.05)
    p.detector.quality_level = 0.1
    p.stereo.max_point_dist = 500
    p.stereo.templ_cols = 9
    p.stereo.subpixel_refinement = 1
    fe = O.Frontend(L, R, p)
    out = fe.process(left, right, 0)
    corners_left = _load_pts("corners_normal_left.txt")
    corners_right = _load_pts("corners_normal_right.txt")
    depth_gt = _load_pts("depth_left.txt").reshape(-1)
    assert len(corners_left) == 35 and len(depth_gt) == 35
    n = out["n_keypoints"]
    assert n > 0 and out["has_stereo"] and out["is_keyframe"] == 1
    assert np.all(out["landmarks_age"] == 1) and np.all(out["landmarks"] >= 0)
    cam = O.Camera(L, R)
    idx = []
    for i in range(n):
        d = np.max(np.abs(corners_left - out["keypoints"][i]), axis=1)
        hits = np.nonzero(d < 3.0)[0]
        assert len(hits) > 0, f"keypoint {out['keypoints'][i]} has no golden corner within 3 px"
        idx.append(int(hits[0]))
    idx = np.array(idx)
    assert np.max(np.abs(corners_left[idx] - out["keypoints"])) <= 2.0
    exp_rect = cam.undistort_keypoints(0, corners_left[idx].astype(np.float32), True, True)
    assert np.max(np.abs(exp_rect - out["left_rect_xy"])) <= 2.0
    assert np.all(out["left_status"] == abi.KP_VALID)
    assert np.all(out["right_status"] == abi.KP_VALID)
    assert np.max(np.abs(corners_right[idx] - out["right_xy"])) <= 2.0
    exp_rrect = cam.undistort_keypoints(1, corners_right[idx].astype(np.float32), True, True)
    assert np.max(np.abs(exp_rrect - out["right_rect_xy"])) <= 2.0
    assert np.max(np.abs(out["keypoints_3d"][:, 2] - depth_gt[idx])) <= 4.0
    v = out["versors"]
    v_exp = v * (depth_gt[idx] / v[:, 2])[:, None]
    assert np.max(np.linalg.norm(v_exp - out["keypoints_3d"], axis=1)) < 5.0


def test_equalize_hist_properties():
    """cv::equalizeHist (UtilsOpenCV.cpp:398-401, equalizeImage: 1).  The reference holds no golden
    image for it (parity unpinned); the restatement is checked on the properties the algorithm
    defines: monotone LUT starting at 0 for the first occupied bin, 255 for the last occupied bin,
    the LUT formula itself in float32, and the constant-image special case."""
    img = np.array(Image.open(os.path.join(G, "left_img_0.png")).convert("L"))
    out = O.equalize_hist(img)
    hist = np.bincount(img.ravel(), minlength=256)
    first = int(np.nonzero(hist)[0][0])
    last = int(np.nonzero(hist)[0][-1])
    scale = np.float32(255.0) / np.float32(img.size - hist[first])
    csum = np.cumsum(hist) - hist[:first + 1].sum()
    lut = np.zeros(256, np.int64)
    for i in range(first + 1, 256):
        lut[i] = int(np.clip(np.rint(np.float32(csum[i]) * scale), 0, 255))
    assert np.array_equal(out, lut[img].astype(np.uint8))
    assert out[img == first].max() == 0 and out[img == last].min() == 255
    assert np.all(np.diff(lut[first:last + 1]) >= 0)
    const = np.full((48, 64), 77, np.uint8)
    assert np.array_equal(O.equalize_hist(const), const)


def _anms_reference_python(kps, need, cols, rows, kind):
    """independent restatement of the Bailo et al. binary search (anms/anms.cpp) for the two variants
    whose covering relation is a plain pixel predicate"""
    n = len(kps)
    x = kps[:, 0].astype(int)
    y = kps[:, 1].astype(int)
    exp1 = rows + cols + 2 * need
    exp2 = 4 * cols + 4 * need + 4 * rows * need + rows * rows + cols * cols - 2 * rows * cols + 4 * rows * cols * need
    exp3 = np.sqrt(float(exp2))
    exp4 = need - 1
    rnd = lambda v: np.floor(abs(v) + 0.5) * np.sign(v)
    high = int(max(-rnd((exp1 + exp3) / exp4), -rnd((exp1 - exp3) / exp4)))
    low = int(np.floor(np.sqrt(n / need)))
    kmin = int(rnd(np.float32(need) - np.float32(need) * np.float32(0.1)))
    kmax = int(rnd(np.float32(need) + np.float32(need) * np.float32(0.1)))
    result, prev = [], -1
    while True:
        r = low + int((high - low) / 2)
        if r == prev or low > high:
            return result
        inc = np.ones(n, bool)
        res = []
        for i in range(n):
            if not inc[i]:
                continue
            res.append(i)
            dx, dy = x - x[i], y - y[i]
            cov = (dx * dx + dy * dy < r * r) if kind == "kdtree" else ((abs(dx) <= r) & (abs(dy) <= r))
            inc &= ~cov
            inc[i] = False
        result = res
        if kmin <= len(res) <= kmax:
            return result
        if len(res) < kmin:
            high = r - 1
        else:
            low = r + 1
        prev = r


@pytest.mark.parametrize("anms_type,name", [(2, "sdc"), (3, "kdtree"), (4, "rangetree"), (5, "ssc")])
def test_anms_radius_search_variants(anms_type, name):
    """anms::Sdc / KdTree / RangeTree / Ssc (anms/anms.cpp:83-436): no golden numbers in the reference
    (its detector tests use TopN and Binning), so: the output is an order-preserving subset of the
    sortIdx-permuted input, its size lands in [K(1-tol), K(1+tol)] on a well-populated image, the kept
    corners respect the spacing the variant implies, and KdTree / RangeTree equal an independent
    Python restatement."""
    img = np.array(Image.open(os.path.join(G, "left_fisheye_img_0.png")).convert("L"))
    h, w = img.shape
    kps, _ = O.good_features_to_track(img, 2000, 0.001, 10, 3)
    p = P.default_frontend_params().detector
    p.non_max_suppression_type = anms_type
    perm = O.sortidx_permutation(len(kps), p.sortidx_policy)
    sorted_kps = kps[perm]
    for need in (100, 300):
        out = O.suppress_non_max(kps, need, w, h, p)
        kmin, kmax = round(need * 0.9), round(need * 1.1)
        assert kmin <= len(out) <= kmax, (name, need, len(out))
        # order-preserving subset of the permuted input
        pos = {tuple(k): i for i, k in enumerate(sorted_kps.tolist())}
        idx = [pos[tuple(k)] for k in out.tolist()]
        assert idx == sorted(idx) and len(set(idx)) == len(idx)
        d = out[:, None, :] - out[None, :, :]
        cheb = np.abs(d).max(-1) + np.eye(len(out)) * 1e9
        assert cheb.min() >= 1
        if name in ("kdtree", "rangetree"):
            exp = _anms_reference_python(sorted_kps, need, w, h, name)
            assert idx == exp, (name, need)
    # numRetPoints < 2: exp4 = 0 in the reference's closed form (UB there) -> nothing is returned
    if name != "sdc":
        assert len(O.suppress_non_max(kps, 1, w, h, p)) == 0


def test_anms_brown():
    """anms::BrownANMS (anms/anms.cpp:51-81; no golden numbers upstream): keypoint i gets the distance to the nearest
    keypoint BEFORE it in the (unsorted) input, the pairs are std::sort-ed by radius descending and the first
    numRetPoints are returned.  Checked against a numpy restatement of the radii: the output radii are non-increasing,
    start with keypoint 0 (FLT_MAX), every returned radius >= every dropped one, and where radii are distinct the
    order is forced.  numRetPoints > size returns the input unchanged."""
    img = np.array(Image.open(os.path.join(G, "left_fisheye_img_0.png")).convert("L"))
    h, w = img.shape
    kps, _ = O.good_features_to_track(img, 1200, 0.001, 10, 3)
    p = P.default_frontend_params().detector
    p.non_max_suppression_type = 1
    n = len(kps)
    assert n > 600
    d = kps[:, None, :].astype(np.float32) - kps[None, :, :].astype(np.float32)
    dist = np.sqrt((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]).astype(np.float32)).astype(np.float32)
    radius = np.full(n, np.finfo(np.float32).max, np.float32)
    for i in range(1, n):
        radius[i] = dist[i, :i].min()
    pos = {tuple(k): i for i, k in enumerate(kps.tolist())}
    for need in (1, 50, 300, n):
        out = O.suppress_non_max(kps, need, w, h, p)
        assert len(out) == need
        idx = [pos[tuple(k)] for k in out.tolist()]
        assert len(set(idx)) == need and idx[0] == 0
        r = radius[idx]
        assert np.all(r[:-1] >= r[1:])
        rest = np.setdiff1d(np.arange(n), idx)
        if len(rest):
            assert r[-1] >= radius[rest].max()
        uniq, counts = np.unique(radius, return_counts=True)
        single = set(uniq[counts == 1].tolist())
        order = np.argsort(-radius, kind="stable")
        for k in range(need):
            if float(radius[order[k]]) in single:
                assert idx[k] == order[k]
    assert np.array_equal(O.suppress_non_max(kps, n + 1, w, h, p), kps)
    assert len(O.suppress_non_max(kps, 0, w, h, p)) == 0


# --------------------------------------------------------------------------- equidistant (cv::fisheye)
def _fisheye_cams():
    return (P.load_camera_params(os.path.join(G, "left_sensor_fisheye.yaml")),
            P.load_camera_params(os.path.join(G, "right_sensor_fisheye.yaml")))


def _fisheye_project(cam, rays):
    """forward equidistant model: theta_d = theta (1 + k1 theta^2 + ... + k4 theta^8)"""
    fx, fy, cx, cy = (cam.intrinsics[i] for i in range(4))
    k = [cam.distortion[i] for i in range(4)]
    x, y = rays[:, 0] / rays[:, 2], rays[:, 1] / rays[:, 2]
    r = np.hypot(x, y)
    th = np.arctan(r)
    thd = th * (1 + k[0] * th**2 + k[1] * th**4 + k[2] * th**6 + k[3] * th**8)
    sc = np.where(r > 0, thd / np.maximum(r, 1e-300), 1.0)
    return np.stack([fx * x * sc + cx, fy * y * sc + cy], 1)


def test_fisheye_rectification_and_undistortion():
    """distortion_model: equidistant (reference: cv::fisheye::stereoRectify / initUndistortRectifyMap /
    undistortPoints; its own tests of this model are DISABLED_, so parity is unpinned).  Checked here:
    the structure every stereo rectification must have (tests/testStereoCamera.cpp:374-440 for radtan),
    round trips against the forward equidistant model, map / keypoint consistency, and product host math
    == oracle."""
    L, R = _fisheye_cams()
    cam = O.Camera(L, R)
    rect = cam.rect
    R1, R2 = np.array(rect.R1).reshape(3, 3), np.array(rect.R2).reshape(3, 3)
    P1, P2 = np.array(rect.P1).reshape(3, 4), np.array(rect.P2).reshape(3, 4)
    for Rm in (R1, R2):
        assert np.allclose(Rm @ Rm.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(Rm) - 1) < 1e-12
    assert np.array_equal(P1[:, :3], P2[:, :3]) and P1[0, 0] == P1[1, 1]
    TL = np.array(L.body_pose_cam).reshape(4, 4)
    TR = np.array(R.body_pose_cam).reshape(4, 4)
    T_lr = np.linalg.inv(TL) @ TR
    assert abs(rect.baseline - np.linalg.norm(T_lr[:3, 3])) < 1e-9
    assert abs(P2[0, 3] + P1[0, 0] * rect.baseline) < 1e-9          # P2[:,3] = -f b
    # relative pose after rectification is a pure x translation
    t_rect = R2 @ (-(T_lr[:3, :3].T @ T_lr[:3, 3]))
    assert abs(t_rect[1]) < 1e-9 and abs(t_rect[2]) < 1e-9 and t_rect[0] < 0
    # round trip: forward model -> undistortPoints (no R, no P) gives the normalised ray back
    rng = np.random.default_rng(0)
    rays = np.stack([rng.uniform(-0.8, 0.8, 200), rng.uniform(-0.5, 0.5, 200), np.ones(200)], 1)
    px = _fisheye_project(L, rays).astype(np.float32)
    und = cam.undistort_keypoints(0, px, use_R=False, use_P=False)
    assert np.allclose(und, rays[:, :2], atol=2e-5)
    # map consistency: the rectification map at a rectified pixel is the forward projection of its ray
    mx, my = cam.maps(0)
    K1inv = np.linalg.inv(P1[:, :3])
    for (u, v) in ((100, 80), (376, 240), (700, 400)):
        ray = R1.T @ (K1inv @ np.array([u, v, 1.0]))
        exp = _fisheye_project(L, ray[None])[0]
        assert abs(mx[v, u] - exp[0]) < 1e-3 and abs(my[v, u] - exp[1]) < 1e-3
    # product host math (kvfe_compute_rectification / _maps) == oracle
    from kimera_vio_amd import frontend as F
    prod = F.compute_rectification(L, R)
    for name in ("R1", "R2", "P1", "P2", "Q"):
        assert np.allclose(np.array(getattr(prod, name)), np.array(getattr(rect, name)), rtol=0, atol=1e-12), name
    assert abs(prod.baseline - rect.baseline) < 1e-14
    pmx, pmy = F.compute_undistort_rectify_maps(L, prod.R1, prod.P1)
    assert np.abs(pmx - mx).max() < 1e-3 and np.abs(pmy - my).max() < 1e-3


def test_get_smart_stereo_measurements():
    """StereoVisionImuFrontendFixture.getSmartStereoMeasurements (tests/testStereoVisionImuFrontend.cpp:360-452):
    12 valid keypoints, 12 without a right match, 12 without a landmark -> 24 measurements (landmark i = i),
    uL / v from the rectified left keypoint, uR from the right one or NaN, no landmark twice."""
    rng = np.random.RandomState(5)
    n_valid = n_missing = n_invalid = 12
    n = n_valid + n_missing + n_invalid
    uL = rng.randint(0, 800, n).astype(np.float32)
    uR = uL + rng.randint(-40, 40, n).astype(np.float32)
    v = rng.randint(0, 600, n).astype(np.float32)
    lmk = np.array(list(range(n_valid + n_missing)) + [-1] * n_invalid, np.int64)
    VALID, NO_RIGHT_RECT = 0, 2
    rstat = np.array([VALID] * n_valid + [NO_RIGHT_RECT] * n_missing + [VALID] * n_invalid, np.uint8)
    ml, mv = O.get_smart_stereo_measurements(lmk, np.stack([uL, v], 1), rstat, np.stack([uR, v], 1))
    assert len(ml) == n_valid + n_missing
    assert len(set(ml.tolist())) == len(ml)
    for lid, (muL, muR, mvv) in zip(ml, mv):
        assert muL == uL[lid] and mvv == v[lid]
        if rstat[lid] == VALID:
            assert muR == uR[lid]
        else:
            assert np.isnan(muR)
    # use_stereo_tracking off (StereoVisionImuFrontend.cpp:509): every uR is NaN
    _, mono = O.get_smart_stereo_measurements(lmk, np.stack([uL, v], 1), rstat, np.stack([uR, v], 1), False)
    assert np.isnan(mono[:, 1]).all()


def test_get_depth_from_rectified_matches(euroc_cam):
    """StereoMatcherFixture.getDepthFromRectifiedMatches (tests/testStereoMatcher.cpp:394-502): 3-D points at
    11 depth-to-baseline ratios x 9 image positions projected with P1 / P2; depth = fx b / disparity within 1e-3,
    0 outside [min_point_dist, max_point_dist] of the default StereoMatchingParams (0.1, 15), 0 for every
    keypoint pair that is not VALID / VALID and for a negative disparity."""
    cam = euroc_cam
    P1 = np.array(cam.rect.P1).reshape(3, 4)
    P2 = np.array(cam.rect.P2).reshape(3, 4)
    b = cam.rect.baseline
    sp = abi.StereoParams()
    sp.min_point_dist, sp.max_point_dist = 0.1, 15.0       # StereoMatchingParams.h:50-52
    lxy, rxy, expd = [], [], []
    for ratio in (0.5, 1.0, 2.0, 3.0, 5.0, 10.0, 15.0, 20.0, 30.0, 50.0, 100.0):
        for x2d in (-0.2, 0, 0.2):
            for y2d in (-0.2, 0, 0.2):
                depth = ratio * b
                X = np.array([x2d * depth, y2d * depth, depth, 1.0])
                pl, pr = P1 @ X, P2 @ X
                lxy.append(pl[:2] / pl[2])
                rxy.append(pr[:2] / pr[2])
                expd.append(depth)
    VALID, NO_LEFT_RECT, NO_RIGHT_RECT, NO_DEPTH, FAILED_ARUN = range(5)
    n = len(expd)
    ls, rs = [VALID] * n, [VALID] * n
    for l_s, r_s, lx in ((VALID, NO_RIGHT_RECT, 1.0), (NO_LEFT_RECT, VALID, 1.0), (NO_DEPTH, FAILED_ARUN, 1.0),
                         (NO_DEPTH, FAILED_ARUN, 3.0)):           # the last one: negative disparity
        lxy.append(np.array([lx, 2.0]))
        rxy.append(np.array([1.0, 2.0]))
        ls.append(l_s)
        rs.append(r_s)
        expd.append(0.0)
    d, _, _ = O.get_depth_from_rectified_matches(cam, sp, np.array(lxy), ls, np.array(rxy), rs)
    checked = 0
    for got, exp in zip(d, expd):
        if exp < sp.min_point_dist or exp > sp.max_point_dist:
            assert abs(got) < 1e-3
        else:
            assert abs(got - exp) < 1e-3, (got, exp)
            checked += 1
    assert checked >= 9 * 9   # ratios 1 .. 100 of a 0.11 m baseline lie inside (0.1, 15) m


@pytest.mark.parametrize("px,rnd,expected,cropped", [
    ((700, 300), False, (700, 300), False),          # CropToSizeInside (tests/testUtilsOpenCV.cpp:280-288)
    ((799, 599), False, (799, 599), False),          # CropToSizeBoundary (:291-317)
    ((800, 600), False, (799, 599), True),
    ((0, 0), False, (0, 0), False),
    ((-1, -1), False, (0, 0), True),
    ((1000, 700), False, (799, 599), True),          # CropToSize (:320-349)
    ((-100, -200), False, (0, 0), True),
    ((700.3, 399.5), True, (700, 400), False),
    ((699.50001, 300.499), True, (700, 300), False),
    ((799.5, 599.5), True, (799, 599), True),        # RoundAndCropToSizeBoundary (:352-366)
    ((-0.499, -0.499), True, (0, 0), False),
    ((1000.4, 700.4), True, (799, 599), True),       # RoundAndCropToSizeOutside (:369-383)
    ((-1000, -800), True, (0, 0), True),
])
def test_crop_to_size(px, rnd, expected, cropped):
    """UtilsOpenCV::cropToSize / roundAndCropToSize in an 800 x 600 frame, every case of the reference's tests."""
    got, c = O.crop_to_size(px, 800, 600, rnd)
    assert got == (float(expected[0]), float(expected[1])) and c == cropped
