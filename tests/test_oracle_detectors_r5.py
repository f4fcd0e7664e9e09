"""Round 5, SURVEY §8 row a15: the detector options besides the minimum eigenvalue -- use_harris_corner_detector_
(cv::cornerHarris inside cv::goodFeaturesToTrack, FeatureDetector.cpp:73-80) and FeatureDetectorType::FAST
(cv::FastFeatureDetector::create(fast_thresh, true), FeatureDetector.cpp:35-40).  The reference holds no vector for either
(its detector tests all run the GFTT minimum-eigenvalue configuration), so the C oracle is checked here against
independent numpy statements of the published algorithms: parity for these two options is pinned on those, not on
reference-held numbers -- DESIGN.md section 2 says so."""
import os

import numpy as np
import pytest
from PIL import Image

import oracle_lib as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gray(name):
    return np.array(Image.open(os.path.join(G, name)).convert("L"))


def r101(i, n):
    i = np.asarray(i)
    i = np.where(i < 0, -i, i)
    return np.where(i >= n, 2 * n - 2 - i, i)


def harris_numpy(img, k, block=3):
    """cv::cornerHarris(u8, blockSize 3, ksize 3, k, BORDER_REFLECT_101) in float32 numpy: Sobel (scale folded into the
    smoothing kernel as OpenCV does), covariance products, unnormalised 3x3 box filter summed in float64, then calcHarris'
    4-wide float lanes over the image as one row with the scalar double-k tail for the last (w*h) % 4 pixels."""
    f32 = np.float32
    h, w = img.shape
    S = img.astype(f32)
    scale = 1.0 / (4.0 * block * 255.0)
    f1, f0 = f32(1.0 * f32(scale)), f32(2.0 * f32(scale))
    xs = np.arange(w)
    xm, xp = r101(xs - 1, w), r101(xs + 1, w)
    Rdx = S[:, xp] - S[:, xm]
    Tdy = (f1 * S[:, xm] + f0 * S) + f1 * S[:, xp]
    ys = np.arange(h)
    ym, yp = r101(ys - 1, h), r101(ys + 1, h)
    dx = (Rdx[ym] + Rdx[yp]) * f1 + Rdx * f0
    dy = Tdy[yp] - Tdy[ym]
    out = []
    for c in (dx * dx, dx * dy, dy * dy):
        c = c.astype(f32).astype(np.float64)
        rs = (c[:, xm] + c) + c[:, xp]
        out.append(((rs[ym] + rs) + rs[yp]).astype(f32))
    a, b, c = out
    kf = f32(k)
    ac = a + c
    res = (a * c - b * b) - (kf * ac) * ac
    n = w * h
    t = n % 4
    if t:
        fa, fb, fc = a.reshape(-1)[n - t:], b.reshape(-1)[n - t:], c.reshape(-1)[n - t:]
        tail = ((fa * fc - fb * fb).astype(np.float64) - k * (fa + fc).astype(np.float64) * (fa + fc).astype(np.float64))
        res.reshape(-1)[n - t:] = tail.astype(f32)
    return res.astype(f32)


@pytest.mark.parametrize("name,crop", [("chessboard.png", None), ("left_fisheye_img_0.png", (40, 200, 61, 131)),
                                       ("left_img_0.png", (100, 300, 33, 47))])
@pytest.mark.parametrize("k", [0.04, 0.06])
def test_corner_harris_matches_numpy_statement(name, crop, k):
    img = gray(name)
    if crop:
        y, x, hh, ww = crop          # odd sizes: (w*h) % 4 != 0 exercises calcHarris' scalar tail
        img = np.ascontiguousarray(img[y:y + hh, x:x + ww])
    got = O.corner_harris(img, k)
    exp = harris_numpy(img, k)
    assert got.shape == exp.shape
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    assert (got > 0).any() and (got < 0).any()   # corners and edges


def test_good_features_to_track_harris_properties():
    """goodFeaturesToTrack(useHarrisDetector = true) on the reference's chessboard image: corners are 3x3 local maxima of
    the Harris response above quality * max, at least min_distance apart, in descending response order -- and they are not
    the minimum-eigenvalue corners' order (the option does something)."""
    img = gray("chessboard.png")
    xy, q = O.good_features_to_track(img, 100, 0.01, 10, harris_k=0.04)
    assert 30 <= len(xy) <= 100
    R = O.corner_harris(img, 0.04)
    thr = np.float32(R.max() * 0.01)
    for (x, y), qv in zip(xy.astype(int), q):
        assert 1 <= x < img.shape[1] - 1 and 1 <= y < img.shape[0] - 1
        assert R[y, x] == qv > thr
        assert R[y, x] == R[y - 1:y + 2, x - 1:x + 2].max()
    assert np.all(np.diff(q) <= 0)
    d = np.linalg.norm(xy[:, None, :] - xy[None, :, :], axis=2) + np.eye(len(xy)) * 1e9
    assert d.min() >= 10
    real = gray("left_img_0.png")
    a, _ = O.good_features_to_track(real, 200, 0.01, 10, harris_k=0.04)
    b, _ = O.good_features_to_track(real, 200, 0.01, 10)
    assert len(a) > 50 and (len(a) != len(b) or not np.array_equal(a, b))
    blank = np.full((40, 50), 128, np.uint8)
    assert len(O.good_features_to_track(blank, 100, 0.01, 10, harris_k=0.04)[0]) == 0


# --------------------------------------------------------------------------------------------- FAST
FAST_RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3),
             (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]   # (dx, dy)


def fast_numpy(img, t, nonmax=True):
    """FAST-9/16 from its definition (Rosten & Drummond; cv::FAST TYPE_9_16): pixel p is a corner when 9 contiguous pixels
    of the radius-3 circle are all darker than p - t or all brighter than p + t; score = the largest threshold for which
    it still is one, minus 1; non-maximum suppression keeps strict 3x3 maxima of the score.  Returns (x, y, score) rows in
    raster order, rows / columns 3 .. size - 4."""
    h, w = img.shape
    I = img.astype(np.int16)
    c = I[3:h - 3, 3:w - 3]
    D = np.stack([c - I[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in FAST_RING])   # v - ring pixel
    D2 = np.concatenate([D, D[:8]])
    arc_min = np.stack([D2[a:a + 9].min(axis=0) for a in range(16)])       # darker arcs: all of v - x > t
    arc_max = np.stack([D2[a:a + 9].max(axis=0) for a in range(16)])       # brighter arcs: all of v - x < -t
    is_corner = (arc_min > t).any(axis=0) | (arc_max < -t).any(axis=0)
    score = np.maximum(arc_min.max(axis=0), (-arc_max).max(axis=0)) - 1
    S = np.zeros((h, w), np.int32)
    S[3:h - 3, 3:w - 3] = np.where(is_corner, score, 0)
    C = np.zeros((h, w), bool)
    C[3:h - 3, 3:w - 3] = is_corner
    if not nonmax:
        ys, xs = np.nonzero(C)
        return np.stack([xs, ys, np.zeros_like(xs)], axis=1).astype(np.float32)
    P = np.pad(S, 1)
    nb = np.stack([P[1 + dy:1 + dy + h, 1 + dx:1 + dx + w] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dx, dy) != (0, 0)])
    keep = C & (S > nb.max(axis=0))
    ys, xs = np.nonzero(keep)
    return np.stack([xs, ys, S[ys, xs]], axis=1).astype(np.float32)


@pytest.mark.parametrize("name", ["left_img_0.png", "left_fisheye_img_0.png", "chessboard.png"])
@pytest.mark.parametrize("t", [0, 10, 35, 120])
def test_fast_matches_numpy_statement(name, t):
    img = gray(name)
    for nonmax in (True, False):
        got = O.fast_detect(img, t, nonmax=nonmax)
        exp = fast_numpy(img, t, nonmax=nonmax)
        assert got.shape == exp.shape, (name, t, nonmax, got.shape, exp.shape)
        assert np.array_equal(got, exp)
    if t == 10 and name != "chessboard.png":
        assert len(got) > 1000


def test_fast_mask_small_and_blank_images():
    img = gray("left_img_0.png")
    rng = np.random.RandomState(9)
    mask = (rng.uniform(size=img.shape) > 0.5).astype(np.uint8) * 255
    allk = O.fast_detect(img, 20)
    got = O.fast_detect(img, 20, mask=mask)
    exp = allk[mask[allk[:, 1].astype(int), allk[:, 0].astype(int)] != 0]
    assert 0 < len(got) < len(allk) and np.array_equal(got, exp)
    assert len(O.fast_detect(np.full((50, 60), 77, np.uint8), 10)) == 0
    tiny = np.ascontiguousarray(img[100:107, 200:207])       # 7 x 7: exactly one candidate pixel
    assert np.array_equal(O.fast_detect(tiny, 5), fast_numpy(tiny, 5))
    assert len(O.fast_detect(np.ascontiguousarray(img[:6, :40]), 5)) == 0   # fewer than 7 rows: nothing


@pytest.mark.parametrize("anms", [0, 4, 6])
def test_feature_detection_fast_goes_through_anms_with_real_responses(anms):
    """FeatureDetectorType::FAST through FeatureDetector::featureDetection: the keypoints reach the ANMS stage with their
    integer responses (cv::sortIdx on real keys), TopN takes the first keypoints in raster order (it receives the unsorted
    list, NonMaximumSuppression.cpp:67), binning keeps at most the quota per bin and prefers high scores."""
    from kimera_vio_amd import _abi as abi
    from kimera_vio_amd import params as P
    img = gray("left_img_0.png")
    d = P.load_detector_params(os.path.join(G, "ForFeatureDetector", "frontendParams-NMS-Binning.yaml"))
    d.feature_detector_type, d.fast_thresh = abi.DET_FAST, 20
    d.non_max_suppression_type, d.max_features_per_frame = anms, 100
    d.enable_subpixel_corner_refinement = 0
    none = np.zeros((0, 2), np.float32)
    got, _ = O.feature_detection(img, none, 100, d)
    raw = O.fast_detect(img, 20)
    assert len(raw) > 1000
    pts = {(x, y): r for x, y, r in raw}
    assert all((x, y) in pts for x, y in got)
    if anms == 0:
        assert np.array_equal(got, raw[:100, :2])
    else:
        assert 50 <= len(got) <= 130
        # stronger than the typical keypoint: the sorted list is walked from the highest response down
        assert np.median([pts[(x, y)] for x, y in got]) >= np.median(raw[:, 2])
    tracked = got[:10] + np.float32(0.3)
    got2, _ = O.feature_detection(img, tracked, 50, d)
    for x, y in got2:          # nothing inside a tracked keypoint's disc
        assert np.min(np.hypot(tracked[:, 0] - x, tracked[:, 1] - y)) > d.min_distance - 1.5
