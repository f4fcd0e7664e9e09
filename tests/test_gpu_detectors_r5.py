"""Round 5, SURVEY §8 row a15: the detector options besides the minimum eigenvalue on the GPU against the oracle --
use_harris_corner_detector_ (cv::cornerHarris response inside goodFeaturesToTrack, FeatureDetector.cpp:73-80).
Tolerance 0: the response is float arithmetic in OpenCV's order on both sides."""
import os

import numpy as np
import pytest

import oracle_lib as O
from kimera_vio_amd import frontend as F
from kimera_vio_amd import params as P
from test_gpu_parity import G, _kf_rotations, _run_sequence, euroc_cams, euroc_params, gray

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def seq():
    z = np.load(os.path.join(G, "micro_euroc_f10_18.npz"))
    return dict(lefts=z["lefts"], rights=z["rights"], ts=z["timestamps"], body_R=z["body_R"])


@pytest.mark.parametrize("k", [0.04, 0.06])
def test_raw_gftt_harris_bit_exact(k):
    """cv::GFTTDetector::detect with useHarrisDetector = true on the reference's detector-test image, with and without
    a user mask; a blank image gives nothing."""
    img = gray("left_fisheye_img_0.png")
    L, R = euroc_cams()
    d = P.load_detector_params(os.path.join(G, "ForFeatureDetector", "frontendParams-noNMS.yaml"))
    d.use_harris_detector, d.k = 1, k
    p = euroc_params()
    p.detector = d
    c = F.Context(L, R, p)
    try:
        got = c.raw_feature_detection(img)
        exp, _ = O.good_features_to_track(img, d.max_nr_keypoints_before_anms, d.quality_level, d.min_distance, 3,
                                          harris_k=k)
        mineig, _ = O.good_features_to_track(img, d.max_nr_keypoints_before_anms, d.quality_level, d.min_distance, 3)
        assert len(exp) > 100 and not (len(exp) == len(mineig) and np.array_equal(exp, mineig))
        assert np.array_equal(got, exp)
        rng = np.random.RandomState(5)
        mask = (rng.uniform(size=img.shape) > 0.3).astype(np.uint8) * 255
        mask[50:150, 200:400] = 0
        got = c.raw_feature_detection(img, mask)
        exp, _ = O.good_features_to_track(img, d.max_nr_keypoints_before_anms, d.quality_level, d.min_distance, 3,
                                          mask=mask, harris_k=k)
        assert np.array_equal(got, exp)
        assert len(c.raw_feature_detection(np.full(img.shape, 200, np.uint8))) == 0
    finally:
        c.close()


def test_feature_detection_harris_with_tracked_mask_and_binning(seq):
    """FeatureDetector::featureDetection with the Harris response: the shipped EuRoC detector parameters (binning ANMS,
    cornerSubPix) + use_harris_corner_detector_, discs around tracked keypoints."""
    img = seq["lefts"][0]
    L, R = euroc_cams()
    p = euroc_params(use_harris_detector=1, k=0.04)
    c = F.Context(L, R, p)
    try:
        d = p.detector
        none = np.zeros((0, 2), np.float32)
        first = c.feature_detection(img, none, 300)
        exp, _ = O.feature_detection(img, none, 300, d)
        assert len(exp) > 50 and np.array_equal(first, exp)
        tracked = exp[::2] + np.float32(0.37)
        need = 300 - len(tracked)
        got = c.feature_detection(img, tracked, need)
        exp2, _ = O.feature_detection(img, tracked, need, d)
        assert len(exp2) > 0 and np.array_equal(got, exp2)
    finally:
        c.close()


def test_frontend_sequence_harris(seq, ocam=None):
    """the whole front-end on the EuRoC frames with use_harris_corner_detector_: 1 -- every field of every frame"""
    L, R = euroc_cams()
    ocam = O.Camera(L, R)
    seq = dict(seq)
    seq["camR"] = _kf_rotations(seq["body_R"], ocam)
    p = euroc_params(use_harris_detector=1, k=0.04)
    fe = [O.Frontend(L, R, p)]
    c = F.Context(L, R, p, batch=1)
    try:
        kinds = _run_sequence(fe, c, seq)
    finally:
        c.close()
    assert [k[2] for k in kinds] == [1, 0, 0, 0, 1, 0, 0, 0, 1]
    assert all(k[3] > 50 for k in kinds[1:])


def test_harris_odd_image_size_takes_the_scalar_tail():
    """(W * H) % 4 != 0: the last pixels of the image go through calcHarris' double-k tail; the response of the last
    row matters through the local-maximum test of the row above."""
    full = gray("left_fisheye_img_0.png")
    img = np.ascontiguousarray(full[100:100 + 241, 200:200 + 323])
    L, R = euroc_cams()
    for cam in (L, R):
        cam.width, cam.height = 323, 241
    d = P.load_detector_params(os.path.join(G, "ForFeatureDetector", "frontendParams-noNMS.yaml"))
    d.use_harris_detector, d.k, d.quality_level = 1, 0.04, 1e-6
    p = euroc_params()
    p.detector = d
    c = F.Context(L, R, p)
    try:
        got = c.raw_feature_detection(img)
        exp, _ = O.good_features_to_track(img, d.max_nr_keypoints_before_anms, d.quality_level, d.min_distance, 3,
                                          harris_k=0.04)
        assert len(exp) > 50 and np.array_equal(got, exp)
    finally:
        c.close()


# --------------------------------------------------------------------------------------------- FAST
def _fast_params(thr, anms=None, max_features=300, subpix=1):
    from kimera_vio_amd import _abi as abi
    p = euroc_params(feature_detector_type=abi.DET_FAST, fast_thresh=thr, max_features_per_frame=max_features,
                     enable_subpixel_corner_refinement=subpix)
    if anms is None:
        p.detector.enable_non_max_suppression = 0
    else:
        p.detector.enable_non_max_suppression = 1
        p.detector.non_max_suppression_type = anms
    return p


@pytest.mark.parametrize("thr", [20, 35, 120])
def test_raw_fast_bit_exact(thr):
    """cv::FastFeatureDetector::create(fast_thresh, true)->detect(img, keypoints, mask): every keypoint, in cv::FAST's
    raster order, with and without a user mask; blank image."""
    L, R = euroc_cams()
    c = F.Context(L, R, _fast_params(thr))
    try:
        for name in ("left_img_0.png", "left_fisheye_img_0.png"):
            img = gray(name)
            exp = O.fast_detect(img, thr)
            got = c.raw_feature_detection(img)
            assert len(exp) > 20 and len(exp) < 8192
            assert np.array_equal(got, exp[:, :2])
            rng = np.random.RandomState(11)
            mask = (rng.uniform(size=img.shape) > 0.4).astype(np.uint8) * 255
            mask[200:300, 100:300] = 0
            got = c.raw_feature_detection(img, mask)
            assert np.array_equal(got, O.fast_detect(img, thr, mask=mask)[:, :2])
        assert len(c.raw_feature_detection(np.full((480, 752), 90, np.uint8))) == 0
    finally:
        c.close()


@pytest.mark.parametrize("anms", [None, 0, 1, 2, 3, 4, 5, 6])
def test_feature_detection_fast_every_anms_type(anms, seq):
    """FeatureDetector::featureDetection with FeatureDetectorType::FAST: cv::sortIdx on the keypoints' integer responses
    (libstdc++'s introsort tie order), every ANMS type, discs around tracked keypoints, cornerSubPix."""
    img = seq["lefts"][0]
    L, R = euroc_cams()
    p = _fast_params(35, anms)
    c = F.Context(L, R, p)
    try:
        d = p.detector
        none = np.zeros((0, 2), np.float32)
        need = 300
        exp, _ = O.feature_detection(img, none, need, d)
        got = c.feature_detection(img, none, need)
        assert len(exp) > 50
        assert np.array_equal(got, exp)
        tracked = exp[::3] + np.float32(0.41)
        exp2, _ = O.feature_detection(img, tracked, need - len(tracked), d)
        got2 = c.feature_detection(img, tracked, need - len(tracked))
        assert len(exp2) > 0 and np.array_equal(got2, exp2)
    finally:
        c.close()


def test_feature_detection_fast_stable_sortidx_policy(seq):
    from kimera_vio_amd import _abi as abi
    img = seq["lefts"][1]
    L, R = euroc_cams()
    p = _fast_params(30, 6, subpix=0)
    p.detector.sortidx_policy = abi.SORTIDX_STABLE
    c = F.Context(L, R, p)
    try:
        none = np.zeros((0, 2), np.float32)
        exp, _ = O.feature_detection(img, none, 200, p.detector)
        assert len(exp) > 50 and np.array_equal(c.feature_detection(img, none, 200), exp)
    finally:
        c.close()


def test_frontend_sequence_fast(seq):
    """the whole front-end on the EuRoC frames with feature_detector_type: 0 (FAST), fast_thresh 35"""
    L, R = euroc_cams()
    ocam = O.Camera(L, R)
    seq = dict(seq)
    seq["camR"] = _kf_rotations(seq["body_R"], ocam)
    p = _fast_params(35, 6)
    fe = [O.Frontend(L, R, p)]
    c = F.Context(L, R, p, batch=1)
    try:
        kinds = _run_sequence(fe, c, seq)
    finally:
        c.close()
    assert [k[2] for k in kinds] == [1, 0, 0, 0, 1, 0, 0, 0, 1]
    assert all(k[3] > 50 for k in kinds[1:])


def test_fast_batched_streams_and_unsupported_types(seq):
    from kimera_vio_amd import _abi as abi
    L, R = euroc_cams()
    ocam = O.Camera(L, R)
    seq = dict(seq)
    seq["camR"] = _kf_rotations(seq["body_R"], ocam)
    p = _fast_params(40, 6)
    B = 3
    fe = [O.Frontend(L, R, p) for _ in range(B)]
    c = F.Context(L, R, p, batch=B)
    try:
        n = len(seq["ts"])
        _run_sequence(fe, c, seq, stream_of=lambda s, i: (i + 2 * s) % n if s != 1 else (n - 1 - i), n=6)
    finally:
        c.close()
    for t in (abi.DET_ORB, abi.DET_AGAST):
        q = euroc_params(feature_detector_type=t)
        with pytest.raises(F.KvfeError) as e:
            F.Context(L, R, q)
        assert e.value.status == abi.KVFE_ERR_UNSUPPORTED
