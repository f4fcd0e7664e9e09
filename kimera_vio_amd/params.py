"""Parameter parsing mirroring the reference's parseYAML methods.

* ``load_camera_params``  -> VIO::CameraParams::parseYAML  (src/frontend/CameraParams.cpp:24-60)
* ``load_frontend_params`` -> VIO::FrontendParams::parseYAML (src/frontend/VisionImuFrontendParams.cpp:80-117),
  which calls FeatureDetectorParams::parseYAML (feature-detector/FeatureDetectorParams.cpp:105-222),
  TrackerParams::parseYAML (VisionImuTrackerParams.cpp:87-135) and
  StereoMatchingParams::parseYAML (StereoMatchingParams.cpp:78-92).

The YAML files are OpenCV-FileStorage flavoured (``%YAML:1.0`` first line), which
pyyaml rejects, so that directive line is dropped before parsing.  As in the
reference (YamlParser.h:41-47) a missing key is an error.
"""
from __future__ import annotations

import yaml

from . import _abi as abi


def _read_yaml(path_or_text: str) -> dict:
    if "\n" in path_or_text:
        text = path_or_text
    else:
        with open(path_or_text, "r") as f:
            text = f.read()
    lines = [ln for ln in text.splitlines() if not ln.startswith("%YAML")]
    return yaml.safe_load("\n".join(lines))


def default_frontend_params() -> abi.FrontendParams:
    """Class defaults of the reference's param structs (header initialisers)."""
    p = abi.FrontendParams()
    d = p.detector
    d.feature_detector_type = abi.DET_GFTT
    d.max_features_per_frame = 400
    d.enable_subpixel_corner_refinement = 1
    d.subpix_window_size = 10
    d.subpix_zero_zone = -1
    d.subpix_max_iters = 10
    d.subpix_epsilon = 0.01
    d.enable_non_max_suppression = 1
    d.non_max_suppression_type = abi.ANMS_RANGETREE
    d.min_distance = 10
    d.max_nr_keypoints_before_anms = 2000
    d.nr_horizontal_bins = 5
    d.nr_vertical_bins = 5
    for i in range(25):
        d.binning_mask[i] = 1
    d.quality_level = 0.001
    d.block_size = 3
    d.use_harris_detector = 0
    d.k = 0.04
    d.sortidx_policy = abi.SORTIDX_LIBSTDCXX
    d.fast_thresh = 10
    t = p.tracker
    t.klt_win_size = 24
    t.klt_max_iter = 30
    t.klt_max_level = 3
    t.max_feature_track_age = 25
    t.klt_eps = 0.01
    t.optical_flow_predictor_type = abi.FLOW_NO_PREDICTION
    t.disparity_threshold = 0.5
    t.min_nr_mono_inliers = 10
    t.min_nr_stereo_inliers = 5
    t.ransac_threshold_mono = 1.0e-6
    t.ransac_threshold_stereo = 1.0
    t.ransac_max_iterations = 100
    t.ransac_randomize = 1          # class default (VisionImuTrackerParams.h:64); every YAML sets 0
    t.ransac_probability = 0.995
    t.ransac_use_1point_stereo = 1
    t.ransac_use_2point_mono = 1
    t.ransac_rng_policy = abi.RNG_LIBSTDCXX_PRE11
    t.pose_2d2d_algorithm = 1  # Pose2d2dAlgorithm::NISTER (VisionImuTrackerParams.h:68)
    p.use_pnp_tracking = 1     # VisionImuFrontendParams.h:59 (every shipped YAML but KinectAzure sets 0)
    p.pnp = abi.PnpParams(abi.PNP_EPNP, 10, 1.0, 0, 0)   # VisionImuTrackerParams.h:73-76
    s = p.stereo
    s.tolerance_template_matching = 0.15
    s.templ_cols = 101
    s.templ_rows = 11
    s.stripe_extra_rows = 0
    s.subpixel_refinement = 0
    s.equalize_image = 0
    s.min_point_dist = 0.1
    s.max_point_dist = 15.0
    p.min_intra_keyframe_time_ns = 0.2 * 10e6   # sic: VisionImuFrontendParams.h:48
    p.max_intra_keyframe_time_ns = 10.0 * 10e6  # sic: VisionImuFrontendParams.h:49
    p.min_number_features = 0
    p.max_disparity_since_lkf = 200.0
    p.use_stereo_tracking = 1
    p.use_ransac = 1  # VisionImuFrontendParams.h:60
    return p


def load_detector_params(path: str, into: abi.DetectorParams | None = None) -> abi.DetectorParams:
    y = _read_yaml(path)
    d = into if into is not None else default_frontend_params().detector
    d.feature_detector_type = int(y["feature_detector_type"])
    d.enable_subpixel_corner_refinement = int(y["enable_subpixel_corner_finder"])
    if d.enable_subpixel_corner_refinement:
        d.subpix_max_iters = int(y["max_iters"])
        d.subpix_epsilon = float(y["epsilon_error"])
        d.subpix_window_size = int(y["window_size"])
        d.subpix_zero_zone = int(y["zero_zone"])
    d.enable_non_max_suppression = int(y["enable_non_max_suppression"])
    d.non_max_suppression_type = int(y["non_max_suppression_type"])
    d.max_features_per_frame = int(y["maxFeaturesPerFrame"])
    d.max_nr_keypoints_before_anms = int(y["max_nr_keypoints_before_anms"])
    d.nr_horizontal_bins = int(y["nr_horizontal_bins"])
    d.nr_vertical_bins = int(y["nr_vertical_bins"])
    total = d.nr_horizontal_bins * d.nr_vertical_bins
    if total > abi.KVFE_MAX_BINS:
        raise ValueError("too many bins")
    mask = list(y["binning_mask"] or [])
    if mask and len(mask) != total:
        raise ValueError("Binning mask size specified by the user is inconsistent")
    for i in range(abi.KVFE_MAX_BINS):
        d.binning_mask[i] = 0
    for i in range(total):
        v = int(mask[i]) if mask else 1
        if v not in (0, 1):
            raise ValueError("Binning mask can only have binary entries {0;1}")
        d.binning_mask[i] = v
    d.quality_level = float(y["quality_level"])
    d.min_distance = int(y["min_distance"])
    d.block_size = int(y["block_size"])
    d.use_harris_detector = int(y["use_harris_detector"])
    d.k = float(y["k"])
    d.fast_thresh = int(y["fast_thresh"])
    return d


def load_frontend_params(path: str, use_ransac: int | None = None) -> abi.FrontendParams:
    """FrontendParams::parseYAML.  ``use_ransac=None`` (default) keeps the YAML's useRANSAC as the
    reference does; tests of the tracking / detection / stereo path alone pass 0 explicitly.
    Like YamlParser::getYamlParam (YamlParser.h:41-47) every key the reference reads is required."""
    y = _read_yaml(path)
    p = default_frontend_params()
    load_detector_params(path, p.detector)
    t = p.tracker
    t.klt_win_size = int(y["klt_win_size"])
    t.klt_max_iter = int(y["klt_max_iter"])
    t.klt_max_level = int(y["klt_max_level"])
    t.klt_eps = float(y["klt_eps"])
    t.max_feature_track_age = int(y["maxFeatureAge"])
    t.disparity_threshold = float(y["disparityThreshold"])
    t.optical_flow_predictor_type = int(y["optical_flow_predictor_type"])
    t.min_nr_mono_inliers = int(y["minNrMonoInliers"])
    t.min_nr_stereo_inliers = int(y["minNrStereoInliers"])
    t.ransac_threshold_mono = float(y["ransac_threshold_mono"])
    t.ransac_threshold_stereo = float(y["ransac_threshold_stereo"])
    t.ransac_max_iterations = int(y["ransac_max_iterations"])
    t.ransac_probability = float(y["ransac_probability"])
    t.ransac_randomize = int(y["ransac_randomize"])
    t.ransac_use_1point_stereo = int(y["ransac_use_1point_stereo"])
    t.ransac_use_2point_mono = int(y["ransac_use_2point_mono"])
    if "2d2d_algorithm" in y:
        t.pose_2d2d_algorithm = int(y["2d2d_algorithm"])
    s = p.stereo
    s.tolerance_template_matching = float(y["toleranceTemplateMatching"])
    s.templ_cols = int(y["templ_cols"])
    s.templ_rows = int(y["templ_rows"])
    s.stripe_extra_rows = int(y["stripe_extra_rows"])
    s.min_point_dist = float(y["minPointDist"])
    s.max_point_dist = float(y["maxPointDist"])
    s.subpixel_refinement = int(y["subpixelRefinementStereo"])
    s.equalize_image = int(y["equalizeImage"])
    p.min_intra_keyframe_time_ns = float(y["min_intra_keyframe_time"]) * 1e9
    p.max_intra_keyframe_time_ns = float(y["max_intra_keyframe_time"]) * 1e9
    p.min_number_features = int(y["minNumberFeatures"])
    p.use_stereo_tracking = int(y["useStereoTracking"])
    p.max_disparity_since_lkf = float(y["max_disparity_since_lkf"])
    p.use_ransac = int(y["useRANSAC"]) if use_ransac is None else int(use_ransac)
    # use_2d2d_tracking / use_3d3d_tracking are parsed by the reference (VisionImuFrontendParams.cpp:104-105)
    # and read nowhere in src/.  use_pnp_tracking gates Tracker::pnp on keyframes (StereoVisionImuFrontend.cpp:389):
    # the step runs it on the device against the landmark map of kvfe_frontend_update_map; EPNP and KneipP3P are
    # implemented, another pnp_algorithm is refused at kvfe_create (KVFE_ERR_UNSUPPORTED).
    int(y["use_2d2d_tracking"]), int(y["use_3d3d_tracking"])
    p.use_pnp_tracking = int(y["use_pnp_tracking"])
    # (the reference's YamlParser aborts on a missing key; its own detector-test YAMLs, which it never parses as tracker
    # parameters, lack these four -- they fall back to the class defaults of VisionImuTrackerParams.h:55-76)
    p.pnp = abi.PnpParams(int(y.get("pnp_algorithm", abi.PNP_EPNP)), int(y.get("min_pnp_inliers", 10)),
                          float(y.get("ransac_threshold_pnp", 1.0)),
                          int(y.get("optimize_2d3d_pose_from_inliers", 0)), 0)
    return p


def load_camera_params(path: str) -> abi.CameraParams:
    y = _read_yaml(path)
    c = abi.CameraParams()
    res = y["resolution"]
    c.width, c.height = int(res[0]), int(res[1])
    intr = y["intrinsics"]
    if len(intr) != 4:
        raise ValueError("intrinsics must be [fu, fv, cu, cv]")
    for i in range(4):
        c.intrinsics[i] = float(intr[i])
    model = str(y["distortion_model"]).lower()
    if model in ("none",):
        c.distortion_model = abi.DIST_NONE
    elif model in ("plumb_bob", "radial-tangential", "radtan"):
        c.distortion_model = abi.DIST_RADTAN
    elif model in ("equidistant",):
        c.distortion_model = abi.DIST_EQUIDISTANT
    else:
        raise ValueError(f"Unrecognized distortion model: {model}")
    coeffs = [float(v) for v in y["distortion_coefficients"]]
    if len(coeffs) < 4 or len(coeffs) > abi.KVFE_MAX_DIST_COEFFS:
        raise ValueError("need 4..8 distortion coefficients")
    c.n_distortion = len(coeffs)
    for i, v in enumerate(coeffs):
        c.distortion[i] = v
    pose = [float(v) for v in y["T_BS"]["data"]]
    if len(pose) != 16:
        raise ValueError("T_BS must have 16 entries")
    for i in range(16):
        c.body_pose_cam[i] = pose[i]
    return c
