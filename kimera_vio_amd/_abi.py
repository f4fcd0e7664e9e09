"""ctypes mirror of include/kvfe.h (POD structs + enums).  Keep in lock-step with the header."""
import ctypes as C

KVFE_MAX_BINS = 256
KVFE_MAX_DIST_COEFFS = 8
SSD_TIE_EXACT, SSD_TIE_F32 = 0, 1
KVFE_N_STAGES = 16

# status
KVFE_OK = 0
KVFE_ERR_INVALID_ARG = -1
KVFE_ERR_UNSUPPORTED = -2
KVFE_ERR_NO_DEVICE = -3
KVFE_ERR_HIP = -4
KVFE_ERR_CAPACITY = -5
KVFE_ERR_NOT_READY = -6

# VIO::KeypointStatus
KP_VALID, KP_NO_LEFT_RECT, KP_NO_RIGHT_RECT, KP_NO_DEPTH, KP_FAILED_ARUN = range(5)
# VIO::AnmsAlgorithmType
ANMS_TOPN, ANMS_BROWN, ANMS_SDC, ANMS_KDTREE, ANMS_RANGETREE, ANMS_SSC, ANMS_BINNING = range(7)
DET_FAST, DET_ORB, DET_AGAST, DET_GFTT = range(4)
FLOW_NO_PREDICTION, FLOW_ROTATIONAL = range(2)
DIST_NONE, DIST_RADTAN, DIST_EQUIDISTANT = range(3)
SORTIDX_LIBSTDCXX, SORTIDX_STABLE = range(2)
TRACKING_VALID, TRACKING_LOW_DISPARITY, TRACKING_FEW_MATCHES, TRACKING_INVALID, TRACKING_DISABLED = range(5)
RNG_LIBSTDCXX_PRE11, RNG_LIBSTDCXX_11 = range(2)
FRONTEND_STEREO, FRONTEND_MONO, FRONTEND_RGBD = range(3)
DEPTH_U16, DEPTH_F32 = range(2)


class CameraParams(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("intrinsics", C.c_double * 4),
        ("distortion_model", C.c_int32), ("n_distortion", C.c_int32),
        ("distortion", C.c_double * KVFE_MAX_DIST_COEFFS),
        ("body_pose_cam", C.c_double * 16),
    ]


class DetectorParams(C.Structure):
    _fields_ = [
        ("feature_detector_type", C.c_int32),
        ("max_features_per_frame", C.c_int32),
        ("enable_subpixel_corner_refinement", C.c_int32),
        ("subpix_window_size", C.c_int32),
        ("subpix_zero_zone", C.c_int32),
        ("subpix_max_iters", C.c_int32),
        ("subpix_epsilon", C.c_double),
        ("enable_non_max_suppression", C.c_int32),
        ("non_max_suppression_type", C.c_int32),
        ("min_distance", C.c_int32),
        ("max_nr_keypoints_before_anms", C.c_int32),
        ("nr_horizontal_bins", C.c_int32), ("nr_vertical_bins", C.c_int32),
        ("binning_mask", C.c_uint8 * KVFE_MAX_BINS),
        ("quality_level", C.c_double),
        ("block_size", C.c_int32),
        ("use_harris_detector", C.c_int32),
        ("k", C.c_double),
        ("sortidx_policy", C.c_int32),
        ("fast_thresh", C.c_int32),
    ]


class TrackerParams(C.Structure):
    _fields_ = [
        ("klt_win_size", C.c_int32), ("klt_max_iter", C.c_int32),
        ("klt_max_level", C.c_int32), ("max_feature_track_age", C.c_int32),
        ("klt_eps", C.c_double),
        ("optical_flow_predictor_type", C.c_int32), ("reserved0", C.c_int32),
        ("disparity_threshold", C.c_double),
        ("min_nr_mono_inliers", C.c_int32), ("min_nr_stereo_inliers", C.c_int32),
        ("ransac_threshold_mono", C.c_double), ("ransac_threshold_stereo", C.c_double),
        ("ransac_max_iterations", C.c_int32), ("ransac_randomize", C.c_int32),
        ("ransac_probability", C.c_double),
        ("ransac_use_1point_stereo", C.c_int32), ("ransac_use_2point_mono", C.c_int32),
        ("ransac_rng_policy", C.c_int32), ("pose_2d2d_algorithm", C.c_int32),
    ]


class StereoParams(C.Structure):
    _fields_ = [
        ("tolerance_template_matching", C.c_double),
        ("templ_cols", C.c_int32), ("templ_rows", C.c_int32),
        ("stripe_extra_rows", C.c_int32), ("subpixel_refinement", C.c_int32),
        ("min_point_dist", C.c_double), ("max_point_dist", C.c_double),
        ("equalize_image", C.c_int32), ("ssd_tie_policy", C.c_int32),
    ]


class DenseStereoParams(C.Structure):
    """kvfe_dense_stereo_params (DenseStereoParams, StereoMatchingParams.h:39-58)"""
    _fields_ = [
        ("use_sgbm", C.c_int32), ("post_filter_disparity", C.c_int32),
        ("median_blur_disparity", C.c_int32), ("pre_filter_cap", C.c_int32),
        ("sad_window_size", C.c_int32), ("min_disparity", C.c_int32),
        ("num_disparities", C.c_int32), ("uniqueness_ratio", C.c_int32),
        ("speckle_range", C.c_int32), ("speckle_window_size", C.c_int32),
        ("texture_threshold", C.c_int32), ("pre_filter_type", C.c_int32),
        ("pre_filter_size", C.c_int32), ("p1", C.c_int32), ("p2", C.c_int32),
        ("disp_12_max_diff", C.c_int32), ("use_mode_hh", C.c_int32), ("reserved0", C.c_int32),
    ]


def dense_stereo_params_default() -> "DenseStereoParams":
    """the values the reference runs with (it never parses them from YAML)"""
    return DenseStereoParams(use_sgbm=1, post_filter_disparity=0, median_blur_disparity=0,
                             pre_filter_cap=31, sad_window_size=11, min_disparity=1,
                             num_disparities=64, uniqueness_ratio=0, speckle_range=3,
                             speckle_window_size=500, texture_threshold=0, pre_filter_type=1,
                             pre_filter_size=9, p1=120, p2=240, disp_12_max_diff=-1, use_mode_hh=1,
                             reserved0=0)


class PnpParams(C.Structure):
    """kvfe_pnp_params (Tracker::pnp, VisionImuTrackerParams.h: pnp_algorithm_, min_pnp_inliers_, ransac_threshold_pnp_)"""
    _fields_ = [
        ("pnp_algorithm", C.c_int32), ("min_pnp_inliers", C.c_int32), ("ransac_threshold_pnp", C.c_double),
        ("optimize_2d3d_pose_from_inliers", C.c_int32), ("reserved0", C.c_int32),
    ]


PNP_KNEIP_P2P, PNP_KNEIP_P3P, PNP_GAO_P3P, PNP_EPNP, PNP_UPNP, PNP_UP3P, PNP_NONLINEAR, PNP_MLPNP = range(8)


def pnp_params_default():
    """class defaults of VisionImuTrackerParams.h:55-76 (= kvfe_default_frontend_params): EPNP, 10 inliers, 1 px
    (the shipped YAMLs set min_pnp_inliers: 20)"""
    return PnpParams(PNP_EPNP, 10, 1.0, 0, 0)


class FrontendParams(C.Structure):
    _fields_ = [
        ("detector", DetectorParams), ("tracker", TrackerParams), ("stereo", StereoParams),
        ("min_intra_keyframe_time_ns", C.c_double),
        ("max_intra_keyframe_time_ns", C.c_double),
        ("min_number_features", C.c_int64),
        ("max_disparity_since_lkf", C.c_double),
        ("use_stereo_tracking", C.c_int32), ("use_ransac", C.c_int32),
        ("use_pnp_tracking", C.c_int32), ("reserved1", C.c_int32), ("pnp", PnpParams),
    ]


class DepthParams(C.Structure):
    """kvfe_depth_params (CameraParams::DepthParams, CameraParams.h:131-155)"""
    _fields_ = [
        ("virtual_baseline", C.c_float), ("depth_to_meters", C.c_float), ("min_depth", C.c_float),
        ("max_depth", C.c_float), ("is_registered", C.c_int32), ("depth_type", C.c_int32),
    ]


def depth_params_default(depth_type: int = 0) -> "DepthParams":
    return DepthParams(virtual_baseline=1.0e-2, depth_to_meters=1.0, min_depth=0.0, max_depth=10.0,
                       is_registered=1, depth_type=depth_type)


class Frame(C.Structure):
    """kvfe_frame (VIO::Frame arrays, Frame.h:160-186)"""
    _fields_ = [("capacity", C.c_int32), ("n_keypoints", C.c_int32), ("keypoints", C.c_void_p),
                ("landmarks", C.c_void_p), ("landmarks_age", C.c_void_p), ("versors", C.c_void_p)]


class Config(C.Structure):
    _fields_ = [
        ("left", CameraParams), ("right", CameraParams), ("params", FrontendParams),
        ("batch", C.c_int32), ("device", C.c_int32),
        ("hip_stream", C.c_void_p),
        ("candidate_capacity", C.c_int32), ("frontend_type", C.c_int32), ("landmark_map_capacity", C.c_int32),
        ("depth", DepthParams),
        ("stream_groups", C.c_int32),
        ("device_frames_persist", C.c_int32), ("single_hip_stream", C.c_int32), ("copy_inputs", C.c_int32),
        ("ssd_impl", C.c_int32), ("lk_impl", C.c_int32),
    ]


class Rectification(C.Structure):
    _fields_ = [
        ("R1", C.c_double * 9), ("R2", C.c_double * 9),
        ("P1", C.c_double * 12), ("P2", C.c_double * 12), ("Q", C.c_double * 16),
        ("roi1", C.c_int32 * 4), ("roi2", C.c_int32 * 4),
        ("baseline", C.c_double),
    ]


class StereoOutput(C.Structure):
    _fields_ = [
        ("left_rect_xy", C.c_void_p), ("left_status", C.c_void_p),
        ("right_rect_xy", C.c_void_p), ("right_status", C.c_void_p),
        ("depth", C.c_void_p), ("right_xy", C.c_void_p), ("keypoints_3d", C.c_void_p),
        ("left_rect_img", C.c_void_p), ("right_rect_img", C.c_void_p),
    ]


class FrameInput(C.Structure):
    _fields_ = [
        ("timestamp_ns", C.c_int64),
        ("keyframe_R_cur_frame", C.c_double * 9),
        ("force_keyframe", C.c_int32), ("reserved0", C.c_int32),
    ]


class FrameOutput(C.Structure):
    _fields_ = [
        ("capacity", C.c_int32), ("n_keypoints", C.c_int32), ("is_keyframe", C.c_int32),
        ("n_tracked", C.c_int32), ("n_detected", C.c_int32), ("n_measurements", C.c_int32),
        ("frame_id", C.c_int64),
        ("landmarks", C.c_void_p), ("landmarks_age", C.c_void_p), ("keypoints", C.c_void_p),
        ("versors", C.c_void_p), ("left_rect_xy", C.c_void_p), ("left_status", C.c_void_p),
        ("right_rect_xy", C.c_void_p), ("right_status", C.c_void_p), ("depth", C.c_void_p),
        ("right_xy", C.c_void_p), ("keypoints_3d", C.c_void_p),
        ("meas_landmark", C.c_void_p), ("meas_uL_uR_v", C.c_void_p),
        ("tracking_status_mono", C.c_int32), ("tracking_status_stereo", C.c_int32),
        ("lkf_T_k_mono", C.c_double * 12), ("lkf_T_k_stereo", C.c_double * 12),
        ("info_mat_stereo_translation", C.c_double * 9),
        ("nr_mono_putatives", C.c_int32), ("nr_mono_inliers", C.c_int32),
        ("mono_ransac_iters", C.c_int32), ("nr_stereo_putatives", C.c_int32),
        ("nr_stereo_inliers", C.c_int32), ("reserved0", C.c_int32),
        ("tracking_status_pnp", C.c_int32), ("nr_pnp_inliers", C.c_int32), ("W_T_k_pnp", C.c_double * 12),
    ]


class RansacOutput(C.Structure):
    _fields_ = [
        ("status", C.c_int32), ("n_inliers", C.c_int32), ("iterations", C.c_int32),
        ("reserved0", C.c_int32), ("pose", C.c_double * 12), ("info", C.c_double * 9),
    ]


class StageTimes(C.Structure):
    _fields_ = [
        ("n_stages", C.c_int32), ("n_samples", C.c_int32),
        ("n_groups", C.c_int32), ("struct_size", C.c_int32),
        ("name", C.c_char_p * KVFE_N_STAGES),
        ("ms_total", C.c_double * KVFE_N_STAGES),
        ("alg_bytes", C.c_double * KVFE_N_STAGES),
        ("ms_active", C.c_double * KVFE_N_STAGES),
        ("active_streams", C.c_double * KVFE_N_STAGES),
        ("alg_bytes_per_stream", C.c_double * KVFE_N_STAGES),
        ("active_launches", C.c_int32 * KVFE_N_STAGES),
    ]


class SyncPacket(C.Structure):
    """kvfe_sync_packet (include/kvfe.h, input side)"""
    _fields_ = [("timestamp_ns", C.c_int64), ("left_tag", C.c_int64), ("right_tag", C.c_int64),
                ("n_imu", C.c_int32), ("reserved0", C.c_int32)]


IMU_DATA_AVAILABLE, IMU_DATA_NOT_YET_AVAILABLE, IMU_DATA_NEVER_AVAILABLE, IMU_QUEUE_SHUTDOWN, \
    IMU_TOO_FEW_MEASUREMENTS = range(5)
(SYNC_PACKET, SYNC_EMPTY, SYNC_WAIT_IMU, SYNC_DROP_OUT_OF_ORDER, SYNC_DROP_NO_IMU, SYNC_DROP_FIRST_FRAME,
 SYNC_DROP_IMU_NEVER, SYNC_DROP_IMU_TOO_FEW, SYNC_DROP_NO_RIGHT, SYNC_SHUTDOWN) = range(10)
