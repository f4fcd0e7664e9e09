"""ctypes binding of csrc/libkvfe.so (the product).  Fails loudly when the library is missing or
cannot be loaded: there is no CPU fallback anywhere in this package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

from . import _abi as abi

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# KVFE_LIB: another build of the same library (A/B timing of kernel variants on one GPU box)
SO_PATH = os.environ.get("KVFE_LIB") or os.path.join(CSRC, "libkvfe.so")

_lib = None


class KvfeError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        super().__init__(f"{where}: kvfe status {status} {detail}".strip())


def build(force: bool = False) -> str:
    """Compile every HIP extension for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j8"]
    if force:
        cmd.append("-B")
    subprocess.run(cmd, check=True)
    return SO_PATH


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(f"{SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(libkvfe has no CPU fallback)")
    # share one HIP runtime with torch when torch is (or will be) in the process
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the library itself
        pass
    L = C.CDLL(SO_PATH)
    vp, sz, i32, f64 = C.c_void_p, C.c_size_t, C.c_int32, C.c_double
    L.kvfe_version.restype = C.c_char_p
    L.kvfe_status_string.restype = C.c_char_p
    L.kvfe_status_string.argtypes = [i32]
    L.kvfe_last_error.restype = C.c_char_p
    L.kvfe_last_error.argtypes = [vp]
    L.kvfe_default_frontend_params.argtypes = [C.POINTER(abi.FrontendParams)]
    L.kvfe_create.argtypes = [C.POINTER(abi.Config), C.POINTER(vp)]
    L.kvfe_destroy.argtypes = [vp]
    L.kvfe_compute_rectification.argtypes = [C.POINTER(abi.CameraParams), C.POINTER(abi.CameraParams),
                                             C.POINTER(abi.Rectification)]
    L.kvfe_compute_undistort_rectify_maps.argtypes = [C.POINTER(abi.CameraParams), vp, vp, vp, vp]
    L.kvfe_get_rectification.argtypes = [vp, C.POINTER(abi.Rectification)]
    L.kvfe_undistort_rectify_image.argtypes = [vp, i32, vp, sz, vp, sz]
    L.kvfe_undistort_rectify_keypoints.argtypes = [vp, i32, vp, i32, i32, i32, vp]
    L.kvfe_get_bearing_vectors.argtypes = [vp, i32, vp, i32, vp]
    L.kvfe_raw_feature_detection.argtypes = [vp, vp, sz, vp, sz, vp, i32, C.POINTER(i32)]
    L.kvfe_feature_detection.argtypes = [vp, vp, sz, vp, i32, i32, vp, i32, C.POINTER(i32)]
    L.kvfe_corner_subpix.argtypes = [vp, vp, sz, vp, i32, i32, i32, i32, f64]
    L.kvfe_calc_optical_flow_pyr_lk.argtypes = [vp, vp, vp, sz, vp, vp, i32, vp, vp]
    L.kvfe_build_optical_flow_pyramid.argtypes = [vp, vp, sz, sz, i32, vp, sz, vp, vp, vp]
    L.kvfe_predict_sparse_flow.argtypes = [vp, vp, i32, vp, vp]
    L.kvfe_get_right_keypoints_rectified.argtypes = [vp, vp, vp, sz, vp, vp, i32, vp, vp, vp]
    L.kvfe_sparse_stereo_reconstruction.argtypes = [vp, vp, vp, sz, vp, i32,
                                                    C.POINTER(abi.StereoOutput)]
    L.kvfe_outlier_rejection_2d2d_given_rotation.argtypes = [vp, vp, vp, i32, vp, vp,
                                                             C.POINTER(abi.RansacOutput)]
    L.kvfe_outlier_rejection_3d3d_given_rotation.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, vp, vp,
                                                             C.POINTER(abi.RansacOutput)]
    L.kvfe_outlier_rejection_3d3d.argtypes = [vp, vp, vp, i32, vp, C.POINTER(abi.RansacOutput)]
    L.kvfe_pnp.argtypes = [vp, C.POINTER(abi.PnpParams), vp, vp, i32, vp, C.POINTER(abi.RansacOutput)]
    L.kvfe_frontend_update_map.argtypes = [vp, i32, vp, vp, i32]
    L.kvfe_outlier_rejection_2d2d.argtypes = [vp, vp, vp, i32, vp, C.POINTER(abi.RansacOutput)]
    L.kvfe_equalize_hist.argtypes = [vp, vp, sz, vp, sz]
    L.kvfe_dense_stereo_params_default.argtypes = [C.POINTER(abi.DenseStereoParams)]
    L.kvfe_dense_stereo_params_default.restype = None
    L.kvfe_dense_stereo_reconstruction.argtypes = [vp, C.POINTER(abi.DenseStereoParams), i32, vp, vp, sz,
                                                   vp, sz]
    L.kvfe_dense_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.kvfe_backproject_disparity_to_3d.argtypes = [vp, vp, sz, vp]
    L.kvfe_dense_debug_volume.argtypes = [vp, i32, vp, sz]
    L.kvfe_frontend_debug_pyramid.argtypes = [vp, i32, vp, vp, sz]
    L.kvfe_frontend_staging_buffer.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(vp)]
    L.kvfe_frontend_staging_wait.argtypes = [vp, i32]
    L.kvfe_frontend_step_staged.argtypes = [vp, i32, vp]
    L.kvfe_frontend_step_host.argtypes = [vp, vp, vp, sz, sz, vp]
    L.kvfe_frontend_step_device.argtypes = [vp, vp, vp, sz, sz, vp]
    L.kvfe_frontend_reset.argtypes = [vp]
    L.kvfe_synchronize.argtypes = [vp]
    L.kvfe_frontend_get_output.argtypes = [vp, i32, C.POINTER(abi.FrameOutput)]
    L.kvfe_frontend_get_output_at.argtypes = [vp, i32, i32, C.POINTER(abi.FrameOutput)]
    L.kvfe_frontend_get_outputs.argtypes = [vp, i32, C.POINTER(abi.FrameOutput)]
    L.kvfe_frontend_view_output.argtypes = [vp, i32, i32, C.POINTER(abi.FrameOutput)]
    L.kvfe_profile_enable.argtypes = [vp, i32]
    L.kvfe_profile_read.argtypes = [vp, C.POINTER(abi.StageTimes)]
    L.kvfe_hbm_copy_probe.argtypes = [C.c_size_t, i32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.kvfe_hbm_copy_probe.restype = C.c_int32
    f32 = C.c_float
    L.kvfe_check_undistorted_rectified_left_keypoints.argtypes = [vp, i32, vp, vp, i32, f32, vp, vp]
    L.kvfe_distort_unrectify_keypoints.argtypes = [vp, i32, vp, vp, i32, vp]
    L.kvfe_undistort_rectify_left_keypoints.argtypes = [vp, vp, i32, vp, vp]
    L.kvfe_distort_unrectify_right_keypoints.argtypes = [vp, vp, vp, i32, vp]
    L.kvfe_undistort_rectify_stereo_frame.argtypes = [vp, vp, vp, sz, vp, vp, sz]
    L.kvfe_get_depth_from_rectified_matches.argtypes = [vp, vp, vp, vp, vp, i32, vp]
    L.kvfe_feature_detection_frame.argtypes = [vp, vp, sz, C.POINTER(abi.Frame), C.POINTER(C.c_int64)]
    L.kvfe_feature_tracking_frame.argtypes = [vp, vp, vp, sz, C.POINTER(abi.Frame), C.POINTER(abi.Frame), vp]
    for fn in NEW_R2_SYMBOLS:
        getattr(L, fn).restype = C.c_int32
    # input side: host code
    i64, pi32, pi64, pf64 = C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double)
    L.kvfe_png_info.argtypes = [vp, sz, pi32, pi32, pi32]
    L.kvfe_png_decode_gray.argtypes = [vp, sz, vp, sz, i32, i32]
    L.kvfe_jpeg_info.argtypes = [vp, sz, pi32, pi32, pi32]
    L.kvfe_jpeg_decode_gray.argtypes = [vp, sz, vp, sz, i32, i32]
    L.kvfe_png_decode_gray_batch.argtypes = [C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), sz, i32, i32, i32, i32,
                                             pi32]
    L.kvfe_imu_buffer_create.argtypes = [i64]
    L.kvfe_imu_buffer_create.restype = vp
    L.kvfe_imu_buffer_destroy.argtypes = [vp]
    L.kvfe_imu_buffer_destroy.restype = None
    L.kvfe_imu_buffer_add.argtypes = [vp, i64, pf64]
    L.kvfe_imu_buffer_add.restype = None
    L.kvfe_imu_buffer_size.argtypes = [vp]
    L.kvfe_imu_buffer_size.restype = i64
    L.kvfe_imu_buffer_shutdown.argtypes = [vp]
    L.kvfe_imu_buffer_shutdown.restype = None
    L.kvfe_imu_buffer_between.argtypes = [vp, i64, i64, i32, pi64, pf64, i32, pi32]
    L.kvfe_imu_buffer_interpolated_upper_border.argtypes = [vp, i64, i64, pi64, pf64, i32, pi32]
    L.kvfe_imu_buffer_interpolated_borders.argtypes = [vp, i64, i64, pi64, pf64, i32, pi32]
    L.kvfe_imu_linear_interpolate.argtypes = [i64, pf64, i64, pf64, i64, pf64]
    L.kvfe_imu_linear_interpolate.restype = None
    L.kvfe_imu_preintegrate_rotation.argtypes = [pi64, pf64, i32, pf64, pf64]
    L.kvfe_imu_preintegrate_rotation.restype = C.c_int32
    L.kvfe_keyframe_R_cur_frame.argtypes = [pf64, pf64, pf64]
    L.kvfe_keyframe_R_cur_frame.restype = None
    L.kvfe_stereo_sync_create.argtypes = [i64]
    L.kvfe_stereo_sync_create.restype = vp
    L.kvfe_stereo_sync_destroy.argtypes = [vp]
    L.kvfe_stereo_sync_destroy.restype = None
    L.kvfe_stereo_sync_set_mode.argtypes = [vp, i32]
    L.kvfe_stereo_sync_set_mode.restype = C.c_int32
    L.kvfe_stereo_sync_fill_left.argtypes = [vp, i64, i64]
    L.kvfe_stereo_sync_fill_left.restype = None
    L.kvfe_stereo_sync_fill_right.argtypes = [vp, i64, i64]
    L.kvfe_stereo_sync_fill_right.restype = None
    L.kvfe_stereo_sync_fill_imu.argtypes = [vp, i64, pf64]
    L.kvfe_stereo_sync_fill_imu.restype = None
    L.kvfe_stereo_sync_do_coarse_imu_camera_temporal_sync.argtypes = [vp]
    L.kvfe_stereo_sync_do_coarse_imu_camera_temporal_sync.restype = None
    L.kvfe_stereo_sync_set_imu_time_shift.argtypes = [vp, f64]
    L.kvfe_stereo_sync_set_imu_time_shift.restype = None
    L.kvfe_stereo_sync_shutdown.argtypes = [vp]
    L.kvfe_stereo_sync_shutdown.restype = None
    L.kvfe_stereo_sync_next.argtypes = [vp, C.POINTER(abi.SyncPacket), pi64, pf64, i32]
    L.kvfe_euroc_parse_camera_csv.argtypes = [C.c_char_p, sz, pi64, i32, pi32]
    L.kvfe_euroc_parse_imu_csv.argtypes = [C.c_char_p, sz, pi64, pf64, i32, pi32]
    for fn in ("kvfe_png_info", "kvfe_png_decode_gray", "kvfe_png_decode_gray_batch", "kvfe_jpeg_info",
               "kvfe_jpeg_decode_gray", "kvfe_imu_buffer_between",
               "kvfe_imu_buffer_interpolated_upper_border", "kvfe_imu_buffer_interpolated_borders",
               "kvfe_stereo_sync_next", "kvfe_euroc_parse_camera_csv", "kvfe_euroc_parse_imu_csv"):
        getattr(L, fn).restype = C.c_int32
    for fn in ("kvfe_create", "kvfe_compute_rectification", "kvfe_compute_undistort_rectify_maps",
               "kvfe_get_rectification", "kvfe_undistort_rectify_image",
               "kvfe_undistort_rectify_keypoints", "kvfe_get_bearing_vectors",
               "kvfe_raw_feature_detection", "kvfe_feature_detection", "kvfe_corner_subpix",
               "kvfe_calc_optical_flow_pyr_lk", "kvfe_build_optical_flow_pyramid", "kvfe_predict_sparse_flow",
               "kvfe_get_right_keypoints_rectified", "kvfe_sparse_stereo_reconstruction",
               "kvfe_outlier_rejection_2d2d_given_rotation",
               "kvfe_outlier_rejection_3d3d_given_rotation", "kvfe_equalize_hist",
               "kvfe_frontend_staging_buffer", "kvfe_frontend_staging_wait", "kvfe_frontend_step_staged",
               "kvfe_frontend_step_host", "kvfe_frontend_step_device", "kvfe_frontend_reset",
               "kvfe_synchronize", "kvfe_frontend_get_output", "kvfe_frontend_get_output_at", "kvfe_frontend_get_outputs", "kvfe_frontend_view_output",
               "kvfe_profile_enable",
               "kvfe_profile_read", "kvfe_dense_stereo_reconstruction", "kvfe_dense_profile_read",
               "kvfe_backproject_disparity_to_3d", "kvfe_dense_debug_volume", "kvfe_frontend_debug_pyramid",
               "kvfe_outlier_rejection_3d3d",
               "kvfe_outlier_rejection_2d2d"):
        getattr(L, fn).restype = C.c_int32
    _lib = L
    return L


NEW_R6_SYMBOLS = ["kvfe_hbm_copy_probe"]
NEW_R3_SYMBOLS = ["kvfe_build_optical_flow_pyramid"]
NEW_R4_SYMBOLS = ["kvfe_frontend_get_output_at", "kvfe_frontend_get_outputs", "kvfe_frontend_view_output"]

NEW_R2_SYMBOLS = [
    "kvfe_check_undistorted_rectified_left_keypoints", "kvfe_distort_unrectify_keypoints",
    "kvfe_undistort_rectify_left_keypoints", "kvfe_distort_unrectify_right_keypoints",
    "kvfe_undistort_rectify_stereo_frame", "kvfe_get_depth_from_rectified_matches",
    "kvfe_feature_detection_frame", "kvfe_feature_tracking_frame", "kvfe_pnp", "kvfe_frontend_update_map",
]

INPUT_SIDE_SYMBOLS = [
    "kvfe_png_info", "kvfe_png_decode_gray", "kvfe_png_decode_gray_batch", "kvfe_jpeg_info", "kvfe_jpeg_decode_gray",
    "kvfe_imu_buffer_create",
    "kvfe_imu_buffer_destroy", "kvfe_imu_buffer_add", "kvfe_imu_buffer_size", "kvfe_imu_buffer_shutdown",
    "kvfe_imu_buffer_between", "kvfe_imu_buffer_interpolated_upper_border", "kvfe_imu_buffer_interpolated_borders",
    "kvfe_imu_linear_interpolate", "kvfe_imu_preintegrate_rotation", "kvfe_keyframe_R_cur_frame",
    "kvfe_stereo_sync_create", "kvfe_stereo_sync_destroy", "kvfe_stereo_sync_set_mode",
    "kvfe_stereo_sync_fill_left", "kvfe_stereo_sync_fill_right", "kvfe_stereo_sync_fill_imu",
    "kvfe_stereo_sync_do_coarse_imu_camera_temporal_sync", "kvfe_stereo_sync_set_imu_time_shift",
    "kvfe_stereo_sync_shutdown", "kvfe_stereo_sync_next", "kvfe_euroc_parse_camera_csv", "kvfe_euroc_parse_imu_csv",
]

EXPORTED_SYMBOLS = NEW_R6_SYMBOLS + NEW_R4_SYMBOLS + NEW_R3_SYMBOLS + NEW_R2_SYMBOLS + INPUT_SIDE_SYMBOLS + [
    "kvfe_version", "kvfe_status_string", "kvfe_last_error", "kvfe_default_frontend_params",
    "kvfe_create", "kvfe_destroy", "kvfe_compute_rectification",
    "kvfe_compute_undistort_rectify_maps", "kvfe_get_rectification", "kvfe_undistort_rectify_image",
    "kvfe_undistort_rectify_keypoints", "kvfe_get_bearing_vectors", "kvfe_raw_feature_detection",
    "kvfe_feature_detection", "kvfe_corner_subpix", "kvfe_calc_optical_flow_pyr_lk",
    "kvfe_predict_sparse_flow", "kvfe_get_right_keypoints_rectified",
    "kvfe_sparse_stereo_reconstruction", "kvfe_outlier_rejection_2d2d_given_rotation",
    "kvfe_outlier_rejection_3d3d_given_rotation", "kvfe_equalize_hist", "kvfe_frontend_staging_buffer",
    "kvfe_frontend_staging_wait", "kvfe_frontend_step_staged", "kvfe_frontend_step_host",
    "kvfe_frontend_step_device",
    "kvfe_frontend_reset", "kvfe_synchronize", "kvfe_frontend_get_output", "kvfe_profile_enable",
    "kvfe_profile_read", "kvfe_dense_stereo_params_default", "kvfe_dense_stereo_reconstruction",
    "kvfe_dense_profile_read", "kvfe_backproject_disparity_to_3d", "kvfe_dense_debug_volume",
    "kvfe_frontend_debug_pyramid",
    "kvfe_outlier_rejection_3d3d", "kvfe_outlier_rejection_2d2d",
]
