"""Field-by-field comparison of one front-end step: GPU library output (kvfe_frontend_get_output)
against the oracle's (StereoVisionImuFrontend::processStereoFrame restatement).  Integer / index /
status fields and float fields alike are compared with tolerance 0 (same IEEE operations in the
same order on both sides, see tests/test_gpu_parity.py)."""
import numpy as np

SCALARS = ("n_keypoints", "is_keyframe", "n_tracked", "n_detected", "n_measurements", "frame_id",
           "tracking_status_mono", "tracking_status_stereo", "nr_mono_putatives", "nr_mono_inliers",
           "nr_stereo_putatives", "nr_stereo_inliers")
FRAME_ARRAYS = ("landmarks", "landmarks_age", "keypoints", "versors")
STEREO_ARRAYS = ("left_rect_xy", "left_status", "right_rect_xy", "right_status", "depth", "right_xy",
                 "keypoints_3d", "meas_landmark")
POSES = ("lkf_T_k_mono", "lkf_T_k_stereo", "info_mat_stereo_translation")


def assert_step_equal(got: dict, exp: dict, where=()):
    for k in SCALARS:
        assert got[k] == exp[k], (*where, k, got[k], exp[k])
    keys = list(FRAME_ARRAYS)
    if exp["has_stereo"] and exp["is_keyframe"]:
        keys += list(STEREO_ARRAYS)
    for k in keys:
        if not np.array_equal(got[k], exp[k]):
            g, e = np.asarray(got[k]), np.asarray(exp[k])
            bad = np.nonzero(np.any(g.reshape(len(g), -1) != e.reshape(len(e), -1), axis=1))[0] if g.shape == e.shape else []
            raise AssertionError((*where, k, g.shape, e.shape, list(bad[:5]),
                                  g[bad[:3]] if len(bad) else None, e[bad[:3]] if len(bad) else None))
    if exp["is_keyframe"]:
        assert np.array_equal(got["meas_uL_uR_v"], exp["meas_uL_uR_v"], equal_nan=True), (*where, "meas_uL_uR_v")
    for k in POSES:
        assert np.array_equal(got[k], exp[k]), (*where, k, got[k], exp[k])
