/*
 * kvfe.h — C ABI of libkvfe: an MI355X-native (gfx950 / HIP) stereo visual
 * front-end that drops in behind Kimera-VIO's
 *   StereoVisionImuFrontend / FeatureDetector / Tracker / StereoMatcher /
 *   UndistorterRectifier / StereoCamera
 * C++ classes.  Kimera-VIO has no FFI layer of its own; the boundary is its
 * concrete class API.  Every entry point below names the reference method it
 * replaces (paths relative to the Kimera-VIO tree).  INTEGRATION.md shows the
 * reference-side adapter a maintainer would add; include/kvfe_adapter.hpp is
 * that adapter written against plain structs.
 *
 * Conventions
 *  - plain pointers + sizes, no C++/torch types; all functions return
 *    kvfe_status (0 = ok, <0 = error), never throw, never abort.
 *  - images: 8-bit gray, row-major, `stride` in bytes.
 *  - keypoints: interleaved float32 (x, y) pairs == cv::Point2f.
 *  - per-point status bytes use the values of VIO::KeypointStatus
 *    (include/kimera-vio/common/vio_types.h:38-44).
 *  - a kvfe_ctx is single-threaded (one HIP stream); distinct contexts are
 *    independent and may live on different threads / GPUs.
 */
#ifndef KVFE_H_
#define KVFE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KVFE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* status / enums                                                            */
/* ------------------------------------------------------------------------- */
typedef int32_t kvfe_status;
enum {
  KVFE_OK = 0,
  KVFE_ERR_INVALID_ARG = -1,   /* reference: CHECK_* abort                    */
  KVFE_ERR_UNSUPPORTED = -2,   /* reference: LOG(FATAL) "not implemented"     */
  KVFE_ERR_NO_DEVICE = -3,     /* no gfx950 device / HIP runtime error        */
  KVFE_ERR_HIP = -4,
  KVFE_ERR_CAPACITY = -5,      /* a device-side list overflowed its capacity  */
  KVFE_ERR_NOT_READY = -6
};

/* VIO::KeypointStatus, include/kimera-vio/common/vio_types.h:38-44 */
enum {
  KVFE_KP_VALID = 0,
  KVFE_KP_NO_LEFT_RECT = 1,
  KVFE_KP_NO_RIGHT_RECT = 2,
  KVFE_KP_NO_DEPTH = 3,
  KVFE_KP_FAILED_ARUN = 4
};

/* VIO::TrackingStatus, include/kimera-vio/frontend/Tracker-definitions.h:126-132 */
enum {
  KVFE_TRACKING_VALID = 0,
  KVFE_TRACKING_LOW_DISPARITY = 1,
  KVFE_TRACKING_FEW_MATCHES = 2,
  KVFE_TRACKING_INVALID = 3,
  KVFE_TRACKING_DISABLED = 4
};

/* std::uniform_int_distribution<int>(0, INT_MAX) over std::mt19937, as drawn by
 * opengv::sac::SampleConsensusProblem::rnd(), is implementation defined.
 * PRE11: libstdc++ <= 10 (GCC 9.4 of the reference's Ubuntu 20.04 image) redraws
 * while the 32-bit output is >= 2^31; V11: libstdc++ >= 11 returns output >> 1. */
enum { KVFE_RNG_LIBSTDCXX_PRE11 = 0, KVFE_RNG_LIBSTDCXX_11 = 1 };

/* VIO::FrontendType (stereo: StereoVisionImuFrontend, mono: MonoVisionImuFrontend, rgbd:
 * RgbdVisionImuFrontend) */
enum { KVFE_FRONTEND_STEREO = 0, KVFE_FRONTEND_MONO = 1, KVFE_FRONTEND_RGBD = 2 };
/* cv::Mat type of the depth image of the RGBD front-end (DepthFrame.cpp:30) */
enum { KVFE_DEPTH_U16 = 0 /* CV_16UC1 */, KVFE_DEPTH_F32 = 1 /* CV_32FC1 */ };

/* VIO::AnmsAlgorithmType (feature-detector/NonMaximumSuppression.h) */
enum {
  KVFE_ANMS_TOPN = 0,
  KVFE_ANMS_BROWN = 1,
  KVFE_ANMS_SDC = 2,
  KVFE_ANMS_KDTREE = 3,
  KVFE_ANMS_RANGETREE = 4,
  KVFE_ANMS_SSC = 5,
  KVFE_ANMS_BINNING = 6
};

/* VIO::FeatureDetectorType */
enum { KVFE_DET_FAST = 0, KVFE_DET_ORB = 1, KVFE_DET_AGAST = 2, KVFE_DET_GFTT = 3 };

/* VIO::OpticalFlowPredictorType */
enum { KVFE_FLOW_NO_PREDICTION = 0, KVFE_FLOW_ROTATIONAL = 1 };

/* VIO::DistortionModel (RADTAN = cv::undistortPoints family, EQUIDISTANT = cv::fisheye family;
 * the omni model of Camera.cpp is not implemented) */
enum { KVFE_DIST_NONE = 0, KVFE_DIST_RADTAN = 1, KVFE_DIST_EQUIDISTANT = 2 };

/* how cv::sortIdx's all-equal-keys permutation is reproduced before ANMS
 * (src/frontend/feature-detector/NonMaximumSuppression.cpp:50-60): the
 * reference truncates the GFTT responses to int (all 0) and calls
 * cv::sortIdx(DESCENDING) = std::sort with an always-false comparator followed
 * by a reversal.  LIBSTDCXX reproduces libstdc++'s introsort permutation
 * (what the reference binary really does); STABLE keeps quality order. */
enum { KVFE_SORTIDX_LIBSTDCXX = 0, KVFE_SORTIDX_STABLE = 1 };

#define KVFE_MAX_BINS 256
#define KVFE_MAX_DIST_COEFFS 8

/* ------------------------------------------------------------------------- */
/* parameter structs (PODs mirroring the reference's param classes)          */
/* ------------------------------------------------------------------------- */

/* VIO::CameraParams (include/kimera-vio/frontend/CameraParams.h) */
typedef struct kvfe_camera_params {
  int32_t width, height;                     /* image_size_                  */
  double intrinsics[4];                      /* fu, fv, cu, cv               */
  int32_t distortion_model;                  /* KVFE_DIST_*                  */
  int32_t n_distortion;                      /* >= 4 for radtan              */
  double distortion[KVFE_MAX_DIST_COEFFS];   /* k1 k2 p1 p2 [k3 k4 k5 k6]    */
  double body_pose_cam[16];                  /* T_BS, row-major 4x4          */
} kvfe_camera_params;

/* VIO::FeatureDetectorParams + SubPixelCornerFinderParams
 * (include/kimera-vio/frontend/feature-detector/FeatureDetectorParams.h:25-106) */
typedef struct kvfe_detector_params {
  int32_t feature_detector_type;             /* KVFE_DET_GFTT (all shipped parameter sets) or KVFE_DET_FAST
                                                (cv::FastFeatureDetector(fast_thresh, true)); ORB / AGAST:
                                                KVFE_ERR_UNSUPPORTED            */
  int32_t max_features_per_frame;            /* maxFeaturesPerFrame          */
  int32_t enable_subpixel_corner_refinement;
  int32_t subpix_window_size;                /* window_size (half window)    */
  int32_t subpix_zero_zone;                  /* zero_zone                    */
  int32_t subpix_max_iters;                  /* max_iters                    */
  double subpix_epsilon;                     /* epsilon_error                */
  int32_t enable_non_max_suppression;
  int32_t non_max_suppression_type;          /* KVFE_ANMS_*                  */
  int32_t min_distance;                      /* min_distance (GFTT + mask r) */
  int32_t max_nr_keypoints_before_anms;
  int32_t nr_horizontal_bins, nr_vertical_bins;
  uint8_t binning_mask[KVFE_MAX_BINS];       /* row-major [vbin][hbin], 0/1  */
  double quality_level;
  int32_t block_size;
  int32_t use_harris_detector;
  double k;
  int32_t sortidx_policy;                    /* KVFE_SORTIDX_*               */
  int32_t fast_thresh;                       /* fast_thresh (KVFE_DET_FAST)  */
} kvfe_detector_params;

/* VIO::TrackerParams (include/kimera-vio/frontend/VisionImuTrackerParams.h:24-85).
 * Geometric outlier rejection (FrontendParams::useRANSAC_) is implemented for
 * ransac_use_2point_mono (opengv TranslationOnlySacProblem), its 5-point alternative
 * (ransac_use_2point_mono = 0: CentralRelativePoseSacProblem, NISTER), ransac_use_1point_stereo
 * (the reference's own voting scheme) and the 3-point Arun problem the stereo branch falls back
 * to (ransac_use_1point_stereo = 0, or a keyframe without gyro rotation); PnP tracking (EPNP) is the
 * separate call kvfe_pnp with its own kvfe_pnp_params.  The other 2d2d algorithms and
 * ransac_randomize = 1 are KVFE_ERR_UNSUPPORTED. */
typedef struct kvfe_tracker_params {
  int32_t klt_win_size;
  int32_t klt_max_iter;
  int32_t klt_max_level;
  int32_t max_feature_track_age;             /* maxFeatureAge                */
  double klt_eps;
  int32_t optical_flow_predictor_type;       /* KVFE_FLOW_*                  */
  int32_t reserved0;
  double disparity_threshold;                /* disparityThreshold           */
  int32_t min_nr_mono_inliers;               /* minNrMonoInliers             */
  int32_t min_nr_stereo_inliers;             /* minNrStereoInliers           */
  double ransac_threshold_mono;              /* 1 - cos on bearing vectors   */
  double ransac_threshold_stereo;            /* Mahalanobis (float32 test)   */
  int32_t ransac_max_iterations;
  int32_t ransac_randomize;                  /* must be 0 (fixed seed 12345) */
  double ransac_probability;
  int32_t ransac_use_1point_stereo;
  int32_t ransac_use_2point_mono;
  int32_t ransac_rng_policy;                 /* KVFE_RNG_*                   */
  int32_t pose_2d2d_algorithm;               /* 2d2d_algorithm (opengv CentralRelativePoseSacProblem::
                                                Algorithm): 1 = NISTER in every shipped YAML; the only
                                                one implemented for ransac_use_2point_mono = 0        */
} kvfe_tracker_params;

/* VIO::StereoMatchingParams (include/kimera-vio/frontend/StereoMatchingParams.h:24-60) */
typedef struct kvfe_stereo_params {
  double tolerance_template_matching;
  int32_t templ_cols, templ_rows, stripe_extra_rows;
  int32_t subpixel_refinement;
  double min_point_dist, max_point_dist;
  int32_t equalize_image;                    /* equalizeImage: cv::equalizeHist on both
                                                input images (UtilsOpenCV.cpp:398-401) */
  int32_t ssd_tie_policy;                    /* KVFE_SSD_TIE_*: which minimum searchRightKeypointEpipolar takes when
                                                candidates are (nearly) tied -- an implementation-defined corner of
                                                cv::matchTemplate + minMaxLoc (StereoMatcher.cpp:388-392), exposed like
                                                sortidx_policy instead of guessed.  On the reference's EuRoC frames it
                                                decides 0 of 16 209 matches (profiles/r4_ssd_tie_exposure.md)          */
} kvfe_stereo_params;
#define KVFE_SSD_TIE_EXACT 0   /* first minimum, row-major, of the EXACT integer SSDs (default)                        */
#define KVFE_SSD_TIE_F32 1     /* first minimum of the SSDs rounded to float32 (round to nearest even): the CV_32F
                                  result matrix of cv::matchTemplate without its DFT rounding noise                    */

/* the PnP members of VIO::TrackerParams (VisionImuTrackerParams.h:72-76), see kvfe_pnp */
typedef struct kvfe_pnp_params {
  int32_t pnp_algorithm;                     /* Pose3d2dAlgorithm: 3 = EPNP, 1 = KneipP3P              */
  int32_t min_pnp_inliers;
  double ransac_threshold_pnp;               /* pixels                                                 */
  int32_t optimize_2d3d_pose_from_inliers;   /* must be 0                                              */
  int32_t reserved0;
} kvfe_pnp_params;

/* VIO::FrontendParams (include/kimera-vio/frontend/VisionImuFrontendParams.h:25-76) */
typedef struct kvfe_frontend_params {
  kvfe_detector_params detector;
  kvfe_tracker_params tracker;
  kvfe_stereo_params stereo;
  double min_intra_keyframe_time_ns;
  double max_intra_keyframe_time_ns;
  int64_t min_number_features;
  double max_disparity_since_lkf;
  int32_t use_stereo_tracking;
  int32_t use_ransac;                        /* useRANSAC                    */
  int32_t use_pnp_tracking;                  /* use_pnp_tracking: Tracker::pnp on keyframes against the
                                                landmark map of kvfe_frontend_update_map
                                                (StereoVisionImuFrontend.cpp:389-399,
                                                RgbdVisionImuFrontend.cpp:328-341)            */
  int32_t reserved1;
  kvfe_pnp_params pnp;
} kvfe_frontend_params;

/* CameraParams::DepthParams (include/kimera-vio/frontend/CameraParams.h:131-155), RGBD front-end only.
 * is_registered must be 1 (an unregistered depth image goes through cv::rgbd::registerDepth in the
 * reference, DepthFrame.cpp:97-110: not implemented). */
typedef struct kvfe_depth_params {
  float virtual_baseline;        /* 1e-2: baseline of the hallucinated right camera          */
  float depth_to_meters;         /* 1.0                                                      */
  float min_depth;               /* 0.0                                                      */
  float max_depth;               /* 10.0                                                     */
  int32_t is_registered;         /* 1                                                        */
  int32_t depth_type;            /* KVFE_DEPTH_U16 | KVFE_DEPTH_F32                          */
} kvfe_depth_params;

typedef struct kvfe_config {
  kvfe_camera_params left, right;
  kvfe_frontend_params params;
  int32_t batch;                 /* number of independent stereo streams     */
  int32_t device;                /* HIP device ordinal                       */
  void* hip_stream;              /* optional hipStream_t owned by the caller: every step is then
                                    stream-ordered on it as a whole (work the caller enqueues on the
                                    stream after a step call runs after the step, including the part
                                    the library runs on its internal side stream).  With a
                                    library-owned stream (NULL) completion is defined by
                                    kvfe_synchronize / kvfe_frontend_get_output only            */
  int32_t candidate_capacity;    /* per-stream GFTT candidate cap, 0 = default: max(W * H / 4, 4096).
                                    Capacities of the detection stage (a step that exceeds one sets
                                    KVFE_ERR_CAPACITY on that stream's output -- never a silently shorter
                                    list): candidates above the quality level <= candidate_capacity; corners
                                    the minimum-distance filter accepts <= 8192 when
                                    max_nr_keypoints_before_anms is <= 0 or larger than that (cv::
                                    goodFeaturesToTrack stops at maxCorners; the shipped 2000 never gets
                                    there); keypoints into the tree / range / SSC non-maximum suppression
                                    <= 4096; kvfe_create refuses max_features_per_frame +
                                    max_nr_keypoints_before_anms + 64 > 4877 keypoints per frame          */
  int32_t frontend_type;         /* KVFE_FRONTEND_*: stereo (default), the monocular front-end
                                    (MonoVisionImuFrontend.cpp: `left` only, `right` ignored) or
                                    the RGBD front-end (RgbdVisionImuFrontend.cpp: `left` is the
                                    colour camera; the `right` images of the step calls are the
                                    depth images, see kvfe_frontend_step_host)                   */
  int32_t landmark_map_capacity; /* landmarks per stream kvfe_frontend_update_map can hold (0 = 8192) */
  kvfe_depth_params depth;       /* RGBD front-end only                                          */
  int32_t stream_groups;         /* 0 = default (1).  The batch is split into this many
                                    groups of streams, each on its own HIP stream, so
                                    that latency-bound and throughput-bound kernels of
                                    different groups can overlap (always 1 with
                                    hip_stream).  Measured on MI355X: 1 is fastest    */
  /* ---- execution options (round 4: these were environment variables of the library) ---- */
  int32_t device_frames_persist; /* kvfe_frontend_step_device only.  1 = the caller guarantees that BOTH images of
                                    a step stay valid and unchanged until the NEXT step has completed (frames
                                    resident in a device ring of >= 2 entries): tracking then reads frame k-1 from
                                    the caller's buffer and the context does not keep its own copy (one image write
                                    per frame less), and the rectification / matching chain of step k -- which
                                    reads both images -- is joined by step k+1's keyframe decision instead of in
                                    front of its tracking launch.  0 (default): no caller
                                    pointer is read after the step it was passed to has COMPLETED -- not "after the
                                    call returned": the call only enqueues, and from round 5 on the step's
                                    rectification / matching chain (which reads both images) runs on the side
                                    stream beside the NEXT steps' tracking.  A caller that rewrites one buffer pair
                                    must wait for the step (kvfe_synchronize, kvfe_frontend_get_output*) first;
                                    rewriting it after merely enqueueing the next step is a race               */
  int32_t single_hip_stream;     /* 1 = every kernel of a step on the context's one HIP stream: no internal side
                                    stream (corner refinement beside rectification / matching) and no output stream.
                                    For callers that need strict single-stream order and for timing kernels alone  */
  int32_t copy_inputs;           /* 1 = the per-stream inputs of a step travel by one H2D copy instead of being read
                                    by the kernels from the mapped pinned ring slot                                */
  int32_t ssd_impl;              /* sparse stereo SSD search (searchRightKeypointEpipolar): 0 = on the matrix cores
                                    where the template / stripe geometry fits their lane maps (default), 1 = the
                                    v_dot4 search for every geometry                                               */
  int32_t lk_impl;               /* pyramidal Lucas-Kanade launch of the front-end step (Tracker::featureTracking):
                                    0 = four points per wavefront, a quad of lanes per SSE lane class (k_lk4.hip; the
                                    24-pixel window, pyramid levels of at least 32 x 32; default), 1 = one wavefront
                                    per point (k_track.hip lk_kernel_sys) for every window.  Results are bit-identical; the component call kvfe_lk_track,
                                    which returns the error array, always takes the second                         */
} kvfe_config;

typedef struct kvfe_ctx kvfe_ctx;

/* rectification constants computed at creation
 * (StereoCamera::computeRectificationParameters, src/frontend/StereoCamera.cpp:292-379) */
typedef struct kvfe_rectification {
  double R1[9], R2[9], P1[12], P2[12], Q[16];
  int32_t roi1[4], roi2[4];
  double baseline;               /* 1 / Q(3,2), StereoCamera.cpp:70-72       */
} kvfe_rectification;

/* ------------------------------------------------------------------------- */
/* lifecycle                                                                 */
/* ------------------------------------------------------------------------- */
KVFE_API const char* kvfe_version(void);
KVFE_API const char* kvfe_status_string(kvfe_status s);
KVFE_API const char* kvfe_last_error(const kvfe_ctx* ctx);

/* fills *p with the class defaults of FeatureDetectorParams / TrackerParams /
 * StereoMatchingParams / FrontendParams (header defaults, not the YAML ones). */
KVFE_API void kvfe_default_frontend_params(kvfe_frontend_params* p);

/* StereoCamera::StereoCamera + UndistorterRectifier ctor x2 + FeatureDetector
 * ctor + Tracker ctor (src/frontend/StereoCamera.cpp:34-94,
 * UndistorterRectifier.cpp:26-31, FeatureDetector.cpp:18-89, Tracker.cpp:54-86).
 * Fails with KVFE_ERR_NO_DEVICE when no gfx950 device is usable: there is no
 * CPU fallback. */
KVFE_API kvfe_status kvfe_create(const kvfe_config* cfg, kvfe_ctx** out);
KVFE_API void kvfe_destroy(kvfe_ctx* ctx);

/* host-only part of creation (no device needed): stereoRectify + maps.
 * Used by the CPU-side tests of the host logic. */
KVFE_API kvfe_status kvfe_compute_rectification(const kvfe_camera_params* left,
                                                const kvfe_camera_params* right,
                                                kvfe_rectification* out);
/* UndistorterRectifier::initUndistortRectifyMaps (UndistorterRectifier.cpp:230-292).
 * cam: 0 = left, 1 = right; map_x/map_y: width*height float32 each. */
KVFE_API kvfe_status kvfe_compute_undistort_rectify_maps(
    const kvfe_camera_params* cam, const double R[9], const double P[12],
    float* map_x, float* map_y);

KVFE_API kvfe_status kvfe_get_rectification(const kvfe_ctx* ctx,
                                            kvfe_rectification* out);

/* ------------------------------------------------------------------------- */
/* component level: host buffers in, host buffers out (one stream)           */
/* ------------------------------------------------------------------------- */

/* UndistorterRectifier::undistortRectifyImage (UndistorterRectifier.cpp:115-128)
 * == cv::remap(INTER_LINEAR, BORDER_REPLICATE).  cam: 0 left, 1 right. */
KVFE_API kvfe_status kvfe_undistort_rectify_image(kvfe_ctx* ctx, int32_t cam,
                                                  const uint8_t* src, size_t src_stride,
                                                  uint8_t* dst, size_t dst_stride);

/* UndistorterRectifier::UndistortRectifyKeypoints (UndistorterRectifier.cpp:33-68)
 * == cv::undistortPoints(K, D, R?, P?) of camera `cam`.
 * use_R/use_P select the rectifier's own R/P (R1,P1 or R2,P2). */
KVFE_API kvfe_status kvfe_undistort_rectify_keypoints(kvfe_ctx* ctx, int32_t cam,
                                                      const float* xy, int32_t n,
                                                      int32_t use_R, int32_t use_P,
                                                      float* out_xy);

/* UndistorterRectifier::GetBearingVector (UndistorterRectifier.cpp:73-113),
 * batched: out_versors is n x 3 float64, unit norm. */
KVFE_API kvfe_status kvfe_get_bearing_vectors(kvfe_ctx* ctx, int32_t cam,
                                              const float* xy, int32_t n,
                                              double* out_versors);

/* cv::equalizeHist as applied by UtilsOpenCV::ReadAndConvertToGrayScale(img, equalize = true)
 * (src/utils/UtilsOpenCV.cpp:390-403; data provider side). */
KVFE_API kvfe_status kvfe_equalize_hist(kvfe_ctx* ctx, const uint8_t* src, size_t src_stride,
                                        uint8_t* dst, size_t dst_stride);

/* ---- dense stereo -------------------------------- */
/* DenseStereoParams (include/kimera-vio/frontend/StereoMatchingParams.h:39-58).  The reference
 * never parses these from YAML (StereoMatcher.cpp:30 default-constructs them), so
 * kvfe_dense_stereo_params_default() is what denseStereoReconstruction runs with. */
typedef struct kvfe_dense_stereo_params {
  int32_t use_sgbm;               /* use_sgbm_ = true                                  */
  int32_t post_filter_disparity;  /* post_filter_disparity_ = false (a no-op upstream) */
  int32_t median_blur_disparity;  /* median_blur_disparity_ = false: cv::medianBlur(5) */
  int32_t pre_filter_cap;         /* 31 */
  int32_t sad_window_size;        /* 11 */
  int32_t min_disparity;          /* 1  */
  int32_t num_disparities;        /* 64 */
  int32_t uniqueness_ratio;       /* 0  */
  int32_t speckle_range;          /* 3  */
  int32_t speckle_window_size;    /* 500 */
  int32_t texture_threshold;      /* 0 (BM) */
  int32_t pre_filter_type;        /* cv::StereoBM::PREFILTER_XSOBEL = 1 (BM) */
  int32_t pre_filter_size;        /* 9 (BM, unused by XSOBEL) */
  int32_t p1;                     /* 120 */
  int32_t p2;                     /* 240 */
  int32_t disp_12_max_diff;       /* -1 */
  int32_t use_mode_hh;            /* use_mode_HH_ = true: cv::StereoSGBM::MODE_HH */
  int32_t reserved0;
} kvfe_dense_stereo_params;
KVFE_API void kvfe_dense_stereo_params_default(kvfe_dense_stereo_params* p);

/* StereoMatcher::denseStereoReconstruction(left_img_rectified, right_img_rectified, disparity_img)
 * (StereoMatcher.cpp:32-121), batched over n_pairs independent rectified pairs of the context's
 * image size: cv::StereoSGBM::compute (MODE_HH, or MODE_SGBM with use_mode_hh = 0; BT pixel cost,
 * 8- / 5-path aggregation, uniqueness, sub-pixel fit, left-right check, 3x3 median,
 * cv::filterSpeckles) and the optional 5x5 median.
 * disparity: n_pairs images of int16 with 4 fractional bits, (min_disparity - 1) * 16 where
 * invalid — the CV_16S matrix cv::StereoSGBM::compute leaves in *disparity_img (compute()
 * re-creates the CV_32F matrix the caller passed as CV_16S; callers divide by 16:
 * tests/testStereoCamera.cpp:296-300).  use_sgbm = 0 runs cv::StereoBM (PREFILTER_XSOBEL; ROI1/ROI2
 * = the context's rectification ROIs, StereoMatcher.cpp:81-86).  KVFE_ERR_UNSUPPORTED:
 * num_disparities > 64, SGBM with min_disparity < 0, BM's normalized-response prefilter or
 * disp_12_max_diff >= 0, cost ranges that
 * leave 16 bits (sad_window_size^2 * 125 + 2 * p2 > 16383).
 * Device memory (allocated at the first call, kept until the context is destroyed, sized by the largest
 * group of pairs a call has launched -- at most 8): three cost volumes of H x width1 x D int16 per pair
 * (42 MB per 752 x 480 pair at 64 disparities) and, for MODE_HH with four or more pairs in a launch, the
 * hand-over buffer of the two-pass aggregation, (row bands - 1) x 2 x pairs x width1 x 528 bytes (180 MB
 * for 8 such pairs); the latter is zeroed whenever the number of pairs in a launch changes (its entries
 * carry a 2-bit launch tag), so a call with 12 pairs (groups of 8 and 4) pays ~50 us per group for it.
 * The waves of that launch wait for each other with BOUNDED polls; if one runs out (a preempted or
 * debugged process) the group is repeated on the eight independent direction sweeps -- same result,
 * ~2.5 x the time -- and kvfe_last_error says so while the call returns KVFE_OK. */
KVFE_API kvfe_status kvfe_dense_stereo_reconstruction(kvfe_ctx* ctx,
                                                      const kvfe_dense_stereo_params* params,
                                                      int32_t n_pairs,
                                                      const uint8_t* const* left_rect,
                                                      const uint8_t* const* right_rect,
                                                      size_t stride,
                                                      int16_t* const* disparity,
                                                      size_t disparity_stride_elems);

/* measurement hook: summed HIP-event time of the kernel sequences of the
 * kvfe_dense_stereo_reconstruction calls since the last read (copies excluded), and the
 * number of pairs they processed; both counters are reset. */
KVFE_API kvfe_status kvfe_dense_profile_read(kvfe_ctx* ctx, double* kernel_ms, int64_t* pairs);

/* test hook: cost volumes [H][width1][D] int16 of the first pair of the last
 * kvfe_dense_stereo_reconstruction call: which = 2: C(p,d) of computeDisparitySGBM (with its +P2
 * bias); 0 and 1: the partial sums of the path costs (uint16): MODE_SGBM and StereoBM keep everything
 * in volume 0; MODE_HH adds four directions into each (pass 1 / pass 2 of computeDisparitySGBM), and
 * S of OpenCV is min(32767, volume 0 + volume 1). */
KVFE_API kvfe_status kvfe_dense_debug_volume(kvfe_ctx* ctx, int32_t which, int16_t* out,
                                             size_t elems);

/* StereoCamera::backProjectDisparityTo3D (StereoCamera.cpp:176-196) ==
 * cv::reprojectImageTo3D(disparity CV_32F, depth CV_32FC3, Q, handleMissingValues = true):
 * disparity is the float image (int16 / 16), xyz is h x w x 3 float; pixels at the minimum
 * disparity of the image get z = 10000.  Q is the context's rectification Q. */
KVFE_API kvfe_status kvfe_backproject_disparity_to_3d(kvfe_ctx* ctx, const float* disparity,
                                                      size_t stride_elems, float* xyz);

/* FeatureDetector::rawFeatureDetection (FeatureDetector.cpp:165-172)
 * == cv::GFTTDetector::detect(img, kps, mask); mask may be NULL (all 255).
 * Returns integer-valued corners in quality-descending order. */
KVFE_API kvfe_status kvfe_raw_feature_detection(kvfe_ctx* ctx,
                                                const uint8_t* img, size_t stride,
                                                const uint8_t* mask, size_t mask_stride,
                                                float* out_xy, int32_t capacity,
                                                int32_t* out_n);

/* private FeatureDetector::featureDetection(const Frame&, need_n_corners)
 * (FeatureDetector.cpp:174-299): mask discs around the tracked keypoints,
 * GFTT, ANMS, cornerSubPix.  tracked_xy: keypoints with landmark != -1. */
KVFE_API kvfe_status kvfe_feature_detection(kvfe_ctx* ctx,
                                            const uint8_t* img, size_t stride,
                                            const float* tracked_xy, int32_t n_tracked,
                                            int32_t need_n_corners,
                                            float* out_xy, int32_t capacity,
                                            int32_t* out_n);

/* cv::cornerSubPix as called at FeatureDetector.cpp:288-292 /
 * StereoMatcher.cpp:404-413; xy is refined in place. */
KVFE_API kvfe_status kvfe_corner_subpix(kvfe_ctx* ctx,
                                        const uint8_t* img, size_t stride,
                                        float* xy, int32_t n,
                                        int32_t half_win, int32_t zero_zone,
                                        int32_t max_iters, double eps);

/* cv::calcOpticalFlowPyrLK as called at Tracker.cpp:137-146
 * (OPTFLOW_USE_INITIAL_FLOW, minEigThreshold 1e-4); cur_xy holds the initial
 * guess on entry and the tracked positions on return. */
KVFE_API kvfe_status kvfe_calc_optical_flow_pyr_lk(kvfe_ctx* ctx,
                                                   const uint8_t* prev_img,
                                                   const uint8_t* cur_img, size_t stride,
                                                   const float* prev_xy, float* cur_xy,
                                                   int32_t n, uint8_t* status,
                                                   float* err);

/* cv::buildOpticalFlowPyramid as cv::calcOpticalFlowPyrLK runs it on both images
 * (Tracker.cpp:137-146): the cv::pyrDown chain, levels 1..L of `n_images` images of
 * the context's size (L from klt_max_level and the window, as in OpenCV).
 * levels_out: per image the levels 1..L densely packed one after the other;
 * level_sizes_out[2 l], [2 l + 1] = width, height of level l = 0..L (may be NULL);
 * level0_copy_out (may be NULL): the dense level-0 copy the device-pointer step
 * keeps for the next frame's tracking (written by the same launch). */
KVFE_API kvfe_status kvfe_build_optical_flow_pyramid(kvfe_ctx* ctx, const uint8_t* imgs,
                                                     size_t row_stride, size_t image_stride,
                                                     int32_t n_images, uint8_t* levels_out,
                                                     size_t levels_capacity,
                                                     int32_t* level_sizes_out,
                                                     int32_t* n_levels_out,
                                                     uint8_t* level0_copy_out);

/* test hook: the front-end's OWN pyramid of the frame of the last step (steps_back 0) or of the one
 * before (steps_back 1) -- what the next tracking launch reads as its previous frame -- after awaiting
 * the context.  levels_out: per stream the levels 1..L packed as kvfe_build_optical_flow_pyramid packs
 * them; level0_copy_out: [batch][H][W], the context-owned level-0 copy of device-pointer steps with
 * device_frames_persist 0 (KVFE_ERR_UNSUPPORTED on other paths).  Either may be NULL.  One stream
 * group only.  (tools/r6/pyr_probe.py: the pyramid launch checked in place, step by step.) */
KVFE_API kvfe_status kvfe_frontend_debug_pyramid(kvfe_ctx* ctx, int32_t steps_back,
                                                 uint8_t* level0_copy_out, uint8_t* levels_out,
                                                 size_t levels_capacity);

/* OpticalFlowPredictor::predictSparseFlow
 * (optical-flow/OpticalFlowPredictor.cpp:27-33,70-126); ref_R_cur row-major. */
KVFE_API kvfe_status kvfe_predict_sparse_flow(kvfe_ctx* ctx, const float* prev_xy,
                                              int32_t n, const double ref_R_cur[9],
                                              float* out_xy);

/* StereoMatcher::getRightKeypointsRectified (StereoMatcher.cpp:196-281):
 * rectified images in, integer right matches out; score = min_val after
 * cv::normalize (always 0 for found matches, -1 for rejected placement). */
KVFE_API kvfe_status kvfe_get_right_keypoints_rectified(
    kvfe_ctx* ctx, const uint8_t* left_rect, const uint8_t* right_rect, size_t stride,
    const float* left_rect_xy, const uint8_t* left_status, int32_t n,
    float* right_rect_xy, uint8_t* right_status, double* score);

/* StereoMatcher::sparseStereoReconstruction(StereoFrame*)
 * (StereoMatcher.cpp:123-175) for one stereo pair given the raw images and the
 * left keypoints.  All outputs have length n; any may be NULL. */
typedef struct kvfe_stereo_output {
  float* left_rect_xy;        /* left_keypoints_rectified_ (.second)          */
  uint8_t* left_status;       /* left_keypoints_rectified_ (.first)           */
  float* right_rect_xy;       /* right_keypoints_rectified_ (.second)         */
  uint8_t* right_status;      /* right_keypoints_rectified_ (.first)          */
  double* depth;              /* keypoints_depth_                             */
  float* right_xy;            /* right_frame_.keypoints_ (distorted px)       */
  double* keypoints_3d;       /* keypoints_3d_, n x 3                         */
  uint8_t* left_rect_img;     /* optional W*H rectified left image            */
  uint8_t* right_rect_img;    /* optional W*H rectified right image           */
} kvfe_stereo_output;

KVFE_API kvfe_status kvfe_sparse_stereo_reconstruction(
    kvfe_ctx* ctx, const uint8_t* left_img, const uint8_t* right_img, size_t stride,
    const float* left_xy, int32_t n, kvfe_stereo_output* out);

/* Tracker::geometricOutlierRejection2d2d(ref_bearings, cur_bearings, matches,
 * inliers, cam_lkf_Pose_cam_kf) with ransac_use_2point_mono (Tracker.cpp:213-318):
 * opengv RANSAC over TranslationOnlySacProblem with the rotation given.
 * f_ref / f_cur: n matched bearing vectors (n x 3 float64).  Outputs: tracking
 * status (VALID / FEW_MATCHES / INVALID), pose 3x4 [R | t], inlier indices into
 * the match list (ascending, capacity n), RANSAC iterations. */
typedef struct kvfe_ransac_output {
  int32_t status;            /* KVFE_TRACKING_*                              */
  int32_t n_inliers;
  int32_t iterations;
  int32_t reserved0;
  double pose[12];           /* row-major 3x4 [R | t]                        */
  double info[9];            /* stereo only: information matrix of t         */
} kvfe_ransac_output;

KVFE_API kvfe_status kvfe_outlier_rejection_2d2d_given_rotation(
    kvfe_ctx* ctx, const double* f_ref, const double* f_cur, int32_t n,
    const double R_ref_cur[9], int32_t* inliers, kvfe_ransac_output* out);

/* Tracker::geometricOutlierRejection2d2d without a rotation prior (Tracker.cpp:262-275): opengv
 * CentralRelativePoseSacProblem, NISTER (5 + 3 points per sample), the problem of
 * ransac_use_2point_mono = 0 (params/D455).  pose = lkf_T_k with a translation of arbitrary norm
 * (the norm follows the scale of the essential matrix: it carries no information upstream either).
 * The polynomial solver is not OpenGV's: poses agree with the reference to rounding, inlier
 * decisions are pinned by the scenes of tests/testTracker.cpp:704-802. */
KVFE_API kvfe_status kvfe_outlier_rejection_2d2d(kvfe_ctx* ctx, const double* f_ref,
                                                 const double* f_cur, int32_t n,
                                                 int32_t* inliers, kvfe_ransac_output* out);

/* Tracker::geometricOutlierRejection3d3dGivenRotation (Tracker.cpp:382-632) +
 * Tracker::getPoint3AndCovariance (:772-818): the 1-point voting scheme on n
 * stereo matches.  Per match and per frame (ref = last keyframe, cur): rectified
 * left pixel (x, y), rectified right pixel x, 3-D point (keypoints_3d_). */
KVFE_API kvfe_status kvfe_outlier_rejection_3d3d_given_rotation(
    kvfe_ctx* ctx, const float* ref_left_rect_xy, const float* ref_right_rect_x,
    const double* ref_points_3d, const float* cur_left_rect_xy,
    const float* cur_right_rect_x, const double* cur_points_3d, int32_t n,
    const double R_ref_cur[9], int32_t* inliers, kvfe_ransac_output* out);

/* Tracker::geometricOutlierRejection3d3d(ref_keypoints_3d, cur_keypoints_3d, matches, inliers)
 * (Tracker.cpp:667-742): opengv 3-point Arun RANSAC (PointCloudSacProblem, fixed seed) over n
 * matched 3-D points (n x 3 float64 each, already gathered by match); the front-end takes this
 * branch when ransac_use_1point_stereo is off or the keyframe has no gyro rotation
 * (VisionImuFrontend.cpp:127-142).  pose = lkf_T_k (points_ref = R points_cur + t); info is zero.
 * The 3x3 SVD is a Jacobi iteration, not Eigen's: poses agree with the reference to rounding,
 * inlier decisions are pinned by the scenes of tests/testTracker.cpp:1004-1186. */
KVFE_API kvfe_status kvfe_outlier_rejection_3d3d(kvfe_ctx* ctx, const double* ref_points_3d,
                                                 const double* cur_points_3d, int32_t n,
                                                 int32_t* inliers, kvfe_ransac_output* out);

/* Tracker::pnp(cam_bearing_vectors, F_points, F_Pose_cam_estimate, inliers) (Tracker.cpp:1122-1288) +
 * VisionImuFrontend::outlierRejectionPnP (VisionImuFrontend.cpp:146-173): 2D-3D RANSAC of the camera pose in the
 * frame F of the landmarks.  Tracker::pnp(const StereoFrame&, ...) (Tracker.cpp:1064-1120) gathers the
 * correspondences on the host -- keypoints with a VALID rectified left keypoint whose landmark id is in the map of
 * optimised landmarks handed over by Tracker::updateMap -- and calls this; the C++ adapter does the same
 * (kvfe::Tracker::pnp / updateMap in include/kvfe_adapter.hpp).  With use_pnp_tracking the step runs the same stage on
 * the device against the map of kvfe_frontend_update_map and reports it in kvfe_frame_output; its result
 * (kfTracking_status_pnp_, W_T_k_pnp_) does not feed back into the keypoint state (VisionImuFrontend.cpp:163
 * "TODO remove outliers").
 * Implemented: pnp_algorithm 3 (EPNP: opengv AbsolutePoseSacProblem, 6 points per sample, fixed seed), the value
 * of every shipped parameter set but params/KinectAzure, and that one's pnp_algorithm 1 (KneipP3P: 3 + 1 points per
 * sample, closed-form quartic); the others return KVFE_ERR_UNSUPPORTED, and so does
 * optimize_2d3d_pose_from_inliers (off everywhere).  The threshold is 1 - cos(atan(sqrt 2 ransac_threshold_pnp /
 * f)) with f the mean of the LEFT camera's fx, fy; iterations / probability / sampler come from the tracker
 * parameters of the context.
 * out: status = VALID when Tracker::pnp succeeded with more than min_pnp_inliers inliers, else FEW_MATCHES;
 * reserved0 = Tracker::pnp's return value; pose = F_Pose_cam 3x4 [R | t]; inliers ascending (capacity n).
 * The dense linear algebra of EPnP is a Jacobi / Householder implementation, not Eigen's: poses agree with the
 * reference to rounding; the inlier decision is pinned by the scene of tests/testTracker.cpp:1613-1800. */
KVFE_API kvfe_status kvfe_pnp(kvfe_ctx* ctx, const kvfe_pnp_params* params, const double* cam_bearing_vectors,
                              const double* F_points, int32_t n, int32_t* inliers, kvfe_ransac_output* out);

/* ---- UndistorterRectifier / StereoCamera / StereoMatcher keypoint methods on their own -------- */
/* (the front-end step runs them fused; these are the reference's public methods as component calls) */

/* UndistorterRectifier::checkUndistortedRectifiedLeftKeypoints(distorted_kps, undistorted_kps, status_kps,
 * pixel_tol = 2.0f) (UndistorterRectifier.h:96-100, UndistorterRectifier.cpp:138-211) of camera `cam`'s
 * rectifier: crop the rectified keypoint to the image, look the map up at the rounded position, VALID if
 * it lands within pixel_tol of the distorted keypoint, else NO_LEFT_RECT; out_xy = the cropped keypoint. */
KVFE_API kvfe_status kvfe_check_undistorted_rectified_left_keypoints(
    kvfe_ctx* ctx, int32_t cam, const float* distorted_xy, const float* undistorted_xy, int32_t n,
    float pixel_tol, float* out_xy, uint8_t* out_status);

/* UndistorterRectifier::distortUnrectifyKeypoints (UndistorterRectifier.h:108-110, .cpp:213-228) of camera
 * `cam`: (map_x, map_y) at the rounded rectified pixel for VALID keypoints, (0, 0) otherwise.  A VALID keypoint
 * whose rounded position lies outside the image is KVFE_ERR_INVALID_ARG (cv::Mat::at out of range upstream). */
KVFE_API kvfe_status kvfe_distort_unrectify_keypoints(kvfe_ctx* ctx, int32_t cam, const float* rect_xy,
                                                      const uint8_t* status, int32_t n, float* out_xy);

/* StereoCamera::undistortRectifyLeftKeypoints (StereoCamera.h:203-213, StereoCamera.cpp:236-260):
 * undistortRectifyKeypoints (K1, D1, R1, P1) + checkUndistortedRectifiedLeftKeypoints. */
KVFE_API kvfe_status kvfe_undistort_rectify_left_keypoints(kvfe_ctx* ctx, const float* xy, int32_t n,
                                                           float* out_xy, uint8_t* out_status);

/* StereoCamera::distortUnrectifyRightKeypoints (StereoCamera.h:215-223, StereoCamera.cpp:262-267). */
KVFE_API kvfe_status kvfe_distort_unrectify_right_keypoints(kvfe_ctx* ctx, const float* rect_xy,
                                                            const uint8_t* status, int32_t n, float* out_xy);

/* StereoCamera::undistortRectifyStereoFrame (StereoCamera.h:225-233, StereoCamera.cpp:269-290): both images
 * through their rectifiers (StereoFrame::setRectifiedImages receives left_rect / right_rect). */
KVFE_API kvfe_status kvfe_undistort_rectify_stereo_frame(kvfe_ctx* ctx, const uint8_t* left, const uint8_t* right,
                                                         size_t src_stride, uint8_t* left_rect,
                                                         uint8_t* right_rect, size_t dst_stride);

/* StereoMatcher::getDepthFromRectifiedMatches(left_keypoints_rectified, right_keypoints_rectified,
 * keypoints_depth) (StereoMatcher.h:85-92, StereoMatcher.cpp:425-483): depth = fx b / (uL - uR) for VALID pairs;
 * right_status is updated in place as upstream (NO_DEPTH for a negative disparity or a depth outside
 * [minPointDist, maxPointDist]; the left status where the left keypoint is invalid), depth 0 where invalid. */
KVFE_API kvfe_status kvfe_get_depth_from_rectified_matches(kvfe_ctx* ctx, const float* left_rect_xy,
                                                           const uint8_t* left_status, const float* right_rect_xy,
                                                           uint8_t* right_status, int32_t n, double* depth);

/* VIO::Frame as far as the hot path touches it (include/kimera-vio/frontend/Frame.h:160-186): parallel arrays of
 * n_keypoints entries in caller-owned storage of `capacity` entries (scores_ is not carried: the reference
 * never sets it, FeatureDetector.cpp:147 "NOT IMPLEMENTED"). */
typedef struct kvfe_frame {
  int32_t capacity;
  int32_t n_keypoints;
  float* keypoints;            /* keypoints_      n x 2 */
  int64_t* landmarks;          /* landmarks_      -1 = no landmark */
  int32_t* landmarks_age;      /* landmarks_age_ */
  double* versors;             /* versors_        n x 3 (input may be NULL for a reference frame) */
} kvfe_frame;

/* void FeatureDetector::featureDetection(Frame* cur_frame, std::optional<cv::Mat> R)
 * (FeatureDetector.h:39-41, FeatureDetector.cpp:94-163): ages of all entries + 1, need = maxFeaturesPerFrame -
 * #landmarks != -1, masked GFTT + ANMS + cornerSubPix, then the new corners are appended with landmark ids
 * *landmark_counter, +1, ... (the function-static `lmk_id` of FeatureDetector.cpp:141, owned by the caller
 * here), age 1 and their bearing vectors (GetBearingVector with the context's R: R1 for a stereo context,
 * none for a mono one).  KVFE_ERR_CAPACITY if the frame does not hold the result. */
KVFE_API kvfe_status kvfe_feature_detection_frame(kvfe_ctx* ctx, const uint8_t* img, size_t stride,
                                                  kvfe_frame* frame, int64_t* landmark_counter);

/* void Tracker::featureTracking(Frame* ref_frame, Frame* cur_frame, const gtsam::Rot3& ref_R_cur,
 * const FeatureDetectorParams&, std::optional<cv::Mat> R) (Tracker.h:70-74, Tracker.cpp:92-211): the reference
 * keypoints with a landmark are predicted (OpticalFlowPredictor) and tracked (calcOpticalFlowPyrLK); survivors
 * not older than maxFeatureAge are pushed to cur_frame (ids, ages, keypoints, bearing vectors), the others
 * get ref_frame->landmarks[i] = -1.  cur_frame must be empty on entry (the reference CHECKs it). */
KVFE_API kvfe_status kvfe_feature_tracking_frame(kvfe_ctx* ctx, const uint8_t* ref_img, const uint8_t* cur_img,
                                                 size_t stride, kvfe_frame* ref_frame, kvfe_frame* cur_frame,
                                                 const double ref_R_cur[9]);

/* ------------------------------------------------------------------------- */
/* front-end level: `batch` independent streams, lock-step, device resident  */
/* ------------------------------------------------------------------------- */

/* per-stream, per-frame input: what StereoVisionImuFrontend::nominalSpinStereo
 * hands to processStereoFrame (StereoVisionImuFrontend.cpp:102-240). */
typedef struct kvfe_frame_input {
  int64_t timestamp_ns;
  double keyframe_R_cur_frame[9];   /* camLrectLkf_R_camLrectK_imu, row-major */
  int32_t force_keyframe;           /* Frame::isKeyframe_ set by the user     */
  int32_t reserved0;
} kvfe_frame_input;

/* StereoVisionImuFrontend::processFirstStereoFrame / processStereoFrame for
 * all streams of the context (StereoVisionImuFrontend.cpp:245-481), including
 * the geometric outlier rejection of keyframes when useRANSAC = 1
 * (VisionImuFrontend.cpp:90-144).  left/right: `batch` images back to back (image s at
 * base + s*image_stride_bytes).  The *_host variant copies from host memory;
 * the *_device variant takes device pointers that are read by THIS step only (they
 * must stay valid until it has completed -- kvfe_synchronize / kvfe_frontend_get_output --,
 * as for any asynchronous call; the left image the next step's tracking needs is copied
 * into the context).  Both only enqueue work.
 * RGBD front-end (RgbdVisionImuFrontend::processFirstFrame / processFrame,
 * RgbdVisionImuFrontend.cpp:184-368): `right` holds the depth images (uint16 or float per
 * kvfe_depth_params::depth_type) with the same geometry in ELEMENTS: row s starts at
 * base + (s*image_stride_bytes + y*row_stride_bytes) * sizeof(depth element). */
KVFE_API kvfe_status kvfe_frontend_step_host(kvfe_ctx* ctx, const uint8_t* left,
                                             const uint8_t* right, size_t row_stride,
                                             size_t image_stride,
                                             const kvfe_frame_input* inputs);
KVFE_API kvfe_status kvfe_frontend_step_device(kvfe_ctx* ctx, const void* left_dev,
                                               const void* right_dev, size_t row_stride,
                                               size_t image_stride,
                                               const kvfe_frame_input* inputs);
/* Staged input (the hand-off the data provider feeds, EurocDataProvider.cpp:146-195
 * / StereoDataProviderModule.cpp:35-91).  The context owns KVFE_STAGING_SLOTS pinned host slots of
 * `batch` left + `batch` right images (tightly packed, width*height bytes each); the data provider
 * decodes straight into a slot and calls kvfe_frontend_step_staged, which uploads the slot on a copy
 * stream (overlapping the previous step's kernels) and enqueues the step.  A slot may be refilled
 * once kvfe_frontend_staging_wait(slot) returns (its upload has completed).  A slot's pinned memory is allocated
 * the first time kvfe_frontend_staging_buffer / kvfe_frontend_step_staged names it (2 * batch * width * height bytes);
 * three slots are what a decoder thread ahead of the front-end needs, eight let a benchmark keep a ring of frames
 * resident on the host. */
#define KVFE_STAGING_SLOTS 8
KVFE_API kvfe_status kvfe_frontend_staging_buffer(kvfe_ctx* ctx, int32_t slot, uint8_t** left,
                                                  uint8_t** right);
KVFE_API kvfe_status kvfe_frontend_staging_wait(kvfe_ctx* ctx, int32_t slot);
KVFE_API kvfe_status kvfe_frontend_step_staged(kvfe_ctx* ctx, int32_t slot,
                                               const kvfe_frame_input* inputs);
KVFE_API kvfe_status kvfe_frontend_reset(kvfe_ctx* ctx);
KVFE_API kvfe_status kvfe_synchronize(kvfe_ctx* ctx);

/* StereoFrontendOutput (StereoVisionImuFrontend-definitions.h:25-91) reduced to
 * the arrays the hot path produces.  Caller allocates `capacity` entries per
 * array (NULL arrays are skipped).
 * Output side (round 4): the step itself packs every stream's output into one contiguous record and sends the
 * records of the step to a pinned host ring slot in ONE device-initiated transfer (off the critical path: beside the
 * next step's tracking launch); kvfe_frontend_get_output waits for that transfer of the LATEST step -- which implies
 * that every kernel of the step has completed -- and copies out of pinned memory.  No blocking device copies.
 * Contract of the accessors (kvfe_frontend_get_output / _at / s, kvfe_frontend_view_output):
 *   - they do NOT synchronise the context: they wait for the records of the step they name and nothing else (use
 *     kvfe_synchronize for "everything enqueued has completed"); a record set that outgrew its transfer is completed
 *     by a copy on a stream of its own, never behind the running step;
 *   - before the first step, and after kvfe_frontend_reset, there is no record: KVFE_ERR_INVALID_ARG (until round 3
 *     this returned KVFE_OK with zero counts -- callers that polled before stepping must check the status);
 *   - they belong to the thread that calls kvfe_frontend_step_*: a context is single-threaded (section "Threading"
 *     of INTEGRATION.md); a consumer thread reading step k while the producer enqueues step k+1 needs the caller's
 *     own mutex around both -- the record data it then reads is stable for KVFE_OUTPUT_RING - 1 further steps. */
typedef struct kvfe_frame_output {
  int32_t capacity;
  int32_t n_keypoints;          /* left_frame_.keypoints_.size()             */
  int32_t is_keyframe;
  int32_t n_tracked;            /* keypoints carried over by featureTracking */
  int32_t n_detected;           /* keypoints appended by featureDetection    */
  int32_t n_measurements;       /* getSmartStereoMeasurements entries        */
  int64_t frame_id;
  int64_t* landmarks;           /* left_frame_.landmarks_                    */
  int32_t* landmarks_age;
  float* keypoints;             /* left_frame_.keypoints_                    */
  double* versors;              /* left_frame_.versors_, n x 3               */
  float* left_rect_xy;
  uint8_t* left_status;
  float* right_rect_xy;
  uint8_t* right_status;
  double* depth;
  float* right_xy;
  double* keypoints_3d;
  int64_t* meas_landmark;       /* StereoMeasurement.first                   */
  double* meas_uL_uR_v;         /* StereoPoint2 (uL, uR or NaN, v), n x 3    */
  /* TrackerStatusSummary (Tracker-definitions.h:135-183) of this frame, valid on
   * keyframes; poses are row-major 3x4 [R | t] of lkf_T_k                     */
  int32_t tracking_status_mono;    /* kfTrackingStatus_mono_,   KVFE_TRACKING_* */
  int32_t tracking_status_stereo;  /* kfTrackingStatus_stereo_                  */
  double lkf_T_k_mono[12];
  double lkf_T_k_stereo[12];
  double info_mat_stereo_translation[9];   /* infoMatStereoTranslation_         */
  /* DebugTrackerInfo (Tracker-definitions.h:78-124) */
  int32_t nr_mono_putatives, nr_mono_inliers, mono_ransac_iters;
  int32_t nr_stereo_putatives, nr_stereo_inliers, reserved0;
  /* use_pnp_tracking: kfTracking_status_pnp_ / W_T_k_pnp_ of the last keyframe (INVALID / identity when off) */
  int32_t tracking_status_pnp;
  int32_t nr_pnp_inliers;
  double W_T_k_pnp[12];
} kvfe_frame_output;

/* Tracker::updateMap (Tracker.h:82-94): the back-end's map of optimised landmarks (id -> world position) of one
 * stream, replacing the previous one; copied into the context (ids need not be sorted; n <= landmark_map_capacity).
 * With use_pnp_tracking the step runs Tracker::pnp(stereoFrame_k) on keyframes right after the stereo outlier
 * rejection (same place as upstream), on the device. */
KVFE_API kvfe_status kvfe_frontend_update_map(kvfe_ctx* ctx, int32_t stream, const int64_t* landmark_ids,
                                              const double* xyz, int32_t n);

KVFE_API kvfe_status kvfe_frontend_get_output(kvfe_ctx* ctx, int32_t stream,
                                              kvfe_frame_output* out);
/* The same for the step `steps_back` steps before the latest one (0 = kvfe_frontend_get_output; at most
 * KVFE_OUTPUT_RING - 1): waits for THAT step's record only, so a consumer reads frame k while frame k+1 is being
 * processed -- how upstream hands StereoFrontendOutput to the back-end queue while the front-end thread goes on
 * (PipelineModule.h:324-343). */
#define KVFE_OUTPUT_RING 3
KVFE_API kvfe_status kvfe_frontend_get_output_at(kvfe_ctx* ctx, int32_t stream, int32_t steps_back,
                                                 kvfe_frame_output* out);
/* all `batch` streams in one call: outs[s] as above (capacity and array pointers set by the caller per stream) */
KVFE_API kvfe_status kvfe_frontend_get_outputs(kvfe_ctx* ctx, int32_t steps_back, kvfe_frame_output* outs);
/* zero-copy: the scalar fields are filled and the array pointers of `out` are pointed INTO the pinned record
 * (capacity = entries held; stereo arrays NULL on a frame without stereo data).  Valid until
 * KVFE_OUTPUT_RING - 1 - steps_back further steps have been enqueued; read-only for the caller. */
KVFE_API kvfe_status kvfe_frontend_view_output(kvfe_ctx* ctx, int32_t stream, int32_t steps_back,
                                               kvfe_frame_output* out);

/* ------------------------------------------------------------------------- */
/* Input side: what sits between the dataset / sensor and    */
/* kvfe_frontend_step_*.  Host code (no device work): the data provider runs */
/* on CPU threads next to the GPU steps, as upstream's does.                  */
/* ------------------------------------------------------------------------- */

/* UtilsOpenCV::ReadAndConvertToGrayScale without the equalisation (src/utils/UtilsOpenCV.cpp:390-399; the
 * equalisation is kvfe_equalize_hist / stereo_matching.equalize_image): cv::imread(IMREAD_ANYCOLOR) of PNG data
 * (8-bit result: 16-bit samples keep their high byte, 1/2/4-bit grey is expanded, alpha dropped, palettes
 * resolved, Adam7 supported) followed by cv::cvtColor(BGR2GRAY) when the file has colour (15-bit BT.601 weights
 * 9798 / 19235 / 3735 of OpenCV 4).  EuRoC frames are 8-bit grey PNGs (EurocDataProvider.cpp:172-188).
 * kvfe_png_info reads the header only.  Errors: KVFE_ERR_INVALID_ARG (not a PNG / corrupt / CRC mismatch /
 * size mismatch), KVFE_ERR_UNSUPPORTED (a PNG feature outside the list above). */
KVFE_API kvfe_status kvfe_png_info(const uint8_t* data, size_t size, int32_t* width, int32_t* height,
                                   int32_t* channels);
KVFE_API kvfe_status kvfe_png_decode_gray(const uint8_t* data, size_t size, uint8_t* dst, size_t dst_stride,
                                          int32_t width, int32_t height);
/* n files by `threads` host threads (<= 0: one per file, at most the processors this process may use -- hardware
 * threads cut down to its affinity mask and its cgroup CPU quota) straight into the
 * caller's buffers -- typically the pinned staging slot of kvfe_frontend_staging_buffer.  status[i] per file
 * (may be NULL); returns the first failure. */
KVFE_API kvfe_status kvfe_png_decode_gray_batch(const uint8_t* const* data, const size_t* sizes,
                                                uint8_t* const* dst, size_t dst_stride, int32_t width,
                                                int32_t height, int32_t n, int32_t threads,
                                                kvfe_status* status);

/* The same for JPEG files (the reference's own front-end test frames, tests/data/ForStereoTracker/ *.jpg, are baseline
 * 4:2:0 JFIF): libjpeg's default decoding pipeline restated in integer arithmetic -- Huffman decoding, the
 * slow-accurate integer IDCT, fancy chroma upsampling, the fixed-point YCbCr -> RGB tables -- then BGR2GRAY for
 * three-component files; bit-identical to libjpeg / libjpeg-turbo.  8-bit baseline / extended-sequential Huffman, 1
 * or 3 components, luma sampling 1x1, 2x1, 2x2, restart intervals; anything else (progressive, arithmetic, CMYK, 12
 * bit) is KVFE_ERR_UNSUPPORTED.  EXIF orientation is not applied.  channels = components of the file. */
KVFE_API kvfe_status kvfe_jpeg_info(const uint8_t* data, size_t size, int32_t* width, int32_t* height,
                                    int32_t* channels);
KVFE_API kvfe_status kvfe_jpeg_decode_gray(const uint8_t* data, size_t size, uint8_t* dst, size_t dst_stride,
                                           int32_t width, int32_t height);

/* utils::ThreadsafeImuBuffer (include/kimera-vio/utils/ThreadsafeImuBuffer.h:46-196, src/utils/
 * ThreadsafeImuBuffer.cpp:48-234 over ThreadsafeTemporalBuffer-inl.h): a time-ordered map of (acc, gyro)
 * samples, thread safe.  Query results keep the upstream enum values. */
enum {
  KVFE_IMU_DATA_AVAILABLE = 0,            /* kDataAvailable                */
  KVFE_IMU_DATA_NOT_YET_AVAILABLE = 1,    /* kDataNotYetAvailable          */
  KVFE_IMU_DATA_NEVER_AVAILABLE = 2,      /* kDataNeverAvailable           */
  KVFE_IMU_QUEUE_SHUTDOWN = 3,            /* kQueueShutdown                */
  KVFE_IMU_TOO_FEW_MEASUREMENTS = 4       /* kTooFewMeasurementsAvailable  */
};
typedef struct kvfe_imu_buffer kvfe_imu_buffer;
/* buffer_length_ns <= 0: unbounded (ThreadsafeImuBuffer(-1)) */
KVFE_API kvfe_imu_buffer* kvfe_imu_buffer_create(int64_t buffer_length_ns);
KVFE_API void kvfe_imu_buffer_destroy(kvfe_imu_buffer* b);
/* addMeasurement: acc_gyr = (ax ay az wx wy wz), "Acceleration first!" (EurocDataProvider.cpp:265-267) */
KVFE_API void kvfe_imu_buffer_add(kvfe_imu_buffer* b, int64_t timestamp_ns, const double acc_gyr[6]);
KVFE_API int64_t kvfe_imu_buffer_size(const kvfe_imu_buffer* b);
KVFE_API void kvfe_imu_buffer_shutdown(kvfe_imu_buffer* b);
/* getImuDataBtwTimestamps / getImuDataInterpolatedUpperBorder / getImuDataInterpolatedBorders.
 * stamps[capacity], acc_gyr[6 * capacity] (column k = sample k, as ImuAccGyrS); *n = samples written
 * (0 unless the result is KVFE_IMU_DATA_AVAILABLE).  Returns the query result, or -1 when `capacity` is
 * too small (*n = the count needed).  t_from >= t_to (a CHECK failure upstream) is KVFE_IMU_DATA_NEVER_AVAILABLE. */
KVFE_API int32_t kvfe_imu_buffer_between(kvfe_imu_buffer* b, int64_t t_from, int64_t t_to,
                                         int32_t get_lower_bound, int64_t* stamps, double* acc_gyr,
                                         int32_t capacity, int32_t* n);
KVFE_API int32_t kvfe_imu_buffer_interpolated_upper_border(kvfe_imu_buffer* b, int64_t t_from, int64_t t_to,
                                                           int64_t* stamps, double* acc_gyr,
                                                           int32_t capacity, int32_t* n);
KVFE_API int32_t kvfe_imu_buffer_interpolated_borders(kvfe_imu_buffer* b, int64_t t_from, int64_t t_to,
                                                      int64_t* stamps, double* acc_gyr, int32_t capacity,
                                                      int32_t* n);
/* ThreadsafeImuBuffer::linearInterpolate */
KVFE_API void kvfe_imu_linear_interpolate(int64_t t0, const double y0[6], int64_t t1, const double y1[6],
                                          int64_t t, double y[6]);

/* The rotation the front-end takes as kvfe_frame_input::keyframe_R_cur_frame, from the gyro samples of the packets:
 * ImuFrontend::preintegrateImuMeasurements (src/imu-frontend/ImuFrontend.cpp:158-173) as far as the rotation goes --
 * for i < n - 1: dt = (stamps[i+1] - stamps[i]) / 1e9, deltaRij <- deltaRij . Expmap((gyro_i - gyro_bias) dt), i.e.
 * gtsam's on-manifold preintegration (GTSAM_TANGENT_PREINTEGRATION=OFF, Dockerfile_20_04:49) with
 * so3::ExpmapFunctor (I + W for |w|^2 <= eps, else I + sin(t) K + 2 sin^2(t/2) K^2, K = W / t); the last sample only
 * contributes its time stamp, as upstream.  deltaRij (row-major 3x3) is updated in place: start it at identity at a
 * keyframe (ImuFrontend::resetIntegrationWithCachedBias), feed every packet since.  KVFE_ERR_INVALID_ARG for n < 2
 * or a non-positive dt (both CHECKs upstream). */
KVFE_API kvfe_status kvfe_imu_preintegrate_rotation(const int64_t* stamps, const double* acc_gyr, int32_t n,
                                                    const double gyro_bias[3], double deltaRij[9]);
/* camLrectLkf_R_camLrectK_imu = body_R_camLrect^-1 . deltaRij . body_R_camLrect (StereoVisionImuFrontend.cpp:143-150;
 * body_R_camLrect = rotation of StereoCamera::getBodyPoseLeftCamRect = R_BS . R1^T) */
KVFE_API void kvfe_keyframe_R_cur_frame(const double body_R_camLrect[9], const double deltaRij[9], double out[9]);

/* StereoDataProviderModule::getInputPacket (src/dataprovider/StereoDataProviderModule.cpp:35-91) over
 * MonoDataProviderModule::getMonoImuSyncPacket (MonoDataProviderModule.cpp:44-118), DataProviderModule::
 * getTimeSyncedImuMeasurements (DataProviderModule.cpp:80-181) and SimpleQueueSynchronizer::syncQueue
 * (pipeline/QueueSynchronizer.h:79-162): left / right frame queues and the IMU buffer in, one
 * StereoImuSyncPacket out.  Frames are (timestamp, caller tag): the images stay with the caller (e.g. in a
 * staging slot), the packet says which left / right tags belong together and carries the IMU samples between
 * the previous packet's frame and this one, borders interpolated.  The decision logic is upstream's; where
 * upstream blocks on a queue (parallel_run) this returns KVFE_SYNC_EMPTY / KVFE_SYNC_WAIT_* and the caller
 * calls again after pushing more data. */
enum {
  KVFE_SYNC_PACKET = 0,               /* a packet was written                                              */
  KVFE_SYNC_EMPTY = 1,                /* no left frame queued                                              */
  KVFE_SYNC_WAIT_IMU = 2,             /* FrameAction::Wait: IMU data up to the frame not there yet; the left
                                         frame is cached and retried by the next call                      */
  KVFE_SYNC_DROP_OUT_OF_ORDER = 3,    /* left timestamp <= the last packet's                               */
  KVFE_SYNC_DROP_NO_IMU = 4,          /* IMU buffer empty                                                  */
  KVFE_SYNC_DROP_FIRST_FRAME = 5,     /* "Skipping first frame": it only sets the previous-frame timestamp */
  KVFE_SYNC_DROP_IMU_NEVER = 6,       /* kDataNeverAvailable (the frame becomes the previous frame)        */
  KVFE_SYNC_DROP_IMU_TOO_FEW = 7,     /* kTooFewMeasurementsAvailable                                      */
  KVFE_SYNC_DROP_NO_RIGHT = 8,        /* "Missing right frame for left frame": right queue empty or ahead  */
  KVFE_SYNC_SHUTDOWN = 9
};
typedef struct kvfe_stereo_sync kvfe_stereo_sync;
typedef struct kvfe_sync_packet {
  int64_t timestamp_ns;
  int64_t left_tag, right_tag;
  int32_t n_imu;                      /* samples written to the caller's arrays */
  int32_t reserved0;
} kvfe_sync_packet;
KVFE_API kvfe_stereo_sync* kvfe_stereo_sync_create(int64_t imu_buffer_length_ns);
/* The same module for the other two providers: KVFE_SYNC_MODE_MONO = MonoDataProviderModule::getInputPacket
 * (MonoDataProviderModule.cpp:30-42: no second queue, right_tag = -1); KVFE_SYNC_MODE_RGBD = RgbdDataProviderModule::
 * getInputPacket (RgbdDataProviderModule.cpp:44-84: the "right" queue holds the depth frames, and the previous-frame
 * timestamp advances BEFORE the depth frame is looked up -- getMonoImuSyncPacket(cache_timestamp = true) --, so a frame
 * dropped for a missing depth image still ends the IMU interval).  Default KVFE_SYNC_MODE_STEREO. */
enum { KVFE_SYNC_MODE_STEREO = 0, KVFE_SYNC_MODE_MONO = 1, KVFE_SYNC_MODE_RGBD = 2 };
KVFE_API kvfe_status kvfe_stereo_sync_set_mode(kvfe_stereo_sync* s, int32_t mode);
KVFE_API void kvfe_stereo_sync_destroy(kvfe_stereo_sync* s);
KVFE_API void kvfe_stereo_sync_fill_left(kvfe_stereo_sync* s, int64_t timestamp_ns, int64_t tag);
KVFE_API void kvfe_stereo_sync_fill_right(kvfe_stereo_sync* s, int64_t timestamp_ns, int64_t tag);
KVFE_API void kvfe_stereo_sync_fill_imu(kvfe_stereo_sync* s, int64_t timestamp_ns, const double acc_gyr[6]);
/* DataProviderModule::doCoarseImuCameraTemporalSync / setImuTimeShift (DataProviderModule.h:106-119) */
KVFE_API void kvfe_stereo_sync_do_coarse_imu_camera_temporal_sync(kvfe_stereo_sync* s);
KVFE_API void kvfe_stereo_sync_set_imu_time_shift(kvfe_stereo_sync* s, double imu_time_shift_s);
KVFE_API void kvfe_stereo_sync_shutdown(kvfe_stereo_sync* s);
/* one getInputPacket(): returns KVFE_SYNC_*; imu_stamps[capacity], imu_acc_gyr[6 * capacity] as above
 * (-1 when capacity is too small: nothing is consumed, packet->n_imu = the count needed) */
KVFE_API int32_t kvfe_stereo_sync_next(kvfe_stereo_sync* s, kvfe_sync_packet* packet, int64_t* imu_stamps,
                                       double* imu_acc_gyr, int32_t capacity);

/* EuRoC index files (CameraImageLists::parseCamImgList, DataProviderInterface-definitions.cpp:74-101;
 * EurocDataProvider::parseImuData, EurocDataProvider.cpp:229-306): the text of mav0/camN/data.csv gives the
 * frame timestamps (the image is <folder>/data/<timestamp>.png), the text of mav0/imu0/data.csv the IMU
 * samples, re-ordered acceleration first.  Both skip the header line; *n = rows found; returns
 * KVFE_ERR_CAPACITY when the arrays are too small (then *n = rows needed), KVFE_ERR_INVALID_ARG on a
 * malformed row or IMU timestamps that are not increasing. */
KVFE_API kvfe_status kvfe_euroc_parse_camera_csv(const char* text, size_t size, int64_t* timestamps,
                                                 int32_t capacity, int32_t* n);
KVFE_API kvfe_status kvfe_euroc_parse_imu_csv(const char* text, size_t size, int64_t* timestamps,
                                              double* acc_gyr, int32_t capacity, int32_t* n);

/* ------------------------------------------------------------------------- */
/* measurement hooks (bench.py): HIP-event timing on the context's stream    */
/* ------------------------------------------------------------------------- */
#define KVFE_N_STAGES 16
typedef struct kvfe_stage_times {
  int32_t n_stages;
  int32_t n_samples;                   /* launches recorded per stage (all groups)  */
  int32_t n_groups;                    /* stream groups (launches per step)         */
  int32_t struct_size;                 /* IN: sizeof(kvfe_stage_times) of the caller's header; kvfe_profile_read writes no
                                          more than that (0 = the layout that ended before ms_active, round 2).  The
                                          struct grew in round 3; a caller built against the shorter layout is safe     */
  const char* name[KVFE_N_STAGES];
  double ms_total[KVFE_N_STAGES];      /* summed over samples                 */
  double alg_bytes[KVFE_N_STAGES];     /* algorithmic bytes per launch with every stream active */
  /* The keyframe-only stages (detection, rectification, stereo matching, outlier rejection) are launched on
   * every step but work only for the streams whose flags of that step say so; the flags of every sampled
   * step are read back.  A stage's time and bytes per launch that DID work:
   *   ms_active / active_launches   and   alg_bytes_per_stream * active_streams / active_launches          */
  double ms_active[KVFE_N_STAGES];            /* summed over the sampled launches with >= 1 active stream */
  double active_streams[KVFE_N_STAGES];       /* active streams summed over those launches               */
  double alg_bytes_per_stream[KVFE_N_STAGES]; /* algorithmic bytes of one active stream                  */
  int32_t active_launches[KVFE_N_STAGES];     /* sampled launches with >= 1 active stream                */
} kvfe_stage_times;
/* on = 0: off; on = N > 0: every N-th step records a begin/end HIP event pair per stage on the
 * stream the stage runs on (recording every step costs ~10 % at 64 streams: 24 extra stream ops) */
KVFE_API kvfe_status kvfe_profile_enable(kvfe_ctx* ctx, int32_t on);
KVFE_API kvfe_status kvfe_profile_read(kvfe_ctx* ctx, kvfe_stage_times* out);
/* what a plain streaming copy (one 16-byte load + one 16-byte store per lane and trip) reaches on the current device, in
 * this process: `iters` copies of `bytes` bytes between two freshly allocated buffers, HIP events around them.
 * *read_plus_write_GBps = 2 * bytes * iters / time.  bench.py prints it beside the roofline fractions (which stay
 * quoted against the guide's 8 TB/s) so that a slow box can be told from a regression.  No reference counterpart.      */
KVFE_API kvfe_status kvfe_hbm_copy_probe(size_t bytes, int32_t iters, double* read_plus_write_GBps, double* ms_per_copy);

#ifdef __cplusplus
}
#endif
#endif /* KVFE_H_ */
