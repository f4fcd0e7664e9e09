// Device-side data layout and kernel launch interface of libkvfe (gfx950 only).
//
// Layout in HBM (one context = `B` independent stereo streams, all arrays are
// struct-of-arrays with the stream index outermost so that one launch covers
// every stream with blockIdx.z / blockIdx.y = stream):
//   images        u8   [B][H][W]            raw left/right (caller or ctx owned), rectified L/R
//   pyramids      u8   [2][B][sum_l w_l*h_l] levels 1..L of the current / previous left image
//   maps          f32x2[2][H][W]            undistort-rectify maps (shared by all streams)
//   candidates    u64  [B][ccap]            (min-eig bits << 32 | pixel index) local maxima
//   frame tables  SoA  [3][B][kcap]         keypoints / landmark ids / ages / versors
//   stereo tables SoA  [B][kcap]            rectified keypoints, statuses, depth, 3D points
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kvfe {

constexpr int MAX_LEVELS = 8;
constexpr int MAX_BINS = 256;
constexpr int MAX_RADIUS = 127;

struct KParams {
  int W, H, B;
  int n_cu;  // compute units of the context's device (launch shapes; filled at context creation)
  int quiet_gate;  // 1: detect_commit / step_finalize / out_pack return unless NO stream of the batch is a keyframe (the
                   // speculative tail of a quiet step, do_step; a few streams)
  int kcap;  // keypoint capacity per stream
  int ccap;  // candidate capacity per stream
  int acap;  // accepted-corner capacity per stream (<= 8192)
  // detector (FeatureDetectorParams)
  int max_features, enable_anms, anms_type, min_distance, max_corners, hbins, vbins, block_size;
  int subpix_enable, subpix_win, subpix_zero, subpix_iters, sortidx_policy;
  int detector;          // FeatureDetectorType (kvfe.h KVFE_DET_*): 0 FAST, 3 GFTT
  int fast_thresh;       // cv::FastFeatureDetector threshold
  int use_harris;        // use_harris_corner_detector_: cv::cornerHarris response instead of the minimum eigenvalue
  double quality, subpix_eps2, harris_k;
  // tracker (TrackerParams)
  int klt_win, klt_iters, klt_maxlevel, max_age, predictor;
  double klt_eps2, disparity_thr;
  // stereo (StereoMatchingParams)
  int templ_cols, templ_rows, stripe_rows, stripe_cols, stereo_subpix, use_stereo_tracking;
  double min_point_dist, max_point_dist, tol_template, fx_rect, baseline;
  // frontend (FrontendParams)
  double min_kf_ns, max_kf_ns, max_disp_lkf;
  long long min_features;
  // geometric outlier rejection (FrontendParams::useRANSAC_, TrackerParams ransac_*)
  int mono;  // MonoVisionImuFrontend: no right camera, keypoints undistorted with R = I, P = K
  int use_ransac, ransac_2pt_mono, ransac_1pt_stereo, ransac_max_iters;
  int use_pnp, pnp_alg, pnp_min_inliers, map_cap;   // use_pnp_tracking (Tracker::pnp on keyframes), landmark map capacity
  double pnp_threshold;                              // 1 - cos(atan(sqrt 2 ransac_threshold_pnp / f))
  int min_mono_inliers, min_stereo_inliers;
  double ransac_thr_mono, ransac_probability;
  float ransac_thr_stereo;      // 1-point voting (float32 Mahalanobis test)
  double ransac_thr_stereo_d;   // 3-point Arun RANSAC (point distance)
  double fy_rect, cx_rect, cy_rect;  // gtsam::Cal3_S2Stereo of the rectified pair (with fx_rect, baseline)
  // RgbdVisionImuFrontend (frontend_type RGBD; `mono` is set too: one camera, R = I, P = K)
  int rgbd, depth_f32;               // depth image element type: uint16 (0) or float (1)
  int meas_right;                    // measurements carry uR of VALID right keypoints (stereo: use_stereo_tracking)
  float depth_to_m, depth_min;       // DepthParams::depth_to_meters_, min_depth_
  float mask_lo_f, mask_hi_f;        // DepthFrame::getDetectionMask bounds (float / uint16 images)
  int mask_lo_u, mask_hi_u;
  double depth_fx_b;                 // intrinsics[0] * virtual_baseline (RgbdFrame.cpp:65)
  // pyramid geometry: level 0 is the raw image, levels 1..nlevels-1 live in the pyramid buffer
  int ssd_dot4;                      // kvfe_config.ssd_impl = 1: v_dot4 SSD search for every geometry
  int ssd_f32;                       // kvfe_stereo_params.ssd_tie_policy = KVFE_SSD_TIE_F32
  int lk_one;                        // kvfe_config.lk_impl = 1: one wavefront per tracked point for every window
  int nlevels;
  int lw[MAX_LEVELS], lh[MAX_LEVELS], loff[MAX_LEVELS];
  int pyr_stride;  // bytes per stream in a pyramid buffer
};

// dense stereo (k_dense.hip): cv::StereoSGBM MODE_HH parameters after OpenCV's own defaulting
struct DenseParams {
  int W, H;
  int minD, D;           // minDisparity, numDisparities (<= 64: one wave lane per disparity)
  int minX1, width1;     // first matchable column, number of matchable columns
  int SW2;               // blockSize / 2
  int P1, P2, ftzero, uniq, disp12;
  int invalid_scaled;    // (minDisparity - 1) * 16
  int speckle_win, speckle_diff;   // filterSpeckles maxSpeckleSize, maxDiff (16 * speckleRange)
  int median5;           // DenseStereoParams::median_blur_disparity_
  int full_dp;           // 1: cv::StereoSGBM::MODE_HH (8 directions), 0: MODE_SGBM (5 directions)
  // cv::StereoBM (use_sgbm_ = false): minX1 = lofs, width1 = width - rofs - ndisp + 1, SW2 = SADWindowSize / 2,
  // ftzero = preFilterCap, uniq = uniquenessRatio, invalid_scaled = FILTERED, speckle_diff = speckleRange
  int bm;                // 1: block matching
  int bm_rofs, bm_texture;
  int bm_roi[4];         // valid-disparity rectangle (x, y, w, h); rows / columns outside are FILTERED
};
struct DenseBuffers {
  uint8_t *left = nullptr, *right = nullptr;   // [n][H][W] rectified images
  uint2* rec = nullptr;                        // [2n][H][W] Birchfield-Tomasi pixel records
  short* vol[3] = {};                          // [n][H][width1][D]
  short* disp[2] = {};                         // [n][H][W]
  int *label = nullptr, *count = nullptr, *runlen = nullptr;   // [n][H][W]
  int cap_pairs = 0;
  size_t vol_elems = 0;                        // elements per pair in vol[]
  // two-pass aggregation (MODE_HH, >= 4 pairs): block-to-block hand-over entries and the launch's ticket / error words
  unsigned long long* hand = nullptr;          // [2 passes][cap_pairs][row bands - 1][width1][64 lanes]
  size_t hand_bytes = 0;
  unsigned* agsync = nullptr;                  // [0] ticket counter, [1] error word (a wait ran out)
  unsigned hand_epoch = 0;                     // tag of the last launch's entries (1..3, 0 = buffer is zero)
  int hand_key[4] = {};                        // (pairs, width1, H, D) of the last launch: a change zeroes the buffer
};
size_t dense_handoff_bytes(const DenseParams& P, int pairs);
constexpr int AGP_SYNC_WORDS = 2 + 16;   // DenseBuffers::agsync: ticket, error, counters of the profiling build
struct ReprojectQ {
  double q[16];
};
size_t dense_volume_elems(const DenseParams& P);
// inputs in B.left / B.right, result in B.disp[0]
// two_pass_allowed = false: the eight direction sweeps whatever the number of pairs (the path a chunk is repeated on when a
// bounded hand-over wait of the two-pass launch ran out)
void launch_dense_sgbm(const DenseParams& P, DenseBuffers& B, int n, hipStream_t st, bool two_pass_allowed = true);
void launch_dense_bm(const DenseParams& P, const DenseBuffers& B, int n, hipStream_t st);
void launch_reproject_to_3d(int W, int H, const float* disp, const ReprojectQ& Q, unsigned* minkey, float* xyz,
                            hipStream_t st);

// undistortPoints constants (double) for one (K, D, RR) combination
struct UndistortDev {
  double fx, fy, cx, cy, ifx, ify;
  double k[8];
  double RR[9];
  int has_dist;  // 0 none, 1 radial-tangential, 2 equidistant (cv::fisheye)
  int pad;
};

// Constant tables shared by all streams (device pointers).
struct Tables {
  const float2* map[2];        // [H][W] (map_x, map_y) per camera
  const int4* rect_box[2];     // per 128 x 16 output tile of a camera: source box (x_lo, y_lo, 16-byte chunks per row, rows)
  const unsigned* rect_tap[2]; // [H][W] packed taps of the tiles whose box fits the LDS stage (k_rectify.hip rectify_pack_kernel)
  const float* subpix_mask;    // (2w+1)^2 Gaussian-ish weights of cv::cornerSubPix
  const float* subpix_mask10;  // same for the hard-coded 10x10 stereo refinement
  const int* circle_hw;        // [radius+1] half widths of cv::circle(FILLED)
  const unsigned char* binning_mask;  // [vbins*hbins]
  const unsigned short* sortidx;      // concatenated permutations for n = 0..max_corners
  const unsigned int* sortidx_off;    // offsets into sortidx
  const int* ransac_rnd;       // SampleConsensusProblem::rnd() stream for seed 12345 (host generated)
  int n_ransac_rnd;
  UndistortDev und_left_R;     // K1, D1, R1        -> bearing vectors
  UndistortDev und_left_RP;    // K1, D1, R1, P1    -> rectified left keypoints
  float Kf[9], Kinvf[9];       // float K / K^-1 of the ORIGINAL left camera (predictor)
};

// One frame's keypoint table (Frame::keypoints_/landmarks_/landmarks_age_/versors_)
struct FrameTab {
  float2* kp;           // [B][kcap]
  long long* lmk;       // [B][kcap]
  int* age;             // [B][kcap]
  double* versor;       // [B][kcap][3]
  int* count;           // [B]
  long long* timestamp; // [B]
};

// StereoFrame tables of the current frame
struct StereoTab {
  float2* left_rect;          // [B][kcap]
  unsigned char* left_status; // [B][kcap]
  float2* right_rect;
  unsigned char* right_status;
  double* depth;
  float2* right_kp;
  double* kp3d;               // [B][kcap][3]
};

// per-stream scalar state
struct StreamState {
  int* flags;             // [B] bit0 initialised, bit1 keyframe this step, bit2 run detection,
                          //     bit3 run stereo, bit4 capacity overflow
  int* n_tracked;         // [B]
  int* n_detected;        // [B]
  int* n_meas;            // [B]
  long long* lmk_counter; // [B] FeatureDetector.cpp:141 (per stream instead of process-wide)
  long long* frame_count; // [B]
  double* kf_R_ref;       // [B][9] keyframe_R_ref_frame_
  const double* kf_R_cur; // [B][9] this step's keyframe_R_cur_frame (input)
  const long long* in_timestamp; // [B]
  const int* in_force_kf; // [B]
  long long* meas_lmk;    // [B][kcap]
  double* meas_uLuRv;     // [B][kcap][3]
  // TrackerStatusSummary / DebugTrackerInfo of the last keyframe (Tracker-definitions.h:78-183)
  int* trk_status;        // [B][2]  kfTrackingStatus_mono_, kfTrackingStatus_stereo_
  double* trk_pose;       // [B][2][12] lkf_T_k_mono_, lkf_T_k_stereo_ (row-major 3x4)
  double* trk_info;       // [B][9]  infoMatStereoTranslation_
  int* trk_counts;        // [B][6]  mono putatives, inliers, iterations; stereo putatives, inliers; -
  // use_pnp_tracking: kfTracking_status_pnp_ / W_T_k_pnp_ and the landmark map of Tracker::updateMap
  int* pnp_status;        // [B]
  int* pnp_counts;        // [B][3]  inliers, iterations, Tracker::pnp's return value
  double* pnp_pose;       // [B][12]
  long long* map_ids;     // [B][map_cap] ascending
  double* map_xyz;        // [B][map_cap][3]
  int* map_n;             // [B]
  // a few streams, synchronous call (do_step, "quiet step"): track_finalize also publishes a stream's flags of this step
  // to a word the host polls -- flags | host_seq << 8, system scope -- so that the host learns the keyframe decision
  // without a copy and a stream synchronisation (null: not published)
  int* host_flags;        // [B] mapped pinned host memory
  int host_seq;           // this step's tag (1 .. 2^23)
};

// VIO::TrackingStatus
enum : int { TRK_VALID = 0, TRK_LOW_DISPARITY = 1, TRK_FEW_MATCHES = 2, TRK_INVALID = 3, TRK_DISABLED = 4 };

// scratch of the geometric outlier rejection kernels
struct RansacScratch {
  int2* matches;          // [B][kcap] (index in lkf, index in k)
  double* f_ref;          // [B][kcap][3] gathered bearing vectors / stereo work values
  double* f_cur;          // [B][kcap][3]
  double* relc;           // [B][kcap][9] float64 covariances of the relative translations
  float* votef;           // [B][kcap][12] float32 relative translations + covariances (voting)
  double* acc;            // [B][kcap][12] float64 information-weighted terms of the inliers
  int* inliers;           // [B][kcap]
  int* n_inliers;         // [B]
  int* cnt;               // [B][kcap] coherent-set sizes of the voting
  int* n_matches;         // [B] stereo matches entering the voting (-1: voting skipped)
};

enum : int {
  FLAG_INIT = 1,
  FLAG_KEYFRAME = 2,
  FLAG_DETECT = 4,
  FLAG_STEREO = 8,
  FLAG_OVERFLOW = 16,
  FLAG_FIRST = 32,  // processFirstStereoFrame: no stereo measurements are produced
};
// KParams::quiet_gate: no stream of the batch is a keyframe, detects or matches in this step (a few streams)
__device__ __forceinline__ bool kvfe_all_quiet(const int* flags, int B) {
  for (int t = 0; t < B; t++)
    if (flags[t] & (FLAG_KEYFRAME | FLAG_DETECT | FLAG_STEREO | FLAG_FIRST)) return false;
  return true;
}


// detection scratch
struct DetectScratch {
  unsigned long long* cand;  // [B][ccap]
  int* cand_count;           // [B]
  unsigned int* maxkey;      // [B] order-preserving key of the masked maximum
  float2* corners;           // [B][acap] GFTT output (integer valued), quality order
  int* n_corners;            // [B]
  float2* newc;              // [B][acap] corners selected by ANMS (pre sub-pixel)
  int* n_new;                // [B]
  int* sp_next;              // [B] next new corner of the stream a refinement slot takes (subpix_group_kernel; zeroed by select)
  int* need;                 // [B]
  // global-memory work arrays of the per-stream select kernel
  unsigned int* cell_items;  // [B][ccap]
  unsigned char* state;      // [B][ccap]
  unsigned long long* sortbuf;  // [B][ccap rounded to pow2]
  int sort_cap;
  // detection mask as a bitmap (k_detect.hip detect_mask_kernel; FeatureDetectorType::FAST)
  unsigned long long* me_maskbits;  // [B][H][me_mask_words(W)] bit x + 64 of row y: pixel (x, y) is masked OUT (null: GFTT)
};
__host__ __device__ inline int me_mask_words(int W) { return ((W + 64 + 63) >> 6) + 1; }   // 64 bits in front, 1 word behind


// LK scratch (component API and frontend share it)
struct LkScratch {
  float2* prev_pts;       // [B][kcap]
  float2* next_pts;       // [B][kcap]
  unsigned char* status;  // [B][kcap]
  float* err;             // [B][kcap]
  int* npts;              // [B]
  const int* skip_age;    // null, or the landmark ages of frame k-1 ([B][kcap], indexed through src_idx): a point older
                          // than maxFeatureAge is reported lost untracked (the front-end step; Tracker.cpp:167-180)
  int* src_idx;           // [B][kcap] index of point i in frame k-1 (keypoints with landmark -1 are
                          //           not tracked, Tracker.cpp:103-112)
  // dispatch order of the tracking launch (results do not depend on it): workgroup b of stream s tracks point
  // order[s][b] -- the points that took the most iterations in the previous frame first, so that the launch does not
  // end on a few slow points that started late; iters = iterations of this launch (saturated), carried to frame k
};

// gtsam::Rot3::equals(Rot3(), 1e-9) (fpEqual without the relative test): "this stream has no usable gyro rotation".
// Shared by the outlier-rejection kernels and by the host, which skips the 3-point kernel's launch when no stream of the
// batch needs it (two copies of the test could drift apart).  NaN / Inf compare false.
__host__ __device__ inline bool rot_is_identity(const double* R) {
  for (int i = 0; i < 9; i++) {
    const double a = R[i], b = (i % 4 == 0) ? 1.0 : 0.0;
    if (a != a) return false;                                   // NaN
    if (a - a != 0.0) return false;                             // +-Inf
    if (a == b) continue;
    const double d = a - b;
    if (!((d < 0 ? -d : d) <= 1e-9)) return false;
  }
  return true;
}

__host__ __device__ inline int reflect101(int p, int len) {
  if ((unsigned)p < (unsigned)len) return p;
  if (len == 1) return 0;
  do {
    if (p < 0)
      p = -p;
    else
      p = 2 * len - 2 - p;
  } while ((unsigned)p >= (unsigned)len);
  return p;
}

// ---- launchers (each enqueues on `st`; no synchronisation) -----------------------------------
// K1: cv::remap of both images per stream.  act_flag: only streams with (flags & act_flag); skip (optional): streams
// with skip[s] != 0 are left alone (rectified earlier in the step).
void launch_rectify(const KParams& P, const Tables& T, const unsigned char* const src[2],
                    size_t src_row_stride, size_t src_img_stride, unsigned char* const dst[2],
                    const int* flags, int act_flag, hipStream_t st, const int* skip = nullptr);
// cv::equalizeHist of B images: dst[s] = lut_s(src[s]); hist: [B][256] int scratch (zeroed here)
void launch_equalize_hist(int W, int H, int B, const unsigned char* src, size_t src_row_stride,
                          size_t src_img_stride, unsigned char* dst, int* hist, hipStream_t st);
// K4a: pyramid levels 1..nlevels-1 of `img` into pyr.
// source boxes of the rectification tiles (context creation; Tables::rect_box)
void launch_rectify_boxes(const float2* map, int W, int H, int4* box, hipStream_t st);
size_t rectify_box_count(int W, int H);
// packed taps of the staged tiles (context creation, behind the boxes; Tables::rect_tap)
void launch_rectify_taps(const float2* map, int W, int H, const int4* box, unsigned* tap, hipStream_t st);
void launch_pyramid(const KParams& P, const unsigned char* img, size_t row_stride,
                    size_t img_stride, unsigned char* pyr, hipStream_t st,
                    unsigned char* level0_copy = nullptr);
// K4c: pyramidal LK for npts[s] points per stream.
void launch_lk(const KParams& P, const unsigned char* prev_img, size_t prev_row_stride,
               size_t prev_img_stride, const unsigned char* prev_pyr, const unsigned char* cur_img,
               size_t cur_row_stride, size_t cur_img_stride, const unsigned char* cur_pyr,
               const LkScratch& lk, int max_pts, hipStream_t st, bool want_err = true);
// k_lk4.hip: four points per wavefront (no error output); false when the window size / pyramid is not covered
bool launch_lk4(const KParams& P, const unsigned char* prev_img, size_t prev_row_stride, size_t prev_img_stride,
                const unsigned char* prev_pyr, const unsigned char* cur_img, size_t cur_row_stride,
                size_t cur_img_stride, const unsigned char* cur_pyr, const LkScratch& lk, int max_pts, hipStream_t st);
// predictor + gather of the reference keypoints (Tracker.cpp:103-129)
void launch_track_prepare(const KParams& P, const Tables& T, const FrameTab& km1,
                          const StreamState& S, const LkScratch& lk, hipStream_t st);
// survivors -> frame k, bearing vectors, keyframe decision (Tracker.cpp:167-189,
// StereoVisionImuFrontend.cpp:313-347, VisionImuFrontend.cpp:175-232)
void launch_track_finalize(const KParams& P, const Tables& T, const FrameTab& km1,
                           const FrameTab& lkf, const FrameTab& k, const StreamState& S,
                           const LkScratch& lk, hipStream_t st);
// K2a-c: min-eigenvalue local maxima + masked maximum.  user_mask may be null.
void launch_mineig(const KParams& P, const Tables& T, const unsigned char* img, size_t row_stride,
                   size_t img_stride, const unsigned char* user_mask, const FrameTab& k,
                   const StreamState& S, const DetectScratch& D, int use_discs, hipStream_t st);
// FeatureDetectorType::FAST: cv::FastFeatureDetector(fast_thresh, nonmaxSuppression = true)::detect into the candidate list
void launch_fast(const KParams& P, const Tables& T, const unsigned char* img, size_t row_stride, size_t img_stride,
                 const unsigned char* user_mask, const FrameTab& k, const StreamState& S, const DetectScratch& D,
                 int use_discs, hipStream_t st);
// K2d + ANMS (per stream): threshold, greedy min-distance, sort, ANMS.
void launch_select(const KParams& P, const Tables& T, const FrameTab& k, const StreamState& S,
                   const DetectScratch& D, int fixed_need /* <0: from frame */, hipStream_t st);
// K3 + append: cornerSubPix on the new corners and append to frame k
void launch_subpix_append(const KParams& P, const Tables& T, const unsigned char* img,
                          size_t row_stride, size_t img_stride, const FrameTab& k,
                          const StreamState& S, const DetectScratch& D, int append,
                          hipStream_t st);
// the per-stream state the next step's tracking reads (keyframe_R_ref_frame_, "initialised", the frame's keypoint
// count): known once the new corners are SELECTED.  launch_subpix_append(append = 2) leaves it to this launch.
int detect_new_bound(const KParams& P);   // upper bound of the new corners per stream and frame
// cv::cornerSubPix on arbitrary points (component API)
void launch_subpix_points(const KParams& P, const float* mask_tab, const unsigned char* img,
                          size_t row_stride, int W, int H, float2* pts, int n, int win,
                          int max_iters, double eps2, hipStream_t st);
// mono front-end: Camera::undistortKeypoints (Camera.cpp:110-133) = the left-keypoint half of K7
void launch_undistort_left(const KParams& P, const Tables& T, const FrameTab& k, const StereoTab& ST,
                           const StreamState& S, int act_flag, int max_kp, hipStream_t st);
// K7 + K5: rectify left keypoints, epipolar SSD, depth, right keypoints, 3D
void launch_stereo(const KParams& P, const Tables& T, const unsigned char* left_rect,
                   const unsigned char* right_rect, const FrameTab& k, const StereoTab& ST,
                   const StreamState& S, int act_flag, int max_kp, int mode /* 0 all, 1 tracked,
                   2 newly detected */, hipStream_t st);
// StereoMatcher::getRightKeypointsRectified only (component API): left_rect/status given
void launch_stereo_match_only(const KParams& P, const Tables& T, const unsigned char* left_rect,
                              const unsigned char* right_rect, const float2* left_rect_kp,
                              const unsigned char* left_status, int n, float2* right_rect_kp,
                              unsigned char* right_status, double* score, hipStream_t st);
// end of step: measurements, lkf <- k for keyframes (incl. the stereo tables the next keyframe's
// outlier rejection reads), rotation bookkeeping
// RgbdVisionImuFrontend: DepthFrame::getDetectionMask for the streams flagged `act_flag`, and
// RgbdFrame::fillStereoFrame for all keypoints of those streams (depth: row_stride / img_stride in elements)
void launch_depth_mask(const KParams& P, const void* depth, size_t row_stride, size_t img_stride,
                       const StreamState& S, int act_flag, unsigned char* mask, hipStream_t st);
void launch_rgbd_fill(const KParams& P, const Tables& T, const void* depth, size_t row_stride, size_t img_stride,
                      const FrameTab& k, const StereoTab& ST, const StreamState& S, int act_flag, int max_kp,
                      hipStream_t st);
void launch_step_finalize(const KParams& P, const FrameTab& k, const FrameTab& lkf,
                          const StereoTab& ST, const StereoTab& LST, const StreamState& S,
                          hipStream_t st);
// VisionImuFrontend::outlierRejectionMono on keyframes (2-point RANSAC, rotation given):
// landmarks of the outliers are set to -1 in frame k, status / pose -> S.trk_*
void launch_mono_ransac(const KParams& P, const Tables& T, const FrameTab& k, const FrameTab& lkf,
                        const StreamState& S, const RansacScratch& RS, hipStream_t st);
// VisionImuFrontend::outlierRejectionStereo on keyframes (1-point voting)
void launch_stereo_ransac(const KParams& P, const Tables& T, const FrameTab& k, const FrameTab& lkf,
                          const StereoTab& ST, const StereoTab& LST, const StreamState& S,
                          const RansacScratch& RS, int max_matches, hipStream_t st, bool need_arun = true);
// component API: the two problems on caller-supplied matches (one stream)
void launch_ransac_2d2d_points(const KParams& P, const Tables& T, const double* f_ref,
                               const double* f_cur, int n, const double* R, const RansacScratch& RS,
                               int* out_status, double* out_pose, int* out_counts, hipStream_t st);
void launch_ransac_3d3d_points(const KParams& P, const Tables& T, const float* ref_left,
                               const float* ref_right_x, const double* ref_p3, const float* cur_left,
                               const float* cur_right_x, const double* cur_p3, int n, const double* R,
                               const RansacScratch& RS, int* out_status, double* out_pose,
                               double* out_info, int* out_counts, hipStream_t st);
void launch_ransac_2d2d_nister_points(const KParams& P, const Tables& T, const double* f_ref, const double* f_cur,
                                      int n, const RansacScratch& RS, int* out_status, double* out_pose,
                                      int* out_counts, hipStream_t st);
// Dynamic LDS a launch of `kernel` may request on the CURRENT device (gfx950: 160 KB per workgroup minus the kernel's
// static LDS); raises the kernel's opt-in limit accordingly (above 64 KB a launch fails -- and aborts the HSA queue --
// without it).  Called at kvfe_create for every kernel whose dynamic LDS scales with the per-frame keypoint capacity.
inline int lds_dynamic_budget(const void* kernel) {
  hipFuncAttributes a;
  if (hipFuncGetAttributes(&a, kernel) != hipSuccess) return 48 * 1024;
  const int budget = 160 * 1024 - (int)a.sharedSizeBytes - 256;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, budget) != hipSuccess) return 64 * 1024;
  return budget;
}
// largest per-frame keypoint capacity (KParams::kcap) the tracking bookkeeping / the outlier-rejection kernels can hold
// in LDS on the current device
int track_max_kcap();
int ransac_max_kcap(bool nister);
// Tracker::pnp, EPNP RANSAC over n 2D-3D correspondences (k_pnp.inl): out_counts = [n_inliers, iterations, success]
// outlierRejectionPnP of the keyframes of this step (frame k after the stereo outlier rejection)
void launch_pnp_frontend(const KParams& P, const Tables& T, const FrameTab& k, const StereoTab& ST, const StreamState& S,
                         const RansacScratch& RS, hipStream_t st);
void launch_pnp(const KParams& P, const Tables& T, int algorithm, const double* f, const double* p, int n,
                double threshold, int min_inliers, int* inliers, int* out_status, double* out_pose, int* out_counts,
                hipStream_t st);
void launch_ransac_3d3d_arun_points(const KParams& P, const Tables& T, const double* p1, const double* p2, int n,
                                    const RansacScratch& RS, int* out_status, double* out_pose, int* out_counts,
                                    hipStream_t st);
// undistort keypoints with an arbitrary UndistortDev (component API)
void launch_undistort_points(const UndistortDev& U, const float2* in, int n, float2* out,
                             double* versors, hipStream_t st);
// component-level keypoint methods (k_components.hip)
void launch_check_undistorted_rectified(const float2* map, int W, int H, const float2* distorted,
                                        const float2* undistorted, int n, float pixel_tol, float2* out_xy,
                                        unsigned char* out_status, hipStream_t st);
void launch_distort_unrectify(const float2* map, int W, int H, const float2* rect_xy, const unsigned char* status, int n,
                              float2* out_xy, hipStream_t st);
void launch_depth_from_matches(const float2* left_xy, const unsigned char* left_status, const float2* right_xy,
                               unsigned char* right_status, int n, double fx_b, double min_dist, double max_dist,
                               double* depth, hipStream_t st);
// ---- output side: one packed record per stream and step (StereoFrontendOutput, StereoVisionImuFrontend-definitions.h:25-91) ----
// step_finalize's successor `out_pack_kernel` gathers everything kvfe_frontend_get_output returns for a stream -- counters,
// TrackerStatusSummary, the frame / stereo / measurement arrays cut to the entries in use -- into ONE contiguous record
// (header + 16-byte aligned arrays); the records of a step travel to a pinned host ring slot in ONE transfer of the DMA
// engine, and kvfe_frontend_get_output is a memcpy out of that slot (round 3: ~27 blocking hipMemcpy per stream).
// (The first form of round 4 moved them with a copy KERNEL writing the mapped slot: every kernel of the step that ended
// while its PCIe writes were in flight completed only when they had drained -- the tracking launch ended with the
// transfer, 0.09 ms late, in most steps.  The DMA engine does not have that effect: +5.8 % on the 64-stream headline.)
constexpr int OUT_HDR_BYTES = 512;
constexpr int OUT_RING = 3;   // steps whose output records are kept (kvfe_frontend_get_output_at: steps_back < OUT_RING)
struct OutHeader {
  int n_keypoints, flags, n_tracked, n_detected, n_meas;
  int trk_status[2], trk_counts[6], pnp_status, pnp_counts[3];
  long long frame_count;
  unsigned long long used_bytes;   // header + arrays of this record
  double trk_pose[24], trk_info[9], pnp_pose[12];
};
static_assert(sizeof(OutHeader) <= OUT_HDR_BYTES, "output header grew past its slot");
struct OutLayout {   // byte offsets inside a record
  size_t lmk, age, kp, versor, left_rect, left_status, right_rect, right_status, depth, right_kp, kp3d, meas_lmk, meas, end;
};
__host__ __device__ inline size_t out_al16(size_t v) { return (v + 15) & ~(size_t)15; }
__host__ __device__ inline OutLayout out_layout(int n, int m, bool stereo) {
  OutLayout L;
  size_t o = OUT_HDR_BYTES;
  L.lmk = o;          o = out_al16(o + sizeof(long long) * (size_t)n);
  L.age = o;          o = out_al16(o + sizeof(int) * (size_t)n);
  L.kp = o;           o = out_al16(o + sizeof(float2) * (size_t)n);
  L.versor = o;       o = out_al16(o + sizeof(double) * 3 * (size_t)n);
  const size_t ns = stereo ? (size_t)n : 0;
  L.left_rect = o;    o = out_al16(o + sizeof(float2) * ns);
  L.left_status = o;  o = out_al16(o + ns);
  L.right_rect = o;   o = out_al16(o + sizeof(float2) * ns);
  L.right_status = o; o = out_al16(o + ns);
  L.depth = o;        o = out_al16(o + sizeof(double) * ns);
  L.right_kp = o;     o = out_al16(o + sizeof(float2) * ns);
  L.kp3d = o;         o = out_al16(o + sizeof(double) * 3 * ns);
  L.meas_lmk = o;     o = out_al16(o + sizeof(long long) * (size_t)m);
  L.meas = o;         o = out_al16(o + sizeof(double) * 3 * (size_t)m);
  L.end = o;
  return L;
}
// A step's slot: [table: B + 1 byte offsets (uint64; entry B = end of the data), 256-byte aligned][records back to back,
// each 256-byte aligned].  Only the bytes in use cross PCIe: the host enqueues the transfer before it can know their
// number, so it transfers what the last completed step needed plus a margin, and an access that finds a record beyond
// that fetches the rest from the staging buffer first (kvfe_api.cpp locate_output).
__host__ __device__ inline size_t out_record_bytes(int n, int m, bool stereo) {
  return (out_layout(n, m, stereo).end + 255) & ~(size_t)255;
}
inline size_t out_table_bytes(int B) { return (sizeof(unsigned long long) * (size_t)(B + 1) + 255) & ~(size_t)255; }
inline size_t out_slot_bytes(int B, int rec_cap) { return out_table_bytes(B) + (size_t)B * out_record_bytes(rec_cap, rec_cap, true); }
void launch_out_pack(const KParams& P, const FrameTab& k, const StereoTab& ST, const StreamState& S, unsigned char* dst,
                     size_t table_bytes, int rec_cap, hipStream_t st);
void launch_mark_lost_tracks(const KParams& P, const FrameTab& km1, const LkScratch& lk, int max_pts, hipStream_t st);
void launch_predict_flow(const KParams& P, const Tables& T, const double* R, const float2* prev,
                         int n, float2* out, hipStream_t st);

}  // namespace kvfe
