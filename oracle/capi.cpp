// TEST INFRASTRUCTURE — C entry points of the CPU oracle for ctypes
// (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).  Never linked into
// or called from the product library.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <vector>

#include "kimera.hpp"
#include "ocv.hpp"
#include "opengv_re.hpp"

using kimera::StatusKeypoint;
using ocv::Point2f;

#define KVO_API extern "C" __attribute__((visibility("default")))

KVO_API const char* kvo_version() { return "kvfe-oracle 0.1 (OpenCV-4.2 restatement; test infrastructure)"; }

namespace ocv { extern int g_stereo_rectify_variant; }
KVO_API void kvo_set_stereo_rectify_variant(int v) { ocv::g_stereo_rectify_variant = v; }

// ---- calib ---------------------------------------------------------------
KVO_API int kvo_compute_rectification(const kvfe_camera_params* l, const kvfe_camera_params* r,
                                      kvfe_rectification* out) {
  kimera::StereoCamera cam;
  cam.init(*l, *r);
  *out = cam.rect;
  return 0;
}

struct kvo_camera {
  kimera::StereoCamera cam;
};

KVO_API kvo_camera* kvo_camera_create(const kvfe_camera_params* l, const kvfe_camera_params* r) {
  kvo_camera* c = new kvo_camera;
  c->cam.init(*l, *r);
  return c;
}
KVO_API void kvo_camera_destroy(kvo_camera* c) { delete c; }
KVO_API void kvo_camera_get_rectification(const kvo_camera* c, kvfe_rectification* out) {
  *out = c->cam.rect;
}
KVO_API void kvo_camera_get_maps(const kvo_camera* c, int cam, float* mx, float* my) {
  std::memcpy(mx, c->cam.map_x[cam].data(), c->cam.map_x[cam].size() * sizeof(float));
  std::memcpy(my, c->cam.map_y[cam].data(), c->cam.map_y[cam].size() * sizeof(float));
}
KVO_API void kvo_camera_rectify_image(const kvo_camera* c, int cam, const uint8_t* src,
                                      size_t stride, uint8_t* dst) {
  c->cam.undistortRectifyImage(cam, src, stride, dst);
}
KVO_API void kvo_camera_undistort_keypoints(const kvo_camera* c, int cam, const float* xy, int n,
                                            int useR, int useP, float* out) {
  c->cam.undistortRectifyKeypoints(cam, (const Point2f*)xy, n, useR != 0, useP != 0,
                                   (Point2f*)out);
}
KVO_API void kvo_camera_bearing_vectors(const kvo_camera* c, int cam, const float* xy, int n,
                                        double* out) {
  for (int i = 0; i < n; i++) c->cam.getBearingVector(cam, Point2f{xy[2 * i], xy[2 * i + 1]}, out + 3 * i);
}
KVO_API void kvo_camera_undistort_rectify_left(const kvo_camera* c, const float* xy, int n,
                                               float* out_xy, uint8_t* out_status) {
  std::vector<Point2f> kps((const Point2f*)xy, (const Point2f*)xy + n);
  std::vector<StatusKeypoint> out;
  c->cam.undistortRectifyLeftKeypoints(kps, out);
  for (int i = 0; i < n; i++) {
    out_xy[2 * i] = out[i].kp.x;
    out_xy[2 * i + 1] = out[i].kp.y;
    out_status[i] = out[i].status;
  }
}

// ---- imgproc ---------------------------------------------------------------
KVO_API void kvo_remap(const uint8_t* src, int w, int h, size_t stride, const float* mx,
                       const float* my, uint8_t* dst) {
  ocv::remap_linear_replicate(src, w, h, stride, dst, w, h, w, mx, my);
}
KVO_API void kvo_corner_min_eigen_val(const uint8_t* img, int w, int h, size_t stride, int block,
                                      float* eig) {
  ocv::cornerMinEigenVal(img, w, h, stride, block, eig);
}
// cv::FAST(TYPE_9_16) / cv::FastFeatureDetector::detect: out = (x, y, response) triples, returns the count
KVO_API int kvo_fast_detect(const uint8_t* img, int w, int h, size_t stride, const uint8_t* mask, size_t mask_stride,
                            int threshold, int nonmax, float* out_xyr, int capacity) {
  std::vector<ocv::FastKeyPoint> k;
  ocv::fastDetect(img, w, h, stride, mask, mask_stride, threshold, nonmax != 0, k);
  const int n = std::min((int)k.size(), capacity);
  for (int i = 0; i < n; i++) {
    out_xyr[3 * i] = k[i].x;
    out_xyr[3 * i + 1] = k[i].y;
    out_xyr[3 * i + 2] = k[i].response;
  }
  return (int)k.size();
}
KVO_API void kvo_corner_harris(const uint8_t* img, int w, int h, size_t stride, int block, double k, float* dst) {
  ocv::cornerHarris(img, w, h, stride, block, k, dst);
}
KVO_API int kvo_good_features_to_track_harris(const uint8_t* img, int w, int h, size_t stride,
                                              const uint8_t* mask, size_t mask_stride, int maxCorners,
                                              double quality, double minDist, int block, double k, float* out_xy,
                                              float* out_quality, int capacity) {
  std::vector<Point2f> c;
  std::vector<float> q;
  ocv::goodFeaturesToTrack(img, w, h, stride, mask, mask_stride, maxCorners, quality, minDist,
                           block, c, &q, true, k);
  int n = std::min((int)c.size(), capacity);
  for (int i = 0; i < n; i++) {
    out_xy[2 * i] = c[i].x;
    out_xy[2 * i + 1] = c[i].y;
    if (out_quality) out_quality[i] = q[i];
  }
  return (int)c.size();
}
KVO_API int kvo_good_features_to_track(const uint8_t* img, int w, int h, size_t stride,
                                       const uint8_t* mask, size_t mask_stride, int maxCorners,
                                       double quality, double minDist, int block, float* out_xy,
                                       float* out_quality, int capacity) {
  std::vector<Point2f> c;
  std::vector<float> q;
  ocv::goodFeaturesToTrack(img, w, h, stride, mask, mask_stride, maxCorners, quality, minDist,
                           block, c, &q);
  int n = std::min((int)c.size(), capacity);
  for (int i = 0; i < n; i++) {
    out_xy[2 * i] = c[i].x;
    out_xy[2 * i + 1] = c[i].y;
    if (out_quality) out_quality[i] = q[i];
  }
  return (int)c.size();
}
KVO_API void kvo_draw_detection_mask(int w, int h, const float* xy, int n, int radius,
                                     uint8_t* mask) {
  std::memset(mask, 255, (size_t)w * h);
  for (int i = 0; i < n; i++)
    ocv::circle_filled(mask, w, h, w, ocv::cvRoundf(xy[2 * i]), ocv::cvRoundf(xy[2 * i + 1]), radius, 0);
}
KVO_API void kvo_corner_subpix(const uint8_t* img, int w, int h, size_t stride, float* xy, int n,
                               int win, int zero_zone, int max_iters, double eps) {
  ocv::cornerSubPix(img, w, h, stride, (Point2f*)xy, n, win, zero_zone, max_iters, eps);
}
KVO_API void kvo_equalize_hist(const uint8_t* src, int w, int h, size_t stride, uint8_t* dst) {
  ocv::equalizeHist(src, w, h, stride, dst, w);
}
KVO_API void kvo_pyr_down(const uint8_t* src, int w, int h, size_t stride, uint8_t* dst) {
  ocv::pyrDown(src, w, h, stride, dst, (w + 1) / 2, (h + 1) / 2, (w + 1) / 2);
}
KVO_API int kvo_calc_optical_flow_pyr_lk(const uint8_t* prev, const uint8_t* next, int w, int h,
                                         size_t stride, const float* prev_xy, float* next_xy,
                                         int n, uint8_t* status, float* err, int win,
                                         int maxLevel, int maxIter, double eps,
                                         int use_initial_flow, double minEigThreshold) {
  return ocv::calcOpticalFlowPyrLK(prev, next, w, h, stride, (const Point2f*)prev_xy,
                                   (Point2f*)next_xy, n, status, err, win, maxLevel, maxIter, eps,
                                   use_initial_flow != 0, minEigThreshold);
}

// ---- dense stereo (parity unpinned) --------------------------------------------
namespace ocv { extern short* g_sgbm_debug_C; extern short* g_sgbm_debug_S; }
// params: minDisparity, numDisparities, blockSize, P1, P2, disp12MaxDiff, preFilterCap,
// uniquenessRatio, speckleWindowSize, speckleRange, mode (0 SGBM, 1 HH)
KVO_API void kvo_stereo_sgbm(const uint8_t* left, const uint8_t* right, int w, int h, size_t stride,
                             const int* params, short* disp, short* debug_C, short* debug_S) {
  ocv::StereoSGBMParams p{params[0], params[1], params[2], params[3], params[4], params[5],
                          params[6], params[7], params[8], params[9], params[10]};
  ocv::g_sgbm_debug_C = debug_C;
  ocv::g_sgbm_debug_S = debug_S;
  ocv::stereoSGBM_compute(left, right, w, h, stride, p, disp, w);
  ocv::g_sgbm_debug_C = ocv::g_sgbm_debug_S = nullptr;
}
// params: preFilterCap, blockSize, minDisparity, numDisparities, textureThreshold, uniquenessRatio,
// speckleRange, speckleWindowSize, roi1[4], roi2[4]
KVO_API void kvo_stereo_bm(const uint8_t* left, const uint8_t* right, int w, int h, size_t stride,
                           const int* params, short* disp) {
  ocv::StereoBMParams p;
  p.preFilterCap = params[0]; p.blockSize = params[1]; p.minDisparity = params[2];
  p.numDisparities = params[3]; p.textureThreshold = params[4]; p.uniquenessRatio = params[5];
  p.speckleRange = params[6]; p.speckleWindowSize = params[7];
  for (int i = 0; i < 4; i++) { p.roi1[i] = params[8 + i]; p.roi2[i] = params[12 + i]; }
  ocv::stereoBM_compute(left, right, w, h, stride, p, disp, w);
}
KVO_API void kvo_median_blur_16s(const short* src, int w, int h, short* dst, int ksize) {
  ocv::medianBlur16s(src, w, h, w, dst, w, ksize);
}
KVO_API void kvo_filter_speckles_16s(short* img, int w, int h, int newVal, int maxSpeckleSize, int maxDiff) {
  ocv::filterSpeckles16s(img, w, h, w, newVal, maxSpeckleSize, maxDiff);
}
KVO_API void kvo_reproject_image_to_3d(const float* disparity, int w, int h, const double* Q,
                                       int handle_missing, float* xyz) {
  ocv::reprojectImageTo3D(disparity, w, h, w, Q, handle_missing != 0, xyz);
}
// StereoMatcher::denseStereoReconstruction (StereoMatcher.cpp:32-121) on rectified images
KVO_API void kvo_dense_stereo_reconstruction(const kvfe_dense_stereo_params* dp, const int* roi1,
                                             const int* roi2, const uint8_t* left_rect,
                                             const uint8_t* right_rect, int w, int h, size_t stride,
                                             short* disp) {
  kimera::denseStereoReconstruction(*dp, roi1, roi2, left_rect, right_rect, w, h, stride, disp);
}

// ---- reference-owned logic ---------------------------------------------------
KVO_API void kvo_sortidx_permutation(int n, int policy, int* idx) {
  std::vector<int> v;
  kimera::sortidx_descending_equal_keys(n, policy, v);
  std::memcpy(idx, v.data(), sizeof(int) * n);
}
// what the reference binary does: std::sort with an always-false comparator + reverse
KVO_API void kvo_sortidx_permutation_stdsort(int n, int* idx) {
  std::vector<int> keys(n, 0);
  for (int i = 0; i < n; i++) idx[i] = i;
  const int* k = keys.data();
  std::sort(idx, idx + n, [k](int a, int b) { return k[a] < k[b]; });
  for (int j = 0; j < n / 2; j++) std::swap(idx[j], idx[n - 1 - j]);
}

KVO_API int kvo_suppress_non_max(const float* xy, int n, int numRet, int cols, int rows,
                                 const kvfe_detector_params* p, float* out_xy, int capacity) {
  std::vector<Point2f> in((const Point2f*)xy, (const Point2f*)xy + n), out;
  if (!kimera::suppressNonMax(in, numRet, cols, rows, *p, out)) return -2;
  int m = std::min((int)out.size(), capacity);
  std::memcpy(out_xy, out.data(), sizeof(Point2f) * m);
  return (int)out.size();
}

KVO_API int kvo_feature_detection(const uint8_t* img, int w, int h, size_t stride,
                                  const float* tracked_xy, int n_tracked, int need,
                                  const kvfe_detector_params* p, float* out_xy, int capacity,
                                  float* raw_xy, int raw_capacity, int* raw_n) {
  std::vector<Point2f> tracked((const Point2f*)tracked_xy, (const Point2f*)tracked_xy + n_tracked);
  std::vector<Point2f> corners, raw;
  if (!kimera::featureDetection(img, w, h, stride, tracked, need, *p, corners, &raw)) return -2;
  int m = std::min((int)corners.size(), capacity);
  std::memcpy(out_xy, corners.data(), sizeof(Point2f) * m);
  if (raw_xy) {
    int r = std::min((int)raw.size(), raw_capacity);
    std::memcpy(raw_xy, raw.data(), sizeof(Point2f) * r);
  }
  if (raw_n) *raw_n = (int)raw.size();
  return (int)corners.size();
}

KVO_API void kvo_predict_sparse_flow(int type, const kvfe_camera_params* cam, const float* prev,
                                     int n, const double* R, float* next) {
  double K[9];
  kimera::camera_matrix(*cam, K);
  kimera::predictSparseFlow(type, K, cam->width, cam->height, (const Point2f*)prev, n, R,
                            (Point2f*)next);
}

KVO_API void kvo_get_right_keypoints_rectified(const uint8_t* left_rect, const uint8_t* right_rect,
                                               int w, int h, size_t stride, const float* left_xy,
                                               const uint8_t* left_status, int n, double fx,
                                               double baseline, const kvfe_stereo_params* p,
                                               float* right_xy, uint8_t* right_status,
                                               double* score) {
  std::vector<StatusKeypoint> left(n), right;
  for (int i = 0; i < n; i++) left[i] = {left_status[i], {left_xy[2 * i], left_xy[2 * i + 1]}};
  std::vector<double> sc;
  kimera::getRightKeypointsRectified(left_rect, right_rect, w, h, stride, left, fx, baseline, *p,
                                     right, &sc);
  for (int i = 0; i < n; i++) {
    right_xy[2 * i] = right[i].kp.x;
    right_xy[2 * i + 1] = right[i].kp.y;
    right_status[i] = right[i].status;
    if (score) score[i] = sc[i];
  }
}

KVO_API void kvo_sparse_stereo_reconstruction(const kvo_camera* c, const kvfe_stereo_params* p,
                                              const uint8_t* left, const uint8_t* right,
                                              size_t stride, const float* left_xy, int n,
                                              kvfe_stereo_output* out) {
  const int w = c->cam.w, h = c->cam.h;
  kimera::StereoFrame sf;
  sf.left.w = w;
  sf.left.h = h;
  sf.left.img.resize((size_t)w * h);
  sf.right_img.resize((size_t)w * h);
  for (int y = 0; y < h; y++) {
    std::memcpy(&sf.left.img[(size_t)y * w], left + y * stride, w);
    std::memcpy(&sf.right_img[(size_t)y * w], right + y * stride, w);
  }
  sf.left.keypoints.assign((const Point2f*)left_xy, (const Point2f*)left_xy + n);
  sf.left.versors.resize((size_t)n * 3);
  for (int i = 0; i < n; i++) c->cam.getBearingVector(0, sf.left.keypoints[i], &sf.left.versors[3 * i]);
  kimera::sparseStereoReconstruction(c->cam, *p, sf);
  for (int i = 0; i < n; i++) {
    if (out->left_rect_xy) {
      out->left_rect_xy[2 * i] = sf.left_kp_rect[i].kp.x;
      out->left_rect_xy[2 * i + 1] = sf.left_kp_rect[i].kp.y;
    }
    if (out->left_status) out->left_status[i] = sf.left_kp_rect[i].status;
    if (out->right_rect_xy) {
      out->right_rect_xy[2 * i] = sf.right_kp_rect[i].kp.x;
      out->right_rect_xy[2 * i + 1] = sf.right_kp_rect[i].kp.y;
    }
    if (out->right_status) out->right_status[i] = sf.right_kp_rect[i].status;
    if (out->depth) out->depth[i] = sf.depth[i];
    if (out->right_xy) {
      out->right_xy[2 * i] = sf.right_kp[i].x;
      out->right_xy[2 * i + 1] = sf.right_kp[i].y;
    }
    if (out->keypoints_3d)
      for (int k = 0; k < 3; k++) out->keypoints_3d[3 * i + k] = sf.kp3d[3 * i + k];
  }
  if (out->left_rect_img) std::memcpy(out->left_rect_img, sf.left_rect.data(), (size_t)w * h);
  if (out->right_rect_img) std::memcpy(out->right_rect_img, sf.right_rect.data(), (size_t)w * h);
}

// ---- geometric outlier rejection -------------------------------------------------
static void fill_ransac_out(const kimera::RansacOut& r, int32_t* inliers, kvfe_ransac_output* out) {
  out->status = r.status;
  out->n_inliers = (int)r.inliers.size();
  out->iterations = r.iterations;
  out->reserved0 = 0;
  std::memcpy(out->pose, r.pose, sizeof(out->pose));
  std::memcpy(out->info, r.info, sizeof(out->info));
  if (inliers)
    for (size_t i = 0; i < r.inliers.size(); i++) inliers[i] = r.inliers[i];
}
KVO_API void kvo_outlier_rejection_2d2d_given_rotation(const double* f_ref, const double* f_cur,
                                                       int n, const double* R,
                                                       const kvfe_tracker_params* tp,
                                                       int32_t* inliers, kvfe_ransac_output* out) {
  fill_ransac_out(kimera::outlierRejection2d2dGivenRot(f_ref, f_cur, n, R, *tp), inliers, out);
}
KVO_API void kvo_outlier_rejection_3d3d_given_rotation(
    const kvo_camera* c, const float* ref_left_xy, const float* ref_right_x, const double* ref_p3,
    const float* cur_left_xy, const float* cur_right_x, const double* cur_p3, int n,
    const double* R, const kvfe_tracker_params* tp, int32_t* inliers, kvfe_ransac_output* out) {
  kimera::StereoCalib K;
  K.fx = c->cam.rect.P1[0];
  K.fy = c->cam.rect.P1[5];
  K.s = c->cam.rect.P1[1];
  K.cx = c->cam.rect.P1[2];
  K.cy = c->cam.rect.P1[6];
  K.b = c->cam.rect.baseline;
  fill_ransac_out(kimera::outlierRejection3d3dGivenRot(ref_left_xy, ref_right_x, ref_p3, cur_left_xy,
                                                       cur_right_x, cur_p3, n, K, R, *tp),
                  inliers, out);
}
KVO_API void kvo_pnp(const double* bearings, const double* points, int n, double avg_focal_length,
                     const kvfe_tracker_params* tp, const kvfe_pnp_params* pp, int32_t* inliers,
                     kvfe_ransac_output* out) {
  bool success = false;
  fill_ransac_out(kimera::pnp(bearings, points, n, avg_focal_length, *tp, *pp, &success), inliers, out);
  out->reserved0 = success ? 1 : 0;
}
KVO_API int kvo_p3p_kneip(const double* bearings, const double* points, const int* idx3, double* sol) {
  return opengv_re::p3p_kneip_solutions(bearings, points, idx3, sol);
}
KVO_API void kvo_quartic_roots(const double* p5, double* roots4) { opengv_re::quartic_roots(p5, roots4); }
KVO_API int kvo_epnp(const double* bearings, const double* points, const int* idx, int n, double* model) {
  return opengv_re::epnp(bearings, points, idx, n, model);
}
KVO_API int kvo_fivept_nister(const double* f1, const double* f2, const int* idx5, double* E_out) {
  return opengv_re::fivept_nister_essentials(f1, f2, idx5, E_out);
}
KVO_API void kvo_outlier_rejection_2d2d(const double* f_ref, const double* f_cur, int n,
                                        const kvfe_tracker_params* tp, int32_t* inliers,
                                        kvfe_ransac_output* out) {
  fill_ransac_out(kimera::outlierRejection2d2d(f_ref, f_cur, n, *tp), inliers, out);
}
KVO_API void kvo_outlier_rejection_3d3d(const double* ref_p3, const double* cur_p3, int n,
                                        const kvfe_tracker_params* tp, int32_t* inliers,
                                        kvfe_ransac_output* out) {
  fill_ransac_out(kimera::outlierRejection3d3d(ref_p3, cur_p3, n, *tp), inliers, out);
}
KVO_API void kvo_get_point3_and_covariance(const kvo_camera* c, double uL, double uR, double v,
                                           const double* p3, const double* Rmat, double* point,
                                           double* cov) {
  kimera::StereoCalib K;
  K.fx = c->cam.rect.P1[0];
  K.fy = c->cam.rect.P1[5];
  K.s = c->cam.rect.P1[1];
  K.cx = c->cam.rect.P1[2];
  K.cy = c->cam.rect.P1[6];
  K.b = c->cam.rect.baseline;
  kimera::getPoint3AndCovariance(K, uL, uR, v, p3, Rmat, point, cov);
}
// opengv::sac::Ransac<PointCloudSacProblem> (3-point Arun), for the reference's 3d3d KAT
#include "opengv_re.hpp"
// Tracker::findMatchingKeypoints / findMatchingStereoKeypoints / computeMedianDisparity and the voting
// loop's Mahalanobis distance on their own (pinned by tests/testTracker.cpp:1320-1533)
KVO_API int kvo_find_matching_keypoints(const int64_t* ref_lmk, int n_ref, const int64_t* cur_lmk, int n_cur,
                                        const uint8_t* ref_right_status, const uint8_t* cur_right_status,
                                        int32_t* out_pairs) {
  kimera::StereoFrame ref, cur;
  ref.left.landmarks.assign(ref_lmk, ref_lmk + n_ref);
  cur.left.landmarks.assign(cur_lmk, cur_lmk + n_cur);
  std::vector<kimera::KeypointMatch> m, ms;
  kimera::findMatchingKeypoints(ref.left, cur.left, m);
  if (ref_right_status && cur_right_status) {  // stereo variant: both right keypoints VALID
    ref.right_kp_rect.resize(n_ref);
    cur.right_kp_rect.resize(n_cur);
    for (int i = 0; i < n_ref; i++) ref.right_kp_rect[i].status = ref_right_status[i];
    for (int i = 0; i < n_cur; i++) cur.right_kp_rect[i].status = cur_right_status[i];
    kimera::findMatchingStereoKeypoints(ref, cur, m, ms);
    m.swap(ms);
  }
  for (size_t i = 0; i < m.size(); i++) {
    out_pairs[2 * i] = (int32_t)m[i].first;
    out_pairs[2 * i + 1] = (int32_t)m[i].second;
  }
  return (int)m.size();
}
KVO_API int kvo_compute_median_disparity(const float* ref_xy, const float* cur_xy, const int32_t* pairs, int n,
                                         double* median) {
  int hi = 0;
  for (int i = 0; i < 2 * n; i++) hi = std::max(hi, pairs[i] + 1);
  std::vector<Point2f> r(hi), c(hi);
  for (int i = 0; i < hi; i++) {
    r[i] = Point2f{ref_xy[2 * i], ref_xy[2 * i + 1]};
    c[i] = Point2f{cur_xy[2 * i], cur_xy[2 * i + 1]};
  }
  std::vector<kimera::KeypointMatch> m(n);
  for (int i = 0; i < n; i++) m[i] = std::make_pair((size_t)pairs[2 * i], (size_t)pairs[2 * i + 1]);
  return kimera::computeMedianDisparity(r, c, m, median) ? 1 : 0;
}
KVO_API int kvo_get_smart_stereo_measurements(const int64_t* lmk, const float* left_rect_xy,
                                              const uint8_t* right_status, const float* right_rect_xy, int n,
                                              int use_stereo_tracking, int64_t* out_lmk, double* out_uLuRv) {
  kimera::StereoFrame sf;
  sf.left.landmarks.assign(lmk, lmk + n);
  sf.left_kp_rect.resize(n);
  sf.right_kp_rect.resize(n);
  for (int i = 0; i < n; i++) {
    sf.left_kp_rect[i].status = KVFE_KP_VALID;
    sf.left_kp_rect[i].kp = Point2f{left_rect_xy[2 * i], left_rect_xy[2 * i + 1]};
    sf.right_kp_rect[i].status = right_status[i];
    sf.right_kp_rect[i].kp = Point2f{right_rect_xy[2 * i], right_rect_xy[2 * i + 1]};
  }
  std::vector<int64_t> ml;
  std::vector<double> mv;
  kimera::smartStereoMeasurements(sf, use_stereo_tracking != 0, ml, mv);
  std::copy(ml.begin(), ml.end(), out_lmk);
  std::copy(mv.begin(), mv.end(), out_uLuRv);
  return (int)ml.size();
}
// StereoMatcher::getDepthFromRectifiedMatches on its own (pinned by tests/testStereoMatcher.cpp:394-502);
// statuses are updated in place as the reference does (NO_DEPTH)
KVO_API void kvo_get_depth_from_rectified_matches(const kvo_camera* c, const kvfe_stereo_params* p, int n,
                                                  const float* left_xy, uint8_t* left_status,
                                                  const float* right_xy, uint8_t* right_status, double* depth) {
  std::vector<StatusKeypoint> l(n), r(n);
  for (int i = 0; i < n; i++) {
    l[i].status = left_status[i];
    l[i].kp = Point2f{left_xy[2 * i], left_xy[2 * i + 1]};
    r[i].status = right_status[i];
    r[i].kp = Point2f{right_xy[2 * i], right_xy[2 * i + 1]};
  }
  std::vector<double> d;
  kimera::getDepthFromRectifiedMatches(l, r, c->cam.fx(), c->cam.baseline(), *p, d);
  for (int i = 0; i < n; i++) {
    depth[i] = d[i];
    left_status[i] = l[i].status;
    right_status[i] = r[i].status;
  }
}
KVO_API int kvo_crop_to_size(float* xy, int w, int h, int round_first) {
  Point2f p{xy[0], xy[1]};
  const bool c = round_first ? kimera::roundAndCropToSize(&p, w, h) : kimera::cropToSize(&p, w, h);
  xy[0] = p.x;
  xy[1] = p.y;
  return c ? 1 : 0;
}
KVO_API float kvo_mahalanobis_f(const float* vi, const float* Ci, const float* vj, const float* Cj) {
  return kimera::mahalanobis_f(vi, Ci, vj, Cj);
}

KVO_API int kvo_ransac_point_cloud(const double* p1, const double* p2, int n, double threshold,
                                   int max_iterations, double probability, int rng_policy,
                                   int32_t* inliers, double* pose, int* iterations) {
  opengv_re::RansacResult r = opengv_re::ransac_point_cloud(p1, p2, n, threshold, max_iterations,
                                                            probability, rng_policy);
  if (iterations) *iterations = r.iterations;
  if (!r.success) return -1;
  std::memcpy(pose, r.coeff, sizeof(double) * 12);
  for (size_t i = 0; i < r.inliers.size(); i++) inliers[i] = r.inliers[i];
  return (int)r.inliers.size();
}
KVO_API void kvo_mt19937_draws(int policy, int n, int32_t* out) {  // SampleConsensusProblem::rnd()
  opengv_re::Mt19937 e;
  e.seed(12345u);
  for (int i = 0; i < n; i++) {
    uint32_t r = e.next();
    if (policy == opengv_re::RNG_LIBSTDCXX_11) {
      out[i] = (int32_t)(r >> 1);
    } else {
      while (r >= 0x80000000u) r = e.next();
      out[i] = (int32_t)r;
    }
  }
}

// ---- UndistorterRectifier / StereoCamera keypoint methods on their own (SURVEY.md 8b component calls) ----
KVO_API void kvo_camera_check_undistorted_rectified(const kvo_camera* c, int cam, const float* dist_xy,
                                                    const float* und_xy, int n, float pixel_tol, float* out_xy,
                                                    uint8_t* out_status) {
  std::vector<Point2f> d((const Point2f*)dist_xy, (const Point2f*)dist_xy + n);
  std::vector<Point2f> u((const Point2f*)und_xy, (const Point2f*)und_xy + n);
  std::vector<StatusKeypoint> out;
  c->cam.checkUndistortedRectifiedKeypoints(cam, d, u, pixel_tol, out);
  for (int i = 0; i < n; i++) {
    out_xy[2 * i] = out[i].kp.x;
    out_xy[2 * i + 1] = out[i].kp.y;
    out_status[i] = out[i].status;
  }
}
KVO_API void kvo_camera_distort_unrectify(const kvo_camera* c, int cam, const float* rect_xy, const uint8_t* status,
                                          int n, float* out_xy) {
  std::vector<StatusKeypoint> r(n);
  for (int i = 0; i < n; i++) r[i] = StatusKeypoint{status[i], {rect_xy[2 * i], rect_xy[2 * i + 1]}};
  std::vector<Point2f> out;
  c->cam.distortUnrectifyKeypoints(cam, r, out);
  for (int i = 0; i < n; i++) {
    out_xy[2 * i] = out[i].x;
    out_xy[2 * i + 1] = out[i].y;
  }
}

// ---- FeatureDetector::featureDetection(Frame*, R) and Tracker::featureTracking as component calls ----
static void frame_in(kimera::Frame& f, const uint8_t* img, int w, int h, size_t stride, int n, const float* kps,
                     const int64_t* lmk, const int32_t* age, const double* versors) {
  f.w = w;
  f.h = h;
  f.img.resize((size_t)w * h);
  for (int y = 0; y < h; y++) std::memcpy(&f.img[(size_t)y * w], img + (size_t)y * stride, w);
  for (int i = 0; i < n; i++) {
    f.keypoints.push_back(Point2f{kps[2 * i], kps[2 * i + 1]});
    f.landmarks.push_back(lmk[i]);
    f.landmarks_age.push_back(age[i]);
    for (int c = 0; c < 3; c++) f.versors.push_back(versors ? versors[3 * i + c] : 0.0);
  }
}
static int frame_out(const kimera::Frame& f, int capacity, float* kps, int64_t* lmk, int32_t* age, double* versors) {
  const int n = (int)f.keypoints.size(), m = std::min(n, capacity);
  for (int i = 0; i < m; i++) {
    kps[2 * i] = f.keypoints[i].x;
    kps[2 * i + 1] = f.keypoints[i].y;
    lmk[i] = f.landmarks[i];
    age[i] = f.landmarks_age[i];
    for (int c = 0; c < 3; c++) versors[3 * i + c] = f.versors[3 * i + c];
  }
  return n;
}
// returns the new keypoint count; *lmk_counter is FeatureDetector.cpp:141's function-static landmark id
KVO_API int kvo_feature_detection_frame(const kvfe_camera_params* l, const kvfe_camera_params* r,
                                        const kvfe_frontend_params* p, int mono, const uint8_t* img, size_t stride,
                                        int capacity, int n, float* kps, int64_t* lmk, int32_t* age, double* versors,
                                        int64_t* lmk_counter) {
  kimera::Frontend fe;
  fe.init(*l, *r, *p, mono != 0);
  fe.lmk_id = *lmk_counter;
  kimera::Frame f;
  frame_in(f, img, l->width, l->height, stride, n, kps, lmk, age, versors);
  fe.featureDetectionFrame(f, nullptr);
  *lmk_counter = fe.lmk_id;
  return frame_out(f, capacity, kps, lmk, age, versors);
}
// ref landmarks are updated in place (-1 where the track was lost or too old); returns the size of cur
KVO_API int kvo_feature_tracking_frame(const kvfe_camera_params* l, const kvfe_camera_params* r,
                                       const kvfe_frontend_params* p, int mono, const uint8_t* ref_img,
                                       const uint8_t* cur_img, size_t stride, int n_ref, const float* ref_kps,
                                       int64_t* ref_lmk, const int32_t* ref_age, const double ref_R_cur[9],
                                       int capacity, float* cur_kps, int64_t* cur_lmk, int32_t* cur_age,
                                       double* cur_versors) {
  kimera::Frontend fe;
  fe.init(*l, *r, *p, mono != 0);
  kimera::Frame ref, cur;
  frame_in(ref, ref_img, l->width, l->height, stride, n_ref, ref_kps, ref_lmk, ref_age, nullptr);
  frame_in(cur, cur_img, l->width, l->height, stride, 0, nullptr, nullptr, nullptr, nullptr);
  fe.featureTracking(ref, cur, ref_R_cur);
  for (int i = 0; i < n_ref; i++) ref_lmk[i] = ref.landmarks[i];
  return frame_out(cur, capacity, cur_kps, cur_lmk, cur_age, cur_versors);
}

// ---- RGBD components on their own (pinned by tests/testRgbdFrame.cpp:85-165, tests/testDepthFrame.cpp:56-171) ----
KVO_API void kvo_depth_detection_mask(const kvfe_depth_params* dp, const void* depth, int w, int h, size_t stride,
                                      uint8_t* mask) {
  std::vector<uint8_t> m;
  kimera::depthDetectionMask(*dp, depth, w, h, stride, m);
  std::memcpy(mask, m.data(), m.size());
}
KVO_API float kvo_depth_at_point(const kvfe_depth_params* dp, const void* depth, int w, int h, size_t stride, float x,
                                 float y) {
  return kimera::depthAtPoint(*dp, depth, w, h, stride, kimera::Point2f{x, y});
}
// RgbdFrame::fillStereoFrame with caller-provided left keypoints / rectified keypoints / statuses / versors
KVO_API void kvo_rgbd_fill_stereo_frame(const kvfe_camera_params* cam, const kvfe_frontend_params* p,
                                        const kvfe_depth_params* dp, const void* depth, int dw, int dh, size_t stride,
                                        int n, const float* left_xy, const float* left_rect_xy,
                                        const uint8_t* left_status, const double* versors, uint8_t* right_status,
                                        float* right_rect_xy, double* depths, double* kp3d, float* right_xy) {
  kimera::Frontend fe;
  fe.initRgbd(*cam, *p, *dp);
  kimera::StereoFrame sf;
  for (int i = 0; i < n; i++) {
    sf.left.keypoints.push_back(kimera::Point2f{left_xy[2 * i], left_xy[2 * i + 1]});
    sf.left_kp_rect.push_back(kimera::StatusKeypoint{left_status[i], {left_rect_xy[2 * i], left_rect_xy[2 * i + 1]}});
    for (int c = 0; c < 3; c++) sf.left.versors.push_back(versors[3 * i + c]);
  }
  fe.fillStereoFrame(sf, depth, stride, dw, dh);
  for (int i = 0; i < n; i++) {
    right_status[i] = (uint8_t)sf.right_kp_rect[i].status;
    right_rect_xy[2 * i] = sf.right_kp_rect[i].kp.x;
    right_rect_xy[2 * i + 1] = sf.right_kp_rect[i].kp.y;
    depths[i] = sf.depth[i];
    for (int c = 0; c < 3; c++) kp3d[3 * i + c] = sf.kp3d[3 * i + c];
    right_xy[2 * i] = sf.right_kp[i].x;
    right_xy[2 * i + 1] = sf.right_kp[i].y;
  }
}

// ---- front-end ---------------------------------------------------------------
struct kvo_frontend {
  kimera::Frontend fe;
};
KVO_API kvo_frontend* kvo_frontend_create(const kvfe_camera_params* l, const kvfe_camera_params* r,
                                          const kvfe_frontend_params* p) {
  kvo_frontend* f = new kvo_frontend;
  f->fe.init(*l, *r, *p);
  return f;
}
KVO_API kvo_frontend* kvo_frontend_create_mono(const kvfe_camera_params* cam,
                                               const kvfe_frontend_params* p) {
  kvo_frontend* f = new kvo_frontend;
  f->fe.init(*cam, *cam, *p, true);
  return f;
}
KVO_API kvo_frontend* kvo_frontend_create_rgbd(const kvfe_camera_params* cam, const kvfe_frontend_params* p,
                                               const kvfe_depth_params* dp) {
  kvo_frontend* f = new kvo_frontend;
  f->fe.initRgbd(*cam, *p, *dp);
  return f;
}
KVO_API void kvo_frontend_destroy(kvo_frontend* f) { delete f; }
KVO_API void kvo_frontend_process(kvo_frontend* f, const uint8_t* left, const uint8_t* right,
                                  size_t stride, const kvfe_frame_input* in) {
  f->fe.process(left, right, stride, *in);
}
KVO_API void kvo_frontend_update_map(kvo_frontend* f, const int64_t* ids, const double* xyz, int n) {
  f->fe.updateMap(ids, xyz, n);
}
KVO_API int kvo_frontend_get_output(kvo_frontend* f, kvfe_frame_output* out) {
  const kimera::StereoFrame& sf = f->fe.current();
  const int n = (int)sf.left.keypoints.size();
  out->n_keypoints = n;
  out->is_keyframe = f->fe.last_is_keyframe ? 1 : 0;
  out->n_tracked = sf.n_tracked;
  out->n_detected = sf.n_detected;
  out->n_measurements = (int)f->fe.meas_lmk.size();
  out->frame_id = sf.left.id;
  {
    const kimera::TrackerStatusSummary& T = f->fe.tracker_status;
    out->tracking_status_mono = T.mono;
    out->tracking_status_stereo = T.stereo;
    std::memcpy(out->lkf_T_k_mono, T.lkf_T_k_mono, sizeof(out->lkf_T_k_mono));
    std::memcpy(out->lkf_T_k_stereo, T.lkf_T_k_stereo, sizeof(out->lkf_T_k_stereo));
    std::memcpy(out->info_mat_stereo_translation, T.info, sizeof(T.info));
    out->nr_mono_putatives = T.nr_mono_putatives;
    out->nr_mono_inliers = T.nr_mono_inliers;
    out->mono_ransac_iters = T.mono_iters;
    out->nr_stereo_putatives = T.nr_stereo_putatives;
    out->nr_stereo_inliers = T.nr_stereo_inliers;
    out->reserved0 = 0;
    out->tracking_status_pnp = T.pnp;
    out->nr_pnp_inliers = T.nr_pnp_inliers;
    std::memcpy(out->W_T_k_pnp, T.W_T_k_pnp, sizeof(out->W_T_k_pnp));
  }
  const int m = std::min(n, out->capacity);
  const bool has_stereo = (int)sf.left_kp_rect.size() == n;
  for (int i = 0; i < m; i++) {
    if (out->landmarks) out->landmarks[i] = sf.left.landmarks[i];
    if (out->landmarks_age) out->landmarks_age[i] = sf.left.landmarks_age[i];
    if (out->keypoints) {
      out->keypoints[2 * i] = sf.left.keypoints[i].x;
      out->keypoints[2 * i + 1] = sf.left.keypoints[i].y;
    }
    if (out->versors)
      for (int k = 0; k < 3; k++) out->versors[3 * i + k] = sf.left.versors[3 * i + k];
    if (!has_stereo) continue;
    if (out->left_rect_xy) {
      out->left_rect_xy[2 * i] = sf.left_kp_rect[i].kp.x;
      out->left_rect_xy[2 * i + 1] = sf.left_kp_rect[i].kp.y;
    }
    if (out->left_status) out->left_status[i] = sf.left_kp_rect[i].status;
    if (out->right_rect_xy) {
      out->right_rect_xy[2 * i] = sf.right_kp_rect[i].kp.x;
      out->right_rect_xy[2 * i + 1] = sf.right_kp_rect[i].kp.y;
    }
    if (out->right_status) out->right_status[i] = sf.right_kp_rect[i].status;
    if (out->depth) out->depth[i] = sf.depth[i];
    if (out->right_xy) {
      out->right_xy[2 * i] = sf.right_kp[i].x;
      out->right_xy[2 * i + 1] = sf.right_kp[i].y;
    }
    if (out->keypoints_3d)
      for (int k = 0; k < 3; k++) out->keypoints_3d[3 * i + k] = sf.kp3d[3 * i + k];
  }
  const int mm = std::min(out->n_measurements, out->capacity);
  for (int i = 0; i < mm; i++) {
    if (out->meas_landmark) out->meas_landmark[i] = f->fe.meas_lmk[i];
    if (out->meas_uL_uR_v)
      for (int k = 0; k < 3; k++) out->meas_uL_uR_v[3 * i + k] = f->fe.meas_uLuRv[3 * i + k];
  }
  return has_stereo ? 1 : 0;
}

// Timed replay for bench.py's cpu_baseline leg: runs `n_frames` stereo pairs of
// one stream through the oracle front-end and returns the elapsed seconds.
KVO_API double kvo_frontend_time_sequence(kvo_frontend* f, const uint8_t* left,
                                          const uint8_t* right, int w, int h, int n_frames,
                                          const kvfe_frame_input* inputs) {
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n_frames; i++)
    f->fe.process(left + (size_t)i * w * h, right + (size_t)i * w * h, w, inputs[i]);
  auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double>(t1 - t0).count();
}
