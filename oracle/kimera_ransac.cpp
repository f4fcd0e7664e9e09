// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
// CPU restatement of Kimera-VIO's geometric outlier rejection (the reference-owned part; the
// OpenGV RANSAC it drives is restated in opengv_re.cpp):
//   Tracker::findMatchingKeypoints / findMatchingStereoKeypoints / computeMedianDisparity
//   Tracker::geometricOutlierRejection2d2d (given rotation)      src/frontend/Tracker.cpp:213-378
//   Tracker::geometricOutlierRejection3d3dGivenRotation (voting) src/frontend/Tracker.cpp:382-661
//   Tracker::getPoint3AndCovariance                              src/frontend/Tracker.cpp:772-818
//   Tracker::removeOutliersMono / removeOutliersStereo           src/frontend/Tracker.cpp:856-917
//   VisionImuFrontend::outlierRejectionMono / outlierRejectionStereo
//                                                                src/frontend/VisionImuFrontend.cpp:90-144
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>

#include "kimera.hpp"
#include "opengv_re.hpp"

namespace kimera {

// gtsam::Rot3::equals(Rot3(), tol) = equal_with_abs_tol(matrix, I, tol) with gtsam::fpEqual
// (gtsam/base/Matrix.h, gtsam/base/Vector.cpp, check_relative_also = false)
static bool fp_equal(double a, double b, double tol) {
  const double DOUBLE_MIN_NORMAL = std::numeric_limits<double>::min() + 1.0;
  const double larger = (std::abs(b) > std::abs(a)) ? std::abs(b) : std::abs(a);
  (void)larger;
  if (std::isnan(a) || std::isnan(b)) return std::isnan(a) && std::isnan(b);
  if (std::isinf(a) || std::isinf(b)) return a == b;
  if (a == b) return true;
  if (a == 0 || b == 0 || (std::abs(a) + std::abs(b)) < DOUBLE_MIN_NORMAL)
    return std::abs(a - b) <= tol * DOUBLE_MIN_NORMAL;
  if (std::abs(a - b) <= tol) return true;
  return false;
}
bool rot_equals_identity(const double R[9], double tol) {
  for (int i = 0; i < 9; i++)
    if (!fp_equal(R[i], (i % 4 == 0) ? 1.0 : 0.0, tol)) return false;
  return true;
}

// Tracker::findMatchingKeypoints (Tracker.cpp:919-946)
void findMatchingKeypoints(const Frame& ref, const Frame& cur, std::vector<KeypointMatch>& out) {
  out.clear();
  std::map<int64_t, size_t> ref_lm_index_map;
  for (size_t i = 0; i < ref.landmarks.size(); ++i)
    if (ref.landmarks[i] != -1) ref_lm_index_map[ref.landmarks[i]] = i;
  for (size_t i = 0; i < cur.landmarks.size(); ++i) {
    const int64_t id = cur.landmarks[i];
    if (id == -1) continue;
    auto it = ref_lm_index_map.find(id);
    if (it != ref_lm_index_map.end()) out.push_back(std::make_pair(it->second, i));
  }
}

// Tracker::findMatchingStereoKeypoints (Tracker.cpp:962-989)
void findMatchingStereoKeypoints(const StereoFrame& ref, const StereoFrame& cur,
                                 const std::vector<KeypointMatch>& mono,
                                 std::vector<KeypointMatch>& out) {
  out.clear();
  for (const KeypointMatch& m : mono)
    if (ref.right_kp_rect[m.first].status == KVFE_KP_VALID &&
        cur.right_kp_rect[m.second].status == KVFE_KP_VALID)
      out.push_back(m);
}

// Tracker::computeMedianDisparity (Tracker.cpp:991-1018)
bool computeMedianDisparity(const std::vector<Point2f>& ref, const std::vector<Point2f>& cur,
                            const std::vector<KeypointMatch>& matches, double* median) {
  std::vector<double> disparity_sq;
  disparity_sq.reserve(matches.size());
  for (const KeypointMatch& rc : matches) {
    const float dx = cur[rc.second].x - ref[rc.first].x;
    const float dy = cur[rc.second].y - ref[rc.first].y;
    const double px_dist = dx * dx + dy * dy;  // float arithmetic, then widened
    disparity_sq.push_back(px_dist);
  }
  if (disparity_sq.empty()) {
    *median = 0.0;
    return false;
  }
  const size_t center = disparity_sq.size() / 2;
  std::nth_element(disparity_sq.begin(), disparity_sq.begin() + center, disparity_sq.end());
  *median = std::sqrt(disparity_sq[center]);
  return true;
}

// Tracker::geometricOutlierRejection2d2d (Tracker.cpp:213-318) + runRansac (Tracker.h:247-296)
RansacOut outlierRejection2d2dGivenRot(const double* f_ref, const double* f_cur, int n,
                                       const double R[9], const kvfe_tracker_params& tp) {
  RansacOut o;
  opengv_re::RansacResult r = opengv_re::ransac_translation_only(
      f_ref, f_cur, n, R, tp.ransac_threshold_mono, tp.ransac_max_iterations,
      tp.ransac_probability, tp.ransac_rng_policy);
  bool success = r.success;
  o.iterations = r.iterations;
  if (success && r.iterations >= tp.ransac_max_iterations && r.inliers.empty()) success = false;
  if (!success) {
    o.status = KVFE_TRACKING_INVALID;
    return o;  // identity pose, no inliers
  }
  o.inliers = r.inliers;
  std::memcpy(o.pose, r.coeff, sizeof(o.pose));
  o.status = KVFE_TRACKING_VALID;
  if ((int)o.inliers.size() < tp.min_nr_mono_inliers) o.status = KVFE_TRACKING_FEW_MATCHES;
  return o;
}

// Tracker::geometricOutlierRejection2d2d, 5-point branch (Tracker.cpp:262-275) -> runRansac<Problem2d2d(NISTER)>
RansacOut outlierRejection2d2d(const double* f_ref, const double* f_cur, int n, const kvfe_tracker_params& tp) {
  RansacOut o;
  opengv_re::RansacResult r = opengv_re::ransac_central_relative_pose_nister(
      f_ref, f_cur, n, tp.ransac_threshold_mono, tp.ransac_max_iterations, tp.ransac_probability,
      tp.ransac_rng_policy);
  bool success = r.success;
  o.iterations = r.iterations;
  if (success && r.iterations >= tp.ransac_max_iterations && r.inliers.empty()) success = false;
  if (!success) {
    o.status = KVFE_TRACKING_INVALID;
    return o;
  }
  o.inliers = r.inliers;
  std::memcpy(o.pose, r.coeff, sizeof(o.pose));
  o.status = KVFE_TRACKING_VALID;
  if ((int)o.inliers.size() < tp.min_nr_mono_inliers) o.status = KVFE_TRACKING_FEW_MATCHES;
  return o;
}

// Tracker::geometricOutlierRejection3d3d (Tracker.cpp:667-742) -> runRansac<Problem3d3d>
// (Tracker.h:247-296; optimize_3d3d_pose_from_inliers_ = false)
RansacOut outlierRejection3d3d(const double* ref_p3, const double* cur_p3, int n,
                               const kvfe_tracker_params& tp) {
  RansacOut o;
  opengv_re::RansacResult r = opengv_re::ransac_point_cloud(
      ref_p3, cur_p3, n, tp.ransac_threshold_stereo, tp.ransac_max_iterations, tp.ransac_probability,
      tp.ransac_rng_policy);
  bool success = r.success;
  o.iterations = r.iterations;
  if (success && r.iterations >= tp.ransac_max_iterations && r.inliers.empty()) success = false;
  if (!success) {
    o.status = KVFE_TRACKING_INVALID;
    return o;
  }
  o.inliers = r.inliers;
  std::memcpy(o.pose, r.coeff, sizeof(o.pose));
  o.status = KVFE_TRACKING_VALID;
  if ((int)o.inliers.size() < tp.min_nr_stereo_inliers) o.status = KVFE_TRACKING_FEW_MATCHES;
  return o;
}

// Tracker::pnp (Tracker.cpp:1122-1288) for pnp_algorithm_ = EPNP / KneipP3P -> runRansac<ProblemPnP> (Tracker.h:247-296,
// optimize_2d3d_pose_from_inliers_ = false), then VisionImuFrontend::outlierRejectionPnP's status
// (VisionImuFrontend.cpp:146-173)
RansacOut pnp(const double* bearings, const double* points, int n, double avg_focal_length,
              const kvfe_tracker_params& tp, const kvfe_pnp_params& pp, bool* success_out) {
  RansacOut o;
  bool success = false;
  if (n == 0) {  // "No 2D-3D correspondences found for 2D-3D RANSAC...": Pose3(), no inliers
    success = false;
  } else {
    const double reprojection_error = pp.ransac_threshold_pnp;
    const double threshold = 1.0 - std::cos(std::atan(std::sqrt(2.0) * reprojection_error / avg_focal_length));
    opengv_re::RansacResult r =
        pp.pnp_algorithm == 1   // Pose3d2dAlgorithm::KneipP3P
            ? opengv_re::ransac_absolute_pose_kneip(bearings, points, n, threshold, tp.ransac_max_iterations,
                                                    tp.ransac_probability, tp.ransac_rng_policy)
            : opengv_re::ransac_absolute_pose_epnp(bearings, points, n, threshold, tp.ransac_max_iterations,
                                                   tp.ransac_probability, tp.ransac_rng_policy);
    success = r.success;
    o.iterations = r.iterations;
    if (success && r.iterations >= tp.ransac_max_iterations && r.inliers.empty()) success = false;
    if (success) {
      o.inliers = r.inliers;
      std::memcpy(o.pose, r.coeff, sizeof(o.pose));
    }
  }
  if (success_out) *success_out = success;
  o.status = (success && (int)o.inliers.size() > pp.min_pnp_inliers) ? KVFE_TRACKING_VALID : KVFE_TRACKING_FEW_MATCHES;
  return o;
}

void Frontend::updateMap(const int64_t* ids, const double* xyz, int n) {
  landmarks_map.clear();
  for (int i = 0; i < n; i++) landmarks_map[ids[i]] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
}

// VisionImuFrontend::outlierRejectionPnP(frame, &status_pnp) with Tracker::pnp(const StereoFrame&) (Tracker.cpp:1064-1120)
void Frontend::outlierRejectionPnP(const StereoFrame& frame) {
  std::vector<double> cam_bearing_vectors, W_points;
  for (size_t i = 0; i < frame.left_kp_rect.size(); i++) {
    const int64_t lmk_id = frame.left.landmarks.at(i);
    if (frame.left_kp_rect[i].status == KVFE_KP_VALID && lmk_id != -1) {
      const auto it = landmarks_map.find(lmk_id);
      if (it != landmarks_map.end()) {
        W_points.insert(W_points.end(), it->second.begin(), it->second.end());
        cam_bearing_vectors.insert(cam_bearing_vectors.end(), frame.kp3d.begin() + 3 * i, frame.kp3d.begin() + 3 * i + 3);
      }
    }
  }
  bool success = false;
  const double focal = 0.5 * (cam.left.intrinsics[0] + cam.left.intrinsics[1]);
  const RansacOut r = pnp(cam_bearing_vectors.data(), W_points.data(), (int)(W_points.size() / 3), focal, p.tracker,
                          p.pnp, &success);
  tracker_status.pnp = r.status;
  std::memcpy(tracker_status.W_T_k_pnp, r.pose, sizeof(r.pose));
  tracker_status.nr_pnp_inliers = (int)r.inliers.size();
}

// gtsam::StereoCamera(Pose3(), K).backproject2(z, boost::none, H2) (gtsam 4.2
// geometry/StereoCamera.cpp) and Tracker::getPoint3AndCovariance (Tracker.cpp:772-818)
void getPoint3AndCovariance(const StereoCalib& K, double uL, double uR, double v, const double p3[3],
                            const double* Rmat, double point[3], double cov[9]) {
  const double disparity = uL - uR;
  const double local_z = K.b * K.fx / disparity;
  const double lx = local_z * (uL - K.cx) / K.fx, ly = local_z * (v - K.cy) / K.fy;
  const double z_partial_uR = local_z / disparity;
  const double x_partial_uR = lx / disparity;
  const double y_partial_uR = ly / disparity;
  double J[9] = {-x_partial_uR + local_z / K.fx, x_partial_uR, 0,
                 -y_partial_uR,                  y_partial_uR, local_z / K.fy,
                 -z_partial_uR,                  z_partial_uR, 0};
  // (the 3-D point itself is keypoints_3d_, not the gtsam back-projection)
  for (int i = 0; i < 3; i++) point[i] = p3[i];
  if (Rmat) {
    double q[3], RJ[9];
    for (int r = 0; r < 3; r++)
      q[r] = (Rmat[r * 3] * p3[0] + Rmat[r * 3 + 1] * p3[1]) + Rmat[r * 3 + 2] * p3[2];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++)
        RJ[r * 3 + c] = (Rmat[r * 3] * J[c] + Rmat[r * 3 + 1] * J[3 + c]) + Rmat[r * 3 + 2] * J[6 + c];
    std::memcpy(point, q, sizeof(q));
    std::memcpy(J, RJ, sizeof(RJ));
  }
  // cov = J * I * J^T
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++)
      cov[r * 3 + c] = (J[r * 3] * J[c * 3] + J[r * 3 + 1] * J[c * 3 + 1]) + J[r * 3 + 2] * J[c * 3 + 2];
}

// Eigen::Matrix3d::inverse() (Eigen/src/LU/InverseImpl.h, compute_inverse<.,.,3>)
static void eigen_inverse3(const double* m, double* r) {
  auto M = [&](int i, int j) { return m[i * 3 + j]; };
  auto cof = [&](int i, int j) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return M(i1, j1) * M(i2, j2) - M(i1, j2) * M(i2, j1);
  };
  const double c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
  const double det = (c00 * M(0, 0) + c10 * M(1, 0)) + c20 * M(2, 0);
  const double invdet = 1.0 / det;
  r[1 * 3 + 0] = cof(0, 1) * invdet;
  r[1 * 3 + 1] = cof(1, 1) * invdet;
  r[2 * 3 + 0] = cof(0, 2) * invdet;
  r[1 * 3 + 2] = cof(2, 1) * invdet;
  r[2 * 3 + 1] = cof(1, 2) * invdet;
  r[2 * 3 + 2] = cof(2, 2) * invdet;
  r[0] = c00 * invdet;
  r[1] = c10 * invdet;
  r[2] = c20 * invdet;
}

// the float32 Mahalanobis test of the voting loop, literally (Tracker.cpp:499-523)
float mahalanobis_f(const float* vi, const float* Ci, const float* vj, const float* Cj) {
  float v[3], O[9];
  for (int k = 0; k < 3; k++) v[k] = vi[k] - vj[k];
  for (int k = 0; k < 9; k++) O[k] = Ci[k] + Cj[k];
#define O_(r, c) O[(r) * 3 + (c)]
  const float dinv = 1 / (O_(0, 0) * (O_(1, 1) * O_(2, 2) - O_(1, 2) * O_(2, 1)) -
                          O_(1, 0) * (O_(0, 1) * O_(2, 2) - O_(0, 2) * O_(2, 1)) +
                          O_(2, 0) * (O_(0, 1) * O_(1, 2) - O_(1, 1) * O_(0, 2)));
  const float d =
      dinv * v[0] *
          (v[0] * (O_(1, 1) * O_(2, 2) - O_(1, 2) * O_(2, 1)) -
           v[1] * (O_(0, 1) * O_(2, 2) - O_(0, 2) * O_(2, 1)) +
           v[2] * (O_(0, 1) * O_(1, 2) - O_(1, 1) * O_(0, 2))) +
      dinv * v[1] *
          (O_(0, 0) * (v[1] * O_(2, 2) - O_(1, 2) * v[2]) -
           O_(1, 0) * (v[0] * O_(2, 2) - O_(0, 2) * v[2]) +
           O_(2, 0) * (v[0] * O_(1, 2) - v[1] * O_(0, 2))) +
      dinv * v[2] *
          (O_(0, 0) * (O_(1, 1) * v[2] - v[1] * O_(2, 1)) -
           O_(1, 0) * (O_(0, 1) * v[2] - v[0] * O_(2, 1)) +
           O_(2, 0) * (O_(0, 1) * v[1] - O_(1, 1) * v[0]));
#undef O_
  return d;
}

// Tracker::geometricOutlierRejection3d3dGivenRotation (Tracker.cpp:382-632)
RansacOut outlierRejection3d3dGivenRot(const float* ref_left_xy, const float* ref_right_x,
                                       const double* ref_p3, const float* cur_left_xy,
                                       const float* cur_right_x, const double* cur_p3, int n,
                                       const StereoCalib& K, const double R[9],
                                       const kvfe_tracker_params& tp) {
  RansacOut o;
  o.iterations = 1;
  std::vector<double> rel_tran((size_t)n * 3), cov_rel((size_t)n * 9);
  std::vector<float> rel_tranf((size_t)n * 3), cov_relf((size_t)n * 9);
  for (int i = 0; i < n; i++) {
    double f_ref[3], cov_ref[9], R_f_cur[3], cov_R_cur[9];
    getPoint3AndCovariance(K, (double)ref_left_xy[2 * i], (double)ref_right_x[i],
                           (double)ref_left_xy[2 * i + 1], ref_p3 + 3 * i, nullptr, f_ref, cov_ref);
    getPoint3AndCovariance(K, (double)cur_left_xy[2 * i], (double)cur_right_x[i],
                           (double)cur_left_xy[2 * i + 1], cur_p3 + 3 * i, R, R_f_cur, cov_R_cur);
    for (int k = 0; k < 3; k++) {
      rel_tran[3 * i + k] = f_ref[k] - R_f_cur[k];
      rel_tranf[3 * i + k] = (float)rel_tran[3 * i + k];
    }
    for (int k = 0; k < 9; k++) {
      cov_rel[9 * i + k] = cov_R_cur[k] + cov_ref[k];
      cov_relf[9 * i + k] = (float)cov_rel[9 * i + k];
    }
  }
  // voting
  std::vector<std::vector<int>> coherent_set(n);
  size_t maxCoherentSetSize = 0, maxCoherentSetId = 0;
  const float threshold = (float)tp.ransac_threshold_stereo;
  for (int i = 0; i < n; i++) {
    coherent_set[i].push_back(i);
    for (int j = i + 1; j < n; j++) {
      const float d = mahalanobis_f(&rel_tranf[3 * i], &cov_relf[9 * i], &rel_tranf[3 * j], &cov_relf[9 * j]);
      if (d < threshold) {
        coherent_set[i].push_back(j);
        coherent_set[j].push_back(i);
      }
    }
    if (coherent_set[i].size() > maxCoherentSetSize) {
      maxCoherentSetSize = coherent_set[i].size();
      maxCoherentSetId = i;
    }
  }
  if (maxCoherentSetSize < 2) {
    o.status = KVFE_TRACKING_INVALID;
    return o;  // Pose3(), zero information
  }
  o.inliers = coherent_set[maxCoherentSetId];
  std::sort(o.inliers.begin(), o.inliers.end());
  o.status = KVFE_TRACKING_VALID;
  if ((int)o.inliers.size() < tp.min_nr_stereo_inliers) o.status = KVFE_TRACKING_FEW_MATCHES;
  double t[3] = {0, 0, 0}, total_info[9] = {0};
  for (int id : o.inliers) {
    double info[9];
    eigen_inverse3(&cov_rel[9 * (size_t)id], info);
    const double* v = &rel_tran[3 * (size_t)id];
    for (int r = 0; r < 3; r++)
      t[r] = t[r] + ((info[r * 3] * v[0] + info[r * 3 + 1] * v[1]) + info[r * 3 + 2] * v[2]);
    for (int k = 0; k < 9; k++) total_info[k] = total_info[k] + info[k];
  }
  double inv_total[9], tt[3];
  eigen_inverse3(total_info, inv_total);
  for (int r = 0; r < 3; r++)
    tt[r] = (inv_total[r * 3] * t[0] + inv_total[r * 3 + 1] * t[1]) + inv_total[r * 3 + 2] * t[2];
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) o.pose[r * 4 + c] = R[r * 3 + c];
    o.pose[r * 4 + 3] = tt[r];
  }
  std::memcpy(o.info, total_info, sizeof(total_info));
  return o;
}

// Tracker::findOutliers (Tracker.cpp:820-838): complement of the (sorted) inlier list
static void findOutliers(size_t n_matches, std::vector<int> inliers, std::vector<int>& outliers) {
  outliers.clear();
  std::sort(inliers.begin(), inliers.end());
  size_t k = 0;
  for (size_t i = 0; i < n_matches; ++i) {
    if (k < inliers.size() && (int)i > inliers[k]) ++k;
    if (k >= inliers.size() || (int)i != inliers[k]) outliers.push_back((int)i);
  }
}

// VisionImuFrontend::outlierRejectionMono (VisionImuFrontend.cpp:90-113) ->
// Tracker::geometricOutlierRejection2d2d(Frame*, Frame*, Pose3) (Tracker.cpp:322-378)
void Frontend::outlierRejectionMono(const double R[9], Frame& ref, Frame& cur) {
  TrackerStatusSummary& S = tracker_status;
  const bool imu_ok = !rot_equals_identity(R, 1e-9);  // time alignment is outside the hot path
  RansacOut result;
  std::vector<KeypointMatch> matches;
  findMatchingKeypoints(ref, cur, matches);
  if (matches.empty()) {
    S.mono = KVFE_TRACKING_INVALID;
    return;
  }
  std::vector<double> f_ref(matches.size() * 3), f_cur(matches.size() * 3);
  for (size_t m = 0; m < matches.size(); m++)
    for (int c = 0; c < 3; c++) {
      f_ref[3 * m + c] = ref.versors[3 * matches[m].first + c];
      f_cur[3 * m + c] = cur.versors[3 * matches[m].second + c];
    }
  // Tracker::geometricOutlierRejection2d2d(bearings, ...) branches on the PARAMETER only (Tracker.cpp:247-275):
  // with ransac_use_2point_mono the 2-point problem runs even without gyro rotation (outlierRejectionMono then
  // passes Pose3(), i.e. R = I, VisionImuFrontend.cpp:108-111); without it the 5-point problem (2d2d_algorithm).
  (void)imu_ok;
  if (p.tracker.ransac_use_2point_mono)
    result = outlierRejection2d2dGivenRot(f_ref.data(), f_cur.data(), (int)matches.size(), R, p.tracker);
  else
    result = outlierRejection2d2d(f_ref.data(), f_cur.data(), (int)matches.size(), p.tracker);
  if (result.status != KVFE_TRACKING_INVALID) {  // debug info is filled on RANSAC success only
    S.nr_mono_putatives = (int)matches.size();
    S.nr_mono_inliers = (int)result.inliers.size();
    S.mono_iters = 0;
  }
  if (result.status != KVFE_TRACKING_FEW_MATCHES) {  // removeOutliersMono (Tracker.cpp:856-884)
    std::vector<int> outliers;
    findOutliers(matches.size(), result.inliers, outliers);
    for (int out : outliers) {
      ref.landmarks[matches[out].first] = -1;
      cur.landmarks[matches[out].second] = -1;
    }
    std::vector<KeypointMatch> kept;
    for (int in : result.inliers) kept.push_back(matches[in]);
    matches = kept;
  }
  if (result.status == KVFE_TRACKING_VALID) {
    double disparity;
    if (computeMedianDisparity(ref.keypoints, cur.keypoints, matches, &disparity))
      if (disparity < p.tracker.disparity_threshold) result.status = KVFE_TRACKING_LOW_DISPARITY;
  }
  S.mono = result.status;
  if (result.status == KVFE_TRACKING_VALID) std::memcpy(S.lkf_T_k_mono, result.pose, sizeof(result.pose));
}

// VisionImuFrontend::outlierRejectionStereo (VisionImuFrontend.cpp:115-144) ->
// Tracker::geometricOutlierRejection3d3dGivenRotation(StereoFrame&, ...) (Tracker.cpp:634-661)
void Frontend::outlierRejectionStereo(const double R[9], StereoFrame& ref, StereoFrame& cur) {
  TrackerStatusSummary& S = tracker_status;
  const bool imu_ok = !rot_equals_identity(R, 1e-9);
  if (!(p.tracker.ransac_use_1point_stereo && imu_ok)) {
    // 3-point RANSAC: Tracker::geometricOutlierRejection3d3d(frame_lkf, frame_k) (Tracker.cpp:744-769),
    // translation information zero (VisionImuFrontend.cpp:140-142)
    std::vector<KeypointMatch> mono3, m3;
    findMatchingKeypoints(ref.left, cur.left, mono3);
    findMatchingStereoKeypoints(ref, cur, mono3, m3);
    const int n3 = (int)m3.size();
    std::vector<double> rp3(3 * (size_t)n3), cp3(3 * (size_t)n3);
    for (int m = 0; m < n3; m++)
      for (int c = 0; c < 3; c++) {
        rp3[3 * m + c] = ref.kp3d[3 * m3[m].first + c];
        cp3[3 * m + c] = cur.kp3d[3 * m3[m].second + c];
      }
    RansacOut res3 = outlierRejection3d3d(rp3.data(), cp3.data(), n3, p.tracker);
    if (res3.status != KVFE_TRACKING_INVALID) {
      S.nr_stereo_putatives = n3;
      S.nr_stereo_inliers = (int)res3.inliers.size();
      std::vector<int> outliers3;   // removeOutliersStereo (Tracker.cpp:763-767)
      findOutliers(m3.size(), res3.inliers, outliers3);
      for (int out : outliers3) {
        const size_t a = m3[out].first, b = m3[out].second;
        ref.right_kp_rect[a].status = KVFE_KP_FAILED_ARUN;
        ref.depth[a] = 0.0;
        cur.right_kp_rect[b].status = KVFE_KP_FAILED_ARUN;
        cur.depth[b] = 0.0;
        for (int c = 0; c < 3; c++) ref.kp3d[3 * a + c] = cur.kp3d[3 * b + c] = 0.0;
      }
    }
    S.stereo = res3.status;
    std::memset(S.info, 0, sizeof(S.info));
    if (res3.status == KVFE_TRACKING_VALID) std::memcpy(S.lkf_T_k_stereo, res3.pose, sizeof(res3.pose));
    return;
  }
  std::vector<KeypointMatch> mono, matches;
  findMatchingKeypoints(ref.left, cur.left, mono);
  findMatchingStereoKeypoints(ref, cur, mono, matches);
  const int n = (int)matches.size();
  std::vector<float> rl(2 * n), rr(n), cl(2 * n), cr(n);
  std::vector<double> rp(3 * n), cp(3 * n);
  for (int m = 0; m < n; m++) {
    const size_t a = matches[m].first, b = matches[m].second;
    rl[2 * m] = ref.left_kp_rect[a].kp.x;
    rl[2 * m + 1] = ref.left_kp_rect[a].kp.y;
    rr[m] = ref.right_kp_rect[a].kp.x;
    cl[2 * m] = cur.left_kp_rect[b].kp.x;
    cl[2 * m + 1] = cur.left_kp_rect[b].kp.y;
    cr[m] = cur.right_kp_rect[b].kp.x;
    for (int c = 0; c < 3; c++) {
      rp[3 * m + c] = ref.kp3d[3 * a + c];
      cp[3 * m + c] = cur.kp3d[3 * b + c];
    }
  }
  StereoCalib K;
  K.fx = cam.rect.P1[0];
  K.fy = cam.rect.P1[5];
  K.s = cam.rect.P1[1];
  K.cx = cam.rect.P1[2];
  K.cy = cam.rect.P1[6];
  K.b = cam.rect.baseline;
  RansacOut result = outlierRejection3d3dGivenRot(rl.data(), rr.data(), rp.data(), cl.data(), cr.data(),
                                                  cp.data(), n, K, R, p.tracker);
  if (result.status != KVFE_TRACKING_INVALID) {
    S.nr_stereo_putatives = n;
    S.nr_stereo_inliers = (int)result.inliers.size();
  }
  // removeOutliersStereo (Tracker.cpp:886-917): FAILED_ARUN / zero depth / zero 3-D point in both frames
  std::vector<int> outliers;
  findOutliers(matches.size(), result.inliers, outliers);
  for (int out : outliers) {
    const size_t a = matches[out].first, b = matches[out].second;
    ref.right_kp_rect[a].status = KVFE_KP_FAILED_ARUN;
    ref.depth[a] = 0.0;
    cur.right_kp_rect[b].status = KVFE_KP_FAILED_ARUN;
    cur.depth[b] = 0.0;
    for (int c = 0; c < 3; c++) ref.kp3d[3 * a + c] = cur.kp3d[3 * b + c] = 0.0;
  }
  S.stereo = result.status;
  std::memcpy(S.info, result.info, sizeof(S.info));
  if (result.status == KVFE_TRACKING_VALID) std::memcpy(S.lkf_T_k_stereo, result.pose, sizeof(result.pose));
}

}  // namespace kimera
