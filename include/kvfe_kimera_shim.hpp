// kvfe_kimera_shim.hpp — the thin layer a Kimera-VIO build puts between its own types and kvfe_adapter.hpp, so that
// the call sites in src/frontend keep the reference's exact signatures:
//
//   void FeatureDetector::featureDetection(Frame* cur_frame, std::optional<cv::Mat> R)       FeatureDetector.h:39-41
//   std::vector<cv::KeyPoint> rawFeatureDetection(const cv::Mat& img, const cv::Mat& mask)    FeatureDetector.h:46-49
//   void Tracker::featureTracking(Frame* ref_frame, Frame* cur_frame, const gtsam::Rot3& ref_R_cur,
//                                 const FeatureDetectorParams&, std::optional<cv::Mat> R)     Tracker.h:70-74
//   void UndistorterRectifier::undistortRectifyImage(const cv::Mat& img, cv::Mat* out) const  UndistorterRectifier.h:76-78
//   void StereoMatcher::getDepthFromRectifiedMatches(StatusKeypointsCV&, StatusKeypointsCV&, Depths*) const
//                                                                                             StereoMatcher.h:85-92
//   void StereoCamera::undistortRectifyStereoFrame(StereoFrame*) const                        StereoCamera.h:225-233
//   void StereoMatcher::sparseStereoReconstruction(StereoFrame* stereo_frame)                StereoMatcher.h:49-52
//   void StereoMatcher::denseStereoReconstruction(const cv::Mat& left_img_rectified,
//                           const cv::Mat& right_img_rectified, cv::Mat* disparity_img)       StereoMatcher.h:54-66
//   FrontendOutputPacketBase::UniquePtr StereoVisionImuFrontend::spinOnce(
//                           FrontendInputPacketBase::UniquePtr&& input)                       VisionImuFrontend.h:63-64
//
// It needs <opencv2/core.hpp> (cv::Mat, cv::Point2f, cv::KeyPoint) and <gtsam/geometry/Rot3.h>; the Kimera types
// (VIO::Frame, VIO::StereoFrame, VIO::FeatureDetectorParams) are template parameters, so this header does not
// include Kimera's.  Layout facts used: cv::Point2f == {float x, y} == kvfe::KeypointCV; VIO::KeypointStatus has the
// values of kvfe::KeypointStatus (vio_types.h:38-44); LandmarkId = long (vio_types.h:55); landmarks_age_ is
// std::vector<size_t>; versors_ is std::vector<gtsam::Vector3> (vio_types.h:67-69).
//
// This container has neither OpenCV nor GTSAM: tests/cpp/shim_check.cpp compiles this header against minimal
// stand-ins (tests/cpp/stubs/) and runs it; INTEGRATION.md shows the two-line change at each reference call site.
#pragma once
#include <cstring>
#include <optional>
#include <stdexcept>
#include <utility>
#include <vector>

#include <gtsam/geometry/Rot3.h>
#include <opencv2/core.hpp>

#include "kvfe_adapter.hpp"

namespace kvfe {
namespace shim {

inline ImageView view(const cv::Mat& m) {
  if (m.empty() || m.type() != CV_8UC1) throw Error(KVFE_ERR_INVALID_ARG, "expected a non-empty CV_8UC1 image");
  return ImageView{m.data, m.rows, m.cols, (size_t)m.step};
}
inline void rot_to_array(const gtsam::Rot3& R, double out[9]) {
  const auto M = R.matrix();
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) out[3 * i + j] = M(i, j);
}

// VIO::Frame <-> kvfe::Frame (copies of the small per-keypoint arrays; the image is a view)
template <class VioFrame>
Frame to_frame(const VioFrame& f) {
  Frame o;
  o.img_ = view(f.img_);
  const size_t n = f.landmarks_.size();
  o.keypoints_.resize(n);
  o.landmarks_.resize(n);
  o.landmarks_age_.resize(n);
  o.versors_.resize(3 * n);
  for (size_t i = 0; i < n; i++) {
    o.keypoints_[i] = KeypointCV{f.keypoints_[i].x, f.keypoints_[i].y};
    o.landmarks_[i] = (int64_t)f.landmarks_[i];
    o.landmarks_age_[i] = (int32_t)f.landmarks_age_[i];
    if (i < f.versors_.size())
      for (int c = 0; c < 3; c++) o.versors_[3 * i + c] = f.versors_[i](c);
  }
  return o;
}
template <class VioFrame>
void from_frame(const Frame& o, VioFrame* f) {
  const size_t n = o.landmarks_.size();
  f->keypoints_.resize(n);
  f->landmarks_.resize(n);
  f->landmarks_age_.resize(n);
  f->scores_.resize(n, 0.0);   // "NOT IMPLEMENTED" upstream (FeatureDetector.cpp:147)
  f->versors_.resize(n);
  for (size_t i = 0; i < n; i++) {
    f->keypoints_[i].x = o.keypoints_[i].x;
    f->keypoints_[i].y = o.keypoints_[i].y;
    f->landmarks_[i] = (decltype(f->landmarks_[i] + 0))o.landmarks_[i];
    f->landmarks_age_[i] = (size_t)o.landmarks_age_[i];
    for (int c = 0; c < 3; c++) f->versors_[i](c) = o.versors_[3 * i + c];
  }
}
template <class StatusKps>
StatusKeypointsCV to_status(const StatusKps& v) {
  StatusKeypointsCV o(v.size());
  for (size_t i = 0; i < v.size(); i++)
    o[i] = {static_cast<KeypointStatus>(static_cast<uint8_t>(v[i].first)), KeypointCV{v[i].second.x, v[i].second.y}};
  return o;
}
template <class StatusKps>
void from_status(const StatusKeypointsCV& o, StatusKps* v) {
  v->resize(o.size());
  for (size_t i = 0; i < o.size(); i++) {
    (*v)[i].first = static_cast<decltype((*v)[i].first)>(static_cast<uint8_t>(o[i].first));
    (*v)[i].second.x = o[i].second.x;
    (*v)[i].second.y = o[i].second.y;
  }
}

// ---- the reference's classes with the reference's signatures -------------------------------------------------
class FeatureDetector {
 public:
  explicit FeatureDetector(Context ctx) : impl_(std::move(ctx)) {}
  // FeatureDetector.h:39-41.  R is accepted for signature compatibility: the context was created from the same
  // StereoCamera, so its R1 is the matrix the reference passes (StereoVisionImuFrontend.cpp:258,418).
  template <class VioFrame>
  void featureDetection(VioFrame* cur_frame, std::optional<cv::Mat> /*R*/ = std::nullopt) {
    Frame f = to_frame(*cur_frame);
    impl_.featureDetection(&f);
    from_frame(f, cur_frame);
  }
  // FeatureDetector.h:46-49
  std::vector<cv::KeyPoint> rawFeatureDetection(const cv::Mat& img, const cv::Mat& mask = cv::Mat()) {
    const ImageView iv = view(img);
    ImageView mv{nullptr, 0, 0, 0};
    if (!mask.empty()) mv = view(mask);
    const KeypointsCV k = impl_.rawFeatureDetection(iv, mask.empty() ? nullptr : &mv);
    std::vector<cv::KeyPoint> out;
    out.reserve(k.size());
    for (const KeypointCV& p : k) out.emplace_back(cv::Point2f(p.x, p.y), 3.0f /* blockSize */);
    return out;
  }

 private:
  kvfe::FeatureDetector impl_;
};

class Tracker {
 public:
  explicit Tracker(Context ctx) : impl_(std::move(ctx)) {}
  // Tracker.h:70-74
  template <class VioFrame, class FeatureDetectorParamsT>
  void featureTracking(VioFrame* ref_frame, VioFrame* cur_frame, const gtsam::Rot3& ref_R_cur,
                       const FeatureDetectorParamsT& /*feature_detector_params*/,
                       std::optional<cv::Mat> /*R*/ = std::nullopt) {
    Frame r = to_frame(*ref_frame), c = to_frame(*cur_frame);
    double R9[9];
    rot_to_array(ref_R_cur, R9);
    impl_.featureTracking(&r, &c, R9);
    for (size_t i = 0; i < r.landmarks_.size(); i++)   // only the landmark ids of the reference frame change
      ref_frame->landmarks_[i] = (decltype(ref_frame->landmarks_[i] + 0))r.landmarks_[i];
    from_frame(c, cur_frame);
  }

  // tracker_params_.pnp_algorithm_ / min_pnp_inliers_ / ransac_threshold_pnp_ / optimize_2d3d_pose_from_inliers_
  // (VisionImuTrackerParams.h:72-76); the remaining RANSAC settings are those of the context
  template <class TrackerParamsT>
  void setPnpParams(const TrackerParamsT& tp) {
    pnp_params_.pnp_algorithm = (int32_t)tp.pnp_algorithm_;
    pnp_params_.min_pnp_inliers = (int32_t)tp.min_pnp_inliers_;
    pnp_params_.ransac_threshold_pnp = tp.ransac_threshold_pnp_;
    pnp_params_.optimize_2d3d_pose_from_inliers = tp.optimize_2d3d_pose_from_inliers_ ? 1 : 0;
  }
  // Tracker.h:82-94, LandmarksMap = std::unordered_map<LandmarkId, gtsam::Point3>
  template <class LandmarksMapT>
  void updateMap(const LandmarksMapT& lmks_map) {
    kvfe::Tracker::LandmarksMap m;
    for (const auto& it : lmks_map) m[(int64_t)it.first] = {it.second(0), it.second(1), it.second(2)};
    impl_.updateMap(m);
  }
  // bool pnp(const BearingVectors& cam_bearing_vectors, const Landmarks& F_points, gtsam::Pose3* F_Pose_cam_estimate,
  //          std::vector<int>* inliers, gtsam::Pose3* F_Pose_cam_prior = nullptr)        (Tracker.h, Tracker.cpp:1122)
  template <class BearingVectorsT, class LandmarksT, class Pose3T>
  bool pnp(const BearingVectorsT& cam_bearing_vectors, const LandmarksT& F_points, Pose3T* F_Pose_cam_estimate,
           std::vector<int>* inliers, Pose3T* /*F_Pose_cam_prior*/ = nullptr) {
    std::vector<double> f, pw;
    for (const auto& b : cam_bearing_vectors) f.insert(f.end(), {b(0), b(1), b(2)});
    for (const auto& q : F_points) pw.insert(pw.end(), {q(0), q(1), q(2)});
    double T[12];
    const bool ok = impl_.pnp(f, pw, pnp_params_, T, inliers);
    *F_Pose_cam_estimate = pose_from_array<Pose3T>(T);
    return ok;
  }
  // bool pnp(const StereoFrame& cur_stereo_frame, gtsam::Pose3* W_Pose_cam_estimate, std::vector<int>* inliers,
  //          gtsam::Pose3* W_Pose_cam_prior = nullptr)                                  (Tracker.cpp:1064-1120)
  template <class VioStereoFrame, class Pose3T>
  bool pnp(const VioStereoFrame& cur_stereo_frame, Pose3T* W_Pose_cam_estimate, std::vector<int>* inliers,
           Pose3T* /*W_Pose_cam_prior*/ = nullptr) {
    const size_t n = cur_stereo_frame.left_keypoints_rectified_.size();
    std::vector<uint8_t> st(n);
    std::vector<int64_t> ids(n);
    std::vector<double> p3(3 * n);
    for (size_t i = 0; i < n; i++) {
      st[i] = (uint8_t)cur_stereo_frame.left_keypoints_rectified_[i].first;
      ids[i] = (int64_t)cur_stereo_frame.left_frame_.landmarks_[i];
      for (int c = 0; c < 3; c++) p3[3 * i + c] = cur_stereo_frame.keypoints_3d_[i](c);
    }
    double T[12];
    const bool ok = impl_.pnp(st.data(), ids.data(), p3.data(), n, pnp_params_, T, inliers);
    *W_Pose_cam_estimate = pose_from_array<Pose3T>(T);
    return ok;
  }

 private:
  template <class Pose3T>
  static Pose3T pose_from_array(const double T[12]) {
    gtsam::Matrix3 M;
    gtsam::Vector3 t;
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) M(r, c) = T[4 * r + c];
      t(r) = T[4 * r + 3];
    }
    return Pose3T(gtsam::Rot3(M), t);
  }
  kvfe::Tracker impl_;
  kvfe_pnp_params pnp_params_ = {3 /* EPNP */, 20, 1.0, 0, 0};
};

class UndistorterRectifier {
 public:
  UndistorterRectifier(Context ctx, int cam) : impl_(std::move(ctx), cam) {}
  // UndistorterRectifier.h:76-78
  void undistortRectifyImage(const cv::Mat& img, cv::Mat* undistorted_img) const {
    undistorted_img->create(img.rows, img.cols, CV_8UC1);
    impl_.undistortRectifyImage(view(img), undistorted_img->data);
  }
  // UndistorterRectifier.h:96-100 / 108-110
  template <class Kps, class StatusKps>
  void checkUndistortedRectifiedLeftKeypoints(const Kps& distorted_kps, const Kps& undistorted_kps, StatusKps* status_kps,
                                              const float& pixel_tol = 2.0f) const {
    KeypointsCV d(distorted_kps.size()), u(undistorted_kps.size());
    for (size_t i = 0; i < d.size(); i++) d[i] = KeypointCV{distorted_kps[i].x, distorted_kps[i].y};
    for (size_t i = 0; i < u.size(); i++) u[i] = KeypointCV{undistorted_kps[i].x, undistorted_kps[i].y};
    StatusKeypointsCV out;
    impl_.checkUndistortedRectifiedLeftKeypoints(d, u, &out, pixel_tol);
    from_status(out, status_kps);
  }
  template <class StatusKps, class Kps>
  void distortUnrectifyKeypoints(const StatusKps& keypoints_rectified, Kps* keypoints_unrectified) const {
    KeypointsCV out;
    impl_.distortUnrectifyKeypoints(to_status(keypoints_rectified), &out);
    keypoints_unrectified->resize(out.size());
    for (size_t i = 0; i < out.size(); i++) {
      (*keypoints_unrectified)[i].x = out[i].x;
      (*keypoints_unrectified)[i].y = out[i].y;
    }
  }

 private:
  kvfe::UndistorterRectifier impl_;
};

class StereoMatcher {
 public:
  explicit StereoMatcher(Context ctx) : impl_(ctx), camera_(ctx) {}
  // StereoMatcher.h:49-52, StereoMatcher.cpp:123-175: rectified images, rectified left / right keypoints with their
  // statuses, depths, distorted right keypoints and 3-D points of `stereo_frame`, from its two images and its left
  // keypoints (one call into the library: the reference's five steps run fused on the device)
  template <class VioStereoFrame>
  void sparseStereoReconstruction(VioStereoFrame* stereo_frame) {
    const cv::Mat& l = stereo_frame->left_frame_.img_;
    const cv::Mat& r = stereo_frame->right_frame_.img_;
    const auto& kps = stereo_frame->left_frame_.keypoints_;
    if (kps.empty()) throw Error(KVFE_ERR_INVALID_ARG, "Call feature detection on left frame first...");   // CHECK_GT upstream
    cv::Mat lr, rr;
    lr.create(l.rows, l.cols, CV_8UC1);
    rr.create(r.rows, r.cols, CV_8UC1);
    camera_.undistortRectifyStereoFrame(view(l), view(r), lr.data, rr.data);
    stereo_frame->setRectifiedImages(lr, rr);
    KeypointsCV k(kps.size());
    for (size_t i = 0; i < k.size(); i++) k[i] = KeypointCV{kps[i].x, kps[i].y};
    const kvfe::StereoMatcher::SparseResult res = impl_.sparseStereoReconstruction(view(l), view(r), k);
    from_status(res.left_keypoints_rectified, &stereo_frame->left_keypoints_rectified_);
    from_status(res.right_keypoints_rectified, &stereo_frame->right_keypoints_rectified_);
    stereo_frame->keypoints_depth_ = res.keypoints_depth;
    stereo_frame->right_frame_.keypoints_.resize(k.size());
    stereo_frame->keypoints_3d_.resize(k.size());
    for (size_t i = 0; i < k.size(); i++) {
      stereo_frame->right_frame_.keypoints_[i].x = res.right_keypoints[i].x;
      stereo_frame->right_frame_.keypoints_[i].y = res.right_keypoints[i].y;
      for (int c = 0; c < 3; c++) stereo_frame->keypoints_3d_[i](c) = res.keypoints_3d[3 * i + c];
    }
  }
  // StereoMatcher.h:54-66, StereoMatcher.cpp:32-121: *disparity_img becomes the CV_16S matrix (4 fractional bits)
  // cv::StereoSGBM / cv::StereoBM ::compute leaves there, after the optional median blur
  void denseStereoReconstruction(const cv::Mat& left_img_rectified, const cv::Mat& right_img_rectified,
                                 cv::Mat* disparity_img) {
    if (!disparity_img || right_img_rectified.cols != left_img_rectified.cols ||
        right_img_rectified.rows != left_img_rectified.rows)
      throw Error(KVFE_ERR_INVALID_ARG, "denseStereoReconstruction: image sizes differ");   // CHECK_EQ upstream
    disparity_img->create(left_img_rectified.rows, left_img_rectified.cols, CV_16SC1);
    impl_.denseStereoReconstruction(view(left_img_rectified), view(right_img_rectified),
                                    reinterpret_cast<int16_t*>(disparity_img->data), disparity_img->step / sizeof(int16_t));
  }
  kvfe_dense_stereo_params& denseStereoParams() { return impl_.dense_stereo_params_; }   // DenseStereoParams
  // StereoMatcher.h:85-92
  template <class StatusKps>
  void getDepthFromRectifiedMatches(StatusKps& left_keypoints_rectified, StatusKps& right_keypoints_rectified,
                                    std::vector<double>* keypoints_depth) const {
    StatusKeypointsCV l = to_status(left_keypoints_rectified), r = to_status(right_keypoints_rectified);
    impl_.getDepthFromRectifiedMatches(l, r, keypoints_depth);
    from_status(r, &right_keypoints_rectified);
  }
  // StereoMatcher.h:62-66 (the overload on rectified images)
  template <class StatusKps>
  void getRightKeypointsRectified(const cv::Mat& left_rectified, const cv::Mat& right_rectified,
                                  const StatusKps& left_keypoints_rectified, StatusKps* right_keypoints_rectified) const {
    StatusKeypointsCV r;
    impl_.getRightKeypointsRectified(view(left_rectified), view(right_rectified), to_status(left_keypoints_rectified), &r);
    from_status(r, right_keypoints_rectified);
  }

 private:
  kvfe::StereoMatcher impl_;
  kvfe::StereoCamera camera_;
};

class StereoCamera {
 public:
  explicit StereoCamera(Context ctx) : impl_(std::move(ctx)) {}
  // StereoCamera.h:225-233: stereo_frame->setRectifiedImages(left, right)
  template <class VioStereoFrame>
  void undistortRectifyStereoFrame(VioStereoFrame* stereo_frame) const {
    const cv::Mat& l = stereo_frame->left_frame_.img_;
    const cv::Mat& r = stereo_frame->right_frame_.img_;
    cv::Mat lr, rr;
    lr.create(l.rows, l.cols, CV_8UC1);
    rr.create(r.rows, r.cols, CV_8UC1);
    impl_.undistortRectifyStereoFrame(view(l), view(r), lr.data, rr.data);
    stereo_frame->setRectifiedImages(lr, rr);
  }
  double getBaseline() const { return impl_.getBaseline(); }

 private:
  kvfe::StereoCamera impl_;
};

// StereoVisionImuFrontend::spinOnce (VisionImuFrontend.h:63-64 -> nominalSpinStereo / bootstrapSpinStereo,
// StereoVisionImuFrontend.cpp:102-276) on the reference's packet types.  What the library computes is handed to a
// caller-supplied `build` callable as a SpinResult, from which the call site constructs its StereoFrontendOutput with
// the reference's own constructor (StereoVisionImuFrontend-definitions.h:30-55): the IMU front-end (pim) and the
// display image stay where they are upstream, on the host.
//   Packet:  getStereoFrame() (left_frame_.img_, right_frame_.img_, timestamp_), getImuStamps() (1 x k int64),
//            getImuAccGyrs() (6 x k double), as StereoImuSyncPacket.h:81-107 / FrontendInputPacketBase.h:37-52
struct SpinResult {
  bool is_keyframe = false;
  int64_t timestamp = 0;
  // TrackerStatusSummary (Tracker-definitions.h:134-183)
  int kfTrackingStatus_mono = 0, kfTrackingStatus_stereo = 0, kfTracking_status_pnp = 0;
  double lkf_T_k_mono[12], lkf_T_k_stereo[12], W_T_k_pnp[12], infoMatStereoTranslation[9];
  // StereoMeasurements: (LandmarkId, StereoPoint2(uL, uR or NaN, v)) (StereoFrame-definitions.h:24-28)
  std::vector<int64_t> meas_landmark;
  std::vector<double> meas_uL_uR_v;   // n x 3
  // stereoFrame_k_: left frame tables and the stereo arrays of a keyframe
  Frame left_frame;
  StatusKeypointsCV left_keypoints_rectified, right_keypoints_rectified;
  std::vector<double> keypoints_depth, keypoints_3d;   // n, n x 3
  KeypointsCV right_keypoints;
  // DebugTrackerInfo (Tracker-definitions.h:78-124)
  int nrMonoPutatives = 0, nrMonoInliers = 0, monoRansacIters = 0, nrStereoPutatives = 0, nrStereoInliers = 0;
};

class StereoVisionImuFrontend {
 public:
  // body_R_camLrect: rotation of StereoCamera::getBodyPoseLeftCamRect (R_BS . R1^T), row-major
  StereoVisionImuFrontend(Context ctx, const double body_R_camLrect[9]) : c_(std::move(ctx)) {
    for (int i = 0; i < 9; i++) {
      body_R_cam_[i] = body_R_camLrect[i];
      deltaRij_[i] = (i % 4 == 0) ? 1.0 : 0.0;
    }
    if (c_.config().batch != 1) throw Error(KVFE_ERR_INVALID_ARG, "the packet interface drives one stream");
  }
  // ImuFrontend::updateBias (the back-end's estimate; only the gyro part enters the rotation)
  void updateImuBias(const double gyro_bias[3]) {
    for (int i = 0; i < 3; i++) gyro_bias_[i] = gyro_bias[i];
  }
  template <class Packet, class Build>
  auto spinOnce(Packet&& input, Build&& build) -> decltype(build(std::declval<const SpinResult&>())) {
    const auto& sf = input.getStereoFrame();
    const auto& stamps = input.getImuStamps();
    const auto& accgyr = input.getImuAccGyrs();
    const int k = (int)stamps.cols();
    // ImuFrontend::preintegrateImuMeasurements, rotation part, since the last keyframe (StereoVisionImuFrontend.cpp:125-150)
    std::vector<int64_t> t(k);
    std::vector<double> ag((size_t)6 * k);
    for (int j = 0; j < k; j++) {
      t[j] = (int64_t)stamps(0, j);
      for (int r = 0; r < 6; r++) ag[(size_t)6 * j + r] = accgyr(r, j);
    }
    if (initialised_)
      c_.check(kvfe_imu_preintegrate_rotation(t.data(), ag.data(), k, gyro_bias_, deltaRij_), "preintegrateImuMeasurements");
    kvfe_frame_input in;
    std::memset(&in, 0, sizeof(in));
    in.timestamp_ns = (int64_t)sf.timestamp_;
    kvfe_keyframe_R_cur_frame(body_R_cam_, deltaRij_, in.keyframe_R_cur_frame);
    const ImageView l = view(sf.left_frame_.img_), r = view(sf.right_frame_.img_);
    if (l.step != r.step) throw Error(KVFE_ERR_INVALID_ARG, "left / right images with different strides");
    c_.check(kvfe_frontend_step_host(c_.get(), l.data, r.data, l.step, l.step * (size_t)l.rows, &in), "spinOnce");
    initialised_ = true;
    // StereoFrontendOutput: read in place from the step's packed record in pinned host memory (kvfe_frontend_view_output:
    // one event wait for this step's record, no device copy, no intermediate buffers)
    SpinResult o;
    kvfe_frame_output f;
    std::memset(&f, 0, sizeof(f));
    c_.check(kvfe_frontend_view_output(c_.get(), 0, 0, &f), "getOutput");
    const int64_t* lmk = f.landmarks;
    const int64_t* mlmk = f.meas_landmark;
    const int32_t* age = f.landmarks_age;
    const float *kp = f.keypoints, *lr = f.left_rect_xy, *rr = f.right_rect_xy, *rxy = f.right_xy;
    const double *ver = f.versors, *dep = f.depth, *p3 = f.keypoints_3d, *muv = f.meas_uL_uR_v;
    const uint8_t *ls = f.left_status, *rs = f.right_status;
    o.is_keyframe = f.is_keyframe != 0;
    o.timestamp = in.timestamp_ns;
    o.kfTrackingStatus_mono = f.tracking_status_mono;
    o.kfTrackingStatus_stereo = f.tracking_status_stereo;
    o.kfTracking_status_pnp = f.tracking_status_pnp;
    std::memcpy(o.lkf_T_k_mono, f.lkf_T_k_mono, sizeof(o.lkf_T_k_mono));
    std::memcpy(o.lkf_T_k_stereo, f.lkf_T_k_stereo, sizeof(o.lkf_T_k_stereo));
    std::memcpy(o.W_T_k_pnp, f.W_T_k_pnp, sizeof(o.W_T_k_pnp));
    std::memcpy(o.infoMatStereoTranslation, f.info_mat_stereo_translation, sizeof(o.infoMatStereoTranslation));
    o.nrMonoPutatives = f.nr_mono_putatives; o.nrMonoInliers = f.nr_mono_inliers; o.monoRansacIters = f.mono_ransac_iters;
    o.nrStereoPutatives = f.nr_stereo_putatives; o.nrStereoInliers = f.nr_stereo_inliers;
    const int n = std::min(f.n_keypoints, f.capacity), m = std::min(f.n_measurements, f.capacity);
    o.meas_landmark.assign(mlmk, mlmk + m);
    o.meas_uL_uR_v.assign(muv, muv + 3 * m);
    o.left_frame.img_ = l;
    o.left_frame.keypoints_.resize(n);
    o.left_frame.landmarks_.assign(lmk, lmk + n);
    o.left_frame.landmarks_age_.assign(age, age + n);
    o.left_frame.versors_.assign(ver, ver + 3 * n);
    o.left_keypoints_rectified.resize(n); o.right_keypoints_rectified.resize(n); o.right_keypoints.resize(n);
    for (int i = 0; i < n; i++) o.left_frame.keypoints_[i] = KeypointCV{kp[2 * i], kp[2 * i + 1]};
    if (ls) {   // a frame with stereo data (every keyframe): the record holds the stereo arrays
      for (int i = 0; i < n; i++) {
        o.left_keypoints_rectified[i] = {static_cast<KeypointStatus>(ls[i]), KeypointCV{lr[2 * i], lr[2 * i + 1]}};
        o.right_keypoints_rectified[i] = {static_cast<KeypointStatus>(rs[i]), KeypointCV{rr[2 * i], rr[2 * i + 1]}};
        o.right_keypoints[i] = KeypointCV{rxy[2 * i], rxy[2 * i + 1]};
      }
      o.keypoints_depth.assign(dep, dep + n);
      o.keypoints_3d.assign(p3, p3 + 3 * n);
    } else {    // (what the zero-initialised buffers of the copying read gave: status 0, zero coordinates)
      for (int i = 0; i < n; i++) {
        o.left_keypoints_rectified[i] = {static_cast<KeypointStatus>(0), KeypointCV{0.f, 0.f}};
        o.right_keypoints_rectified[i] = {static_cast<KeypointStatus>(0), KeypointCV{0.f, 0.f}};
        o.right_keypoints[i] = KeypointCV{0.f, 0.f};
      }
      o.keypoints_depth.assign(n, 0.0);
      o.keypoints_3d.assign(3 * (size_t)n, 0.0);
    }
    if (o.is_keyframe)   // ImuFrontend::resetIntegrationWithCachedBias (StereoVisionImuFrontend.cpp:203)
      for (int i = 0; i < 9; i++) deltaRij_[i] = (i % 4 == 0) ? 1.0 : 0.0;
    return build(static_cast<const SpinResult&>(o));
  }

 private:
  Context c_;
  double body_R_cam_[9], deltaRij_[9], gyro_bias_[3] = {0, 0, 0};
  bool initialised_ = false;
};

}  // namespace shim
}  // namespace kvfe
