// kvfe_adapter.hpp — header-only C++ adapter over the C ABI of libkvfe (kvfe.h) that carries the
// class and method names of Kimera-VIO's stereo front-end, written against plain structs so that
// it compiles without OpenCV / GTSAM.  INTEGRATION.md shows how a Kimera-VIO maintainer maps
// cv::Mat / cv::Point2f / gtsam::Rot3 / StatusKeypointsCV onto these (they are layout compatible:
// KeypointCV == cv::Point2f == {float x, y}; StatusKeypointCV == {KeypointStatus, KeypointCV}).
//
// Reference interfaces mirrored (paths relative to the Kimera-VIO tree):
//   UndistorterRectifier   include/kimera-vio/frontend/UndistorterRectifier.h:51-114
//   StereoCamera           include/kimera-vio/frontend/StereoCamera.h:203-233
//   FeatureDetector        include/kimera-vio/frontend/feature-detector/FeatureDetector.h:39-49
//   Tracker                include/kimera-vio/frontend/Tracker.h:70-74
//   StereoMatcher          include/kimera-vio/frontend/StereoMatcher.h:49-92
//   StereoVisionImuFrontend include/kimera-vio/frontend/StereoVisionImuFrontend.h (spinOnce path)
//   utils::ThreadsafeImuBuffer include/kimera-vio/utils/ThreadsafeImuBuffer.h:46-196                (input side)
//   StereoDataProviderModule include/kimera-vio/dataprovider/StereoDataProviderModule.h:30-77        (input side)
//   UtilsOpenCV::ReadAndConvertToGrayScale include/kimera-vio/utils/UtilsOpenCV.h (PNG files)       (input side)
// Error behaviour: the reference CHECK-aborts on contract violations; the adapter throws
// kvfe::Error carrying the kvfe_status and kvfe_last_error() text instead (never UB, never a
// silent CPU fallback — there is none).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <array>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "kvfe.h"

namespace kvfe {

struct Error : std::runtime_error {
  kvfe_status status;
  Error(kvfe_status s, const std::string& what) : std::runtime_error(what), status(s) {}
};

// KeypointCV / StatusKeypointCV / KeypointStatus (include/kimera-vio/common/vio_types.h:38-66)
struct KeypointCV {
  float x, y;
};
enum class KeypointStatus : uint8_t {
  VALID = KVFE_KP_VALID,
  NO_LEFT_RECT = KVFE_KP_NO_LEFT_RECT,
  NO_RIGHT_RECT = KVFE_KP_NO_RIGHT_RECT,
  NO_DEPTH = KVFE_KP_NO_DEPTH,
  FAILED_ARUN = KVFE_KP_FAILED_ARUN
};
using KeypointsCV = std::vector<KeypointCV>;
using StatusKeypointCV = std::pair<KeypointStatus, KeypointCV>;
using StatusKeypointsCV = std::vector<StatusKeypointCV>;

// 8-bit gray image view (cv::Mat CV_8UC1: data, rows, cols, step)
struct ImageView {
  const uint8_t* data;
  int rows, cols;
  size_t step;
};

// VIO::Frame as far as the hot path touches it (include/kimera-vio/frontend/Frame.h:160-186): img_ and the
// parallel arrays keypoints_ / landmarks_ / landmarks_age_ / versors_ (scores_ is never set upstream)
struct Frame {
  ImageView img_{nullptr, 0, 0, 0};
  KeypointsCV keypoints_;
  std::vector<int64_t> landmarks_;
  std::vector<int32_t> landmarks_age_;
  std::vector<double> versors_;  // n x 3

  // Frame::getNrValidKeypoints / getValidKeypoints / findLmkIdFromPixel (Frame.h:97-139)
  size_t getNrValidKeypoints() const {
    size_t count = 0;
    for (int64_t l : landmarks_) count += l != -1;
    return count;
  }
  KeypointsCV getValidKeypoints() const {
    KeypointsCV valid;
    for (size_t i = 0; i < landmarks_.size() && i < keypoints_.size(); i++)
      if (landmarks_[i] != -1) valid.push_back(keypoints_[i]);
    return valid;
  }
  static int64_t findLmkIdFromPixel(const KeypointCV& px, const KeypointsCV& keypoints,
                                    const std::vector<int64_t>& landmarks, size_t* idx_in_keypoints = nullptr) {
    for (size_t i = 0; i < keypoints.size(); i++)
      if (keypoints[i].x == px.x && keypoints[i].y == px.y) {
        if (idx_in_keypoints) *idx_in_keypoints = i;
        return landmarks.at(i);
      }
    return -1;
  }
};

// gtsam::Pose3 as far as the front-end's outputs need it (rotation matrix + translation; compose / inverse /
// between with gtsam's formulas: R = R1 R2, t = t1 + R1 t2; inverse = (R^T, -R^T t))
struct Pose3 {
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double t[3] = {0, 0, 0};
  Pose3 compose(const Pose3& o) const {
    Pose3 r;
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) r.R[3 * i + j] = R[3 * i] * o.R[j] + R[3 * i + 1] * o.R[3 + j] + R[3 * i + 2] * o.R[6 + j];
      r.t[i] = t[i] + (R[3 * i] * o.t[0] + R[3 * i + 1] * o.t[1] + R[3 * i + 2] * o.t[2]);
    }
    return r;
  }
  Pose3 inverse() const {
    Pose3 r;
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) r.R[3 * i + j] = R[3 * j + i];
      r.t[i] = -(R[i] * t[0] + R[3 + i] * t[1] + R[6 + i] * t[2]);
    }
    return r;
  }
  Pose3 between(const Pose3& o) const { return inverse().compose(o); }
  // row-major 3x4 [R | t] (kvfe_frame_output::lkf_T_k_*) / row-major 4x4 (kvfe_camera_params::body_pose_cam)
  static Pose3 from3x4(const double* m) {
    Pose3 r;
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) r.R[3 * i + j] = m[4 * i + j];
      r.t[i] = m[4 * i + 3];
    }
    return r;
  }
  static Pose3 from4x4(const double* m) { return from3x4(m); }
};

// StereoCamera::B_Pose_camLrect_ / B_Pose_camRrect_ (src/frontend/StereoCamera.cpp:55-66):
//   camL_Pose_camLrect = (R1^-1, 0);  B_Pose_camLrect = left.body_Pose_cam . camL_Pose_camLrect
// and the same with R2 for the right camera -- upstream composes that one with the LEFT camera's body pose as well
// (StereoCamera.cpp:66), which is reproduced, not corrected.  Host arithmetic: no context needed.
inline Pose3 bodyPoseCamRect(const double R_rect[9], const kvfe_camera_params& left_cam_params) {
  Pose3 cam_Pose_camrect;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) cam_Pose_camrect.R[3 * i + j] = R_rect[3 * j + i];   // cvMatToGtsamRot3(R).inverse()
  return Pose3::from4x4(left_cam_params.body_pose_cam).compose(cam_Pose_camrect);
}
// gtsam::Cal3_S2Stereo of the rectified pair (StereoCamera.cpp:73-80: UtilsOpenCV::Cvmat2Cal3_S2(P1) + baseline)
struct Cal3_S2Stereo {
  double fx, fy, skew, px, py, baseline;
};
inline Cal3_S2Stereo stereoCalib(const kvfe_rectification& r) {
  return Cal3_S2Stereo{r.P1[0], r.P1[5], r.P1[1], r.P1[2], r.P1[6], r.baseline};
}
// StereoVisionImuFrontend::getRelativePoseBodyMono / getRelativePoseBodyStereo (StereoVisionImuFrontend.cpp:699-716):
// lkfBody_T_kBody = body_Pose_cam . lkf_T_k . body_Pose_cam^-1 with the rectified LEFT camera's body pose
inline Pose3 relativePoseBody(const Pose3& body_Pose_camLrect, const double lkf_T_k[12]) {
  return body_Pose_camLrect.compose(Pose3::from3x4(lkf_T_k)).compose(body_Pose_camLrect.inverse());
}

namespace detail {
// kvfe_frame view of a Frame whose vectors have been resized to `capacity`
inline kvfe_frame frame_view(Frame* f, int n, int capacity) {
  f->keypoints_.resize(capacity);
  f->landmarks_.resize(capacity);
  f->landmarks_age_.resize(capacity);
  f->versors_.resize((size_t)capacity * 3);
  kvfe_frame v;
  v.capacity = capacity;
  v.n_keypoints = n;
  v.keypoints = &f->keypoints_.data()->x;
  v.landmarks = f->landmarks_.data();
  v.landmarks_age = f->landmarks_age_.data();
  v.versors = f->versors_.data();
  return v;
}
inline void frame_shrink(Frame* f, int n) {
  f->keypoints_.resize(n);
  f->landmarks_.resize(n);
  f->landmarks_age_.resize(n);
  f->versors_.resize((size_t)n * 3);
}
inline int frame_capacity(const kvfe_config& cfg) {
  return cfg.params.detector.max_features_per_frame + cfg.params.detector.max_nr_keypoints_before_anms + 64;
}
}  // namespace detail

// Owns a kvfe_ctx: StereoCamera + the four front-end classes share it, as the reference's
// StereoVisionImuFrontend owns its StereoCamera::ConstPtr, Tracker, FeatureDetector and
// StereoMatcher.
class Context {
 public:
  Context(const kvfe_camera_params& left, const kvfe_camera_params& right,
          const kvfe_frontend_params& params, int batch = 1, int device = 0) {
    std::memset(&cfg_, 0, sizeof(cfg_));
    cfg_.left = left;
    cfg_.right = right;
    cfg_.params = params;
    cfg_.batch = batch;
    cfg_.device = device;
    kvfe_ctx* c = nullptr;
    const kvfe_status s = kvfe_create(&cfg_, &c);
    if (s != KVFE_OK) throw Error(s, std::string("kvfe_create: ") + kvfe_status_string(s));
    ctx_.reset(c, kvfe_destroy);
  }
  // any kvfe_config (frontend_type KVFE_FRONTEND_MONO / _RGBD, stream groups, a caller-owned HIP stream ...)
  explicit Context(const kvfe_config& cfg) : cfg_(cfg) {
    kvfe_ctx* c = nullptr;
    const kvfe_status s = kvfe_create(&cfg_, &c);
    if (s != KVFE_OK) throw Error(s, std::string("kvfe_create: ") + kvfe_status_string(s));
    ctx_.reset(c, kvfe_destroy);
  }
  kvfe_ctx* get() const { return ctx_.get(); }
  const kvfe_config& config() const { return cfg_; }
  void check(kvfe_status s, const char* where) const {
    if (s != KVFE_OK)
      throw Error(s, std::string(where) + ": " + kvfe_status_string(s) + " (" +
                         kvfe_last_error(ctx_.get()) + ")");
  }

 private:
  kvfe_config cfg_;
  std::shared_ptr<kvfe_ctx> ctx_;
};

// UndistorterRectifier (src/frontend/UndistorterRectifier.cpp); cam: 0 = left, 1 = right
class UndistorterRectifier {
 public:
  UndistorterRectifier(Context ctx, int cam) : c_(std::move(ctx)), cam_(cam) {}
  // undistortRectifyImage (UndistorterRectifier.cpp:115-128); dst: cols*rows bytes, tightly packed
  void undistortRectifyImage(const ImageView& img, uint8_t* dst) const {
    c_.check(kvfe_undistort_rectify_image(c_.get(), cam_, img.data, img.step, dst, (size_t)img.cols),
             "undistortRectifyImage");
  }
  // undistortRectifyKeypoints (UndistorterRectifier.cpp:130-136)
  void undistortRectifyKeypoints(const KeypointsCV& kps, KeypointsCV* out) const {
    out->resize(kps.size());
    c_.check(kvfe_undistort_rectify_keypoints(c_.get(), cam_, &kps.data()->x, (int32_t)kps.size(), 1, 1,
                                              &out->data()->x),
             "undistortRectifyKeypoints");
  }
  // GetBearingVector (UndistorterRectifier.cpp:73-113), batched: n x 3 doubles
  void getBearingVectors(const KeypointsCV& kps, std::vector<double>* versors) const {
    versors->resize(kps.size() * 3);
    c_.check(kvfe_get_bearing_vectors(c_.get(), cam_, &kps.data()->x, (int32_t)kps.size(), versors->data()),
             "GetBearingVector");
  }
  // checkUndistortedRectifiedLeftKeypoints(distorted_kps, undistorted_kps, status_kps, pixel_tol = 2.0f)
  // (UndistorterRectifier.h:96-100, UndistorterRectifier.cpp:138-211)
  void checkUndistortedRectifiedLeftKeypoints(const KeypointsCV& distorted_kps, const KeypointsCV& undistorted_kps,
                                              StatusKeypointsCV* status_kps, const float& pixel_tol = 2.0f) const {
    if (distorted_kps.size() != undistorted_kps.size())   // CHECK_EQ upstream
      throw Error(KVFE_ERR_INVALID_ARG, "checkUndistortedRectifiedLeftKeypoints: size mismatch");
    const int32_t n = (int32_t)distorted_kps.size();
    KeypointsCV out(n);
    std::vector<uint8_t> st(n);
    c_.check(kvfe_check_undistorted_rectified_left_keypoints(c_.get(), cam_, n ? &distorted_kps.data()->x : nullptr,
                                                             n ? &undistorted_kps.data()->x : nullptr, n, pixel_tol,
                                                             n ? &out.data()->x : reinterpret_cast<float*>(&out),
                                                             n ? st.data() : reinterpret_cast<uint8_t*>(&st)),
             "checkUndistortedRectifiedLeftKeypoints");
    status_kps->resize(n);
    for (int32_t i = 0; i < n; i++) (*status_kps)[i] = {static_cast<KeypointStatus>(st[i]), out[i]};
  }
  // distortUnrectifyKeypoints(keypoints_rectified, keypoints_unrectified) (UndistorterRectifier.cpp:213-228)
  void distortUnrectifyKeypoints(const StatusKeypointsCV& keypoints_rectified, KeypointsCV* keypoints_unrectified) const {
    const int32_t n = (int32_t)keypoints_rectified.size();
    KeypointsCV r(n);
    std::vector<uint8_t> st(n);
    for (int32_t i = 0; i < n; i++) {
      st[i] = static_cast<uint8_t>(keypoints_rectified[i].first);
      r[i] = keypoints_rectified[i].second;
    }
    keypoints_unrectified->assign(n, KeypointCV{0.f, 0.f});
    if (n == 0) return;
    c_.check(kvfe_distort_unrectify_keypoints(c_.get(), cam_, &r.data()->x, st.data(), n,
                                              &keypoints_unrectified->data()->x),
             "distortUnrectifyKeypoints");
  }

 private:
  Context c_;
  int cam_;
};

// StereoCamera (src/frontend/StereoCamera.cpp:34-94,236-379)
class StereoCamera {
 public:
  explicit StereoCamera(Context ctx) : c_(std::move(ctx)) {
    c_.check(kvfe_get_rectification(c_.get(), &rect_), "computeRectificationParameters");
  }
  double getBaseline() const { return rect_.baseline; }
  const double* getR1() const { return rect_.R1; }
  const double* getR2() const { return rect_.R2; }
  const double* getP1() const { return rect_.P1; }
  const double* getP2() const { return rect_.P2; }
  const double* getQ() const { return rect_.Q; }
  Cal3_S2Stereo getStereoCalib() const { return stereoCalib(rect_); }
  // getBodyPoseLeftCamRect / getBodyPoseRightCamRect (StereoCamera.h; the context does not keep the YAML's T_BS, the
  // caller passes the left camera's parameters it created the context with)
  Pose3 getBodyPoseLeftCamRect(const kvfe_camera_params& left_cam_params) const { return bodyPoseCamRect(rect_.R1, left_cam_params); }
  Pose3 getBodyPoseRightCamRect(const kvfe_camera_params& left_cam_params) const { return bodyPoseCamRect(rect_.R2, left_cam_params); }
  const Context& context() const { return c_; }
  // undistortRectifyLeftKeypoints(keypoints, status_keypoints_rectified) (StereoCamera.h:203-213, .cpp:236-260)
  void undistortRectifyLeftKeypoints(const KeypointsCV& keypoints, StatusKeypointsCV* status_keypoints_rectified) const {
    const int32_t n = (int32_t)keypoints.size();
    KeypointsCV out(n);
    std::vector<uint8_t> st(n);
    status_keypoints_rectified->resize(n);
    if (n == 0) return;
    c_.check(kvfe_undistort_rectify_left_keypoints(c_.get(), &keypoints.data()->x, n, &out.data()->x, st.data()),
             "undistortRectifyLeftKeypoints");
    for (int32_t i = 0; i < n; i++) (*status_keypoints_rectified)[i] = {static_cast<KeypointStatus>(st[i]), out[i]};
  }
  // distortUnrectifyRightKeypoints(status_keypoints_rectified, keypoints) (StereoCamera.h:215-223, .cpp:262-267)
  void distortUnrectifyRightKeypoints(const StatusKeypointsCV& status_keypoints_rectified, KeypointsCV* keypoints) const {
    UndistorterRectifier(c_, 1).distortUnrectifyKeypoints(status_keypoints_rectified, keypoints);
  }
  // undistortRectifyStereoFrame(StereoFrame*) (StereoCamera.h:225-233, .cpp:269-290): the two rectified images
  // (StereoFrame::setRectifiedImages), tightly packed cols*rows bytes each
  void undistortRectifyStereoFrame(const ImageView& left_img, const ImageView& right_img, uint8_t* left_img_rectified,
                                   uint8_t* right_img_rectified) const {
    c_.check(kvfe_undistort_rectify_stereo_frame(c_.get(), left_img.data, right_img.data, left_img.step,
                                                 left_img_rectified, right_img_rectified, (size_t)left_img.cols),
             "undistortRectifyStereoFrame");
  }
  // backProjectDisparityTo3D(disparity_img CV_32F, depth CV_32FC3) (StereoCamera.cpp:176-196):
  // disparity: h x w float (CV_16S / 16), depth: h x w x 3 float
  void backProjectDisparityTo3D(const float* disparity, size_t stride_elems, float* depth) const {
    c_.check(kvfe_backproject_disparity_to_3d(c_.get(), disparity, stride_elems, depth),
             "backProjectDisparityTo3D");
  }

 private:
  Context c_;
  kvfe_rectification rect_;
};

// FeatureDetector (src/frontend/feature-detector/FeatureDetector.cpp)
class FeatureDetector {
 public:
  explicit FeatureDetector(Context ctx) : c_(std::move(ctx)) {}
  // rawFeatureDetection (FeatureDetector.cpp:165-172); mask may be null
  KeypointsCV rawFeatureDetection(const ImageView& img, const ImageView* mask = nullptr) const {
    KeypointsCV out(8192);
    int32_t n = 0;
    c_.check(kvfe_raw_feature_detection(c_.get(), img.data, img.step, mask ? mask->data : nullptr,
                                        mask ? mask->step : 0, &out.data()->x, (int32_t)out.size(), &n),
             "rawFeatureDetection");
    out.resize(n);
    return out;
  }
  // void featureDetection(Frame* cur_frame, std::optional<cv::Mat> R) (FeatureDetector.h:39-41,
  // FeatureDetector.cpp:94-163).  R is the context's (R1 for a stereo context, none for a mono one).  New landmarks
  // take their ids from landmarkCounter(): the reference keeps a function-static `lmk_id` (FeatureDetector.cpp:141)
  // shared by every FeatureDetector of the process, and so does this adapter.
  static int64_t& landmarkCounter() {
    static int64_t lmk_id = 0;
    return lmk_id;
  }
  void featureDetection(Frame* cur_frame) const {
    const int cap = detail::frame_capacity(c_.config());
    const int n = (int)cur_frame->landmarks_.size();
    kvfe_frame v = detail::frame_view(cur_frame, n, cap);
    const kvfe_status s = kvfe_feature_detection_frame(c_.get(), cur_frame->img_.data, cur_frame->img_.step, &v,
                                                       &landmarkCounter());
    detail::frame_shrink(cur_frame, s == KVFE_OK ? v.n_keypoints : n);
    c_.check(s, "featureDetection");
  }
  // private featureDetection(const Frame&, need_n_corners) (FeatureDetector.cpp:174-299):
  // `tracked` = keypoints of the frame whose landmark id is not -1
  KeypointsCV featureDetection(const ImageView& img, const KeypointsCV& tracked, int need_n_corners) const {
    KeypointsCV out(8192);
    int32_t n = 0;
    c_.check(kvfe_feature_detection(c_.get(), img.data, img.step, &tracked.data()->x,
                                    (int32_t)tracked.size(), need_n_corners, &out.data()->x,
                                    (int32_t)out.size(), &n),
             "featureDetection");
    out.resize(n);
    return out;
  }

 private:
  Context c_;
};

// Tracker (src/frontend/Tracker.cpp): featureTracking core (:92-211: prediction + pyramidal LK) and the
// IMU-aided geometric outlier rejection (:213-661)
class Tracker {
 public:
  explicit Tracker(Context ctx) : c_(std::move(ctx)) {}
  // px_ref: valid reference keypoints; ref_R_cur row-major 3x3 (gtsam::Rot3::matrix()).
  // Returns px_cur, status (1 = tracked) and error, as cv::calcOpticalFlowPyrLK does.
  void featureTracking(const ImageView& ref_img, const ImageView& cur_img, const KeypointsCV& px_ref,
                       const double ref_R_cur[9], KeypointsCV* px_cur, std::vector<uint8_t>* status,
                       std::vector<float>* error) const {
    const int32_t n = (int32_t)px_ref.size();
    px_cur->resize(n);
    status->resize(n);
    error->resize(n);
    c_.check(kvfe_predict_sparse_flow(c_.get(), &px_ref.data()->x, n, ref_R_cur, &px_cur->data()->x),
             "predictSparseFlow");
    c_.check(kvfe_calc_optical_flow_pyr_lk(c_.get(), ref_img.data, cur_img.data, ref_img.step,
                                           &px_ref.data()->x, &px_cur->data()->x, n, status->data(),
                                           error->data()),
             "calcOpticalFlowPyrLK");
  }

  // void featureTracking(Frame* ref_frame, Frame* cur_frame, const gtsam::Rot3& ref_R_cur,
  //                      const FeatureDetectorParams&, std::optional<cv::Mat> R) (Tracker.h:70-74, Tracker.cpp:92-211)
  // incl. the survivor bookkeeping: ids / ages / keypoints / versors pushed to cur_frame, ref_frame->landmarks_[i] = -1
  // for lost or too old tracks.  ref_R_cur: gtsam::Rot3::matrix() row-major; R: the context's (see FeatureDetector).
  void featureTracking(Frame* ref_frame, Frame* cur_frame, const double ref_R_cur[9]) const {
    if (!cur_frame->keypoints_.empty() || !cur_frame->landmarks_.empty())   // CHECK(cur_frame->keypoints_.empty())
      throw Error(KVFE_ERR_INVALID_ARG, "featureTracking: cur_frame must be empty");
    const int cap = detail::frame_capacity(c_.config());
    const int n = (int)ref_frame->landmarks_.size();
    kvfe_frame r = detail::frame_view(ref_frame, n, n > 0 ? n : 1);
    kvfe_frame k = detail::frame_view(cur_frame, 0, cap);
    const kvfe_status s = kvfe_feature_tracking_frame(c_.get(), ref_frame->img_.data, cur_frame->img_.data,
                                                      ref_frame->img_.step, &r, &k, ref_R_cur);
    detail::frame_shrink(ref_frame, n);
    detail::frame_shrink(cur_frame, s == KVFE_OK ? k.n_keypoints : 0);
    c_.check(s, "featureTracking");
  }

  // TrackingStatusPose (Tracker-definitions.h:126-133) with the pose as row-major 3x4 [R | t]
  struct TrackingStatusPose {
    int status = KVFE_TRACKING_INVALID;  // VIO::TrackingStatus
    double pose[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    double info[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // stereo: infoMatStereoTranslation_
  };
  // geometricOutlierRejection2d2d(ref_bearings, cur_bearings, matches, inliers, cam_lkf_Pose_cam_kf)
  // with ransac_use_2point_mono_ (Tracker.cpp:213-318); f_ref / f_cur: the matched bearing vectors
  // (n x 3), R: cam_lkf_Pose_cam_kf.rotation().matrix() row-major
  TrackingStatusPose geometricOutlierRejection2d2d(const std::vector<double>& f_ref,
                                                   const std::vector<double>& f_cur, const double R[9],
                                                   std::vector<int>* inliers) const {
    const int32_t n = (int32_t)(f_ref.size() / 3);
    std::vector<int32_t> inl((size_t)(n > 0 ? n : 1));
    kvfe_ransac_output o;
    c_.check(kvfe_outlier_rejection_2d2d_given_rotation(c_.get(), f_ref.data(), f_cur.data(), n, R,
                                                        inl.data(), &o),
             "geometricOutlierRejection2d2d");
    inliers->assign(inl.begin(), inl.begin() + o.n_inliers);
    TrackingStatusPose r;
    r.status = o.status;
    std::memcpy(r.pose, o.pose, sizeof(r.pose));
    return r;
  }
  // geometricOutlierRejection3d3dGivenRotation (Tracker.cpp:382-632) on the stereo matches: per match
  // the rectified left pixel, the rectified right x and keypoints_3d_ of the reference (last
  // keyframe) and current frame
  TrackingStatusPose geometricOutlierRejection3d3dGivenRotation(
      const KeypointsCV& ref_left_rect, const std::vector<float>& ref_right_x,
      const std::vector<double>& ref_points_3d, const KeypointsCV& cur_left_rect,
      const std::vector<float>& cur_right_x, const std::vector<double>& cur_points_3d, const double R[9],
      std::vector<int>* inliers) const {
    const int32_t n = (int32_t)ref_left_rect.size();
    std::vector<int32_t> inl((size_t)(n > 0 ? n : 1));
    kvfe_ransac_output o;
    c_.check(kvfe_outlier_rejection_3d3d_given_rotation(
                 c_.get(), n ? &ref_left_rect.data()->x : nullptr, ref_right_x.data(), ref_points_3d.data(),
                 n ? &cur_left_rect.data()->x : nullptr, cur_right_x.data(), cur_points_3d.data(), n, R,
                 inl.data(), &o),
             "geometricOutlierRejection3d3dGivenRotation");
    inliers->assign(inl.begin(), inl.begin() + o.n_inliers);
    TrackingStatusPose r;
    r.status = o.status;
    std::memcpy(r.pose, o.pose, sizeof(r.pose));
    std::memcpy(r.info, o.info, sizeof(r.info));
    return r;
  }
  // geometricOutlierRejection2d2d without rotation prior (Tracker.cpp:262-275): 5-point Nister RANSAC
  TrackingStatusPose geometricOutlierRejection2d2d(const std::vector<double>& f_ref,
                                                   const std::vector<double>& f_cur,
                                                   std::vector<int>* inliers) const {
    const int32_t n = (int32_t)(f_ref.size() / 3);
    std::vector<int32_t> inl((size_t)(n > 0 ? n : 1));
    kvfe_ransac_output o;
    c_.check(kvfe_outlier_rejection_2d2d(c_.get(), f_ref.data(), f_cur.data(), n, inl.data(), &o),
             "geometricOutlierRejection2d2d");
    inliers->assign(inl.begin(), inl.begin() + o.n_inliers);
    TrackingStatusPose r;
    r.status = o.status;
    std::memcpy(r.pose, o.pose, sizeof(r.pose));
    std::memset(r.info, 0, sizeof(r.info));
    return r;
  }
  // geometricOutlierRejection3d3d (Tracker.cpp:667-742): 3-point Arun RANSAC on the matched keypoints_3d_
  TrackingStatusPose geometricOutlierRejection3d3d(const std::vector<double>& ref_points_3d,
                                                   const std::vector<double>& cur_points_3d,
                                                   std::vector<int>* inliers) const {
    const int32_t n = (int32_t)(ref_points_3d.size() / 3);
    std::vector<int32_t> inl((size_t)(n > 0 ? n : 1));
    kvfe_ransac_output o;
    c_.check(kvfe_outlier_rejection_3d3d(c_.get(), ref_points_3d.data(), cur_points_3d.data(), n, inl.data(), &o),
             "geometricOutlierRejection3d3d");
    inliers->assign(inl.begin(), inl.begin() + o.n_inliers);
    TrackingStatusPose r;
    r.status = o.status;
    std::memcpy(r.pose, o.pose, sizeof(r.pose));
    std::memset(r.info, 0, sizeof(r.info));
    return r;
  }

  // ---- PnP tracking (use_pnp_tracking) -----------------------------------------------------------------------------
  // LandmarksMap (common/vio_types.h): landmark id -> optimised 3-D position in the world frame
  using LandmarksMap = std::unordered_map<int64_t, std::array<double, 3>>;
  // void updateMap(const LandmarksMap&) (Tracker.h:82-94): the backend's time horizon, copied under a mutex upstream
  void updateMap(const LandmarksMap& lmks_map) {
    std::lock_guard<std::mutex> lock(landmarks_map_mtx_);
    landmarks_map_ = lmks_map;
  }
  // bool pnp(const BearingVectors& cam_bearing_vectors, const Landmarks& F_points, gtsam::Pose3* F_Pose_cam_estimate,
  //          std::vector<int>* inliers, gtsam::Pose3* F_Pose_cam_prior) (Tracker.cpp:1122-1288); bearings / points
  // n x 3, pose row-major 3x4 [R | t].  `status` (optional) receives outlierRejectionPnP's TrackingStatus.
  bool pnp(const std::vector<double>& cam_bearing_vectors, const std::vector<double>& F_points,
           const kvfe_pnp_params& params, double F_Pose_cam_estimate[12], std::vector<int>* inliers,
           int* status = nullptr) const {
    const int32_t n = (int32_t)(F_points.size() / 3);
    if (cam_bearing_vectors.size() != F_points.size())   // CHECK_EQ(cam_bearing_vectors.size(), F_points.size())
      throw Error(KVFE_ERR_INVALID_ARG, "pnp: bearing vectors and points differ in number");
    std::vector<int32_t> inl((size_t)(n > 0 ? n : 1));
    kvfe_ransac_output o;
    c_.check(kvfe_pnp(c_.get(), &params, cam_bearing_vectors.data(), F_points.data(), n, inl.data(), &o), "pnp");
    if (inliers) inliers->assign(inl.begin(), inl.begin() + o.n_inliers);
    std::memcpy(F_Pose_cam_estimate, o.pose, sizeof(o.pose));
    if (status) *status = o.status;
    return o.reserved0 != 0;
  }
  // bool pnp(const StereoFrame& cur_stereo_frame, gtsam::Pose3* W_Pose_cam_estimate, std::vector<int>* inliers, ...)
  // (Tracker.cpp:1064-1120): the 2D-3D correspondences are the keypoints with a VALID rectified left keypoint whose
  // landmark id is in the map of updateMap.  The frame is passed as the three members the reference reads:
  // left_keypoints_rectified_[i].first, left_frame_.landmarks_[i], keypoints_3d_[i] (n x 3) -- e.g. left_status /
  // landmarks / keypoints_3d of kvfe_frame_output.
  bool pnp(const uint8_t* left_keypoints_rectified_status, const int64_t* landmarks, const double* keypoints_3d,
           size_t n, const kvfe_pnp_params& params, double W_Pose_cam_estimate[12], std::vector<int>* inliers,
           int* status = nullptr) const {
    LandmarksMap copy_W_landmarks_map;
    {
      std::lock_guard<std::mutex> lock(landmarks_map_mtx_);
      copy_W_landmarks_map = landmarks_map_;
    }
    std::vector<double> cam_bearing_vectors, W_points;
    for (size_t i = 0; i < n; i++) {
      const int64_t lmk_id = landmarks[i];
      if (left_keypoints_rectified_status[i] == KVFE_KP_VALID && lmk_id != -1) {
        const auto it = copy_W_landmarks_map.find(lmk_id);
        if (it != copy_W_landmarks_map.end()) {
          W_points.insert(W_points.end(), it->second.begin(), it->second.end());
          cam_bearing_vectors.insert(cam_bearing_vectors.end(), keypoints_3d + 3 * i, keypoints_3d + 3 * i + 3);
        }
      }
    }
    return pnp(cam_bearing_vectors, W_points, params, W_Pose_cam_estimate, inliers, status);
  }
  // VisionImuFrontend::outlierRejectionPnP(frame, &status_pnp) (VisionImuFrontend.cpp:146-173) on the output of a
  // keyframe step: fills kfTracking_status_pnp_ / W_T_k_pnp_ of the TrackerStatusSummary
  TrackingStatusPose outlierRejectionPnP(const kvfe_frame_output& frame, const kvfe_pnp_params& params) const {
    TrackingStatusPose r;
    std::vector<int> inliers;
    pnp(frame.left_status, frame.landmarks, frame.keypoints_3d, (size_t)frame.n_keypoints, params, r.pose, &inliers,
        &r.status);
    return r;
  }

 private:
  Context c_;
  mutable std::mutex landmarks_map_mtx_;
  LandmarksMap landmarks_map_;
};

// StereoMatcher (src/frontend/StereoMatcher.cpp:123-483)
class StereoMatcher {
 public:
  explicit StereoMatcher(Context ctx) : c_(std::move(ctx)) {}
  struct SparseResult {
    StatusKeypointsCV left_keypoints_rectified, right_keypoints_rectified;
    std::vector<double> keypoints_depth;     // n
    KeypointsCV right_keypoints;             // distorted right pixels
    std::vector<double> keypoints_3d;        // n x 3
  };
  // sparseStereoReconstruction(StereoFrame*) (StereoMatcher.cpp:123-175)
  SparseResult sparseStereoReconstruction(const ImageView& left, const ImageView& right,
                                          const KeypointsCV& left_keypoints) const {
    const int32_t n = (int32_t)left_keypoints.size();
    std::vector<KeypointCV> lr(n), rr(n);
    std::vector<uint8_t> ls(n), rs(n);
    SparseResult r;
    r.keypoints_depth.resize(n);
    r.right_keypoints.resize(n);
    r.keypoints_3d.resize((size_t)n * 3);
    kvfe_stereo_output o;
    std::memset(&o, 0, sizeof(o));
    o.left_rect_xy = &lr.data()->x;
    o.left_status = ls.data();
    o.right_rect_xy = &rr.data()->x;
    o.right_status = rs.data();
    o.depth = r.keypoints_depth.data();
    o.right_xy = &r.right_keypoints.data()->x;
    o.keypoints_3d = r.keypoints_3d.data();
    c_.check(kvfe_sparse_stereo_reconstruction(c_.get(), left.data, right.data, left.step,
                                               &left_keypoints.data()->x, n, &o),
             "sparseStereoReconstruction");
    r.left_keypoints_rectified.resize(n);
    r.right_keypoints_rectified.resize(n);
    for (int32_t i = 0; i < n; i++) {
      r.left_keypoints_rectified[i] = {static_cast<KeypointStatus>(ls[i]), lr[i]};
      r.right_keypoints_rectified[i] = {static_cast<KeypointStatus>(rs[i]), rr[i]};
    }
    return r;
  }
  // getDepthFromRectifiedMatches(left_keypoints_rectified, right_keypoints_rectified, keypoints_depth)
  // (StereoMatcher.h:85-92, StereoMatcher.cpp:425-483); right statuses are updated in place as upstream
  void getDepthFromRectifiedMatches(StatusKeypointsCV& left_keypoints_rectified,
                                    StatusKeypointsCV& right_keypoints_rectified,
                                    std::vector<double>* keypoints_depth) const {
    if (left_keypoints_rectified.size() != right_keypoints_rectified.size())   // CHECK_EQ upstream
      throw Error(KVFE_ERR_INVALID_ARG, "getDepthFromRectifiedMatches: size mismatch");
    const int32_t n = (int32_t)left_keypoints_rectified.size();
    KeypointsCV l(n), r(n);
    std::vector<uint8_t> ls(n), rs(n);
    for (int32_t i = 0; i < n; i++) {
      ls[i] = static_cast<uint8_t>(left_keypoints_rectified[i].first);
      l[i] = left_keypoints_rectified[i].second;
      rs[i] = static_cast<uint8_t>(right_keypoints_rectified[i].first);
      r[i] = right_keypoints_rectified[i].second;
    }
    keypoints_depth->assign(n, 0.0);
    if (n == 0) return;
    c_.check(kvfe_get_depth_from_rectified_matches(c_.get(), &l.data()->x, ls.data(), &r.data()->x, rs.data(), n,
                                                   keypoints_depth->data()),
             "getDepthFromRectifiedMatches");
    for (int32_t i = 0; i < n; i++) right_keypoints_rectified[i].first = static_cast<KeypointStatus>(rs[i]);
  }
  // denseStereoReconstruction(left_img_rectified, right_img_rectified, disparity_img)
  // (StereoMatcher.cpp:32-121).  disparity: h x w int16 with 4 fractional bits — the CV_16S matrix
  // cv::StereoSGBM::compute leaves in *disparity_img; dense_stereo_params_ defaults to the reference's
  // member initialisers (StereoMatchingParams.h:39-58).
  void denseStereoReconstruction(const ImageView& left_img_rectified, const ImageView& right_img_rectified,
                                 int16_t* disparity, size_t disparity_stride_elems) const {
    kvfe_dense_stereo_params p = dense_stereo_params_;
    const uint8_t* l = left_img_rectified.data;
    const uint8_t* r = right_img_rectified.data;
    c_.check(kvfe_dense_stereo_reconstruction(c_.get(), &p, 1, &l, &r, left_img_rectified.step, &disparity,
                                              disparity_stride_elems),
             "denseStereoReconstruction");
  }
  kvfe_dense_stereo_params dense_stereo_params_ = [] {
    kvfe_dense_stereo_params p;
    kvfe_dense_stereo_params_default(&p);
    return p;
  }();
  // getRightKeypointsRectified (StereoMatcher.cpp:196-281) on already rectified images
  void getRightKeypointsRectified(const ImageView& left_rectified, const ImageView& right_rectified,
                                  const StatusKeypointsCV& left_keypoints_rectified,
                                  StatusKeypointsCV* right_keypoints_rectified) const {
    const int32_t n = (int32_t)left_keypoints_rectified.size();
    std::vector<KeypointCV> l(n), r(n);
    std::vector<uint8_t> ls(n), rs(n);
    for (int32_t i = 0; i < n; i++) {
      ls[i] = static_cast<uint8_t>(left_keypoints_rectified[i].first);
      l[i] = left_keypoints_rectified[i].second;
    }
    c_.check(kvfe_get_right_keypoints_rectified(c_.get(), left_rectified.data, right_rectified.data,
                                                left_rectified.step, &l.data()->x, ls.data(), n,
                                                &r.data()->x, rs.data(), nullptr),
             "getRightKeypointsRectified");
    right_keypoints_rectified->resize(n);
    for (int32_t i = 0; i < n; i++)
      (*right_keypoints_rectified)[i] = {static_cast<KeypointStatus>(rs[i]), r[i]};
  }

 private:
  Context c_;
};

// StereoVisionImuFrontend::spinOnce visual path (StereoVisionImuFrontend.cpp:102-481) for
// `batch` independent streams.  IMU preintegration stays on the host: the caller passes
// keyframe_R_cur_frame (camLrectLkf_R_camLrectK_imu) per stream.
class StereoVisionImuFrontend {
 public:
  // getRelativePoseBodyMono / getRelativePoseBodyStereo (StereoVisionImuFrontend.cpp:699-716) of one stream's output
  static Pose3 getRelativePoseBodyMono(const kvfe_frame_output& out, const Pose3& body_Pose_camLrect) {
    return relativePoseBody(body_Pose_camLrect, out.lkf_T_k_mono);
  }
  static Pose3 getRelativePoseBodyStereo(const kvfe_frame_output& out, const Pose3& body_Pose_camLrect) {
    return relativePoseBody(body_Pose_camLrect, out.lkf_T_k_stereo);
  }
  explicit StereoVisionImuFrontend(Context ctx) : c_(std::move(ctx)) {}
  // images: `batch` images back to back (host memory)
  void spinOnce(const uint8_t* left, const uint8_t* right, size_t row_stride, size_t image_stride,
                const kvfe_frame_input* inputs) {
    c_.check(kvfe_frontend_step_host(c_.get(), left, right, row_stride, image_stride, inputs), "spinOnce");
  }
  // same with device pointers (batched many-sequence mode, inputs already in HBM)
  void spinOnceDevice(const void* left_dev, const void* right_dev, size_t row_stride,
                      size_t image_stride, const kvfe_frame_input* inputs) {
    c_.check(kvfe_frontend_step_device(c_.get(), left_dev, right_dev, row_stride, image_stride, inputs),
             "spinOnceDevice");
  }
  // StereoFrontendOutput of one stream; arrays sized by out->capacity
  void getOutput(int stream, kvfe_frame_output* out) {
    c_.check(kvfe_frontend_get_output(c_.get(), stream, out), "getOutput");
  }
  void reset() { c_.check(kvfe_frontend_reset(c_.get()), "reset"); }
  // Tracker::updateMap through the front-end (use_pnp_tracking: the step's outlierRejectionPnP reads this map)
  void updateMap(int stream, const Tracker::LandmarksMap& lmks_map) {
    std::vector<int64_t> ids;
    std::vector<double> xyz;
    ids.reserve(lmks_map.size());
    xyz.reserve(3 * lmks_map.size());
    for (const auto& it : lmks_map) {
      ids.push_back(it.first);
      xyz.insert(xyz.end(), it.second.begin(), it.second.end());
    }
    c_.check(kvfe_frontend_update_map(c_.get(), stream, ids.data(), xyz.data(), (int32_t)ids.size()), "updateMap");
  }

 private:
  Context c_;
};

// MonoVisionImuFrontend::spinOnce visual path (src/frontend/MonoVisionImuFrontend.cpp:196-369) for `batch`
// independent streams: create the context with kvfe_config.frontend_type = KVFE_FRONTEND_MONO (one camera: R = I,
// P = K; measurements are (u, NaN, v), getSmartMonoMeasurements :340-369).
class MonoVisionImuFrontend {
 public:
  explicit MonoVisionImuFrontend(Context ctx) : c_(std::move(ctx)) {
    if (c_.config().frontend_type != KVFE_FRONTEND_MONO)
      throw Error(KVFE_ERR_INVALID_ARG, "MonoVisionImuFrontend needs a KVFE_FRONTEND_MONO context");
  }
  static kvfe_config config(const kvfe_camera_params& cam, const kvfe_frontend_params& params, int batch = 1,
                            int device = 0) {
    kvfe_config cfg;
    std::memset(&cfg, 0, sizeof(cfg));
    cfg.left = cam;
    cfg.right = cam;
    cfg.params = params;
    cfg.batch = batch;
    cfg.device = device;
    cfg.frontend_type = KVFE_FRONTEND_MONO;
    return cfg;
  }
  void spinOnce(const uint8_t* img, size_t row_stride, size_t image_stride, const kvfe_frame_input* inputs) {
    c_.check(kvfe_frontend_step_host(c_.get(), img, nullptr, row_stride, image_stride, inputs), "spinOnce");
  }
  void getOutput(int stream, kvfe_frame_output* out) {
    c_.check(kvfe_frontend_get_output(c_.get(), stream, out), "getOutput");
  }
  void reset() { c_.check(kvfe_frontend_reset(c_.get()), "reset"); }

 private:
  Context c_;
};

// RgbdVisionImuFrontend::spinOnce visual path (RgbdVisionImuFrontend.cpp:184-431) for `batch` independent
// streams: create the context with kvfe_config.frontend_type = KVFE_FRONTEND_RGBD and kvfe_config.depth =
// CameraParams::DepthParams.  `depth` are the registered depth images (uint16 or float per depth_type) with
// the geometry of the intensity images in elements.
class RgbdVisionImuFrontend {
 public:
  explicit RgbdVisionImuFrontend(Context ctx) : c_(std::move(ctx)) {}
  void spinOnce(const uint8_t* intensity, const void* depth, size_t row_stride, size_t image_stride,
                const kvfe_frame_input* inputs) {
    c_.check(kvfe_frontend_step_host(c_.get(), intensity, static_cast<const uint8_t*>(depth), row_stride,
                                     image_stride, inputs),
             "spinOnce");
  }
  void getOutput(int stream, kvfe_frame_output* out) {
    c_.check(kvfe_frontend_get_output(c_.get(), stream, out), "getOutput");
  }
  void reset() { c_.check(kvfe_frontend_reset(c_.get()), "reset"); }

 private:
  Context c_;
};

// ------------------------------------------------------------------------------------------------------------
// Input side: what feeds spinOnce.  Host code of the library; see kvfe.h for the semantics.
// ------------------------------------------------------------------------------------------------------------

// UtilsOpenCV::ReadAndConvertToGrayScale(img_name, equalize = false) for PNG / baseline JPEG data in memory: the decoded
// 8-bit grey image (rows * cols bytes, tightly packed).  Equalisation is the front-end's equalize_image parameter
// / kvfe_equalize_hist.
inline std::vector<uint8_t> ReadAndConvertToGrayScale(const uint8_t* png, size_t size, int* rows, int* cols) {
  int32_t w = 0, h = 0, c = 0;
  const bool jpeg = size > 2 && png[0] == 0xFF && png[1] == 0xD8;   // baseline JPEG (kvfe_jpeg_*) or PNG (kvfe_png_*)
  kvfe_status st = jpeg ? kvfe_jpeg_info(png, size, &w, &h, &c) : kvfe_png_info(png, size, &w, &h, &c);
  if (st != KVFE_OK) throw Error(st, "ReadAndConvertToGrayScale: not a PNG / baseline JPEG file this library decodes");
  std::vector<uint8_t> img((size_t)w * h);
  st = jpeg ? kvfe_jpeg_decode_gray(png, size, img.data(), (size_t)w, w, h)
            : kvfe_png_decode_gray(png, size, img.data(), (size_t)w, w, h);
  if (st != KVFE_OK) throw Error(st, "ReadAndConvertToGrayScale: corrupt image data");
  if (rows) *rows = h;
  if (cols) *cols = w;
  return img;
}

// ImuStampS (1 x n int64) / ImuAccGyrS (6 x n, column k = sample k) as plain vectors
struct ImuMeasurements {
  std::vector<int64_t> timestamps;
  std::vector<double> acc_gyr;   // 6 per sample: ax ay az wx wy wz
  int cols() const { return (int)timestamps.size(); }
};

namespace utils {
class ThreadsafeImuBuffer {
 public:
  enum class QueryResult {
    kDataAvailable = KVFE_IMU_DATA_AVAILABLE,
    kDataNotYetAvailable = KVFE_IMU_DATA_NOT_YET_AVAILABLE,
    kDataNeverAvailable = KVFE_IMU_DATA_NEVER_AVAILABLE,
    kQueueShutdown = KVFE_IMU_QUEUE_SHUTDOWN,
    kTooFewMeasurementsAvailable = KVFE_IMU_TOO_FEW_MEASUREMENTS
  };
  explicit ThreadsafeImuBuffer(int64_t buffer_length_ns) : b_(kvfe_imu_buffer_create(buffer_length_ns)) {
    if (!b_) throw std::bad_alloc();
  }
  ~ThreadsafeImuBuffer() { kvfe_imu_buffer_destroy(b_); }
  ThreadsafeImuBuffer(const ThreadsafeImuBuffer&) = delete;
  ThreadsafeImuBuffer& operator=(const ThreadsafeImuBuffer&) = delete;

  void addMeasurement(int64_t timestamp_ns, const double acc_gyr[6]) { kvfe_imu_buffer_add(b_, timestamp_ns, acc_gyr); }
  size_t size() const { return (size_t)kvfe_imu_buffer_size(b_); }
  void shutdown() { kvfe_imu_buffer_shutdown(b_); }
  QueryResult getImuDataBtwTimestamps(int64_t from, int64_t to, ImuMeasurements* out, bool get_lower_bound = false) {
    return query(out, [&](int64_t* t, double* v, int32_t cap, int32_t* n) {
      return kvfe_imu_buffer_between(b_, from, to, get_lower_bound ? 1 : 0, t, v, cap, n);
    });
  }
  QueryResult getImuDataInterpolatedUpperBorder(int64_t from, int64_t to, ImuMeasurements* out) {
    return query(out, [&](int64_t* t, double* v, int32_t cap, int32_t* n) {
      return kvfe_imu_buffer_interpolated_upper_border(b_, from, to, t, v, cap, n);
    });
  }
  QueryResult getImuDataInterpolatedBorders(int64_t from, int64_t to, ImuMeasurements* out) {
    return query(out, [&](int64_t* t, double* v, int32_t cap, int32_t* n) {
      return kvfe_imu_buffer_interpolated_borders(b_, from, to, t, v, cap, n);
    });
  }
  static void linearInterpolate(int64_t t0, const double y0[6], int64_t t1, const double y1[6], int64_t t,
                                double y[6]) {
    kvfe_imu_linear_interpolate(t0, y0, t1, y1, t, y);
  }

 private:
  template <class F>
  static QueryResult query(ImuMeasurements* out, F f) {
    int32_t cap = 64, n = 0;
    for (;;) {
      out->timestamps.resize(cap);
      out->acc_gyr.resize((size_t)6 * cap);
      const int32_t r = f(out->timestamps.data(), out->acc_gyr.data(), cap, &n);
      if (r == -1) {
        cap = n;
        continue;
      }
      out->timestamps.resize(n);
      out->acc_gyr.resize((size_t)6 * n);
      return static_cast<QueryResult>(r);
    }
  }
  kvfe_imu_buffer* b_;
};
}  // namespace utils

// ImuFrontend::preintegrateImuMeasurements as far as the front-end needs it (src/imu-frontend/ImuFrontend.cpp:158-173):
// the gyro rotation since the last keyframe, advanced by one packet's samples; resetIntegration() at a keyframe
// (ImuFrontend::resetIntegrationWithCachedBias).  keyframeRcurFrame() is the kvfe_frame_input::keyframe_R_cur_frame of
// StereoVisionImuFrontend.cpp:143-150.
class ImuRotationPreintegrator {
 public:
  explicit ImuRotationPreintegrator(const double gyro_bias[3] = nullptr) {
    for (int i = 0; i < 3; i++) bias_[i] = gyro_bias ? gyro_bias[i] : 0.0;
    resetIntegration();
  }
  void resetIntegration() {
    for (int i = 0; i < 9; i++) deltaRij_[i] = (i % 4 == 0) ? 1.0 : 0.0;
  }
  void updateBias(const double gyro_bias[3]) {
    for (int i = 0; i < 3; i++) bias_[i] = gyro_bias[i];
  }
  void preintegrateImuMeasurements(const ImuMeasurements& imu) {
    const kvfe_status st = kvfe_imu_preintegrate_rotation(imu.timestamps.data(), imu.acc_gyr.data(), imu.cols(), bias_, deltaRij_);
    if (st != KVFE_OK) throw Error(st, "preintegrateImuMeasurements: fewer than two samples or a non-positive time step");
  }
  const double* deltaRij() const { return deltaRij_; }
  void keyframeRcurFrame(const Pose3& body_Pose_camLrect, double out[9]) const {
    kvfe_keyframe_R_cur_frame(body_Pose_camLrect.R, deltaRij_, out);
  }

 private:
  double bias_[3], deltaRij_[9];
};

// StereoImuSyncPacket (include/kimera-vio/frontend/StereoImuSyncPacket.h:81-107) without the images: the frames
// are referred to by the tags the caller queued them with
struct StereoImuSyncPacket {
  int64_t timestamp = 0;
  int64_t left_frame_tag = 0, right_frame_tag = 0;
  ImuMeasurements imu;
};

// StereoDataProviderModule in sequential mode: fill the queues, getInputPacket() = one spin
class StereoDataProviderModule {
 public:
  explicit StereoDataProviderModule(int64_t imu_buffer_length_ns = -1, int mode = KVFE_SYNC_MODE_STEREO)
      : s_(kvfe_stereo_sync_create(imu_buffer_length_ns)) {
    if (!s_) throw std::bad_alloc();
    kvfe_stereo_sync_set_mode(s_, mode);
  }
  ~StereoDataProviderModule() { kvfe_stereo_sync_destroy(s_); }
  StereoDataProviderModule(const StereoDataProviderModule&) = delete;
  StereoDataProviderModule& operator=(const StereoDataProviderModule&) = delete;

  void fillLeftFrameQueue(int64_t timestamp_ns, int64_t tag) { kvfe_stereo_sync_fill_left(s_, timestamp_ns, tag); }
  void fillRightFrameQueue(int64_t timestamp_ns, int64_t tag) { kvfe_stereo_sync_fill_right(s_, timestamp_ns, tag); }
  void fillImuQueue(int64_t timestamp_ns, const double acc_gyr[6]) { kvfe_stereo_sync_fill_imu(s_, timestamp_ns, acc_gyr); }
  void doCoarseImuCameraTemporalSync() { kvfe_stereo_sync_do_coarse_imu_camera_temporal_sync(s_); }
  void setImuTimeShift(double imu_time_shift_s) { kvfe_stereo_sync_set_imu_time_shift(s_, imu_time_shift_s); }
  void shutdown() { kvfe_stereo_sync_shutdown(s_); }
  // true: *packet written; false: nothing this spin -- lastAction() is the KVFE_SYNC_* reason
  bool getInputPacket(StereoImuSyncPacket* packet) {
    int32_t cap = 64;
    for (;;) {
      packet->imu.timestamps.resize(cap);
      packet->imu.acc_gyr.resize((size_t)6 * cap);
      kvfe_sync_packet pk;
      const int32_t r = kvfe_stereo_sync_next(s_, &pk, packet->imu.timestamps.data(), packet->imu.acc_gyr.data(), cap);
      if (r == -1) {
        cap = pk.n_imu;
        continue;
      }
      last_action_ = r;
      packet->imu.timestamps.resize(r == KVFE_SYNC_PACKET ? pk.n_imu : 0);
      packet->imu.acc_gyr.resize(r == KVFE_SYNC_PACKET ? (size_t)6 * pk.n_imu : 0);
      if (r != KVFE_SYNC_PACKET) return false;
      packet->timestamp = pk.timestamp_ns;
      packet->left_frame_tag = pk.left_tag;
      packet->right_frame_tag = pk.right_tag;
      return true;
    }
  }
  int lastAction() const { return last_action_; }

 private:
  kvfe_stereo_sync* s_;
  int last_action_ = KVFE_SYNC_EMPTY;
};

// MonoDataProviderModule (MonoDataProviderModule.cpp:30-118): no second queue, packets carry right_frame_tag = -1
class MonoDataProviderModule : public StereoDataProviderModule {
 public:
  explicit MonoDataProviderModule(int64_t imu_buffer_length_ns = -1)
      : StereoDataProviderModule(imu_buffer_length_ns, KVFE_SYNC_MODE_MONO) {}
};
// RgbdDataProviderModule (RgbdDataProviderModule.cpp:44-84): the second queue holds the depth frames
class RgbdDataProviderModule : public StereoDataProviderModule {
 public:
  explicit RgbdDataProviderModule(int64_t imu_buffer_length_ns = -1)
      : StereoDataProviderModule(imu_buffer_length_ns, KVFE_SYNC_MODE_RGBD) {}
  void fillDepthFrameQueue(int64_t timestamp_ns, int64_t tag) { fillRightFrameQueue(timestamp_ns, tag); }
};

// EurocDataProvider (include/kimera-vio/dataprovider/EurocDataProvider.h; src/dataprovider/EurocDataProvider.cpp:
// 108-195 spin / spinOnce, 229-306 parseImuData, 437-482 parseDataset / parseCameraData, 741-751 clipFinalFrame) in
// sequential mode: parses <dataset_path>/mav0/{cam0,cam1,imu0}/data.csv, spin() sends every IMU sample once and then
// the frame pairs initial_k .. final_k - 1, decoded to 8-bit grey, through the callbacks.
class EurocDataProvider {
 public:
  using ImuCallback = std::function<void(int64_t timestamp, const double acc_gyr[6])>;
  using FrameCallback = std::function<void(int64_t k, int64_t timestamp, const uint8_t* img, int rows, int cols)>;

  EurocDataProvider(const std::string& dataset_path, int64_t initial_k, int64_t final_k) : path_(dataset_path) {
    imu_ = parse_imu(read(path_ + "/mav0/imu0/data.csv"));
    left_ = parse_cam(read(path_ + "/mav0/cam0/data.csv"));
    right_ = parse_cam(read(path_ + "/mav0/cam1/data.csv"));
    initial_k_ = initial_k;
    final_k_ = std::min<int64_t>(final_k, (int64_t)left_.size());   // clipFinalFrame
    if (!(final_k_ > initial_k_)) throw Error(KVFE_ERR_INVALID_ARG, "EurocDataProvider: final_k must exceed initial_k");
    current_k_ = initial_k_;
  }
  void registerImuSingleCallback(ImuCallback cb) { imu_cb_ = std::move(cb); }
  void registerLeftFrameCallback(FrameCallback cb) { left_cb_ = std::move(cb); }
  void registerRightFrameCallback(FrameCallback cb) { right_cb_ = std::move(cb); }
  size_t getNumImages() const { return left_.size(); }
  int64_t timestampAtFrame(int64_t k) const { return left_.at((size_t)k); }
  bool hasData() const { return current_k_ < final_k_; }
  std::string getLeftImgName(int64_t k) const { return img_name("cam0", left_, k); }
  std::string getRightImgName(int64_t k) const { return img_name("cam1", right_, k); }

  bool spinOnce() {
    if (current_k_ >= final_k_) return false;
    const int64_t k = current_k_, t = timestampAtFrame(k);
    std::vector<uint8_t> lf, rf;
    const std::string ln = getLeftImgName(k), rn = getRightImgName(k);
    if (!ln.empty() && !rn.empty() && try_read(ln, &lf) && try_read(rn, &rf)) {
      int r0 = 0, c0 = 0, r1 = 0, c1 = 0;
      const std::vector<uint8_t> li = ReadAndConvertToGrayScale(lf.data(), lf.size(), &r0, &c0);
      const std::vector<uint8_t> ri = ReadAndConvertToGrayScale(rf.data(), rf.size(), &r1, &c1);
      if (left_cb_) left_cb_(k, t, li.data(), r0, c0);
      if (right_cb_) right_cb_(k, t, ri.data(), r1, c1);
    }   // else: "Missing left/right stereo pair, proceeding to the next one."
    current_k_++;
    return true;
  }
  bool spin() {
    if (!imu_sent_) {
      if (imu_cb_)
        for (size_t i = 0; i < imu_.timestamps.size(); i++) imu_cb_(imu_.timestamps[i], &imu_.acc_gyr[6 * i]);
      imu_sent_ = true;
    }
    while (spinOnce()) {
    }
    return false;
  }
  const ImuMeasurements& imuMeasurements() const { return imu_; }

 private:
  static bool try_read(const std::string& name, std::vector<uint8_t>* out) {
    std::FILE* f = std::fopen(name.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out->resize(n > 0 ? (size_t)n : 0);
    const size_t got = out->empty() ? 0 : std::fread(out->data(), 1, out->size(), f);
    std::fclose(f);
    return got == out->size();
  }
  static std::vector<uint8_t> read(const std::string& name) {
    std::vector<uint8_t> d;
    if (!try_read(name, &d)) throw Error(KVFE_ERR_INVALID_ARG, "EurocDataProvider: cannot open file: " + name);
    return d;
  }
  static std::vector<int64_t> parse_cam(const std::vector<uint8_t>& text) {
    int32_t n = 0;
    kvfe_status st = kvfe_euroc_parse_camera_csv(reinterpret_cast<const char*>(text.data()), text.size(), nullptr, 0, &n);
    if (st != KVFE_OK && st != KVFE_ERR_CAPACITY) throw Error(st, "EurocDataProvider: malformed camera data.csv");
    std::vector<int64_t> ts((size_t)n);
    st = kvfe_euroc_parse_camera_csv(reinterpret_cast<const char*>(text.data()), text.size(), ts.data(), n, &n);
    if (st != KVFE_OK) throw Error(st, "EurocDataProvider: malformed camera data.csv");
    return ts;
  }
  static ImuMeasurements parse_imu(const std::vector<uint8_t>& text) {
    int32_t n = 0;
    kvfe_status st =
        kvfe_euroc_parse_imu_csv(reinterpret_cast<const char*>(text.data()), text.size(), nullptr, nullptr, 0, &n);
    if (st != KVFE_OK && st != KVFE_ERR_CAPACITY) throw Error(st, "EurocDataProvider: malformed imu data.csv");
    ImuMeasurements m;
    m.timestamps.resize((size_t)n);
    m.acc_gyr.resize((size_t)6 * n);
    st = kvfe_euroc_parse_imu_csv(reinterpret_cast<const char*>(text.data()), text.size(), m.timestamps.data(),
                                  m.acc_gyr.data(), n, &n);
    if (st != KVFE_OK) throw Error(st, "EurocDataProvider: malformed imu data.csv");
    return m;
  }
  std::string img_name(const char* cam, const std::vector<int64_t>& stamps, int64_t k) const {
    if (k < 0 || (size_t)k >= stamps.size()) return std::string();
    return path_ + "/mav0/" + cam + "/data/" + std::to_string(stamps[(size_t)k]) + ".png";
  }
  std::string path_;
  ImuMeasurements imu_;
  std::vector<int64_t> left_, right_;
  int64_t initial_k_ = 0, final_k_ = 0, current_k_ = 0;
  bool imu_sent_ = false;
  ImuCallback imu_cb_;
  FrameCallback left_cb_, right_cb_;
};

}  // namespace kvfe
