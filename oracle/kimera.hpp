// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
// CPU restatement of the reference-owned logic of Kimera-VIO's stereo front-end
// (the parts that live in /root/reference, as opposed to OpenCV): every function
// cites the reference file:line it follows.  Built on oracle/ocv.hpp.
#pragma once
#include <cstdint>
#include <array>
#include <map>
#include <vector>

#include "../include/kvfe.h"  // POD parameter structs only (no product code is linked)
#include "ocv.hpp"

namespace kimera {

using ocv::Point2f;

struct StatusKeypoint {
  uint8_t status;
  Point2f kp;
};

// StereoCamera + UndistorterRectifier x2 (src/frontend/StereoCamera.cpp:34-94,
// src/frontend/UndistorterRectifier.cpp:26-31,230-292)
struct StereoCamera {
  kvfe_camera_params left, right;
  double K1[9], K2[9];
  kvfe_rectification rect;
  std::vector<float> map_x[2], map_y[2];
  int w, h;
  void init(const kvfe_camera_params& l, const kvfe_camera_params& r);
  // VIO::Camera (src/frontend/Camera.cpp:29-49): one camera, undistorter with R = I and P = K
  void initMono(const kvfe_camera_params& c);
  double fx() const { return rect.P1[0]; }
  double baseline() const { return rect.baseline; }
  // UndistorterRectifier::undistortRectifyImage (UndistorterRectifier.cpp:115-128)
  void undistortRectifyImage(int cam, const uint8_t* src, size_t stride, uint8_t* dst) const;
  // UndistorterRectifier::UndistortRectifyKeypoints (UndistorterRectifier.cpp:33-68)
  void undistortRectifyKeypoints(int cam, const Point2f* in, int n, bool useR, bool useP,
                                 Point2f* out) const;
  // UndistorterRectifier::GetBearingVector (UndistorterRectifier.cpp:73-113)
  void getBearingVector(int cam, Point2f kp, double versor[3]) const;
  // StereoCamera::undistortRectifyLeftKeypoints (StereoCamera.cpp:236-260) =
  // undistortRectifyKeypoints + checkUndistortedRectifiedLeftKeypoints
  // (UndistorterRectifier.cpp:138-211, pixel_tol = 2.0f)
  void undistortRectifyLeftKeypoints(const std::vector<Point2f>& kps,
                                     std::vector<StatusKeypoint>& out) const;
  // UndistorterRectifier::distortUnrectifyKeypoints (UndistorterRectifier.cpp:213-228)
  void distortUnrectifyRightKeypoints(const std::vector<StatusKeypoint>& rect,
                                      std::vector<Point2f>& out) const;
  // the two UndistorterRectifier methods on their own, for either camera's rectifier
  void checkUndistortedRectifiedKeypoints(int cam, const std::vector<Point2f>& distorted,
                                          const std::vector<Point2f>& undistorted, float pixel_tol,
                                          std::vector<StatusKeypoint>& out) const;
  void distortUnrectifyKeypoints(int cam, const std::vector<StatusKeypoint>& rect, std::vector<Point2f>& out) const;
};

void camera_matrix(const kvfe_camera_params& c, double K[9]);

// cv::sortIdx(all-equal int keys, SORT_DESCENDING) permutation
// (NonMaximumSuppression.cpp:50-60): libstdc++ introsort with an always-false
// comparator, then reversed.  policy: KVFE_SORTIDX_*.
void sortidx_descending_equal_keys(int n, int policy, std::vector<int>& idx);

// AdaptiveNonMaximumSuppression::suppressNonMax (NonMaximumSuppression.cpp:33-122)
// for TopN, Binning (the types the shipped YAMLs and the reference tests use), BrownANMS and the
// radius-search variants Sdc / KdTree / RangeTree / Ssc (anms/anms.cpp).
// responses: the keypoints' cv::KeyPoint::response (FAST scores); null = all-equal keys (GFTT)
bool suppressNonMax(const std::vector<Point2f>& keypoints, int numRetPoints, int cols, int rows,
                    const kvfe_detector_params& p, std::vector<Point2f>& out,
                    const std::vector<float>* responses = nullptr);

// private FeatureDetector::featureDetection(const Frame&, need) (FeatureDetector.cpp:174-299)
bool featureDetection(const uint8_t* img, int w, int h, size_t stride,
                      const std::vector<Point2f>& tracked, int need_n_corners,
                      const kvfe_detector_params& p, std::vector<Point2f>& new_corners,
                      std::vector<Point2f>* raw_gftt = nullptr,
                      const uint8_t* detection_mask = nullptr /* Frame::detection_mask_, w*h */);

// OpticalFlowPredictor::predictSparseFlow (OpticalFlowPredictor.cpp:27-33,70-126)
void predictSparseFlow(int type, const double K[9], int w, int h, const Point2f* prev, int n,
                       const double ref_R_cur[9], Point2f* next);

// StereoMatcher::getRightKeypointsRectified + searchRightKeypointEpipolar
// (StereoMatcher.cpp:196-423)
void getRightKeypointsRectified(const uint8_t* left_rect, const uint8_t* right_rect, int w, int h,
                                size_t stride, const std::vector<StatusKeypoint>& left,
                                double fx, double baseline, const kvfe_stereo_params& p,
                                std::vector<StatusKeypoint>& right, std::vector<double>* scores);

// StereoMatcher::getDepthFromRectifiedMatches (StereoMatcher.cpp:425-483)
void getDepthFromRectifiedMatches(std::vector<StatusKeypoint>& left,
                                  std::vector<StatusKeypoint>& right, double fx, double baseline,
                                  const kvfe_stereo_params& p, std::vector<double>& depth);

struct Frame {
  int64_t id = 0, timestamp = 0;
  bool isKeyframe = false;
  int w = 0, h = 0;
  std::vector<uint8_t> img;
  std::vector<Point2f> keypoints;
  std::vector<int64_t> landmarks;
  std::vector<int32_t> landmarks_age;
  std::vector<double> versors;  // n x 3
};

struct StereoFrame {
  Frame left;
  std::vector<uint8_t> right_img, left_rect, right_rect;
  std::vector<StatusKeypoint> left_kp_rect, right_kp_rect;
  std::vector<double> depth;
  std::vector<Point2f> right_kp;
  std::vector<double> kp3d;
  int n_tracked = 0, n_detected = 0;
};

// ---- geometric outlier rejection (Tracker.cpp:213-1018) ------------------------------------------
typedef std::pair<size_t, size_t> KeypointMatch;  // (index in ref frame, index in cur frame)

// TrackerStatusSummary + DebugTrackerInfo (Tracker-definitions.h:78-183); poses 3x4 [R | t]
struct TrackerStatusSummary {
  int mono = KVFE_TRACKING_INVALID, stereo = KVFE_TRACKING_INVALID;
  double lkf_T_k_mono[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  double lkf_T_k_stereo[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  double info[9] = {0};
  int nr_mono_putatives = 0, nr_mono_inliers = 0, mono_iters = 0;
  int nr_stereo_putatives = 0, nr_stereo_inliers = 0;
  int pnp = KVFE_TRACKING_INVALID;   // kfTracking_status_pnp_
  double W_T_k_pnp[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  int nr_pnp_inliers = 0;
};

// Tracker::findMatchingKeypoints (Tracker.cpp:919-946)
void findMatchingKeypoints(const Frame& ref, const Frame& cur, std::vector<KeypointMatch>& out);
// Tracker::findMatchingStereoKeypoints (Tracker.cpp:948-989)
void findMatchingStereoKeypoints(const StereoFrame& ref, const StereoFrame& cur,
                                 const std::vector<KeypointMatch>& mono,
                                 std::vector<KeypointMatch>& out);
// the float32 Mahalanobis distance of the 1-point voting loop (Tracker.cpp:499-523): (vi - vj)^T (Ci + Cj)^-1 (vi - vj)
float mahalanobis_f(const float* vi, const float* Ci, const float* vj, const float* Cj);
// Tracker::computeMedianDisparity (Tracker.cpp:991-1018)
bool computeMedianDisparity(const std::vector<Point2f>& ref, const std::vector<Point2f>& cur,
                            const std::vector<KeypointMatch>& matches, double* median);

struct RansacOut {
  int status = KVFE_TRACKING_INVALID;
  double pose[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  double info[9] = {0};
  std::vector<int> inliers;
  int iterations = 0;
};
// Tracker::geometricOutlierRejection2d2d(bearings, ..., cam_lkf_Pose_cam_kf) with
// ransac_use_2point_mono_ (Tracker.cpp:213-318) over n already gathered matches
RansacOut outlierRejection2d2dGivenRot(const double* f_ref, const double* f_cur, int n,
                                       const double R[9], const kvfe_tracker_params& tp);
// Tracker::pnp(bearings, F_points, ...) (Tracker.cpp:1122-1288, EPNP) + the status of
// VisionImuFrontend::outlierRejectionPnP (VisionImuFrontend.cpp:146-173); *success = Tracker::pnp's return value;
// avg_focal_length = 0.5 (fx + fy) of the camera the tracker was built with
RansacOut pnp(const double* bearings, const double* points, int n, double avg_focal_length,
              const kvfe_tracker_params& tp, const kvfe_pnp_params& pp, bool* success);
// Tracker::geometricOutlierRejection2d2d without a rotation prior (ransac_use_2point_mono_ = false or no gyro
// rotation): 5-point Nister RANSAC (Tracker.cpp:213-318, Problem2d2d::NISTER) over n gathered matches
RansacOut outlierRejection2d2d(const double* f_ref, const double* f_cur, int n, const kvfe_tracker_params& tp);
// Tracker::geometricOutlierRejection3d3d(ref_keypoints_3d, cur_keypoints_3d, matches, inliers)
// (Tracker.cpp:667-742): 3-point Arun RANSAC over n already gathered matches
RansacOut outlierRejection3d3d(const double* ref_p3, const double* cur_p3, int n,
                               const kvfe_tracker_params& tp);
// gtsam::Cal3_S2Stereo of the rectified pair (StereoCamera.cpp:75-83)
struct StereoCalib {
  double fx, fy, s, cx, cy, b;
};
// Tracker::getPoint3AndCovariance (Tracker.cpp:772-818), stereo_point_covariance = I
void getPoint3AndCovariance(const StereoCalib& K, double uL, double uR, double v, const double p3[3],
                            const double* Rmat /* may be null */, double point[3], double cov[9]);
// Tracker::geometricOutlierRejection3d3dGivenRotation (Tracker.cpp:382-632) over n stereo matches
RansacOut outlierRejection3d3dGivenRot(const float* ref_left_xy, const float* ref_right_x,
                                       const double* ref_p3, const float* cur_left_xy,
                                       const float* cur_right_x, const double* cur_p3, int n,
                                       const StereoCalib& K, const double R[9],
                                       const kvfe_tracker_params& tp);

// StereoMatcher::sparseStereoReconstruction(StereoFrame*) (StereoMatcher.cpp:123-175)
void sparseStereoReconstruction(const StereoCamera& cam, const kvfe_stereo_params& p,
                                StereoFrame& sf);

// StereoVisionImuFrontend (processFirstStereoFrame / processStereoFrame incl. the useRANSAC
// branch for the 2-point mono / 1-point stereo problems;
// src/frontend/StereoVisionImuFrontend.cpp:245-531) +
// VisionImuFrontend::shouldBeKeyframe (VisionImuFrontend.cpp:175-232) +
// Tracker::featureTracking (Tracker.cpp:92-211) +
// FeatureDetector::featureDetection(Frame*, R) (FeatureDetector.cpp:94-163).
// StereoMatcher::denseStereoReconstruction (StereoMatcher.cpp:32-121) on rectified images; roi1/roi2:
// StereoCamera::getROI1/2 (x, y, w, h), used by the StereoBM branch only.  disp: CV_16S, stride w.
void denseStereoReconstruction(const kvfe_dense_stereo_params& dp, const int roi1[4], const int roi2[4],
                               const uint8_t* left_rect, const uint8_t* right_rect, int w, int h,
                               size_t stride, short* disp);

// DepthFrame::getDetectionMask (DepthFrame.cpp:76-96) / DepthFrame::getDepthAtPoint (DepthFrame.cpp:40-74) on a depth
// image of (w, h) elements, uint16 or float per kvfe_depth_params::depth_type
void depthDetectionMask(const kvfe_depth_params& dp, const void* depth, int w, int h, size_t stride,
                        std::vector<uint8_t>& mask);
float depthAtPoint(const kvfe_depth_params& dp, const void* depth, int w, int h, size_t stride, Point2f pt);

struct Frontend {
  StereoCamera cam;
  kvfe_frontend_params p;
  int64_t lmk_id = 0;  // FeatureDetector.cpp:141 (static counter, here per stream)
  int64_t frame_count = 0;
  bool initialized = false;
  StereoFrame km1, lkf, k;
  bool km1_is_lkf = false;
  double keyframe_R_ref_frame[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  std::vector<int64_t> meas_lmk;
  std::vector<double> meas_uLuRv;
  bool last_is_keyframe = false;
  TrackerStatusSummary tracker_status;  // tracker_status_summary_ (persists between keyframes)

  bool mono = false;  // MonoVisionImuFrontend (src/frontend/MonoVisionImuFrontend.cpp:196-335)
  bool rgbd = false;  // RgbdVisionImuFrontend (src/frontend/RgbdVisionImuFrontend.cpp:184-368)
  kvfe_depth_params depth_params{};
  void init(const kvfe_camera_params& l, const kvfe_camera_params& r,
            const kvfe_frontend_params& fp, bool mono_frontend = false);
  void initRgbd(const kvfe_camera_params& cam, const kvfe_frontend_params& fp, const kvfe_depth_params& dp);
  // stereo / mono: right = right image (ignored by mono); rgbd: right = depth image (uint16 or float, `stride`
  // elements per row)
  void process(const uint8_t* left, const uint8_t* right, size_t stride,
               const kvfe_frame_input& in);
  const StereoFrame& current() const { return km1; }
  // Tracker::updateMap (Tracker.h:82-94) and VisionImuFrontend::outlierRejectionPnP (VisionImuFrontend.cpp:146-173)
  // = Tracker::pnp(const StereoFrame&) (Tracker.cpp:1064-1120) on the keyframe
  std::map<int64_t, std::array<double, 3>> landmarks_map;
  void updateMap(const int64_t* ids, const double* xyz, int n);
  void outlierRejectionPnP(const StereoFrame& frame);

 // RgbdFrame::fillStereoFrame (public so that the reference's component KAT can drive it, capi.cpp);
  // (dw, dh) = size of the depth image; <= 0: the camera resolution
  void fillStereoFrame(StereoFrame& sf, const void* depth, size_t depth_stride, int dw = 0, int dh = 0) const;
  // FeatureDetector::featureDetection(Frame*, R) / Tracker::featureTracking as component calls (capi.cpp)
  void featureDetectionFrame(Frame& f, int* n_detected, const uint8_t* detection_mask = nullptr);
  void featureTracking(Frame& ref, Frame& cur, const double ref_R_cur[9]);

 private:
  bool shouldBeKeyframe(const Frame& frame, const Frame& frame_lkf) const;
  void getSmartStereoMeasurements(const StereoFrame& sf);
  // VisionImuFrontend::outlierRejectionMono / outlierRejectionStereo (VisionImuFrontend.cpp:90-144)
  void outlierRejectionMono(const double R[9], Frame& lkf_left, Frame& k_left);
  void outlierRejectionStereo(const double R[9], StereoFrame& lkf_sf, StereoFrame& k_sf);
  void processMono(const kvfe_frame_input& in);
  void processRgbd(const kvfe_frame_input& in, const void* depth, size_t depth_stride);
  void depthDetectionMask(const void* depth, size_t depth_stride, std::vector<uint8_t>& mask) const;
};

// UtilsOpenCV::cropToSize / roundAndCropToSize (UtilsOpenCV.cpp:215-247): clamp to [0, w-1] x [0, h-1], true if moved
bool cropToSize(Point2f* px, int w, int h);
bool roundAndCropToSize(Point2f* px, int w, int h);

// StereoVisionImuFrontend::getSmartStereoMeasurements (StereoVisionImuFrontend.cpp:485-531): (landmark id,
// uL, uR or NaN, v) of every keypoint with a landmark
void smartStereoMeasurements(const StereoFrame& sf, bool use_stereo_tracking, std::vector<int64_t>& meas_lmk,
                             std::vector<double>& meas_uLuRv);

// gtsam::Rot3::equals(Rot3(), 1e-9) as used for `given_rot` (VisionImuFrontend.cpp:97,125)
bool rot_equals_identity(const double R[9], double tol);

}  // namespace kimera
