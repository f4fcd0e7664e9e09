// shim_check: compiles include/kvfe_kimera_shim.hpp against the stand-in OpenCV / GTSAM headers of tests/cpp/stubs
// and drives the reference-signature methods on Kimera-shaped Frame / StereoFrame structs (same member names and
// types as include/kimera-vio/frontend/Frame.h:160-186, StereoFrame.h:137-171).  Input / output format as
// adapter_sequence.cpp; tests/test_gpu_parity.py compares the records with the oracle.
#include <algorithm>
#include <array>
#include <chrono>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <cmath>
#include <unordered_map>
#include <vector>

#include "kvfe_kimera_shim.hpp"

namespace VIO {   // Kimera-shaped types (only what the hot path touches)
using LandmarkId = long int;
enum class KeypointStatus { VALID, NO_LEFT_RECT, NO_RIGHT_RECT, NO_DEPTH, FAILED_ARUN };
using KeypointCV = cv::Point2f;
using KeypointsCV = std::vector<KeypointCV>;
using StatusKeypointCV = std::pair<KeypointStatus, KeypointCV>;
using StatusKeypointsCV = std::vector<StatusKeypointCV>;
struct FeatureDetectorParams {};
struct Frame {
  cv::Mat img_;
  KeypointsCV keypoints_;
  std::vector<double> scores_;
  std::vector<LandmarkId> landmarks_;
  std::vector<size_t> landmarks_age_;
  std::vector<gtsam::Vector3> versors_;
};
enum class Pose3d2dAlgorithm { KneipP2P = 0, KneipP3P = 1, GaoP3P = 2, EPNP = 3, UPNP = 4 };
struct TrackerParams {   // the PnP members of VisionImuTrackerParams.h:72-76
  Pose3d2dAlgorithm pnp_algorithm_ = Pose3d2dAlgorithm::EPNP;
  int min_pnp_inliers_ = 10;
  double ransac_threshold_pnp_ = 1.0;
  bool optimize_2d3d_pose_from_inliers_ = false;
};
using BearingVectors = std::vector<gtsam::Vector3>;
using Landmarks = std::vector<gtsam::Point3>;
using LandmarksMap = std::unordered_map<LandmarkId, gtsam::Point3>;
using Timestamp = std::int64_t;
struct StereoFrame {   // StereoFrame.h:137-171
  Timestamp timestamp_ = 0;
  Frame left_frame_, right_frame_;
  StatusKeypointsCV left_keypoints_rectified_, right_keypoints_rectified_;
  std::vector<double> keypoints_depth_;
  BearingVectors keypoints_3d_;
  cv::Mat left_img_rectified_, right_img_rectified_;
  void setRectifiedImages(const cv::Mat& l, const cv::Mat& r) {
    left_img_rectified_ = l;
    right_img_rectified_ = r;
  }
};
// Eigen-shaped stand-ins of ImuStampS (1 x k int64) / ImuAccGyrS (6 x k double): cols(), operator()(row, col)
struct ImuStampS {
  std::vector<std::int64_t> v;
  long cols() const { return (long)v.size(); }
  std::int64_t operator()(int, int c) const { return v[c]; }
};
struct ImuAccGyrS {
  std::vector<double> v;   // column-major 6 x k
  long cols() const { return (long)v.size() / 6; }
  double operator()(int r, int c) const { return v[(size_t)6 * c + r]; }
};
struct StereoImuSyncPacket {   // StereoImuSyncPacket.h:81-107 (+ FrontendInputPacketBase.h:37-52)
  StereoFrame stereo_frame_;
  ImuStampS imu_stamps_;
  ImuAccGyrS imu_accgyrs_;
  const StereoFrame& getStereoFrame() const { return stereo_frame_; }
  const ImuStampS& getImuStamps() const { return imu_stamps_; }
  const ImuAccGyrS& getImuAccGyrs() const { return imu_accgyrs_; }
};
enum class TrackingStatus { VALID, LOW_DISPARITY, FEW_MATCHES, INVALID, DISABLED };
struct TrackerStatusSummary {   // Tracker-definitions.h:134-183
  TrackingStatus kfTrackingStatus_mono_ = TrackingStatus::INVALID, kfTrackingStatus_stereo_ = TrackingStatus::INVALID;
  gtsam::Pose3 lkf_T_k_mono_, lkf_T_k_stereo_;
};
using StereoMeasurement = std::pair<LandmarkId, std::array<double, 3>>;   // StereoPoint2(uL, uR, v)
using StatusStereoMeasurements = std::pair<TrackerStatusSummary, std::vector<StereoMeasurement>>;
struct StereoFrontendOutput {   // the constructor arguments of StereoVisionImuFrontend-definitions.h:30-42 this path fills
  bool is_keyframe_;
  std::shared_ptr<StatusStereoMeasurements> status_stereo_measurements_;
  StereoFrame stereo_frame_lkf_;
  ImuAccGyrS imu_acc_gyrs_;
  StereoFrontendOutput(bool is_keyframe, std::shared_ptr<StatusStereoMeasurements> m, const StereoFrame& sf,
                       const ImuAccGyrS& ag)
      : is_keyframe_(is_keyframe), status_stereo_measurements_(std::move(m)), stereo_frame_lkf_(sf), imu_acc_gyrs_(ag) {}
};
}  // namespace VIO

namespace {
struct Writer {
  FILE* f;
  void put(const char* tag, const void* p, size_t n) {
    char t[16] = {0};
    std::snprintf(t, sizeof(t), "%s", tag);
    const int64_t nb = (int64_t)n;
    std::fwrite(t, 1, 16, f);
    std::fwrite(&nb, sizeof(nb), 1, f);
    if (n) std::fwrite(p, 1, n, f);
  }
};
void put_frame(Writer& w, const std::string& pre, const VIO::Frame& f) {
  std::vector<float> kp;
  std::vector<int64_t> lmk;
  std::vector<int32_t> age;
  std::vector<double> ver;
  for (size_t i = 0; i < f.landmarks_.size(); i++) {
    kp.push_back(f.keypoints_[i].x);
    kp.push_back(f.keypoints_[i].y);
    lmk.push_back(f.landmarks_[i]);
    age.push_back((int32_t)f.landmarks_age_[i]);
    for (int c = 0; c < 3; c++) ver.push_back(f.versors_[i](c));
  }
  w.put((pre + "_kp").c_str(), kp.data(), kp.size() * 4);
  w.put((pre + "_lmk").c_str(), lmk.data(), lmk.size() * 8);
  w.put((pre + "_age").c_str(), age.data(), age.size() * 4);
  w.put((pre + "_ver").c_str(), ver.data(), ver.size() * 8);
}
bool read_exact(FILE* f, void* p, size_t n) { return std::fread(p, 1, n, f) == n; }
}  // namespace

int main(int argc, char** argv) {
  // shim_check in.bin out.bin            the parity records (tests/test_gpu_components_r2.py)
  // shim_check in.bin out.bin bench N    N timed StereoVisionImuFrontend::spinOnce calls over the input frames walked
  //                                      ping-pong, every call returning its StereoFrontendOutput (bench.py: single_stream_spinonce)
  const int bench_n = (argc == 5 && std::string(argv[3]) == "bench") ? std::atoi(argv[4]) : 0;
  if (argc != 3 && bench_n <= 0) return 2;
  FILE* fi = std::fopen(argv[1], "rb");
  FILE* fo = std::fopen(argv[2], "wb");
  if (!fi || !fo) return 2;
  kvfe_config cfg;
  int32_t hdr[3];
  if (!read_exact(fi, &cfg, sizeof(cfg)) || !read_exact(fi, hdr, sizeof(hdr))) return 2;
  const int n_frames = hdr[0], W = hdr[1], H = hdr[2];
  const size_t N = (size_t)W * H;
  if (n_frames < 2) return 2;
  std::vector<kvfe_frame_input> inputs(n_frames);
  std::vector<std::vector<uint8_t>> lefts(n_frames), rights(n_frames);
  for (int i = 0; i < n_frames; i++) {
    lefts[i].resize(N);
    rights[i].resize(N);
    if (!read_exact(fi, &inputs[i], sizeof(kvfe_frame_input)) || !read_exact(fi, lefts[i].data(), N) ||
        !read_exact(fi, rights[i].data(), N))
      return 2;
  }
  std::fclose(fi);
  Writer w{fo};
  if (bench_n > 0) {
    try {
      kvfe::Context fctx(cfg.left, cfg.right, cfg.params, 1, cfg.device);
      kvfe::StereoCamera cam(fctx);
      const kvfe::Pose3 bl = cam.getBodyPoseLeftCamRect(cfg.left);
      kvfe::shim::StereoVisionImuFrontend frontend(fctx, bl.R);
      const int warm = 20;
      size_t total_meas = 0, keyframes = 0;
      std::chrono::steady_clock::time_point t_begin;
      for (int it = 0; it < warm + bench_n; it++) {
        if (it == warm) t_begin = std::chrono::steady_clock::now();
        const int period = 2 * (n_frames - 1), j = it % period, i = j < n_frames ? j : period - j;
        VIO::StereoImuSyncPacket pk;
        pk.stereo_frame_.timestamp_ = (int64_t)it * 50000000;
        pk.stereo_frame_.left_frame_.img_ = cv::Mat(H, W, CV_8UC1, lefts[i].data(), (size_t)W);
        pk.stereo_frame_.right_frame_.img_ = cv::Mat(H, W, CV_8UC1, rights[i].data(), (size_t)W);
        const int64_t t1 = pk.stereo_frame_.timestamp_, t0 = t1 - 50000000;
        // bench input: keyframe_R_cur_frame[0..2] of frame i carries the body-frame gyro rate (rad/s) of the interval
        // i-1 -> i, so that the rotation the shim preintegrates is the one the images show; walking backwards it is
        // the negative rate of the interval just undone
        const bool fwd = j < n_frames && j > 0;
        const double* g = fwd ? inputs[i].keyframe_R_cur_frame : inputs[std::min(i + 1, n_frames - 1)].keyframe_R_cur_frame;
        const double sgn = it == 0 ? 0.0 : (fwd ? 1.0 : -1.0);
        for (int q = 0; q <= 10; q++) {
          pk.imu_stamps_.v.push_back(t0 + (t1 - t0) * q / 10);
          const double row[6] = {0.1, 9.8, 0.2, sgn * g[0], sgn * g[1], sgn * g[2]};   // acc, gyro
          pk.imu_accgyrs_.v.insert(pk.imu_accgyrs_.v.end(), row, row + 6);
        }
        auto out = frontend.spinOnce(std::move(pk), [&](const kvfe::shim::SpinResult& r) {
          auto m = std::make_shared<VIO::StatusStereoMeasurements>();
          m->first.kfTrackingStatus_mono_ = static_cast<VIO::TrackingStatus>(r.kfTrackingStatus_mono);
          m->first.kfTrackingStatus_stereo_ = static_cast<VIO::TrackingStatus>(r.kfTrackingStatus_stereo);
          for (size_t q = 0; q < r.meas_landmark.size(); q++)
            m->second.push_back({(VIO::LandmarkId)r.meas_landmark[q],
                                 {r.meas_uL_uR_v[3 * q], r.meas_uL_uR_v[3 * q + 1], r.meas_uL_uR_v[3 * q + 2]}});
          VIO::StereoFrame sfo;
          sfo.timestamp_ = r.timestamp;
          kvfe::shim::from_frame(r.left_frame, &sfo.left_frame_);
          kvfe::shim::from_status(r.left_keypoints_rectified, &sfo.left_keypoints_rectified_);
          kvfe::shim::from_status(r.right_keypoints_rectified, &sfo.right_keypoints_rectified_);
          sfo.keypoints_depth_ = r.keypoints_depth;
          return std::make_unique<VIO::StereoFrontendOutput>(r.is_keyframe, m, sfo, VIO::ImuAccGyrS());
        });
        if (it >= warm) {
          total_meas += out->status_stereo_measurements_->second.size();
          keyframes += out->is_keyframe_ ? 1 : 0;
        }
      }
      const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
      std::printf("{\"spins\": %d, \"seconds\": %.6f, \"pairs_per_s\": %.2f, \"ms_per_pair\": %.5f, "
                  "\"keyframes\": %zu, \"measurements_per_spin\": %.1f}\n",
                  bench_n, sec, bench_n / sec, 1e3 * sec / bench_n, keyframes, (double)total_meas / bench_n);
    } catch (const kvfe::Error& e) {
      std::fprintf(stderr, "kvfe::Error %d: %s\n", (int)e.status, e.what());
      return 1;
    }
    std::fclose(fo);
    return 0;
  }
  try {
    kvfe::Context ctx(cfg.left, cfg.right, cfg.params, 1, cfg.device);
    kvfe::shim::FeatureDetector feature_detector(ctx);
    kvfe::shim::Tracker tracker(ctx);
    kvfe::shim::StereoCamera stereo_camera(ctx);
    kvfe::shim::StereoMatcher stereo_matcher(ctx);
    kvfe::FeatureDetector::landmarkCounter() = 0;

    VIO::Frame ref, cur;
    ref.img_ = cv::Mat(H, W, CV_8UC1, lefts[0].data(), (size_t)W);
    cur.img_ = cv::Mat(H, W, CV_8UC1, lefts[1].data(), (size_t)W);
    feature_detector.featureDetection(&ref, std::nullopt);                       // FeatureDetector.h:39-41
    put_frame(w, "s_f0", ref);
    gtsam::Matrix3 M;
    for (int i = 0; i < 9; i++) M.m[i] = inputs[1].keyframe_R_cur_frame[i];
    tracker.featureTracking(&ref, &cur, gtsam::Rot3(M), VIO::FeatureDetectorParams(), std::nullopt);   // Tracker.h:70-74
    put_frame(w, "s_ref", ref);
    put_frame(w, "s_f1", cur);
    const std::vector<cv::KeyPoint> raw = feature_detector.rawFeatureDetection(ref.img_);
    std::vector<float> rawxy;
    for (const auto& k : raw) {
      rawxy.push_back(k.pt.x);
      rawxy.push_back(k.pt.y);
    }
    w.put("s_raw", rawxy.data(), rawxy.size() * 4);

    {   // Tracker::updateMap + both Tracker::pnp overloads (Tracker.h): a cube of landmarks seen from (I, [0 0 -2])
      VIO::TrackerParams tp;
      tracker.setPnpParams(tp);
      VIO::LandmarksMap map;
      VIO::StereoFrame pf;
      VIO::BearingVectors bearings;
      VIO::Landmarks points;
      int id = 0;
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++)
          for (int c = 0; c < 3; c++) {
            gtsam::Point3 X;
            X(0) = 0.5 * a - 0.4 + 0.03 * c;
            X(1) = 0.45 * b - 0.5 + 0.02 * a;
            X(2) = 0.5 * c + 0.01 * b;
            gtsam::Vector3 f;
            const double pc[3] = {X(0), X(1), X(2) + 2.0};
            const double nrm = std::sqrt(pc[0] * pc[0] + pc[1] * pc[1] + pc[2] * pc[2]);
            for (int k = 0; k < 3; k++) f(k) = pc[k] / nrm;
            map[id] = X;
            pf.left_frame_.landmarks_.push_back(id);
            pf.left_keypoints_rectified_.push_back({VIO::KeypointStatus::VALID, {0.f, 0.f}});
            pf.keypoints_3d_.push_back(f);
            bearings.push_back(f);
            points.push_back(X);
            id++;
          }
      pf.left_frame_.landmarks_[3] = -1;                                           // dropped: no landmark id
      pf.left_keypoints_rectified_[5].first = VIO::KeypointStatus::NO_LEFT_RECT;   // dropped: not VALID
      map.erase(7);                                                                // dropped: out of the time horizon
      tracker.updateMap(map);
      gtsam::Pose3 T1, T2;
      std::vector<int> in1, in2;
      const bool ok1 = tracker.pnp(bearings, points, &T1, &in1);
      const bool ok2 = tracker.pnp(pf, &T2, &in2);
      const int32_t head[4] = {ok1, ok2, (int32_t)in1.size(), (int32_t)in2.size()};
      w.put("s_pnp", head, sizeof(head));
      double tt[6] = {T1.translation()(0), T1.translation()(1), T1.translation()(2),
                      T2.translation()(0), T2.translation()(1), T2.translation()(2)};
      w.put("s_pnp_t", tt, sizeof(tt));
    }

    VIO::StereoFrame sf;
    sf.left_frame_.img_ = ref.img_;
    sf.right_frame_.img_ = cv::Mat(H, W, CV_8UC1, rights[0].data(), (size_t)W);
    stereo_camera.undistortRectifyStereoFrame(&sf);                               // StereoCamera.h:225-233
    w.put("s_lrect", sf.left_img_rectified_.data, N);
    w.put("s_rrect", sf.right_img_rectified_.data, N);

    VIO::StatusKeypointsCV left{{VIO::KeypointStatus::VALID, {400.f, 200.f}},
                                {VIO::KeypointStatus::VALID, {300.f, 100.f}},
                                {VIO::KeypointStatus::NO_LEFT_RECT, {10.f, 10.f}}};
    VIO::StatusKeypointsCV right{{VIO::KeypointStatus::VALID, {380.f, 200.f}},
                                 {VIO::KeypointStatus::VALID, {310.f, 100.f}},
                                 {VIO::KeypointStatus::VALID, {5.f, 10.f}}};
    {   // StereoMatcher::sparseStereoReconstruction(StereoFrame*) (StereoMatcher.h:49-52) on the first frame's corners
      VIO::StereoFrame s2;
      s2.left_frame_ = ref;                    // (keypoints / versors of featureDetection above; image = left 0)
      s2.left_frame_.img_ = cv::Mat(H, W, CV_8UC1, lefts[0].data(), (size_t)W);
      s2.right_frame_.img_ = cv::Mat(H, W, CV_8UC1, rights[0].data(), (size_t)W);
      stereo_matcher.sparseStereoReconstruction(&s2);
      std::vector<float> lrx, rrx, rk;
      std::vector<uint8_t> lst, rst;
      std::vector<double> p3;
      for (size_t i = 0; i < s2.left_keypoints_rectified_.size(); i++) {
        lst.push_back((uint8_t)s2.left_keypoints_rectified_[i].first);
        rst.push_back((uint8_t)s2.right_keypoints_rectified_[i].first);
        lrx.push_back(s2.left_keypoints_rectified_[i].second.x);
        lrx.push_back(s2.left_keypoints_rectified_[i].second.y);
        rrx.push_back(s2.right_keypoints_rectified_[i].second.x);
        rrx.push_back(s2.right_keypoints_rectified_[i].second.y);
        rk.push_back(s2.right_frame_.keypoints_[i].x);
        rk.push_back(s2.right_frame_.keypoints_[i].y);
        for (int c = 0; c < 3; c++) p3.push_back(s2.keypoints_3d_[i](c));
      }
      w.put("s_sp_lst", lst.data(), lst.size());
      w.put("s_sp_rst", rst.data(), rst.size());
      w.put("s_sp_lr", lrx.data(), lrx.size() * 4);
      w.put("s_sp_rr", rrx.data(), rrx.size() * 4);
      w.put("s_sp_rk", rk.data(), rk.size() * 4);
      w.put("s_sp_dep", s2.keypoints_depth_.data(), s2.keypoints_depth_.size() * 8);
      w.put("s_sp_p3", p3.data(), p3.size() * 8);
      w.put("s_sp_lrect", s2.left_img_rectified_.data, N);
      // StereoMatcher::denseStereoReconstruction(const cv::Mat&, const cv::Mat&, cv::Mat*) (StereoMatcher.h:54-66)
      cv::Mat disp;
      stereo_matcher.denseStereoReconstruction(s2.left_img_rectified_, s2.right_img_rectified_, &disp);
      w.put("s_disp", disp.data, (size_t)disp.rows * disp.step);
    }
    {   // StereoVisionImuFrontend::spinOnce(StereoImuSyncPacket&&) -> StereoFrontendOutput on every frame of the input:
      // IMU samples are synthesised from the input rotations' time stamps (constant rate about z), the rotation the
      // step receives is what the shim integrates from them
      kvfe::Context fctx(cfg.left, cfg.right, cfg.params, 1, cfg.device);
      kvfe::StereoCamera cam(fctx);
      const kvfe::Pose3 bl = cam.getBodyPoseLeftCamRect(cfg.left);
      kvfe::shim::StereoVisionImuFrontend frontend(fctx, bl.R);
      std::vector<int32_t> kf;
      std::vector<int32_t> nmeas;
      std::vector<double> Rk, last_meas;
      std::vector<int64_t> last_lmk;
      for (int i = 0; i < n_frames; i++) {
        VIO::StereoImuSyncPacket pk;
        pk.stereo_frame_.timestamp_ = inputs[i].timestamp_ns;
        pk.stereo_frame_.left_frame_.img_ = cv::Mat(H, W, CV_8UC1, lefts[i].data(), (size_t)W);
        pk.stereo_frame_.right_frame_.img_ = cv::Mat(H, W, CV_8UC1, rights[i].data(), (size_t)W);
        const int64_t t1 = inputs[i].timestamp_ns, t0 = i ? inputs[i - 1].timestamp_ns : t1 - 50000000;
        for (int j = 0; j <= 10; j++) {
          pk.imu_stamps_.v.push_back(t0 + (t1 - t0) * j / 10);
          const double row[6] = {0.1, 9.8, 0.2, 0.01, -0.02, 0.3};   // acc, gyro
          pk.imu_accgyrs_.v.insert(pk.imu_accgyrs_.v.end(), row, row + 6);
        }
        auto out = frontend.spinOnce(std::move(pk), [&](const kvfe::shim::SpinResult& r) {
          auto m = std::make_shared<VIO::StatusStereoMeasurements>();
          m->first.kfTrackingStatus_mono_ = static_cast<VIO::TrackingStatus>(r.kfTrackingStatus_mono);
          m->first.kfTrackingStatus_stereo_ = static_cast<VIO::TrackingStatus>(r.kfTrackingStatus_stereo);
          for (size_t q = 0; q < r.meas_landmark.size(); q++)
            m->second.push_back({(VIO::LandmarkId)r.meas_landmark[q],
                                 {r.meas_uL_uR_v[3 * q], r.meas_uL_uR_v[3 * q + 1], r.meas_uL_uR_v[3 * q + 2]}});
          VIO::StereoFrame sfo;
          sfo.timestamp_ = r.timestamp;
          kvfe::shim::from_frame(r.left_frame, &sfo.left_frame_);
          kvfe::shim::from_status(r.left_keypoints_rectified, &sfo.left_keypoints_rectified_);
          kvfe::shim::from_status(r.right_keypoints_rectified, &sfo.right_keypoints_rectified_);
          sfo.keypoints_depth_ = r.keypoints_depth;
          return std::make_unique<VIO::StereoFrontendOutput>(r.is_keyframe, m, sfo, VIO::ImuAccGyrS());
        });
        kf.push_back(out->is_keyframe_ ? 1 : 0);
        nmeas.push_back((int32_t)out->status_stereo_measurements_->second.size());
        if (i == n_frames - 1) {
          for (const auto& mm : out->status_stereo_measurements_->second) {
            last_lmk.push_back(mm.first);
            last_meas.insert(last_meas.end(), mm.second.begin(), mm.second.end());
          }
          put_frame(w, "s_fe_last", out->stereo_frame_lkf_.left_frame_);
        }
      }
      w.put("s_fe_kf", kf.data(), kf.size() * 4);
      w.put("s_fe_nmeas", nmeas.data(), nmeas.size() * 4);
      w.put("s_fe_lmk", last_lmk.data(), last_lmk.size() * 8);
      w.put("s_fe_meas", last_meas.data(), last_meas.size() * 8);
    }
    std::vector<double> depths;
    stereo_matcher.getDepthFromRectifiedMatches(left, right, &depths);            // StereoMatcher.h:85-92
    std::vector<uint8_t> rs;
    for (const auto& k : right) rs.push_back((uint8_t)k.first);
    w.put("s_depth", depths.data(), depths.size() * 8);
    w.put("s_rstat", rs.data(), rs.size());
  } catch (const kvfe::Error& e) {
    std::fprintf(stderr, "kvfe::Error %d: %s\n", (int)e.status, e.what());
    std::fclose(fo);
    return 1;
  }
  std::fclose(fo);
  return 0;
}
