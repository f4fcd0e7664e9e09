// adapter_sequence: a C++ host program written the way a Kimera-VIO maintainer would use libkvfe —
// only through include/kvfe_adapter.hpp, i.e. through the reference's own class and method names
// (FeatureDetector::featureDetection, Tracker::featureTracking,
// StereoMatcher::sparseStereoReconstruction, UndistorterRectifier::undistortRectifyImage,
// StereoCamera::getBaseline, StereoVisionImuFrontend::spinOnce).  tests/test_gpu_parity.py feeds it
// EuRoC frames and compares what it writes with the CPU oracle, so the C++ side of the drop-in
// boundary is exercised end to end, not just the ctypes mirror.
//
//   adapter_sequence <in.bin> <out.bin>
// in.bin : kvfe_config | int32 n_frames, W, H | n_frames x { kvfe_frame_input, left[W*H], right[W*H] }
// out.bin: a flat sequence of records { char tag[16]; int64 nbytes; payload }
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "kvfe_adapter.hpp"

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
  template <class T>
  void vec(const char* tag, const std::vector<T>& v) { put(tag, v.data(), v.size() * sizeof(T)); }
  template <class T>
  void val(const char* tag, const T& v) { put(tag, &v, sizeof(T)); }
};
bool read_exact(FILE* f, void* p, size_t n) { return std::fread(p, 1, n, f) == n; }
}  // namespace

int main(int argc, char** argv) {
  if (argc != 3) {
    std::fprintf(stderr, "usage: %s <in.bin> <out.bin>\n", argv[0]);
    return 2;
  }
  FILE* fi = std::fopen(argv[1], "rb");
  FILE* fo = std::fopen(argv[2], "wb");
  if (!fi || !fo) return 2;
  kvfe_config cfg;
  int32_t hdr[3];
  if (!read_exact(fi, &cfg, sizeof(cfg)) || !read_exact(fi, hdr, sizeof(hdr))) return 2;
  const int n_frames = hdr[0], W = hdr[1], H = hdr[2];
  const size_t N = (size_t)W * H;
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
  try {
    kvfe::Context ctx(cfg.left, cfg.right, cfg.params, /*batch=*/1, cfg.device);
    auto view = [&](const std::vector<uint8_t>& v) { return kvfe::ImageView{v.data(), H, W, (size_t)W}; };

    // ---- the four classes north_star names, one call each, as the reference's unit tests do ----
    kvfe::StereoCamera stereo_camera(ctx);
    w.val("baseline", stereo_camera.getBaseline());
    w.put("P1", stereo_camera.getP1(), 12 * sizeof(double));

    kvfe::UndistorterRectifier left_rectifier(ctx, 0);
    std::vector<uint8_t> left_rect(N);
    left_rectifier.undistortRectifyImage(view(lefts[0]), left_rect.data());
    w.vec("left_rect", left_rect);

    kvfe::FeatureDetector feature_detector(ctx);
    const kvfe::KeypointsCV none;
    const kvfe::KeypointsCV corners = feature_detector.featureDetection(
        view(lefts[0]), none, cfg.params.detector.max_features_per_frame);
    w.vec("corners", corners);
    std::vector<double> versors;
    left_rectifier.getBearingVectors(corners, &versors);
    w.vec("versors", versors);

    if (n_frames > 1) {
      kvfe::Tracker tracker(ctx);
      kvfe::KeypointsCV px_cur;
      std::vector<uint8_t> status;
      std::vector<float> error;
      tracker.featureTracking(view(lefts[0]), view(lefts[1]), corners, inputs[1].keyframe_R_cur_frame,
                              &px_cur, &status, &error);
      w.vec("lk_px", px_cur);
      w.vec("lk_status", status);
      w.vec("lk_err", error);
    }

    kvfe::StereoMatcher stereo_matcher(ctx);
    const auto sr = stereo_matcher.sparseStereoReconstruction(view(lefts[0]), view(rights[0]), corners);
    {
      std::vector<uint8_t> ls, rs;
      kvfe::KeypointsCV lr, rr;
      for (const auto& k : sr.left_keypoints_rectified) {
        ls.push_back((uint8_t)k.first);
        lr.push_back(k.second);
      }
      for (const auto& k : sr.right_keypoints_rectified) {
        rs.push_back((uint8_t)k.first);
        rr.push_back(k.second);
      }
      w.vec("st_lstat", ls);
      w.vec("st_lrect", lr);
      w.vec("st_rstat", rs);
      w.vec("st_rrect", rr);
      w.vec("st_depth", sr.keypoints_depth);
      w.vec("st_3d", sr.keypoints_3d);
    }

    // ---- the remaining public keypoint / frame methods (round 2), each called once ---------------
    {
      auto put_status = [&](const char* tag_st, const char* tag_xy, const kvfe::StatusKeypointsCV& v) {
        std::vector<uint8_t> st;
        kvfe::KeypointsCV xy;
        for (const auto& k : v) {
          st.push_back((uint8_t)k.first);
          xy.push_back(k.second);
        }
        w.vec(tag_st, st);
        w.vec(tag_xy, xy);
      };
      // StereoCamera::undistortRectifyLeftKeypoints
      kvfe::StatusKeypointsCV left_rect_kps;
      stereo_camera.undistortRectifyLeftKeypoints(corners, &left_rect_kps);
      put_status("r2_url_st", "r2_url_xy", left_rect_kps);
      // UndistorterRectifier::undistortRectifyKeypoints + checkUndistortedRectifiedLeftKeypoints (tolerance 0.5)
      kvfe::KeypointsCV und;
      left_rectifier.undistortRectifyKeypoints(corners, &und);
      kvfe::StatusKeypointsCV checked;
      left_rectifier.checkUndistortedRectifiedLeftKeypoints(corners, und, &checked, 0.5f);
      put_status("r2_chk_st", "r2_chk_xy", checked);
      // StereoCamera::undistortRectifyStereoFrame
      std::vector<uint8_t> lrect(N), rrect(N);
      stereo_camera.undistortRectifyStereoFrame(view(lefts[0]), view(rights[0]), lrect.data(), rrect.data());
      w.vec("r2_lrect", lrect);
      w.vec("r2_rrect", rrect);
      // StereoMatcher::getRightKeypointsRectified + getDepthFromRectifiedMatches + distortUnrectifyRightKeypoints:
      // the steps of sparseStereoReconstruction one by one
      kvfe::StatusKeypointsCV right_rect_kps;
      stereo_matcher.getRightKeypointsRectified(view(lrect), view(rrect), left_rect_kps, &right_rect_kps);
      std::vector<double> depths;
      stereo_matcher.getDepthFromRectifiedMatches(left_rect_kps, right_rect_kps, &depths);
      put_status("r2_dep_st", "r2_dep_xy", right_rect_kps);
      w.vec("r2_depth", depths);
      kvfe::KeypointsCV right_kps;
      stereo_camera.distortUnrectifyRightKeypoints(right_rect_kps, &right_kps);
      w.vec("r2_rkps", right_kps);
      // FeatureDetector::featureDetection(Frame*) -> Tracker::featureTracking(Frame*, Frame*, R) -> featureDetection
      auto put_frame = [&](const std::string& pre, const kvfe::Frame& f) {
        w.vec((pre + "_kp").c_str(), f.keypoints_);
        w.vec((pre + "_lmk").c_str(), f.landmarks_);
        w.vec((pre + "_age").c_str(), f.landmarks_age_);
        w.vec((pre + "_ver").c_str(), f.versors_);
      };
      kvfe::FeatureDetector::landmarkCounter() = 0;
      kvfe::Frame f0, f1;
      f0.img_ = view(lefts[0]);
      feature_detector.featureDetection(&f0);
      put_frame("r2_f0", f0);
      if (n_frames > 1) {
        kvfe::Tracker tracker2(ctx);
        f1.img_ = view(lefts[1]);
        tracker2.featureTracking(&f0, &f1, inputs[1].keyframe_R_cur_frame);
        w.vec("r2_ref_lmk", f0.landmarks_);
        put_frame("r2_f1t", f1);
        feature_detector.featureDetection(&f1);
        put_frame("r2_f1d", f1);
        w.val("r2_counter", kvfe::FeatureDetector::landmarkCounter());
      }
    }

    // ---- MonoVisionImuFrontend::spinOnce on the left images (its own context) ----------------------------
    {
      kvfe::MonoVisionImuFrontend mono(kvfe::Context(kvfe::MonoVisionImuFrontend::config(cfg.left, cfg.params, 1, cfg.device)));
      const int cap = 4096;
      std::vector<int64_t> landmarks(cap), meas_lmk(cap);
      std::vector<float> kp(2 * cap);
      std::vector<double> meas(3 * cap);
      for (int i = 0; i < std::min(n_frames, 3); i++) {
        mono.spinOnce(lefts[i].data(), (size_t)W, N, &inputs[i]);
        kvfe_frame_output o;
        std::memset(&o, 0, sizeof(o));
        o.capacity = cap;
        o.landmarks = landmarks.data();
        o.keypoints = kp.data();
        o.meas_landmark = meas_lmk.data();
        o.meas_uL_uR_v = meas.data();
        mono.getOutput(0, &o);
        const int32_t head[4] = {o.n_keypoints, o.is_keyframe, o.n_tracked, o.n_measurements};
        w.put("m_head", head, sizeof(head));
        w.put("m_lmk", landmarks.data(), (size_t)o.n_keypoints * 8);
        w.put("m_kp", kp.data(), (size_t)o.n_keypoints * 8);
        w.put("m_meas", meas.data(), (size_t)o.n_measurements * 24);
      }
    }

    // ---- StereoVisionImuFrontend::spinOnce over the sequence -----------------------------------
    kvfe::StereoVisionImuFrontend frontend(ctx);
    const int cap = 4096;
    std::vector<int64_t> landmarks(cap), meas_lmk(cap);
    std::vector<int32_t> age(cap);
    std::vector<float> kp(2 * cap), lrect(2 * cap), rrect(2 * cap), rxy(2 * cap);
    std::vector<double> vers(3 * cap), depth(cap), p3d(3 * cap), meas(3 * cap);
    std::vector<uint8_t> lstat(cap), rstat(cap);
    // PnP tracking next to the step (use_pnp_tracking: StereoVisionImuFrontend.cpp:389-399): the map the backend would
    // hand over through Tracker::updateMap is played by the 3-D points of frame 0 (world = camera frame of frame 0)
    kvfe::Tracker pnp_tracker(ctx);
    kvfe::Tracker::LandmarksMap lmk_map;
    kvfe_pnp_params pnp_params;
    std::memset(&pnp_params, 0, sizeof(pnp_params));
    pnp_params.pnp_algorithm = 3;   // EPNP
    pnp_params.min_pnp_inliers = 20;
    pnp_params.ransac_threshold_pnp = 1.0;
    for (int i = 0; i < n_frames; i++) {
      frontend.spinOnce(lefts[i].data(), rights[i].data(), (size_t)W, N, &inputs[i]);
      kvfe_frame_output o;
      std::memset(&o, 0, sizeof(o));
      o.capacity = cap;
      o.landmarks = landmarks.data();
      o.landmarks_age = age.data();
      o.keypoints = kp.data();
      o.versors = vers.data();
      o.left_rect_xy = lrect.data();
      o.left_status = lstat.data();
      o.right_rect_xy = rrect.data();
      o.right_status = rstat.data();
      o.depth = depth.data();
      o.right_xy = rxy.data();
      o.keypoints_3d = p3d.data();
      o.meas_landmark = meas_lmk.data();
      o.meas_uL_uR_v = meas.data();
      frontend.getOutput(0, &o);
      const int32_t head[6] = {o.n_keypoints, o.is_keyframe, o.n_tracked, o.n_detected, o.n_measurements,
                               (int32_t)o.frame_id};
      const size_t n = (size_t)o.n_keypoints, m = (size_t)o.n_measurements;
      w.put("f_head", head, sizeof(head));
      w.put("f_lmk", landmarks.data(), n * 8);
      w.put("f_age", age.data(), n * 4);
      w.put("f_kp", kp.data(), n * 8);
      w.put("f_versors", vers.data(), n * 24);
      w.put("f_lrect", lrect.data(), n * 8);
      w.put("f_lstat", lstat.data(), n);
      w.put("f_rrect", rrect.data(), n * 8);
      w.put("f_rstat", rstat.data(), n);
      w.put("f_depth", depth.data(), n * 8);
      w.put("f_3d", p3d.data(), n * 24);
      w.put("f_mlmk", meas_lmk.data(), m * 8);
      w.put("f_meas", meas.data(), m * 24);
      const int32_t trk[6] = {o.tracking_status_mono, o.tracking_status_stereo, o.nr_mono_putatives,
                              o.nr_mono_inliers, o.nr_stereo_putatives, o.nr_stereo_inliers};
      w.put("f_trk", trk, sizeof(trk));
      w.put("f_Tmono", o.lkf_T_k_mono, sizeof(o.lkf_T_k_mono));
      w.put("f_Tstereo", o.lkf_T_k_stereo, sizeof(o.lkf_T_k_stereo));
      if (i == 0) {
        for (size_t q = 0; q < n; q++)
          if (rstat[q] == KVFE_KP_VALID && landmarks[q] != -1)
            lmk_map[landmarks[q]] = {p3d[3 * q], p3d[3 * q + 1], p3d[3 * q + 2]};
        pnp_tracker.updateMap(lmk_map);
      } else if (o.is_keyframe) {
        const kvfe::Tracker::TrackingStatusPose sp = pnp_tracker.outlierRejectionPnP(o, pnp_params);
        const int32_t ps[2] = {sp.status, (int32_t)lmk_map.size()};
        w.put("p_status", ps, sizeof(ps));
        w.put("p_pose", sp.pose, sizeof(sp.pose));
      }
    }

    // ---- error behaviour: a contract violation surfaces as kvfe::Error, never as a crash --------
    int32_t threw = 0;
    try {
      kvfe::ImageView bad{nullptr, H, W, (size_t)W};
      feature_detector.rawFeatureDetection(bad);
    } catch (const kvfe::Error& e) {
      threw = (int32_t)e.status;
    }
    w.val("err_status", threw);
  } catch (const kvfe::Error& e) {
    std::fprintf(stderr, "kvfe::Error %d: %s\n", (int)e.status, e.what());
    std::fclose(fo);
    return 1;
  }
  std::fclose(fo);
  return 0;
}
