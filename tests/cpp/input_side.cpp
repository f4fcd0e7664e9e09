// Host program over include/kvfe_adapter.hpp only (plain g++, links libkvfe.so): the input-side classes with the
// reference's names, driven through cases of tests/testStereoProvider.cpp and tests/testThreadsafeImuBuffer.cpp.
// Prints one line per check; tests/test_host_logic.py compares them with the expected transcript.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>

#include "kvfe_adapter.hpp"

using kvfe::StereoDataProviderModule;
using kvfe::StereoImuSyncPacket;
using Buf = kvfe::utils::ThreadsafeImuBuffer;

static void spin(const char* what, StereoDataProviderModule& m) {
  StereoImuSyncPacket p;
  const bool got = m.getInputPacket(&p);
  std::printf("%s: %s action=%d", what, got ? "packet" : "none", m.lastAction());
  if (got) {
    std::printf(" t=%lld tags=%lld,%lld imu=", (long long)p.timestamp, (long long)p.left_frame_tag,
                (long long)p.right_frame_tag);
    for (size_t i = 0; i < p.imu.timestamps.size(); i++) std::printf("%s%lld", i ? "," : "", (long long)p.imu.timestamps[i]);
  }
  std::printf("\n");
}

int main(int argc, char** argv) {
  const double zero[6] = {0, 0, 0, 0, 0, 0};
  {   // testStereoProvider.cpp:526-560 dropRightFrame
    StereoDataProviderModule m;
    int64_t id = 0;
    auto frame = [&](int64_t t, bool right = true) {
      m.fillLeftFrameQueue(t, id);
      if (right) m.fillRightFrameQueue(t, id);
      id++;
    };
    m.fillImuQueue(0, zero);
    frame(1);
    spin("first", m);
    for (int64_t t : {2, 3, 4}) m.fillImuQueue(t, zero);
    frame(5, false);
    m.fillImuQueue(6, zero);
    m.fillImuQueue(7, zero);
    frame(8);
    m.fillImuQueue(9, zero);
    spin("no right", m);
    spin("valid", m);
    spin("empty", m);
  }
  {   // testThreadsafeImuBuffer.cpp:194-279
    Buf b(-1);
    for (int64_t t : {10, 15, 20, 25, 30, 40, 50}) {
      const double v[6] = {(double)t, (double)t, (double)t, (double)t, (double)t, (double)t};
      b.addMeasurement(t, v);
    }
    kvfe::ImuMeasurements m;
    const Buf::QueryResult r = b.getImuDataInterpolatedBorders(21, 29, &m);
    std::printf("borders(21,29): result=%d cols=%d stamps=%lld,%lld,%lld values=%g,%g,%g\n", (int)r, m.cols(),
                (long long)m.timestamps[0], (long long)m.timestamps[1], (long long)m.timestamps[2], m.acc_gyr[0],
                m.acc_gyr[6], m.acc_gyr[12]);
    const Buf::QueryResult r2 = b.getImuDataInterpolatedBorders(40, 51, &m);
    std::printf("borders(40,51): result=%d cols=%d\n", (int)r2, m.cols());
    const Buf::QueryResult r3 = b.getImuDataBtwTimestamps(21, 24, &m);
    std::printf("between(21,24): result=%d cols=%d\n", (int)r3, m.cols());
  }
  {   // tests/testFrame.cpp:82-99 getNrValidKeypoints, :186-200 findLmkIdFromPixel
    kvfe::Frame f;
    for (int i = 0; i < 200; i++) {
      if (i % 5 == 0) f.landmarks_.push_back(-1);
      f.landmarks_.push_back(i);
    }
    f.keypoints_.resize(f.landmarks_.size());
    for (size_t i = 0; i < f.keypoints_.size(); i++) f.keypoints_[i] = kvfe::KeypointCV{(float)(3 * i) + 0.5f, (float)i};
    size_t idx = 0;
    const long long id = kvfe::Frame::findLmkIdFromPixel(f.keypoints_[7], f.keypoints_, f.landmarks_, &idx);
    std::printf("frame: valid=%zu valid_kps=%zu lmk_of_px7=%lld idx=%zu missing=%lld\n", f.getNrValidKeypoints(),
                f.getValidKeypoints().size(), id, idx,
                (long long)kvfe::Frame::findLmkIdFromPixel(kvfe::KeypointCV{-1.f, -1.f}, f.keypoints_, f.landmarks_));
  }
  if (argc > 2 && argv[2][0]) {   // an EuRoC-layout dataset: EurocDataProvider -> StereoDataProviderModule -> packets
    kvfe::EurocDataProvider prov(argv[2], 0, 1 << 30);
    StereoDataProviderModule sync;
    unsigned long long pix = 0;
    prov.registerImuSingleCallback([&](int64_t t, const double* ag) { sync.fillImuQueue(t, ag); });
    prov.registerLeftFrameCallback([&](int64_t k, int64_t t, const uint8_t* img, int rows, int cols) {
      for (int i = 0; i < rows * cols; i++) pix += img[i];
      sync.fillLeftFrameQueue(t, k);
    });
    prov.registerRightFrameCallback([&](int64_t k, int64_t t, const uint8_t* img, int rows, int cols) {
      for (int i = 0; i < rows * cols; i++) pix += 3ull * img[i];
      sync.fillRightFrameQueue(t, k);
    });
    prov.spin();
    std::printf("dataset: images=%zu imu=%d pixel_sum=%llu\n", prov.getNumImages(), prov.imuMeasurements().cols(), pix);
    for (;;) {
      StereoImuSyncPacket p;
      const bool got = sync.getInputPacket(&p);
      if (!got && sync.lastAction() == KVFE_SYNC_EMPTY) break;
      if (got)
        std::printf("packet: t=%lld tag=%lld,%lld n_imu=%d first=%lld last=%lld acc0=%.17g\n", (long long)p.timestamp,
                    (long long)p.left_frame_tag, (long long)p.right_frame_tag, p.imu.cols(), (long long)p.imu.timestamps.front(),
                    (long long)p.imu.timestamps.back(), p.imu.acc_gyr[0]);
      else
        std::printf("dropped: action=%d\n", sync.lastAction());
      if (!got && sync.lastAction() == KVFE_SYNC_WAIT_IMU) break;   // (upstream would block here until IMU data arrives)
    }
  }
  if (argc > 3) {   // a text file with the two cameras' parameters: rectified body poses, stereo calibration, relative pose
    std::ifstream f(argv[3]);
    kvfe_camera_params cam[2];
    std::memset(cam, 0, sizeof(cam));
    for (int c = 0; c < 2; c++) {
      f >> cam[c].width >> cam[c].height;
      for (int i = 0; i < 4; i++) f >> cam[c].intrinsics[i];
      cam[c].distortion_model = KVFE_DIST_RADTAN;
      cam[c].n_distortion = 4;
      for (int i = 0; i < 4; i++) f >> cam[c].distortion[i];
      for (int i = 0; i < 16; i++) f >> cam[c].body_pose_cam[i];
    }
    kvfe_rectification r;
    if (kvfe_compute_rectification(&cam[0], &cam[1], &r) != KVFE_OK) return 3;
    const kvfe::Pose3 bl = kvfe::bodyPoseCamRect(r.R1, cam[0]), br = kvfe::bodyPoseCamRect(r.R2, cam[0]);
    const kvfe::Cal3_S2Stereo k = kvfe::stereoCalib(r);
    const double lkf_T_k[12] = {0.9998, -0.01, 0.015, 0.05, 0.0101, 0.99995, -0.002, -0.02, -0.01498, 0.00215, 0.99989, 0.3};
    const kvfe::Pose3 rel = kvfe::relativePoseBody(bl, lkf_T_k);
    auto show = [](const char* name, const kvfe::Pose3& p) {
      std::printf("%s:", name);
      for (int i = 0; i < 9; i++) std::printf(" %.17g", p.R[i]);
      for (int i = 0; i < 3; i++) std::printf(" %.17g", p.t[i]);
      std::printf("\n");
    };
    show("B_Pose_camLrect", bl);
    show("B_Pose_camRrect", br);
    show("relative_pose_body", rel);
    std::printf("stereo_calib: %.17g %.17g %.17g %.17g %.17g %.17g\n", k.fx, k.fy, k.skew, k.px, k.py, k.baseline);
  }
  if (argc > 1 && argv[1][0]) {   // a PNG file: size and a checksum of the decoded grey image
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    int rows = 0, cols = 0;
    const std::vector<uint8_t> img = kvfe::ReadAndConvertToGrayScale(data.data(), data.size(), &rows, &cols);
    unsigned long long sum = 0;
    for (size_t i = 0; i < img.size(); i++) sum += (unsigned long long)img[i] * (i % 251 + 1);
    std::printf("png: %dx%d checksum=%llu\n", cols, rows, sum);
  }
  return 0;
}
