// Host-side (init-time, float64) calibration math of libkvfe:
//   StereoCamera::computeRectificationParameters -> cv::stereoRectify
//     (reference: src/frontend/StereoCamera.cpp:292-379)
//   UndistorterRectifier::initUndistortRectifyMaps -> cv::initUndistortRectifyMap
//     (reference: src/frontend/UndistorterRectifier.cpp:230-292)
//   cv::undistortPoints (reference: src/frontend/UndistorterRectifier.cpp:33-68)
//   RotationalOpticalFlowPredictor homography (optical-flow/OpticalFlowPredictor.cpp:70-92)
//   cv::sortIdx permutation tables (feature-detector/NonMaximumSuppression.cpp:50-60)
//   cv::circle row spans (feature-detector/FeatureDetector.cpp:196-201)
// All of it runs once per context; the per-frame work is on the GPU.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/kvfe.h"

namespace kvfe {

struct M3 {
  double m[9];
  double& operator()(int r, int c) { return m[r * 3 + c]; }
  double operator()(int r, int c) const { return m[r * 3 + c]; }
  static M3 eye() { return M3{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
  M3 t() const {
    M3 o;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) o(r, c) = (*this)(c, r);
    return o;
  }
};
struct V3 {
  double v[3];
};

M3 mul(const M3& a, const M3& b);
V3 mul(const M3& a, const V3& x);
M3 inv3(const M3& a);
M3 rodrigues(const V3& r);
V3 rodrigues_inv(const M3& R);

M3 camera_matrix(const kvfe_camera_params& c);

// undistortion of one pixel, float in / float out:
//   has_dist 1 (radial-tangential, cv::undistortPoints): K^-1, 5 fixed-point iterations
//   has_dist 2 (equidistant, cv::fisheye::undistortPoints): K^-1, Newton iterations on theta, tan
// then RR = (P[:, :3] *) R applied.
struct UndistortCtx {
  double fx, fy, cx, cy, ifx, ify;
  double k[8];
  int has_dist;  // 0 none, 1 radtan, 2 equidistant
  M3 RR;
};
UndistortCtx make_undistort_ctx(const kvfe_camera_params& cam, const double* R /*9|null*/,
                                const double* P /*12|null*/);
void undistort_point(const UndistortCtx& u, float x_in, float y_in, float* x_out, float* y_out);

kvfe_status stereo_rectify(const kvfe_camera_params& left, const kvfe_camera_params& right,
                           kvfe_rectification* out);

kvfe_status init_undistort_rectify_map(const kvfe_camera_params& cam, const double R[9],
                                       const double P[12], float* map_x, float* map_y);

// cv::circle(FILLED) half-widths: pixel (x, y) is painted iff |y-cy| <= r and
// |x-cx| <= hw[|y-cy|].
std::vector<int> circle_half_widths(int radius);

// permutation that cv::sortIdx(all-equal keys, DESCENDING) applies to n elements.
void sortidx_permutation(int n, int policy, uint16_t* out);

}  // namespace kvfe
