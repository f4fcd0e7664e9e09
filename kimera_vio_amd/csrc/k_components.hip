// Component-level keypoint methods of UndistorterRectifier / StereoCamera / StereoMatcher on their own, i.e. the
// pieces that the front-end step runs fused inside stereo_left_kernel / stereo_match_kernel / track_finalize:
//   UndistorterRectifier::checkUndistortedRectifiedLeftKeypoints   src/frontend/UndistorterRectifier.cpp:138-211
//   UndistorterRectifier::distortUnrectifyKeypoints                 src/frontend/UndistorterRectifier.cpp:213-228
//   StereoMatcher::getDepthFromRectifiedMatches                     src/frontend/StereoMatcher.cpp:425-483
//   Tracker::featureTracking's `ref_frame->landmarks_[i] = -1`      src/frontend/Tracker.cpp:167-178
// One lane per keypoint; pure table look-ups and a few float / double operations in the reference's order.
#include "../../include/kvfe.h"
#include "kvfe_dev.hpp"

namespace kvfe {

__global__ void check_undistorted_rectified_kernel(const float2* __restrict__ map, int W, int H,
                                                   const float2* __restrict__ distorted,
                                                   const float2* __restrict__ undistorted, int n, float pixel_tol,
                                                   float2* __restrict__ out_xy, unsigned char* __restrict__ out_status) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float2 d = distorted[i];
  float ux = undistorted[i].x, uy = undistorted[i].y;
  // cropToSize (UtilsOpenCV.cpp:215-235)
  bool cropped = false;
  const float maxw = (float)(W - 1), maxh = (float)(H - 1);
  if (ux > maxw) {
    ux = maxw;
    cropped = true;
  } else if (ux < 0.0f) {
    ux = 0.0f;
    cropped = true;
  }
  if (uy > maxh) {
    uy = maxh;
    cropped = true;
  } else if (uy < 0.0f) {
    uy = 0.0f;
    cropped = true;
  }
  const int ry = (int)roundf(uy), rx = (int)roundf(ux);
  const float2 e = map[(size_t)ry * W + rx];
  unsigned char status = KVFE_KP_VALID;
  if (cropped || fabsf(d.x - e.x) > pixel_tol || fabsf(d.y - e.y) > pixel_tol) status = KVFE_KP_NO_LEFT_RECT;
  out_xy[i] = make_float2(ux, uy);
  out_status[i] = status;
}

void launch_check_undistorted_rectified(const float2* map, int W, int H, const float2* distorted,
                                        const float2* undistorted, int n, float pixel_tol, float2* out_xy,
                                        unsigned char* out_status, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(check_undistorted_rectified_kernel, dim3((n + 255) / 256), dim3(256), 0, st, map, W, H,
                     distorted, undistorted, n, pixel_tol, out_xy, out_status);
}

__global__ void distort_unrectify_kernel(const float2* __restrict__ map, int W, const float2* __restrict__ rect_xy,
                                         const unsigned char* __restrict__ status, int n, float2* __restrict__ out_xy) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float2 o = make_float2(0.f, 0.f);
  if (status[i] == KVFE_KP_VALID) {
    const float2 px = rect_xy[i];
    o = map[(size_t)(int)roundf(px.y) * W + (int)roundf(px.x)];
  }
  out_xy[i] = o;
}

void launch_distort_unrectify(const float2* map, int W, const float2* rect_xy, const unsigned char* status, int n,
                              float2* out_xy, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(distort_unrectify_kernel, dim3((n + 255) / 256), dim3(256), 0, st, map, W, rect_xy, status, n,
                     out_xy);
}

__global__ void depth_from_matches_kernel(const float2* __restrict__ left_xy, const unsigned char* __restrict__ left_status,
                                          const float2* __restrict__ right_xy, unsigned char* __restrict__ right_status,
                                          int n, double fx_b, double min_dist, double max_dist,
                                          double* __restrict__ depth) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned char ls = left_status[i];
  unsigned char rs = right_status[i];
  double d = 0.0;
  if (ls == KVFE_KP_VALID && rs == KVFE_KP_VALID) {
    const double disparity = (double)(left_xy[i].x - right_xy[i].x);   // float subtraction, then widened
    if (disparity >= 0.0) {
      const double z = fx_b / disparity;
      if (z < min_dist || z > max_dist)
        rs = KVFE_KP_NO_DEPTH;
      else
        d = z;
    } else {
      rs = KVFE_KP_NO_DEPTH;
    }
  } else if (ls != KVFE_KP_VALID && rs != ls) {
    rs = ls;   // "cannot have a valid right keypoint without a valid left keypoint"
  }
  right_status[i] = rs;
  depth[i] = d;
}

void launch_depth_from_matches(const float2* left_xy, const unsigned char* left_status, const float2* right_xy,
                               unsigned char* right_status, int n, double fx_b, double min_dist, double max_dist,
                               double* depth, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(depth_from_matches_kernel, dim3((n + 255) / 256), dim3(256), 0, st, left_xy, left_status,
                     right_xy, right_status, n, fx_b, min_dist, max_dist, depth);
}

// Tracker.cpp:167-178: the reference marks lost / too old tracks in the REFERENCE frame (they guide detection later)
__global__ void mark_lost_tracks_kernel(KParams P, FrameTab KM1, LkScratch lk) {
  const int s = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= lk.npts[s]) return;
  const size_t so = (size_t)s * P.kcap;
  const int src = lk.src_idx[so + i];
  if (!lk.status[so + i] || KM1.age[so + src] > P.max_age) KM1.lmk[so + src] = -1;
}

void launch_mark_lost_tracks(const KParams& P, const FrameTab& km1, const LkScratch& lk, int max_pts, hipStream_t st) {
  if (max_pts <= 0) return;
  hipLaunchKernelGGL(mark_lost_tracks_kernel, dim3((max_pts + 255) / 256, P.B), dim3(256), 0, st, P, km1, lk);
}

}  // namespace kvfe
