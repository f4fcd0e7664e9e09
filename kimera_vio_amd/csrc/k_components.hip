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

__global__ void distort_unrectify_kernel(const float2* __restrict__ map, int W, int H, const float2* __restrict__ rect_xy,
                                         const unsigned char* __restrict__ status, int n, float2* __restrict__ out_xy) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float2 o = make_float2(0.f, 0.f);
  if (status[i] == KVFE_KP_VALID) {
    // a VALID keypoint outside the image: upstream's map_x_.at<float>() is an unchecked read (undefined); here the map
    // entry of the nearest pixel (k_stereo.hip stereo_match_kernel, oracle/kimera.cpp)
    const float2 px = rect_xy[i];
    const int ry = min(max((int)roundf(px.y), 0), H - 1), rx = min(max((int)roundf(px.x), 0), W - 1);
    o = map[(size_t)ry * W + rx];
  }
  out_xy[i] = o;
}

void launch_distort_unrectify(const float2* map, int W, int H, const float2* rect_xy, const unsigned char* status, int n,
                              float2* out_xy, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(distort_unrectify_kernel, dim3((n + 255) / 256), dim3(256), 0, st, map, W, H, rect_xy, status, n,
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

// ---- output side (kvfe_dev.hpp "output side") ------------------------------------------------------------------------
// One block per stream: the header by thread 0, the arrays by coalesced element copies.  `dst` is a device staging
// buffer (many streams: the records then cross PCIe in one DMA transfer on the output stream, beside the next step's
// tracking launch) or the mapped pinned ring slot itself (a few streams: one launch less on the latency path).
// `rec_cap` = entries a record has room for (the host side cuts with the same number).
// The records of a step lie back to back behind a table of their offsets (kvfe_dev.hpp "output side"): a block derives
// its own offset from the counts of the streams before it (at most 63 triples of ints).
template <typename T>
__device__ __forceinline__ void out_copy_arr(unsigned char* rec, size_t off, const T* __restrict__ src, size_t n) {
  T* d = reinterpret_cast<T*>(rec + off);
  for (size_t i = threadIdx.x; i < n; i += blockDim.x) d[i] = src[i];
}

__global__ __launch_bounds__(256) void out_pack_kernel(KParams P, FrameTab K, StereoTab ST, StreamState S,
                                                       unsigned char* __restrict__ dst, size_t table_bytes, int rec_cap) {
  const int s = blockIdx.x;
  if (P.quiet_gate && !kvfe_all_quiet(S.flags, P.B)) return;
  __shared__ unsigned long long sh_off;
  if (threadIdx.x == 0) sh_off = 0ull;
  __syncthreads();
  for (int t = threadIdx.x; t < s; t += blockDim.x)
    atomicAdd(&sh_off, (unsigned long long)out_record_bytes(min(K.count[t], rec_cap), min(S.n_meas[t], rec_cap),
                                                            (S.flags[t] & FLAG_STEREO) != 0));
  __syncthreads();
  const size_t off = table_bytes + (size_t)sh_off;
  unsigned char* rec = dst + off;
  const int flags = S.flags[s];
  const int n = min(K.count[s], rec_cap), m = min(S.n_meas[s], rec_cap);
  const bool stereo = (flags & FLAG_STEREO) != 0;
  const OutLayout L = out_layout(n, m, stereo);
  if (threadIdx.x == 0) {
    OutHeader* h = reinterpret_cast<OutHeader*>(rec);
    h->n_keypoints = K.count[s];
    h->flags = flags | (K.count[s] > rec_cap ? FLAG_OVERFLOW : 0);   // (cannot happen while pts_bound holds; loud if it does)
    h->n_tracked = S.n_tracked[s];
    h->n_detected = S.n_detected[s];
    h->n_meas = S.n_meas[s];
    h->trk_status[0] = S.trk_status[2 * s];
    h->trk_status[1] = S.trk_status[2 * s + 1];
    for (int i = 0; i < 6; i++) h->trk_counts[i] = S.trk_counts[6 * s + i];
    h->pnp_status = S.pnp_status[s];
    for (int i = 0; i < 3; i++) h->pnp_counts[i] = S.pnp_counts[3 * s + i];
    h->frame_count = S.frame_count[s];
    h->used_bytes = (unsigned long long)L.end;
    unsigned long long* tab = reinterpret_cast<unsigned long long*>(dst);
    tab[s] = (unsigned long long)off;
    if (s == (int)gridDim.x - 1) tab[gridDim.x] = (unsigned long long)(off + out_record_bytes(n, m, stereo));   // end of the step's data
  } else if (threadIdx.x >= 64 && threadIdx.x < 64 + 45) {
    OutHeader* h = reinterpret_cast<OutHeader*>(rec);
    const int i = threadIdx.x - 64;
    if (i < 24) h->trk_pose[i] = S.trk_pose[24 * (size_t)s + i];
    else if (i < 33) h->trk_info[i - 24] = S.trk_info[9 * (size_t)s + i - 24];
    else h->pnp_pose[i - 33] = S.pnp_pose[12 * (size_t)s + i - 33];
  }
  const size_t so = (size_t)s * P.kcap;
  out_copy_arr(rec, L.lmk, K.lmk + so, (size_t)n);
  out_copy_arr(rec, L.age, K.age + so, (size_t)n);
  out_copy_arr(rec, L.kp, K.kp + so, (size_t)n);
  out_copy_arr(rec, L.versor, K.versor + so * 3, (size_t)n * 3);
  if (stereo) {
    out_copy_arr(rec, L.left_rect, ST.left_rect + so, (size_t)n);
    out_copy_arr(rec, L.left_status, ST.left_status + so, (size_t)n);
    out_copy_arr(rec, L.right_rect, ST.right_rect + so, (size_t)n);
    out_copy_arr(rec, L.right_status, ST.right_status + so, (size_t)n);
    out_copy_arr(rec, L.depth, ST.depth + so, (size_t)n);
    out_copy_arr(rec, L.right_kp, ST.right_kp + so, (size_t)n);
    out_copy_arr(rec, L.kp3d, ST.kp3d + so * 3, (size_t)n * 3);
  }
  out_copy_arr(rec, L.meas_lmk, S.meas_lmk + so, (size_t)m);
  out_copy_arr(rec, L.meas, S.meas_uLuRv + so * 3, (size_t)m * 3);
}

void launch_out_pack(const KParams& P, const FrameTab& k, const StereoTab& ST, const StreamState& S, unsigned char* dst,
                     size_t table_bytes, int rec_cap, hipStream_t st) {
  hipLaunchKernelGGL(out_pack_kernel, dim3((unsigned)P.B), dim3(256), 0, st, P, k, ST, S, dst, table_bytes, rec_cap);
}

// ---------------------------------------------------------------------------------------------
// kvfe_hbm_copy_probe (measurement hook, bench.py): what a plain streaming copy reaches on THIS device in THIS process.
// The dense kernels' roofline fractions are quoted against the guide's 8 TB/s; boxes of this pool differ by up to 2 x on
// the same binary, and this number beside them tells a slow box from a regression.  One 16-byte load and one 16-byte
// store per lane and trip, grid-stride, 2048 workgroups of 256 (8 per compute unit).
// ---------------------------------------------------------------------------------------------
typedef float probe_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void hbm_copy_kernel(const probe_f4* __restrict__ src, probe_f4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {   // four independent requests in flight per lane
    const probe_f4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
    const probe_f4 c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
    __builtin_nontemporal_store(a, dst + i);
    __builtin_nontemporal_store(b, dst + i + stride);
    __builtin_nontemporal_store(c, dst + i + 2 * stride);
    __builtin_nontemporal_store(d, dst + i + 3 * stride);
  }
  for (; i < n; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

}  // namespace kvfe

extern "C" KVFE_API kvfe_status kvfe_hbm_copy_probe(size_t bytes, int32_t iters, double* read_plus_write_GBps,
                                                    double* ms_per_copy) {
  if (bytes < 16 || iters < 1 || !read_plus_write_GBps) return KVFE_ERR_INVALID_ARG;
  const size_t n = bytes / 16;
  kvfe::probe_f4 *src = nullptr, *dst = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipStream_t st = nullptr;
  kvfe_status rc = KVFE_ERR_HIP;
  float ms = 0.f;
  if (hipMalloc(&src, n * 16) != hipSuccess || hipMalloc(&dst, n * 16) != hipSuccess) goto out;
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) goto out;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) goto out;
  if (hipMemsetAsync(src, 0x5a, n * 16, st) != hipSuccess || hipMemsetAsync(dst, 0, n * 16, st) != hipSuccess) goto out;
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL(kvfe::hbm_copy_kernel, dim3(2048), dim3(256), 0, st, src, dst, n);
  if (hipEventRecord(e0, st) != hipSuccess) goto out;
  for (int i = 0; i < iters; i++) hipLaunchKernelGGL(kvfe::hbm_copy_kernel, dim3(2048), dim3(256), 0, st, src, dst, n);
  if (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) goto out;
  if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess || !(ms > 0.f)) goto out;
  {
    unsigned char probe[16];
    if (hipMemcpy(probe, reinterpret_cast<unsigned char*>(dst) + (n - 1) * 16, 16, hipMemcpyDeviceToHost) != hipSuccess) goto out;
    for (unsigned char b : probe)
      if (b != 0x5a) goto out;   // (the copy really ran)
  }
  *read_plus_write_GBps = 2.0 * (double)(n * 16) * iters / ((double)ms * 1e6);
  if (ms_per_copy) *ms_per_copy = (double)ms / iters;
  rc = KVFE_OK;
out:
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (st) (void)hipStreamDestroy(st);
  if (src) (void)hipFree(src);
  if (dst) (void)hipFree(dst);
  return rc;
}
