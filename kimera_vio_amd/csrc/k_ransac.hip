// Geometric outlier rejection of keyframes (FrontendParams::useRANSAC_), device resident so that a
// keyframe still needs no host round trip:
//   VisionImuFrontend::outlierRejectionMono / outlierRejectionStereo  src/frontend/VisionImuFrontend.cpp:90-144
//   Tracker::geometricOutlierRejection2d2d (rotation given)           src/frontend/Tracker.cpp:213-378
//     = opengv::sac::Ransac<TranslationOnlySacProblem> (2-point), Tracker::runRansac, Tracker.h:247-296
//   Tracker::geometricOutlierRejection3d3dGivenRotation (1-point voting, the reference's own scheme)
//                                                                      src/frontend/Tracker.cpp:382-661
//   Tracker::getPoint3AndCovariance                                    src/frontend/Tracker.cpp:772-818
//   Tracker::findMatchingKeypoints / findMatchingStereoKeypoints / computeMedianDisparity / removeOutliers*
//                                                                      src/frontend/Tracker.cpp:856-1018
//
// One 256-thread workgroup per stream.  RANSAC iterations are sequential (adaptive stopping), the
// scoring of every hypothesis is parallel over the matches; the O(n^2) float32 Mahalanobis voting is
// parallel over rows.  All float64 / float32 expressions are evaluated in the order of the CPU path
// (oracle/opengv_re.cpp, oracle/kimera_ransac.cpp), no FMA contraction.  The random numbers of
// opengv's sampler (std::mt19937 seeded with 12345 through uniform_int_distribution) are a fixed
// stream, generated once on the host (Tables::ransac_rnd).
#include "kvfe_dev.hpp"

#include <algorithm>
#include <climits>

namespace kvfe {

constexpr int RS_T = 256;

__device__ __forceinline__ double rs_dot3(const double* a, const double* b) {
  return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}
__device__ __forceinline__ void rs_cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ void rs_matvec3(const double* R, const double* v, double* o) {
  for (int r = 0; r < 3; r++) o[r] = (R[r * 3] * v[0] + R[r * 3 + 1] * v[1]) + R[r * 3 + 2] * v[2];
}

// gtsam::Rot3::equals(Rot3(), 1e-9): ONE predicate for the device and for the host's "is the 3-point kernel needed at
// all" test (kvfe_dev.hpp: rot_is_identity)
__device__ __forceinline__ bool rs_rot_is_identity(const double* R) { return rot_is_identity(R); }

// exclusive scan of one int per thread over the 256-thread block; *total = block sum
__device__ __forceinline__ int rs_scan(int v, int* wave_tot /*[4]*/, int* total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int inc = v;
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(inc, off);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wave_tot[wv] = inc;
  __syncthreads();
  int base = 0, tot = 0;
  for (int i = 0; i < RS_T / 64; i++) {
    const int t = wave_tot[i];
    if (i < wv) base += t;
    tot += t;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}
__device__ __forceinline__ int rs_block_sum(int v, int* wave_tot) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = v;
  __syncthreads();
  const int tot = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
  __syncthreads();
  return tot;
}

// ---------------------------------------------------------------------------------------------
// Tracker::findMatchingKeypoints (+ findMatchingStereoKeypoints when ST/LST are given): matches
// (index in lkf, index in k) in the order of k's keypoints.  Valid landmark ids of a frame are
// strictly increasing, so the std::map lookup is a binary search over the compacted valid ids.
// LDS: ids [kcap] i64 | idx [kcap] i32.
// ---------------------------------------------------------------------------------------------
__device__ int rs_build_matches(const KParams& P, const FrameTab& K, const FrameTab& LKF,
                                const unsigned char* k_rstat, const unsigned char* l_rstat, int s,
                                int n_tracked, long long* ids, int* idx, int* wave_tot, int2* matches) {
  const int tid = threadIdx.x;
  const size_t so = (size_t)s * P.kcap;
  // only the tracked part of frame k: both rejections run before / concurrently with the detection
  // that appends new entries behind it (and whose landmark ids cannot match the last keyframe anyway)
  const int nl = LKF.count[s], nk = n_tracked;
  __shared__ int sh_off;
  if (tid == 0) sh_off = 0;
  __syncthreads();
  for (int base = 0; base < nl; base += RS_T) {
    const int j = base + tid;
    long long id = -1;
    if (j < nl) id = LKF.lmk[so + j];
    int tot;
    const int pos = rs_scan(id != -1 ? 1 : 0, wave_tot, &tot);
    const int off = sh_off;
    if (id != -1) {
      ids[off + pos] = id;
      idx[off + pos] = j;
    }
    __syncthreads();
    if (tid == 0) sh_off = off + tot;
    __syncthreads();
  }
  const int nv = sh_off;
  __syncthreads();
  if (tid == 0) sh_off = 0;
  __syncthreads();
  for (int base = 0; base < nk; base += RS_T) {
    const int i = base + tid;
    int j = -1;
    if (i < nk) {
      const long long id = K.lmk[so + i];
      if (id != -1) {
        int lo = 0, hi = nv - 1;
        while (lo <= hi) {
          const int mid = (lo + hi) >> 1;
          const long long v = ids[mid];
          if (v == id) {
            j = idx[mid];
            break;
          }
          if (v < id)
            lo = mid + 1;
          else
            hi = mid - 1;
        }
      }
      if (j >= 0 && k_rstat && !(l_rstat[so + j] == 0 && k_rstat[so + i] == 0)) j = -1;
    }
    int tot;
    const int pos = rs_scan(j >= 0 ? 1 : 0, wave_tot, &tot);
    const int off = sh_off;
    if (j >= 0) matches[off + pos] = make_int2(j, i);
    __syncthreads();
    if (tid == 0) sh_off = off + tot;
    __syncthreads();
  }
  const int n = sh_off;
  __syncthreads();
  return n;
}

// ---------------------------------------------------------------------------------------------
// opengv TranslationOnlySacProblem
// ---------------------------------------------------------------------------------------------
struct RsModel {
  double R[9], t[3], inv[12];  // model [R | t] and the inverse solution [R^T | -R^T t]
};

// relative_pose::twopt(adapter, unrotate = true, i0, i1)
__device__ void rs_twopt(const double* f1, const double* f2, const double* R12, int i0, int i1,
                         RsModel* M) {
  double a1[3], b1[3], fa[3], fb[3], a2[3], b2[3];
  for (int c = 0; c < 3; c++) {
    a1[c] = f1[3 * (size_t)i0 + c];
    b1[c] = f1[3 * (size_t)i1 + c];
    fa[c] = f2[3 * (size_t)i0 + c];
    fb[c] = f2[3 * (size_t)i1 + c];
  }
  rs_matvec3(R12, fa, a2);
  rs_matvec3(R12, fb, b2);
  double n1[3], n2[3], t[3];
  rs_cross3(a1, a2, n1);
  rs_cross3(b1, b2, n2);
  rs_cross3(n1, n2, t);
  const double nrm = sqrt(rs_dot3(t, t));
  for (int i = 0; i < 3; i++) t[i] = t[i] / nrm;
  const double flow[3] = {a1[0] - a2[0], a1[1] - a2[1], a1[2] - a2[2]};
  if (rs_dot3(flow, t) < 0)
    for (int i = 0; i < 3; i++) t[i] = -t[i];
  for (int i = 0; i < 9; i++) M->R[i] = R12[i];
  for (int i = 0; i < 3; i++) M->t[i] = t[i];
  double Rt[9], it[3];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) Rt[r * 3 + c] = R12[c * 3 + r];
  rs_matvec3(Rt, t, it);
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) M->inv[r * 4 + c] = Rt[r * 3 + c];
    M->inv[r * 4 + 3] = -it[r];
  }
}

// getSelectedDistancesToModel for one correspondence: triangulate2, reproject, 1 - cos in both views
__device__ double rs_distance(const RsModel& M, const double* a, const double* b) {
  double bu[3];
  rs_matvec3(M.R, b, bu);
  const double b0 = rs_dot3(M.t, a), b1 = rs_dot3(M.t, bu);
  const double A00 = rs_dot3(a, a), A10 = rs_dot3(a, bu), A01 = -A10, A11 = -rs_dot3(bu, bu);
  const double det = A00 * A11 - A10 * A01;
  const double invdet = 1.0 / det;
  const double i00 = A11 * invdet, i10 = -A10 * invdet, i01 = -A01 * invdet, i11 = A00 * invdet;
  const double l0 = i00 * b0 + i01 * b1, l1 = i10 * b0 + i11 * b1;
  double p[3];
  for (int c = 0; c < 3; c++) {
    const double xm = l0 * a[c];
    const double xn = M.t[c] + l1 * bu[c];
    p[c] = (xm + xn) / 2;
  }
  double r1[3], r2[3];
  const double n1 = sqrt(rs_dot3(p, p));
  for (int c = 0; c < 3; c++) r1[c] = p[c] / n1;
  for (int r = 0; r < 3; r++)
    r2[r] = ((M.inv[r * 4] * p[0] + M.inv[r * 4 + 1] * p[1]) + M.inv[r * 4 + 2] * p[2]) + M.inv[r * 4 + 3] * 1.0;
  const double n2 = sqrt(rs_dot3(r2, r2));
  for (int c = 0; c < 3; c++) r2[c] = r2[c] / n2;
  const double e1 = 1.0 - rs_dot3(a, r1);
  const double e2 = 1.0 - rs_dot3(b, r2);
  return e1 + e2;
}

struct Rs2d2dResult {
  int status, n_inliers, iterations;
  double pose[12];
};

__device__ Rs2d2dResult rs_ransac_nister(const KParams& P, const Tables& T, const double* f1, const double* f2,
                                         int n, int* shuffled, int* wave_tot, int* inliers);   // 5-point, below

// opengv::sac::Ransac<TranslationOnlySacProblem>::computeModel + the checks of Tracker::runRansac and
// Tracker::geometricOutlierRejection2d2d.  Every thread of the block calls it with the same
// arguments and gets the same result; `inliers` receives the ascending inlier indices.
// LDS: shuffled [n] int.
__device__ Rs2d2dResult rs_ransac_2d2d(const KParams& P, const Tables& T, const double* f1,
                                       const double* f2, int n, const double* R12, int* shuffled,
                                       int* wave_tot, int* inliers) {
  const int tid = threadIdx.x;
  __shared__ int sh_sel[2];
  __shared__ int sh_cnt;
  Rs2d2dResult res;
  res.status = TRK_INVALID;
  res.n_inliers = 0;
  res.iterations = 0;
  for (int i = 0; i < 12; i++) res.pose[i] = (i % 5 == 0) ? 1.0 : 0.0;  // Pose3()
  for (int i = tid; i < n; i += RS_T) shuffled[i] = i;
  __syncthreads();
  int iterations = 0, best = -INT_MAX, draw = 0;
  double k = 1.0;
  bool have_model = false;
  RsModel Mbest;
  if (n >= 2) {
    while ((double)iterations < k) {
      if (tid == 0) {  // SampleConsensusProblem::drawIndexSample (the shuffle persists across iterations)
        for (int i = 0; i < 2; ++i) {
          const int r = T.ransac_rnd[min(draw + i, T.n_ransac_rnd - 1)];
          const int j = i + (int)((unsigned)r % (unsigned)(n - i));
          const int tmp = shuffled[i];
          shuffled[i] = shuffled[j];
          shuffled[j] = tmp;
        }
        sh_sel[0] = shuffled[0];
        sh_sel[1] = shuffled[1];
      }
      draw += 2;
      __syncthreads();
      RsModel M;
      rs_twopt(f1, f2, R12, sh_sel[0], sh_sel[1], &M);
      int cnt = 0;
      for (int i = tid; i < n; i += RS_T) {
        double a[3], b[3];
        for (int c = 0; c < 3; c++) {
          a[c] = f1[3 * (size_t)i + c];
          b[c] = f2[3 * (size_t)i + c];
        }
        if (rs_distance(M, a, b) < P.ransac_thr_mono) cnt++;
      }
      cnt = rs_block_sum(cnt, wave_tot);
      if (cnt > best) {
        best = cnt;
        Mbest = M;
        have_model = true;
        const double w = (double)best / (double)n;
        double p_no_outliers = 1.0 - pow(w, 2.0);
        p_no_outliers = fmax(2.220446049250313e-16, p_no_outliers);
        p_no_outliers = fmin(1.0 - 2.220446049250313e-16, p_no_outliers);
        k = log(1.0 - P.ransac_probability) / log(p_no_outliers);
      }
      ++iterations;
      if (iterations > P.ransac_max_iters) break;
    }
  } else {
    iterations = INT_MAX;  // getSamples: not enough correspondences
  }
  res.iterations = iterations;
  if (!have_model) return res;
  // selectWithinDistance
  if (tid == 0) sh_cnt = 0;
  __syncthreads();
  for (int base = 0; base < n; base += RS_T) {
    const int i = base + tid;
    bool in = false;
    if (i < n) {
      double a[3], b[3];
      for (int c = 0; c < 3; c++) {
        a[c] = f1[3 * (size_t)i + c];
        b[c] = f2[3 * (size_t)i + c];
      }
      in = rs_distance(Mbest, a, b) < P.ransac_thr_mono;
    }
    int tot;
    const int pos = rs_scan(in ? 1 : 0, wave_tot, &tot);
    const int off = sh_cnt;
    if (in) inliers[off + pos] = i;
    __syncthreads();
    if (tid == 0) sh_cnt = off + tot;
    __syncthreads();
  }
  const int n_in = sh_cnt;
  __syncthreads();
  if (iterations >= P.ransac_max_iters && n_in == 0) return res;  // Tracker.h:270-273
  res.n_inliers = n_in;
  res.status = n_in < P.min_mono_inliers ? TRK_FEW_MATCHES : TRK_VALID;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) res.pose[r * 4 + c] = Mbest.R[r * 3 + c];
    res.pose[r * 4 + 3] = Mbest.t[r];
  }
  return res;
}

// k-th smallest (0-based) of m non-negative floats in LDS = the value std::nth_element leaves at
// position k: 4 x 8-bit radix select on the bit patterns (order preserving for non-negative floats)
__device__ float rs_kth(const float* v, int m, int kth, int* hist /*[256]*/, int* sh2 /*[2]*/) {
  const unsigned* keys = reinterpret_cast<const unsigned*>(v);
  const int tid = threadIdx.x;
  unsigned prefix = 0;
  for (int pass = 3; pass >= 0; pass--) {
    const int shift = 8 * pass;
    for (int i = tid; i < 256; i += RS_T) hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < m; i += RS_T) {
      const unsigned key = keys[i];
      if (pass == 3 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(key >> shift) & 255u], 1);
    }
    __syncthreads();
    if (tid < 64) {
      const int c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
      const int tot = c0 + c1 + c2 + c3;
      int inc = tot;
      for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off);
        if (tid >= off) inc += t;
      }
      const int before = inc - tot;
      if (kth >= before && kth < inc) {  // exactly one lane
        int r = kth - before, d;
        if (r < c0)
          d = 0;
        else if (r < c0 + c1) {
          d = 1;
          r -= c0;
        } else if (r < c0 + c1 + c2) {
          d = 2;
          r -= c0 + c1;
        } else {
          d = 3;
          r -= c0 + c1 + c2;
        }
        sh2[0] = 4 * tid + d;
        sh2[1] = r;
      }
    }
    __syncthreads();
    prefix |= (unsigned)sh2[0] << shift;
    kth = sh2[1];
    __syncthreads();
  }
  return __uint_as_float(prefix);
}

// LDS carve-up shared by the two front-end kernels
struct RsLds {
  long long* ids;  // [kcap]
  int* idx;        // [kcap]
  int* work;       // [kcap] shuffled indices / inlier flags
  float* disp;     // [kcap]
};
__host__ __device__ inline size_t rs_lds_bytes(int kcap) {
  return (size_t)kcap * (sizeof(long long) + 3 * sizeof(int)) + 16;
}
__device__ __forceinline__ RsLds rs_carve(unsigned char* raw, int kcap) {
  RsLds L;
  L.ids = reinterpret_cast<long long*>(raw);
  L.idx = reinterpret_cast<int*>(L.ids + kcap);
  L.work = L.idx + kcap;
  L.disp = reinterpret_cast<float*>(L.work + kcap);
  return L;
}

// ---------------------------------------------------------------------------------------------
// VisionImuFrontend::outlierRejectionMono for the keyframes of this step
// ---------------------------------------------------------------------------------------------
// NISTER: the 5-point problem (ransac_use_2point_mono = 0) -- its own instantiation, so that the registers and
// the scratch memory of the polynomial solver do not weigh on the 2-point kernel every shipped Euroc-style
// configuration runs
template <bool NISTER>
__global__ __launch_bounds__(RS_T) void mono_ransac_kernel(KParams P, Tables T, FrameTab K,
                                                           FrameTab LKF, StreamState S,
                                                           RansacScratch RS) {
  const int s = blockIdx.x, tid = threadIdx.x;
  const int flags = S.flags[s];
  if (P.mono && (flags & FLAG_INIT) && !(flags & FLAG_KEYFRAME)) {
    // MonoVisionImuFrontend::processFrame resets both statuses on every frame (:264-265)
    if (tid == 0) {
      S.trk_status[2 * (size_t)s] = TRK_INVALID;
      // (RgbdVisionImuFrontend::processFrame resets both to INVALID, RgbdVisionImuFrontend.cpp:269-270)
      S.trk_status[2 * (size_t)s + 1] = P.rgbd ? TRK_INVALID : TRK_DISABLED;
    }
    return;
  }
  // stereo: tracker_status_summary_ only changes on keyframes of the nominal path
  if (!(flags & FLAG_KEYFRAME) || (flags & FLAG_FIRST)) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ int wave_tot[RS_T / 64];
  __shared__ int hist[256];
  __shared__ int sh2[2];
  const RsLds L = rs_carve(lds_raw, P.kcap);
  const size_t so = (size_t)s * P.kcap;
  int* st = S.trk_status + 2 * (size_t)s;
  if (!P.use_ransac) {  // StereoVisionImuFrontend.cpp:400-404
    if (tid == 0) st[0] = st[1] = TRK_DISABLED;
    return;
  }
  if (tid == 0) {
    st[0] = TRK_INVALID;                               // :353-354
    st[1] = (P.mono && !P.rgbd) ? TRK_DISABLED : TRK_INVALID;
  }
  const double* R = S.kf_R_cur + (size_t)s * 9;
  // Tracker::geometricOutlierRejection2d2d branches on the parameter only (Tracker.cpp:247-275): the 2-point
  // problem also runs for a keyframe without gyro rotation (R = I then, VisionImuFrontend.cpp:108-111)
  int2* matches = RS.matches + so;
  const int n = rs_build_matches(P, K, LKF, nullptr, nullptr, s, S.n_tracked[s], L.ids, L.idx, wave_tot, matches);
  if (n == 0) return;
  double* f1 = RS.f_ref + so * 3;
  double* f2 = RS.f_cur + so * 3;
  for (int m = tid; m < n; m += RS_T) {
    const int2 mm = matches[m];
    for (int c = 0; c < 3; c++) {
      f1[3 * (size_t)m + c] = LKF.versor[(so + mm.x) * 3 + c];
      f2[3 * (size_t)m + c] = K.versor[(so + mm.y) * 3 + c];
    }
  }
  __syncthreads();
  double Rl[9];
  for (int i = 0; i < 9; i++) Rl[i] = R[i];
  int* inliers = RS.inliers + so;
  Rs2d2dResult res;
  if (NISTER)
    res = rs_ransac_nister(P, T, f1, f2, n, L.work, wave_tot, inliers);
  else
    res = rs_ransac_2d2d(P, T, f1, f2, n, Rl, L.work, wave_tot, inliers);
  int status = res.status;
  if (status != TRK_FEW_MATCHES) {  // removeOutliersMono: landmarks of the outliers -> -1 in frame k
    for (int m = tid; m < n; m += RS_T) L.work[m] = 0;
    __syncthreads();
    for (int q = tid; q < res.n_inliers; q += RS_T) L.work[inliers[q]] = 1;
    __syncthreads();
    for (int m = tid; m < n; m += RS_T)
      if (!L.work[m]) K.lmk[so + matches[m].y] = -1;
  }
  if (status == TRK_VALID) {  // computeMedianDisparity over the inlier matches (Tracker.cpp:361-373)
    const int m = res.n_inliers;
    for (int q = tid; q < m; q += RS_T) {
      const int2 mm = matches[inliers[q]];
      const float2 c = K.kp[so + mm.y], r = LKF.kp[so + mm.x];
      const float dx = c.x - r.x, dy = c.y - r.y;
      L.disp[q] = dx * dx + dy * dy;
    }
    __syncthreads();
    if (m > 0) {
      const double disparity = sqrt((double)rs_kth(L.disp, m, m / 2, hist, sh2));
      if (disparity < P.disparity_thr) status = TRK_LOW_DISPARITY;
    }
  }
  if (tid == 0) {
    st[0] = status;
    if (res.status != TRK_INVALID) {
      int* cn = S.trk_counts + 6 * (size_t)s;
      cn[0] = n;
      cn[1] = res.n_inliers;
      cn[2] = 0;  // monoRansacIters_ is not accessible in the reference either (Tracker.cpp:297-298)
    }
    if (status == TRK_VALID) {
      double* pose = S.trk_pose + 24 * (size_t)s;
      for (int i = 0; i < 12; i++) pose[i] = res.pose[i];
    }
  }
}

void launch_mono_ransac(const KParams& P, const Tables& T, const FrameTab& k, const FrameTab& lkf,
                        const StreamState& S, const RansacScratch& RS, hipStream_t st) {
  if (P.ransac_2pt_mono)
    hipLaunchKernelGGL(mono_ransac_kernel<false>, dim3(P.B), dim3(RS_T), rs_lds_bytes(P.kcap), st, P, T, k, lkf,
                       S, RS);
  else
    hipLaunchKernelGGL(mono_ransac_kernel<true>, dim3(P.B), dim3(RS_T), rs_lds_bytes(P.kcap), st, P, T, k, lkf,
                       S, RS);
}

// ---------------------------------------------------------------------------------------------
// 1-point voting
// ---------------------------------------------------------------------------------------------
// gtsam::StereoCamera(Pose3(), K).backproject2 Jacobian + Tracker::getPoint3AndCovariance
__device__ void rs_point3_cov(const KParams& P, double uL, double uR, double v, const double* p3,
                              const double* Rmat, double* point, double* cov) {
  const double disparity = uL - uR;
  const double local_z = P.baseline * P.fx_rect / disparity;
  const double lx = local_z * (uL - P.cx_rect) / P.fx_rect, ly = local_z * (v - P.cy_rect) / P.fy_rect;
  const double z_partial_uR = local_z / disparity;
  const double x_partial_uR = lx / disparity;
  const double y_partial_uR = ly / disparity;
  double J[9] = {-x_partial_uR + local_z / P.fx_rect, x_partial_uR, 0,
                 -y_partial_uR,                       y_partial_uR, local_z / P.fy_rect,
                 -z_partial_uR,                       z_partial_uR, 0};
  for (int i = 0; i < 3; i++) point[i] = p3[i];
  if (Rmat) {
    double q[3], RJ[9];
    rs_matvec3(Rmat, p3, q);
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++)
        RJ[r * 3 + c] = (Rmat[r * 3] * J[c] + Rmat[r * 3 + 1] * J[3 + c]) + Rmat[r * 3 + 2] * J[6 + c];
    for (int i = 0; i < 3; i++) point[i] = q[i];
    for (int i = 0; i < 9; i++) J[i] = RJ[i];
  }
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++)
      cov[r * 3 + c] = (J[r * 3] * J[c * 3] + J[r * 3 + 1] * J[c * 3 + 1]) + J[r * 3 + 2] * J[c * 3 + 2];
}

// Eigen::Matrix3d::inverse() (cofactor formula of Eigen/src/LU/InverseImpl.h)
__device__ void rs_inverse3(const double* m, double* r) {
#define M_(i, j) m[(i) * 3 + (j)]
#define COF_(i, j) \
  (M_(((i) + 1) % 3, ((j) + 1) % 3) * M_(((i) + 2) % 3, ((j) + 2) % 3) - M_(((i) + 1) % 3, ((j) + 2) % 3) * M_(((i) + 2) % 3, ((j) + 1) % 3))
  const double c00 = COF_(0, 0), c10 = COF_(1, 0), c20 = COF_(2, 0);
  const double det = (c00 * M_(0, 0) + c10 * M_(1, 0)) + c20 * M_(2, 0);
  const double invdet = 1.0 / det;
  r[3] = COF_(0, 1) * invdet;
  r[4] = COF_(1, 1) * invdet;
  r[6] = COF_(0, 2) * invdet;
  r[5] = COF_(2, 1) * invdet;
  r[7] = COF_(1, 2) * invdet;
  r[8] = COF_(2, 2) * invdet;
  r[0] = c00 * invdet;
  r[1] = c10 * invdet;
  r[2] = c20 * invdet;
#undef COF_
#undef M_
}

// the float32 Mahalanobis test of the voting loop, literally (Tracker.cpp:499-523); a = row i, b = row j
__device__ __forceinline__ float rs_mahalanobis(const float* a, const float* b) {
  float v[3], O[9];
  for (int k = 0; k < 3; k++) v[k] = a[k] - b[k];
  for (int k = 0; k < 9; k++) O[k] = a[3 + k] + b[3 + k];
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

struct Rs3d3dResult {
  int status, n_inliers;
  double t[3], info[9];
};

// The voting runs as three launches so that the O(n^2) coherence tests of all streams spread over
// the whole GPU instead of one workgroup per stream:
//   prepare : matches, float64 relative translations / covariances and their float32 copies
//   tiles   : one wavefront per 64 x 64 tile (i-block, j-block >= i-block) of the pair matrix
//   finish  : first largest coherent set, its members, information-weighted translation
// relv / relc: [n][3] / [n][9] float64, votef: [n][12] float32, cnt: [n] coherent-set sizes.
template <typename GetMatch>
__device__ void rs_vote_prepare(const KParams& P, int n, const double* R, GetMatch get, double* relv,
                                double* relc, float* votef, int* cnt) {
  for (int m = threadIdx.x; m < n; m += RS_T) {
    double rl[3], rp[3], cl[3], cp[3];  // (uL, uR, v), 3-D point of ref / cur
    get(m, rl, rp, cl, cp);
    double f_ref[3], cov_ref[9], R_f_cur[3], cov_R_cur[9];
    rs_point3_cov(P, rl[0], rl[1], rl[2], rp, nullptr, f_ref, cov_ref);
    rs_point3_cov(P, cl[0], cl[1], cl[2], cp, R, R_f_cur, cov_R_cur);
    for (int k = 0; k < 3; k++) {
      const double v = f_ref[k] - R_f_cur[k];
      relv[3 * (size_t)m + k] = v;
      votef[12 * (size_t)m + k] = (float)v;
    }
    for (int k = 0; k < 9; k++) {
      const double c = cov_R_cur[k] + cov_ref[k];
      relc[9 * (size_t)m + k] = c;
      votef[12 * (size_t)m + 3 + k] = (float)c;
    }
    cnt[m] = 1;  // a vector is coherent with itself
  }
}

// one wavefront: rows ib*64.., columns jb*64.. (jb >= ib); lane = row, the column is uniform
__device__ void rs_vote_tile(const KParams& P, int n, int tile, const float* __restrict__ votef,
                             int* cnt) {
  // tile -> (ib, jb) of the upper triangle, row-major
  const int nb = (n + 63) >> 6;
  int ib = 0, rem = tile;
  while (ib < nb && rem >= nb - ib) {
    rem -= nb - ib;
    ib++;
  }
  if (ib >= nb) return;
  const int jb = ib + rem;
  const int lane = threadIdx.x;
  const int i = ib * 64 + lane;
  const bool row = i < n;
  float a[12];
  for (int k = 0; k < 12; k++) a[k] = row ? votef[12 * (size_t)i + k] : 0.f;
  const float thr = P.ransac_thr_stereo;
  const int j0 = jb * 64, j1 = min(n, j0 + 64);
  int mine = 0;
  for (int j = j0; j < j1; j++) {
    float b[12];
    for (int k = 0; k < 12; k++) b[k] = votef[12 * (size_t)j + k];  // uniform address
    const bool coh = row && j > i && rs_mahalanobis(a, b) < thr;
    const unsigned long long m = __ballot(coh);
    if (m) {
      if (coh) mine++;
      if (lane == 0) atomicAdd(&cnt[j], __popcll(m));
    }
  }
  if (mine) atomicAdd(&cnt[i], mine);
}

__device__ Rs3d3dResult rs_vote_finish(const KParams& P, int n, const double* relv, const double* relc,
                                       const float* votef, double* acc, const int* cnt, int* wave_tot,
                                       int* inliers) {
  const int tid = threadIdx.x;
  __shared__ int sh_best, sh_cnt;
  __shared__ double sh_sum[12];
  __shared__ unsigned long long sh_key[RS_T / 64];
  Rs3d3dResult res;
  res.status = TRK_INVALID;
  res.n_inliers = 0;
  for (int i = 0; i < 3; i++) res.t[i] = 0;
  for (int i = 0; i < 9; i++) res.info[i] = 0;
  const float thr = P.ransac_thr_stereo;
  // first index with the largest coherent set
  int best_c = 0, best_i = INT_MAX;
  for (int i = tid; i < n; i += RS_T) {
    const int c = cnt[i];
    if (c > best_c) {  // ascending i per thread: strict > keeps the first
      best_c = c;
      best_i = i;
    }
  }
  unsigned long long key = ((unsigned long long)(unsigned)best_c << 32) | (unsigned)(INT_MAX - best_i);
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(key, off);
    key = o > key ? o : key;
  }
  if ((tid & 63) == 0) sh_key[tid >> 6] = key;
  __syncthreads();
  if (tid == 0) {
    unsigned long long k2 = sh_key[0];
    for (int w = 1; w < RS_T / 64; w++) k2 = sh_key[w] > k2 ? sh_key[w] : k2;
    sh_best = (int)(k2 >> 32) >= 2 ? INT_MAX - (int)(unsigned)(k2 & 0xffffffffu) : -1;
    sh_cnt = 0;
  }
  __syncthreads();
  const int best = sh_best;
  if (n == 0 || best < 0) return res;  // maxCoherentSetSize < 2
  // members of the winning coherent set, ascending (= the sorted inlier list)
  {
    float a[12];
    for (int k = 0; k < 12; k++) a[k] = votef[12 * (size_t)best + k];
    for (int base = 0; base < n; base += RS_T) {
      const int j = base + tid;
      bool in = false;
      if (j < n) {
        if (j == best) {
          in = true;
        } else {
          float b[12];
          for (int k = 0; k < 12; k++) b[k] = votef[12 * (size_t)j + k];
          in = (j > best ? rs_mahalanobis(a, b) : rs_mahalanobis(b, a)) < thr;
        }
      }
      int tot;
      const int pos = rs_scan(in ? 1 : 0, wave_tot, &tot);
      const int off = sh_cnt;
      if (in) inliers[off + pos] = j;
      __syncthreads();
      if (tid == 0) sh_cnt = off + tot;
      __syncthreads();
    }
  }
  const int n_in = sh_cnt;
  __syncthreads();
  res.n_inliers = n_in;
  res.status = n_in < P.min_stereo_inliers ? TRK_FEW_MATCHES : TRK_VALID;
  // t = (sum_i info_i)^-1 sum_i info_i v_i, summed in inlier order (Tracker.cpp:584-592): the terms
  // are computed in parallel, staged in LDS component-major, and twelve lanes walk one chain each
  constexpr int CH = 256;  // inliers per LDS chunk
  __shared__ double sh_terms[12][CH + 1];
  double sum = 0.0;
  for (int base = 0; base < n_in; base += CH) {
    const int q = base + tid;
    if (q < n_in) {
      const int id = inliers[q];
      double info[9], iv[3];
      rs_inverse3(relc + 9 * (size_t)id, info);
      rs_matvec3(info, relv + 3 * (size_t)id, iv);
      for (int k = 0; k < 3; k++) sh_terms[k][tid] = iv[k];
      for (int k = 0; k < 9; k++) sh_terms[3 + k][tid] = info[k];
    }
    __syncthreads();
    if (tid < 12) {
      const int m = min(CH, n_in - base);
      for (int q2 = 0; q2 < m; q2++) sum = sum + sh_terms[tid][q2];
    }
    __syncthreads();
  }
  if (tid < 12) sh_sum[tid] = sum;
  __syncthreads();
  double total[9], inv_total[9], tsum[3];
  for (int k = 0; k < 3; k++) tsum[k] = sh_sum[k];
  for (int k = 0; k < 9; k++) total[k] = sh_sum[3 + k];
  rs_inverse3(total, inv_total);
  rs_matvec3(inv_total, tsum, res.t);
  for (int k = 0; k < 9; k++) res.info[k] = total[k];
  __syncthreads();
  return res;
}

// ---------------------------------------------------------------------------------------------
// opengv PointCloudSacProblem: 3-point Arun (Tracker::geometricOutlierRejection3d3d, Tracker.cpp:667-742)
// ---------------------------------------------------------------------------------------------
// one-sided Jacobi SVD of a 3x3 matrix; operation for operation the oracle's svd3 (oracle/opengv_re.cpp)
__device__ void rs_svd3(const double* H, double* U, double* S, double* V) {
  double A[9];
  for (int i = 0; i < 9; i++) {
    A[i] = H[i];
    V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  }
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int r = 0; r < 3; r++) {
          alpha += A[r * 3 + p] * A[r * 3 + p];
          beta += A[r * 3 + q] * A[r * 3 + q];
          gamma += A[r * 3 + p] * A[r * 3 + q];
        }
        off = fmax(off, fabs(gamma) / sqrt(fmax(alpha * beta, 1e-300)));
        if (fabs(gamma) <= 1e-300) continue;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + tt * tt), sn = c * tt;
        for (int r = 0; r < 3; r++) {
          const double ap = A[r * 3 + p], aq = A[r * 3 + q];
          A[r * 3 + p] = c * ap - sn * aq;
          A[r * 3 + q] = sn * ap + c * aq;
          const double vp = V[r * 3 + p], vq = V[r * 3 + q];
          V[r * 3 + p] = c * vp - sn * vq;
          V[r * 3 + q] = sn * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  for (int c = 0; c < 3; c++) S[c] = sqrt(A[c] * A[c] + A[3 + c] * A[3 + c] + A[6 + c] * A[6 + c]);
  int order[3] = {0, 1, 2};
  for (int i = 1; i < 3; i++)
    for (int j = i; j > 0 && S[order[j]] > S[order[j - 1]]; j--) {
      const int t = order[j];
      order[j] = order[j - 1];
      order[j - 1] = t;
    }
  double A2[9], V2[9], S2[3];
  for (int c = 0; c < 3; c++) {
    S2[c] = S[order[c]];
    for (int r = 0; r < 3; r++) {
      A2[r * 3 + c] = A[r * 3 + order[c]];
      V2[r * 3 + c] = V[r * 3 + order[c]];
    }
  }
  for (int i = 0; i < 9; i++) V[i] = V2[i];
  for (int i = 0; i < 3; i++) S[i] = S2[i];
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) U[r * 3 + c] = S[c] > 1e-300 ? A2[r * 3 + c] / S[c] : 0.0;
  if (S[2] <= 1e-12 * fmax(S[0], 1e-300)) {
    const double u0[3] = {U[0], U[3], U[6]}, u1[3] = {U[1], U[4], U[7]};
    double u2[3];
    rs_cross3(u0, u1, u2);
    const double n = sqrt(rs_dot3(u2, u2));
    if (n > 0)
      for (int r = 0; r < 3; r++) U[r * 3 + 2] = u2[r] / n;
  }
}

// point_cloud::threept_arun over three correspondences -> model [R | t] (3x4 row-major): p1 = R p2 + t
__device__ void rs_arun3(const double* p1, const double* p2, const int* sel, double* model) {
  double c1[3] = {0, 0, 0}, c2[3] = {0, 0, 0};
  for (int k = 0; k < 3; k++)
    for (int c = 0; c < 3; c++) {
      c1[c] += p1[3 * (size_t)sel[k] + c];
      c2[c] += p2[3 * (size_t)sel[k] + c];
    }
  for (int c = 0; c < 3; c++) {
    c1[c] = c1[c] / 3.0;
    c2[c] = c2[c] / 3.0;
  }
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int k = 0; k < 3; k++) {
    double f[3], fp[3];
    for (int c = 0; c < 3; c++) {
      f[c] = p1[3 * (size_t)sel[k] + c] - c1[c];
      fp[c] = p2[3 * (size_t)sel[k] + c] - c2[c];
    }
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) H[r * 3 + c] += fp[r] * f[c];
  }
  double U[9], S[3], V[9], R[9];
  rs_svd3(H, U, S, V);
  auto mulVUt = [&](const double* Vm) {
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++)
        R[r * 3 + c] = (Vm[r * 3] * U[c * 3] + Vm[r * 3 + 1] * U[c * 3 + 1]) + Vm[r * 3 + 2] * U[c * 3 + 2];
  };
  mulVUt(V);
  const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) +
                     R[2] * (R[3] * R[7] - R[4] * R[6]);
  if (det < 0) {
    double Vp[9];
    for (int i = 0; i < 9; i++) Vp[i] = V[i];
    for (int r = 0; r < 3; r++) Vp[r * 3 + 2] = -Vp[r * 3 + 2];
    mulVUt(Vp);
  }
  double Rc2[3];
  rs_matvec3(R, c2, Rc2);
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) model[r * 4 + c] = R[r * 3 + c];
    model[r * 4 + 3] = c1[r] - Rc2[r];
  }
}
__device__ __forceinline__ double rs_arun_distance(const double* model, const double* a, const double* b) {
  const double R[9] = {model[0], model[1], model[2], model[4], model[5], model[6], model[8], model[9], model[10]};
  double q[3], d[3];
  rs_matvec3(R, b, q);
  for (int c = 0; c < 3; c++) d[c] = a[c] - (q[c] + model[c * 4 + 3]);
  return sqrt(rs_dot3(d, d));
}

// opengv::sac::Ransac<PointCloudSacProblem>::computeModel + the checks of Tracker::runRansac and
// Tracker::geometricOutlierRejection3d3d; same calling convention as rs_ransac_2d2d
__device__ Rs2d2dResult rs_ransac_arun(const KParams& P, const Tables& T, const double* p1, const double* p2,
                                       int n, int* shuffled, int* wave_tot, int* inliers) {
  const int tid = threadIdx.x;
  __shared__ int sh_sel3[3];
  __shared__ int sh_cnt3;
  __shared__ double sh_model[12];
  Rs2d2dResult res;
  res.status = TRK_INVALID;
  res.n_inliers = 0;
  res.iterations = 0;
  for (int i = 0; i < 12; i++) res.pose[i] = (i % 5 == 0) ? 1.0 : 0.0;  // Pose3()
  for (int i = tid; i < n; i += RS_T) shuffled[i] = i;
  __syncthreads();
  int iterations = 0, best = -INT_MAX, draw = 0;
  double k = 1.0;
  bool have_model = false;
  double Mbest[12];
  if (n >= 3) {
    while ((double)iterations < k) {
      if (tid == 0) {  // drawIndexSample, then threept_arun on the sample
        for (int i = 0; i < 3; ++i) {
          const int r = T.ransac_rnd[min(draw + i, T.n_ransac_rnd - 1)];
          const int j = i + (int)((unsigned)r % (unsigned)(n - i));
          const int tmp = shuffled[i];
          shuffled[i] = shuffled[j];
          shuffled[j] = tmp;
        }
        for (int i = 0; i < 3; i++) sh_sel3[i] = shuffled[i];
        double m[12];
        rs_arun3(p1, p2, sh_sel3, m);
        for (int i = 0; i < 12; i++) sh_model[i] = m[i];
      }
      draw += 3;
      __syncthreads();
      double M[12];
      for (int i = 0; i < 12; i++) M[i] = sh_model[i];
      int cnt = 0;
      for (int i = tid; i < n; i += RS_T)
        if (rs_arun_distance(M, p1 + 3 * (size_t)i, p2 + 3 * (size_t)i) < P.ransac_thr_stereo_d) cnt++;
      cnt = rs_block_sum(cnt, wave_tot);
      if (cnt > best) {
        best = cnt;
        for (int i = 0; i < 12; i++) Mbest[i] = M[i];
        have_model = true;
        const double w = (double)best / (double)n;
        double p_no_outliers = 1.0 - pow(w, 3.0);
        p_no_outliers = fmax(2.220446049250313e-16, p_no_outliers);
        p_no_outliers = fmin(1.0 - 2.220446049250313e-16, p_no_outliers);
        k = log(1.0 - P.ransac_probability) / log(p_no_outliers);
      }
      ++iterations;
      if (iterations > P.ransac_max_iters) break;
    }
  } else {
    iterations = INT_MAX;  // getSamples: not enough correspondences
  }
  res.iterations = iterations;
  if (!have_model) return res;
  if (tid == 0) sh_cnt3 = 0;
  __syncthreads();
  for (int base = 0; base < n; base += RS_T) {   // selectWithinDistance
    const int i = base + tid;
    const bool in = i < n && rs_arun_distance(Mbest, p1 + 3 * (size_t)min(i, n - 1), p2 + 3 * (size_t)min(i, n - 1)) <
                                 P.ransac_thr_stereo_d;
    int tot;
    const int pos = rs_scan(in ? 1 : 0, wave_tot, &tot);
    const int off = sh_cnt3;
    if (in) inliers[off + pos] = i;
    __syncthreads();
    if (tid == 0) sh_cnt3 = off + tot;
    __syncthreads();
  }
  const int n_in = sh_cnt3;
  __syncthreads();
  if (iterations >= P.ransac_max_iters && n_in == 0) return res;  // Tracker.h:270-273
  res.n_inliers = n_in;
  res.status = n_in < P.min_stereo_inliers ? TRK_FEW_MATCHES : TRK_VALID;
  for (int i = 0; i < 12; i++) res.pose[i] = Mbest[i];
  return res;
}

// outlierRejectionStereo, 3-point branch (VisionImuFrontend.cpp:137-142): streams whose keyframe has no
// usable gyro rotation, or every stream when ransac_use_1point_stereo is off
__global__ __launch_bounds__(RS_T) void stereo_arun_kernel(KParams P, Tables T, FrameTab K, FrameTab LKF,
                                                           StereoTab ST, StereoTab LST, StreamState S,
                                                           RansacScratch RS) {
  const int s = blockIdx.x, tid = threadIdx.x;
  const int flags = S.flags[s];
  if (!(flags & FLAG_KEYFRAME) || (flags & FLAG_FIRST) || !P.use_ransac || !P.use_stereo_tracking) return;
  const double* R = S.kf_R_cur + (size_t)s * 9;
  if (P.ransac_1pt_stereo && !rs_rot_is_identity(R)) return;   // handled by the 1-point voting
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ int wave_tot[RS_T / 64];
  const RsLds L = rs_carve(lds_raw, P.kcap);
  const size_t so = (size_t)s * P.kcap;
  int2* matches = RS.matches + so;
  const int n = rs_build_matches(P, K, LKF, ST.right_status, LST.right_status, s, S.n_tracked[s], L.ids, L.idx,
                                 wave_tot, matches);
  double* p1 = RS.f_ref + so * 3;
  double* p2 = RS.f_cur + so * 3;
  for (int m = tid; m < n; m += RS_T) {
    const int2 mm = matches[m];
    for (int c = 0; c < 3; c++) {
      p1[3 * (size_t)m + c] = LST.kp3d[(so + mm.x) * 3 + c];
      p2[3 * (size_t)m + c] = ST.kp3d[(so + mm.y) * 3 + c];
    }
  }
  __syncthreads();
  const Rs2d2dResult res = rs_ransac_arun(P, T, p1, p2, n, L.work, wave_tot, RS.inliers + so);
  if (tid == 0) {
    S.trk_status[2 * (size_t)s + 1] = res.status;
    if (res.status != TRK_INVALID) {
      int* cn = S.trk_counts + 6 * (size_t)s;
      cn[3] = n;
      cn[4] = res.n_inliers;
    }
    if (res.status == TRK_VALID) {
      double* pose = S.trk_pose + 24 * (size_t)s + 12;
      for (int i = 0; i < 12; i++) pose[i] = res.pose[i];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// opengv CentralRelativePoseSacProblem(NISTER): the 5-point problem of ransac_use_2point_mono = 0
// (Tracker.cpp:262-275).  The solver is, operation for operation, the oracle's (oracle/opengv_re.cpp: Jacobi null
// space, the ten cubic constraints, Gauss-Jordan in Nister's monomial order, det B(z) of degree 10, Sturm
// bracketing + bisection); the four decompositions of every essential matrix are scored on the 5 + 3 sample
// points like CentralRelativePoseSacProblem::computeModelCoefficients.  One thread solves one hypothesis; a
// batch of hypotheses is solved in parallel and replayed in order (the sample stream does not depend on results).
// ---------------------------------------------------------------------------------------------
// per-hypothesis work area in LDS (dynamic indexing of private arrays would go to scratch memory)
struct NpSlot {
  double Q[5][9];
  double N[4][9];
  double M[10][20];
  double chain[12][12];   // Sturm chain, coefficient k of z^k
  double leaf_a[10], leaf_b[10];   // isolating intervals in the order the interval stack yields them
  double roots[10];
  double qual[10];        // best summed reprojection error over the decompositions of root k
  int cdeg[12];
  int leaf_cnt[10];
  int nleaf, ok;
};
// linear x linear -> quadratic [x2 y2 z2 xy xz yz x y z 1]
__device__ __forceinline__ void np_lin_mul(const double* a, const double* b, double* q) {
  q[0] = a[0] * b[0];
  q[1] = a[1] * b[1];
  q[2] = a[2] * b[2];
  q[3] = a[0] * b[1] + a[1] * b[0];
  q[4] = a[0] * b[2] + a[2] * b[0];
  q[5] = a[1] * b[2] + a[2] * b[1];
  q[6] = a[0] * b[3] + a[3] * b[0];
  q[7] = a[1] * b[3] + a[3] * b[1];
  q[8] = a[2] * b[3] + a[3] * b[2];
  q[9] = a[3] * b[3];
}
// quadratic x linear -> cubic in Nister's monomial order, out += sign * product
__device__ __forceinline__ void np_quadlin_acc(const double* q, const double* l, double sign, double* out) {
  out[0] += sign * (q[0] * l[0]);
  out[1] += sign * (q[1] * l[1]);
  out[2] += sign * (q[0] * l[1] + q[3] * l[0]);
  out[3] += sign * (q[1] * l[0] + q[3] * l[1]);
  out[4] += sign * (q[0] * l[2] + q[4] * l[0]);
  out[5] += sign * (q[0] * l[3] + q[6] * l[0]);
  out[6] += sign * (q[1] * l[2] + q[5] * l[1]);
  out[7] += sign * (q[1] * l[3] + q[7] * l[1]);
  out[8] += sign * ((q[3] * l[2] + q[4] * l[1]) + q[5] * l[0]);
  out[9] += sign * ((q[3] * l[3] + q[6] * l[1]) + q[7] * l[0]);
  out[10] += sign * (q[2] * l[0] + q[4] * l[2]);
  out[11] += sign * ((q[4] * l[3] + q[6] * l[2]) + q[8] * l[0]);
  out[12] += sign * (q[6] * l[3] + q[9] * l[0]);
  out[13] += sign * (q[2] * l[1] + q[5] * l[2]);
  out[14] += sign * ((q[5] * l[3] + q[7] * l[2]) + q[8] * l[1]);
  out[15] += sign * (q[7] * l[3] + q[9] * l[1]);
  out[16] += sign * (q[2] * l[2]);
  out[17] += sign * (q[2] * l[3] + q[8] * l[2]);
  out[18] += sign * (q[8] * l[3] + q[9] * l[2]);
  out[19] += sign * (q[9] * l[3]);
}
// four orthonormal null vectors of W.Q (5 x 9): Gauss-Jordan with complete pivoting, Gram-Schmidt
__device__ bool np_nullspace(NpSlot& W) {
  int pc[5];
  unsigned used = 0;
  for (int s = 0; s < 5; s++) {
    int br = s, bc = -1;
    double bv = -1.0;
    for (int r = s; r < 5; r++)
      for (int c = 0; c < 9; c++) {
        if (used & (1u << c)) continue;
        const double v = fabs(W.Q[r][c]);
        if (v > bv) {
          bv = v;
          br = r;
          bc = c;
        }
      }
    if (!(bv > 1e-300)) return false;
    if (br != s)
      for (int c = 0; c < 9; c++) {
        const double t = W.Q[br][c];
        W.Q[br][c] = W.Q[s][c];
        W.Q[s][c] = t;
      }
    used |= 1u << bc;
    pc[s] = bc;
    const double inv = 1.0 / W.Q[s][bc];
    for (int c = 0; c < 9; c++) W.Q[s][c] *= inv;
    for (int r = 0; r < 5; r++) {
      if (r == s) continue;
      const double f = W.Q[r][bc];
      if (f == 0.0) continue;
      for (int c = 0; c < 9; c++) W.Q[r][c] -= f * W.Q[s][c];
    }
  }
  int j = 0;
  for (int fc = 0; fc < 9; fc++) {
    if (used & (1u << fc)) continue;
    for (int c = 0; c < 9; c++) W.N[j][c] = 0.0;
    W.N[j][fc] = 1.0;
    for (int s = 0; s < 5; s++) W.N[j][pc[s]] = -W.Q[s][fc];
    j++;
  }
  for (int a = 0; a < 4; a++) {
    for (int b = 0; b < a; b++) {
      double d = 0;
      for (int c = 0; c < 9; c++) d += W.N[a][c] * W.N[b][c];
      for (int c = 0; c < 9; c++) W.N[a][c] -= d * W.N[b][c];
    }
    double nn = 0;
    for (int c = 0; c < 9; c++) nn += W.N[a][c] * W.N[a][c];
    nn = sqrt(nn);
    if (!(nn > 1e-300)) return false;
    for (int c = 0; c < 9; c++) W.N[a][c] = W.N[a][c] / nn;
  }
  return true;
}
// r[i + j] += a[i] * b[j], i ascending then j ascending (Poly1 multiplication of the oracle), fixed degrees
template <int DA, int DB>
__device__ __forceinline__ void np_conv(const double* a, const double* b, double* r) {
#pragma unroll
  for (int i = 0; i <= DA + DB; i++) r[i] = 0.0;
#pragma unroll
  for (int i = 0; i <= DA; i++)
#pragma unroll
    for (int j = 0; j <= DB; j++) r[i + j] += a[i] * b[j];
}
// Horner on a register copy of one chain polynomial: the steps above its degree are skipped by selects, so the
// operations are those of the plain loop from `deg` down to 0 without an LDS round trip inside the dependent chain
__device__ __forceinline__ double np_horner11(const double (&c)[11], int deg, double z) {
  double v = 0;
#pragma unroll
  for (int i = 10; i >= 0; i--) {
    const double t = v * z + c[i];
    v = i <= deg ? t : v;
  }
  return v;
}
// sign changes of the Sturm chain at z: all polynomials advance together (chain k has degree <= 10 - k), which
// gives the scheduler nc independent Horner chains instead of one LDS-latency-bound chain after the other
__device__ int np_sturm_changes(const NpSlot& W, const int (&degs)[11], double z) {
  double v[11];
#pragma unroll
  for (int k = 0; k < 11; k++) v[k] = 0;
#pragma unroll
  for (int i = 10; i >= 0; i--) {
#pragma unroll
    for (int k = 0; k <= 10 - i; k++) {
      const double t = v[k] * z + W.chain[k][i];
      v[k] = i <= degs[k] ? t : v[k];
    }
  }
  int n = 0, prev = 0;
#pragma unroll
  for (int k = 0; k < 11; k++) {
    const int sgn = v[k] > 0 ? 1 : v[k] < 0 ? -1 : 0;
    if (sgn != 0) {
      if (prev != 0 && sgn != prev) n++;
      prev = sgn;
    }
  }
  return n;
}
// Real roots of the polynomial in W.chain[0] (degree W.cdeg[0]), step 1: Sturm chain + interval subdivision.
// Every interval the stack loop of the oracle would refine is recorded (in that order) instead of refined on the
// spot: the bisections are independent of the subdivision and of each other, so they run one per lane afterwards.
__device__ void np_isolate(NpSlot& W) {
  W.nleaf = 0;
  {
    int deg = W.cdeg[0];
    double m = 0;
    for (int i = 0; i <= deg; i++) m = fmax(m, fabs(W.chain[0][i]));
    while (deg > 0 && fabs(W.chain[0][deg]) <= 1e-14 * m) deg--;
    W.cdeg[0] = deg;
  }
  const int pdeg = W.cdeg[0];
  if (pdeg < 1) return;
  int nc = 1;
  for (int i = 1; i <= pdeg; i++) W.chain[1][i - 1] = i * W.chain[0][i];
  W.cdeg[1] = pdeg - 1;
  nc = 2;
  while (W.cdeg[nc - 1] > 0 && nc < 12) {
    const int da = W.cdeg[nc - 2], db = W.cdeg[nc - 1];
    double* a = W.chain[nc];   // remainder built in place in the next slot
    for (int i = 0; i <= da; i++) a[i] = W.chain[nc - 2][i];
    const double* b = W.chain[nc - 1];
    for (int i = da; i >= db; i--) {
      const double f = a[i] / b[db];
      for (int j = 0; j <= db; j++) a[i - db + j] -= f * b[j];
      a[i] = 0.0;
    }
    int deg = max(db - 1, 0);
    double m = 0;
    for (int i = 0; i <= deg; i++) m = fmax(m, fabs(a[i]));
    if (m == 0) break;
    for (int i = 0; i <= deg; i++) a[i] = -a[i] / m;
    while (deg > 0 && fabs(a[deg]) <= 1e-13) deg--;
    W.cdeg[nc] = deg;
    nc++;
  }
  // chain k of the nc built has degree cdeg[k] <= 10 - k; the others never contribute (degree -1 -> value 0)
  int degs[11];
#pragma unroll
  for (int kk = 0; kk < 11; kk++) degs[kk] = kk < nc ? W.cdeg[kk] : -1;
  double bound = 0;
  for (int i = 0; i < pdeg; i++) bound = fmax(bound, fabs(W.chain[0][i] / W.chain[0][pdeg]));
  bound += 1.0;
  int nroots = 0;
  // interval stack (depth <= 64) in the constraint-matrix area of the slot, which is dead by now
  double* sa = &W.M[0][0];
  double* sb = sa + 64;
  int* sna = reinterpret_cast<int*>(sb + 64);
  int* snb = sna + 64;
  sa[0] = -bound;
  sb[0] = bound;
  sna[0] = np_sturm_changes(W, degs, -bound);
  snb[0] = np_sturm_changes(W, degs, bound);
  int sp = 1;
  while (sp > 0 && nroots < 10) {
    --sp;
    const double ia = sa[sp], ib = sb[sp];
    const int na = sna[sp], nb = snb[sp];
    const int cnt = na - nb;
    if (cnt <= 0) continue;
    const double mid = 0.5 * (ia + ib);
    if (cnt == 1 || ib - ia < 1e-13 * fmax(1.0, fabs(mid))) {
      W.leaf_a[nroots] = ia;
      W.leaf_b[nroots] = ib;
      W.leaf_cnt[nroots] = cnt;
      nroots++;
      continue;
    }
    const int nm = np_sturm_changes(W, degs, mid);
    if (sp + 2 <= 64) {
      sa[sp] = mid;
      sb[sp] = ib;
      sna[sp] = nm;
      snb[sp] = nb;
      sp++;
      sa[sp] = ia;
      sb[sp] = mid;
      sna[sp] = na;
      snb[sp] = nm;
      sp++;
    }
  }
  W.nleaf = nroots;
}
// step 2: bisection of isolating interval r (a cluster the chain cannot split further yields its midpoint)
__device__ double np_bisect_leaf(const NpSlot& W, int r) {
  double c0[11];
#pragma unroll
  for (int i = 0; i < 11; i++) c0[i] = W.chain[0][i];
  const int pdeg = W.cdeg[0], cnt = W.leaf_cnt[r];
  double a = W.leaf_a[r], b = W.leaf_b[r];
  double fa = np_horner11(c0, pdeg, a);
  for (int it = 0; it < 200 && b - a > 1e-16 * fmax(1.0, fabs(a) + fabs(b)); it++) {
    const double m = 0.5 * (a + b);
    if (m <= a || m >= b) break;
    const double fm = np_horner11(c0, pdeg, m);
    if (cnt == 1 && ((fa < 0) != (fm < 0))) {
      b = m;
    } else if (cnt == 1) {
      a = m;
      fa = fm;
    } else
      break;
  }
  return 0.5 * (a + b);
}
// step 3: ascending order (std::sort in the oracle; the values are distinct)
__device__ void np_sort_roots(NpSlot& W) {
  const int n = W.nleaf;
  for (int i = 1; i < n; i++)
    for (int j = i; j > 0 && W.roots[j] < W.roots[j - 1]; j--) {
      const double t = W.roots[j];
      W.roots[j] = W.roots[j - 1];
      W.roots[j - 1] = t;
    }
}
__device__ void np_set_model(const double* R, const double* t, RsModel* M) {
  for (int i = 0; i < 9; i++) M->R[i] = R[i];
  for (int i = 0; i < 3; i++) M->t[i] = t[i];
  double Rt[9], it[3];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) Rt[r * 3 + c] = R[c * 3 + r];
  rs_matvec3(Rt, t, it);
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) M->inv[r * 4 + c] = Rt[r * 3 + c];
    M->inv[r * 4 + 3] = -it[r];
  }
}
// the four [R | t] of an essential matrix (CentralRelativePoseSacProblem.cpp, NISTER case) from its SVD; j
// selects one
__device__ void np_decompose(const double* U, const double* S, const double* V, int j, RsModel* M) {
  const double W[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1}, Wt[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
  const double* Wm = (j & 1) ? Wt : W;
  double UW[9], R[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) UW[r * 3 + c] = (U[r * 3] * Wm[c] + U[r * 3 + 1] * Wm[3 + c]) + U[r * 3 + 2] * Wm[6 + c];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++)
      R[r * 3 + c] = (UW[r * 3] * V[c * 3] + UW[r * 3 + 1] * V[c * 3 + 1]) + UW[r * 3 + 2] * V[c * 3 + 2];
  const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) +
                     R[2] * (R[3] * R[7] - R[4] * R[6]);
  if (det < 0)
    for (int i = 0; i < 9; i++) R[i] = -R[i];
  const double scale = S[0], sg = (j & 2) ? -1.0 : 1.0;
  const double t[3] = {sg * (scale * U[2]), sg * (scale * U[5]), sg * (scale * U[8])};
  np_set_model(R, t, M);
}
// CentralRelativePoseSacProblem::computeModelCoefficients(NISTER) for one sample of 5 + 3 indices:
// relative_pose::fivept_nister on the first five, then every decomposition of every essential matrix scored on
// all eight (the lowest summed reprojection error wins, first one on ties).  The block solves NP_BATCH samples
// at a time in stages that change the work distribution but not one operation of the sequential algorithm:
//   A  one lane per sample     null space of the epipolar constraints, the 10 x 20 constraint matrix
//   B  16 lanes per sample     Gauss-Jordan elimination, two columns per lane
//   C  one lane per sample     B(z), its determinant, Sturm chain, root isolation
//   D  one lane per root       bisection
//   E  one lane per root       essential matrix, SVD, the four decompositions scored on the eight points
//   F  one lane per root       the winner (lowest error, first on ties) publishes its model
// stage A
__device__ bool np_stage_a(const double* f1, const double* f2, const int* s8, NpSlot& W) {
  for (int i = 0; i < 5; i++) {
    const double* f = f1 + 3 * (size_t)s8[i];
    const double* fp = f2 + 3 * (size_t)s8[i];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) W.Q[i][3 * r + c] = f[c] * fp[r];
  }
  if (!np_nullspace(W)) return false;
  {
    double El[3][3][4];   // entry (r, c) of E = x A + y B + z C + D as a linear polynomial
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
        for (int k = 0; k < 4; k++) El[r][c][k] = W.N[k][3 * r + c];
    double row[20];
    {   // det(E)
      double m0[10], m1[10], m2[10], ta[10], tb[10];
      np_lin_mul(El[1][1], El[2][2], ta);
      np_lin_mul(El[1][2], El[2][1], tb);
#pragma unroll
      for (int i = 0; i < 10; i++) m0[i] = ta[i] - tb[i];
      np_lin_mul(El[1][0], El[2][2], ta);
      np_lin_mul(El[1][2], El[2][0], tb);
#pragma unroll
      for (int i = 0; i < 10; i++) m1[i] = ta[i] - tb[i];
      np_lin_mul(El[1][0], El[2][1], ta);
      np_lin_mul(El[1][1], El[2][0], tb);
#pragma unroll
      for (int i = 0; i < 10; i++) m2[i] = ta[i] - tb[i];
#pragma unroll
      for (int i = 0; i < 20; i++) row[i] = 0.0;
      np_quadlin_acc(m0, El[0][0], 1.0, row);
      np_quadlin_acc(m1, El[0][1], -1.0, row);
      np_quadlin_acc(m2, El[0][2], 1.0, row);
#pragma unroll
      for (int i = 0; i < 20; i++) W.M[0][i] = row[i];
    }
    // 2 E E^T E - trace(E E^T) E, row by row of E E^T (recomputed: same operations, same values)
    auto eet = [&](int r, int c, double* q) {
      double t[10];
      np_lin_mul(El[r][0], El[c][0], q);
      np_lin_mul(El[r][1], El[c][1], t);
#pragma unroll
      for (int i = 0; i < 10; i++) q[i] += t[i];
      np_lin_mul(El[r][2], El[c][2], t);
#pragma unroll
      for (int i = 0; i < 10; i++) q[i] += t[i];
    };
    double tr[10];
    {
      double d0[10], d1[10], d2[10];
      eet(0, 0, d0);
      eet(1, 1, d1);
      eet(2, 2, d2);
#pragma unroll
      for (int i = 0; i < 10; i++) tr[i] = (d0[i] + d1[i]) + d2[i];
    }
#pragma unroll
    for (int r = 0; r < 3; r++) {
      double e0[10], e1[10], e2[10];
      eet(r, 0, e0);
      eet(r, 1, e1);
      eet(r, 2, e2);
#pragma unroll
      for (int c = 0; c < 3; c++) {
#pragma unroll
        for (int i = 0; i < 20; i++) row[i] = 0.0;
        np_quadlin_acc(e0, El[0][c], 2.0, row);
        np_quadlin_acc(e1, El[1][c], 2.0, row);
        np_quadlin_acc(e2, El[2][c], 2.0, row);
        np_quadlin_acc(tr, El[r][c], -1.0, row);
#pragma unroll
        for (int i = 0; i < 20; i++) W.M[1 + 3 * r + c][i] = row[i];
      }
    }
  }
  return true;
}
// stage B: Gauss-Jordan with partial pivoting on the first ten columns; lane l of the 16 of a sample owns columns
// l and l + 16.  Every element sees the operations of the sequential elimination in the same order; the barriers
// separate the reads of a pivot column from the updates of its owner.  Called by all threads of the block.
__device__ void np_stage_b(NpSlot& W, int l) {
  bool ok = W.ok != 0;
  const int c0 = l, c1 = l + 16;
  const bool has1 = c1 < 20;
  for (int col = 0; col < 10; col++) {
    int piv = col;
    if (ok) {
      for (int r = col + 1; r < 10; r++)
        if (fabs(W.M[r][col]) > fabs(W.M[piv][col])) piv = r;
      if (fabs(W.M[piv][col]) < 1e-300) ok = false;
    }
    __syncthreads();
    if (ok && piv != col) {
      double t = W.M[piv][c0];
      W.M[piv][c0] = W.M[col][c0];
      W.M[col][c0] = t;
      if (has1) {
        t = W.M[piv][c1];
        W.M[piv][c1] = W.M[col][c1];
        W.M[col][c1] = t;
      }
    }
    __syncthreads();
    double inv = 0;
    if (ok) inv = 1.0 / W.M[col][col];
    __syncthreads();
    if (ok) {
      W.M[col][c0] *= inv;
      if (has1) W.M[col][c1] *= inv;
    }
    __syncthreads();
    double f[10];
#pragma unroll
    for (int r = 0; r < 10; r++) f[r] = ok ? W.M[r][col] : 0.0;
    __syncthreads();
    if (ok) {
      const double p0 = W.M[col][c0], p1 = has1 ? W.M[col][c1] : 0.0;
#pragma unroll
      for (int r = 0; r < 10; r++) {
        if (r == col || f[r] == 0.0) continue;
        W.M[r][c0] -= f[r] * p0;
        if (has1) W.M[r][c1] -= f[r] * p1;
      }
    }
    __syncthreads();
  }
  if (l == 0 && !ok) W.ok = 0;
  __syncthreads();
}
// stage C
__device__ void np_stage_c(NpSlot& W) {
  // rows <k> = <e> - z<f>, <l> = <g> - z<h>, <m> = <i> - z<j>: B(z) [x y 1]^T = 0
  double Bx[3][4], By[3][4], Bc[3][5];
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const double* e = W.M[4 + 2 * q] + 10;
    const double* f = W.M[5 + 2 * q] + 10;
    Bx[q][0] = e[2];
    Bx[q][1] = e[1] - f[2];
    Bx[q][2] = e[0] - f[1];
    Bx[q][3] = -f[0];
    By[q][0] = e[5];
    By[q][1] = e[4] - f[5];
    By[q][2] = e[3] - f[4];
    By[q][3] = -f[3];
    Bc[q][0] = e[9];
    Bc[q][1] = e[8] - f[9];
    Bc[q][2] = e[7] - f[8];
    Bc[q][3] = e[6] - f[7];
    Bc[q][4] = -f[6];
  }
  {   // det B(z) = B00 (B11 B22 - B12 B21) - B01 (B10 B22 - B12 B20) + B02 (B10 B21 - B11 B20)
    double t1[8], t2[8], d7[8], T0[11], T1[11], T2[11], x6a[7], x6b[7], d6[7];
    np_conv<3, 4>(By[1], Bc[2], t1);
    np_conv<4, 3>(Bc[1], By[2], t2);
#pragma unroll
    for (int i = 0; i < 8; i++) d7[i] = (0.0 + t1[i]) - t2[i];
    np_conv<3, 7>(Bx[0], d7, T0);
    np_conv<3, 4>(Bx[1], Bc[2], t1);
    np_conv<4, 3>(Bc[1], Bx[2], t2);
#pragma unroll
    for (int i = 0; i < 8; i++) d7[i] = (0.0 + t1[i]) - t2[i];
    np_conv<3, 7>(By[0], d7, T1);
    np_conv<3, 3>(By[1], Bx[2], x6a);
    np_conv<3, 3>(Bx[1], By[2], x6b);
#pragma unroll
    for (int i = 0; i < 7; i++) d6[i] = (0.0 + x6a[i]) - x6b[i];
    np_conv<4, 6>(Bc[0], d6, T2);
#pragma unroll
    for (int i = 0; i < 11; i++) {
      const double inner = (0.0 + T0[i]) - T1[i];
      const double t2one = 0.0 + T2[i] * 1.0;
      W.chain[0][i] = (0.0 + inner) - t2one;
    }
    W.cdeg[0] = 10;
  }
  // B(z) stays in the (dead) epipolar-constraint area for stage E
  double* Bs = &W.Q[0][0];
#pragma unroll
  for (int q = 0; q < 3; q++) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      Bs[13 * q + i] = Bx[q][i];
      Bs[13 * q + 4 + i] = By[q][i];
    }
#pragma unroll
    for (int i = 0; i < 5; i++) Bs[13 * q + 8 + i] = Bc[q][i];
  }
  np_isolate(W);
}
// stage E: root k of the sample -> best of its four decompositions (returns its summed error, 1e6 when none)
__device__ double np_stage_e(const double* f1, const double* f2, const int* s8, const NpSlot& W, int k, RsModel* out) {
  const double z = W.roots[k];
  const double* Bs = &W.Q[0][0];
  double b[3][3];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    double v = 0;
#pragma unroll
    for (int i = 3; i >= 0; i--) v = v * z + Bs[13 * r + i];
    b[r][0] = v;
    v = 0;
#pragma unroll
    for (int i = 3; i >= 0; i--) v = v * z + Bs[13 * r + 4 + i];
    b[r][1] = v;
    v = 0;
#pragma unroll
    for (int i = 4; i >= 0; i--) v = v * z + Bs[13 * r + 8 + i];
    b[r][2] = v;
  }
  double bestdet = 0;
  int r0 = 0, r1 = 1;
#pragma unroll
  for (int pi = 0; pi < 3; pi++) {
    const int pa = pi == 2 ? 1 : 0, pb = pi == 0 ? 1 : 2;
    const double dd = b[pa][0] * b[pb][1] - b[pa][1] * b[pb][0];
    if (fabs(dd) > fabs(bestdet)) {
      bestdet = dd;
      r0 = pa;
      r1 = pb;
    }
  }
  double bestQuality = 1000000.0;
  if (bestdet == 0) return bestQuality;
  const double b00 = r0 == 0 ? b[0][0] : b[1][0], b01 = r0 == 0 ? b[0][1] : b[1][1], b02 = r0 == 0 ? b[0][2] : b[1][2];
  const double b10 = r1 == 1 ? b[1][0] : b[2][0], b11 = r1 == 1 ? b[1][1] : b[2][1], b12 = r1 == 1 ? b[1][2] : b[2][2];
  const double x = (-b02 * b11 + b12 * b01) / bestdet;
  const double y = (-b00 * b12 + b10 * b02) / bestdet;
  double E[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int a = i + 3 * j;
      E[3 * i + j] = ((x * W.N[0][a] + y * W.N[1][a]) + z * W.N[2][a]) + W.N[3][a];
    }
  double U[9], S[3], V[9];
  rs_svd3(E, U, S, V);
  for (int j = 0; j < 4; j++) {
    RsModel M;
    np_decompose(U, S, V, j, &M);
    double quality = 0.0;
    for (int q = 0; q < 8; q++) {
      double a[3], bb[3];
      for (int c = 0; c < 3; c++) {
        a[c] = f1[3 * (size_t)s8[q] + c];
        bb[c] = f2[3 * (size_t)s8[q] + c];
      }
      quality += rs_distance(M, a, bb);
    }
    if (quality < bestQuality) {
      bestQuality = quality;
      *out = M;
    }
  }
  return bestQuality;
}

constexpr int NP_BATCH = 16;   // hypotheses solved in parallel per round (one LDS work area each)
// opengv::sac::Ransac<CentralRelativePoseSacProblem>::computeModel + the checks of Tracker::runRansac and
// Tracker::geometricOutlierRejection2d2d; calling convention of rs_ransac_2d2d
__device__ Rs2d2dResult rs_ransac_nister(const KParams& P, const Tables& T, const double* f1, const double* f2,
                                         int n, int* shuffled, int* wave_tot, int* inliers) {
  const int tid = threadIdx.x;
  __shared__ int sh_sel8[NP_BATCH][8];
  __shared__ int sh_okm[NP_BATCH];
  __shared__ int sh_cntm[NP_BATCH];
  __shared__ RsModel sh_models[NP_BATCH];
  __shared__ NpSlot sh_slots[NP_BATCH];
  __shared__ int sh_state[4];   // 0: stop flag, 1: best hypothesis slot of this round (-1 none), 2: n inliers, 3: spare
  __shared__ RsModel sh_best;
  Rs2d2dResult res;
  res.status = TRK_INVALID;
  res.n_inliers = 0;
  res.iterations = 0;
  for (int i = 0; i < 12; i++) res.pose[i] = (i % 5 == 0) ? 1.0 : 0.0;  // Pose3()
  for (int i = tid; i < n; i += RS_T) shuffled[i] = i;
  __syncthreads();
  // replay state (identical in every thread: all read the same shared results)
  int iterations = 0, best = -INT_MAX, draw = 0;
  unsigned skipped = 0;
  const unsigned max_skip = (unsigned)P.ransac_max_iters * 10u;
  double k = 1.0;
  bool have_model = false, done = false;
  if (n < 8) {
    iterations = INT_MAX;  // getSamples: not enough correspondences
    done = true;
  }
  while (!done) {
    if (tid == 0) {   // drawIndexSample for the next NP_BATCH hypotheses (the shuffle persists)
      for (int h = 0; h < NP_BATCH; h++) {
        for (int i = 0; i < 8; ++i) {
          const int r = T.ransac_rnd[min(draw + 8 * h + i, T.n_ransac_rnd - 1)];
          const int j = i + (int)((unsigned)r % (unsigned)(n - i));
          const int tmp = shuffled[i];
          shuffled[i] = shuffled[j];
          shuffled[j] = tmp;
        }
        for (int i = 0; i < 8; i++) sh_sel8[h][i] = shuffled[i];
      }
    }
    draw += 8 * NP_BATCH;
    __syncthreads();
    {
      static_assert(NP_BATCH * 16 == RS_T, "16 lanes per sample");
      const int h = tid >> 4, l = tid & 15;
      NpSlot& W = sh_slots[h];
      if (l == 0) {
        W.ok = np_stage_a(f1, f2, sh_sel8[h], W) ? 1 : 0;
        W.nleaf = 0;
        sh_okm[h] = 0;
        sh_cntm[h] = 0;
      }
      __syncthreads();
      np_stage_b(W, l);
      if (l == 0 && W.ok) np_stage_c(W);
      __syncthreads();
      const int nr = W.nleaf;
      if (l < nr) W.roots[l] = np_bisect_leaf(W, l);   // stage D
      __syncthreads();
      if (l == 0) np_sort_roots(W);
      __syncthreads();
      RsModel M;
      double q = 1000000.0;
      if (l < nr) q = np_stage_e(f1, f2, sh_sel8[h], W, l, &M);
      if (l < 10) W.qual[l] = q;
      __syncthreads();
      if (l < nr && q < 1000000.0) {   // stage F
        bool win = true;
        for (int k = 0; k < nr; k++) {
          const double qk = W.qual[k];
          if (k < l ? qk <= q : k > l ? qk < q : false) win = false;
        }
        if (win) {
          sh_models[h] = M;
          sh_okm[h] = 1;
        }
      }
    }
    __syncthreads();
    // countWithinDistance of every hypothesis of the round: (hypothesis, match) pairs over the block
    for (int h = 0; h < NP_BATCH; h++) {
      if (!sh_okm[h]) continue;
      const RsModel M = sh_models[h];
      int cnt = 0;
      for (int i = tid; i < n; i += RS_T) {
        double a[3], b[3];
        for (int c = 0; c < 3; c++) {
          a[c] = f1[3 * (size_t)i + c];
          b[c] = f2[3 * (size_t)i + c];
        }
        if (rs_distance(M, a, b) < P.ransac_thr_mono) cnt++;
      }
      cnt = rs_block_sum(cnt, wave_tot);
      if (tid == 0) sh_cntm[h] = cnt;
    }
    __syncthreads();
    // replay the sequential loop of Ransac::computeModel over the round
    for (int h = 0; h < NP_BATCH && !done; h++) {
      if (!((double)iterations < k && skipped < max_skip)) {
        done = true;
        break;
      }
      if (!sh_okm[h]) {
        ++skipped;
        continue;
      }
      const int cnt = sh_cntm[h];
      if (cnt > best) {
        best = cnt;
        have_model = true;
        if (tid == 0) sh_best = sh_models[h];
        const double w = (double)best / (double)n;
        double p_no_outliers = 1.0 - pow(w, 8.0);
        p_no_outliers = fmax(2.220446049250313e-16, p_no_outliers);
        p_no_outliers = fmin(1.0 - 2.220446049250313e-16, p_no_outliers);
        k = log(1.0 - P.ransac_probability) / log(p_no_outliers);
      }
      ++iterations;
      if (iterations > P.ransac_max_iters) done = true;
    }
    if (!((double)iterations < k && skipped < max_skip)) done = true;
    __syncthreads();
  }
  res.iterations = iterations;
  if (!have_model) return res;
  __syncthreads();
  const RsModel Mbest = sh_best;
  if (tid == 0) sh_state[2] = 0;
  __syncthreads();
  for (int base = 0; base < n; base += RS_T) {   // selectWithinDistance
    const int i = base + tid;
    bool in = false;
    if (i < n) {
      double a[3], b[3];
      for (int c = 0; c < 3; c++) {
        a[c] = f1[3 * (size_t)i + c];
        b[c] = f2[3 * (size_t)i + c];
      }
      in = rs_distance(Mbest, a, b) < P.ransac_thr_mono;
    }
    int tot;
    const int pos = rs_scan(in ? 1 : 0, wave_tot, &tot);
    const int off = sh_state[2];
    if (in) inliers[off + pos] = i;
    __syncthreads();
    if (tid == 0) sh_state[2] = off + tot;
    __syncthreads();
  }
  const int n_in = sh_state[2];
  __syncthreads();
  if (iterations >= P.ransac_max_iters && n_in == 0) return res;  // Tracker.h:270-273
  res.n_inliers = n_in;
  res.status = n_in < P.min_mono_inliers ? TRK_FEW_MATCHES : TRK_VALID;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) res.pose[r * 4 + c] = Mbest.R[r * 3 + c];
    res.pose[r * 4 + 3] = Mbest.t[r];
  }
  return res;
}

// ---- front-end: the three launches over all streams ------------------------------------------------
__global__ __launch_bounds__(RS_T) void stereo_ransac_prepare_kernel(KParams P, Tables T, FrameTab K,
                                                                     FrameTab LKF, StereoTab ST,
                                                                     StereoTab LST, StreamState S,
                                                                     RansacScratch RS) {
  const int s = blockIdx.x, tid = threadIdx.x;
  const int flags = S.flags[s];
  if (tid == 0) RS.n_matches[s] = -1;
  if (!(flags & FLAG_KEYFRAME) || (flags & FLAG_FIRST) || !P.use_ransac) return;
  if (!P.use_stereo_tracking) return;  // status stays INVALID (StereoVisionImuFrontend.cpp:380-385)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ int wave_tot[RS_T / 64];
  const RsLds L = rs_carve(lds_raw, P.kcap);
  const size_t so = (size_t)s * P.kcap;
  const double* R = S.kf_R_cur + (size_t)s * 9;
  const bool imu_ok = !rs_rot_is_identity(R);
  if (!(P.ransac_1pt_stereo && imu_ok)) {  // 3-point problem (stereo_arun_kernel); zero information
    if (tid < 9) S.trk_info[9 * (size_t)s + tid] = 0.0;
    return;
  }
  int2* matches = RS.matches + so;
  const int n = rs_build_matches(P, K, LKF, ST.right_status, LST.right_status, s, S.n_tracked[s],
                                 L.ids, L.idx, wave_tot, matches);
  double Rl[9];
  for (int i = 0; i < 9; i++) Rl[i] = R[i];
  auto get = [&](int m, double* rl, double* rp, double* cl, double* cp) {
    const int2 mm = matches[m];
    const float2 a = LST.left_rect[so + mm.x], b = ST.left_rect[so + mm.y];
    rl[0] = (double)a.x;
    rl[1] = (double)LST.right_rect[so + mm.x].x;
    rl[2] = (double)a.y;
    cl[0] = (double)b.x;
    cl[1] = (double)ST.right_rect[so + mm.y].x;
    cl[2] = (double)b.y;
    for (int c = 0; c < 3; c++) {
      rp[c] = LST.kp3d[(so + mm.x) * 3 + c];
      cp[c] = ST.kp3d[(so + mm.y) * 3 + c];
    }
  };
  rs_vote_prepare(P, n, Rl, get, RS.f_ref + so * 3, RS.relc + so * 9, RS.votef + so * 12, RS.cnt + so);
  if (tid == 0) RS.n_matches[s] = n;
}

__global__ __launch_bounds__(64) void stereo_ransac_tile_kernel(KParams P, RansacScratch RS) {
  const int s = blockIdx.y;
  const int n = RS.n_matches[s];
  if (n <= 1) return;
  const size_t so = (size_t)s * P.kcap;
  rs_vote_tile(P, n, blockIdx.x, RS.votef + so * 12, RS.cnt + so);
}

__global__ __launch_bounds__(RS_T) void stereo_ransac_finish_kernel(KParams P, StreamState S,
                                                                    RansacScratch RS) {
  const int s = blockIdx.x, tid = threadIdx.x;
  const int n = RS.n_matches[s];
  if (n < 0) return;
  __shared__ int wave_tot[RS_T / 64];
  const size_t so = (size_t)s * P.kcap;
  Rs3d3dResult res = rs_vote_finish(P, n, RS.f_ref + so * 3, RS.relc + so * 9, RS.votef + so * 12,
                                    RS.acc + so * 12, RS.cnt + so, wave_tot, RS.inliers + so);
  if (tid == 0) {
    S.trk_status[2 * (size_t)s + 1] = res.status;
    if (res.status != TRK_INVALID) {
      int* cn = S.trk_counts + 6 * (size_t)s;
      cn[3] = n;
      cn[4] = res.n_inliers;
    }
    if (res.status == TRK_VALID) {
      const double* R = S.kf_R_cur + (size_t)s * 9;
      double* pose = S.trk_pose + 24 * (size_t)s + 12;
      for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) pose[r * 4 + c] = R[r * 3 + c];
        pose[r * 4 + 3] = res.t[r];
      }
    }
  }
  if (tid < 9) S.trk_info[9 * (size_t)s + tid] = res.info[tid];
}

static int rs_n_tiles(int max_matches) {
  const int nb = (max_matches + 63) / 64;
  return nb * (nb + 1) / 2;
}

void launch_stereo_ransac(const KParams& P, const Tables& T, const FrameTab& k, const FrameTab& lkf,
                          const StereoTab& ST, const StereoTab& LST, const StreamState& S,
                          const RansacScratch& RS, int max_matches, hipStream_t st, bool need_arun) {
  hipLaunchKernelGGL(stereo_ransac_prepare_kernel, dim3(P.B), dim3(RS_T), rs_lds_bytes(P.kcap), st, P, T,
                     k, lkf, ST, LST, S, RS);
  hipLaunchKernelGGL(stereo_ransac_tile_kernel, dim3(rs_n_tiles(max_matches), P.B), dim3(64), 0, st, P, RS);
  hipLaunchKernelGGL(stereo_ransac_finish_kernel, dim3(P.B), dim3(RS_T), 0, st, P, S, RS);
  // the 3-point kernel serves the streams the voting does not (no usable gyro rotation, or ransac_use_1point_stereo
  // off); the caller knows the step's rotations, so a step in which every stream votes does not launch it
  if (need_arun)
    hipLaunchKernelGGL(stereo_arun_kernel, dim3(P.B), dim3(RS_T), rs_lds_bytes(P.kcap), st, P, T, k, lkf, ST, LST,
                       S, RS);
}

// ---------------------------------------------------------------------------------------------
// component API: the two problems on caller-supplied matches
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RS_T) void ransac_2d2d_points_kernel(KParams P, Tables T,
                                                                  const double* f_ref,
                                                                  const double* f_cur, int n,
                                                                  const double* R, RansacScratch RS,
                                                                  int* out_status, double* out_pose,
                                                                  int* out_counts) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ int wave_tot[RS_T / 64];
  const RsLds L = rs_carve(lds_raw, P.kcap);
  double Rl[9];
  for (int i = 0; i < 9; i++) Rl[i] = R[i];
  Rs2d2dResult res = rs_ransac_2d2d(P, T, f_ref, f_cur, n, Rl, L.work, wave_tot, RS.inliers);
  if (threadIdx.x == 0) {
    out_status[0] = res.status;
    out_counts[0] = n;
    out_counts[1] = res.n_inliers;
    out_counts[2] = res.iterations;
    for (int i = 0; i < 12; i++) out_pose[i] = res.pose[i];
    RS.n_inliers[0] = res.n_inliers;
  }
}

void launch_ransac_2d2d_points(const KParams& P, const Tables& T, const double* f_ref,
                               const double* f_cur, int n, const double* R, const RansacScratch& RS,
                               int* out_status, double* out_pose, int* out_counts, hipStream_t st) {
  hipLaunchKernelGGL(ransac_2d2d_points_kernel, dim3(1), dim3(RS_T), rs_lds_bytes(P.kcap), st, P, T,
                     f_ref, f_cur, n, R, RS, out_status, out_pose, out_counts);
}

// Tracker::geometricOutlierRejection2d2d without rotation prior (5-point Nister) on caller-supplied matches
__global__ __launch_bounds__(RS_T) void ransac_2d2d_nister_points_kernel(KParams P, Tables T, const double* f_ref,
                                                                         const double* f_cur, int n,
                                                                         RansacScratch RS, int* out_status,
                                                                         double* out_pose, int* out_counts) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ int wave_tot[RS_T / 64];
  const RsLds L = rs_carve(lds_raw, P.kcap);
  Rs2d2dResult res = rs_ransac_nister(P, T, f_ref, f_cur, n, L.work, wave_tot, RS.inliers);
  if (threadIdx.x == 0) {
    out_status[0] = res.status;
    out_counts[0] = n;
    out_counts[1] = res.n_inliers;
    out_counts[2] = res.iterations;
    for (int i = 0; i < 12; i++) out_pose[i] = res.pose[i];
    RS.n_inliers[0] = res.n_inliers;
  }
}
void launch_ransac_2d2d_nister_points(const KParams& P, const Tables& T, const double* f_ref, const double* f_cur,
                                      int n, const RansacScratch& RS, int* out_status, double* out_pose,
                                      int* out_counts, hipStream_t st) {
  hipLaunchKernelGGL(ransac_2d2d_nister_points_kernel, dim3(1), dim3(RS_T), rs_lds_bytes(P.kcap), st, P, T, f_ref,
                     f_cur, n, RS, out_status, out_pose, out_counts);
}

__global__ __launch_bounds__(RS_T) void ransac_3d3d_prepare_kernel(
    KParams P, const float* ref_left, const float* ref_right_x, const double* ref_p3,
    const float* cur_left, const float* cur_right_x, const double* cur_p3, int n, const double* R,
    RansacScratch RS) {
  double Rl[9];
  for (int i = 0; i < 9; i++) Rl[i] = R[i];
  auto get = [&](int m, double* rl, double* rp, double* cl, double* cp) {
    rl[0] = (double)ref_left[2 * m];
    rl[1] = (double)ref_right_x[m];
    rl[2] = (double)ref_left[2 * m + 1];
    cl[0] = (double)cur_left[2 * m];
    cl[1] = (double)cur_right_x[m];
    cl[2] = (double)cur_left[2 * m + 1];
    for (int c = 0; c < 3; c++) {
      rp[c] = ref_p3[3 * (size_t)m + c];
      cp[c] = cur_p3[3 * (size_t)m + c];
    }
  };
  rs_vote_prepare(P, n, Rl, get, RS.f_ref, RS.relc, RS.votef, RS.cnt);
  if (threadIdx.x == 0) RS.n_matches[0] = n;
}

__global__ __launch_bounds__(RS_T) void ransac_3d3d_finish_kernel(KParams P, int n, const double* R,
                                                                  RansacScratch RS, int* out_status,
                                                                  double* out_pose, double* out_info,
                                                                  int* out_counts) {
  __shared__ int wave_tot[RS_T / 64];
  Rs3d3dResult res = rs_vote_finish(P, n, RS.f_ref, RS.relc, RS.votef, RS.acc, RS.cnt, wave_tot, RS.inliers);
  if (threadIdx.x == 0) {
    out_status[0] = res.status;
    out_counts[0] = n;
    out_counts[1] = res.n_inliers;
    out_counts[2] = 1;
    RS.n_inliers[0] = res.n_inliers;
    for (int i = 0; i < 12; i++) out_pose[i] = (i % 5 == 0) ? 1.0 : 0.0;
    if (res.status != TRK_INVALID)
      for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) out_pose[r * 4 + c] = R[r * 3 + c];
        out_pose[r * 4 + 3] = res.t[r];
      }
    for (int i = 0; i < 9; i++) out_info[i] = res.info[i];
  }
}

void launch_ransac_3d3d_points(const KParams& P, const Tables& T, const float* ref_left,
                               const float* ref_right_x, const double* ref_p3, const float* cur_left,
                               const float* cur_right_x, const double* cur_p3, int n, const double* R,
                               const RansacScratch& RS, int* out_status, double* out_pose,
                               double* out_info, int* out_counts, hipStream_t st) {
  hipLaunchKernelGGL(ransac_3d3d_prepare_kernel, dim3(1), dim3(RS_T), 0, st, P, ref_left, ref_right_x,
                     ref_p3, cur_left, cur_right_x, cur_p3, n, R, RS);
  if (n > 1)
    hipLaunchKernelGGL(stereo_ransac_tile_kernel, dim3(rs_n_tiles(n), 1), dim3(64), 0, st, P, RS);
  hipLaunchKernelGGL(ransac_3d3d_finish_kernel, dim3(1), dim3(RS_T), 0, st, P, n, R, RS, out_status,
                     out_pose, out_info, out_counts);
}

// Tracker::geometricOutlierRejection3d3d on caller-supplied matched points (one problem)
__global__ __launch_bounds__(RS_T) void ransac_3d3d_arun_points_kernel(KParams P, Tables T, const double* p1,
                                                                       const double* p2, int n, RansacScratch RS,
                                                                       int* out_status, double* out_pose,
                                                                       int* out_counts) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ int wave_tot[RS_T / 64];
  const RsLds L = rs_carve(lds_raw, P.kcap);
  const Rs2d2dResult res = rs_ransac_arun(P, T, p1, p2, n, L.work, wave_tot, RS.inliers);
  if (threadIdx.x == 0) {
    out_status[0] = res.status;
    out_counts[0] = res.n_inliers;
    out_counts[1] = res.iterations;
    for (int i = 0; i < 12; i++) out_pose[i] = res.pose[i];
  }
}
void launch_ransac_3d3d_arun_points(const KParams& P, const Tables& T, const double* p1, const double* p2, int n,
                                    const RansacScratch& RS, int* out_status, double* out_pose, int* out_counts,
                                    hipStream_t st) {
  hipLaunchKernelGGL(ransac_3d3d_arun_points_kernel, dim3(1), dim3(RS_T), rs_lds_bytes(P.kcap), st, P, T, p1, p2,
                     n, RS, out_status, out_pose, out_counts);
}

int ransac_max_kcap(bool nister) {
  int budget = lds_dynamic_budget(reinterpret_cast<const void*>(mono_ransac_kernel<false>));
  if (nister) budget = std::min(budget, lds_dynamic_budget(reinterpret_cast<const void*>(mono_ransac_kernel<true>)));
  const void* others[] = {reinterpret_cast<const void*>(stereo_ransac_prepare_kernel),
                          reinterpret_cast<const void*>(stereo_arun_kernel),
                          reinterpret_cast<const void*>(ransac_2d2d_points_kernel),
                          reinterpret_cast<const void*>(ransac_3d3d_arun_points_kernel)};
  for (const void* k : others) budget = std::min(budget, lds_dynamic_budget(k));
  if (nister)
    budget = std::min(budget, lds_dynamic_budget(reinterpret_cast<const void*>(ransac_2d2d_nister_points_kernel)));
  return (int)((budget - 16) / (sizeof(long long) + 3 * sizeof(int)));
}

#include "k_pnp.inl"

}  // namespace kvfe
