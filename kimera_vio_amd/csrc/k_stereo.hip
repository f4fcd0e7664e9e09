// K7 + K5  sparse stereo:  StereoMatcher::sparseStereoReconstruction (src/frontend/StereoMatcher.cpp:123-175)
//   undistortRectifyLeftKeypoints  (StereoCamera.cpp:236-260, UndistorterRectifier.cpp:138-211)
//   getRightKeypointsRectified / searchRightKeypointEpipolar (StereoMatcher.cpp:196-423):
//     cv::matchTemplate(TM_SQDIFF) + normalize + minMaxLoc == exact integer SSD + first argmin
//   getDepthFromRectifiedMatches (StereoMatcher.cpp:425-483)
//   distortUnrectifyRightKeypoints (UndistorterRectifier.cpp:213-228), keypoints_3d_ (:157-174)
// and the end-of-frame bookkeeping of StereoVisionImuFrontend::processStereoFrame (:448-475) /
// getSmartStereoMeasurements (:485-531).
//
// One wavefront per keypoint: template and stripe rows are staged in LDS, every lane owns a set
// of horizontal offsets and accumulates an exact int32 SSD; a 64-bit (ssd, offset) min-reduction
// by wave shuffles returns the first minimum, as cv::minMaxLoc does.
#include "kvfe_dev.hpp"

#include <cstdlib>
#include <utility>

namespace kvfe {

#include "kvfe_undistort.inl"

#include "kvfe_subpix.inl"

// keypoint range of a stereo launch: all keypoints of the frame, only the tracked ones (known as
// soon as tracking is done, so their matching overlaps corner refinement on another HIP stream),
// or only the newly detected ones
enum : int { STEREO_ALL = 0, STEREO_TRACKED = 1, STEREO_NEW = 2 };
__device__ __forceinline__ bool stereo_range(const FrameTab& K, const StreamState& S, int s, int mode,
                                             int idx, int* i) {
  const int nt = S.n_tracked[s];
  if (mode == STEREO_TRACKED) {
    *i = idx;
    return idx < nt;
  }
  if (mode == STEREO_NEW) {
    *i = nt + idx;
    return *i < K.count[s];
  }
  *i = idx;
  return idx < K.count[s];
}

__global__ void stereo_left_kernel(KParams P, Tables T, FrameTab K, StereoTab ST, StreamState S,
                                   int act_flag, int mode) {
  const int s = blockIdx.y;
  if (!(S.flags[s] & act_flag)) return;
  int i;
  if (!stereo_range(K, S, s, mode, blockIdx.x * blockDim.x + threadIdx.x, &i)) return;
  const size_t o = (size_t)s * P.kcap + i;
  const float2 d = K.kp[o];
  float ux, uy;
  undistort_point_dev(T.und_left_RP, d.x, d.y, &ux, &uy);
  // cropToSize (UtilsOpenCV.cpp:215-235)
  bool cropped = false;
  const float maxw = (float)(P.W - 1), maxh = (float)(P.H - 1);
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
  const float2 e = T.map[0][(size_t)ry * P.W + rx];
  unsigned char status = 0;
  if (cropped)
    status = 1;  // NO_LEFT_RECT
  else if (fabsf(d.x - e.x) > 2.0f || fabsf(d.y - e.y) > 2.0f)
    status = 1;
  ST.left_rect[o] = make_float2(ux, uy);
  ST.left_status[o] = status;
}

// LDS geometry of match_one (dwords), shared by the kernel and the launcher
struct StereoGeom {
  int tcw, MC, TSW, SSW, P2N;
  int KS, TPW, SPB;      // MFMA search (ssd_search_mfma): K steps of 64 bytes, template row pitch (dwords), stripe row pitch (bytes)
  bool mfma_ok;          // its lane maps hold: one stripe dword and one template dword per lane and row
  size_t match_bytes;
};
__host__ __device__ inline StereoGeom stereo_geom(const KParams& P) {
  StereoGeom g;
  g.tcw = (P.templ_cols + 3) >> 2;              // template dwords per row
  g.MC = (g.tcw + 3 + 3) >> 2;                  // 16-byte stripe chunks walked per row
  g.TSW = 4 * (g.MC + 1);                       // [4 zeros | template | zeros]
  const int rw = P.stripe_cols - P.templ_cols + 1;
  const int NJ = (3 + rw + 15) >> 4;
  int ssw = 4 * NJ + 4 * g.MC + 4;              // furthest chunk any task reads
  const int ndw = (3 + P.stripe_cols + 3) >> 2;
  if (ssw < ndw + 4) ssw = (ndw + 4 + 3) & ~3;
  g.SSW = ssw;
  g.P2N = 4 * ndw + 4;
  size_t b = 4 * ((size_t)P.templ_rows * g.TSW + (size_t)P.stripe_rows * g.SSW + (size_t)g.P2N);
  // MFMA search: template rows [16 zero bytes | template | zeros] of 64 KS + 32 bytes, stripe rows of whole 16-byte
  // chunks up to the furthest one an A operand reads (offset block NJ - 1, lane group 3, last K step)
  g.KS = (P.templ_cols + 15 + 63) >> 6;
  g.TPW = 16 * g.KS + 8;
  const int nch = NJ + 3 + 4 * (g.KS - 1), nchd = (ndw + 3) >> 2;
  g.SPB = 16 * (nch > nchd ? nch : nchd);
  g.mfma_ok = ndw <= 64 && g.TPW <= 64 && rw >= 1;
  if (g.mfma_ok) {
    // (one template row of zeros and one stripe row of anything behind the real ones: the phantom row of an odd tr)
    const size_t bm = 4 * (size_t)(P.templ_rows + 1) * g.TPW + (size_t)(P.stripe_rows + 1) * g.SPB + 4 * (size_t)g.P2N;
    if (bm > b) b = bm;
  }
  g.match_bytes = (b + 15) & ~(size_t)15;
  return g;
}

typedef int v4i_t __attribute__((ext_vector_type(4)));

// inclusive prefix sum over the wavefront: Hillis-Steele inside the 16-lane DPP rows (row_shr, zero fill), then the
// row totals with row_bcast:15 / row_bcast:31 -- six DPP additions, no LDS permutes
__device__ __forceinline__ unsigned wave_incl_scan(unsigned v) {
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
  return v;
}

// The SSD search of one keypoint on the matrix cores (round 3).  cv::matchTemplate(TM_SQDIFF) over one stripe is, per
// template row, a correlation c(u) = sum_k T[k] S[u + k]: with u = 16 oh + j it is the matrix product
//     C[oh][j] = sum_k' A[oh][k'] B[k'][j],   A[oh][k'] = S[16 oh + k'],   B[k'][j] = T[k' - j]  (0 outside the template)
// -- A's rows are the stripe read at 16-byte steps (one aligned ds_read_b128 per lane: lane (oh, g) holds bytes
// 16 (oh + g + 4 s) .. +15 of the row for K step s), B's columns the template shifted by j bytes (five dwords of the
// zero-padded template row funnel-shifted by v_alignbyte) -- and v_mfma_i32_16x16x64_i8 does 16 x 16 x 64 of the
// products per instruction: tr x KS MFMAs (22 for the shipped 101 x 11 template) replace ~770 v_dot4 wave-instructions.
// The instruction multiplies SIGNED bytes, so both images are staged as x - 128 (x ^ 0x80): the SSD
// sum (T - S)^2 = sum T'^2 + sum S'^2 - 2 sum T'S' is the same integer in the shifted domain, the zero padding of the
// template stays zero, and every sum is exact in int32 (1111 * 128^2 < 2^25).  Stripe bytes beyond the staged ones
// are never initialised: they only ever meet zero template bytes.
// LDS bytes: template (tr + 1) x 4 TPW | stripe (sr + 1) x SPB | P2 (prefix of the per-column sums of S'^2 over the tr rows).
// Returns the wave's candidate keys (ssd << 32 | oy * rw + ox), one per lane.
// KSC: the number of K steps at compile time (2 for the shipped template; 0 = any, plain loop); SWAP: operands
// exchanged (KVFE_SSD_IMPL=3, a bring-up switch: it must FAIL the parity tests).
// kvfe_stereo_params.ssd_tie_policy: the value the minimum is taken over -- the exact integer SSD, or the SSD rounded
// to float32 (v_cvt_f32_u32 rounds to nearest even; positive floats order like their bit patterns), i.e. what a CV_32F
// result matrix holds.  The offset in the low word keeps "first minimum in row-major order" in both cases.
__device__ __forceinline__ unsigned ssd_key(const KParams& P, unsigned ssd) {
  return P.ssd_f32 ? __float_as_uint((float)ssd) : ssd;
}

template <int KSC, bool SWAP>
__device__ __forceinline__ unsigned long long ssd_search_mfma(const KParams& P, const StereoGeom& G,
                                                              const unsigned char* __restrict__ L,
                                                              const unsigned char* __restrict__ R, unsigned char* lds,
                                                              int lane, int tcx, int tcy, int scx, int scy) {
  const int W = P.W, tc = P.templ_cols, tr = P.templ_rows, sc = P.stripe_cols, sr = P.stripe_rows;
  const int KS = G.KS, TPW = G.TPW, SPB = G.SPB, SPW = SPB >> 2, tcw = G.tcw, W4 = W >> 2;
  unsigned* Xw = reinterpret_cast<unsigned*>(lds);
  unsigned char* stpb = lds + (size_t)(tr + 1) * TPW * 4;
  unsigned* stpw = reinterpret_cast<unsigned*>(stpb);
  unsigned* P2 = stpw + (sr + 1) * SPW;
  const int sh0 = scx & 3, x_al = scx - sh0;
  const int ndw = (sh0 + sc + 3) >> 2;
  int t2 = 0;
  if (lane < TPW) {   // template dword m = lane - 4 of every row (the four dwords in front of it are the zero pad)
    const int m = lane - 4, tsh = tcx & 3, tail = tc & 3;
    const bool ld = m >= 0 && m < tcw;
    const unsigned* src = reinterpret_cast<const unsigned*>(L + (size_t)tcy * W + (tcx - tsh)) + (ld ? m : 0);
    const unsigned keep = (m == tcw - 1 && tail) ? (1u << (8 * tail)) - 1u : 0xffffffffu;
    for (int y = 0; y < tr; y++) {
      unsigned v = 0;
      if (ld) v = (__builtin_amdgcn_alignbyte(src[1], src[0], tsh) ^ 0x80808080u) & keep;
      src += W4;
      Xw[y * TPW + lane] = v;
      t2 = __builtin_amdgcn_sdot4((int)v, (int)v, t2, false);
    }
    Xw[tr * TPW + lane] = 0u;
  }
  if (lane < ndw) {
    const unsigned* rs = reinterpret_cast<const unsigned*>(R + (size_t)scy * W + x_al) + lane;
    for (int y = 0; y < sr; y++) {
      stpw[y * SPW + lane] = rs[0] ^ 0x80808080u;
      rs += W4;
    }
  }
  __syncthreads();
  const unsigned t2u = (unsigned)__builtin_amdgcn_readlane((int)wave_incl_scan((unsigned)t2), 63);
  const int rw = sc - tc + 1, rh = sr - tr + 1;
  const int NJ = (sh0 + rw + 15) >> 4;
  const int j = lane & 15, g = lane >> 4;
  const int jc = (j + 3) >> 2, bsh = (4 - (j & 3)) & 3;   // B: bytes 64 s + 16 g + 16 - j .. of the padded template row
  unsigned long long best = ~0ull;
  for (int oy = 0; oy < rh; oy++) {
    {  // P2: lane = stripe dword
      int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
      if (lane < ndw)
        for (int y = 0; y < tr; y++) {
          const int v = (int)stpw[(oy + y) * SPW + lane];
          const int b0 = __builtin_amdgcn_sbfe(v, 0, 8), b1 = __builtin_amdgcn_sbfe(v, 8, 8),
                    b2 = __builtin_amdgcn_sbfe(v, 16, 8), b3 = v >> 24;
          c0 += __mul24(b0, b0);
          c1 += __mul24(b1, b1);
          c2 += __mul24(b2, b2);
          c3 += __mul24(b3, b3);
        }
      const unsigned run = (unsigned)(c0 + c1 + c2 + c3);
      const unsigned inc = wave_incl_scan(run);
      const unsigned base = inc - run;
      if (lane < ndw)
        *reinterpret_cast<uint4*>(P2 + 4 * lane) =
            make_uint4(base, base + (unsigned)c0, base + (unsigned)(c0 + c1), base + (unsigned)(c0 + c1 + c2));
      if (lane == 63) P2[4 * ndw] = inc;
      __syncthreads();
    }
    for (int mt = 0; 16 * mt < NJ; mt++) {
      const int ia = min(16 * mt + j, NJ - 1);   // A: offset block of this lane (unused rows re-read the last one)
      const unsigned char* arow = stpb + (size_t)oy * SPB + 16 * (ia + g);
      const unsigned* brow = Xw + (4 * g + 4 - jc);
      v4i_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
      if (KSC > 0) {
        // two rows per trip, two operand sets: the operands of the next row are requested before the MFMAs of this one
        // are issued, nothing is copied between trips.  An odd tr runs one phantom row: template row tr is all zeros
        // and stripe row oy + tr is staged memory of any content.  K step s accumulates into accumulator s & 1.
        constexpr int KN = KSC > 0 ? KSC : 1;
        v4i_t a0[KN], a1[KN];
        unsigned x0[KN][5], x1[KN][5];
#define KVFE_MM_LOAD(a_, x_)                                              \
  _Pragma("unroll") for (int s = 0; s < KN; s++) {                        \
    a_[s] = *reinterpret_cast<const v4i_t*>(arow + 64 * s);               \
    _Pragma("unroll") for (int w = 0; w < 5; w++) x_[s][w] = brow[16 * s + w]; \
  }                                                                       \
  arow += SPB;                                                            \
  brow += TPW;
#define KVFE_MM_MULT(a_, x_)                                                                   \
  _Pragma("unroll") for (int s = 0; s < KN; s++) {                                             \
    v4i_t B;                                                                                   \
    B.x = (int)__builtin_amdgcn_alignbyte(x_[s][1], x_[s][0], bsh);                            \
    B.y = (int)__builtin_amdgcn_alignbyte(x_[s][2], x_[s][1], bsh);                            \
    B.z = (int)__builtin_amdgcn_alignbyte(x_[s][3], x_[s][2], bsh);                            \
    B.w = (int)__builtin_amdgcn_alignbyte(x_[s][4], x_[s][3], bsh);                            \
    if (s & 1)                                                                                 \
      acc1 = SWAP ? __builtin_amdgcn_mfma_i32_16x16x64_i8(B, a_[s], acc1, 0, 0, 0)             \
                  : __builtin_amdgcn_mfma_i32_16x16x64_i8(a_[s], B, acc1, 0, 0, 0);            \
    else                                                                                       \
      acc0 = SWAP ? __builtin_amdgcn_mfma_i32_16x16x64_i8(B, a_[s], acc0, 0, 0, 0)             \
                  : __builtin_amdgcn_mfma_i32_16x16x64_i8(a_[s], B, acc0, 0, 0, 0);            \
  }
        KVFE_MM_LOAD(a0, x0)
        for (int y = 0; y < tr; y += 2) {
          KVFE_MM_LOAD(a1, x1)
          KVFE_MM_MULT(a0, x0)
          if (y + 2 < tr) {   // (uniform) the last trip has nothing further to request
            KVFE_MM_LOAD(a0, x0)
          }
          KVFE_MM_MULT(a1, x1)
        }
#undef KVFE_MM_LOAD
#undef KVFE_MM_MULT
      } else {
        for (int y = 0; y < tr; y++) {
          for (int s = 0; s < KS; s++) {
            const v4i_t A = *reinterpret_cast<const v4i_t*>(arow + 64 * s);
            const unsigned* bp = brow + 16 * s;
            const unsigned x0 = bp[0], x1 = bp[1], x2 = bp[2], x3 = bp[3], x4 = bp[4];
            v4i_t B;
            B.x = (int)__builtin_amdgcn_alignbyte(x1, x0, bsh);
            B.y = (int)__builtin_amdgcn_alignbyte(x2, x1, bsh);
            B.z = (int)__builtin_amdgcn_alignbyte(x3, x2, bsh);
            B.w = (int)__builtin_amdgcn_alignbyte(x4, x3, bsh);
            acc0 = SWAP ? __builtin_amdgcn_mfma_i32_16x16x64_i8(B, A, acc0, 0, 0, 0)
                        : __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, acc0, 0, 0, 0);
          }
          arow += SPB;
          brow += TPW;
        }
      }
      acc0 += acc1;
      // D: column (lane & 15) = j, rows 4 (lane >> 4) + r = offset block
      const int cs[4] = {acc0.x, acc0.y, acc0.z, acc0.w};
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int oh = 16 * mt + 4 * g + r;
        const int u = 16 * oh + j;
        const int ox = u - sh0;
        if (oh < NJ && ox >= 0 && ox < rw) {
          const unsigned ss = P2[u + tc] - P2[u];
          const unsigned ssd = t2u + ss - 2u * (unsigned)cs[r];
          const unsigned long long key = ((unsigned long long)ssd_key(P, ssd) << 32) | (unsigned)(oy * rw + ox);
          best = key < best ? key : best;
        }
      }
    }
    __syncthreads();
  }
  return best;
}

// core of searchRightKeypointEpipolar for one keypoint, executed by one wavefront
// SUBPIX = StereoMatchingParams::subpixel_refinement_ (off in every shipped parameter set): the
// refinement code is compiled only into the <true> instantiation so that the common kernel keeps a
// small register footprint.
template <bool SUBPIX>
__device__ void match_one(const KParams& P, const Tables& T, const unsigned char* __restrict__ L,
                          const unsigned char* __restrict__ R, float2 lkp, unsigned char* lds,
                          int lane, float2* out_kp, int* out_status, double* out_score, int impl) {
  const int W = P.W, H = P.H, tc = P.templ_cols, tr = P.templ_rows, sc = P.stripe_cols,
            sr = P.stripe_rows;
  const int rx = (int)roundf(lkp.x), ry = (int)roundf(lkp.y);
  int temp_corner_y = ry - (tr - 1) / 2;
  if (temp_corner_y < 0 || temp_corner_y + tr > H - 1) {
    *out_score = -1.0;
    *out_status = 2;  // NO_RIGHT_RECT
    *out_kp = make_float2(0.f, 0.f);
    return;
  }
  int offset_temp = 0;
  int temp_corner_x = rx - (tc - 1) / 2;
  if (temp_corner_x < 0) {
    offset_temp = temp_corner_x;
    temp_corner_x = 0;
  }
  if (temp_corner_x + tc > W - 1) {
    offset_temp = (temp_corner_x + tc) - (W - 1);
    temp_corner_x -= offset_temp;
  }
  const int stripe_corner_y = ry - (sr - 1) / 2;
  if (stripe_corner_y < 0 || stripe_corner_y + sr > H - 1) {
    *out_score = -1.0;
    *out_status = 2;
    *out_kp = make_float2(0.f, 0.f);
    return;
  }
  int stripe_corner_x = rx + (tc - 1) / 2 - sc;
  if (stripe_corner_x + sc > W - 1) {
    const int offset_stripe = (stripe_corner_x + sc) - (W - 1);
    stripe_corner_x -= offset_stripe;
  }
  if (stripe_corner_x < 0) stripe_corner_x = 0;

  // ---- LDS layout (dwords) -----------------------------------------------------------------------
  //   template : tr rows x TSW, row = [4 zero dwords | tcw template dwords | zeros]; the last
  //              template dword is zero padded, so padding contributes nothing to sum(T*S)
  //   stripe   : sr rows x SSW, staged from the dword-aligned position left of the stripe
  //              (LDS byte u <-> stripe offset u - sh0), zero filled beyond the loaded dwords
  //   P2       : exclusive prefix sums of the per-column sum of squares over the tr stripe rows
  // SSD(o) = sum T^2 + sum S_o^2 - 2 sum T*S_o, exact in uint32 (101*11*255^2 < 2^27).
  // sum S_o^2 comes from P2; sum T*S_o from v_dot4_u32_u8: a lane owns FOUR consecutive dword
  // offsets of one byte phase r = lane & 3 (task (J, r): LDS bytes u = 16 J + 4 a + r, a < 4),
  // walks the stripe in 16-byte chunks (one ds_read_b128 + 4 v_alignbyte for the phase) and reuses
  // each chunk for the four offsets against a sliding window of template dwords: 16 dot4 per
  // 2 LDS reads.
  const StereoGeom G = stereo_geom(P);
  const int tcw = G.tcw, MC = G.MC, TSW = G.TSW, SSW = G.SSW;
  unsigned* tplw = reinterpret_cast<unsigned*>(lds);
  unsigned* stpw = tplw + tr * TSW;
  unsigned* P2 = stpw + sr * SSW;
  const uint4* tpl4 = reinterpret_cast<const uint4*>(lds);   // rows are whole uint4s (TSW, SSW % 4 == 0)
  const uint4* stp4 = tpl4 + tr * (TSW >> 2);
  const int TSW4 = TSW >> 2, SSW4 = SSW >> 2;
  unsigned char* tplb = reinterpret_cast<unsigned char*>(tplw);
  const bool dword_rows = (W & 3) == 0 && ((size_t)R & 3) == 0;
  const int sh0 = dword_rows ? (stripe_corner_x & 3) : 0;
  const int x_al = stripe_corner_x - sh0;
  const int ncol = sh0 + sc;                 // staged stripe bytes per row that hold image data
  const int ndw = (ncol + 3) >> 2;
  const int rw = sc - tc + 1, rh = sr - tr + 1;
  unsigned long long best = ~0ull;
  if (impl != 0 && G.mfma_ok && dword_rows && ((size_t)L & 3) == 0) {
    if (G.KS == 2)
      best = ssd_search_mfma<2, false>(P, G, L, R, lds, lane, temp_corner_x, temp_corner_y, stripe_corner_x, stripe_corner_y);
    else
      best = ssd_search_mfma<0, false>(P, G, L, R, lds, lane, temp_corner_x, temp_corner_y, stripe_corner_x, stripe_corner_y);
  } else {
  {  // zero fill, 16 bytes per store (both areas are whole uint4s and contiguous)
    uint4* z4 = reinterpret_cast<uint4*>(lds);
    const int nz = (tr * TSW + sr * SSW) >> 2;
    for (int e = lane; e < nz; e += 64) z4[e] = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();
  if (dword_rows && ((size_t)L & 3) == 0 && tcw <= 32 && ndw <= 64) {
    // template: aligned dwords of the left image shifted into place by v_alignbyte (no byte loads, no divisions);
    // 32 lanes per row, two rows per trip.  The dword behind the last one of a row is inside the image (the template
    // never touches the last image row), the bytes beyond the template's last column are masked off.
    const int tsh = temp_corner_x & 3;
    const unsigned char* Lrow0 = L + (size_t)temp_corner_y * W + (temp_corner_x - tsh);
    const int q = lane & 31;
    const int tail = tc & 3;  // valid bytes of the last template dword (0: all four)
    for (int y = lane >> 5; y < tr; y += 2) {
      if (q < tcw) {
        const unsigned* src = reinterpret_cast<const unsigned*>(Lrow0 + (size_t)y * W) + q;
        const unsigned d0 = src[0], d1 = src[1];
        unsigned v = __builtin_amdgcn_alignbyte(d1, d0, tsh);
        if (q == tcw - 1 && tail) v &= (1u << (8 * tail)) - 1u;
        tplw[y * TSW + 4 + q] = v;
      }
    }
    for (int y = 0; y < sr; y++)
      if (lane < ndw)
        stpw[y * SSW + lane] =
            *reinterpret_cast<const unsigned*>(R + (size_t)(stripe_corner_y + y) * W + x_al + 4 * lane);
  } else {
    for (int e = lane; e < tr * tc; e += 64) {
      const int y = e / tc, x = e - y * tc;
      tplb[(y * TSW + 4) * 4 + x] = L[(size_t)(temp_corner_y + y) * W + temp_corner_x + x];
    }
    if (dword_rows) {
      for (int e = lane; e < sr * ndw; e += 64) {
        const int y = e / ndw, q = e - y * ndw;
        stpw[y * SSW + q] =
            *reinterpret_cast<const unsigned*>(R + (size_t)(stripe_corner_y + y) * W + x_al + 4 * q);
      }
    } else {
      unsigned char* stpb = reinterpret_cast<unsigned char*>(stpw);
      for (int e = lane; e < sr * sc; e += 64) {
        const int y = e / sc, x = e - y * sc;
        stpb[y * SSW * 4 + x] = R[(size_t)(stripe_corner_y + y) * W + stripe_corner_x + x];
      }
    }
  }
  __syncthreads();
  unsigned t2 = 0;
  for (int e = lane; e < tr * TSW; e += 64) t2 = __builtin_amdgcn_udot4(tplw[e], tplw[e], t2, false);
  for (int off = 32; off > 0; off >>= 1) t2 += (unsigned)__shfl_xor((int)t2, off);
  const int NJ = (sh0 + rw + 15) >> 4;        // tasks: (J, r), J < NJ, r < 4
  const int r = lane & 3;
  const int cpl = ((ndw + 63) >> 6);          // stripe dwords per lane for the column sums
  for (int oy = 0; oy < rh; oy++) {
    // ---- P2: exclusive prefix of the column sums of squares over rows oy .. oy+tr-1 -----------
    {
      unsigned run = 0;
      const int q0 = lane * cpl;
      // first pass: lane total
      for (int q = q0; q < min(q0 + cpl, ndw); q++)
        for (int y = 0; y < tr; y++) {
          const unsigned v = stpw[(oy + y) * SSW + q];
          run = __builtin_amdgcn_udot4(v, v, run, false);
        }
      unsigned inc = run;
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = (unsigned)__shfl_up((int)inc, off);
        if (lane >= off) inc += t;
      }
      unsigned base = inc - run;  // exclusive prefix of this lane's first column
      __syncthreads();
      for (int q = q0; q < min(q0 + cpl, ndw); q++) {
        unsigned c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        for (int y = 0; y < tr; y++) {
          const unsigned v = stpw[(oy + y) * SSW + q];
          const unsigned b0 = v & 0xffu, b1 = (v >> 8) & 0xffu, b2 = (v >> 16) & 0xffu, b3 = v >> 24;
          c0 += b0 * b0;
          c1 += b1 * b1;
          c2 += b2 * b2;
          c3 += b3 * b3;
        }
        P2[4 * q] = base;
        P2[4 * q + 1] = base + c0;
        P2[4 * q + 2] = base + c0 + c1;
        P2[4 * q + 3] = base + c0 + c1 + c2;
        base += c0 + c1 + c2 + c3;
      }
      if (lane == 63) P2[4 * ndw] = inc;  // total (lane 63 ends the last segment or is empty)
      __syncthreads();
    }
    // With the shipped parameters a stripe has about 100 offsets = 28 tasks for 64 lanes.  The SSD sums are exact
    // integers, so the template rows of a task are split over 2 (4) lane groups that take the same tasks and are
    // added up afterwards: 56 of 64 lanes busy instead of 28.
    const int ntask = NJ * 4;
    const int nsplit = ntask <= 16 ? 4 : (ntask <= 32 ? 2 : 1);
    const int gsz = 64 / nsplit;
    const int split = lane / gsz;
    const int y_lo = (tr * split) / nsplit, y_hi = (tr * (split + 1)) / nsplit;
    for (int tbase = 0; tbase < ntask; tbase += gsz) {
      const int task = tbase + (lane & (gsz - 1));
      const bool active = task < ntask;
      const int J = active ? task >> 2 : 0;
      unsigned acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
      // one 16-byte stripe chunk against the template window {tp = dwords 4m-4..4m-1, tq = 4m..4m+3}
#define KVFE_SSD_CHUNK(tp, tq, raw, nxt)                                        \
  {                                                                             \
    const unsigned s0 = __builtin_amdgcn_alignbyte(raw.y, raw.x, r);            \
    const unsigned s1 = __builtin_amdgcn_alignbyte(raw.z, raw.y, r);            \
    const unsigned s2 = __builtin_amdgcn_alignbyte(raw.w, raw.z, r);            \
    const unsigned s3 = __builtin_amdgcn_alignbyte(nxt.x, raw.w, r);            \
    acc0 = __builtin_amdgcn_udot4(tq.x, s0, acc0, false);                       \
    acc1 = __builtin_amdgcn_udot4(tp.w, s0, acc1, false);                       \
    acc2 = __builtin_amdgcn_udot4(tp.z, s0, acc2, false);                       \
    acc3 = __builtin_amdgcn_udot4(tp.y, s0, acc3, false);                       \
    acc0 = __builtin_amdgcn_udot4(tq.y, s1, acc0, false);                       \
    acc1 = __builtin_amdgcn_udot4(tq.x, s1, acc1, false);                       \
    acc2 = __builtin_amdgcn_udot4(tp.w, s1, acc2, false);                       \
    acc3 = __builtin_amdgcn_udot4(tp.z, s1, acc3, false);                       \
    acc0 = __builtin_amdgcn_udot4(tq.z, s2, acc0, false);                       \
    acc1 = __builtin_amdgcn_udot4(tq.y, s2, acc1, false);                       \
    acc2 = __builtin_amdgcn_udot4(tq.x, s2, acc2, false);                       \
    acc3 = __builtin_amdgcn_udot4(tp.w, s2, acc3, false);                       \
    acc0 = __builtin_amdgcn_udot4(tq.w, s3, acc0, false);                       \
    acc1 = __builtin_amdgcn_udot4(tq.z, s3, acc1, false);                       \
    acc2 = __builtin_amdgcn_udot4(tq.y, s3, acc2, false);                       \
    acc3 = __builtin_amdgcn_udot4(tq.x, s3, acc3, false);                       \
  }
      for (int y = y_lo; y < (active ? y_hi : y_lo); y++) {
        const uint4* trow = tpl4 + y * TSW4;
        const uint4* srow = stp4 + (oy + y) * SSW4 + J;
        uint4 ta = trow[0];  // zeros (template dwords -4 .. -1)
        uint4 ra = srow[0];
        int m = 0;
        for (; m + 2 <= MC; m += 2) {  // two chunks per trip: the register windows swap roles
          const uint4 tb = trow[m + 1], rb = srow[m + 1];
          KVFE_SSD_CHUNK(ta, tb, ra, rb)
          ta = trow[m + 2];
          ra = srow[m + 2];
          KVFE_SSD_CHUNK(tb, ta, rb, ra)
        }
        if (m < MC) {
          const uint4 tb = trow[m + 1], rb = srow[m + 1];
          KVFE_SSD_CHUNK(ta, tb, ra, rb)
        }
      }
#undef KVFE_SSD_CHUNK
      if (nsplit >= 2) {  // (wave-uniform) add the row groups: lanes l, l ^ 32 (and l ^ 16) hold the same task
        acc0 += (unsigned)__shfl_xor((int)acc0, 32);
        acc1 += (unsigned)__shfl_xor((int)acc1, 32);
        acc2 += (unsigned)__shfl_xor((int)acc2, 32);
        acc3 += (unsigned)__shfl_xor((int)acc3, 32);
        if (nsplit == 4) {
          acc0 += (unsigned)__shfl_xor((int)acc0, 16);
          acc1 += (unsigned)__shfl_xor((int)acc1, 16);
          acc2 += (unsigned)__shfl_xor((int)acc2, 16);
          acc3 += (unsigned)__shfl_xor((int)acc3, 16);
        }
      }
      const unsigned ts[4] = {acc0, acc1, acc2, acc3};
#pragma unroll
      for (int a = 0; a < 4; a++) {
        const int u = 16 * J + 4 * a + r;
        const int ox = u - sh0;
        if (active && ox >= 0 && ox < rw) {
          const unsigned ss = P2[u + tc] - P2[u];
          const unsigned ssd = t2 + ss - 2u * ts[a];
          const unsigned long long key = ((unsigned long long)ssd_key(P, ssd) << 32) | (unsigned)(oy * rw + ox);
          best = key < best ? key : best;
        }
      }
    }
    __syncthreads();
  }
  }  // dot4 search
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long other = __shfl_xor(best, off);
    best = other < best ? other : best;
  }
  __syncthreads();
  const int o = (int)(unsigned)best;
  const int by = o / rw, bx = o - by * rw;
  const int mx = bx + stripe_corner_x + (tc - 1) / 2 + offset_temp;
  const int my = by + stripe_corner_y + (tr - 1) / 2;
  float2 match = make_float2((float)mx, (float)my);
  if (SUBPIX) {  // cv::cornerSubPix(right_rectified, (10,10), (-1,-1), 40 it, 0.001)
    match = corner_subpix_wave<10>(R, (size_t)W, W, H, match, 10, 40, 0.001 * 0.001, T.subpix_mask10,
                               lds + G.match_bytes, lane);
  }
  const double min_val = 0.0;  // normalised minimum (cv::normalize MINMAX) is always 0
  *out_score = min_val;
  *out_status = (min_val < P.tol_template) ? 0 : 2;
  *out_kp = match;
}

template <bool SUBPIX>
__global__ __launch_bounds__(64) void stereo_match_kernel(KParams P, Tables T,
                                                          const unsigned char* __restrict__ Lr,
                                                          const unsigned char* __restrict__ Rr,
                                                          FrameTab K, StereoTab ST, StreamState S,
                                                          int act_flag, int mode, int impl) {
  const int s = blockIdx.y;
  if (!(S.flags[s] & act_flag)) return;
  int i;
  if (!stereo_range(K, S, s, mode, blockIdx.x, &i)) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int lane = threadIdx.x;
  const size_t o = (size_t)s * P.kcap + i;
  const unsigned char* L = Lr + (size_t)s * P.W * P.H;
  const unsigned char* R = Rr + (size_t)s * P.W * P.H;
  const int lstatus = ST.left_status[o];
  const float2 lkp = ST.left_rect[o];
  float2 rkp = make_float2(0.f, 0.f);
  int rstatus = lstatus;
  double score = -1.0;
  if (lstatus == 0) match_one<SUBPIX>(P, T, L, R, lkp, lds_raw, lane, &rkp, &rstatus, &score, impl);
  if (lane != 0) return;
  // getDepthFromRectifiedMatches (StereoMatcher.cpp:425-483)
  double depth = 0.0;
  if (lstatus == 0 && rstatus == 0) {
    const double disparity = (double)(lkp.x - rkp.x);
    if (disparity >= 0.0) {
      const double d = (P.fx_rect * P.baseline) / disparity;
      if (d < P.min_point_dist || d > P.max_point_dist)
        rstatus = 3;  // NO_DEPTH
      else
        depth = d;
    } else {
      rstatus = 3;
    }
  } else if (lstatus != 0 && rstatus != lstatus) {
    rstatus = lstatus;
  }
  ST.right_rect[o] = rkp;
  ST.right_status[o] = (unsigned char)rstatus;
  ST.depth[o] = depth;
  // distortUnrectifyKeypoints (UndistorterRectifier.cpp:213-228)
  float2 rraw = make_float2(0.f, 0.f);
  if (rstatus == 0) {
    // (cornerSubPix may leave a match up to its half window OUTSIDE the image -- a stripe at the image edge -- and upstream's
    // map_x_.at<float>(round(y), round(x)) is an unchecked read there (cv::Mat::at asserts in debug builds only): undefined
    // upstream, defined here and in the oracle as the map entry of the nearest pixel.  Found by tools/fuzz_components.py
    // (seed 63, configuration 66: a memory access fault when the map happened to end a mapping).)
    const int yy = min(max((int)roundf(rkp.y), 0), P.H - 1), xx = min(max((int)roundf(rkp.x), 0), P.W - 1);
    rraw = T.map[1][(size_t)yy * P.W + xx];
  }
  ST.right_kp[o] = rraw;
  // keypoints_3d_ (StereoMatcher.cpp:157-174)
  double x3 = 0, y3 = 0, z3 = 0;
  if (rstatus == 0) {
    const double* v = K.versor + o * 3;
    x3 = v[0] * depth / v[2];
    y3 = v[1] * depth / v[2];
    z3 = v[2] * depth / v[2];
  }
  ST.kp3d[o * 3] = x3;
  ST.kp3d[o * 3 + 1] = y3;
  ST.kp3d[o * 3 + 2] = z3;
}

// kvfe_config.ssd_impl: 0 (default) = the SSD search on the matrix cores where its lane maps fit (ssd_search_mfma),
// 1 = the v_dot4 search everywhere (also the path of the geometries that do not fit)
static int stereo_ssd_impl(const KParams& P) { return P.ssd_dot4 ? 0 : 1; }

static size_t stereo_lds_bytes(const KParams& P) {
  size_t b = stereo_geom(P).match_bytes;
  if (P.stereo_subpix) b += subpix_geom(10).bytes;
  return b;
}

void launch_stereo(const KParams& P, const Tables& T, const unsigned char* left_rect,
                   const unsigned char* right_rect, const FrameTab& k, const StereoTab& ST,
                   const StreamState& S, int act_flag, int max_kp, int mode, hipStream_t st) {
  const int nb = max_kp > 0 ? (max_kp < P.kcap ? max_kp : P.kcap) : P.kcap;
  hipLaunchKernelGGL(stereo_left_kernel, dim3((nb + 63) / 64, P.B), dim3(64), 0, st, P, T, k,
                     ST, S, act_flag, mode);
  if (P.stereo_subpix)
    hipLaunchKernelGGL(stereo_match_kernel<true>, dim3(nb, P.B), dim3(64), stereo_lds_bytes(P), st,
                       P, T, left_rect, right_rect, k, ST, S, act_flag, mode, stereo_ssd_impl(P));
  else
    hipLaunchKernelGGL(stereo_match_kernel<false>, dim3(nb, P.B), dim3(64), stereo_lds_bytes(P), st,
                       P, T, left_rect, right_rect, k, ST, S, act_flag, mode, stereo_ssd_impl(P));
}

void launch_undistort_left(const KParams& P, const Tables& T, const FrameTab& k, const StereoTab& ST,
                           const StreamState& S, int act_flag, int max_kp, hipStream_t st) {
  const int nb = max_kp > 0 ? (max_kp < P.kcap ? max_kp : P.kcap) : P.kcap;
  hipLaunchKernelGGL(stereo_left_kernel, dim3((nb + 63) / 64, P.B), dim3(64), 0, st, P, T, k, ST, S,
                     act_flag, STEREO_ALL);
}

template <bool SUBPIX>
__global__ __launch_bounds__(64) void stereo_match_only_kernel(
    KParams P, Tables T, const unsigned char* __restrict__ L, const unsigned char* __restrict__ R,
    const float2* __restrict__ lkps, const unsigned char* __restrict__ lstat, int n,
    float2* rkps, unsigned char* rstat, double* scores, int impl) {
  const int i = blockIdx.x;
  if (i >= n) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  float2 rkp = make_float2(0.f, 0.f);
  int rstatus = lstat[i];
  double score = -1.0;
  if (rstatus == 0)
    match_one<SUBPIX>(P, T, L, R, lkps[i], lds_raw, threadIdx.x, &rkp, &rstatus, &score, impl);
  if (threadIdx.x == 0) {
    rkps[i] = rkp;
    rstat[i] = (unsigned char)rstatus;
    if (scores) scores[i] = score;
  }
}

void launch_stereo_match_only(const KParams& P, const Tables& T, const unsigned char* left_rect,
                              const unsigned char* right_rect, const float2* left_rect_kp,
                              const unsigned char* left_status, int n, float2* right_rect_kp,
                              unsigned char* right_status, double* score, hipStream_t st) {
  if (n <= 0) return;
  if (P.stereo_subpix)
    hipLaunchKernelGGL(stereo_match_only_kernel<true>, dim3(n), dim3(64), stereo_lds_bytes(P), st, P,
                       T, left_rect, right_rect, left_rect_kp, left_status, n, right_rect_kp,
                       right_status, score, stereo_ssd_impl(P));
  else
    hipLaunchKernelGGL(stereo_match_only_kernel<false>, dim3(n), dim3(64), stereo_lds_bytes(P), st, P,
                       T, left_rect, right_rect, left_rect_kp, left_status, n, right_rect_kp,
                       right_status, score, stereo_ssd_impl(P));
}

// ---------------------------------------------------------------------------------------------
// end of frame
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void step_finalize_kernel(KParams P, FrameTab K, FrameTab LKF,
                                                            StereoTab ST, StereoTab LST,
                                                            StreamState S) {
  const int s = blockIdx.x, tid = threadIdx.x;
  if (P.quiet_gate && !kvfe_all_quiet(S.flags, P.B)) return;
  const size_t so = (size_t)s * P.kcap;
  const int flags = S.flags[s];
  const int n = K.count[s];
  __shared__ int wave_tot[4];
  __shared__ int sh_off;
  if (flags & FLAG_KEYFRAME) {
    const bool first = (flags & FLAG_FIRST) != 0;  // bootstrapSpinStereo returns no measurements
    if (tid == 0) sh_off = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 256) {
      const int i = base + tid;
      const bool in = i < n;
      long long lmk = -1;
      if (in) {
        // stereoFrame_lkf_ = stereoFrame_k_ (what the next keyframe decision and the next
        // geometric outlier rejection read of it)
        lmk = K.lmk[so + i];
        LKF.kp[so + i] = K.kp[so + i];
        LKF.lmk[so + i] = lmk;
        if (P.use_ransac) {
          for (int c = 0; c < 3; c++) {
            LKF.versor[(so + i) * 3 + c] = K.versor[(so + i) * 3 + c];
            LST.kp3d[(so + i) * 3 + c] = ST.kp3d[(so + i) * 3 + c];
          }
          LST.left_rect[so + i] = ST.left_rect[so + i];
          LST.right_rect[so + i] = ST.right_rect[so + i];
          LST.right_status[so + i] = ST.right_status[so + i];
        }
      }
      // getSmartStereoMeasurements (StereoVisionImuFrontend.cpp:485-531): landmarks that geometric
      // outlier rejection set to -1 carry no measurement, the others keep their order
      const bool meas = in && !first && lmk != -1;
      const int lane = tid & 63, wv = tid >> 6;
      int inc = meas ? 1 : 0;
      for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
      }
      if (lane == 63) wave_tot[wv] = inc;
      __syncthreads();
      int wbase = 0, tot = 0;
      for (int w = 0; w < 4; w++) {
        if (w < wv) wbase += wave_tot[w];
        tot += wave_tot[w];
      }
      const int off0 = sh_off;
      if (meas) {
        const size_t o = so + off0 + wbase + inc - 1;
        const float2 l = ST.left_rect[so + i];
        double uR = __longlong_as_double(0x7ff8000000000000LL);  // quiet NaN
        if (P.meas_right && ST.right_status[so + i] == 0) uR = (double)ST.right_rect[so + i].x;
        S.meas_lmk[o] = lmk;
        S.meas_uLuRv[o * 3] = (double)l.x;
        S.meas_uLuRv[o * 3 + 1] = uR;
        S.meas_uLuRv[o * 3 + 2] = (double)l.y;
      }
      __syncthreads();
      if (tid == 0) sh_off = off0 + tot;
      __syncthreads();
    }
    if (tid == 0) {
      LKF.count[s] = n;
      LKF.timestamp[s] = K.timestamp[s];
      S.n_meas[s] = sh_off;
    }
  } else {
    if (tid == 0) S.n_meas[s] = 0;
  }
  // (keyframe_R_ref_frame_ and the "initialised" flag are written by detect_commit_kernel, earlier in the step: the next
  // step's tracking reads them and must not wait for this kernel)
  if (tid == 0) S.frame_count[s] += 1;
}

// ---------------------------------------------------------------------------------------------
// RgbdVisionImuFrontend: the right view is hallucinated from the registered depth image
// ---------------------------------------------------------------------------------------------
// DepthFrame::getDetectionMask (DepthFrame.cpp:76-96): cv::inRange(depth, min, max) -> 255 / 0
__global__ __launch_bounds__(256) void depth_mask_kernel(KParams P, const void* __restrict__ depth,
                                                         size_t row_stride, size_t img_stride, StreamState S,
                                                         int act_flag, unsigned char* __restrict__ mask) {
  const int s = blockIdx.z, y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
  if (!(S.flags[s] & act_flag) || x >= P.W) return;
  const size_t e = (size_t)s * img_stride + (size_t)y * row_stride + x;
  bool in;
  if (P.depth_f32) {
    const float v = static_cast<const float*>(depth)[e];
    in = P.mask_lo_f <= v && v <= P.mask_hi_f;
  } else {
    const int v = static_cast<const unsigned short*>(depth)[e];
    in = P.mask_lo_u <= v && v <= P.mask_hi_u;
  }
  mask[((size_t)s * P.H + y) * P.W + x] = in ? 255 : 0;
}
void launch_depth_mask(const KParams& P, const void* depth, size_t row_stride, size_t img_stride,
                       const StreamState& S, int act_flag, unsigned char* mask, hipStream_t st) {
  hipLaunchKernelGGL(depth_mask_kernel, dim3((P.W + 255) / 256, P.H, P.B), dim3(256), 0, st, P, depth, row_stride,
                     img_stride, S, act_flag, mask);
}

// RgbdFrame::fillStereoFrame (RgbdFrame.cpp:48-115) with DepthFrame::getDepthAtPoint (DepthFrame.cpp:40-74) and
// RgbdCamera::distortKeypoints (UndistorterRectifier.cpp:213-228), one thread per keypoint
__global__ void rgbd_fill_kernel(KParams P, Tables T, const void* __restrict__ depth, size_t row_stride,
                                 size_t img_stride, FrameTab K, StereoTab ST, StreamState S, int act_flag) {
  const int s = blockIdx.y;
  if (!(S.flags[s] & act_flag)) return;
  int i;
  if (!stereo_range(K, S, s, STEREO_ALL, blockIdx.x * blockDim.x + threadIdx.x, &i)) return;
  const size_t o = (size_t)s * P.kcap + i;
  const int lstat = ST.left_status[o];
  float2 rr = make_float2(0.f, 0.f), rk = make_float2(0.f, 0.f);
  int rstat = lstat;
  double dep = 0.0, p3[3] = {0.0, 0.0, 0.0};
  if (lstat == 0) {
    const float2 pt = K.kp[o], l = ST.left_rect[o];
    const int x = (int)pt.x, y = (int)pt.y;
    float d = __int_as_float(0x7fc00000);   // quiet NaN
    if (!(x < 0 || x >= P.W || y < 0 || y >= P.H)) {
      const size_t e = (size_t)s * img_stride + (size_t)y * row_stride + x;
      d = P.depth_f32 ? static_cast<const float*>(depth)[e] : (float)static_cast<const unsigned short*>(depth)[e];
      d *= P.depth_to_m;
      if (d < P.depth_min) d = __int_as_float(0x7fc00000);
    }
    rstat = 3;   // NO_DEPTH
    if (isfinite(d)) {
      const float disparity = (float)(P.depth_fx_b / (double)d);
      const float uR = l.x - disparity;
      if (!(uR < 0.0f)) {
        rstat = 0;
        rr = make_float2(uR, l.y);
        dep = (double)d;
        const double v2 = K.versor[o * 3 + 2];
        for (int c = 0; c < 3; c++) p3[c] = K.versor[o * 3 + c] * (double)d / v2;
        const int ry = min(max((int)roundf(rr.y), 0), P.H - 1), rx = min(max((int)roundf(rr.x), 0), P.W - 1);   // (see stereo_match_kernel)
        rk = T.map[0][(size_t)ry * P.W + rx];
      }
    }
  }
  ST.right_rect[o] = rr;
  ST.right_status[o] = (unsigned char)rstat;
  ST.depth[o] = dep;
  ST.right_kp[o] = rk;
  for (int c = 0; c < 3; c++) ST.kp3d[o * 3 + c] = p3[c];
}
void launch_rgbd_fill(const KParams& P, const Tables& T, const void* depth, size_t row_stride, size_t img_stride,
                      const FrameTab& k, const StereoTab& ST, const StreamState& S, int act_flag, int max_kp,
                      hipStream_t st) {
  const int nb = max_kp > 0 ? (max_kp < P.kcap ? max_kp : P.kcap) : P.kcap;
  hipLaunchKernelGGL(rgbd_fill_kernel, dim3((nb + 63) / 64, P.B), dim3(64), 0, st, P, T, depth, row_stride,
                     img_stride, k, ST, S, act_flag);
}

void launch_step_finalize(const KParams& P, const FrameTab& k, const FrameTab& lkf,
                          const StereoTab& ST, const StereoTab& LST, const StreamState& S,
                          hipStream_t st) {
  hipLaunchKernelGGL(step_finalize_kernel, dim3(P.B), dim3(256), 0, st, P, k, lkf, ST, LST, S);
}

}  // namespace kvfe
