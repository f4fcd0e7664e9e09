// K4c (round 6)  pyramidal Lucas-Kanade, EIGHT points per wavefront, one float chain per lane
//      == cv::calcOpticalFlowPyrLK (OPTFLOW_USE_INITIAL_FLOW) as called by Tracker::featureTracking,
//         /root/reference/src/frontend/Tracker.cpp:137-146
//
// k_track.hip's lk_kernel_sys gives a point a whole wavefront: OpenCV's SSE accumulators are strictly sequential float
// chains (4 lane classes x & 3, times {b1, b2} per iteration and {A11, A12, A22} per level), so a wave that owns ONE point
// has 8 (12) chains to walk and 4 useful lanes in each of them -- 1 560 of its 4 870 vector instructions per point.
// Here a wave owns 8 points and a lane owns ONE chain:
//
//   lane = 8 * slot + 2 * g + h      slot = point of the wave, g = SSE lane class (window column & 3), h = chain type
//
// * pixels: the 6 columns {g + 4c} of a class are split between the lane pair by rows: lane h owns the window rows
//   {4m + 2h, 4m + 2h + 1} (72 pixels at WIN = 24).  Their template values (folded into the blend accumulator, as in
//   lk_kernel_sys) stay in registers for the level.
// * chains: lane h = 0 walks the class's b1 chain (and A11), lane h = 1 its b2 chain (and A22), over ALL 24 rows in
//   OpenCV's order.  The packed differences of a row pair are handed to the partner lane with one DPP move
//   (quad_perm broadcast of the owner's value to both lanes: "slot" 2m comes from lane h = 0, slot 2m + 1 from lane
//   h = 1, so both lanes run the same instruction stream in row order); each lane keeps the gradient of ITS type for all
//   144 pixels of the class.  A12 is walked redundantly by both lanes from broadcast products.
// * the eight lanes of a point are a little SIMD machine of their own for the per-level set-up: lane l8 walks column group
//   l8 of the Scharr pass down the rows (every source row loaded once), and stages column group l8 of the current-frame
//   window.  All per-point scalars are computed redundantly by the point's lanes.
// * LDS: 2.2 KB per point.  The set-up of a level runs in two halves of 12 window rows (byte patch + derivative patch of a
//   half: 1.9 KB), the current-frame window is staged over it as 16-bit (pixel, pixel + 1) byte pairs.
//   17.5 KB per wave: the register budget (2 waves per SIMD), not LDS, sets the occupancy.
// * the points of a wave iterate in lock step; a point that has converged (or left the image) is masked.
//
// Every integer is exact and every float chain has the order of lk_kernel_sys (= OpenCV's SSE2 build), so the results are
// bit-identical to it and to the oracle (tests/test_gpu_parity.py, tests/test_gpu_bench_configs.py, tools/fuzz_frontend.py).
// The error output of calcOpticalFlowPyrLK is not computed here (the front-end step drops it): launch_lk keeps
// lk_kernel_sys for callers that want it, for windows other than 16 / 24 and behind kvfe_config / KVFE_LK_IMPL=1.
#include "kvfe_dev.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace kvfe {

#include "kvfe_lk.inl"

// BORDER_REFLECT_101 index for -len < p < 2 len - 1 (one fold), branch-free so that the loads of a pass stay in flight
// together: min(|p|, 2 len - 2 - |p|).  launch_lk8 refuses pyramids whose smallest level is narrower than LK8_MIN_DIM
// (a window plus its staging margin could fold twice).
constexpr int LK8_MIN_DIM = 32;
__device__ __forceinline__ int refl1(int p, int len) {
  const int q = max(p, -p);
  return min(q, 2 * len - 2 - q);
}

template <int WIN>
struct Lk8 {
  static constexpr int NC = WIN / 4;         // window columns of one class
  static constexpr int NCH = WIN / 8;        // 8-pixel chunks of a row = b terms per row, class and type
  static constexpr int NM = WIN / 4;         // row-pair groups of a lane (rows 4m + 2h + {0, 1})
  static constexpr int NHALF = 2;            // set-up passes per level
  static constexpr int MH = NM / NHALF;      // row-pair groups per pass
  static constexpr int HR = 4 * MH;          // window rows per pass
  static constexpr int PROWS = HR + 3;       // byte patch rows of a pass
  static constexpr int DROWS = HR + 1;       // derivative rows of a pass
  static constexpr int WS = WIN + 3, WP = WIN + 1;
  static constexpr int NT = (WP + 3) / 4;    // column groups of the Scharr pass (4 outputs each)
  static constexpr int PSTR = 4 * NT;        // byte patch row stride
  static constexpr int DSTR = (4 * NT) | 1;  // derivative patch row stride in dwords
  static constexpr int PATCH_B = PROWS * PSTR;
  static constexpr int DXY_B = DROWS * DSTR * 4;
  static constexpr int JM = 3;               // margin of the staged current-frame window
  static constexpr int JS = WIN + 1 + 2 * JM;   // staged rows
  static constexpr int JW = JS - 1;             // staged byte pairs per row
  static constexpr int NJ = (JW + 3) / 4;       // column groups of the staging pass (4 pairs each)
  static constexpr int JSTRB = 4 * (2 * NJ + 1);   // row stride in bytes: an odd number of dwords
  static constexpr int J_B = JS * JSTRB;
  static constexpr int NEED_B = (PATCH_B + DXY_B > J_B ? PATCH_B + DXY_B : J_B);
  // region stride: a multiple of 16 bytes that moves consecutive points by 4 banks
  static constexpr int RB = ((NEED_B + 127) / 128) * 128 + 16;
  static_assert(NM % NHALF == 0, "row-pair groups split evenly over the passes");
  static_assert(WP == 4 * (NT - 1) + 1, "last derivative group holds exactly one output");
  static_assert(JW == 4 * (NJ - 1) + 2, "last staging group holds exactly two pairs");
  static_assert(NT <= 8 && NJ <= 8, "one column group per lane of a point");
  static_assert(PATCH_B % 4 == 0, "derivative patch starts on a dword");
};


// -DKVFE_LK8_PROF (tools/r6/gpu_lk8_prof.sh; never in the product build): cycle stamps per phase summed over the waves of
// all launches, wave lifetimes of the launches, printed at exit
#ifdef KVFE_LK8_PROF
constexpr int LK8P_N = 12, LK8P_WAVES = 8192;
__device__ unsigned long long kvfe_lk8_prof[LK8P_N];
__device__ unsigned long long kvfe_lk8_life[LK8P_WAVES * 2];   // (start, lifetime) in 10 ns ticks of the last launch
#define LK8P_DECL unsigned long long p8_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, p8_last = __builtin_readcyclecounter(); \
  const unsigned long long p8_t0 = __builtin_amdgcn_s_memrealtime(); unsigned p8_iters = 0, p8_stages = 0
#define LK8P(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); p8_acc[i] += t_ - p8_last; p8_last = t_; } while (0)
#define LK8P_COUNT(v) (v)++
#else
#define LK8P_DECL do { } while (0)
#define LK8P(i) do { } while (0)
#define LK8P_COUNT(v) do { } while (0)
#endif

// value of the pair's lane h = 0 (sp == 0) or h = 1 (sp == 1) in both lanes of the pair
template <int SP>
__device__ __forceinline__ int pair_bcast(int v) {
  return __builtin_amdgcn_update_dpp(0, v, SP ? 0xF5 : 0xA0, 0xf, 0xf, true);   // quad_perm [1,1,3,3] / [0,0,2,2]
}
template <int SP>
__device__ __forceinline__ float pair_bcast_f(float v) {
  return __builtin_bit_cast(float, pair_bcast<SP>(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ int pair_swap(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);               // quad_perm [1,0,3,2]
}


// ---- hand-scheduled blocks (window of 24: six columns per class).  hipcc only emits the accumulate-in-place dot product
// (v_dot2c: a v_mov per start value) and pads every dot product -> other-VALU dependency with s_nop; in these blocks the
// three-operand form is issued six wide, so that every consumer sits at least three instructions behind its producer (the
// wait states a non-dot reader of a dot product needs), and the blocks end with the wait states a DPP / SDWA-sensitive
// reader outside needs (the compiler cannot see into inline asm; tools/check_dpp_hazard.py scans the ISA).
// One window row of a lane (six pixels): blend of the current frame minus the folded template, >> 9, packed in pairs
// (v_ashrrev ... dst_sel:WORD_1 writes the odd pixel's difference into the high half of the even one's register).
__device__ __forceinline__ void lk8_diff_row6(const int (&Ea)[6], const int (&Eb)[6], const int (&ra)[6], int wq0, int wq1,
                                              int (&dd)[3]) {
  int t0, t1, t2, t3, t4, t5;
  asm("v_dot2_i32_i16 %3, %15, %28, %21\n\t"
      "v_dot2_i32_i16 %4, %16, %28, %22\n\t"
      "v_dot2_i32_i16 %5, %17, %28, %23\n\t"
      "v_dot2_i32_i16 %6, %18, %28, %24\n\t"
      "v_dot2_i32_i16 %7, %19, %28, %25\n\t"
      "v_dot2_i32_i16 %8, %20, %28, %26\n\t"
      "v_dot2_i32_i16 %3, %9, %27, %3\n\t"
      "v_dot2_i32_i16 %4, %10, %27, %4\n\t"
      "v_dot2_i32_i16 %5, %11, %27, %5\n\t"
      "v_dot2_i32_i16 %6, %12, %27, %6\n\t"
      "v_dot2_i32_i16 %7, %13, %27, %7\n\t"
      "v_dot2_i32_i16 %8, %14, %27, %8\n\t"
      "v_ashrrev_i32 %0, 9, %3\n\t"
      "v_ashrrev_i32 %1, 9, %5\n\t"
      "v_ashrrev_i32 %2, 9, %7\n\t"
      "v_ashrrev_i32_sdwa %0, 9, %4 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_ashrrev_i32_sdwa %1, 9, %6 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_ashrrev_i32_sdwa %2, 9, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "s_nop 1"
      : "=&v"(dd[0]), "=&v"(dd[1]), "=&v"(dd[2]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5)
      : "v"(Ea[0]), "v"(Ea[1]), "v"(Ea[2]), "v"(Ea[3]), "v"(Ea[4]), "v"(Ea[5]), "v"(Eb[0]), "v"(Eb[1]), "v"(Eb[2]),
        "v"(Eb[3]), "v"(Eb[4]), "v"(Eb[5]), "v"(ra[0]), "v"(ra[1]), "v"(ra[2]), "v"(ra[3]), "v"(ra[4]), "v"(ra[5]),
        "v"(wq0), "v"(wq1));
}
// six chain terms: (d(x) G(x) + d(x + 4) G(x + 4)) as float
__device__ __forceinline__ void lk8_terms6(const int (&d)[6], const int (&gq)[6], float (&f)[6]) {
  asm("v_dot2_i32_i16 %0, %6, %12, 0\n\t"
      "v_dot2_i32_i16 %1, %7, %13, 0\n\t"
      "v_dot2_i32_i16 %2, %8, %14, 0\n\t"
      "v_dot2_i32_i16 %3, %9, %15, 0\n\t"
      "v_dot2_i32_i16 %4, %10, %16, 0\n\t"
      "v_dot2_i32_i16 %5, %11, %17, 0\n\t"
      "v_cvt_f32_i32 %0, %0\n\t"
      "v_cvt_f32_i32 %1, %1\n\t"
      "v_cvt_f32_i32 %2, %2\n\t"
      "v_cvt_f32_i32 %3, %3\n\t"
      "v_cvt_f32_i32 %4, %4\n\t"
      "v_cvt_f32_i32 %5, %5"
      : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]), "=&v"(f[4]), "=&v"(f[5])
      : "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(gq[0]), "v"(gq[1]), "v"(gq[2]), "v"(gq[3]),
        "v"(gq[4]), "v"(gq[5]));
}
// one row of six bilinear blends (a0 . w0 + a1 . w1 + c) >> SH of (value, right neighbour) pairs: the template and
// gradient gather of a level
template <int SH>
__device__ __forceinline__ void lk8_blend_row6(const int (&Ea)[6], const int (&Eb)[6], int wq0, int wq1, int c0,
                                               int (&o)[6]) {
  asm("v_dot2_i32_i16 %0, %12, %19, %20\n\t"
      "v_dot2_i32_i16 %1, %13, %19, %20\n\t"
      "v_dot2_i32_i16 %2, %14, %19, %20\n\t"
      "v_dot2_i32_i16 %3, %15, %19, %20\n\t"
      "v_dot2_i32_i16 %4, %16, %19, %20\n\t"
      "v_dot2_i32_i16 %5, %17, %19, %20\n\t"
      "v_dot2_i32_i16 %0, %6, %18, %0\n\t"
      "v_dot2_i32_i16 %1, %7, %18, %1\n\t"
      "v_dot2_i32_i16 %2, %8, %18, %2\n\t"
      "v_dot2_i32_i16 %3, %9, %18, %3\n\t"
      "v_dot2_i32_i16 %4, %10, %18, %4\n\t"
      "v_dot2_i32_i16 %5, %11, %18, %5\n\t"
      "v_ashrrev_i32 %0, %21, %0\n\t"
      "v_ashrrev_i32 %1, %21, %1\n\t"
      "v_ashrrev_i32 %2, %21, %2\n\t"
      "v_ashrrev_i32 %3, %21, %3\n\t"
      "v_ashrrev_i32 %4, %21, %4\n\t"
      "v_ashrrev_i32 %5, %21, %5"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5])
      : "v"(Ea[0]), "v"(Ea[1]), "v"(Ea[2]), "v"(Ea[3]), "v"(Ea[4]), "v"(Ea[5]), "v"(Eb[0]), "v"(Eb[1]), "v"(Eb[2]),
        "v"(Eb[3]), "v"(Eb[4]), "v"(Eb[5]), "v"(wq0), "v"(wq1), "s"(c0), "n"(SH));
}

template <int WIN>
__global__ __launch_bounds__(64, 2) void lk8_kernel(KParams P, const unsigned char* prev_img, size_t prev_row_stride,
                                                    size_t prev_img_stride, const unsigned char* prev_pyr,
                                                    const unsigned char* cur_img, size_t cur_row_stride,
                                                    size_t cur_img_stride, const unsigned char* cur_pyr, LkScratch lk,
                                                    int iter_cap) {
  using C = Lk8<WIN>;
  constexpr int NC = C::NC, NCH = C::NCH, NM = C::NM, MH = C::MH, HR = C::HR, PROWS = C::PROWS, DROWS = C::DROWS,
                WS = C::WS, WP = C::WP, NT = C::NT, PSTR = C::PSTR, DSTR = C::DSTR, JM = C::JM, JS = C::JS, JW = C::JW,
                NJ = C::NJ, JSTRB = C::JSTRB;
  __shared__ __attribute__((aligned(16))) unsigned char lds[8 * C::RB];

  const int s = blockIdx.y;
  const int npts = lk.npts[s];
  if ((int)blockIdx.x * 8 >= npts) return;
  const int lane = threadIdx.x;
  const int slot = lane >> 3, l8 = lane & 7, g = l8 >> 1, h = l8 & 1;
  const int pbase = lane & ~7;
  const int pt = blockIdx.x * 8 + slot;
  bool valid = pt < npts;
  const size_t po = (size_t)s * P.kcap + (valid ? pt : 0);
  // Tracker.cpp:167-180 drops a point whose landmark is older than maxFeatureAge whatever its tracking result; the step
  // passes the ages and such a point is reported lost without being tracked (see lk_kernel_sys)
  if (valid && lk.skip_age && lk.skip_age[(size_t)s * P.kcap + lk.src_idx[po]] > P.max_age) {
    if (l8 == 0) lk.status[po] = 0;
    valid = false;
  }
  if (!__any(valid)) return;

  unsigned char* const reg = lds + slot * C::RB;
  unsigned char* const patch = reg;
  int* const dxy = reinterpret_cast<int*>(reg + C::PATCH_B);

  const unsigned char* pimg = prev_img + (size_t)s * prev_img_stride;
  const unsigned char* cimg = cur_img + (size_t)s * cur_img_stride;
  const unsigned char* ppyr = prev_pyr + (size_t)s * P.pyr_stride;
  const unsigned char* cpyr = cur_pyr + (size_t)s * P.pyr_stride;

  const float2 prevPt0 = lk.prev_pts[po];
  float2 nextOut = lk.next_pts[po];   // initial flow
  int status = 1;
  const int maxLevel = P.nlevels - 1;
  const float FLT_SCALE = 1.f / (1 << 20);
  const float halfWin = (WIN - 1) * 0.5f;
  const int klt_iters = P.klt_iters;
  const double klt_eps2 = P.klt_eps2;
  // chain type of this lane: h = 0 keeps Ix (b1, A11), h = 1 keeps Iy (b2, A22); the other component is handed over
  const unsigned selK = h ? 0x07060302u : 0x05040100u, selS = h ? 0x05040100u : 0x07060302u;

  LK8P_DECL;
  for (int level = maxLevel; level >= 0; level--) {
    LK8P(6);
    const float2 entryOut = nextOut;   // (the state a deferred point restarts this level from)
    const LevelImg LI = level_img(P, pimg, prev_row_stride, ppyr, level);
    const LevelImg LJ = level_img(P, cimg, cur_row_stride, cpyr, level);
    const float lscale = (float)(1. / (1 << level));
    float2 prevPt = make_float2(prevPt0.x * lscale, prevPt0.y * lscale);
    float2 nextPt;
    if (level == maxLevel)
      nextPt = make_float2(nextOut.x * lscale, nextOut.y * lscale);
    else
      nextPt = make_float2(nextOut.x * 2.f, nextOut.y * 2.f);
    nextOut = nextPt;

    prevPt.x -= halfWin;
    prevPt.y -= halfWin;
    int ipx = (int)floorf(prevPt.x), ipy = (int)floorf(prevPt.y);
    bool lvl = valid;   // this point runs this level
    if (lvl && (ipx < -WIN || ipx >= LI.w || ipy < -WIN || ipy >= LI.h)) {
      if (level == 0) status = 0;
      lvl = false;
    }
    if (!lvl) ipx = ipy = 0;
    int wq0, wq1;
    {
      const float a = prevPt.x - ipx, b = prevPt.y - ipy;
      int iw00, iw01, iw10, iw11;
      lk_weights(a, b, &iw00, &iw01, &iw10, &iw11);
      wq0 = pack_lo16(iw00, iw01);
      wq1 = pack_lo16(iw10, iw11);
    }
    // Column group l8 of the Scharr pass: derivative positions x = 4 l8 .. 4 l8 + 3, source byte columns X0 .. X0 + 5.
    // A group whose eight bytes lie inside the image row reads two dwords per row; a group on the left / right border
    // reads its six bytes one by one at BORDER_REFLECT_101 columns.  Rows are reflected per row.  The derivative is zero
    // at positions outside the image (BORDER_CONSTANT): column masks per lane, row mask per row.
    const int X0 = ipx - 1 + 4 * l8;
    const bool edgeI = X0 < 0 || X0 + (l8 < NT - 1 ? 8 : 4) > LI.w;
    int ecol[6];
#pragma unroll
    for (int k = 0; k < 6; k++) ecol[k] = refl1(X0 + k, LI.w);
    int cmask[4];
#pragma unroll
    for (int k = 0; k < 4; k++) cmask[k] = (unsigned)(ipx + 4 * l8 + k) < (unsigned)LI.w ? -1 : 0;

    // template (folded: 2^8 - (I << 9), see lk_kernel_sys) of this lane's pixels and the gradient of this lane's type
    // for all pixels of the class, as (column 8cc + g, column 8cc + 4 + g) pairs per row and chunk
    int rA[NM][2][NC];
    int G[2 * NM][2][NCH];
    float accKK = 0.f, acc12 = 0.f;

#pragma unroll
    for (int hf = 0; hf < C::NHALF; hf++) {
      __syncthreads();   // the readers of the region (previous pass / level) are done
      LK8P(hf == 0 ? 0 : 2);
      if (lvl && l8 < NT) {
        // walked down the rows of the pass: every source row is loaded once, the middle row of a 3-row window doubles
        // as the byte patch of the template gather
        const int hoff = l8 < NT - 1 ? 4 : 0;   // (the last group holds one output: its second dword repeats the first)
        int lo[PROWS], hi[PROWS];
        unsigned ro[PROWS];   // byte offset of the row in the level (32 bit: the loads take the uniform base as scalar)
#pragma unroll
        for (int pr = 0; pr < PROWS; pr++) ro[pr] = (unsigned)refl1(ipy - 1 + HR * hf + pr, LI.h) * (unsigned)LI.stride;
        if (!edgeI) {
#pragma unroll
          for (int pr = 0; pr < PROWS; pr++) {
            lo[pr] = *reinterpret_cast<const int_u*>(LI.p + (ro[pr] + (unsigned)X0));
            hi[pr] = *reinterpret_cast<const int_u*>(LI.p + (ro[pr] + (unsigned)(X0 + hoff)));
          }
        } else {
#pragma unroll
          for (int pr = 0; pr < PROWS; pr++) {
            const unsigned char* r = LI.p + ro[pr];
            lo[pr] = (int)r[ecol[0]] | ((int)r[ecol[1]] << 8) | ((int)r[ecol[2]] << 16) | ((int)r[ecol[3]] << 24);
            hi[pr] = (int)r[ecol[4]] | ((int)r[ecol[5]] << 8);
          }
        }
        __builtin_amdgcn_sched_barrier(0);   // (all loads of the pass in flight before the first is consumed)
        const v2us k3 = {3, 3}, k10 = {10, 10};
        v2us a01, a23, a45, b01, b23, b45;   // the two rows above the one being unpacked
#pragma unroll
        for (int pr = 0; pr < PROWS; pr++) {
          *reinterpret_cast<int*>(patch + pr * PSTR + 4 * l8) = lo[pr];
          const v2us c01 = as_v2us(perm_b32(0, lo[pr], 0x0c010c00u)), c23 = as_v2us(perm_b32(0, lo[pr], 0x0c030c02u)),
                     c45 = as_v2us(perm_b32(0, hi[pr], 0x0c010c00u));
          if (pr >= 2) {
            const int y = pr - 2;
            const v2us s01 = (a01 + c01) * k3 + b01 * k10, s23 = (a23 + c23) * k3 + b23 * k10,
                       s45 = (a45 + c45) * k3 + b45 * k10;
            const v2us d01 = c01 - a01, d23 = c23 - a23, d45 = c45 - a45;
            const v2us vx12 = s23 - s01, vx34 = s45 - s23;
            const v2us m12 = as_v2us(perm_b32(as_i32(d23), as_i32(d01), 0x05040302u));
            const v2us m34 = as_v2us(perm_b32(as_i32(d45), as_i32(d23), 0x05040302u));
            const v2us vy12 = (d01 + d23) * k3 + m12 * k10, vy34 = (d23 + d45) * k3 + m34 * k10;
            const int rmask = (unsigned)(ipy + HR * hf + y) < (unsigned)LI.h ? -1 : 0;
            int* o = dxy + y * DSTR + 4 * l8;
            o[0] = pack_lo16(as_i32(vx12), as_i32(vy12)) & (cmask[0] & rmask);
            o[1] = pack_hi16(as_i32(vx12), as_i32(vy12)) & (cmask[1] & rmask);
            o[2] = pack_lo16(as_i32(vx34), as_i32(vy34)) & (cmask[2] & rmask);
            o[3] = pack_hi16(as_i32(vx34), as_i32(vy34)) & (cmask[3] & rmask);
          }
          a01 = b01; a23 = b23; a45 = b45;
          b01 = c01; b23 = c23; b45 = c45;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();
      LK8P(1);

      // bilinear template and gradient window of this lane's pixels of the pass; gradient hand-over; A chains
#pragma unroll
      for (int ml = 0; ml < MH; ml++) {
        const int m = hf * MH + ml;
        const int yl = 4 * ml + 2 * h;   // first local row of the group (lane-dependent: in the base address)
        int pp[3][NC], kK[3][NC], kS[3][NC];
#pragma unroll
        for (int rr = 0; rr < 3; rr++)
#pragma unroll
          for (int c = 0; c < NC; c++) {
            const unsigned char* s0 = patch + (yl + 1 + rr) * PSTR + (g + 4 * c + 1);
            pp[rr][c] = (int)s0[0] | ((int)s0[1] << 16);
            const int* d = dxy + (yl + rr) * DSTR + (g + 4 * c);
            const int d0 = d[0], d1 = d[1];
            kK[rr][c] = perm_b32(d1, d0, selK);
            kS[rr][c] = perm_b32(d1, d0, selS);
          }
        int vK[2][NC], vS[2][NC];
        float p12[2][NC];
        if constexpr (NC == 6) {
#pragma unroll
          for (int r = 0; r < 2; r++) {
            int iv[6];
            lk8_blend_row6<9>(pp[r], pp[r + 1], wq0, wq1, 1 << 8, iv);
#pragma unroll
            for (int c = 0; c < NC; c++) rA[m][r][c] = (1 << 8) - (iv[c] << 9);
            lk8_blend_row6<14>(kK[r], kK[r + 1], wq0, wq1, 1 << 13, vK[r]);
            lk8_blend_row6<14>(kS[r], kS[r + 1], wq0, wq1, 1 << 13, vS[r]);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 2; r++)
#pragma unroll
            for (int c = 0; c < NC; c++) {
              const int ival = dot2_i16(pp[r][c], wq0, dot2_i16(pp[r + 1][c], wq1, 1 << 8)) >> 9;
              rA[m][r][c] = (1 << 8) - (ival << 9);
              vK[r][c] = dot2_i16(kK[r][c], wq0, dot2_i16(kK[r + 1][c], wq1, 1 << 13)) >> 14;
              vS[r][c] = dot2_i16(kS[r][c], wq0, dot2_i16(kS[r + 1][c], wq1, 1 << 13)) >> 14;
            }
        }
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
          for (int c = 0; c < NC; c++) p12[r][c] = (float)vK[r][c] * (float)vS[r][c];
        // slot 2m = rows {4m, 4m + 1} (owned by the pair's lane h = 0), slot 2m + 1 = rows {4m + 2, 4m + 3} (lane h = 1)
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
          for (int cc = 0; cc < NCH; cc++) {
            const int gk = pack_lo16(vK[r][2 * cc], vK[r][2 * cc + 1]);
            const int gs = pack_lo16(vS[r][2 * cc], vS[r][2 * cc + 1]);
            const int rv = pair_swap(gs);   // the partner's pixels, this lane's type
            G[2 * m][r][cc] = h ? rv : gk;
            G[2 * m + 1][r][cc] = h ? gk : rv;
          }
        // A12: both lanes walk it from broadcast products (row order: slot 2m, then slot 2m + 1)
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
          for (int c = 0; c < NC; c++) acc12 = acc12 + pair_bcast_f<0>(p12[r][c]);
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
          for (int c = 0; c < NC; c++) acc12 = acc12 + pair_bcast_f<1>(p12[r][c]);
        // A11 (h = 0) / A22 (h = 1)
#pragma unroll
        for (int sp = 0; sp < 2; sp++)
#pragma unroll
          for (int r = 0; r < 2; r++)
#pragma unroll
            for (int cc = 0; cc < NCH; cc++) {
              const int gg = G[2 * m + sp][r][cc];
              const float f0 = (float)(short)(gg & 0xffff), f1 = (float)(gg >> 16);
              accKK = accKK + f0 * f0;
              accKK = accKK + f1 * f1;
            }
      }
    }

    LK8P(2);
    float A11, A12, A22;
    {
      const float k0 = __shfl(accKK, pbase + 0), k1 = __shfl(accKK, pbase + 2), k2 = __shfl(accKK, pbase + 4),
                  k3 = __shfl(accKK, pbase + 6);
      const float q0 = __shfl(accKK, pbase + 1), q1 = __shfl(accKK, pbase + 3), q2 = __shfl(accKK, pbase + 5),
                  q3 = __shfl(accKK, pbase + 7);
      const float x0 = __shfl(acc12, pbase + 0), x1 = __shfl(acc12, pbase + 2), x2 = __shfl(acc12, pbase + 4),
                  x3 = __shfl(acc12, pbase + 6);
      float iA11 = 0.f, iA12 = 0.f, iA22 = 0.f;
      iA11 += k0 + k1 + k2 + k3;
      iA12 += x0 + x1 + x2 + x3;
      iA22 += q0 + q1 + q2 + q3;
      A11 = iA11 * FLT_SCALE;
      A12 = iA12 * FLT_SCALE;
      A22 = iA22 * FLT_SCALE;
    }
    float D = A11 * A22 - A12 * A12;
    const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * WIN * WIN);
    if (lvl && (minEig < 1e-4f || D < 1.1920929e-07f)) {
      if (level == 0) status = 0;
      lvl = false;
    }
    D = 1.f / D;

    nextPt.x -= halfWin;
    nextPt.y -= halfWin;
    float2 prevDelta = make_float2(0.f, 0.f);
    int jx0 = 0, jy0 = 0;
    bool jvalid = false;
    bool active = lvl;
    bool may_defer = true;
    int jboff = 2 * h * JSTRB + 2 * g;   // (a point that never staged a window reads its region's garbage: never used)
    LK8P(3);
    for (int j = 0; j < klt_iters; j++) {
      int inx = 0, iny = 0;
      bool need = false;
      if (active) {
        inx = (int)floorf(nextPt.x);
        iny = (int)floorf(nextPt.y);
        if (inx < -WIN || inx >= LJ.w || iny < -WIN || iny >= LJ.h) {
          if (level == 0) status = 0;
          active = false;
        } else {
          const float a = nextPt.x - inx, b = nextPt.y - iny;
          int iw00, iw01, iw10, iw11;
          lk_weights(a, b, &iw00, &iw01, &iw10, &iw11);
          wq0 = pack_lo16(iw00, iw01);
          wq1 = pack_lo16(iw10, iw11);
          need = !jvalid || inx < jx0 || iny < jy0 || inx + WIN + 1 > jx0 + JS || iny + WIN + 1 > jy0 + JS;
        }
      }
      if (!__any(active)) break;
      if (__any(need)) {
        LK8P(5);
        LK8P_COUNT(p8_stages);
        // (re)stage the current-level window of the points that left theirs, as (pixel | next pixel << 8) byte pairs
        __syncthreads();
        if (need) {
          jx0 = inx - JM;
          jy0 = iny - JM;
          jvalid = true;
          if (l8 < NJ) {
            // column group l8: four pairs of a row out of five bytes (the last group: two pairs out of three), rows and,
            // for a group on the left / right border, byte columns at BORDER_REFLECT_101 positions
            const int XJ = jx0 + 4 * l8;
            const int hoff = l8 < NJ - 1 ? 4 : 0;
            constexpr int CH = 8;   // rows per batch of loads
            if (XJ >= 0 && XJ + (l8 < NJ - 1 ? 8 : 4) <= LJ.w) {
#pragma unroll
              for (int y0 = 0; y0 < JS; y0 += CH) {
                int lo[CH], hi[CH];
#pragma unroll
                for (int k = 0; k < CH; k++)
                  if (y0 + k < JS) {
                    const unsigned ro = (unsigned)refl1(jy0 + y0 + k, LJ.h) * (unsigned)LJ.stride + (unsigned)XJ;
                    lo[k] = *reinterpret_cast<const int_u*>(LJ.p + ro);
                    hi[k] = *reinterpret_cast<const int_u*>(LJ.p + (ro + (unsigned)hoff));
                  }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < CH; k++)
                  if (y0 + k < JS) {
                    int* o = reinterpret_cast<int*>(reg + (y0 + k) * JSTRB + 8 * l8);
                    o[0] = perm_b32(hi[k], lo[k], 0x02010100u);   // (b0 b1)(b1 b2)
                    o[1] = perm_b32(hi[k], lo[k], 0x04030302u);   // (b2 b3)(b3 b4)
                  }
                __builtin_amdgcn_sched_barrier(0);
              }
            } else {
              int jc[5];
#pragma unroll
              for (int k = 0; k < 5; k++) jc[k] = refl1(XJ + k, LJ.w);
#pragma unroll
              for (int y0 = 0; y0 < JS; y0 += CH) {
                int lo[CH], hi[CH];
#pragma unroll
                for (int k = 0; k < CH; k++)
                  if (y0 + k < JS) {
                    const unsigned char* r = LJ.p + (unsigned)refl1(jy0 + y0 + k, LJ.h) * (unsigned)LJ.stride;
                    lo[k] = (int)r[jc[0]] | ((int)r[jc[1]] << 8) | ((int)r[jc[2]] << 16) | ((int)r[jc[3]] << 24);
                    hi[k] = (int)r[jc[4]];
                  }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < CH; k++)
                  if (y0 + k < JS) {
                    int* o = reinterpret_cast<int*>(reg + (y0 + k) * JSTRB + 8 * l8);
                    o[0] = perm_b32(hi[k], lo[k], 0x02010100u);
                    o[1] = perm_b32(hi[k], lo[k], 0x04030302u);
                  }
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
        }
        __syncthreads();
        LK8P(4);
      }
      LK8P_COUNT(p8_iters);
      if (active) jboff = (iny - jy0 + 2 * h) * JSTRB + 2 * (inx - jx0 + g);
      const unsigned char* jb = reg + jboff;

      // this lane's pixels: blend of the current frame minus the template; packed differences handed to both lanes of
      // the pair in row order; the lane's chain: terms (d(x) G(x) + d(x+4) G(x+4)) as float, added in OpenCV's order
      float acc = 0.f;
#pragma unroll
      for (int m = 0; m < NM; m++) {
        int E[3][NC];
#pragma unroll
        for (int rr = 0; rr < 3; rr++)
#pragma unroll
          for (int c = 0; c < NC; c++) {
            const int pr = *reinterpret_cast<const unsigned short*>(jb + (4 * m + rr) * JSTRB + 8 * c);
            E[rr][c] = perm_b32(0, pr, 0x0c010c00u);
          }
        int ddo[2][NCH];
        if constexpr (NC == 6) {
          lk8_diff_row6(E[0], E[1], rA[m][0], wq0, wq1, ddo[0]);
          lk8_diff_row6(E[1], E[2], rA[m][1], wq0, wq1, ddo[1]);
          int de[6], dq[6];
          float f[6];
#pragma unroll
          for (int k = 0; k < 6; k++) de[k] = pair_bcast<0>(ddo[k / 3][k % 3]);
          lk8_terms6(de, reinterpret_cast<const int(&)[6]>(G[2 * m][0][0]), f);
#pragma unroll
          for (int k = 0; k < 6; k++) acc = acc + f[k];
#pragma unroll
          for (int k = 0; k < 6; k++) dq[k] = pair_bcast<1>(ddo[k / 3][k % 3]);
          lk8_terms6(dq, reinterpret_cast<const int(&)[6]>(G[2 * m + 1][0][0]), f);
#pragma unroll
          for (int k = 0; k < 6; k++) acc = acc + f[k];
        } else {
#pragma unroll
          for (int r = 0; r < 2; r++)
#pragma unroll
            for (int cc = 0; cc < NCH; cc++) {
              const int d0 = dot2_i16(E[r][2 * cc], wq0, dot2_i16(E[r + 1][2 * cc], wq1, rA[m][r][2 * cc])) >> 9;
              const int d1 = dot2_i16(E[r][2 * cc + 1], wq0, dot2_i16(E[r + 1][2 * cc + 1], wq1, rA[m][r][2 * cc + 1])) >> 9;
              ddo[r][cc] = pack_lo16(d0, d1);
            }
#pragma unroll
          for (int r = 0; r < 2; r++)
#pragma unroll
            for (int cc = 0; cc < NCH; cc++)
              acc = acc + (float)dot2_i16(pair_bcast<0>(ddo[r][cc]), G[2 * m][r][cc], 0);
#pragma unroll
          for (int r = 0; r < 2; r++)
#pragma unroll
            for (int cc = 0; cc < NCH; cc++)
              acc = acc + (float)dot2_i16(pair_bcast<1>(ddo[r][cc]), G[2 * m + 1][r][cc], 0);
        }
      }
      // bbuf = qb0 + qb1 ; ib1 += bbuf[0] + bbuf[2] ; ib2 += bbuf[1] + bbuf[3]   (lane 2g + h holds class g, type h)
      const float t0 = __shfl(acc, pbase + 0), t1 = __shfl(acc, pbase + 1), t2 = __shfl(acc, pbase + 2),
                  t3 = __shfl(acc, pbase + 3), t4 = __shfl(acc, pbase + 4), t5 = __shfl(acc, pbase + 5),
                  t6 = __shfl(acc, pbase + 6), t7 = __shfl(acc, pbase + 7);
      if (active) {
        const float bb0 = t0 + t4, bb1 = t1 + t5, bb2 = t2 + t6, bb3 = t3 + t7;
        float ib1 = 0.f, ib2 = 0.f;
        ib1 += bb0 + bb2;
        ib2 += bb1 + bb3;
        const float b1 = ib1 * FLT_SCALE, b2 = ib2 * FLT_SCALE;
        const float2 delta = make_float2((float)((A12 * b2 - A22 * b1) * D), (float)((A12 * b1 - A11 * b2) * D));
        nextPt.x += delta.x;
        nextPt.y += delta.y;
        nextOut = make_float2(nextPt.x + halfWin, nextPt.y + halfWin);
        if ((double)delta.x * (double)delta.x + (double)delta.y * (double)delta.y <= klt_eps2) {
          active = false;
        } else if (j > 0 && fabs((double)(delta.x + prevDelta.x)) < 0.01 && fabs((double)(delta.y + prevDelta.y)) < 0.01) {
          nextOut.x -= delta.x * 0.5f;
          nextOut.y -= delta.y * 0.5f;
          active = false;
        }
        prevDelta = delta;
        // still iterating after iter_cap iterations of this level: the other points of the wave would wait for this one in
        // lock step -- hand it to a one-point wave (lk_kernel_sys, deferred pass), which redoes the level from its start
        if (active && may_defer && j + 1 >= iter_cap && j + 1 < klt_iters) {
          int di = 0;
          if (l8 == 0) di = atomicAdd(&lk.defer_cnt[s], 1);
          di = __shfl(di, pbase);
          if (di < lk.defer_cap) {
            if (l8 == 0) {
              lk.defer_pt[(size_t)s * lk.defer_cap + di] = pt | (level << 24);
              lk.next_pts[po] = entryOut;
            }
            valid = false;
            active = false;
          } else {
            may_defer = false;   // (list full: the point stays)
          }
        }
      }
    }

    LK8P(5);
    if (valid && status && level == 0) {
      // (calcOpticalFlowPyrLK clears the status of a point whose final window left the image whenever an error array is
      // passed, and the reference passes one, Tracker.cpp:137-139; the error itself is not computed here)
      const float2 np = make_float2(nextOut.x - halfWin, nextOut.y - halfWin);
      const int inx = (int)floorf(np.x), iny = (int)floorf(np.y);
      if (inx < -WIN || inx >= LJ.w || iny < -WIN || iny >= LJ.h) status = 0;
    }
  }
#ifdef KVFE_LK8_PROF
  if (lane == 0) {
    for (int i = 0; i < 8; i++) atomicAdd(&kvfe_lk8_prof[i], p8_acc[i]);
    atomicAdd(&kvfe_lk8_prof[8], 1ull);
    atomicAdd(&kvfe_lk8_prof[9], (unsigned long long)p8_iters);
    atomicAdd(&kvfe_lk8_prof[10], (unsigned long long)p8_stages);
    const unsigned wid = blockIdx.y * gridDim.x + blockIdx.x;
    if (wid < LK8P_WAVES) {
      kvfe_lk8_life[2 * wid] = p8_t0;
      kvfe_lk8_life[2 * wid + 1] = __builtin_amdgcn_s_memrealtime() - p8_t0;
    }
  }
#endif
  if (valid && l8 == 0) {
    lk.next_pts[po] = nextOut;
    lk.status[po] = (unsigned char)status;
    lk.err[po] = 0.f;
  }
}

bool launch_lk8(const KParams& P, const unsigned char* prev_img, size_t prev_row_stride, size_t prev_img_stride,
                const unsigned char* prev_pyr, const unsigned char* cur_img, size_t cur_row_stride,
                size_t cur_img_stride, const unsigned char* cur_pyr, const LkScratch& lk, int max_pts, hipStream_t st,
                int iter_cap) {
  for (int l = 0; l < P.nlevels; l++)
    if (P.lw[l] < LK8_MIN_DIM || P.lh[l] < LK8_MIN_DIM) return false;   // (refl1: one fold)
  const dim3 grid((max_pts + 7) / 8, P.B), block(64);
#ifdef KVFE_LK8_PROF
  {
    static bool reg = false;
    if (!reg) {
      reg = true;
      std::atexit([] {
        unsigned long long h[LK8P_N];
        static unsigned long long life[LK8P_WAVES * 2];
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(h, HIP_SYMBOL(kvfe_lk8_prof), sizeof(h));
        hipMemcpyFromSymbol(life, HIP_SYMBOL(kvfe_lk8_life), sizeof(life));
        const double n = h[8] ? (double)h[8] : 1.0;
        std::fprintf(stderr, "KVFE_LK8_PROF waves %.0f  cycles per wave: header %.0f  column walk %.0f  gather+A %.0f  A sums %.0f  "
                     "staging %.0f  iterations %.0f  between levels %.0f | wave-iterations %.2f  staging passes %.2f\n",
                     n, h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[5] / n, h[6] / n, h[9] / n, h[10] / n);
        std::vector<double> st, lf;
        double t0 = 1e300;
        for (int w = 0; w < LK8P_WAVES; w++)
          if (life[2 * w + 1]) { st.push_back((double)life[2 * w]); lf.push_back((double)life[2 * w + 1]); t0 = std::min(t0, st.back()); }
        if (!lf.empty()) {
          double tend = 0, tls = 0;
          for (size_t i = 0; i < lf.size(); i++) { tend = std::max(tend, st[i] - t0 + lf[i]); tls = std::max(tls, st[i] - t0); }
          std::vector<double> q = lf;
          std::sort(q.begin(), q.end());
          std::fprintf(stderr, "KVFE_LK8_PROF last launch: %zu waves, ends at %.0f (10 ns ticks), last wave starts at %.0f; lifetimes p10 %.0f "
                       "p50 %.0f p90 %.0f p99 %.0f max %.0f\n", lf.size(), tend, tls, q[q.size() / 10], q[q.size() / 2],
                       q[q.size() * 9 / 10], q[q.size() * 99 / 100], q.back());
          std::fprintf(stderr, "KVFE_LK8_PROF resident waves along the launch:");
          for (int k = 0; k < 20; k++) {
            const double tm = tend * (k + 0.5) / 20;
            int res = 0;
            for (size_t i = 0; i < lf.size(); i++) res += (st[i] - t0 <= tm && tm < st[i] - t0 + lf[i]);
            std::fprintf(stderr, " %d", res);
          }
          std::fprintf(stderr, "\n");
        }
      });
    }
  }
#endif
  switch (P.klt_win) {
    case 24:
      hipLaunchKernelGGL(lk8_kernel<24>, grid, block, 0, st, P, prev_img, prev_row_stride, prev_img_stride, prev_pyr,
                         cur_img, cur_row_stride, cur_img_stride, cur_pyr, lk, iter_cap);
      return true;
    default:
      return false;
  }
}

}  // namespace kvfe
