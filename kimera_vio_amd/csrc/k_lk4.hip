// K4c (round 6)  pyramidal Lucas-Kanade, FOUR points per wavefront
//      == cv::calcOpticalFlowPyrLK (OPTFLOW_USE_INITIAL_FLOW) as called by Tracker::featureTracking,
//         /root/reference/src/frontend/Tracker.cpp:137-146
//
// k_track.hip's lk_kernel_sys gives a point a whole wavefront: OpenCV's SSE accumulators are strictly sequential float
// chains (4 lane classes x & 3, times {b1, b2} per iteration and {A11, A12, A22} per level), so a wave that owns ONE point
// walks them as a 12-stage systolic pipeline with 4 useful lanes per stage -- 1 560 of its 4 870 vector instructions per
// point.  Here a DPP row of 16 lanes owns a point and a QUAD of lanes owns a lane class:
//
//   lane = 16 * slot + 4 * g + q      slot = point of the wave, g = SSE lane class (window column & 3), q = row owner
//
// * pixels: lane q of a quad owns the window rows {8j + 2q, 8j + 2q + 1} of the class's six columns {g + 4c}: 36 pixels at
//   WIN = 24.  Template values (folded into the blend accumulator, as in lk_kernel_sys) and both gradient components of the
//   lane's pixels stay in registers for the level: 72 registers.
// * terms: the owner computes the chain terms of its pixels (d(x) Ix(x) + d(x+4) Ix(x+4) as float, same for Iy).
// * chains: `v_add_f32_dpp acc, term, acc quad_perm:[o,o,o,o]` adds owner o's term in ALL four lanes of the quad, so every
//   lane of a quad walks both chains of its class in OpenCV's order (rows 0, 1 from owner 0, rows 2, 3 from owner 1, ...):
//   one instruction per chain addition, no hand-over, no select; the two chains are independent and interleave.
//   A11 / A12 / A22 the same way once per level.  The four classes are combined by two DPP row rotations.
// * the sixteen lanes of a point are a SIMD machine of their own for the per-level set-up: lane (t, half) walks column group
//   t of the Scharr pass down one half of the rows (every source row loaded once, all loads of the pass in flight
//   together), and stages column group t of one half of the current-frame window.  Borders: rows and, for a group that
//   touches the left / right border, byte columns at BORDER_REFLECT_101 positions (branch-free single fold); the
//   derivative is masked to zero outside the image (BORDER_CONSTANT).  Per-point scalars are computed redundantly.
// * LDS: 3.3 KB per point (byte patch + derivative patch of a level; the current-frame window is staged over it as 16-bit
//   (pixel, pixel + 1) byte pairs), 13 KB per wave; about 130 registers: three waves per SIMD.
// * the four points of a wave iterate in lock step; a point that has converged (or left the image) is masked.
//
// Every integer is exact and every float chain has the order of lk_kernel_sys (= OpenCV's SSE2 build), so the results are
// bit-identical to it and to the oracle (tests/test_gpu_parity.py, tests/test_gpu_bench_configs.py, tools/fuzz_frontend.py).
// The error output of calcOpticalFlowPyrLK is not computed here (the front-end step drops it): launch_lk keeps
// lk_kernel_sys for callers that want it, for windows other than 24, for pyramids with a level narrower than LK4_MIN_DIM
// and behind kvfe_config.lk_impl = 1 / KVFE_LK_IMPL=1.
#include "kvfe_dev.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace kvfe {

#include "kvfe_lk.inl"

// BORDER_REFLECT_101 index for -len < p < 2 len - 1 (one fold), branch-free so that the loads of a pass stay in flight
// together: min(|p|, 2 len - 2 - |p|).  The indices this kernel forms lie in [-(WIN + 3), len + WIN + 4] (a window may start
// WIN pixels outside the image, calcOpticalFlowPyrLK's own bound, plus the staging margin of 3 and the fifth byte of the
// last column group): one fold covers them from len = WIN + 6 = 30 on.  launch_lk4 refuses pyramids with a smaller level
// (the shipped five-level pyramid of 752 x 480 ends at 47 x 30).
constexpr int LK4_MIN_DIM = 30;
__device__ __forceinline__ int refl1(int p, int len) {
  const int q = max(p, -p);
  return min(q, 2 * len - 2 - q);
}

template <int WIN>
struct Lk4 {
  static constexpr int NC = WIN / 4;         // window columns of one class
  static constexpr int NCH = WIN / 8;        // 8-pixel chunks of a row = chain terms per row, class and component
  static constexpr int NG = WIN / 8;         // row-pair groups of a lane (rows 8j + 2q + {0, 1})
  static constexpr int WS = WIN + 3, WP = WIN + 1;
  static constexpr int NT = (WP + 3) / 4;    // column groups of the Scharr pass (4 outputs each)
  static constexpr int PSTR = 4 * NT;        // byte patch row stride
  static constexpr int DSTR = WP;            // derivative patch row stride in dwords (odd)
  static constexpr int PATCH_B = WS * PSTR;
  static constexpr int DXY_B = WP * DSTR * 4;
  static constexpr int HROWS = (WP + 1) / 2;   // derivative rows per half of the lanes (the middle row is written twice)
  static constexpr int PROWS = HROWS + 2;      // source rows a half walks
  static constexpr int JM = 3;               // margin of the staged current-frame window
  static constexpr int JS = WIN + 1 + 2 * JM;   // staged rows
  static constexpr int JW = JS - 1;             // staged byte pairs per row
  static constexpr int NJ = (JW + 3) / 4;       // column groups of the staging pass (4 pairs each)
  static constexpr int JSTRB = 4 * (2 * NJ + 1);   // row stride in bytes: an odd number of dwords
  static constexpr int J_B = JS * JSTRB;
  static constexpr int JH = (JS + 1) / 2;       // staged rows per half of the lanes
  static constexpr int NEED_B = (PATCH_B + DXY_B > J_B ? PATCH_B + DXY_B : J_B);
  // region stride: consecutive points 8 banks apart
  static constexpr int RB = ((NEED_B + 127) / 128) * 128 + 32;
  static_assert(WP == 4 * (NT - 1) + 1, "last derivative group holds exactly one output");
  static_assert(JW == 4 * (NJ - 1) + 2, "last staging group holds exactly two pairs");
  static_assert(NT <= 8 && NJ <= 8, "one column group per lane of a half");
  static_assert(PATCH_B % 4 == 0 && (WP & 1), "derivative patch starts on a dword; odd row stride");
  static_assert(NC == 6, "the hand-scheduled blocks are six pixels wide");
};

// -DKVFE_LK4_PROF (tools/r6/gpu_lk4_prof.sh; never in the product build): cycle stamps per phase summed over the waves of
// all launches, printed at exit
#ifdef KVFE_LK4_PROF
constexpr int LK4P_N = 12;
__device__ unsigned long long kvfe_lk4_prof[LK4P_N];
#define LK4P_DECL unsigned long long p4_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, p4_last = __builtin_readcyclecounter(); \
  unsigned p4_iters = 0, p4_stages = 0
#define LK4P(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); p4_acc[i] += t_ - p4_last; p4_last = t_; } while (0)
#define LK4P_COUNT(v) (v)++
#else
#define LK4P_DECL do { } while (0)
#define LK4P(i) do { } while (0)
#define LK4P_COUNT(v) do { } while (0)
#endif

// value of the quad's lane O in all four lanes of the quad
template <int O>
__device__ __forceinline__ float quad_bcast_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), O * 0x55, 0xf, 0xf, true));
}
// rotation of the 16-lane DPP row by N lanes (the classes of a point are its four quads)
template <int N>
__device__ __forceinline__ float row_ror_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, true));
}

// ---- hand-scheduled blocks (window of 24: six columns per class).  hipcc only emits the accumulate-in-place dot product
// (v_dot2c: a v_mov per start value) and pads every dot product -> other-VALU dependency with s_nop; in these blocks the
// three-operand form is issued six wide, so that every consumer sits at least three instructions behind its producer (the
// wait states a non-dot reader of a dot product needs), and the blocks end with the wait states a DPP / SDWA-sensitive
// reader outside needs (the compiler cannot see into inline asm; tools/check_dpp_hazard.py scans the ISA).
// One window row of a lane (six pixels): blend of the current frame minus the folded template, >> 9, packed in pairs
// (v_ashrrev ... dst_sel:WORD_1 writes the odd pixel's difference into the high half of the even one's register).
__device__ __forceinline__ void lk4_diff_row6(const int (&Ea)[6], const int (&Eb)[6], const int (&ra)[6], int wq0, int wq1,
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
      "s_nop 0"
      : "=&v"(dd[0]), "=&v"(dd[1]), "=&v"(dd[2]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5)
      : "v"(Ea[0]), "v"(Ea[1]), "v"(Ea[2]), "v"(Ea[3]), "v"(Ea[4]), "v"(Ea[5]), "v"(Eb[0]), "v"(Eb[1]), "v"(Eb[2]),
        "v"(Eb[3]), "v"(Eb[4]), "v"(Eb[5]), "v"(ra[0]), "v"(ra[1]), "v"(ra[2]), "v"(ra[3]), "v"(ra[4]), "v"(ra[5]),
        "v"(wq0), "v"(wq1));
}
// the chain terms of two window rows (three packed difference pairs each): (d(x) G(x) + d(x + 4) G(x + 4)) as float, for
// both gradient components; ends with the two wait states the DPP additions behind it need
__device__ __forceinline__ void lk4_terms12(const int (&d)[6], const int (&gx)[6], const int (&gy)[6], float (&fx)[6],
                                            float (&fy)[6]) {
  asm("v_dot2_i32_i16 %0, %12, %18, 0\n\t"
      "v_dot2_i32_i16 %1, %13, %19, 0\n\t"
      "v_dot2_i32_i16 %2, %14, %20, 0\n\t"
      "v_dot2_i32_i16 %3, %15, %21, 0\n\t"
      "v_dot2_i32_i16 %4, %16, %22, 0\n\t"
      "v_dot2_i32_i16 %5, %17, %23, 0\n\t"
      "v_dot2_i32_i16 %6, %12, %24, 0\n\t"
      "v_dot2_i32_i16 %7, %13, %25, 0\n\t"
      "v_dot2_i32_i16 %8, %14, %26, 0\n\t"
      "v_dot2_i32_i16 %9, %15, %27, 0\n\t"
      "v_dot2_i32_i16 %10, %16, %28, 0\n\t"
      "v_dot2_i32_i16 %11, %17, %29, 0\n\t"
      "v_cvt_f32_i32 %0, %0\n\t"
      "v_cvt_f32_i32 %1, %1\n\t"
      "v_cvt_f32_i32 %2, %2\n\t"
      "v_cvt_f32_i32 %3, %3\n\t"
      "v_cvt_f32_i32 %4, %4\n\t"
      "v_cvt_f32_i32 %5, %5\n\t"
      "v_cvt_f32_i32 %6, %6\n\t"
      "v_cvt_f32_i32 %7, %7\n\t"
      "v_cvt_f32_i32 %8, %8\n\t"
      "v_cvt_f32_i32 %9, %9\n\t"
      "v_cvt_f32_i32 %10, %10\n\t"
      "v_cvt_f32_i32 %11, %11\n\t"
      "s_nop 1"
      : "=&v"(fx[0]), "=&v"(fx[1]), "=&v"(fx[2]), "=&v"(fx[3]), "=&v"(fx[4]), "=&v"(fx[5]), "=&v"(fy[0]), "=&v"(fy[1]),
        "=&v"(fy[2]), "=&v"(fy[3]), "=&v"(fy[4]), "=&v"(fy[5])
      : "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(gx[0]), "v"(gx[1]), "v"(gx[2]), "v"(gx[3]),
        "v"(gx[4]), "v"(gx[5]), "v"(gy[0]), "v"(gy[1]), "v"(gy[2]), "v"(gy[3]), "v"(gy[4]), "v"(gy[5]));
}
// one row of six bilinear blends (a0 . w0 + a1 . w1 + c) >> SH of (value, right neighbour) pairs: the template and
// gradient gather of a level
template <int SH>
__device__ __forceinline__ void lk4_blend_row6(const int (&Ea)[6], const int (&Eb)[6], int wq0, int wq1, int c0,
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

// The chains.  `v_add_f32_dpp acc, term, acc quad_perm:[o,o,o,o]` adds owner o's term in all four lanes of the quad.  Written
// as asm because hipcc pads EVERY VALU write -> DPP-instruction read with two wait states, also when the freshly written
// register is the plain operand (the accumulator, src1) and not the permuted one (src0): one s_nop per addition of a
// two-chain interleave.  The hardware hazard is on the DPP source only (it is read ahead of the operand forwarding); the
// terms were written at least two instructions earlier (the producing blocks end in s_nop 1).
#define LK4_QP(o) "quad_perm:[" #o "," #o "," #o "," #o "] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define LK4_ADD2(o, a, b) "v_add_f32_dpp %0, %" #a ", %0 " LK4_QP(o) "v_add_f32_dpp %1, %" #b ", %1 " LK4_QP(o)
#define LK4_OWNER2(o) LK4_ADD2(o, 2, 8) LK4_ADD2(o, 3, 9) LK4_ADD2(o, 4, 10) LK4_ADD2(o, 5, 11) LK4_ADD2(o, 6, 12) LK4_ADD2(o, 7, 13)
#define LK4_ADD1(o, a) "v_add_f32_dpp %0, %" #a ", %0 " LK4_QP(o)
#define LK4_OWNER1(o, b) LK4_ADD1(o, b##1) LK4_ADD1(o, b##2) LK4_ADD1(o, b##3) LK4_ADD1(o, b##4) LK4_ADD1(o, b##5) LK4_ADD1(o, b##6)
// two independent chains, interleaved: the six terms ta / tb of every owner in owner (= row) order
__device__ __forceinline__ void chain_owners_x2(float& a, const float (&ta)[6], float& b, const float (&tb)[6]) {
  asm(LK4_OWNER2(0) LK4_OWNER2(1) LK4_OWNER2(2) LK4_OWNER2(3) "s_nop 0"
      : "+v"(a), "+v"(b)
      : "v"(ta[0]), "v"(ta[1]), "v"(ta[2]), "v"(ta[3]), "v"(ta[4]), "v"(ta[5]), "v"(tb[0]), "v"(tb[1]), "v"(tb[2]),
        "v"(tb[3]), "v"(tb[4]), "v"(tb[5]));
}
// two rows per owner (the A chains of a level: twelve terms of every owner), two chains
__device__ __forceinline__ void chain_owner_rows_x2(float& a, const float (&ta)[2][6], float& b, const float (&tb)[2][6]) {
#define LK4_O2R(o) LK4_ADD2(o, 2, 14) LK4_ADD2(o, 3, 15) LK4_ADD2(o, 4, 16) LK4_ADD2(o, 5, 17) LK4_ADD2(o, 6, 18) LK4_ADD2(o, 7, 19) \
                   LK4_ADD2(o, 8, 20) LK4_ADD2(o, 9, 21) LK4_ADD2(o, 10, 22) LK4_ADD2(o, 11, 23) LK4_ADD2(o, 12, 24) LK4_ADD2(o, 13, 25)
  asm(LK4_O2R(0) LK4_O2R(1) LK4_O2R(2) LK4_O2R(3) "s_nop 0"
      : "+v"(a), "+v"(b)
      : "v"(ta[0][0]), "v"(ta[0][1]), "v"(ta[0][2]), "v"(ta[0][3]), "v"(ta[0][4]), "v"(ta[0][5]), "v"(ta[1][0]), "v"(ta[1][1]),
        "v"(ta[1][2]), "v"(ta[1][3]), "v"(ta[1][4]), "v"(ta[1][5]), "v"(tb[0][0]), "v"(tb[0][1]), "v"(tb[0][2]), "v"(tb[0][3]),
        "v"(tb[0][4]), "v"(tb[0][5]), "v"(tb[1][0]), "v"(tb[1][1]), "v"(tb[1][2]), "v"(tb[1][3]), "v"(tb[1][4]), "v"(tb[1][5]));
#undef LK4_O2R
}
// ... and one chain
__device__ __forceinline__ void chain_owner_rows(float& a, const float (&ta)[2][6]) {
#define LK4_O1R(o) LK4_ADD1(o, 1) LK4_ADD1(o, 2) LK4_ADD1(o, 3) LK4_ADD1(o, 4) LK4_ADD1(o, 5) LK4_ADD1(o, 6) \
                   LK4_ADD1(o, 7) LK4_ADD1(o, 8) LK4_ADD1(o, 9) LK4_ADD1(o, 10) LK4_ADD1(o, 11) LK4_ADD1(o, 12)
  asm(LK4_O1R(0) LK4_O1R(1) LK4_O1R(2) LK4_O1R(3) "s_nop 0"
      : "+v"(a)
      : "v"(ta[0][0]), "v"(ta[0][1]), "v"(ta[0][2]), "v"(ta[0][3]), "v"(ta[0][4]), "v"(ta[0][5]), "v"(ta[1][0]), "v"(ta[1][1]),
        "v"(ta[1][2]), "v"(ta[1][3]), "v"(ta[1][4]), "v"(ta[1][5]));
#undef LK4_O1R
}

template <int WIN>
__global__ __launch_bounds__(64, 3) void lk4_kernel(KParams P, const unsigned char* prev_img, size_t prev_row_stride,
                                                    size_t prev_img_stride, const unsigned char* prev_pyr,
                                                    const unsigned char* cur_img, size_t cur_row_stride,
                                                    size_t cur_img_stride, const unsigned char* cur_pyr, LkScratch lk,
                                                    int order) {
  using C = Lk4<WIN>;
  constexpr int NC = C::NC, NCH = C::NCH, NG = C::NG, WS = C::WS, WP = C::WP, NT = C::NT, PSTR = C::PSTR,
                DSTR = C::DSTR, HROWS = C::HROWS, PROWS = C::PROWS, JM = C::JM, JS = C::JS, JW = C::JW, NJ = C::NJ,
                JSTRB = C::JSTRB, JH = C::JH;
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * C::RB];

  // Dispatch order (results do not depend on it).  The launch ends on the few waves that hold a point running all 30
  // iterations on every level (~0.11 ms for such a wave alone), so those should start EARLY.  They sit at the END of a
  // stream's list -- the youngest landmarks: corners the detector appended last, which have not yet survived a track -- so
  // a stream's blocks are taken in reverse (order & 1), and consecutive workgroups take the same block of consecutive
  // streams (order & 2) so that every stream's last blocks are among the first dispatched: 0.325 -> 0.262 ms at 64 x 600
  // points (tools/r6/gpu_lk4_order.sh; KVFE_LK4_ORDER=0..3 is the A/B switch, 3 the default)
  int bx = blockIdx.x, s = blockIdx.y;
  if (order & 2) {
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    s = lin % gridDim.y;
    bx = lin / gridDim.y;
  }
  const int npts = lk.npts[s];
  if (order & 1) bx = (npts + 3) / 4 - 1 - bx;
  if (bx < 0 || bx * 4 >= npts) return;
  const int lane = threadIdx.x;
  const int slot = lane >> 4, l16 = lane & 15, g = l16 >> 2, q = l16 & 3;
  const int tcol = l16 & 7, half = l16 >> 3;   // set-up role: column group, half of the rows
  const int pbase = lane & ~15;
  const int pt = bx * 4 + slot;
  bool valid = pt < npts;
  const size_t po = (size_t)s * P.kcap + (valid ? pt : 0);
  // Tracker.cpp:167-180 drops a point whose landmark is older than maxFeatureAge whatever its tracking result; the step
  // passes the ages and such a point is reported lost without being tracked (see lk_kernel_sys)
  if (valid && lk.skip_age && lk.skip_age[(size_t)s * P.kcap + lk.src_idx[po]] > P.max_age) {
    if (l16 == 0) lk.status[po] = 0;
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

  LK4P_DECL;
  for (int level = maxLevel; level >= 0; level--) {
    LK4P(6);
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

    __syncthreads();   // the readers of the region (previous level) are done
    LK4P(0);
    if (lvl && tcol < NT) {
      // Column group tcol of the Scharr pass (derivative positions x = 4 tcol .. 4 tcol + 3, source byte columns
      // X0 .. X0 + 5) walked down this lane's half of the rows: every source row is loaded once (two dwords at byte
      // addresses; six single bytes at reflected columns for a group on the left / right border), the middle row of the
      // 3-row window doubles as the byte patch of the template gather.
      const int X0 = ipx - 1 + 4 * tcol;
      const int hoff = tcol < NT - 1 ? 4 : 0;   // (the last group holds one output: its second dword repeats the first)
      const bool edge = X0 < 0 || X0 + (tcol < NT - 1 ? 8 : 4) > LI.w;
      const int r0 = half * (WP - HROWS);       // first derivative row of this half (the middle row is written twice)
      int lo[PROWS], hi[PROWS];
      unsigned ro[PROWS];   // byte offset of the row in the level (32 bit: the loads take the uniform base as scalar)
#pragma unroll
      for (int pr = 0; pr < PROWS; pr++) ro[pr] = (unsigned)refl1(ipy - 1 + r0 + pr, LI.h) * (unsigned)LI.stride;
      if (!edge) {
#pragma unroll
        for (int pr = 0; pr < PROWS; pr++) {
          lo[pr] = *reinterpret_cast<const int_u*>(LI.p + (ro[pr] + (unsigned)X0));
          hi[pr] = *reinterpret_cast<const int_u*>(LI.p + (ro[pr] + (unsigned)(X0 + hoff)));
        }
      } else {
        int ecol[6];
#pragma unroll
        for (int k = 0; k < 6; k++) ecol[k] = refl1(X0 + k, LI.w);
#pragma unroll
        for (int pr = 0; pr < PROWS; pr++) {
          const unsigned char* r = LI.p + ro[pr];
          lo[pr] = (int)r[ecol[0]] | ((int)r[ecol[1]] << 8) | ((int)r[ecol[2]] << 16) | ((int)r[ecol[3]] << 24);
          hi[pr] = (int)r[ecol[4]] | ((int)r[ecol[5]] << 8);
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // (all loads of the pass in flight before the first is consumed)
      // the derivative is zero at positions outside the image (BORDER_CONSTANT): column masks, row mask per row
      int cmask[4];
#pragma unroll
      for (int k = 0; k < 4; k++) cmask[k] = (unsigned)(ipx + 4 * tcol + k) < (unsigned)LI.w ? -1 : 0;
      const v2us k3 = {3, 3}, k10 = {10, 10};
      v2us a01, a23, a45, b01, b23, b45;   // the two rows above the one being unpacked
      unsigned char* prow = patch + r0 * PSTR + 4 * tcol;
      int* drow = dxy + r0 * DSTR + 4 * tcol;
#pragma unroll
      for (int pr = 0; pr < PROWS; pr++) {
        *reinterpret_cast<int*>(prow + pr * PSTR) = lo[pr];
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
          const int rmask = (unsigned)(ipy + r0 + y) < (unsigned)LI.h ? -1 : 0;
          int* o = drow + y * DSTR;
          o[0] = pack_lo16(as_i32(vx12), as_i32(vy12)) & (cmask[0] & rmask);
          if (tcol < NT - 1) {   // (the last group's other outputs would land in the next row)
            o[1] = pack_hi16(as_i32(vx12), as_i32(vy12)) & (cmask[1] & rmask);
            o[2] = pack_lo16(as_i32(vx34), as_i32(vy34)) & (cmask[2] & rmask);
            o[3] = pack_hi16(as_i32(vx34), as_i32(vy34)) & (cmask[3] & rmask);
          }
        }
        a01 = b01; a23 = b23; a45 = b45;
        b01 = c01; b23 = c23; b45 = c45;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    LK4P(1);

    // bilinear template (folded: 2^8 - (I << 9), see lk_kernel_sys) and gradient window of this lane's pixels, as
    // (column 8cc + g, column 8cc + 4 + g) pairs per row and chunk for the gradient; the A chains
    int rA[NG][2][NC];
    int GX[NG][2][NCH], GY[NG][2][NCH];
    float a11 = 0.f, a12 = 0.f, a22 = 0.f;
#pragma unroll
    for (int j = 0; j < NG; j++) {
      const int yl = 8 * j + 2 * q;   // first window row of the group (lane-dependent: in the base address)
      int pp[3][NC], kx[3][NC], ky[3][NC];
#pragma unroll
      for (int rr = 0; rr < 3; rr++)
#pragma unroll
        for (int c = 0; c < NC; c++) {
          const unsigned char* s0 = patch + (yl + 1 + rr) * PSTR + (g + 4 * c + 1);
          pp[rr][c] = (int)s0[0] | ((int)s0[1] << 16);
          const int* d = dxy + (yl + rr) * DSTR + (g + 4 * c);
          const int d0 = d[0], d1 = d[1];
          kx[rr][c] = pack_lo16(d0, d1);
          ky[rr][c] = pack_hi16(d0, d1);
        }
      float pxx[2][NC], pxy[2][NC], pyy[2][NC];
#pragma unroll
      for (int r = 0; r < 2; r++) {
        int iv[6], vx[6], vy[6];
        lk4_blend_row6<9>(pp[r], pp[r + 1], wq0, wq1, 1 << 8, iv);
        lk4_blend_row6<14>(kx[r], kx[r + 1], wq0, wq1, 1 << 13, vx);
        lk4_blend_row6<14>(ky[r], ky[r + 1], wq0, wq1, 1 << 13, vy);
#pragma unroll
        for (int c = 0; c < NC; c++) {
          rA[j][r][c] = (1 << 8) - (iv[c] << 9);
          const float fx = (float)vx[c], fy = (float)vy[c];
          pxx[r][c] = fx * fx;
          pxy[r][c] = fx * fy;
          pyy[r][c] = fy * fy;
        }
#pragma unroll
        for (int cc = 0; cc < NCH; cc++) {
          GX[j][r][cc] = pack_lo16(vx[2 * cc], vx[2 * cc + 1]);
          GY[j][r][cc] = pack_lo16(vy[2 * cc], vy[2 * cc + 1]);
        }
      }
      // rows 8j + 2o + {0, 1} come from owner o: row order = owner order.  (The products are plain VALU results of the
      // compiler's own instructions: two wait states before the first DPP read.)
      asm volatile("s_nop 1");
      chain_owner_rows_x2(a11, pxx, a12, pxy);
      chain_owner_rows(a22, pyy);
    }
    LK4P(2);

    float A11, A12, A22;
    {
      // iA = 0 + (((c0 + c1) + c2) + c3) over the classes (quads of the point's DPP row)
      const float k0 = __shfl(a11, pbase + 0), k1 = __shfl(a11, pbase + 4), k2 = __shfl(a11, pbase + 8),
                  k3 = __shfl(a11, pbase + 12);
      const float x0 = __shfl(a12, pbase + 0), x1 = __shfl(a12, pbase + 4), x2 = __shfl(a12, pbase + 8),
                  x3 = __shfl(a12, pbase + 12);
      const float q0 = __shfl(a22, pbase + 0), q1 = __shfl(a22, pbase + 4), q2 = __shfl(a22, pbase + 8),
                  q3 = __shfl(a22, pbase + 12);
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
    int jboff = 2 * q * JSTRB + 2 * g;   // (a point that never staged a window reads its region's garbage: never used)
    LK4P(3);
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
        // (re)stage the current-level window of the points that left theirs, as (pixel | next pixel << 8) byte pairs:
        // column group tcol (four pairs of a row out of five bytes; the last group two pairs out of three) of this lane's
        // half of the rows
        LK4P(5);
        LK4P_COUNT(p4_stages);
        __syncthreads();
        if (need) {
          jx0 = inx - JM;
          jy0 = iny - JM;
          jvalid = true;
          if (tcol < NJ) {
            const int XJ = jx0 + 4 * tcol;
            const int hoff = tcol < NJ - 1 ? 4 : 0;
            const int y0 = half * (JS - JH);   // (an odd number of rows: the middle one is written twice)
            int lo[JH], hi[JH];
            if (XJ >= 0 && XJ + (tcol < NJ - 1 ? 8 : 4) <= LJ.w) {
#pragma unroll
              for (int k = 0; k < JH; k++) {
                const unsigned ro = (unsigned)refl1(jy0 + y0 + k, LJ.h) * (unsigned)LJ.stride + (unsigned)XJ;
                lo[k] = *reinterpret_cast<const int_u*>(LJ.p + ro);
                hi[k] = *reinterpret_cast<const int_u*>(LJ.p + (ro + (unsigned)hoff));
              }
            } else {
              int jc[5];
#pragma unroll
              for (int k = 0; k < 5; k++) jc[k] = refl1(XJ + k, LJ.w);
#pragma unroll
              for (int k = 0; k < JH; k++) {
                const unsigned char* r = LJ.p + (unsigned)refl1(jy0 + y0 + k, LJ.h) * (unsigned)LJ.stride;
                lo[k] = (int)r[jc[0]] | ((int)r[jc[1]] << 8) | ((int)r[jc[2]] << 16) | ((int)r[jc[3]] << 24);
                hi[k] = (int)r[jc[4]];
              }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < JH; k++) {
              int* o = reinterpret_cast<int*>(reg + (y0 + k) * JSTRB + 8 * tcol);
              o[0] = perm_b32(hi[k], lo[k], 0x02010100u);   // (b0 b1)(b1 b2)
              o[1] = perm_b32(hi[k], lo[k], 0x04030302u);   // (b2 b3)(b3 b4)
            }
          }
        }
        __syncthreads();
        LK4P(4);
      }
      LK4P_COUNT(p4_iters);
      if (active) jboff = (iny - jy0 + 2 * q) * JSTRB + 2 * (inx - jx0 + g);
      const unsigned char* jb = reg + jboff;

      // this lane's pixels: blend of the current frame minus the template, packed difference pairs, chain terms for both
      // gradient components; the chains: every lane of a quad adds the owners' terms in row order
      float b1c = 0.f, b2c = 0.f;
#pragma unroll
      for (int jg = 0; jg < NG; jg++) {
        int E[3][NC];
#pragma unroll
        for (int rr = 0; rr < 3; rr++)
#pragma unroll
          for (int c = 0; c < NC; c++) {
            const int pr = *reinterpret_cast<const unsigned short*>(jb + (8 * jg + rr) * JSTRB + 8 * c);
            E[rr][c] = perm_b32(0, pr, 0x0c010c00u);
          }
        int dd[6];
        lk4_diff_row6(E[0], E[1], rA[jg][0], wq0, wq1, reinterpret_cast<int(&)[3]>(dd[0]));
        lk4_diff_row6(E[1], E[2], rA[jg][1], wq0, wq1, reinterpret_cast<int(&)[3]>(dd[3]));
        float tx[6], ty[6];
        lk4_terms12(dd, reinterpret_cast<const int(&)[6]>(GX[jg][0][0]), reinterpret_cast<const int(&)[6]>(GY[jg][0][0]),
                    tx, ty);
        chain_owners_x2(b1c, tx, b2c, ty);
      }
      // bbuf = qb0 + qb1 ; ib1 += bbuf[0] + bbuf[2] ; ib2 += bbuf[1] + bbuf[3]: classes (0, 2) and (1, 3) first -- the
      // row rotated by two quads, then by one (float addition is commutative: every lane ends with the same two sums)
      const float bb02 = b1c + row_ror_f<8>(b1c), bb13 = b2c + row_ror_f<8>(b2c);
      const float s1 = bb02 + row_ror_f<4>(bb02), s2 = bb13 + row_ror_f<4>(bb13);
      if (active) {
        float ib1 = 0.f, ib2 = 0.f;
        ib1 += s1;
        ib2 += s2;
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
      }
    }

    LK4P(5);
    if (valid && status && level == 0) {
      // (calcOpticalFlowPyrLK clears the status of a point whose final window left the image whenever an error array is
      // passed, and the reference passes one, Tracker.cpp:137-139; the error itself is not computed here)
      const float2 np = make_float2(nextOut.x - halfWin, nextOut.y - halfWin);
      const int inx = (int)floorf(np.x), iny = (int)floorf(np.y);
      if (inx < -WIN || inx >= LJ.w || iny < -WIN || iny >= LJ.h) status = 0;
    }
  }
#ifdef KVFE_LK4_PROF
  if (lane == 0) {
    for (int i = 0; i < 8; i++) atomicAdd(&kvfe_lk4_prof[i], p4_acc[i]);
    atomicAdd(&kvfe_lk4_prof[8], 1ull);
    atomicAdd(&kvfe_lk4_prof[9], (unsigned long long)p4_iters);
    atomicAdd(&kvfe_lk4_prof[10], (unsigned long long)p4_stages);
  }
#endif
  if (valid && l16 == 0) {
    lk.next_pts[po] = nextOut;
    lk.status[po] = (unsigned char)status;
    lk.err[po] = 0.f;
  }
}

bool launch_lk4(const KParams& P, const unsigned char* prev_img, size_t prev_row_stride, size_t prev_img_stride,
                const unsigned char* prev_pyr, const unsigned char* cur_img, size_t cur_row_stride,
                size_t cur_img_stride, const unsigned char* cur_pyr, const LkScratch& lk, int max_pts, hipStream_t st) {
  if (P.klt_win != 24) return false;
  for (int l = 0; l < P.nlevels; l++)
    if (P.lw[l] < LK4_MIN_DIM || P.lh[l] < LK4_MIN_DIM) return false;   // (refl1: one fold)
  const dim3 grid((max_pts + 3) / 4, P.B), block(64);
  static const int order = [] { const char* e = std::getenv("KVFE_LK4_ORDER"); return e ? std::atoi(e) : 3; }();
#ifdef KVFE_LK4_PROF
  {
    static bool reg = false;
    if (!reg) {
      reg = true;
      std::atexit([] {
        unsigned long long h[LK4P_N];
        (void)hipDeviceSynchronize();
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(kvfe_lk4_prof), sizeof(h));
        const double n = h[8] ? (double)h[8] : 1.0;
        std::fprintf(stderr, "KVFE_LK4_PROF waves %.0f  cycles per wave: header %.0f  column walk %.0f  gather+A %.0f  A sums %.0f  "
                     "staging %.0f  iterations %.0f  between levels %.0f | wave-iterations %.2f  staging passes %.2f\n",
                     n, h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[5] / n, h[6] / n, h[9] / n, h[10] / n);
      });
    }
  }
#endif
  hipLaunchKernelGGL(lk4_kernel<24>, grid, block, 0, st, P, prev_img, prev_row_stride, prev_img_stride, prev_pyr, cur_img,
                     cur_row_stride, cur_img_stride, cur_pyr, lk, order);
  return true;
}

}  // namespace kvfe
