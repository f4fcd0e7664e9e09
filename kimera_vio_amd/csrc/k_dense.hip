// Dense stereo correspondence for gfx950: cv::StereoSGBM (MODE_HH) as reached through
// StereoMatcher::denseStereoReconstruction (reference: src/frontend/StereoMatcher.cpp:32-121,
// DenseStereoParams include/kimera-vio/frontend/StereoMatchingParams.h:39-58), bit-exact against
// oracle/ocv_stereo.cpp (integer arithmetic throughout).
//
// Data layout in HBM, per rectified pair (W x H, D disparities, width1 = W - minX1 columns that
// can be matched):
//   rec    u8x8 [2][H][W]          per pixel of the left / right image: Sobel-x response through the
//                                  clip table and raw intensity, each with the min / max over the
//                                  half-pixel neighbourhood (Birchfield-Tomasi)
//   vol0-2 i16  [H][width1][D]     cost volumes, disparity innermost so that one wave (lane = d) reads
//                                  one 128-byte line per pixel: pixel cost -> row sums -> C(p,d);
//                                  the first two are then reused for the two 4-path sums
//   disp   i16  [2][H][W], labels i32 [2][H][W]
// Kernels (all HBM / latency bound; no contraction, so no MFMA):
//   dense_prefilter   1 thread / pixel
//   dense_bt_cost     wave = one pixel, lane = disparity
//   dense_hsum/vsum   sliding window sums along x then y (2 lines read per line written)
//   dense_cost_fused  MODE_HH: the three kernels above in one (only C(p,d) reaches HBM)
//   dense_aggregate   one wave per scan-line path (8 directions), lane = disparity; neighbours d-1 / d+1
//                     through DPP wave shifts, min_k L_r through DPP + readlane; the running path cost
//                     stays in a register, the 4-path sums are accumulated in place (u16)
//   dense_select      block = image row: winner-take-all, uniqueness, sub-pixel fit, the right-view
//                     disparity buffer as 64-bit LDS atomic-min keys, left-right check
//   dense_median3/5, speckle filter as connected-component labelling (row runs + union-find)
#include "kvfe_dev.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace kvfe {

namespace {

constexpr int DISP_SHIFT = 4;
constexpr int DISP_SCALE = 16;
constexpr int MAXC = 32767;

__device__ __forceinline__ int dpp_wave_shr1(int v, int fill) {   // lane i <- lane i-1, lane 0 <- fill
  return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ int dpp_wave_shl1(int v, int fill) {   // lane i <- lane i+1, lane 63 <- fill
  return __builtin_amdgcn_update_dpp(fill, v, 0x130, 0xf, 0xf, false);
}
// minimum over the 64 lanes, returned uniformly
__device__ __forceinline__ int wave_min(int v) {
  // the minimum of every row of 16 lanes, written out (round 6): from the builtins the compiler makes v_mov_b32 +
  // v_mov_b32_dpp + v_min_i32 and a wait per rotation -- 16 issue slots of the ~70 of a sweep step, and the few-pairs launch
  // is bound by the instructions of its steps (profiles/r6_analysis.md section 17).  A DPP operand written by the
  // instruction before needs two wait states (tools/check_dpp_hazard.py scans this file).
  asm("s_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0"   // (the v_readlane_b32 behind the statement reads what the last instruction wrote: one wait state, which
      : "+v"(v));   //  the compiler pads between its own instructions and cannot see here)
  const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  return min(min(a, b), min(c, d));
}

// ---- calcPixelCostBT, first half: per-pixel records -------------------------------------------
// (stereosgbm.cpp calcPixelCostBT: prow1/prow2 rows and the v0/v1, u0/u1 half-pixel bounds)
__global__ __launch_bounds__(256) void dense_prefilter_kernel(DenseParams P, const uint8_t* __restrict__ left,
                                                              const uint8_t* __restrict__ right,
                                                              uint2* __restrict__ rec) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, img = blockIdx.z & 1, pair = blockIdx.z >> 1;
  if (x >= P.W) return;
  const size_t plane = (size_t)P.W * P.H;
  const uint8_t* src = (img ? right : left) + pair * plane;
  const uint8_t* r1 = src + (size_t)y * P.W;
  const uint8_t* rn = y > 0 ? r1 - P.W : r1;
  const uint8_t* rs = y < P.H - 1 ? r1 + P.W : r1;
  const int ft = P.ftzero;
  auto chan = [&](int xx, int& p0, int& p1) {
    if (xx <= 0 || xx >= P.W - 1) {   // first / last column of both channels: tab[0]
      p0 = p1 = ft;
      return;
    }
    const int g = (r1[xx + 1] - r1[xx - 1]) * 2 + rn[xx + 1] - rn[xx - 1] + rs[xx + 1] - rs[xx - 1];
    p0 = min(max(g, -ft), ft) + ft;
    p1 = r1[xx];
  };
  int a0, a1, b0, b1, c0, c1;
  chan(x, a0, a1);
  chan(x - 1, b0, b1);
  chan(x + 1, c0, c1);
  const int l0 = x > 0 ? (a0 + b0) / 2 : a0, h0 = x < P.W - 1 ? (a0 + c0) / 2 : a0;
  const int l1 = x > 0 ? (a1 + b1) / 2 : a1, h1 = x < P.W - 1 ? (a1 + c1) / 2 : a1;
  const int lo0 = min(min(l0, h0), a0), hi0 = max(max(l0, h0), a0);
  const int lo1 = min(min(l1, h1), a1), hi1 = max(max(l1, h1), a1);
  uint2 o;
  o.x = (unsigned)a0 | ((unsigned)lo0 << 8) | ((unsigned)hi0 << 16) | ((unsigned)a1 << 24);
  o.y = (unsigned)lo1 | ((unsigned)hi1 << 8);
  rec[((size_t)blockIdx.z * P.H + y) * P.W + x] = o;
}

// ---- calcPixelCostBT, second half: cost(x, d), wave = pixel, lane = d -----------------------------
constexpr int BT_XPB = 64;   // columns per block (16 per wave)
__global__ __launch_bounds__(256) void dense_bt_cost_kernel(DenseParams P, const uint2* __restrict__ rec,
                                                            short* __restrict__ pix) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int y = blockIdx.y, pair = blockIdx.z;
  const int xb = blockIdx.x * BT_XPB + wv * (BT_XPB / 4);
  const uint2* recL = rec + ((size_t)(pair * 2) * P.H + y) * P.W;
  const uint2* recR = rec + ((size_t)(pair * 2 + 1) * P.H + y) * P.W;
  short* out = pix + (((size_t)pair * P.H + y) * P.width1) * P.D;
  const bool act = lane < P.D;
  for (int i = 0; i < BT_XPB / 4; i++) {
    const int x = xb + i;
    if (x >= P.width1) break;
    const int xa = x + P.minX1;
    const uint2 L = recL[xa];
    const int xr = xa - (lane + P.minD);
    const uint2 R = recR[act ? xr : xa];
    const int u = L.x & 255, u0 = (L.x >> 8) & 255, u1 = (L.x >> 16) & 255;
    const int v = R.x & 255, v0 = (R.x >> 8) & 255, v1 = (R.x >> 16) & 255;
    const int c0 = max(max(0, u - v1), v0 - u), c1 = max(max(0, v - u1), u0 - v);
    const int ur = L.x >> 24, ur0 = L.y & 255, ur1 = (L.y >> 8) & 255;
    const int vr = R.x >> 24, vr0 = R.y & 255, vr1 = (R.y >> 8) & 255;
    const int e0 = max(max(0, ur - vr1), vr0 - ur), e1 = max(max(0, vr - ur1), ur0 - vr);
    if (act) out[(size_t)x * P.D + lane] = (short)(min(c0, c1) + (min(e0, e1) >> 2));
  }
}

// ---- window sums (computeDisparitySGBM: hsumAdd / C recurrences, written as clamped box sums) ---------
// hs(y,x,d) = sum_{i=-SW2..SW2} pix(y, clamp(x+i, 0, width1-1), d)
constexpr int HS_CHUNK = 32;
__global__ __launch_bounds__(256) void dense_hsum_kernel(DenseParams P, const short* __restrict__ pix,
                                                         short* __restrict__ hs) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int y = blockIdx.y, pair = blockIdx.z;
  const int x0 = (blockIdx.x * 4 + wv) * HS_CHUNK;
  if (x0 >= P.width1 || lane >= P.D) return;
  const size_t row = ((size_t)pair * P.H + y) * P.width1;
  const short* in = pix + row * P.D + lane;
  short* out = hs + row * P.D + lane;
  const int w1 = P.width1 - 1;
  int s = 0;
  for (int i = -P.SW2; i <= P.SW2; i++) s += in[(size_t)min(max(x0 + i, 0), w1) * P.D];
  out[(size_t)x0 * P.D] = (short)s;
  const int xe = min(x0 + HS_CHUNK, P.width1);
#pragma unroll 8
  for (int x = x0 + 1; x < xe; x++) {
    s += in[(size_t)min(x + P.SW2, w1) * P.D] - in[(size_t)max(x - P.SW2 - 1, 0) * P.D];
    out[(size_t)x * P.D] = (short)s;
  }
}

// C(y,x,d) = P2 + sum_{k=-SH2..SH2} hs(clamp(y+k, 0, H-1), x, d); OpenCV's recurrence leaves C at its
// initial value P2 in column 0 of every row but the first, and in the last SH2 rows (y + SH2 >= H)
constexpr int VS_CHUNK = 32;
__global__ __launch_bounds__(256) void dense_vsum_kernel(DenseParams P, const short* __restrict__ hs,
                                                         short* __restrict__ Cv) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int x = blockIdx.x * 4 + wv, pair = blockIdx.z;
  const int y0 = blockIdx.y * VS_CHUNK;
  if (x >= P.width1 || lane >= P.D) return;
  const size_t rs = (size_t)P.width1 * P.D;
  const short* in = hs + (size_t)pair * P.H * rs + (size_t)x * P.D + lane;
  short* out = Cv + (size_t)pair * P.H * rs + (size_t)x * P.D + lane;
  const int h1 = P.H - 1, SH2 = P.SW2;
  const int ye = min(y0 + VS_CHUNK, P.H);
  // the clamped window sum slides for every row; the rows / column OpenCV's recurrence does not touch just
  // do not use it.  MODE_HH keeps C for every row, untouched entries hold the initial P2; MODE_SGBM reuses
  // one row buffer, so untouched entries repeat what the row above held: column 0 keeps the first row's
  // value and the last SH2 rows repeat row H-1-SH2.
  const int ylast = max(P.H - 1 - SH2, 0);   // last row whose C is computed (y + SH2 < H), row 0 always is
  auto window = [&](int yc) {
    int t = 0;
    for (int k = -SH2; k <= SH2; k++) t += in[(size_t)min(max(yc + k, 0), h1) * rs];
    return t;
  };
  int s = window(y0);
  int frozen = 0;
  if (!P.full_dp) frozen = x == 0 ? window(0) : window(ylast);
  auto value = [&](int y, int cur) {
    if (P.bm) return cur;   // cv::StereoBM: plain window sum of the row SADs (rows clamped to the image)
    const bool untouched = y > 0 && (x == 0 || y + SH2 >= P.H);
    if (!untouched) return P.P2 + cur;
    return P.full_dp ? P.P2 : P.P2 + frozen;
  };
  out[(size_t)y0 * rs] = (short)value(y0, s);
#pragma unroll 8
  for (int y = y0 + 1; y < ye; y++) {
    s += in[(size_t)min(y + SH2, h1) * rs] - in[(size_t)max(y - SH2 - 1, 0) * rs];
    out[(size_t)y * rs] = (short)value(y, s);
  }
}

// ---- C(p,d) in one kernel (MODE_HH) -------------------------------------------------------------------
// The three kernels above move the cost volume five times (pixel costs written and read, row sums written and read, C
// written): 0.19 GB of the 1.30 GB a 752x480 pair costs.  Here one wave owns NCOL matchable columns x RC rows of a pair
// (lane = disparity) and walks its rows top-down: the pixel costs of the NCOL + 2 SW2 columns of a row are evaluated from
// the 8-byte records into registers, the row sums slide along them, and the column sums slide down through a ring of the
// last 2 SW2 + 1 row sums in LDS (two columns per dword).  Only C is written.
// The records of a row reach the wave through three vector loads issued one row ahead (lane l: the left record of window
// column l, the right records lo + l and lo + 64 + l) and a 1 KB LDS stage: column j then reads the left record at a
// wave-uniform address (broadcast) and its right record at position 63 + j - lane -- immediate offsets, no address
// arithmetic beyond the clamp of the window column (hsum's: columns outside [0, width1) repeat the edge column), no
// global-load latency inside the row.
// Price: the pixel costs of the halo columns and rows are evaluated by two waves ((NCOL + 2 SW2) / NCOL x
// (RC + 2 SW2) / RC evaluations per output).  Same integers as the three kernels: every sum is below 2^15
// (11 x 11 x 189 + P2).  MODE_SGBM keeps the three kernels (its frozen rows need sums of other rows).
// Measured (8 pairs of 752x480, 64 disparities): 0.35 ms against 0.70 ms for the three kernels; 16 / 12 / 8 columns per
// wave give 0.356 / 0.346 / 0.35 ms (LDS ring 22.5 / 16.9 / 11.3 KB per wave against 1.6 / 1.8 / 2.3 evaluations per
// output) -- profiles/r4_analysis.md.
#ifndef KVFE_CF_NCOL
#define KVFE_CF_NCOL 12
#endif
constexpr int CF_NCOL = KVFE_CF_NCOL;
template <int SW2>
__global__ __launch_bounds__(64) void dense_cost_fused_kernel(DenseParams P, const uint2* __restrict__ rec,
                                                              short* __restrict__ Cv, int RC) {
  constexpr int NCOL = CF_NCOL, NR = 2 * SW2 + 1, NP = NCOL + 2 * SW2, NK = NCOL / 2;
  static_assert(NCOL % 2 == 0 && NP <= 64, "two columns per ring dword; one lane per window column");
  __shared__ unsigned ring[NR * NK * 64];
  __shared__ uint2 stage_r[128];
  __shared__ uint2 stage_l[NP];
  const int lane = threadIdx.x;
  const int x0 = blockIdx.x * NCOL, y0 = blockIdx.y * RC, pair = blockIdx.z;
  const int W = P.W, H = P.H, W1 = P.width1, w1 = W1 - 1, D = P.D;
  const bool act = lane < D;
  const size_t plane = (size_t)H * W;
  const uint2* recL = rec + (size_t)(pair * 2) * plane;
  const uint2* recR = recL + plane;
  short* out = Cv + (size_t)pair * H * W1 * D + min(lane, D - 1);
  // image columns this lane fetches every row (clamped into the row: what a clamped right-image address returns
  // belongs to an idle lane or to a window column nobody reads)
  const int xl0 = x0 - SW2 + P.minX1;              // image column of window column 0
  const int lo = xl0 - P.minD - 63;                // right record at stage position 0
  const int cl = min(max(x0 - SW2 + lane, 0), w1) + P.minX1;   // (clamped like the window column it serves)
  const int ca = min(max(lo + lane, 0), W - 1), cb = min(max(lo + 64 + lane, 0), W - 1);
  const uint2* sr = stage_r + (63 - lane);         // + j: the right record of window column j at this lane's disparity
  // (the left record's address is wave-uniform; hidden from the compiler, which would otherwise move all 26 records
  // through v_readfirstlane into scalar registers and spill them)
  int zero_v;
  asm volatile("v_mov_b32 %0, 0" : "=v"(zero_v));
  const uint2* sl = stage_l + zero_v;
#pragma unroll
  for (int i = 0; i < NR * NK; i++) ring[i * 64 + lane] = 0u;
  unsigned Vp[NK];
#pragma unroll
  for (int k = 0; k < NK; k++) Vp[k] = 0u;
  int slot = 0;
  const int ye = min(y0 + RC, H);
  // window columns left of column 0 / right of column width1-1 repeat the edge column (hsum's clamp)
  const int jl = x0 == 0 ? SW2 : 0, jr = min(w1 - x0 + SW2, NP - 1);
  uint2 nl, na, nb;
  {
    const size_t ro = (size_t)min(max(y0 - SW2, 0), H - 1) * W;
    nl = recL[ro + cl];
    na = recR[ro + ca];
    nb = recR[ro + cb];
  }
  for (int v = y0 - SW2; v < ye + SW2; v++) {
    // (the block is ONE wave -- __launch_bounds__(64), checked by the launcher -- so a barrier is a wave barrier and
    // free; it is what orders the previous row's cross-lane reads of the stage before these writes and the writes
    // before this row's reads, instead of the in-order LDS pipe and the compiler's caution doing so by accident)
    __syncthreads();
    if (lane < NP) stage_l[lane] = nl;
    stage_r[lane] = na;
    stage_r[64 + lane] = nb;
    __syncthreads();
    {   // the next row's records (the last iteration re-reads its own row)
      const size_t ro = (size_t)min(max(min(v + 1, ye + SW2 - 1), 0), H - 1) * W;
      nl = recL[ro + cl];
      na = recR[ro + ca];
      nb = recR[ro + cb];
    }
    int pix[NP];
#pragma unroll
    for (int j = 0; j < NP; j++) {
      const int jc = min(max(j, jl), jr);   // (wave-uniform: hsum's clamp of the window column)
      const uint2 L = sl[j];
      const uint2 R = sr[jc];
      const int u = L.x & 255, u0 = (L.x >> 8) & 255, u1 = (L.x >> 16) & 255;
      const int q = R.x & 255, q0 = (R.x >> 8) & 255, q1 = (R.x >> 16) & 255;
      const int c0 = max(max(0, u - q1), q0 - u), c1 = max(max(0, q - u1), u0 - q);
      const int ur = L.x >> 24, ur0 = L.y & 255, ur1 = (L.y >> 8) & 255;
      const int qr = R.x >> 24, qr0 = R.y & 255, qr1 = (R.y >> 8) & 255;
      const int e0 = max(max(0, ur - qr1), qr0 - ur), e1 = max(max(0, qr - ur1), ur0 - qr);
      pix[j] = min(c0, c1) + (min(e0, e1) >> 2);
    }
    int s = 0;
#pragma unroll
    for (int j = 0; j < NR; j++) s += pix[j];
    unsigned hp[NK];
#pragma unroll
    for (int c = 0; c < NCOL; c++) {
      if (c > 0) s += pix[c + 2 * SW2] - pix[c - 1];
      if (c & 1)
        hp[c >> 1] |= (unsigned)s << 16;
      else
        hp[c >> 1] = (unsigned)s;
    }
    // column sums: + this row, - the row that leaves the window (the halves never borrow from each other: both stay
    // true window sums in [0, 2^16))
    unsigned* rg = ring + (size_t)slot * NK * 64 + lane;
#pragma unroll
    for (int k = 0; k < NK; k++) {
      const unsigned old = rg[k * 64];
      Vp[k] = Vp[k] + hp[k] - old;
      rg[k * 64] = hp[k];
    }
    slot = slot + 1 == NR ? 0 : slot + 1;
    const int y = v - SW2;
    if (y >= y0) {
      // (OpenCV's recurrence leaves C at its initial P2 in column 0 of every row but the first and in the last SW2 rows)
      const bool rowdead = y > 0 && y + SW2 >= H;
      short* o = out + ((size_t)y * W1 + x0) * D;
#pragma unroll
      for (int c = 0; c < NCOL; c++) {
        const int val = (int)((Vp[c >> 1] >> ((c & 1) * 16)) & 0xffffu);
        const bool dead = rowdead || (y > 0 && x0 + c == 0);
        if (act && x0 + c < W1) o[(size_t)c * D] = (short)(P.P2 + (dead ? 0 : val));
      }
    }
  }
}

// ---- path aggregation ---------------------------------------------------------------------------
// L_r(p,d) = C(p,d) + min(L_r(p-r,d), L_r(p-r,d-1)+P1, L_r(p-r,d+1)+P1, min_k L_r(p-r,k)+P2) - (min_k L_r(p-r,k)+P2)
// along one scan line; outside the volume L_r = 0 for every d (the zeroed borders of OpenCV's Lr / minLr
// buffers), L_r(., -1) = L_r(., D) = SHRT_MAX.  SX,SY = direction of travel (p - r is the previous pixel).
// The four path costs of a pass are added into one u16 volume with saturation at 65535.  OpenCV keeps
// S = saturate_cast<short>(S + L0 + L1 + L2 + L3) per pass; every L_r >= 0 (C >= P2 and the min term is
// >= min_k L_r - delta = -P2), so saturate(saturate(A) + B) == min(32767, A + B) and the order in which the
// eight directions are accumulated does not matter.
// MODE: 0 = this sweep writes the sum, 1 = read-modify-write (sweeps launched one after the other),
// 2 = packed 32-bit atomic adds of two u16 sums (all sweeps of a pair in one launch, see below)
enum { AGG_WRITE = 0, AGG_RMW = 1, AGG_ATOMIC = 2 };
template <int SX, int SY, int MODE>
__device__ __forceinline__ void dense_aggregate_path(const DenseParams& P, const short* __restrict__ Cv,
                                                     unsigned short* __restrict__ sum, int path, int pair) {
  const int lane = threadIdx.x & 63;
  constexpr bool FIRST = MODE != AGG_RMW;   // no read of the running sum
  const int W1 = P.width1, H = P.H;
  int x, y, len;
  if (SY == 0) {
    if (path >= H) return;
    y = path;
    x = SX > 0 ? 0 : W1 - 1;
    len = W1;
  } else if (SX == 0) {
    if (path >= W1) return;
    x = path;
    y = SY > 0 ? 0 : H - 1;
    len = H;
  } else {
    if (path >= W1 + H - 1) return;
    if (path < W1) {
      x = path;
      y = SY > 0 ? 0 : H - 1;
    } else {
      const int j = path - W1;
      x = SX > 0 ? 0 : W1 - 1;
      y = SY > 0 ? j + 1 : H - 2 - j;
    }
    const int nx = SX > 0 ? W1 - x : x + 1, ny = SY > 0 ? H - y : y + 1;
    len = min(nx, ny);
  }
  const bool act = lane < P.D;
  // lanes >= D load lane D-1's element (no divergent branch around the loads: a load under a lane
  // predicate makes the compiler wait for it at the end of the branch, which serialises the prefetch)
  const size_t base = (size_t)pair * H * W1 * P.D + ((size_t)y * W1 + x) * P.D + min(lane, P.D - 1);
  const long step = ((long)SY * W1 + SX) * P.D;
  const short* cp = Cv + base;
  unsigned short* sp = sum + base;
  int Lp = act ? 0 : MAXC;   // lanes >= D stand for d = D, D+1, ...: never below SHRT_MAX
  int minp = 0;
  const int P1 = P.P1, P2 = P.P2;
  // software pipeline: the loads of the next pixels do not depend on the recurrence
  constexpr int PF = 8;
  int cbuf[PF];
  int sbuf[PF];
  const int last = len - 1;
#pragma unroll
  for (int i = 0; i < PF; i++) {
    cbuf[i] = cp[(long)min(i, last) * step];
    if (!FIRST) sbuf[i] = sp[(long)min(i, last) * step];
  }
  // element offsets of the pixel requested PF steps ahead and of the pixel stored, advanced by `step` (round 6: two 64-bit
  // products per step were a third of the scalar instructions of a step, and a step's instructions are what bounds the
  // few-pairs launch -- profiles/r6_analysis.md section 17)
  long tn = (long)min(PF, last) * step, ts = 0;
  for (int t0 = 0; t0 < len; t0 += PF) {
#pragma unroll
    for (int i = 0; i < PF; i++) {
      const bool live = t0 + i <= last;  // the tail repeats the last pixel (nothing is stored for it)
      const int c = cbuf[i];
      const int sprev = FIRST ? 0 : sbuf[i];
      cbuf[i] = cp[tn];
      if (!FIRST) sbuf[i] = sp[tn];
      if (t0 + i + PF < last) tn += step;
      const int delta = minp + P2;
      const int lm = dpp_wave_shr1(Lp, MAXC), lq = dpp_wave_shl1(Lp, MAXC);
      const int m = min(min(Lp, delta), min(lm, lq) + P1);
      const int L = c + m - delta;
      if (live) {   // wave-uniform
        Lp = act ? L : MAXC;
        minp = wave_min(Lp);
        if (MODE == AGG_ATOMIC) {
          // even lanes add (L_d | L_{d+1} << 16): four sweeps share a volume, 4 * 16383 < 65536, so no carry
          const int hi = dpp_wave_shl1(act ? L : 0, 0);
          if (act && !(lane & 1))
            atomicAdd(reinterpret_cast<unsigned*>(sp + ts), (unsigned)L | ((unsigned)hi << 16));
        } else if (act) {
          sp[ts] = (unsigned short)min(sprev + L, 65535);
        }
        ts += step;
      }
    }
  }
}

template <int SX, int SY, bool FIRST>
__global__ __launch_bounds__(256) void dense_aggregate_kernel(DenseParams P, const short* __restrict__ Cv,
                                                              unsigned short* __restrict__ sum) {
  dense_aggregate_path<SX, SY, FIRST ? AGG_WRITE : AGG_RMW>(P, Cv, sum, blockIdx.x * 4 + (threadIdx.x >> 6),
                                                            blockIdx.y);
}

// Up to three pairs: a sweep has only 480-1170 scan lines per pair, so a launch per direction is latency bound
// (0.09-0.13 ms each).  Here all directions of a pair run in ONE launch (blockIdx.y = direction) and add into
// two zeroed volumes with packed atomics: directions 0-3 into sumA, 4-7 into sumB (S = min(32767, A + B)).
__global__ __launch_bounds__(256) void dense_aggregate_all_kernel(DenseParams P, const short* __restrict__ Cv,
                                                                  unsigned short* __restrict__ sumA,
                                                                  unsigned short* __restrict__ sumB) {
  const int path = blockIdx.x * 4 + (threadIdx.x >> 6), pair = blockIdx.z;
  switch (blockIdx.y) {
    case 0: dense_aggregate_path<1, 0, AGG_ATOMIC>(P, Cv, sumA, path, pair); break;
    case 1: dense_aggregate_path<1, 1, AGG_ATOMIC>(P, Cv, sumA, path, pair); break;
    case 2: dense_aggregate_path<0, 1, AGG_ATOMIC>(P, Cv, sumA, path, pair); break;
    case 3: dense_aggregate_path<-1, 1, AGG_ATOMIC>(P, Cv, sumA, path, pair); break;
    case 4: dense_aggregate_path<-1, 0, AGG_ATOMIC>(P, Cv, sumB, path, pair); break;
    case 5: dense_aggregate_path<1, -1, AGG_ATOMIC>(P, Cv, sumB, path, pair); break;
    case 6: dense_aggregate_path<0, -1, AGG_ATOMIC>(P, Cv, sumB, path, pair); break;
    default: dense_aggregate_path<-1, -1, AGG_ATOMIC>(P, Cv, sumB, path, pair); break;
  }
}

// ---- two-pass aggregation (MODE_HH, four or more pairs) -------------------------------------------------------
// computeDisparitySGBM's own order: pass 1 walks the rows top-down and every row left to right with the four paths whose
// previous pixel lies behind ((x-1,y), (x-1,y-1), (x,y-1), (x+1,y-1)), pass 2 walks them bottom-up and right to left
// with the other four.  Here a wave is one image row of one pass and does the four paths of a pixel in ONE set of
// instructions: DPP row q of the wave (16 lanes) is path q (0 horizontal, 1 the diagonal whose predecessor is behind, 2
// vertical, 3 the diagonal whose predecessor is ahead), lane g of a row holds disparities 4g .. 4g+3 as two packed
// 16-bit pairs.  C is read twice and each partial sum written once per pixel (eight sweeps: C read eight times, S written
// once and read-modified-written seven times), and a pixel of a pass costs ~50 vector instructions (a wave per path
// with lane = disparity: ~190, which was slower than the eight sweeps -- profiles/r5_analysis.md).
//   * path q of the row r at step i (column x_i) continues path q of row r-1 at step i-2+q: the row before publishes one
//     ENTRY per step (its four new cost vectors, 8 bytes per lane, + the three minima), and lane (q, g) reads its 8 bytes
//     out of entry i-2+q -- one LDS read with per-lane addresses; row 0 of the wave (horizontal) takes its own previous
//     result instead.  A row may run step i as soon as the row before has finished step i+1;
//   * the sum of the four paths is formed from the entry the wave has just written (lanes of row 0 read the other three
//     rows back), one step later so that the LDS latency is not waited for;
//   * a block is 15 consecutive rows + a loader wave; entries go from wave to wave through a ring of 8 in LDS, guarded
//     by a per-wave step counter both ways;
//   * from the last row of a block to the first row of the next they go through HBM as relaxed device-scope atomics, each
//     word tagged with the launch's epoch (two spare bits: costs are <= 16383), so the reader needs no fence: the next
//     block's loader wave polls the words themselves (requested AGP_LPF steps ahead; a word that has not arrived is
//     loaded again) and puts them into the ring its first row reads -- every row wave is the same LDS-fed code, and the
//     registers of the prefetch are the loader's (a row that prefetches for itself needs 84 and halves the occupancy);
//   * blocks take their (band, pair, pass) from a ticket counter, band-major: a block only ever waits for a block with a
//     smaller ticket, which is running -- no co-residency assumption, no deadlock whatever the dispatch order;
//   * every wait is bounded (a budget of AGP_WAIT_TURNS polls per wave); a wave whose budget is used up stops waiting,
//     finishes with what it reads and sets the launch's error word: the call returns an error instead of hanging the
//     device.
// Same integers as dense_aggregate_path: 16383 stands for SHRT_MAX at d = -1, d = D and in the lanes past D (every cost
// is <= 16383 -- dense_params() refuses parameters that could exceed it -- so 16383 + P1 is never the minimum, which is
// all SHRT_MAX does upstream), zero predecessors outside the volume.
constexpr int AGP_ROWS = 15;    // image rows per block
constexpr int AGP_WAVES = 16;   // + the loader wave (wave 0)
constexpr int AGP_LPF = 4;      // steps the loader requests ahead
constexpr int AGP_MIN_PAIRS = 4;   // fewer pairs: too few rows in flight, the single-launch atomic sweeps are faster
constexpr int AGP_RING = 8;
constexpr int AGP_PF = 6;      // requests of C in flight per row
constexpr int AGP_PF_G = 6;  // ... in the last row of a block: its device-scope stores of the entries complete late and sit
                               // in the same in-order counter as the requests (with 6, these rows -- and through the rings the
                               // whole block -- ran at the stores' latency / 2.75 steps)
constexpr unsigned AGP_INF2 = 0x3FFF3FFFu;
constexpr unsigned AGP_TAGMASK = 0x80008000u;

typedef short agp_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short agp_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned agp_pk_min(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned,
                            __builtin_elementwise_min(__builtin_bit_cast(agp_s2, a), __builtin_bit_cast(agp_s2, b)));
}
__device__ __forceinline__ unsigned agp_pk_add(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_bit_cast(agp_u2, a) + __builtin_bit_cast(agp_u2, b));
}
__device__ __forceinline__ unsigned agp_pk_sub(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_bit_cast(agp_u2, a) - __builtin_bit_cast(agp_u2, b));
}

// A wait is a loop over "read the counter, sleep"; every wave has a budget for all its waits together (AGP_WAIT_TURNS
// loops of up to 64 turns: seconds, far beyond a launch); when it is used up the wave stops waiting, finishes its row
// with whatever it reads, and sets the launch's error word.
constexpr unsigned AGP_WAIT_TURNS = 1u << 18;   // x 64 turns of >= 512 cycles

typedef __attribute__((address_space(3))) volatile unsigned long long agp_lds_u64;
typedef __attribute__((address_space(3))) volatile unsigned agp_lds_u32;
enum { AGP_IN_ZERO = 0, AGP_IN_LDS = 1, AGP_OUT_NONE = 0, AGP_OUT_LDS = 1, AGP_OUT_GLOBAL = 2 };

// wait until the step counter *p has reached `need`; returns the counter (wave-uniform).  Three loops of a few
// instructions each, the sleep growing with the time already waited: a row that follows its neighbour step by step
// waits a turn or two, but the rows of a pass start one after the other, and the ones that have not started yet would
// otherwise spend the launch polling every 64 cycles on the SIMDs the running rows need (profiles/r5_analysis.md: 60 %
// of the kernel's scalar instructions).
template <int SLEEP>
__device__ __forceinline__ unsigned agp_wait_loop(agp_lds_u32* p, unsigned need, unsigned seen, unsigned turns,
                                                  unsigned& budget) {
  while (seen < need && turns != 0) {
    __builtin_amdgcn_s_sleep(SLEEP);
    turns--;
    seen = __builtin_amdgcn_readfirstlane(*p);
  }
  budget -= budget != 0 ? 1u : 0u;   // (per loop, not per turn: the budget is a bound, not a clock)
  return seen;
}
// the same for a row's input: every turn reads the counter AND the lane's words of the entry behind it (LDS serves a
// wave's requests in order: words read behind a counter that is high enough are valid), so that the turn that sees the
// counter arrive has the entry already -- no second round trip after the wait
template <int SLEEP>
__device__ __forceinline__ unsigned agp_wait_in_loop(agp_lds_u32* p, agp_lds_u64* pe, agp_lds_u32* pm, unsigned need,
                                                     unsigned seen, unsigned turns, unsigned& budget,
                                                     unsigned long long& v, unsigned& m) {
  while (seen < need && turns != 0) {
    __builtin_amdgcn_s_sleep(SLEEP);
    turns--;
    const unsigned pr = *p;
    v = *pe;
    m = *pm;
    seen = __builtin_amdgcn_readfirstlane(pr);
  }
  budget -= budget != 0 ? 1u : 0u;
  return seen;
}
__device__ __forceinline__ unsigned agp_wait_in(agp_lds_u32* p, agp_lds_u64* pe, agp_lds_u32* pm, unsigned need,
                                                unsigned seen, unsigned& budget, unsigned long long& v, unsigned& m) {
#ifdef KVFE_AGP_FREERUN
  return need;
#endif
  if (budget == 0) return seen;
  seen = agp_wait_in_loop<1>(p, pe, pm, need, seen, 8, budget, v, m);
  if (seen >= need) return seen;
  seen = agp_wait_in_loop<3>(p, pe, pm, need, seen, 16, budget, v, m);
  while (seen < need && budget > 3) seen = agp_wait_in_loop<8>(p, pe, pm, need, seen, 64, budget, v, m);
  if (seen < need) budget = 0;
  return seen;
}

__device__ __forceinline__ unsigned agp_wait(agp_lds_u32* p, unsigned need, unsigned& budget) {
#ifdef KVFE_AGP_FREERUN   // measurement only (wrong results): nobody waits, every row runs at its own speed
  return need;
#endif
  unsigned seen = __builtin_amdgcn_readfirstlane(*p);
  if (seen >= need || budget == 0) return seen;
  seen = agp_wait_loop<1>(p, need, seen, 8, budget);
  if (seen >= need) return seen;
  seen = agp_wait_loop<3>(p, need, seen, 16, budget);
  while (seen < need && budget > 3) seen = agp_wait_loop<8>(p, need, seen, 64, budget);
  if (seen < need) budget = 0;
  return seen;
}

// minimum over the 16 lanes of a DPP row of a value whose two halves are equal (so the 32-bit order is the 16-bit one);
// (written out: the compiler keeps v_mov_b32_dpp + v_min_u32 apart, three instructions per rotation instead of one; a DPP
// operand written by the instruction before needs two wait states)
__device__ __forceinline__ unsigned agp_row_min(unsigned x) {
  asm("s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf"
      : "+v"(x));
  return x;
}

struct AgpRow {
  // wave-uniform bases (scalar registers); the lane's part of an address is a 32-bit offset computed in agp_row
  const char* cp;             // C of the row, column 0 (the volume has a guard band: requests run AGP_PF columns past a row)
  char* sp;                   // the pass's partial sum of the row
  char* hout;                 // entries for the block after: [step][64 lanes] x 8 bytes
  char* hmin_out;             // their minima: [step][4] words
  agp_lds_u64* ring_in;       // ring written by the wave of the row before: [slot][64]
  agp_lds_u32* minring_in;    //   its minima: [slot][4]
  agp_lds_u64* ring_out;      // this wave's ring (or its two scratch slots when nobody reads it)
  agp_lds_u32* minring_out;
  agp_lds_u32 *prog_in, *prog_me, *prog_out;
  unsigned* err;
  int W1, D, P1, P2;
  unsigned etag;
  bool down;
};

// one row of one pass; IN / OUT: where the row before's entries come from and where this row's go (wave-uniform, so
// every memory operation of the loop is unconditional and the compiler can count the loads in flight)
template <int IN, int OUT, bool FULL>
__device__ __forceinline__ void agp_row(const AgpRow& R) {
  constexpr int OM = OUT == AGP_OUT_LDS ? AGP_RING - 1 : 1;   // slots of its own ring the wave uses, - 1
  constexpr int PF = OUT == AGP_OUT_GLOBAL ? AGP_PF_G : AGP_PF;
  const int lane = threadIdx.x & 63;
  const int q = lane >> 4, g = lane & 15;
  const int W1 = R.W1, D = R.D, last = W1 - 1;
  const bool down = R.down;
  const int cstep = down ? D * 2 : -D * 2;                                // bytes of C / S from a step's column to the next
  const unsigned coff = 8u * (unsigned)min(g, D / 4 - 1);                 // this lane's four disparities in a column
  const unsigned hoff = 8u * (unsigned)lane, moff = 4u * (unsigned)q;    // this lane's word of an entry / of the minima
  const bool actl = 4 * g < D;
  constexpr bool all_active = FULL;   // D == 64: no lane of a row lies past the last disparity
  const unsigned zero2 = actl ? 0u : AGP_INF2;
  const unsigned P1P1 = (unsigned)R.P1 * 0x10001u, P2P2 = (unsigned)R.P2 * 0x10001u;
  const unsigned etag = R.etag;
  const bool row0 = q == 0;
  const bool sum_lane = row0 && actl;
  const bool min_lane = g == 0;
  const int qm2 = q - 2;
  unsigned budget = AGP_WAIT_TURNS;
  unsigned seen_in = 0, seen_out = 0;
  const unsigned progaddr = (unsigned)(size_t)R.prog_me;   // LDS byte address of this wave's step counter
  unsigned progv = 0;                                      // its value
#ifdef KVFE_AGP_PROF
  const long long prof_t0 = wall_clock64();
  unsigned prof_in_turns = 0, prof_out_turns = 0, prof_in_steps = 0, prof_out_steps = 0;
#endif
  // C and the sum through buffer descriptors of the row: lane offset in a register, column offset in a scalar that runs
  // with the steps -- no vector instruction per address (a request a few columns past the row's end in the bottom-up
  // pass has a negative = huge offset: out of range, returns 0, never used)
  const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(R.cp), 0, 0x7fffffff, 0x00027000);
  const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(R.sp, 0, 0x7fffffff, 0x00027000);
  int cnext = (down ? 0 : last) * (D * 2);   // byte offset of the column whose C is requested next
  int sprev = cnext;                         // ... of the column whose sum is stored next
  char* hptr = R.hout;
  char* hmptr = R.hmin_out;

  // requests: C of the next PF steps (one 64-bit value each: as two words they are copied at the loop's back edge)
  unsigned long long cbuf[PF];
#pragma unroll
  for (int u = 0; u < PF; u++) {
    cbuf[u] = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(crsrc, coff, cnext, 0));
    cnext += cstep;
  }
  unsigned pA = zero2, pB = zero2, pM = 0;   // this lane's result of the step before (row 0: the horizontal path's input)
  // the neighbour lanes' values; lanes 0 / 15 of a row keep 16383 for good (DPP leaves lanes without a source alone)
  unsigned nbp = AGP_INF2, nbn = AGP_INF2;

  // the sum of the four paths of step j, from the entry this wave wrote at step j: every lane reads rows 1 - 3 of the
  // entry (no lane mask around the LDS reads, so that they are requested together with the step's other reads and
  // waited for once), row 0 adds its own part (hA, hB) and stores
  auto sum_read = [&](int j, unsigned long long& r1, unsigned long long& r2, unsigned long long& r3) {
    const int so = (j & OM) * 64 + g;
    r1 = R.ring_out[so + 16];
    r2 = R.ring_out[so + 32];
    r3 = R.ring_out[so + 48];
  };
  auto sum_store = [&](unsigned hA, unsigned hB, unsigned long long r1, unsigned long long r2, unsigned long long r3) {
    uint2 sum;
    sum.x = agp_pk_add(agp_pk_add(hA, (unsigned)r1), agp_pk_add((unsigned)r2, (unsigned)r3));
    sum.y = agp_pk_add(agp_pk_add(hB, (unsigned)(r1 >> 32)), agp_pk_add((unsigned)(r2 >> 32), (unsigned)(r3 >> 32)));
    typedef unsigned agp_v2u __attribute__((ext_vector_type(2)));
    if (sum_lane) __builtin_amdgcn_raw_buffer_store_b64(agp_v2u{sum.x, sum.y}, srsrc, coff, sprev, 0);
    sprev += cstep;
  };

  // one step; u = i mod PF selects the request register (compile-time in the unrolled groups below); EDGE: the step
  // may be the first or the last of the row (entries -1 and W1 lie outside the volume)
  auto step = [&](auto edge_tag, const int u, const int i) {
    constexpr bool EDGE = decltype(edge_tag)::value;
    unsigned long long craw = cbuf[u];
    unsigned inA = zero2, inB = zero2, inM = 0;
    unsigned long long r1 = 0, r2 = 0, r3 = 0;
    if (IN == AGP_IN_LDS) {
      // the row before's step counter, the entry and the minima requested together: LDS serves a wave's requests in
      // order, so an entry read behind a counter that is high enough is valid -- one wait instead of two
      const unsigned need = EDGE ? (unsigned)min(i + 2, W1) : (unsigned)(i + 2);
      const int slot = (i + qm2) & (AGP_RING - 1);   // entry i - 2 + q: the one this lane's path continues
      const unsigned pr = *R.prog_in;
      unsigned long long v = R.ring_in[slot * 64 + lane];
      inM = R.minring_in[slot * 4 + q];
      if (!EDGE || i > 0) sum_read(i - 1, r1, r2, r3);
      if (seen_in < need) {
        seen_in = __builtin_amdgcn_readfirstlane(pr);
        if (seen_in < need) {
#ifdef KVFE_AGP_PROF
          const unsigned b0 = budget;
#endif
          seen_in = agp_wait_in(R.prog_in, R.ring_in + (slot * 64 + lane), R.minring_in + (slot * 4 + q), need, seen_in,
                                budget, v, inM);
#ifdef KVFE_AGP_PROF
          prof_in_turns += b0 - budget;
          prof_in_steps += b0 != budget ? 1 : 0;
#endif
        }
      }
      inA = (unsigned)v;
      inB = (unsigned)(v >> 32);
      if (EDGE && (i == 0 || i == last)) {   // wave-uniform
        const int e = i + qm2;
        const bool inr = e >= 0 && e <= last;
        inA = inr ? inA : zero2;
        inB = inr ? inB : zero2;
        inM = inr ? inM : 0u;
      }
    } else if (!EDGE || i > 0) {
      sum_read(i - 1, r1, r2, r3);
    }
    asm volatile("" : "+v"(craw));   // first use of the requested C: here, not earlier
    const unsigned cA = (unsigned)craw, cB = (unsigned)(craw >> 32);
    // (the next request after the use: issued before it, old and new value are alive together and the registers
    // rotate through copies at the loop's back edge -- behind a wait for every load)
    cbuf[u] = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(crsrc, coff, cnext, 0));
    cnext += cstep;
    // the sum of the step before (its entry has been in LDS for a step).  Behind the wait for C: the compiler cannot count
    // a store under a lane mask, so it keeps only PF - 1 operations in flight at that wait, stores included
    if (!EDGE || i > 0) sum_store(pA, pB, r1, r2, r3);
    inA = row0 ? pA : inA;
    inB = row0 ? pB : inB;
    inM = row0 ? pM : inM;
    // L(d) = C(d) + min(Lp(d), Lp(d-1) + P1, Lp(d+1) + P1, min Lp + P2) - (min Lp + P2), four disparities at a time
    const unsigned delta = agp_pk_add(inM, P2P2);
    nbp = (unsigned)__builtin_amdgcn_update_dpp((int)nbp, (int)inB, 0x111, 0xf, 0xf, false);   // row_shr:1
    nbn = (unsigned)__builtin_amdgcn_update_dpp((int)nbn, (int)inA, 0x101, 0xf, 0xf, false);   // row_shl:1
    const unsigned lmA = __builtin_amdgcn_alignbit(inA, nbp, 16);   // (L[4g-1], L[4g])
    const unsigned lmB = __builtin_amdgcn_alignbit(inB, inA, 16);   // (L[4g+1], L[4g+2])
    const unsigned lqB = __builtin_amdgcn_alignbit(nbn, inB, 16);   // (L[4g+3], L[4g+4])
    const unsigned tA = agp_pk_add(agp_pk_min(lmA, lmB), P1P1);
    const unsigned tB = agp_pk_add(agp_pk_min(lmB, lqB), P1P1);
    const unsigned mA = agp_pk_min(agp_pk_min(inA, delta), tA);
    const unsigned mB = agp_pk_min(agp_pk_min(inB, delta), tB);
    unsigned nA = agp_pk_add(cA, agp_pk_sub(mA, delta));
    unsigned nB = agp_pk_add(cB, agp_pk_sub(mB, delta));
    if (!all_active) {
      nA = actl ? nA : AGP_INF2;
      nB = actl ? nB : AGP_INF2;
    }
    unsigned x = agp_pk_min(nA, nB);
    x = agp_pk_min(x, __builtin_amdgcn_alignbit(x, x, 16));
    const unsigned nM = agp_row_min(x);
    // publish this row's entry i (in LDS always: the sum is formed from it)
    if (OUT == AGP_OUT_LDS) {
      // the slot's previous entry (i - RING) is read last at the reader's step i - RING + 1 (signed: nothing to wait for
      // during the first RING - 2 steps)
      const int need = i - AGP_RING + 2;
      if ((int)seen_out < need) {
#ifdef KVFE_AGP_PROF
        const unsigned b0 = budget;
#endif
        seen_out = agp_wait(R.prog_out, (unsigned)need, budget);
#ifdef KVFE_AGP_PROF
        prof_out_turns += b0 - budget;
        prof_out_steps += b0 != budget ? 1 : 0;
#endif
      }
    }
    const unsigned long long o = (unsigned long long)nA | ((unsigned long long)nB << 32);
    R.ring_out[(i & OM) * 64 + lane] = o;
    if (OUT == AGP_OUT_LDS) {
      // lane 0 of every row writes the row's minimum (the lane mask set and reset by hand: every lane is active here, and
      // the compiler's version is a saved mask, a branch around the store and a restore)
      const unsigned maddr = (unsigned)(size_t)(R.minring_out + (i & OM) * 4) + moff;
      asm volatile("s_mov_b64 exec, %2\n\tds_write_b32 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(maddr), "v"(nM),
                   "s"(0x0001000100010001ull)
                   : "memory");
    }
    if (OUT == AGP_OUT_GLOBAL) {
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(hptr + hoff), o | etag, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      if (min_lane)
        __hip_atomic_store(reinterpret_cast<unsigned*>(hmptr + moff), nM | etag, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      hptr += 512;
      hmptr += 16;
    }
    progv += 1u;
    asm volatile("s_mov_b64 exec, 1\n\tds_write_b32 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(progaddr),
                 "v"(progv)
                 : "memory");
    pA = nA;
    pB = nB;
    pM = nM;
  };
  // groups of PF steps without a condition in between (a conditional update of the request registers costs copies
  // at the top of the next step, and a copy is a use: the wait for the load moves up to it).  The first group and the
  // last one or two carry the tests for the row's ends; the groups in between do not.
  auto edge_group = [&](int i0) {
#pragma unroll
    for (int u = 0; u < PF; u++)
      if (i0 + u <= last) step(std::true_type{}, u, i0 + u);
  };
  // (the first group without the test for the row's end -- the launch needs W1 > PF: with conditional requests in
  // front of the loop the compiler no longer knows how many are in flight at its top and waits for all of them)
  int i0 = PF;
#pragma unroll
  for (int u = 0; u < PF; u++) step(std::true_type{}, u, u);
  for (; i0 + PF <= last; i0 += PF) {   // every step of the group has 0 < i < last
#pragma unroll
    for (int u = 0; u < PF; u++) step(std::false_type{}, u, i0 + u);
  }
  for (; i0 <= last; i0 += PF) edge_group(i0);
  {
    unsigned long long r1, r2, r3;
    sum_read(last, r1, r2, r3);
    sum_store(pA, pB, r1, r2, r3);
  }
#ifndef KVFE_AGP_FREERUN
  if (budget == 0 && lane == 0) atomicOr(R.err, 1u);
#endif
#ifdef KVFE_AGP_PROF
  if (lane == 0) {
    const long long t1 = wall_clock64();
    unsigned* st = R.err + 1;   // sync[2..]
    const unsigned dur = (unsigned)(t1 - prof_t0);
    atomicAdd(st + 0, 1u);
    atomicAdd(st + 1, dur);
    atomicMax(st + 2, dur);
    atomicAdd(st + 3, prof_in_turns);
    atomicAdd(st + 4, prof_out_turns);
    atomicAdd(st + 5, prof_in_steps);
    atomicAdd(st + 6, prof_out_steps);
    atomicMin(st + 7, (unsigned)prof_t0);
    atomicMax(st + 8, (unsigned)t1);
    if (IN == AGP_IN_ZERO) {
      atomicAdd(st + 11, dur);
      atomicAdd(st + 12, 1u);
    }
  }
#endif
}

// the loader wave of a block: the entries the block before wrote to HBM -> the ring the block's first row reads
__device__ __forceinline__ void agp_loader(const char* hin, const char* hmin_in, agp_lds_u64* ring_out,
                                           agp_lds_u32* minring_out, agp_lds_u32* prog_me, agp_lds_u32* prog_out,
                                           unsigned* err, int W1, unsigned etag) {
  const int lane = threadIdx.x & 63;
  const int last = W1 - 1;
  const unsigned hoff = 8u * (unsigned)lane, moff = 4u * (unsigned)(lane & 3);
  unsigned budget = AGP_WAIT_TURNS;
  unsigned seen_out = 0;
  auto h_at = [&](int e) {
    return __hip_atomic_load(reinterpret_cast<unsigned long long*>(const_cast<char*>(hin) + ((size_t)e * 512u + hoff)),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto hm_at = [&](int e) {   // (all lanes load one of the four words: no lane predicate around the request)
    return __hip_atomic_load(reinterpret_cast<unsigned*>(const_cast<char*>(hmin_in) + ((size_t)e * 16u + moff)),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
#ifdef KVFE_AGP_PROF
  unsigned prof_turns = 0, prof_steps = 0;
#endif
  unsigned long long gL[AGP_LPF];
  unsigned gM[AGP_LPF];
#pragma unroll
  for (int u = 0; u < AGP_LPF; u++) {
    gL[u] = h_at(min(u, last));
    gM[u] = hm_at(min(u, last));
  }
  auto step = [&](const int u, const int j) {
    unsigned long long gl = gL[u];
    unsigned gm = gM[u];
    asm volatile("" : "+v"(gl), "+v"(gm));   // first use of the requested words: here, not earlier
    gL[u] = h_at(min(j + AGP_LPF, last));
    gM[u] = hm_at(min(j + AGP_LPF, last));
    // word 0 of the minima is not read by anyone (row 0 of a wave is the horizontal path)
#ifdef KVFE_AGP_PROF
    const unsigned b0 = budget;
#endif
#ifdef KVFE_AGP_FREERUN
    budget = 0;
#endif
    while (__builtin_amdgcn_ballot_w64((((unsigned)gl & AGP_TAGMASK) != etag) ||
                                       ((lane & 3) != 0 && (gm & AGP_TAGMASK) != etag)) != 0ull &&
           budget != 0) {
      __builtin_amdgcn_s_sleep(1);
      budget--;
      gl = h_at(j);
      gm = hm_at(j);
    }
#ifdef KVFE_AGP_PROF
    prof_turns += b0 - budget;
    prof_steps += b0 != budget ? 1 : 0;
#endif
    if (j >= AGP_RING - 1) {   // as in agp_row: entry j - RING is read last at the reader's step j - RING + 1
      const unsigned need = (unsigned)(j - AGP_RING + 2);
      if (seen_out < need) seen_out = agp_wait(prog_out, need, budget);
    }
    const int slot = j & (AGP_RING - 1);
    ring_out[slot * 64 + lane] = gl & ~(unsigned long long)AGP_TAGMASK;
    if (lane < 4) minring_out[slot * 4 + lane] = gm & ~AGP_TAGMASK;
    if (lane == 0) *prog_me = (unsigned)j + 1u;
  };
  int j0 = 0;
  for (; j0 + AGP_LPF <= W1; j0 += AGP_LPF) {
#pragma unroll
    for (int u = 0; u < AGP_LPF; u++) step(u, j0 + u);
  }
#pragma unroll
  for (int u = 0; u < AGP_LPF; u++)
    if (j0 + u <= last) step(u, j0 + u);
#ifndef KVFE_AGP_FREERUN
  if (budget == 0 && lane == 0) atomicOr(err, 1u);
#endif
#ifdef KVFE_AGP_PROF
  if (lane == 0) {
    atomicAdd(err + 1 + 9, prof_turns);
    atomicAdd(err + 1 + 10, prof_steps);
  }
#endif
}

#ifdef KVFE_AGP_PROF
// profiling build only: the launch's counters -> the first words of the C volume (read with kvfe_dense_debug_volume)
__global__ void agp_prof_copy_kernel(const unsigned* sync, unsigned* out) {
  if (threadIdx.x < 16) out[threadIdx.x] = sync[2 + threadIdx.x];
}
#endif

// (eight waves per SIMD = two blocks per CU: 64 registers)
__global__ __launch_bounds__(AGP_WAVES * 64) void dense_aggregate_pass_kernel(
    DenseParams P, const short* __restrict__ Cv, unsigned short* __restrict__ sumA, unsigned short* __restrict__ sumB,
    unsigned long long* __restrict__ hand, unsigned* __restrict__ hand_min, unsigned* __restrict__ sync, int n,
    int cap_pairs, int nbands, unsigned epoch) {
  // ring k: written by wave k, read by wave k + 1; wave 15 has two scratch slots behind the rings
  __shared__ unsigned long long ring_s[((AGP_WAVES - 1) * AGP_RING + 2) * 64];
  __shared__ unsigned minring_s[(AGP_WAVES - 1) * AGP_RING * 4];
  __shared__ unsigned prog_s[AGP_WAVES];
  __shared__ unsigned ticket_s;
  // (LDS-qualified pointers: through a generic pointer a volatile access stays a flat one)
  agp_lds_u64* const ring = (agp_lds_u64*)ring_s;
  agp_lds_u32* const minring = (agp_lds_u32*)minring_s;
  agp_lds_u32* const prog = (agp_lds_u32*)prog_s;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (threadIdx.x == 0) ticket_s = atomicAdd(&sync[0], 1u);
  if (threadIdx.x < AGP_WAVES) prog_s[threadIdx.x] = 0;
  __syncthreads();
  const int t = __builtin_amdgcn_readfirstlane((int)ticket_s);
  const int band = t / (2 * n), rem = t - band * 2 * n;
  const int pair = rem >> 1;
  const bool down = (rem & 1) == 0;
  const int W1 = P.width1, H = P.H, D = P.D;
  if (band >= nbands) return;
  const unsigned etag = ((epoch & 1u) << 15) | ((epoch & 2u) << 30);
  // hand-over words: [pass][pair][boundary][step] x 64 lanes (entries) / x 4 (minima)
  const size_t hpp = (size_t)max(nbands - 1, 1) * W1;
  const size_t hbase = ((size_t)(down ? 0 : 1) * cap_pairs + pair) * hpp;
  const size_t hin = hbase + (size_t)max(band - 1, 0) * W1, hout = hbase + (size_t)min(band, max(nbands - 2, 0)) * W1;
  if (w == 0) {
    if (band > 0)
      agp_loader(reinterpret_cast<const char*>(hand + hin * 64), reinterpret_cast<const char*>(hand_min + hin * 4), ring,
                 minring, prog, prog + 1, sync + 1, W1, etag);
    return;
  }
  const int r = band * AGP_ROWS + (w - 1);   // row in the order of the pass
  if (r >= H) return;
  const int y = down ? r : H - 1 - r;
  const int in_kind = r == 0 ? AGP_IN_ZERO : AGP_IN_LDS;
  const int out_kind = r == H - 1 ? AGP_OUT_NONE : (w == AGP_WAVES - 1 ? AGP_OUT_GLOBAL : AGP_OUT_LDS);
  const size_t rowbase = (((size_t)pair * H + y) * W1) * D;
  AgpRow R;
  R.cp = reinterpret_cast<const char*>(Cv + rowbase);
  R.sp = reinterpret_cast<char*>((down ? sumA : sumB) + rowbase);
  R.hout = reinterpret_cast<char*>(hand + hout * 64);
  R.hmin_out = reinterpret_cast<char*>(hand_min + hout * 4);
  R.ring_in = ring + (size_t)(w - 1) * AGP_RING * 64;
  R.minring_in = minring + (size_t)(w - 1) * AGP_RING * 4;
  R.ring_out = ring + (size_t)w * AGP_RING * 64;           // wave 15: the two scratch slots behind the rings
  R.minring_out = minring + (size_t)min(w, AGP_WAVES - 2) * AGP_RING * 4;
  R.prog_in = prog + (w - 1);
  R.prog_me = prog + w;
  R.prog_out = prog + min(w + 1, AGP_WAVES - 1);
  R.err = sync + 1;
  R.W1 = W1;
  R.D = D;
  R.P1 = P.P1;
  R.P2 = P.P2;
  R.etag = etag;
  R.down = down;
  const bool full = D == 64;
#define AGP_CASE(I, O)                               \
  case I * 3 + O:                                    \
    if (full) agp_row<I, O, true>(R);                \
    else agp_row<I, O, false>(R);                    \
    break;
  switch (in_kind * 3 + out_kind) {   // wave-uniform
    AGP_CASE(AGP_IN_ZERO, AGP_OUT_NONE)
    AGP_CASE(AGP_IN_ZERO, AGP_OUT_LDS)
    AGP_CASE(AGP_IN_ZERO, AGP_OUT_GLOBAL)
    AGP_CASE(AGP_IN_LDS, AGP_OUT_NONE)
    AGP_CASE(AGP_IN_LDS, AGP_OUT_LDS)
    default:
      if (full) agp_row<AGP_IN_LDS, AGP_OUT_GLOBAL, true>(R);
      else agp_row<AGP_IN_LDS, AGP_OUT_GLOBAL, false>(R);
      break;
  }
#undef AGP_CASE
}

// ---- disparity selection (computeDisparitySGBM, pass == npasses block) ---------------------------------
constexpr int SEL_MAXW = 2048;
// Block = one image row.  Each wave takes tiles of 64 columns: it reads the tile with lane = disparity
// (one 128-byte line per column), transposes it through LDS, and then works with lane = column, so that the
// minimum search, the uniqueness test, the parabola fit and the integer division run on 64 pixels at once.
constexpr int SEL_PITCH = 66;   // halfwords per disparity row of the transposed tile (bank-conflict padding)
__global__ __launch_bounds__(256) void dense_select_kernel(DenseParams P, const unsigned short* __restrict__ sum,
                                                           const unsigned short* __restrict__ sum2,
                                                           short* __restrict__ disp) {
  __shared__ unsigned long long d2key[SEL_MAXW];
  __shared__ short d1[SEL_MAXW];
  __shared__ unsigned short tile[4][64 * SEL_PITCH];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int y = blockIdx.x, pair = blockIdx.y;
  const int W = P.W, W1 = P.width1, D = P.D;
  const short INVALID = (short)P.invalid_scaled;
  for (int x = threadIdx.x; x < W; x += 256) {
    d2key[x] = ~0ull;
    d1[x] = INVALID;
  }
  __syncthreads();
  const size_t row = (((size_t)pair * P.H + y) * W1) * D;
  const unsigned short* srow = sum + row + min(lane, D - 1);
  const unsigned short* srow2 = sum2 ? sum2 + row + min(lane, D - 1) : nullptr;
  unsigned short* T = tile[wv];
  const int ntiles = (W1 + 63) / 64;
  for (int tl = wv; tl < ntiles; tl += 4) {
    const int x0 = tl * 64;
    // lane = disparity: column j of the tile -> T[lane][j]
    if (srow2) {   // two partial sums (single-launch aggregation): each < 65536, their sum saturates here
#pragma unroll 16
      for (int j = 0; j < 64; j++) {
        const size_t o = (size_t)min(x0 + j, W1 - 1) * D;
        T[lane * SEL_PITCH + j] = (unsigned short)min((int)srow[o] + (int)srow2[o], 65535);
      }
    } else {
#pragma unroll 16
      for (int j = 0; j < 64; j++) T[lane * SEL_PITCH + j] = srow[(size_t)min(x0 + j, W1 - 1) * D];
    }
    // (one wave owns the tile: the LDS writes above are complete before the reads below are issued in order)
    __builtin_amdgcn_wave_barrier();
    const int x = x0 + lane;
    // lane = column
    int minS = MAXC, best = -1;
    for (int d = 0; d < D; d++) {
      const int S = min(MAXC, (int)T[d * SEL_PITCH + lane]);   // saturate_cast<short>
      if (S < minS) {
        minS = S;
        best = d;
      }
    }
    bool ok = x < W1 && best >= 0;   // best = -1: every S saturated, the pixel keeps INVALID_DISP_SCALED
    {   // uniqueness: any other disparity (|d - best| > 1) whose cost is within the ratio invalidates
      bool viol = false;
      for (int d = 0; d < D; d++) {
        const int S = min(MAXC, (int)T[d * SEL_PITCH + lane]);
        viol |= (S * (100 - P.uniq) < minS * 100) && abs(best - d) > 1;
      }
      ok = ok && !viol;
    }
    if (ok) {
      const int x2 = x + P.minX1 - best - P.minD;
      // scanning x downwards, a strictly smaller cost replaces: lowest cost, then largest x
      atomicMin(&d2key[x2], ((unsigned long long)minS << 32) | ((unsigned long long)(0xFFFF - x) << 16) |
                                (unsigned long long)best);
      int d;
      if (0 < best && best < D - 1) {
        const int Sm = min(MAXC, (int)T[(best - 1) * SEL_PITCH + lane]);
        const int Sq = min(MAXC, (int)T[(best + 1) * SEL_PITCH + lane]);
        const int denom2 = max(Sm + Sq - 2 * minS, 1);
        d = best * DISP_SCALE + ((Sm - Sq) * DISP_SCALE + denom2) / (denom2 * 2);
      } else
        d = best * DISP_SCALE;
      d1[x + P.minX1] = (short)(d + P.minD * DISP_SCALE);
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  short* out = disp + ((size_t)pair * P.H + y) * W;
  for (int x = threadIdx.x; x < W; x += 256) {
    int dv = d1[x];
    if (x >= P.minX1 && x < P.minX1 + W1 && dv != INVALID) {
      const int _d = dv >> DISP_SHIFT, d_ = (dv + DISP_SCALE - 1) >> DISP_SHIFT;
      const int _x = x - _d, x_ = x - d_;
      auto disp2 = [&](int xx) {
        const unsigned long long k = d2key[xx];
        // untouched entries hold INVALID_DISP_SCALED like OpenCV's disp2ptr (so for minDisparity >= 2
        // they pass the ">= minD" test below, as they do upstream)
        return k == ~0ull ? P.invalid_scaled : (int)(k & 0xFFFF) + P.minD;
      };
      bool bad = false;
      if (0 <= _x && _x < W && 0 <= x_ && x_ < W) {
        const int a = disp2(_x), b = disp2(x_);
        bad = a >= P.minD && abs(a - _d) > P.disp12 && b >= P.minD && abs(b - d_) > P.disp12;
      }
      if (bad) dv = INVALID;
    }
    out[x] = (short)dv;
  }
}

// ---------------------------------------------------------------------------------------------
// cv::StereoBM (stereobm.cpp, PREFILTER_XSOBEL, CV_16S): use_sgbm_ = false (StereoMatcher.cpp:66-90)
// ---------------------------------------------------------------------------------------------
// prefilterXSobel: rows reflect-101, first / last column and (odd H) the last row hold ftzero
__global__ __launch_bounds__(256) void bm_prefilter_kernel(DenseParams P, const uint8_t* __restrict__ left,
                                                           const uint8_t* __restrict__ right,
                                                           uint8_t* __restrict__ pre) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, img = blockIdx.z & 1, pair = blockIdx.z >> 1;
  if (x >= P.W) return;
  const size_t plane = (size_t)P.W * P.H;
  const uint8_t* src = (img ? right : left) + pair * plane;
  const int ft = P.ftzero;
  int v = ft;   // tab[0 + OFS]
  const bool last_odd_row = (P.H & 1) && y == P.H - 1;
  if (x > 0 && x < P.W - 1 && !last_odd_row) {
    const int ym = y > 0 ? y - 1 : (P.H > 1 ? 1 : 0), yp = y < P.H - 1 ? y + 1 : (P.H > 1 ? P.H - 2 : 0);
    const uint8_t *r0 = src + (size_t)ym * P.W, *r1 = src + (size_t)y * P.W, *r2 = src + (size_t)yp * P.W;
    const int g = (r0[x + 1] - r0[x - 1]) + (r1[x + 1] - r1[x - 1]) * 2 + (r2[x + 1] - r2[x - 1]);
    v = g < -ft ? 0 : g > ft ? ft * 2 : g + ft;
  }
  pre[((size_t)blockIdx.z * P.H + y) * P.W + x] = (uint8_t)v;
}

// hsad(y, x, d) = sum over the window columns c = x-wsz2 .. x+wsz2 of |L'(y, clampL(c)) - R'(y, rbase(c) + d)|
// with findStereoCorrespondenceBM's own column clamps (left pixel clamped, right BASE clamped); d indexes
// disparity ndisp - 1 - d + mindisp.  wave = pixel, lane = d.
__global__ __launch_bounds__(256) void bm_hsad_kernel(DenseParams P, const uint8_t* __restrict__ pre,
                                                      short* __restrict__ hs) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int y = blockIdx.y, pair = blockIdx.z;
  const uint8_t* L = pre + ((size_t)(pair * 2) * P.H + y) * P.W;
  const uint8_t* R = pre + ((size_t)(pair * 2 + 1) * P.H + y) * P.W;
  short* out = hs + (((size_t)pair * P.H + y) * P.width1) * P.D;
  const int lofs = P.minX1, rofs = P.bm_rofs, wsz2 = P.SW2, nd = P.D;
  const int dl = min(lane, nd - 1);
  for (int i = 0; i < BT_XPB / 4; i++) {
    const int x = blockIdx.x * BT_XPB + wv * (BT_XPB / 4) + i;
    if (x >= P.width1) break;
    int s = 0;
    for (int c = x - wsz2; c <= x + wsz2; c++) {
      const int lv = L[lofs + min(max(c, -lofs), P.W - 1 - lofs)];
      const int rv = R[rofs + min(max(c, -rofs), P.W - nd - rofs) + dl];
      s += abs(lv - rv);
    }
    if (lane < nd) out[(size_t)x * nd + lane] = (short)s;
  }
}

// texture sum of findStereoCorrespondenceBM: sum over the window of |L'(clamped) - ftzero|
__global__ __launch_bounds__(256) void bm_texture_kernel(DenseParams P, const uint8_t* __restrict__ pre,
                                                         int* __restrict__ tex) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, pair = blockIdx.z;
  if (x >= P.width1) return;
  const uint8_t* L = pre + ((size_t)(pair * 2) * P.H) * P.W;
  int s = 0;
  for (int k = -P.SW2; k <= P.SW2; k++) {
    const uint8_t* row = L + (size_t)min(max(y + k, 0), P.H - 1) * P.W;
    for (int c = x - P.SW2; c <= x + P.SW2; c++)
      s += abs((int)row[P.minX1 + min(max(c, -P.minX1), P.W - 1 - P.minX1)] - P.ftzero);
  }
  tex[((size_t)pair * P.H + y) * P.W + x] = s;
}

// minimum SAD, texture / uniqueness tests, sub-pixel fit (the tail of findStereoCorrespondenceBM's y loop),
// FILTERED outside the valid-disparity rectangle (FindStereoCorrespInvoker).  Same tile transposition as
// dense_select_kernel: lane = d while loading, lane = column while deciding.
__global__ __launch_bounds__(256) void bm_select_kernel(DenseParams P, const short* __restrict__ sad,
                                                        const int* __restrict__ tex, short* __restrict__ disp) {
  __shared__ unsigned short tile[4][64 * SEL_PITCH];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int y = blockIdx.x, pair = blockIdx.y;
  const int W = P.W, W1 = P.width1, D = P.D;
  const short FILTERED = (short)P.invalid_scaled;
  short* out = disp + ((size_t)pair * P.H + y) * W;
  for (int x = threadIdx.x; x < W; x += 256) out[x] = FILTERED;
  if (y < P.bm_roi[1] || y >= P.bm_roi[1] + P.bm_roi[3]) return;
  __syncthreads();
  const size_t row = (((size_t)pair * P.H + y) * W1) * D;
  const short* srow = sad + row + min(lane, D - 1);
  unsigned short* T = tile[wv];
  const int ntiles = (W1 + 63) / 64;
  for (int tl = wv; tl < ntiles; tl += 4) {
    const int x0 = tl * 64;
#pragma unroll 16
    for (int j = 0; j < 64; j++) T[lane * SEL_PITCH + j] = (unsigned short)srow[(size_t)min(x0 + j, W1 - 1) * D];
    __builtin_amdgcn_wave_barrier();
    const int x = x0 + lane;
    int minsad = 0x7fffffff, mind = -1;
    for (int d = 0; d < D; d++) {
      const int v = T[d * SEL_PITCH + lane];
      if (v < minsad) {
        minsad = v;
        mind = d;
      }
    }
    const int xo = P.minX1 + x;   // output column
    bool ok = x < W1 && xo >= P.bm_roi[0] && xo < P.bm_roi[0] + P.bm_roi[2];
    if (ok && P.bm_texture > 0) ok = tex[((size_t)pair * P.H + y) * W + x] >= P.bm_texture;
    if (ok && P.uniq > 0) {
      const int thresh = minsad + (minsad * P.uniq / 100);
      bool viol = false;
      for (int d = 0; d < D; d++) {
        const int v = T[d * SEL_PITCH + lane];
        viol |= (d < mind - 1 || d > mind + 1) && v <= thresh;
      }
      ok = !viol;
    }
    if (ok) {
      // sad[-1] = sad[1]; sad[ndisp] = sad[ndisp-2]
      const int ip = mind + 1 < D ? mind + 1 : D - 2, in = mind - 1 >= 0 ? mind - 1 : 1;
      const int p = T[ip * SEL_PITCH + lane], n = T[in * SEL_PITCH + lane];
      const int dd = p + n - 2 * minsad + abs(p - n);
      out[xo] = (short)(((D - mind - 1 + P.minD) * 256 + (dd != 0 ? (p - n) * 256 / dd : 0) + 15) >> 4);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- cv::medianBlur on CV_16S (replicated borders) -----------------------------------------------
__device__ __forceinline__ void cswap(int& a, int& b) {
  const int t = min(a, b);
  b = max(a, b);
  a = t;
}
__global__ __launch_bounds__(256) void dense_median3_kernel(int W, int H, const short* __restrict__ src,
                                                            short* __restrict__ dst) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const short* s = src + (size_t)blockIdx.z * W * H;
  int v[9];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const int yy = min(max(y + j - 1, 0), H - 1);
#pragma unroll
    for (int i = 0; i < 3; i++) v[j * 3 + i] = s[(size_t)yy * W + min(max(x + i - 1, 0), W - 1)];
  }
  // 19-exchange median-of-9 network
  cswap(v[1], v[2]); cswap(v[4], v[5]); cswap(v[7], v[8]);
  cswap(v[0], v[1]); cswap(v[3], v[4]); cswap(v[6], v[7]);
  cswap(v[1], v[2]); cswap(v[4], v[5]); cswap(v[7], v[8]);
  cswap(v[0], v[3]); cswap(v[5], v[8]); cswap(v[4], v[7]);
  cswap(v[3], v[6]); cswap(v[1], v[4]); cswap(v[2], v[5]);
  cswap(v[4], v[7]); cswap(v[4], v[2]); cswap(v[6], v[4]);
  cswap(v[4], v[2]);
  dst[(size_t)blockIdx.z * W * H + (size_t)y * W + x] = (short)v[4];
}
__global__ __launch_bounds__(256) void dense_median5_kernel(int W, int H, const short* __restrict__ src,
                                                            short* __restrict__ dst) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const short* s = src + (size_t)blockIdx.z * W * H;
  int v[25];
#pragma unroll
  for (int j = 0; j < 5; j++) {
    const int yy = min(max(y + j - 2, 0), H - 1);
#pragma unroll
    for (int i = 0; i < 5; i++) v[j * 5 + i] = s[(size_t)yy * W + min(max(x + i - 2, 0), W - 1)];
  }
  // partial selection: 13 rounds of "move the minimum of the remaining elements to the front"
#pragma unroll
  for (int k = 0; k < 13; k++)
#pragma unroll
    for (int i = 24; i > k; i--) cswap(v[i - 1], v[i]);
  dst[(size_t)blockIdx.z * W * H + (size_t)y * W + x] = (short)v[12];
}

// ---- cv::filterSpeckles as connected-component labelling ------------------------------------------------
// A pixel is removed iff its 4-connected component (edges between valid neighbours whose values differ by
// at most maxDiff) has at most maxSpeckleSize pixels — the flood fill's result does not depend on scan order.
// Step 1: block = row: every pixel gets the index of the first pixel of its horizontal run.
__global__ __launch_bounds__(256) void speckle_rows_kernel(int W, int H, int newVal, int maxDiff,
                                                           const short* __restrict__ disp, int* __restrict__ label,
                                                           int* __restrict__ count, int* __restrict__ runlen) {
  __shared__ int start[SEL_MAXW];
  __shared__ int rl[SEL_MAXW];
  __shared__ int wave_tot[4];
  const int y = blockIdx.x;
  const size_t img = (size_t)blockIdx.y * W * H;
  const short* d = disp + img + (size_t)y * W;
  const int per = (W + 255) / 256;
  const int x0 = min((int)threadIdx.x * per, W), x1 = min(x0 + per, W);
  // start[x] = running maximum of (x if a run starts at x else -1): inclusive prefix max over the row
  int run = -1;
  for (int x = x0; x < x1; x++) {
    const int v = d[x];
    const bool valid = v != newVal;
    const bool joins = valid && x > 0 && d[x - 1] != newVal && abs(v - d[x - 1]) <= maxDiff;
    if (valid && !joins) run = x;
    start[x] = run;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int inc = run;
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(inc, off);
    if (lane >= off) inc = max(inc, o);
  }
  if (lane == 63) wave_tot[wv] = inc;
  __syncthreads();
  int pre = __shfl_up(inc, 1);
  if (lane == 0) pre = -1;
  for (int k = 0; k < wv; k++) pre = max(pre, wave_tot[k]);
  for (int x = x0; x < x1; x++) {
    start[x] = max(start[x], pre);
    rl[x] = 0;
  }
  __syncthreads();
  for (int x = threadIdx.x; x < W; x += 256)
    if (d[x] != newVal) atomicAdd(&rl[start[x]], 1);
  __syncthreads();
  for (int x = threadIdx.x; x < W; x += 256) {
    const size_t i = img + (size_t)y * W + x;
    label[i] = d[x] != newVal ? y * W + start[x] : -1;
    count[i] = 0;
    runlen[i] = rl[x];   // pixels of the run, at its first pixel; 0 elsewhere
  }
}
__device__ __forceinline__ int cc_find(const int* L, int i) {
  int p;
  while ((p = __hip_atomic_load(&L[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != i) i = p;
  return i;
}
// find with path halving, for the launches BEHIND the merge (no union runs beside them): an entry is only ever replaced by
// an ancestor of its own tree, so every concurrent reader still reaches the same root, and the trees -- as deep as a
// component is tall, one link per image row -- flatten while the launch walks them (round 6: the count and apply launches
// of a single 752 x 480 pair took 43 + 23 us of dependent loads)
__device__ __forceinline__ int cc_find_halving(int* L, int i) {
  for (;;) {
    const int p = __hip_atomic_load(&L[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p == i) return i;
    const int g = __hip_atomic_load(&L[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (g == p) return p;
    __hip_atomic_store(&L[i], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    i = g;
  }
}
__device__ void cc_union(int* L, int a, int b) {
  for (;;) {
    a = cc_find(L, a);
    b = cc_find(L, b);
    if (a == b) return;
    if (a < b) {
      const int t = a;
      a = b;
      b = t;
    }
    const int old = atomicMin(&L[a], b);
    if (old == a) return;
    a = old;
  }
}
// Step 2: vertical links (only where a run changes on either row, or at the first column of a run pair)
__global__ __launch_bounds__(256) void speckle_merge_kernel(int W, int H, int newVal, int maxDiff,
                                                            const short* __restrict__ disp, int* __restrict__ label) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= W || y >= H - 1) return;
  const size_t img = (size_t)blockIdx.z * W * H;
  const short* d = disp + img;
  int* L = label + img;
  const int i = y * W + x;
  const int v = d[i], u = d[i + W];
  if (v == newVal || u == newVal || abs(v - u) > maxDiff) return;
  // skip links already implied by the left neighbour's link (same runs above and below, and linked)
  if (x > 0) {
    const int vl = d[i - 1], ul = d[i + W - 1];
    if (vl != newVal && ul != newVal && abs(vl - ul) <= maxDiff && abs(v - vl) <= maxDiff && abs(u - ul) <= maxDiff)
      return;
  }
  cc_union(L, i, i + W);
}
// Step 3: component sizes, one atomic per horizontal run
__global__ __launch_bounds__(256) void speckle_count_kernel(int W, int H, int* __restrict__ label,
                                                            const int* __restrict__ runlen, int* __restrict__ count) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const size_t img = (size_t)blockIdx.z * W * H;
  const int i = y * W + x;
  const int n = runlen[img + i];
  if (n == 0) return;
  atomicAdd(&count[img + cc_find_halving(label + img, i)], n);
}
__global__ __launch_bounds__(256) void speckle_apply_kernel(int W, int H, int newVal, int maxSize,
                                                            int* __restrict__ label, const int* __restrict__ count,
                                                            short* __restrict__ disp) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const size_t img = (size_t)blockIdx.z * W * H;
  const size_t i = img + (size_t)y * W + x;
  const int r = label[i];
  if (r >= 0 && count[img + cc_find_halving(label + img, r)] <= maxSize) disp[i] = (short)newVal;
}

// cv::reprojectImageTo3D(CV_32F -> CV_32FC3, handleMissingValues = true)
__global__ __launch_bounds__(256) void dense_min_kernel(int n, const float* __restrict__ disp, unsigned* __restrict__ out) {
  // order-preserving map float -> unsigned, atomicMin
  float m = 3.402823466e+38f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) m = fminf(m, disp[i]);
  for (int off = 32; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0) {
    const unsigned b = __float_as_uint(m);
    atomicMin(out, (b & 0x80000000u) ? ~b : (b | 0x80000000u));
  }
}
__global__ __launch_bounds__(256) void dense_reproject_kernel(int W, int H, const float* __restrict__ disp,
                                                              const unsigned* __restrict__ minkey, ReprojectQ Q,
                                                              float* __restrict__ xyz) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const unsigned k = *minkey;
  const double minDisparity = (double)__uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
  const double d = disp[(size_t)y * W + x];
  const double v[4] = {(double)x, (double)y, d, 1.0};
  double hom[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    double s = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) s += Q.q[i * 4 + j] * v[j];
    hom[i] = s;
  }
  const double ialpha = 1. / hom[3];
  float* o = xyz + ((size_t)y * W + x) * 3;
  o[0] = (float)((double)(float)hom[0] * ialpha);
  o[1] = (float)((double)(float)hom[1] * ialpha);
  o[2] = fabs(d - minDisparity) <= 1.192092896e-07 ? 10000.f : (float)((double)(float)hom[2] * ialpha);
}

}  // namespace

size_t dense_volume_elems(const DenseParams& P) { return (size_t)P.H * P.width1 * P.D; }

// bytes of the two-pass aggregation's block-to-block entries (0: that path is not used for this call)
size_t dense_handoff_bytes(const DenseParams& P, int pairs) {
  if (!P.full_dp || P.bm || pairs < AGP_MIN_PAIRS || P.width1 <= AGP_PF_G) return 0;
  const size_t nbands = (size_t)(P.H + AGP_ROWS - 1) / AGP_ROWS;
  // per step of a block boundary: the entry (64 lanes x 8 bytes) and the three minima (4 words)
  return std::max<size_t>(1, nbands - 1) * 2 * pairs * P.width1 * (64 * sizeof(unsigned long long) + 4 * sizeof(unsigned));
}

void launch_dense_sgbm(const DenseParams& P, DenseBuffers& B, int n, hipStream_t st, bool two_pass_allowed) {
  const dim3 blk(256);
  dense_prefilter_kernel<<<dim3((P.W + 255) / 256, P.H, 2 * n), blk, 0, st>>>(P, B.left, B.right, B.rec);
  if (P.full_dp && P.SW2 >= 1 && P.SW2 <= 5) {
    // rows per wave: enough waves for the chip when few pairs are in the call, less halo work when many are
    const int RC = n >= 4 ? 48 : 16;
    const dim3 g((P.width1 + CF_NCOL - 1) / CF_NCOL, (P.H + RC - 1) / RC, n);
    switch (P.SW2) {
      case 1: dense_cost_fused_kernel<1><<<g, dim3(64), 0, st>>>(P, B.rec, B.vol[2], RC); break;
      case 2: dense_cost_fused_kernel<2><<<g, dim3(64), 0, st>>>(P, B.rec, B.vol[2], RC); break;
      case 3: dense_cost_fused_kernel<3><<<g, dim3(64), 0, st>>>(P, B.rec, B.vol[2], RC); break;
      case 4: dense_cost_fused_kernel<4><<<g, dim3(64), 0, st>>>(P, B.rec, B.vol[2], RC); break;
      default: dense_cost_fused_kernel<5><<<g, dim3(64), 0, st>>>(P, B.rec, B.vol[2], RC); break;
    }
  } else {
    dense_bt_cost_kernel<<<dim3((P.width1 + BT_XPB - 1) / BT_XPB, P.H, n), blk, 0, st>>>(P, B.rec, B.vol[0]);
    dense_hsum_kernel<<<dim3((P.width1 + 4 * HS_CHUNK - 1) / (4 * HS_CHUNK), P.H, n), blk, 0, st>>>(P, B.vol[0],
                                                                                                   B.vol[1]);
    dense_vsum_kernel<<<dim3((P.width1 + 3) / 4, (P.H + VS_CHUNK - 1) / VS_CHUNK, n), blk, 0, st>>>(P, B.vol[1],
                                                                                                   B.vol[2]);
  }
  const short* Cv = B.vol[2];
  unsigned short* sA = (unsigned short*)B.vol[0];
  unsigned short* sB = (unsigned short*)B.vol[1];
  const int nh = (P.H + 3) / 4, nw = (P.width1 + 3) / 4, nd = (P.width1 + P.H - 1 + 3) / 4;
  if (two_pass_allowed && P.full_dp && n >= AGP_MIN_PAIRS && P.width1 > AGP_PF_G && B.hand &&
      B.hand_bytes >= dense_handoff_bytes(P, B.cap_pairs)) {
    // computeDisparitySGBM's two passes, one launch: rows of a pass are waves that hand their path costs down
    const int nbands = (P.H + AGP_ROWS - 1) / AGP_ROWS;
    const int key[4] = {n, P.width1, P.H, P.D};
    if (memcmp(key, B.hand_key, sizeof(key)) != 0 || B.hand_epoch == 0) {
      // entries of another layout could carry this launch's tag: start from a zeroed buffer
      (void)hipMemsetAsync(B.hand, 0, B.hand_bytes, st);
      memcpy(B.hand_key, key, sizeof(key));
      B.hand_epoch = 0;
    }
    B.hand_epoch = B.hand_epoch % 3 + 1;
    (void)hipMemsetAsync(B.agsync, 0, AGP_SYNC_WORDS * sizeof(unsigned), st);
#ifdef KVFE_AGP_PROF
    (void)hipMemsetAsync(B.agsync + 2 + 7, 0xff, sizeof(unsigned), st);   // the minimum of the start times
#endif
    // (the minima lie behind the entries of the context's pair capacity)
    unsigned* hand_min = reinterpret_cast<unsigned*>(
        B.hand + (size_t)std::max(1, nbands - 1) * 2 * B.cap_pairs * P.width1 * 64);
    dense_aggregate_pass_kernel<<<dim3(nbands * 2 * n), dim3(AGP_WAVES * 64), 0, st>>>(
        P, Cv, sA, sB, B.hand, hand_min, B.agsync, n, B.cap_pairs, nbands, B.hand_epoch);
    dense_select_kernel<<<dim3(P.H, n), blk, 0, st>>>(P, sA, sB, B.disp[0]);
#ifdef KVFE_AGP_PROF
    agp_prof_copy_kernel<<<1, 64, 0, st>>>(B.agsync, reinterpret_cast<unsigned*>(B.vol[2]));
#endif
  } else if (n <= 3 && P.full_dp && two_pass_allowed) {
    // few pairs: every direction of a pair in one launch, packed atomic adds into two zeroed volumes
    const size_t bytes = sizeof(short) * dense_volume_elems(P) * n;
    (void)hipMemsetAsync(sA, 0, bytes, st);
    (void)hipMemsetAsync(sB, 0, bytes, st);
    dense_aggregate_all_kernel<<<dim3(nd, 8, n), blk, 0, st>>>(P, Cv, sA, sB);
    dense_select_kernel<<<dim3(P.H, n), blk, 0, st>>>(P, sA, sB, B.disp[0]);
  } else {
    // pass 1 of computeDisparitySGBM: previous pixel at (x-1,y), (x-1,y-1), (x,y-1), (x+1,y-1)
    dense_aggregate_kernel<1, 0, true><<<dim3(nh, n), blk, 0, st>>>(P, Cv, sA);
    dense_aggregate_kernel<1, 1, false><<<dim3(nd, n), blk, 0, st>>>(P, Cv, sA);
    dense_aggregate_kernel<0, 1, false><<<dim3(nw, n), blk, 0, st>>>(P, Cv, sA);
    dense_aggregate_kernel<-1, 1, false><<<dim3(nd, n), blk, 0, st>>>(P, Cv, sA);
    // MODE_SGBM: the fifth direction, previous pixel at (x+1,y); MODE_HH pass 2: (x+1,y), (x-1,y+1), (x,y+1),
    // (x+1,y+1)
    dense_aggregate_kernel<-1, 0, false><<<dim3(nh, n), blk, 0, st>>>(P, Cv, sA);
    if (P.full_dp) {
      dense_aggregate_kernel<1, -1, false><<<dim3(nd, n), blk, 0, st>>>(P, Cv, sA);
      dense_aggregate_kernel<0, -1, false><<<dim3(nw, n), blk, 0, st>>>(P, Cv, sA);
      dense_aggregate_kernel<-1, -1, false><<<dim3(nd, n), blk, 0, st>>>(P, Cv, sA);
    }
    dense_select_kernel<<<dim3(P.H, n), blk, 0, st>>>(P, sA, nullptr, B.disp[0]);
  }
  const dim3 gpx((P.W + 255) / 256, P.H, n);
  dense_median3_kernel<<<gpx, blk, 0, st>>>(P.W, P.H, B.disp[0], B.disp[1]);
  short* cur = B.disp[1];
  short* other = B.disp[0];
  if (P.speckle_win > 0) {
    speckle_rows_kernel<<<dim3(P.H, n), blk, 0, st>>>(P.W, P.H, P.invalid_scaled, P.speckle_diff, cur, B.label,
                                                      B.count, B.runlen);
    speckle_merge_kernel<<<gpx, blk, 0, st>>>(P.W, P.H, P.invalid_scaled, P.speckle_diff, cur, B.label);
    speckle_count_kernel<<<gpx, blk, 0, st>>>(P.W, P.H, B.label, B.runlen, B.count);
    speckle_apply_kernel<<<gpx, blk, 0, st>>>(P.W, P.H, P.invalid_scaled, P.speckle_win, B.label, B.count, cur);
  }
  if (P.median5) {
    dense_median5_kernel<<<gpx, blk, 0, st>>>(P.W, P.H, cur, other);
    cur = other;
  }
  if (cur != B.disp[0])
    (void)hipMemcpyAsync(B.disp[0], cur, sizeof(short) * (size_t)P.W * P.H * n, hipMemcpyDeviceToDevice, st);
}

// cv::StereoBM::compute: inputs in B.left / B.right, result in B.disp[0].  Reuses the SGBM buffers: the
// pre-filtered images live in B.rec, hsad in vol[0], the window SADs in vol[1], the texture sums in B.count.
void launch_dense_bm(const DenseParams& P, const DenseBuffers& B, int n, hipStream_t st) {
  const dim3 blk(256);
  uint8_t* pre = reinterpret_cast<uint8_t*>(B.rec);
  bm_prefilter_kernel<<<dim3((P.W + 255) / 256, P.H, 2 * n), blk, 0, st>>>(P, B.left, B.right, pre);
  bm_hsad_kernel<<<dim3((P.width1 + BT_XPB - 1) / BT_XPB, P.H, n), blk, 0, st>>>(P, pre, B.vol[0]);
  dense_vsum_kernel<<<dim3((P.width1 + 3) / 4, (P.H + VS_CHUNK - 1) / VS_CHUNK, n), blk, 0, st>>>(P, B.vol[0],
                                                                                                 B.vol[1]);
  if (P.bm_texture > 0)
    bm_texture_kernel<<<dim3((P.width1 + 255) / 256, P.H, n), blk, 0, st>>>(P, pre, B.count);
  bm_select_kernel<<<dim3(P.H, n), blk, 0, st>>>(P, B.vol[1], B.count, B.disp[0]);
  const dim3 gpx((P.W + 255) / 256, P.H, n);
  short* cur = B.disp[0];
  if (P.speckle_win > 0) {
    speckle_rows_kernel<<<dim3(P.H, n), blk, 0, st>>>(P.W, P.H, P.invalid_scaled, P.speckle_diff, cur, B.label,
                                                      B.count, B.runlen);
    speckle_merge_kernel<<<gpx, blk, 0, st>>>(P.W, P.H, P.invalid_scaled, P.speckle_diff, cur, B.label);
    speckle_count_kernel<<<gpx, blk, 0, st>>>(P.W, P.H, B.label, B.runlen, B.count);
    speckle_apply_kernel<<<gpx, blk, 0, st>>>(P.W, P.H, P.invalid_scaled, P.speckle_win, B.label, B.count, cur);
  }
  if (P.median5) {
    dense_median5_kernel<<<gpx, blk, 0, st>>>(P.W, P.H, cur, B.disp[1]);
    (void)hipMemcpyAsync(B.disp[0], B.disp[1], sizeof(short) * (size_t)P.W * P.H * n, hipMemcpyDeviceToDevice, st);
  }
}

void launch_reproject_to_3d(int W, int H, const float* disp, const ReprojectQ& Q, unsigned* minkey, float* xyz,
                            hipStream_t st) {
  (void)hipMemsetAsync(minkey, 0xff, sizeof(unsigned), st);
  dense_min_kernel<<<dim3(64), dim3(256), 0, st>>>(W * H, disp, minkey);
  dense_reproject_kernel<<<dim3((W + 255) / 256, H), dim3(256), 0, st>>>(W, H, disp, minkey, Q, xyz);
}

}  // namespace kvfe
