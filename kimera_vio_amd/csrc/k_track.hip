// K4c  pyramidal Lucas-Kanade       == cv::calcOpticalFlowPyrLK (OPTFLOW_USE_INITIAL_FLOW)
//      reference: Tracker::featureTracking, src/frontend/Tracker.cpp:92-211 (call at :137-146)
//      + RotationalOpticalFlowPredictor::predictSparseFlow (optical-flow/OpticalFlowPredictor.cpp:70-126)
//      + survivor bookkeeping (Tracker.cpp:167-189) and VisionImuFrontend::shouldBeKeyframe
//        (src/frontend/VisionImuFrontend.cpp:175-232) evaluated on the device so that a frame
//        needs no host round trip.
//
// One wavefront per tracked point walks all pyramid levels.  Window / derivative arithmetic is
// integer (14-bit bilinear weights, x32 template scaling, Scharr int16) and therefore exact.  The
// float accumulators of the 2x2 system (A11,A12,A22 per level; b1,b2 per iteration) are summed in
// the lane order of OpenCV's x86-64 (SSE2) build: four/eight independent running float sums over
// (row, 4/8-pixel chunk), combined at the end — each chain is owned by one lane, so the tracked
// positions are bit-identical to the reference CPU front-end.
#include "kvfe_dev.hpp"

#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <vector>

namespace kvfe {

#include "kvfe_undistort.inl"

#include "kvfe_lk.inl"

// ---------------------------------------------------------------------------------------------
// Generic kernel (any window size): template / derivative windows and the per-iteration
// products go through LDS, the SSE-lane float chains are walked by lanes 0-14.  Used only when
// the systolic kernel below does not cover the window size.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void lk_kernel_generic(KParams P, const unsigned char* prev_img,
                                                        size_t prev_row_stride,
                                                        size_t prev_img_stride,
                                                        const unsigned char* prev_pyr,
                                                        const unsigned char* cur_img,
                                                        size_t cur_row_stride,
                                                        size_t cur_img_stride,
                                                        const unsigned char* cur_pyr,
                                                        LkScratch lk) {
  const int s = blockIdx.y, pt = blockIdx.x;
  if (pt >= lk.npts[s]) return;
  const int lane = threadIdx.x;
  const int win = P.klt_win, w2 = win * win, wp = win + 1, ws = win + 3;
  extern __shared__ unsigned char lds_raw[];
  short* Iwin = reinterpret_cast<short*>(lds_raw);
  short* dIx = Iwin + w2;
  short* dIy = dIx + w2;
  short* dpx = dIy + w2;          // (win+1)^2 derivative patch
  short* dpy = dpx + wp * wp;
  int* prod1 = reinterpret_cast<int*>(dpy + wp * wp + ((wp * wp + 3 * w2) & 1));
  int* prod2 = prod1 + w2;
  unsigned char* patch = reinterpret_cast<unsigned char*>(prod2 + w2);  // (win+3)^2
  __shared__ float chain[16];

  const unsigned char* pimg = prev_img + (size_t)s * prev_img_stride;
  const unsigned char* cimg = cur_img + (size_t)s * cur_img_stride;
  const unsigned char* ppyr = prev_pyr + (size_t)s * P.pyr_stride;
  const unsigned char* cpyr = cur_pyr + (size_t)s * P.pyr_stride;

  const size_t po = (size_t)s * P.kcap + pt;
  // Tracker.cpp:167-180 drops a point whose landmark is older than maxFeatureAge WHATEVER its tracking result: the
  // reference tracks it and throws the result away.  The step passes the ages (lk.skip_age), and such a point is
  // reported lost without being tracked -- nothing reads its position or error.  (All features of a synthetic stream are
  // born in the bootstrap frame: every 26th step the whole list is in this state.)
  if (lk.skip_age && lk.skip_age[(size_t)s * P.kcap + lk.src_idx[po]] > P.max_age) {
    if (threadIdx.x == 0) lk.status[po] = 0;
    return;
  }
  const float2 prevPt0 = lk.prev_pts[po];
  float2 nextOut = lk.next_pts[po];  // initial flow
  int status = 1;
  float errOut = 0.f;
  const int maxLevel = P.nlevels - 1;
  const float FLT_SCALE = 1.f / (1 << 20);
  const float halfWin = (win - 1) * 0.5f;
  const int W_BITS1 = 14;

  for (int level = maxLevel; level >= 0; level--) {
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
    const int ipx = (int)floorf(prevPt.x), ipy = (int)floorf(prevPt.y);
    if (ipx < -win || ipx >= LI.w || ipy < -win || ipy >= LI.h) {
      if (level == 0) {
        status = 0;
        errOut = 0.f;
      }
      continue;
    }
    float a = prevPt.x - ipx, b = prevPt.y - ipy;
    int iw00, iw01, iw10, iw11;
    lk_weights(a, b, &iw00, &iw01, &iw10, &iw11);

    // stage the (win+3)^2 neighbourhood of the previous level (REFLECT_101 padded image)
    for (int e = lane; e < ws * ws; e += 64) {
      const int py = e / ws, px = e - py * ws;
      patch[e] = (unsigned char)at101(LI, ipx - 1 + px, ipy - 1 + py);
    }
    __syncthreads();
    // Scharr derivative at the (win+1)^2 positions; zero outside the image (BORDER_CONSTANT)
    for (int e = lane; e < wp * wp; e += 64) {
      const int y = e / wp, x = e - y * wp;
      const int gx = ipx + x, gy = ipy + y;
      short vx = 0, vy = 0;
      if (gx >= 0 && gx < LI.w && gy >= 0 && gy < LI.h) {
        const unsigned char* r0 = patch + y * ws + x;  // row gy-1, col gx-1
        const unsigned char* r1 = r0 + ws;
        const unsigned char* r2 = r1 + ws;
        const int t0m = (r0[0] + r2[0]) * 3 + r1[0] * 10, t0p = (r0[2] + r2[2]) * 3 + r1[2] * 10;
        const int t1m = r2[0] - r0[0], t1c = r2[1] - r0[1], t1p = r2[2] - r0[2];
        vx = (short)(t0p - t0m);
        vy = (short)((t1p + t1m) * 3 + t1c * 10);
      }
      dpx[e] = vx;
      dpy[e] = vy;
    }
    __syncthreads();
    // bilinear template and derivative window
    for (int e = lane; e < w2; e += 64) {
      const int y = e / win, x = e - y * win;
      const unsigned char* s0 = patch + (y + 1) * ws + (x + 1);
      const int ival = (s0[0] * iw00 + s0[1] * iw01 + s0[ws] * iw10 + s0[ws + 1] * iw11 +
                        (1 << (W_BITS1 - 5 - 1))) >> (W_BITS1 - 5);
      const int d = y * wp + x;
      const int ixval = (dpx[d] * iw00 + dpx[d + 1] * iw01 + dpx[d + wp] * iw10 +
                         dpx[d + wp + 1] * iw11 + (1 << (W_BITS1 - 1))) >> W_BITS1;
      const int iyval = (dpy[d] * iw00 + dpy[d + 1] * iw01 + dpy[d + wp] * iw10 +
                         dpy[d + wp + 1] * iw11 + (1 << (W_BITS1 - 1))) >> W_BITS1;
      Iwin[e] = (short)max(-32768, min(32767, ival));
      dIx[e] = (short)max(-32768, min(32767, ixval));
      dIy[e] = (short)max(-32768, min(32767, iyval));
    }
    __syncthreads();
    // A11/A12/A22: lanes 0-11 own the SSE lanes (sum q, lane l = pixel x%4), lanes 12-14 the tails
    {
      const int nq = win / 4, tail0 = nq * 4;
      float acc = 0.f;
      if (lane < 12) {
        const int q = lane >> 2, l = lane & 3;
        for (int y = 0; y < win; y++) {
          const short* rx = dIx + y * win;
          const short* ry = dIy + y * win;
          for (int xc = 0; xc < nq; xc++) {
            const float fx = (float)rx[4 * xc + l], fy = (float)ry[4 * xc + l];
            const float t = q == 0 ? fx * fx : (q == 1 ? fx * fy : fy * fy);
            acc = acc + t;
          }
        }
      } else if (lane < 15) {
        const int q = lane - 12;
        for (int y = 0; y < win; y++)
          for (int x = tail0; x < win; x++) {
            const int ix = dIx[y * win + x], iy = dIy[y * win + x];
            const int t = q == 0 ? ix * ix : (q == 1 ? ix * iy : iy * iy);
            acc += (float)t;
          }
      }
      if (lane < 16) chain[lane] = acc;
    }
    __syncthreads();
    float iA11 = chain[12], iA12 = chain[13], iA22 = chain[14];
    iA11 += chain[0] + chain[1] + chain[2] + chain[3];
    iA12 += chain[4] + chain[5] + chain[6] + chain[7];
    iA22 += chain[8] + chain[9] + chain[10] + chain[11];
    __syncthreads();
    const float A11 = iA11 * FLT_SCALE, A12 = iA12 * FLT_SCALE, A22 = iA22 * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    const float minEig =
        (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
    if (minEig < 1e-4f || D < 1.1920929e-07f) {
      if (level == 0) status = 0;
      continue;
    }
    D = 1.f / D;

    nextPt.x -= halfWin;
    nextPt.y -= halfWin;
    float2 prevDelta = make_float2(0.f, 0.f);
    for (int j = 0; j < P.klt_iters; j++) {
      const int inx = (int)floorf(nextPt.x), iny = (int)floorf(nextPt.y);
      if (inx < -win || inx >= LJ.w || iny < -win || iny >= LJ.h) {
        if (level == 0) status = 0;
        break;
      }
      a = nextPt.x - inx;
      b = nextPt.y - iny;
      lk_weights(a, b, &iw00, &iw01, &iw10, &iw11);
      float ib1, ib2;
      for (int e = lane; e < w2; e += 64) {
        const int y = e / win, x = e - y * win;
        const int gx = inx + x, gy = iny + y;
        const int y0 = reflect101(gy, LJ.h), y1 = reflect101(gy + 1, LJ.h);
        const int x0 = reflect101(gx, LJ.w), x1 = reflect101(gx + 1, LJ.w);
        const unsigned char* r0 = LJ.p + (size_t)y0 * LJ.stride;
        const unsigned char* r1 = LJ.p + (size_t)y1 * LJ.stride;
        const int t = (r0[x0] * iw00 + r0[x1] * iw01 + r1[x0] * iw10 + r1[x1] * iw11 +
                       (1 << (W_BITS1 - 5 - 1))) >> (W_BITS1 - 5);
        const int diff = t - (int)Iwin[e];
        prod1[e] = diff * (int)dIx[e];
        prod2[e] = diff * (int)dIy[e];
      }
      __syncthreads();
      {
        const int n8 = win / 8, tail0 = n8 * 8;
        float acc = 0.f;
        if (lane < 8) {
          const int g = lane >> 1;
          const int* pr = (lane & 1) ? prod2 : prod1;
          for (int y = 0; y < win; y++) {
            const int* r = pr + y * win;
            for (int c = 0; c < n8; c++) acc = acc + (float)(r[8 * c + g] + r[8 * c + g + 4]);
          }
        } else if (lane < 10) {
          const int* pr = (lane & 1) ? prod2 : prod1;
          for (int y = 0; y < win; y++)
            for (int x = tail0; x < win; x++) acc += (float)pr[y * win + x];
        }
        if (lane < 10) chain[lane] = acc;
      }
      __syncthreads();
      // bbuf = qb0 + qb1 ; ib1 += bbuf[0] + bbuf[2] ; ib2 += bbuf[1] + bbuf[3]
      const float bb0 = chain[0] + chain[4], bb1 = chain[1] + chain[5], bb2 = chain[2] + chain[6],
                  bb3 = chain[3] + chain[7];
      ib1 = chain[8];
      ib2 = chain[9];
      ib1 += bb0 + bb2;
      ib2 += bb1 + bb3;
      __syncthreads();
      const float b1 = ib1 * FLT_SCALE, b2 = ib2 * FLT_SCALE;
      const float2 delta =
          make_float2((float)((A12 * b2 - A22 * b1) * D), (float)((A12 * b1 - A11 * b2) * D));
      nextPt.x += delta.x;
      nextPt.y += delta.y;
      nextOut = make_float2(nextPt.x + halfWin, nextPt.y + halfWin);
      if ((double)delta.x * (double)delta.x + (double)delta.y * (double)delta.y <= P.klt_eps2)
        break;
      if (j > 0 && fabs((double)(delta.x + prevDelta.x)) < 0.01 &&
          fabs((double)(delta.y + prevDelta.y)) < 0.01) {
        nextOut.x -= delta.x * 0.5f;
        nextOut.y -= delta.y * 0.5f;
        break;
      }
      prevDelta = delta;
    }

    if (status && level == 0) {
      const float2 np = make_float2(nextOut.x - halfWin, nextOut.y - halfWin);
      const int inx = (int)floorf(np.x), iny = (int)floorf(np.y);
      if (inx < -win || inx >= LJ.w || iny < -win || iny >= LJ.h) {
        status = 0;
        continue;
      }
      const float aa = np.x - inx, bb = np.y - iny;
      lk_weights(aa, bb, &iw00, &iw01, &iw10, &iw11);
      // errval is a float sum of integers < 2^24: exact in any order
      int esum = 0;
      for (int e = lane; e < w2; e += 64) {
        const int y = e / win, x = e - y * win;
        const int gx = inx + x, gy = iny + y;
        const int y0 = reflect101(gy, LJ.h), y1 = reflect101(gy + 1, LJ.h);
        const int x0 = reflect101(gx, LJ.w), x1 = reflect101(gx + 1, LJ.w);
        const unsigned char* r0 = LJ.p + (size_t)y0 * LJ.stride;
        const unsigned char* r1 = LJ.p + (size_t)y1 * LJ.stride;
        const int t = (r0[x0] * iw00 + r0[x1] * iw01 + r1[x0] * iw10 + r1[x1] * iw11 +
                       (1 << (W_BITS1 - 5 - 1))) >> (W_BITS1 - 5);
        esum += abs(t - (int)Iwin[e]);
      }
      for (int off = 32; off > 0; off >>= 1) esum += __shfl_xor(esum, off);
      errOut = (float)esum * 1.f / (float)(32 * win * win);
    }
  }
  if (lane == 0) {
    lk.next_pts[po] = nextOut;
    lk.status[po] = (unsigned char)status;
    lk.err[po] = errOut;
  }
}

static size_t lk_generic_lds_bytes(int win) {
  const int w2 = win * win, wp = win + 1, ws = win + 3;
  size_t shorts = (size_t)3 * w2 + 2 * wp * wp;
  shorts += shorts & 1;
  return shorts * 2 + sizeof(int) * 2 * w2 + (((size_t)ws * ws + 3) & ~(size_t)3) + 16;
}

// ---------------------------------------------------------------------------------------------
// Systolic kernel, WIN in {16, 24, 32} (WIN % 8 == 0): one wavefront per point, no LDS in the
// accumulation chains.
//
// Lane map: g = lane >> 4 (one DPP row of 16 lanes per SSE lane class x & 3), q = lane & 15;
// lanes with q < WIN/2 own window rows {2q, 2q+1} and columns {g + 4m, m < WIN/4}: their template,
// gradient and product values never leave registers.  OpenCV's four-lane SSE accumulators are
// strictly sequential float chains over (row, chunk); a chain is therefore a systolic pipeline
// along the DPP row: in stage t every lane adds its own terms to the carry it received (lane t
// holds the true partial sum at stage t) and hands the result to lane q+1 with row_shr:1.  The
// two b components (and A11/A12) ride in one v_pk_add_f32.  The bilinear samples are
// v_dot2_i32_i16 on (pixel, pixel+1) pairs staged once per level in LDS.
// ---------------------------------------------------------------------------------------------

// -DKVFE_LK_PROF (tools/gpu_lk_prof.sh; never in the product build): cycle-counter deltas per phase of
// lk_kernel_sys, summed over all waves of a launch and printed at exit
#ifdef KVFE_LK_PROF
constexpr int LKP_WAVES = 1 << 17, LKP_SLOTS = 12;
__device__ unsigned long long kvfe_lk_prof[(size_t)LKP_WAVES * LKP_SLOTS];   // one record per wave: no contention
#define LKP_DECL unsigned long long lkp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lkp_last = __builtin_readcyclecounter(); \
  const unsigned long long lkp_t0 = __builtin_amdgcn_s_memrealtime(); unsigned lkp_iters = 0, lkp_stages = 0
#if KVFE_LK_PROF == 2   // start / end of every wave only (the phase stamps cost a few per cent)
#define LKP(i) do { } while (0)
#else
#define LKP(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); lkp_acc[i] += t_ - lkp_last; lkp_last = t_; } while (0)
#endif
#define LKP_COUNT(v) (v)++
#else
#define LKP_DECL do { } while (0)
#define LKP(i) do { } while (0)
#define LKP_COUNT(v) do { } while (0)
#endif


template <int WIN>
struct LkSys {
  static constexpr int NC = WIN / 4;        // columns per lane
  static constexpr int NQ = WIN / 2;        // active lanes per DPP row
  static constexpr int NPX = 2 * NC;        // pixels per lane
  static constexpr int NST = NC;            // b-chain steps per lane (2 rows x NC/2 chunks)
  static constexpr int WS = WIN + 3;        // previous-level byte patch side
  static constexpr int WP = WIN + 1;        // derivative patch side
  static constexpr int JM = 3;              // margin of the staged current-level window
  static constexpr int JS = WIN + 1 + 2 * JM;   // staged rows; JS-1 pair columns
  static constexpr int JSTR = ((JS - 1 - 1 + 15) / 16) * 16 + 1;  // == 1 (mod 16): conflict free
  static constexpr int PSTR = (WS + 3) & ~3;   // byte patch row stride: rows start on a dword
  static constexpr int NT = (WP + 3) / 4;       // derivative tasks per row (4 outputs each)
  static constexpr int DSTR = (4 * NT) | 1;     // derivative patch row stride (odd: spread over the banks)
  static constexpr int NJ = (JS - 1 + 3) / 4;   // pair-staging tasks per row (4 pairs each)
  static constexpr int PATCH_B = WS * PSTR;
  static constexpr int DXY_W = WP * DSTR;
  // the derivative patch is dead once the template registers are filled: the pair window of the iterations aliases it
  static constexpr int LDS_BYTES = PATCH_B + 4 * (DXY_W > JS * JSTR ? DXY_W : JS * JSTR);
  static_assert(WP == 4 * (NT - 1) + 1, "last derivative task holds exactly one output");
  static_assert(JS - 1 == 4 * (NJ - 1) + 2 && 4 * NJ <= JSTR, "last pair task holds exactly two pairs");
};


// seven waves per SIMD up to the reference's window of 24 (72 VGPRs, 8 spilled to scratch, 4.8 KB of LDS per point; round 4,
// tools/r4/gpu_ab.sh: 0.534 -> 0.530 ms against six waves at 80 VGPRs without spills -- the launch hardly notices the
// seventh wave: it is bound by instruction issue of all categories plus its tail); a window of 32 keeps 16 pixels per
// lane in registers and stays at five
template <int WIN>
__global__ __launch_bounds__(64, (WIN <= 24 ? 7 : 5)) void lk_kernel_sys(KParams P, const unsigned char* prev_img,
                                                    size_t prev_row_stride,
                                                    size_t prev_img_stride,
                                                    const unsigned char* prev_pyr,
                                                    const unsigned char* cur_img,
                                                    size_t cur_row_stride, size_t cur_img_stride,
                                                    const unsigned char* cur_pyr, LkScratch lk, int use_order) {
  using C = LkSys<WIN>;
  constexpr int NC = C::NC, NQ = C::NQ, NPX = C::NPX, NST = C::NST, WS = C::WS, WP = C::WP,
                JM = C::JM, JS = C::JS, JSTR = C::JSTR, PSTR = C::PSTR, NT = C::NT, DSTR = C::DSTR,
                NJ = C::NJ;
  // blockIdx = (point, stream); flags bit 3: the caller does not read the error output (the front-end step).
  // (Round 3 measured two more launch forms -- points ordered by the previous frame's iteration count, and the launch
  // split over two HIP streams; both lost their A/B and were removed in round 4)
  const bool want_err = !(use_order & 8);
  const int s = blockIdx.y, rank = blockIdx.x;
  if (rank >= lk.npts[s]) return;
  const int pt = rank;
  int iters_total = 0;
  const int lane = threadIdx.x;
  const int g = lane >> 4, q = lane & 15;
  const bool active = q < NQ;
  const int qa = active ? q : 0;  // idle lanes shadow lane q = 0 (their results are never read)

  __shared__ __attribute__((aligned(16))) unsigned char lds[C::LDS_BYTES];
  unsigned char* patch = lds;                                            // WS x WS bytes
  int* dxy = reinterpret_cast<int*>(lds + C::PATCH_B);                   // WP x WP (dx | dy << 16)
  int* jp = dxy;                                                         // JS x JSTR pixel pairs (over dxy)

  const unsigned char* pimg = prev_img + (size_t)s * prev_img_stride;
  const unsigned char* cimg = cur_img + (size_t)s * cur_img_stride;
  const unsigned char* ppyr = prev_pyr + (size_t)s * P.pyr_stride;
  const unsigned char* cpyr = cur_pyr + (size_t)s * P.pyr_stride;

  const size_t po = (size_t)s * P.kcap + pt;
  // Tracker.cpp:167-180 drops a point whose landmark is older than maxFeatureAge WHATEVER its tracking result: the
  // reference tracks it and throws the result away.  The step passes the ages (lk.skip_age), and such a point is
  // reported lost without being tracked -- nothing reads its position or error.  (All features of a synthetic stream are
  // born in the bootstrap frame: every 26th step the whole list is in this state.)
  if (lk.skip_age && lk.skip_age[(size_t)s * P.kcap + lk.src_idx[po]] > P.max_age) {
    if (threadIdx.x == 0) lk.status[po] = 0;
    return;
  }
  const float2 prevPt0 = lk.prev_pts[po];
  float2 nextOut = lk.next_pts[po];  // initial flow
  int status = 1;
  float errOut = 0.f;
  const int maxLevel = P.nlevels - 1;
  const float FLT_SCALE = 1.f / (1 << 20);
  const float halfWin = (WIN - 1) * 0.5f;
  const int klt_iters = P.klt_iters;
  const double klt_eps2 = P.klt_eps2;
  // window offset of this lane's first pixel (row 2q, column g)
  const int y0w = 2 * qa, x0w = g;
  LKP_DECL;

  for (int level = maxLevel; level >= 0; level--) {
    LKP(7);
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
    const int ipx = (int)floorf(prevPt.x), ipy = (int)floorf(prevPt.y);
    if (ipx < -WIN || ipx >= LI.w || ipy < -WIN || ipy >= LI.h) {
      if (level == 0) {
        status = 0;
        errOut = 0.f;
      }
      continue;
    }
    float a = prevPt.x - ipx, b = prevPt.y - ipy;
    int iw00, iw01, iw10, iw11;
    lk_weights(a, b, &iw00, &iw01, &iw10, &iw11);
    int wq0 = pack_lo16(iw00, iw01), wq1 = pack_lo16(iw10, iw11);

    __syncthreads();  // previous level's readers of patch / dxy / jp are done
    LKP(0);
    // The (WIN+3)^2 byte neighbourhood of the previous level (REFLECT_101 padded image) and the Scharr derivative
    // at the (WIN+1)^2 window positions (zero outside the image: BORDER_CONSTANT).
    if (ipx - 1 >= 0 && ipy - 1 >= 0 && ipx - 1 + 4 * NT <= LI.w && ipy - 1 + WS <= LI.h) {
      // Interior window: one task = four horizontally adjacent derivative positions (x = 4t .. 4t+3 of row y),
      // read as three rows of 6 bytes straight from the image (dword loads at byte addresses), the vertical
      // passes of the separable filter in packed 16-bit arithmetic on column pairs; the middle row doubles as the
      // byte patch of the template gather.  Every sum stays below 2^12 in magnitude: exact.
      const unsigned char* base = LI.p + (size_t)(ipy - 1) * LI.stride + (ipx - 1);
#pragma unroll 1
      for (int it = 0; it < (WP * NT + 63) / 64; it++) {
        const int e = lane + 64 * it;
        const int y = e / NT, t = e - y * NT;
        if (y < WP) {
          // (the last task of a row holds one output and needs its first three bytes only: its second dword,
          // which would reach past the window, is a repeat of the first and feeds outputs nobody reads)
          const unsigned char* r = base + (size_t)y * LI.stride + 4 * t;
          const int hoff = t < NT - 1 ? 4 : 0;
          int lo[3], hi[3];
#pragma unroll
          for (int k = 0; k < 3; k++) {
            lo[k] = *reinterpret_cast<const int_u*>(r + (size_t)k * LI.stride);
            hi[k] = *reinterpret_cast<const int_u*>(r + (size_t)k * LI.stride + hoff);
          }
          *reinterpret_cast<int*>(patch + (y + 1) * PSTR + 4 * t) = lo[1];
          v2us c01[3], c23[3], c45[3];
#pragma unroll
          for (int k = 0; k < 3; k++) {
            c01[k] = as_v2us(perm_b32(0, lo[k], 0x0c010c00u));
            c23[k] = as_v2us(perm_b32(0, lo[k], 0x0c030c02u));
            c45[k] = as_v2us(perm_b32(0, hi[k], 0x0c010c00u));
          }
          const v2us k3 = {3, 3}, k10 = {10, 10};
          const v2us s01 = (c01[0] + c01[2]) * k3 + c01[1] * k10, s23 = (c23[0] + c23[2]) * k3 + c23[1] * k10,
                     s45 = (c45[0] + c45[2]) * k3 + c45[1] * k10;
          const v2us d01 = c01[2] - c01[0], d23 = c23[2] - c23[0], d45 = c45[2] - c45[0];
          const v2us vx12 = s23 - s01, vx34 = s45 - s23;
          const v2us m12 = as_v2us(perm_b32(as_i32(d23), as_i32(d01), 0x05040302u));
          const v2us m34 = as_v2us(perm_b32(as_i32(d45), as_i32(d23), 0x05040302u));
          const v2us vy12 = (d01 + d23) * k3 + m12 * k10, vy34 = (d23 + d45) * k3 + m34 * k10;
          int* o = dxy + y * DSTR + 4 * t;
          o[0] = pack_lo16(as_i32(vx12), as_i32(vy12));
          o[1] = pack_hi16(as_i32(vx12), as_i32(vy12));
          o[2] = pack_lo16(as_i32(vx34), as_i32(vy34));
          o[3] = pack_hi16(as_i32(vx34), as_i32(vy34));
        }
      }
      __syncthreads();
    } else {
      for (int e = lane; e < WS * WS; e += 64) {
        const int py = e / WS, px = e - py * WS;
        patch[py * PSTR + px] = (unsigned char)at101(LI, ipx - 1 + px, ipy - 1 + py);
      }
      __syncthreads();
      for (int e = lane; e < WP * WP; e += 64) {
        const int y = e / WP, x = e - y * WP;
        const int gx = ipx + x, gy = ipy + y;
        int vx = 0, vy = 0;
        if (gx >= 0 && gx < LI.w && gy >= 0 && gy < LI.h) {
          const unsigned char* r0 = patch + y * PSTR + x;  // row gy-1, col gx-1
          const unsigned char* r1 = r0 + PSTR;
          const unsigned char* r2 = r1 + PSTR;
          const int t0m = (r0[0] + r2[0]) * 3 + r1[0] * 10, t0p = (r0[2] + r2[2]) * 3 + r1[2] * 10;
          const int t1m = r2[0] - r0[0], t1c = r2[1] - r0[1], t1p = r2[2] - r0[2];
          vx = t0p - t0m;
          vy = (t1p + t1m) * 3 + t1c * 10;
        }
        dxy[y * DSTR + x] = pack_lo16(vx, vy);
      }
      __syncthreads();
    }
    LKP(1);
    // bilinear template and derivative window of this lane's pixels -> registers
    // rA = 2^8 - (I << 9): the template value folded into the start value of the blend's accumulator -- the blend of the
    // current frame, its rounding and the subtraction of I are then two dot products and a shift:
    // ((b + 2^8) >> 9) - I == (b + 2^8 - (I << 9)) >> 9 for every integer b
    int rA[NPX], rgxy[NPX];  // rgxy = (Ix & 0xffff) | (Iy << 16)
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int m = 0; m < NC; m++) {
        const int k = r * NC + m;
        const int y = y0w + r, x = x0w + 4 * m;
        const unsigned char* s0 = patch + (y + 1) * PSTR + (x + 1);
        const int p0 = perm_b32(0, *reinterpret_cast<const ushort_u*>(s0), 0x0c010c00u);
        const int p1 = perm_b32(0, *reinterpret_cast<const ushort_u*>(s0 + PSTR), 0x0c010c00u);
        const int ival = dot2_i16(p0, wq0, dot2_i16(p1, wq1, 1 << 8)) >> 9;
        const int* d = dxy + y * DSTR + x;
        const int d00 = d[0], d01 = d[1], d10 = d[DSTR], d11 = d[DSTR + 1];
        const int ixval = dot2_i16(pack_lo16(d00, d01), wq0, dot2_i16(pack_lo16(d10, d11), wq1, 1 << 13)) >> 14;
        const int iyval = dot2_i16(pack_hi16(d00, d01), wq0, dot2_i16(pack_hi16(d10, d11), wq1, 1 << 13)) >> 14;
        // no saturation (OpenCV stores these as short without one): 0 <= ival <= 255 * 2^14 >> 9 = 8160 and
        // |ixval|, |iyval| <= 4080 (a bilinear blend of Scharr sums of at most 16 * 255).  No zeroing of the idle lanes
        // either (q >= NQ shadow lane 0's pixels): they sit BEHIND lane NQ-1 in the row_shr pipeline of their DPP row,
        // so nothing they compute reaches a lane that is read -- and a select here is re-issued in every iteration.
        rA[k] = (1 << 8) - (ival << 9);
        rgxy[k] = pack_lo16(ixval, iyval);
      }
    LKP(2);
    // A11/A12/A22 chains (SSE lane l = g; order: row, then column chunk)
    float A11, A12, A22;
    {
      v2f pa[NPX];
      float pc[NPX];
#pragma unroll
      for (int k = 0; k < NPX; k++) {
        const float fx = (float)(short)(rgxy[k] & 0xffff), fy = (float)(rgxy[k] >> 16);
        pa[k] = v2f{fx * fx, fx * fy};
        pc[k] = fy * fy;
      }
      // stage st: carry from lane q-1 (0 into lane 0 of the row) + this lane's terms, in order.  The hand-over is fused
      // into the stage's first addition (v_add_f32_dpp: the shifted carry IS the left operand), so a stage is
      // 1 + (NPX - 1) additions per component instead of a move and NPX additions.
      v2f t = {0.f, 0.f};
      float t22 = 0.f;
#pragma unroll
      for (int st = 0; st < NQ; st++) {
        {
          float cx = t.x, cy = t.y;
          dpp_shr1_add3(cx, cy, t22, pa[0].x, pa[0].y, pc[0]);
          t = v2f{cx, cy};
        }
#pragma unroll
        for (int k = 1; k < NPX; k++) {
          t = t + pa[k];
          t22 = t22 + pc[k];
        }
      }
      constexpr int L0 = NQ - 1, L1 = 16 + NQ - 1, L2 = 32 + NQ - 1, L3 = 48 + NQ - 1;
      float iA11 = 0.f, iA12 = 0.f, iA22 = 0.f;
      iA11 += lane_f(t.x, L0) + lane_f(t.x, L1) + lane_f(t.x, L2) + lane_f(t.x, L3);
      iA12 += lane_f(t.y, L0) + lane_f(t.y, L1) + lane_f(t.y, L2) + lane_f(t.y, L3);
      iA22 += lane_f(t22, L0) + lane_f(t22, L1) + lane_f(t22, L2) + lane_f(t22, L3);
      A11 = iA11 * FLT_SCALE;
      A12 = iA12 * FLT_SCALE;
      A22 = iA22 * FLT_SCALE;
    }
    float D = A11 * A22 - A12 * A12;
    const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                         (float)(2 * WIN * WIN);
    if (minEig < 1e-4f || D < 1.1920929e-07f) {
      if (level == 0) status = 0;
      continue;
    }
    D = 1.f / D;
    LKP(3);

    nextPt.x -= halfWin;
    nextPt.y -= halfWin;
    float2 prevDelta = make_float2(0.f, 0.f);
    int jx0 = 0, jy0 = 0;
    bool jvalid = false;
    // (re)stage the current-level window as (pixel, pixel+1) pairs around (inx, iny)
    auto stage_j = [&](int inx, int iny) {
      jx0 = inx - JM;
      jy0 = iny - JM;
      __syncthreads();
      LKP(5);
      LKP_COUNT(lkp_stages);
      const bool interior = jx0 >= 0 && jy0 >= 0 && jx0 + 4 * NJ <= LJ.w && jy0 + JS <= LJ.h;
      if (interior) {
        // one task = four pairs of a row out of five bytes (two dword loads at byte addresses)
        const unsigned char* base = LJ.p + (size_t)jy0 * LJ.stride + jx0;
#pragma unroll 1
        for (int it = 0; it < (JS * NJ + 63) / 64; it++) {
          const int e = lane + 64 * it;
          const int yy = e / NJ, i = e - yy * NJ;
          if (yy < JS) {
            const unsigned char* r = base + (size_t)yy * LJ.stride + 4 * i;
            const int lo = *reinterpret_cast<const int_u*>(r);
            const int hi = *reinterpret_cast<const int_u*>(r + (i < NJ - 1 ? 4 : 0));  // last task: two pairs, lo only
            int* o = jp + yy * JSTR + 4 * i;
            o[0] = perm_b32(hi, lo, 0x0c010c00u);
            o[1] = perm_b32(hi, lo, 0x0c020c01u);
            o[2] = perm_b32(hi, lo, 0x0c030c02u);
            o[3] = perm_b32(hi, lo, 0x0c040c03u);
          }
        }
      } else {
        for (int e = lane; e < JS * (JS - 1); e += 64) {
          const int yy = e / (JS - 1), xx = e - yy * (JS - 1);
          jp[yy * JSTR + xx] =
              at101(LJ, jx0 + xx, jy0 + yy) | (at101(LJ, jx0 + xx + 1, jy0 + yy) << 16);
        }
      }
      __syncthreads();
      jvalid = true;
      LKP(4);
    };
    for (int j = 0; j < klt_iters; j++) {
      LKP(5);
      LKP_COUNT(lkp_iters);
      iters_total++;
      const int inx = (int)floorf(nextPt.x), iny = (int)floorf(nextPt.y);
      if (inx < -WIN || inx >= LJ.w || iny < -WIN || iny >= LJ.h) {
        if (level == 0) status = 0;
        break;
      }
      a = nextPt.x - inx;
      b = nextPt.y - iny;
      lk_weights(a, b, &iw00, &iw01, &iw10, &iw11);
      wq0 = pack_lo16(iw00, iw01);
      wq1 = pack_lo16(iw10, iw11);
      if (!jvalid || inx < jx0 || iny < jy0 || inx + WIN + 1 > jx0 + JS || iny + WIN + 1 > jy0 + JS)
        stage_j(inx, iny);
      const int* jb = jp + (iny - jy0 + y0w) * JSTR + (inx - jx0 + x0w);
      int diff[NPX];
#pragma unroll
      for (int r = 0; r < 2; r++)
#pragma unroll
        for (int m = 0; m < NC; m++) {
          const int k = r * NC + m;
          diff[k] = dot2x2_i16(jb[(r + 1) * JSTR + 4 * m], wq1, rA[k], jb[r * JSTR + 4 * m], wq0) >> 9;
        }
      // chain terms: madd pairs (x, x+4) of one row chunk, converted to float
      v2f term[NST];
#pragma unroll
      for (int r = 0; r < 2; r++)
#pragma unroll
        for (int c = 0; c < NC / 2; c++) {
          const int k0 = r * NC + 2 * c, k1 = k0 + 1;
          const int dd = pack_lo16(diff[k0], diff[k1]);
          int m1, m2;
          dot2_pair0_i16(dd, pack_lo16(rgxy[k0], rgxy[k1]), pack_hi16(rgxy[k0], rgxy[k1]), m1, m2);
          term[r * (NC / 2) + c] = v2f{(float)m1, (float)m2};
        }
      v2f t = {0.f, 0.f};
#pragma unroll
      for (int st = 0; st < NQ; st++) {
        {
          float cx = t.x, cy = t.y;
          dpp_shr1_add2(cx, cy, term[0].x, term[0].y);
          t = v2f{cx, cy};
        }
#pragma unroll
        for (int i = 1; i < NST; i++) t = t + term[i];
      }
      constexpr int L0 = NQ - 1, L1 = 16 + NQ - 1, L2 = 32 + NQ - 1, L3 = 48 + NQ - 1;
      // bbuf = qb0 + qb1 ; ib1 += bbuf[0] + bbuf[2] ; ib2 += bbuf[1] + bbuf[3]
      const float bb0 = lane_f(t.x, L0) + lane_f(t.x, L2), bb1 = lane_f(t.y, L0) + lane_f(t.y, L2);
      const float bb2 = lane_f(t.x, L1) + lane_f(t.x, L3), bb3 = lane_f(t.y, L1) + lane_f(t.y, L3);
      float ib1 = 0.f, ib2 = 0.f;
      ib1 += bb0 + bb2;
      ib2 += bb1 + bb3;
      const float b1 = ib1 * FLT_SCALE, b2 = ib2 * FLT_SCALE;
      const float2 delta =
          make_float2((float)((A12 * b2 - A22 * b1) * D), (float)((A12 * b1 - A11 * b2) * D));
      nextPt.x += delta.x;
      nextPt.y += delta.y;
      nextOut = make_float2(nextPt.x + halfWin, nextPt.y + halfWin);
      if ((double)delta.x * (double)delta.x + (double)delta.y * (double)delta.y <= klt_eps2) break;
      if (j > 0 && fabs((double)(delta.x + prevDelta.x)) < 0.01 &&
          fabs((double)(delta.y + prevDelta.y)) < 0.01) {
        nextOut.x -= delta.x * 0.5f;
        nextOut.y -= delta.y * 0.5f;
        break;
      }
      prevDelta = delta;
    }
    LKP(5);

    if (status && level == 0) {
      const float2 np = make_float2(nextOut.x - halfWin, nextOut.y - halfWin);
      const int inx = (int)floorf(np.x), iny = (int)floorf(np.y);
      if (inx < -WIN || inx >= LJ.w || iny < -WIN || iny >= LJ.h) {
        status = 0;
        continue;
      }
      // (the window test above is part of the tracking result -- calcOpticalFlowPyrLK clears the status there whenever
      // an error array is passed, and the reference passes one, Tracker.cpp:137-139 -- the error itself is an output
      // nobody reads in the front-end step: Tracker::featureTracking drops the vector)
      if (!want_err) continue;
      const float aa = np.x - inx, bb = np.y - iny;
      lk_weights(aa, bb, &iw00, &iw01, &iw10, &iw11);
      wq0 = pack_lo16(iw00, iw01);
      wq1 = pack_lo16(iw10, iw11);
      if (!jvalid || inx < jx0 || iny < jy0 || inx + WIN + 1 > jx0 + JS || iny + WIN + 1 > jy0 + JS)
        stage_j(inx, iny);
      const int* jb = jp + (iny - jy0 + y0w) * JSTR + (inx - jx0 + x0w);
      // errval is a float sum of integers < 2^24: exact in any order
      int esum = 0;
#pragma unroll
      for (int r = 0; r < 2; r++)
#pragma unroll
        for (int m = 0; m < NC; m++) {
          const int k = r * NC + m;
          const int t = dot2x2_i16(jb[(r + 1) * JSTR + 4 * m], wq1, rA[k], jb[r * JSTR + 4 * m], wq0) >> 9;
          esum += active ? abs(t) : 0;
        }
      for (int off = 32; off > 0; off >>= 1) esum += __shfl_xor(esum, off);
      errOut = (float)esum * 1.f / (float)(32 * WIN * WIN);
      LKP(6);
    }
  }
  if (lane == 0) {
    lk.next_pts[po] = nextOut;
    lk.status[po] = (unsigned char)status;
    lk.err[po] = errOut;
#ifdef KVFE_LK_PROF
    const size_t wid = (size_t)blockIdx.y * gridDim.x + blockIdx.x;   // (dispatch order)
    if (wid < (size_t)LKP_WAVES) {
      unsigned long long* o = kvfe_lk_prof + wid * LKP_SLOTS;
      for (int i = 0; i < 8; i++) o[i] = lkp_acc[i];
      o[8] = 1ull;
      o[9] = lkp_iters;
      o[10] = lkp_stages;
      o[11] = lkp_t0;                                 // 100 MHz device-wide clock
      o[7] += 0;
      o[8] = 1ull | (__builtin_amdgcn_s_memrealtime() - lkp_t0) << 8;   // wave lifetime in 10 ns ticks
    }
#endif
  }
}

void launch_lk(const KParams& P, const unsigned char* prev_img, size_t prev_row_stride,
               size_t prev_img_stride, const unsigned char* prev_pyr, const unsigned char* cur_img,
               size_t cur_row_stride, size_t cur_img_stride, const unsigned char* cur_pyr,
               const LkScratch& lk, int max_pts, hipStream_t st, bool want_err) {
  if (max_pts <= 0) return;
  // the front-end step (no error output): four points per wavefront (k_lk4.hip) unless kvfe_config.lk_impl = 1 /
  // KVFE_LK_IMPL=1 (the A/B switch of tools/ and bench.py) asks for the one-point kernel
  static const int env_impl = [] { const char* e = std::getenv("KVFE_LK_IMPL"); return e ? std::atoi(e) : -1; }();
  const bool one = env_impl >= 0 ? env_impl == 1 : P.lk_one != 0;
  if (!want_err && !one &&
      launch_lk4(P, prev_img, prev_row_stride, prev_img_stride, prev_pyr, cur_img, cur_row_stride, cur_img_stride,
                 cur_pyr, lk, max_pts, st))
    return;
  const int flags = want_err ? 0 : 8;   // (bit 3: no error output -- Tracker::featureTracking drops that vector)
  const dim3 grid(max_pts, P.B), block(64);
#define KVFE_LK_SYS(WINSZ)                                                                                          \
  hipLaunchKernelGGL(lk_kernel_sys<WINSZ>, grid, block, 0, st, P, prev_img, prev_row_stride, prev_img_stride, prev_pyr, \
                     cur_img, cur_row_stride, cur_img_stride, cur_pyr, lk, flags)
  switch (P.klt_win) {
    case 16: KVFE_LK_SYS(16); break;
    case 24: KVFE_LK_SYS(24); break;
    case 32: KVFE_LK_SYS(32); break;
    default:
      hipLaunchKernelGGL(lk_kernel_generic, grid, block, lk_generic_lds_bytes(P.klt_win), st, P,
                         prev_img, prev_row_stride, prev_img_stride, prev_pyr, cur_img,
                         cur_row_stride, cur_img_stride, cur_pyr, lk);
  }
#undef KVFE_LK_SYS
#ifdef KVFE_LK_PROF
  {
    static double h[LKP_SLOTS];
    static std::vector<unsigned long long> buf((size_t)LKP_WAVES * LKP_SLOTS);
    static double wmax = 0;
    static bool reg = false;
    void* dev = nullptr;
    hipGetSymbolAddress(&dev, HIP_SYMBOL(kvfe_lk_prof));
    hipStreamSynchronize(st);
    hipMemcpy(buf.data(), dev, buf.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    hipMemset(dev, 0, buf.size() * sizeof(unsigned long long));
    static double tl[8], ithist[8];   // timeline of the launch: when 50 / 90 / 99 / 100 % of the waves have ended
    static long nl = 0;
    std::vector<double> ends;
    double t_first = 1e300;
    for (size_t w = 0; w < (size_t)LKP_WAVES; w++) {
      const unsigned long long* o = buf.data() + w * LKP_SLOTS;
      if (!o[8]) continue;
      double tot = 0;
      for (int i = 0; i < 8; i++) tot += (double)o[i];
      if (tot > wmax) wmax = tot;
      for (int i = 0; i < 11; i++) h[i] += i == 8 ? 1.0 : (double)o[i];
      if ((double)o[11] < t_first) t_first = (double)o[11];
      ends.push_back((double)o[11] + (double)(o[8] >> 8));
      const int it = (int)o[9];
      ithist[it <= 3 ? 0 : it <= 6 ? 1 : it <= 12 ? 2 : it <= 24 ? 3 : it <= 48 ? 4 : 5] += 1;
    }
    if (!ends.empty() && nl == 5) {   // one launch in detail: resident waves and mean lifetime along the launch
      std::vector<std::pair<double, double>> se;
      for (size_t w = 0; w < (size_t)LKP_WAVES; w++) {
        const unsigned long long* o = buf.data() + w * LKP_SLOTS;
        if (o[8]) se.push_back({(double)o[11] - t_first, (double)(o[8] >> 8)});
      }
      double t_end = 0, t_last_start = 0, life_sum = 0;
      for (auto& e : se) {
        t_end = std::max(t_end, e.first + e.second);
        t_last_start = std::max(t_last_start, e.first);
        life_sum += e.second;
      }
      {
        std::vector<double> lf;
        for (auto& e : se) lf.push_back(e.second);
        std::sort(lf.begin(), lf.end());
        std::fprintf(stderr, "KVFE_LK_PROF one launch: last wave started at %.0f, lifetimes (10 ns): mean %.0f p50 %.0f p90 %.0f p99 %.0f p99.9 %.0f max %.0f; ten longest waves [start, lifetime]:",
                     t_last_start, life_sum / se.size(), lf[lf.size() / 2], lf[lf.size() * 9 / 10], lf[lf.size() * 99 / 100], lf[lf.size() * 999 / 1000], lf.back());
        std::vector<std::pair<double, double>> by = se;
        std::sort(by.begin(), by.end(), [](const std::pair<double, double>& a, const std::pair<double, double>& b) { return a.second > b.second; });
        for (int k = 0; k < 10 && k < (int)by.size(); k++) std::fprintf(stderr, " [%.0f, %.0f]", by[k].first, by[k].second);
        std::fprintf(stderr, "\n");
      }
      std::fprintf(stderr, "KVFE_LK_PROF one launch, %zu waves, %.0f ticks of 10 ns: [t: resident waves, mean lifetime of the waves started in the slice]",
                   se.size(), t_end);
      for (int k = 0; k < 20; k++) {
        const double t0 = t_end * k / 20, t1 = t_end * (k + 1) / 20, tm = 0.5 * (t0 + t1);
        int res = 0, ns = 0;
        double life = 0;
        for (auto& e : se) {
          if (e.first <= tm && tm < e.first + e.second) res++;
          if (e.first >= t0 && e.first < t1) { ns++; life += e.second; }
        }
        std::fprintf(stderr, " [%.0f: %d, %d started, %.0f]", tm, res, ns, ns ? life / ns : 0.0);
      }
      std::fprintf(stderr, "\n");
    }
    if (!ends.empty()) {
      std::sort(ends.begin(), ends.end());
      const size_t n = ends.size();
      tl[0] += ends[n / 2] - t_first;
      tl[1] += ends[n * 9 / 10] - t_first;
      tl[2] += ends[n * 99 / 100] - t_first;
      tl[3] += ends[n - 1] - t_first;
      tl[4] += ends[n * 999 / 1000] - t_first;
      nl++;
    }
    if (!reg) {
      reg = true;
      std::atexit([] {
        const double n = h[8] ? h[8] : 1.0;
        std::fprintf(stderr,
                     "KVFE_LK_PROF waves %.0f  cycles per wave: level-header %.0f  patch+scharr %.0f  gather %.0f  "
                     "A-chains %.0f  stage_j %.0f  iterations %.0f  err %.0f  between-levels %.0f  (slowest wave %.0f) | "
                     "iterations %.2f stage_j calls %.2f\n",
                     h[8], h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[5] / n, h[6] / n, h[7] / n, wmax,
                     h[9] / n, h[10] / n);
        std::fprintf(stderr,
                     "KVFE_LK_PROF launch timeline (10 ns ticks after the first wave started): 50 %% of the waves ended by "
                     "%.0f, 90 %% by %.0f, 99 %% by %.0f, 99.9 %% by %.0f, all by %.0f | iterations per wave (3 levels): <=3 %.3f  <=6 "
                     "%.3f  <=12 %.3f  <=24 %.3f  <=48 %.3f  more %.3f\n",
                     tl[0] / nl, tl[1] / nl, tl[2] / nl, tl[4] / nl, tl[3] / nl, ithist[0] / n, ithist[1] / n, ithist[2] / n,
                     ithist[3] / n, ithist[4] / n, ithist[5] / n);
      });
    }
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// predictor (one block per stream)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void matx33f_mul(const float* a, const float* b, float* c) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      float s = 0;
      for (int k = 0; k < 3; k++) s += a[i * 3 + k] * b[k * 3 + j];
      c[i * 3 + j] = s;
    }
}

__device__ __forceinline__ double quaternion_w(const double* m) {  // Eigen::Quaterniond(R).w()
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    return 0.5 * t;
  }
  int i = 0;
  if (m[4] > m[0]) i = 1;
  if (m[8] > m[i * 4]) i = 2;
  const int j = (i + 1) % 3, k = (j + 1) % 3;
  t = sqrt(m[i * 4] - m[j * 4] - m[k * 4] + 1.0);
  t = 0.5 / t;
  return (m[k * 3 + j] - m[j * 3 + k]) * t;
}

// H (float 3x3) for the rotational predictor, or identity flag
__device__ bool predictor_homography(const KParams& P, const Tables& T, const double* ref_R_cur,
                                     float* H) {
  if (P.predictor == 0) return false;
  if (fabs(1.0 - fabs(quaternion_w(ref_R_cur))) < 1e-4) return false;
  float Rt[9], KR[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Rt[i * 3 + j] = (float)ref_R_cur[j * 3 + i];
  matx33f_mul(T.Kf, Rt, KR);
  matx33f_mul(KR, T.Kinvf, H);
  return true;
}

__device__ __forceinline__ float2 predict_point(const float* H, float2 p, int W, int Hh) {
  const float p1[3] = {p.x, p.y, 1.0f};
  float p2[3];
  for (int r = 0; r < 3; r++) {
    float s = 0;
    for (int k = 0; k < 3; k++) s += H[r * 3 + k] * p1[k];
    p2[r] = s;
  }
  float2 q = p;
  if (p2[2] > 0.0f) q = make_float2(p2[0] / p2[2], p2[1] / p2[2]);
  if (0.0f <= q.x && q.x < (float)W && 0.0f <= q.y && q.y < (float)Hh) return q;
  return p;
}

__global__ __launch_bounds__(256) void track_prepare_kernel(KParams P, Tables T, FrameTab KM1,
                                                            StreamState S, LkScratch lk) {
  const int s = blockIdx.x;
  __shared__ float Hs[9];
  __shared__ int use_h;
  if (!(S.flags[s] & FLAG_INIT)) {
    if (threadIdx.x == 0) lk.npts[s] = 0;
    return;
  }
  if (threadIdx.x == 0) {
    // ref_frame_R_cur_frame = keyframe_R_ref_frame^-1 * keyframe_R_cur_frame
    // (StereoVisionImuFrontend.cpp:304-305)
    const double* Rref = S.kf_R_ref + (size_t)s * 9;
    const double* Rcur = S.kf_R_cur + (size_t)s * 9;
    double R[9];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        double acc = 0;
        for (int k = 0; k < 3; k++) acc += Rref[k * 3 + i] * Rcur[k * 3 + j];
        R[i * 3 + j] = acc;
      }
    float H[9];
    const bool u = predictor_homography(P, T, R, H);
    use_h = u ? 1 : 0;
    for (int i = 0; i < 9; i++) Hs[i] = H[i];
  }
  __syncthreads();
  // Tracker.cpp:103-112: only keypoints with a valid landmark are tracked (geometric outlier
  // rejection leaves landmark -1 entries in a keyframe); src_idx maps point -> index in frame k-1
  const int n = KM1.count[s];
  const size_t so = (size_t)s * P.kcap;
  __shared__ int wave_tot[4];
  __shared__ int sh_off;
  if (threadIdx.x == 0) sh_off = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 256) {
    const int i = base + threadIdx.x;
    const bool valid = i < n && KM1.lmk[so + i] != -1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = valid ? 1 : 0;
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
    if (valid) {
      const int o = off0 + wbase + inc - 1;
      const float2 p = KM1.kp[so + i];
      lk.prev_pts[so + o] = p;
      lk.next_pts[so + o] = use_h ? predict_point(Hs, p, P.W, P.H) : p;
      lk.src_idx[so + o] = i;
    }
    __syncthreads();
    if (threadIdx.x == 0) sh_off = off0 + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) lk.npts[s] = sh_off;
}

void launch_track_prepare(const KParams& P, const Tables& T, const FrameTab& km1,
                          const StreamState& S, const LkScratch& lk, hipStream_t st) {
  hipLaunchKernelGGL(track_prepare_kernel, dim3(P.B), dim3(256), 0, st, P, T, km1, S, lk);
}

__global__ void predict_flow_kernel(KParams P, Tables T, const double* R, const float2* prev,
                                    int n, float2* out) {
  __shared__ float Hs[9];
  __shared__ int use_h;
  if (threadIdx.x == 0) {
    double Rl[9];
    for (int i = 0; i < 9; i++) Rl[i] = R[i];
    float H[9];
    use_h = predictor_homography(P, T, Rl, H) ? 1 : 0;
    for (int i = 0; i < 9; i++) Hs[i] = H[i];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x)
    out[i] = use_h ? predict_point(Hs, prev[i], P.W, P.H) : prev[i];
}

void launch_predict_flow(const KParams& P, const Tables& T, const double* R, const float2* prev,
                         int n, float2* out, hipStream_t st) {
  hipLaunchKernelGGL(predict_flow_kernel, dim3(1), dim3(256), 0, st, P, T, R, prev, n, out);
}

// ---------------------------------------------------------------------------------------------
// survivors -> frame k; keyframe decision
// ---------------------------------------------------------------------------------------------
// 1024 threads per stream (round 5; 256 until then): a stream's ~600 points are ONE pass of each loop below instead of
// three or four, and every pass is a chain of dependent memory round trips, a block scan and -- in the first loop -- the
// ~400 dependent float64 instructions of a bearing vector.
constexpr int TF_T = 1024;

__device__ __forceinline__ int block_scan_tf(int v, int* wave_tot, int* total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int inc = v;
  for (int off = 1; off < 64; off <<= 1) {
    int t = __shfl_up(inc, off);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wave_tot[wv] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < TF_T / 64; i++) {
    const int t = wave_tot[i];
    if (i < wv) base += t;
    tot += t;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// see StreamState::host_flags
__device__ __forceinline__ void publish_flags(const StreamState& S, int s, int f) {
  if (!S.host_flags) return;
  __hip_atomic_store(&S.host_flags[s], f | (S.host_seq << 8), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(TF_T) void track_finalize_kernel(KParams P, Tables T, FrameTab KM1,
                                                              FrameTab LKF, FrameTab K,
                                                              StreamState S, LkScratch lk) {
  const int s = blockIdx.x, tid = threadIdx.x;
  extern __shared__ unsigned char lds_raw[];
  long long* lkf_ids = reinterpret_cast<long long*>(lds_raw);        // [kcap]
  float* disp = reinterpret_cast<float*>(lkf_ids + P.kcap);          // [kcap]
  __shared__ int wave_tot[TF_T / 64];
  __shared__ int sh_cnt, sh_m, sh_digit, sh_rank;
  __shared__ int hist[256];
  const size_t so = (size_t)s * P.kcap;
  const long long ts = S.in_timestamp[s];

  if (!(S.flags[s] & FLAG_INIT)) {  // processFirstStereoFrame (StereoVisionImuFrontend.cpp:245-276)
    if (tid == 0) {
      K.count[s] = 0;
      K.timestamp[s] = ts;
      S.n_tracked[s] = 0;
      S.n_detected[s] = 0;
      const int f0 = (S.flags[s] & FLAG_OVERFLOW) | FLAG_KEYFRAME | FLAG_DETECT | FLAG_STEREO | FLAG_FIRST;
      S.flags[s] = f0;
      publish_flags(S, s, f0);
    }
    return;
  }
  // ---- Tracker.cpp:167-189 ---------------------------------------------------------------------
  const int n = lk.npts[s];
  if (tid == 0) sh_cnt = 0;
  __syncthreads();
  for (int base = 0; base < n; base += TF_T) {
    const int i = base + tid;
    bool keep = false;
    const int src = i < n ? lk.src_idx[so + i] : 0;
    if (i < n) keep = lk.status[so + i] && !(KM1.age[so + src] > P.max_age);
    int tot;
    const int pos = block_scan_tf(keep ? 1 : 0, wave_tot, &tot);
    const int off = sh_cnt;
    if (keep) {
      const size_t o = so + off + pos;
      const float2 p = lk.next_pts[so + i];
      K.kp[o] = p;
      K.lmk[o] = KM1.lmk[so + src];
      K.age[o] = KM1.age[so + src];
      double v[3];
      bearing_vector(T.und_left_R, p.x, p.y, v);
      K.versor[o * 3] = v[0];
      K.versor[o * 3 + 1] = v[1];
      K.versor[o * 3 + 2] = v[2];
    }
    __syncthreads();
    if (tid == 0) sh_cnt = off + tot;
    __syncthreads();
  }
  const int nk = sh_cnt;
  __syncthreads();
  if (nk == 0 && !P.mono) {  // StereoVisionImuFrontend.cpp:313-323: re-detect and move on (the mono
                             // front-end has no such early return, MonoVisionImuFrontend.cpp:248-262)
    if (tid == 0) {
      K.count[s] = 0;
      K.timestamp[s] = ts;
      S.n_tracked[s] = 0;
      S.n_detected[s] = 0;
      const int f0 = (S.flags[s] & (FLAG_OVERFLOW | FLAG_INIT)) | FLAG_DETECT;
      S.flags[s] = f0;
      publish_flags(S, s, f0);
    }
    return;
  }
  // ---- shouldBeKeyframe (VisionImuFrontend.cpp:175-232) ----------------------------------------
  // findMatchingKeypoints: landmark ids of a frame are strictly increasing (tracked ids keep their
  // order, new ids are larger than every earlier id), so the std::map lookup is a binary search.
  // (the last keyframe may hold landmark -1 entries left by its outlier rejection: they are
  // squeezed out of the search list, lkf_pos keeps the position of each remaining id)
  int* lkf_pos = reinterpret_cast<int*>(disp + P.kcap);
  const int nl_all = LKF.count[s];
  if (tid == 0) sh_cnt = 0;
  __syncthreads();
  for (int base = 0; base < nl_all; base += TF_T) {
    const int j = base + tid;
    long long id = -1;
    if (j < nl_all) id = LKF.lmk[so + j];
    int tot;
    const int pos = block_scan_tf(id != -1 ? 1 : 0, wave_tot, &tot);
    const int off = sh_cnt;
    if (id != -1) {
      lkf_ids[off + pos] = id;
      lkf_pos[off + pos] = j;
    }
    __syncthreads();
    if (tid == 0) sh_cnt = off + tot;
    __syncthreads();
  }
  const int nl = sh_cnt;
  if (tid == 0) sh_m = 0;
  __syncthreads();
  for (int i = tid; i < nk; i += TF_T) {
    const long long id = K.lmk[so + i];
    int lo = 0, hi = nl - 1, j = -1;
    while (lo <= hi) {
      const int mid = (lo + hi) >> 1;
      const long long v = lkf_ids[mid];
      if (v == id) {
        j = lkf_pos[mid];
        break;
      }
      if (v < id)
        lo = mid + 1;
      else
        hi = mid - 1;
    }
    if (j >= 0) {
      const float2 c = K.kp[so + i], r = LKF.kp[so + j];
      const float dx = c.x - r.x, dy = c.y - r.y;
      const int slot = atomicAdd(&sh_m, 1);
      disp[slot] = dx * dx + dy * dy;
    }
  }
  __syncthreads();
  const int m = sh_m;
  double disparity = 0.0;
  if (m > 0) {
    // std::nth_element(center = m/2): radix select on the (non-negative) float bit patterns
    unsigned* keys = reinterpret_cast<unsigned*>(disp);
    unsigned prefix = 0;
    int kth = m / 2;
    for (int pass = 3; pass >= 0; pass--) {
      const int shift = 8 * pass;
      for (int i = tid; i < 256; i += TF_T) hist[i] = 0;
      __syncthreads();
      for (int i = tid; i < m; i += TF_T) {
        const unsigned key = keys[i];
        if (pass == 3 || (key >> (shift + 8)) == (prefix >> (shift + 8)))
          atomicAdd(&hist[(key >> shift) & 255u], 1);
      }
      __syncthreads();
      if (tid < 64) {
        const int c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2],
                  c3 = hist[4 * tid + 3];
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
          sh_digit = 4 * tid + d;
          sh_rank = r;
        }
      }
      __syncthreads();
      prefix |= (unsigned)sh_digit << shift;
      kth = sh_rank;
      __syncthreads();
    }
    disparity = sqrt((double)__uint_as_float(prefix));
  }
  if (tid == 0) {
    const long long kf_diff_ns = ts - LKF.timestamp[s];
    const bool min_time_elapsed = (double)kf_diff_ns >= P.min_kf_ns;
    const bool max_time_elapsed = (double)kf_diff_ns >= P.max_kf_ns;
    const bool nr_features_low = (long long)nk <= P.min_features;
    const bool is_disparity_low = disparity < P.disparity_thr;
    // kfTrackingStatus_mono_ of the last keyframe (LOW_DISPARITY only arises with useRANSAC)
    // (MonoVisionImuFrontend::processFrame resets the status to INVALID before this test, :264-265)
    const bool disparity_low_first_time =
        is_disparity_low && (P.mono || !(S.trk_status[2 * (size_t)s] == TRK_LOW_DISPARITY));
    const bool enough_disparity = !is_disparity_low;
    const bool max_disparity_reached = disparity > P.max_disp_lkf;
    const bool disparity_flipped = (enough_disparity || disparity_low_first_time) && min_time_elapsed;
    const bool kf = max_time_elapsed || max_disparity_reached || disparity_flipped ||
                    nr_features_low || (S.in_force_kf[s] != 0);
    K.count[s] = nk;
    K.timestamp[s] = ts;
    S.n_tracked[s] = nk;
    S.n_detected[s] = 0;
    int f = S.flags[s] & (FLAG_OVERFLOW | FLAG_INIT);
    if (kf) f |= FLAG_KEYFRAME | FLAG_DETECT | FLAG_STEREO;
    S.flags[s] = f;
    publish_flags(S, s, f);
  }
}

int track_max_kcap() {
  const int budget = lds_dynamic_budget(reinterpret_cast<const void*>(track_finalize_kernel));
  return budget / (int)(sizeof(long long) + sizeof(float) + sizeof(int));
}

void launch_track_finalize(const KParams& P, const Tables& T, const FrameTab& km1,
                           const FrameTab& lkf, const FrameTab& k, const StreamState& S,
                           const LkScratch& lk, hipStream_t st) {
  const size_t lds = (sizeof(long long) + sizeof(float) + sizeof(int)) * (size_t)P.kcap;
  hipLaunchKernelGGL(track_finalize_kernel, dim3(P.B), dim3(TF_T), lds, st, P, T, km1, lkf, k, S,
                     lk);
}

}  // namespace kvfe
