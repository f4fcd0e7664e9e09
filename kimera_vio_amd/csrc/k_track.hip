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

namespace kvfe {

#include "kvfe_undistort.inl"

struct LevelImg {
  const unsigned char* p;
  int w, h;
  size_t stride;
};

__device__ __forceinline__ LevelImg level_img(const KParams& P, const unsigned char* img0,
                                              size_t row_stride, const unsigned char* pyr, int l) {
  LevelImg L;
  if (l == 0) {
    L.p = img0;
    L.stride = row_stride;
  } else {
    L.p = pyr + P.loff[l];
    L.stride = (size_t)P.lw[l];
  }
  L.w = P.lw[l];
  L.h = P.lh[l];
  return L;
}

__device__ __forceinline__ int at101(const LevelImg& L, int x, int y) {
  return L.p[(size_t)reflect101(y, L.h) * L.stride + reflect101(x, L.w)];
}

__device__ __forceinline__ void lk_weights(float a, float b, int* w00, int* w01, int* w10,
                                           int* w11) {
  const int W_BITS = 14;
  *w00 = __float2int_rn((1.f - a) * (1.f - b) * (1 << W_BITS));
  *w01 = __float2int_rn(a * (1.f - b) * (1 << W_BITS));
  *w10 = __float2int_rn((1.f - a) * b * (1 << W_BITS));
  *w11 = (1 << W_BITS) - *w00 - *w01 - *w10;
}

// NPX > 0: window pixels per lane held in registers (win*win <= 64*NPX), current-frame window
// staged in LDS, exact-sum fast path for the b vector.  NPX == 0: generic fallback.
template <int NPX>
__global__ __launch_bounds__(64) void lk_kernel(KParams P, const unsigned char* prev_img,
                                                size_t prev_row_stride, size_t prev_img_stride,
                                                const unsigned char* prev_pyr,
                                                const unsigned char* cur_img,
                                                size_t cur_row_stride, size_t cur_img_stride,
                                                const unsigned char* cur_pyr, LkScratch lk) {
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
  constexpr int JM = 3;                 // margin of the staged current-frame window
  const int JS = win + 1 + 2 * JM;      // staged window side
  unsigned char* jwin = patch + ((ws * ws + 3) & ~3);  // JS^2
  __shared__ float chain[16];
  // per-lane window pixels (fixed for the whole kernel)
  constexpr int NP = NPX > 0 ? NPX : 1;
  int pxy[NP];
  if (NPX > 0) {
#pragma unroll
    for (int k = 0; k < NP; k++) {
      const int e = lane + 64 * k;
      const int y = e / win, x = e - y * win;
      pxy[k] = e < w2 ? (y * 64 + x) : -1;  // packed (y, x), x < 64
    }
  }

  const unsigned char* pimg = prev_img + (size_t)s * prev_img_stride;
  const unsigned char* cimg = cur_img + (size_t)s * cur_img_stride;
  const unsigned char* ppyr = prev_pyr + (size_t)s * P.pyr_stride;
  const unsigned char* cpyr = cur_pyr + (size_t)s * P.pyr_stride;

  const size_t po = (size_t)s * P.kcap + pt;
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
    // register copies of the template / derivative window and LDS staging of the J window
    int ri[NP], rgx[NP], rgy[NP];
    int jx0 = 0, jy0 = 0;
    bool jvalid = false;
    if (NPX > 0) {
#pragma unroll
      for (int k = 0; k < NP; k++) {
        const int e = lane + 64 * k;
        const bool ok = pxy[k] >= 0;
        ri[k] = ok ? (int)Iwin[e] : 0;
        rgx[k] = ok ? (int)dIx[e] : 0;
        rgy[k] = ok ? (int)dIy[e] : 0;
      }
    }
    auto stage_j = [&](int inx, int iny) {
      jx0 = inx - JM;
      jy0 = iny - JM;
      __syncthreads();
      for (int e = lane; e < JS * JS; e += 64) {
        const int yy = e / JS, xx = e - yy * JS;
        jwin[e] = (unsigned char)at101(LJ, jx0 + xx, jy0 + yy);
      }
      __syncthreads();
      jvalid = true;
    };
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
      if (NPX > 0) {
        if (!jvalid || inx < jx0 || iny < jy0 || inx + win + 1 > jx0 + JS || iny + win + 1 > jy0 + JS)
          stage_j(inx, iny);
        const int ob = (iny - jy0) * JS + (inx - jx0);
        int p1[NP], p2[NP];
        int s1 = 0, s2 = 0;
        unsigned a1 = 0, a2 = 0;
#pragma unroll
        for (int k = 0; k < NP; k++) {
          const int yx = pxy[k] < 0 ? 0 : pxy[k];
          const unsigned char* r0 = jwin + ob + (yx >> 6) * JS + (yx & 63);
          const int t = (r0[0] * iw00 + r0[1] * iw01 + r0[JS] * iw10 + r0[JS + 1] * iw11 +
                         (1 << (W_BITS1 - 5 - 1))) >> (W_BITS1 - 5);
          const int diff = pxy[k] < 0 ? 0 : t - ri[k];
          p1[k] = diff * rgx[k];
          p2[k] = diff * rgy[k];
          s1 += p1[k];
          s2 += p2[k];
          a1 += (unsigned)abs(p1[k]);
          a2 += (unsigned)abs(p2[k]);
        }
        // exactness test: if sum|terms| < 2^24 every partial sum of every SSE lane chain is an
        // integer below 2^24, so the float chains equal the exact integer sum in any order.
        unsigned am = max(a1, a2);
        am = min(am, 1u << 25);
        for (int off = 32; off > 0; off >>= 1) {
          s1 += __shfl_xor(s1, off);
          s2 += __shfl_xor(s2, off);
          am += (unsigned)__shfl_xor((int)am, off);
        }
        if (am < (1u << 24)) {
          ib1 = (float)s1;
          ib2 = (float)s2;
        } else {
#pragma unroll
          for (int k = 0; k < NP; k++) {
            const int e = lane + 64 * k;
            if (pxy[k] >= 0) {
              prod1[e] = p1[k];
              prod2[e] = p2[k];
            }
          }
          __syncthreads();
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
          __syncthreads();
          const float bb0 = chain[0] + chain[4], bb1 = chain[1] + chain[5],
                      bb2 = chain[2] + chain[6], bb3 = chain[3] + chain[7];
          ib1 = chain[8];
          ib2 = chain[9];
          ib1 += bb0 + bb2;
          ib2 += bb1 + bb3;
          __syncthreads();
        }
      } else {
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
      }
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

static size_t lk_lds_bytes(int win) {
  const int w2 = win * win, wp = win + 1, ws = win + 3;
  size_t shorts = (size_t)3 * w2 + 2 * wp * wp;
  shorts += shorts & 1;
  const int js = win + 1 + 2 * 3;
  return shorts * 2 + sizeof(int) * 2 * w2 + (((size_t)ws * ws + 3) & ~(size_t)3) + (size_t)js * js + 16;
}

void launch_lk(const KParams& P, const unsigned char* prev_img, size_t prev_row_stride,
               size_t prev_img_stride, const unsigned char* prev_pyr, const unsigned char* cur_img,
               size_t cur_row_stride, size_t cur_img_stride, const unsigned char* cur_pyr,
               const LkScratch& lk, int max_pts, hipStream_t st) {
  if (max_pts <= 0) return;
  if (P.klt_win * P.klt_win <= 64 * 9 && P.klt_win <= 60)
    hipLaunchKernelGGL(lk_kernel<9>, dim3(max_pts, P.B), dim3(64), lk_lds_bytes(P.klt_win), st, P,
                       prev_img, prev_row_stride, prev_img_stride, prev_pyr, cur_img,
                       cur_row_stride, cur_img_stride, cur_pyr, lk);
  else
    hipLaunchKernelGGL(lk_kernel<0>, dim3(max_pts, P.B), dim3(64), lk_lds_bytes(P.klt_win), st, P,
                       prev_img, prev_row_stride, prev_img_stride, prev_pyr, cur_img,
                       cur_row_stride, cur_img_stride, cur_pyr, lk);
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
  const int n = KM1.count[s];
  for (int i = threadIdx.x; i < n; i += 256) {
    const float2 p = KM1.kp[(size_t)s * P.kcap + i];
    lk.prev_pts[(size_t)s * P.kcap + i] = p;
    lk.next_pts[(size_t)s * P.kcap + i] = use_h ? predict_point(Hs, p, P.W, P.H) : p;
  }
  if (threadIdx.x == 0) lk.npts[s] = n;
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
constexpr int TF_T = 256;

__device__ __forceinline__ int block_scan256(int v, int* wave_tot, int* total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int inc = v;
  for (int off = 1; off < 64; off <<= 1) {
    int t = __shfl_up(inc, off);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wave_tot[wv] = inc;
  __syncthreads();
  int base = 0, tot = 0;
  for (int i = 0; i < TF_T / 64; i++) {
    const int t = wave_tot[i];
    if (i < wv) base += t;
    tot += t;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
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
      S.flags[s] = (S.flags[s] & FLAG_OVERFLOW) | FLAG_KEYFRAME | FLAG_DETECT | FLAG_STEREO | FLAG_FIRST;
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
    if (i < n) keep = lk.status[so + i] && !(KM1.age[so + i] > P.max_age);
    int tot;
    const int pos = block_scan256(keep ? 1 : 0, wave_tot, &tot);
    const int off = sh_cnt;
    if (keep) {
      const size_t o = so + off + pos;
      const float2 p = lk.next_pts[so + i];
      K.kp[o] = p;
      K.lmk[o] = KM1.lmk[so + i];
      K.age[o] = KM1.age[so + i];
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
  if (nk == 0) {  // StereoVisionImuFrontend.cpp:313-323: re-detect and move on
    if (tid == 0) {
      K.count[s] = 0;
      K.timestamp[s] = ts;
      S.n_tracked[s] = 0;
      S.n_detected[s] = 0;
      S.flags[s] = (S.flags[s] & (FLAG_OVERFLOW | FLAG_INIT)) | FLAG_DETECT;
    }
    return;
  }
  // ---- shouldBeKeyframe (VisionImuFrontend.cpp:175-232) ----------------------------------------
  // findMatchingKeypoints: landmark ids of a frame are strictly increasing (tracked ids keep their
  // order, new ids are larger than every earlier id), so the std::map lookup is a binary search.
  const int nl = LKF.count[s];
  for (int i = tid; i < nl; i += TF_T) lkf_ids[i] = LKF.lmk[so + i];
  if (tid == 0) sh_m = 0;
  __syncthreads();
  for (int i = tid; i < nk; i += TF_T) {
    const long long id = K.lmk[so + i];
    int lo = 0, hi = nl - 1, j = -1;
    while (lo <= hi) {
      const int mid = (lo + hi) >> 1;
      const long long v = lkf_ids[mid];
      if (v == id) {
        j = mid;
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
    const bool disparity_low_first_time = is_disparity_low;  // status is never LOW_DISPARITY w/o RANSAC
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
  }
}

void launch_track_finalize(const KParams& P, const Tables& T, const FrameTab& km1,
                           const FrameTab& lkf, const FrameTab& k, const StreamState& S,
                           const LkScratch& lk, hipStream_t st) {
  const size_t lds = (sizeof(long long) + sizeof(float)) * (size_t)P.kcap;
  hipLaunchKernelGGL(track_finalize_kernel, dim3(P.B), dim3(TF_T), lds, st, P, T, km1, lkf, k, S,
                     lk);
}

}  // namespace kvfe
