// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
// CPU restatement of the OpenCV 4.2 dense stereo correspondence the reference reaches through
// StereoMatcher::denseStereoReconstruction (src/frontend/StereoMatcher.cpp:32-121):
// cv::StereoSGBM (MODE_HH by default, DenseStereoParams: StereoMatchingParams.h:39-58) or
// cv::StereoBM, then cv::filterSpeckles inside compute(), an optional cv::medianBlur(5), and
// cv::reprojectImageTo3D behind StereoCamera::backProjectDisparityTo3D (StereoCamera.cpp:176-196).
// The routines follow the structure of calib3d/src/stereosgbm.cpp and stereobm.cpp (generic C++ paths,
// no SIMD/OpenCL) — including their border quirks, because those decide the disparities at the image
// edge.  OpenCV is not in /root/reference; the reference has no numeric test of this path
// (tests/testStereoMatcher.cpp:131 and testStereoCamera.cpp:264 only check self-consistency):
// PARITY UNPINNED.
#include <algorithm>
#include <climits>
#include <cstdlib>

#include "ocv.hpp"

namespace ocv {

// test hooks: when set, computeDisparitySGBM (MODE_HH) copies its cost volume C (with the +P2 bias)
// and the summed path costs S, both [height][width1][D] shorts
short* g_sgbm_debug_C = nullptr;
short* g_sgbm_debug_S = nullptr;

namespace {
typedef short CostType;
typedef short DispType;
typedef uint8_t PixType;
const int DISP_SHIFT = 4;
const int DISP_SCALE = 1 << DISP_SHIFT;
const CostType MAX_COST = SHRT_MAX;

// stereosgbm.cpp calcPixelCostBT (cn == 1, full x range).  cost: width1*D entries, [x - minX1][d - minD].
void calcPixelCostBT(const uint8_t* img1, const uint8_t* img2, int width, int height, size_t step, int y,
                     int minD, int maxD, CostType* cost, const PixType* tab /* already + TAB_OFS */) {
  const int minX1 = std::max(maxD, 0), maxX1 = width + std::min(minD, 0);
  const int D = maxD - minD, width1 = maxX1 - minX1;
  const int minX2 = std::max(minX1 - maxD, 0), maxX2 = std::min(maxX1 - minD, width);
  const PixType *row1 = img1 + (size_t)y * step, *row2 = img2 + (size_t)y * step;
  // prow[c][x]: c = 0 Sobel-x response through the clip table, c = 1 raw intensity; the first and the
  // last column of both channels hold tab[0] (prow2 is stored mirrored in OpenCV: same values)
  std::vector<PixType> p1(2 * (size_t)width), p2(2 * (size_t)width);
  for (int c = 0; c < 2; c++)
    p1[width * c] = p1[width * c + width - 1] = p2[width * c] = p2[width * c + width - 1] = tab[0];
  const long n1 = y > 0 ? -(long)step : 0, s1 = y < height - 1 ? (long)step : 0;
  int minX_cmn = std::min(minX1, minX2) - 1, maxX_cmn = std::max(maxX1, maxX2) + 1;
  minX_cmn = std::max(minX_cmn, 1);
  maxX_cmn = std::min(maxX_cmn, width - 1);
  for (int x = minX_cmn; x < maxX_cmn; x++) {
    p1[x] = tab[(row1[x + 1] - row1[x - 1]) * 2 + row1[x + n1 + 1] - row1[x + n1 - 1] + row1[x + s1 + 1] -
                row1[x + s1 - 1]];
    p2[x] = tab[(row2[x + 1] - row2[x - 1]) * 2 + row2[x + n1 + 1] - row2[x + n1 - 1] + row2[x + s1 + 1] -
                row2[x + s1 - 1]];
    p1[x + width] = row1[x];
    p2[x + width] = row2[x];
  }
  std::fill(cost, cost + (size_t)width1 * D, (CostType)0);
  std::vector<PixType> lo(width), hi(width);
  for (int c = 0; c < 2; c++) {
    const PixType *prow1 = p1.data() + (size_t)width * c, *prow2 = p2.data() + (size_t)width * c;
    const int diff_scale = c < 1 ? 0 : 2;
    // v0 = min(row2[x-1/2], row2[x], row2[x+1/2]), v1 = max(...) (OpenCV fills x in (minX2, maxX2]; the
    // entries read below lie inside that range)
    for (int x = std::max(minX2, 0); x < std::min(maxX2 + 1, width); x++) {
      const int v = prow2[x];
      const int vl = x < width - 1 ? (v + prow2[x + 1]) / 2 : v;   // mirrored index x-1 <-> column x+1
      const int vr = x > 0 ? (v + prow2[x - 1]) / 2 : v;
      lo[x] = (PixType)std::min(std::min(vl, vr), v);
      hi[x] = (PixType)std::max(std::max(vl, vr), v);
    }
    for (int x = minX1; x < maxX1; x++) {
      const int u = prow1[x];
      const int ul = x > 0 ? (u + prow1[x - 1]) / 2 : u;
      const int ur = x < width - 1 ? (u + prow1[x + 1]) / 2 : u;
      const int u0 = std::min(std::min(ul, ur), u), u1 = std::max(std::max(ul, ur), u);
      CostType* cx = cost + (size_t)(x - minX1) * D;
      for (int d = minD; d < maxD; d++) {
        const int v = prow2[x - d], v0 = lo[x - d], v1 = hi[x - d];
        const int c0 = std::max(std::max(0, u - v1), v0 - u);
        const int c1 = std::max(std::max(0, v - u1), u0 - v);
        cx[d - minD] = (CostType)(cx[d - minD] + (std::min(c0, c1) >> diff_scale));
      }
    }
  }
}

inline CostType sat_cost(int v) { return (CostType)(v < SHRT_MIN ? SHRT_MIN : v > SHRT_MAX ? SHRT_MAX : v); }

// stereosgbm.cpp computeDisparitySGBM (MODE_SGBM: one pass, 5 directions; MODE_HH: two passes, 8).
void computeDisparitySGBM(const uint8_t* img1, const uint8_t* img2, int width, int height, size_t step,
                          const StereoSGBMParams& params, DispType* disp1, size_t dstep) {
  const int minD = params.minDisparity, maxD = minD + params.numDisparities;
  const int SADWindowSize = params.blockSize > 0 ? params.blockSize : 5;
  const int ftzero = std::max(params.preFilterCap, 15) | 1;
  const int uniquenessRatio = params.uniquenessRatio >= 0 ? params.uniquenessRatio : 10;
  const int disp12MaxDiff = params.disp12MaxDiff > 0 ? params.disp12MaxDiff : 1;
  const int P1 = params.P1 > 0 ? params.P1 : 2, P2 = std::max(params.P2 > 0 ? params.P2 : 5, P1 + 1);
  const int minX1 = std::max(maxD, 0), maxX1 = width + std::min(minD, 0);
  const int D = maxD - minD, width1 = maxX1 - minX1;
  const int INVALID_DISP = minD - 1, INVALID_DISP_SCALED = INVALID_DISP * DISP_SCALE;
  const int SW2 = SADWindowSize / 2, SH2 = SADWindowSize / 2;
  const bool fullDP = params.mode == STEREO_SGBM_MODE_HH;
  const int npasses = fullDP ? 2 : 1;
  const int TAB_OFS = 256 * 4, TAB_SIZE = 256 + TAB_OFS * 2;
  std::vector<PixType> clipTab(TAB_SIZE);
  for (int k = 0; k < TAB_SIZE; k++)
    clipTab[k] = (PixType)(std::min(std::max(k - TAB_OFS, -ftzero), ftzero) + ftzero);

  if (minX1 >= maxX1) {
    for (int y = 0; y < height; y++)
      for (int x = 0; x < width; x++) disp1[y * dstep + x] = (DispType)INVALID_DISP_SCALED;
    return;
  }
  // per x: NR2 = 8 direction slots of D2 = D + 16 entries, one border column either side, two rows
  const int NR2 = 8, D2 = D + 16, NRD2 = NR2 * D2;
  const size_t costBufSize = (size_t)width1 * D;
  const size_t CSBufSize = costBufSize * (fullDP ? height : 1);
  const size_t minLrSize = (size_t)(width1 + 2) * NR2, LrSize = minLrSize * D2;
  const int hsumBufNRows = SH2 * 2 + 2;
  std::vector<CostType> Cbuf(CSBufSize, (CostType)P2), Sbuf(CSBufSize, 0);   // "add P2 to every C(x,y)"
  std::vector<CostType> hsumBuf(costBufSize * hsumBufNRows), pixDiff(costBufSize);
  std::vector<CostType> LrStore[2], minLrStore[2];
  std::vector<CostType> disp2cost(width);
  std::vector<DispType> disp2ptr(width);

  for (int pass = 1; pass <= npasses; pass++) {
    int x1, y1, x2, y2, dx, dy;
    if (pass == 1) {
      y1 = 0; y2 = height; dy = 1; x1 = 0; x2 = width1; dx = 1;
    } else {
      y1 = height - 1; y2 = -1; dy = -1; x1 = width1 - 1; x2 = -1; dx = -1;
    }
    CostType *Lr[2], *minLr[2];
    for (int k = 0; k < 2; k++) {
      LrStore[k].assign(LrSize + 16, 0);
      minLrStore[k].assign(minLrSize, 0);
      Lr[k] = LrStore[k].data() + NRD2 + 8;
      minLr[k] = minLrStore[k].data() + NR2;
    }
    for (int y = y1; y != y2; y += dy) {
      DispType* disp1ptr = disp1 + (size_t)y * dstep;
      CostType* C = Cbuf.data() + (!fullDP ? 0 : y * costBufSize);
      CostType* S = Sbuf.data() + (!fullDP ? 0 : y * costBufSize);
      if (pass == 1) {   // compute C on the first pass, reuse it on the second
        const int dy1 = y == 0 ? 0 : y + SH2, dy2 = y == 0 ? SH2 : dy1;
        for (int k = dy1; k <= dy2; k++) {
          CostType* hsumAdd = hsumBuf.data() + (std::min(k, height - 1) % hsumBufNRows) * costBufSize;
          if (k < height) {
            calcPixelCostBT(img1, img2, width, height, step, k, minD, maxD, pixDiff.data(),
                            clipTab.data() + TAB_OFS);
            std::fill(hsumAdd, hsumAdd + D, (CostType)0);
            for (int x = 0; x <= SW2 * D; x += D) {
              const int scale = x == 0 ? SW2 + 1 : 1;
              for (int d = 0; d < D; d++) hsumAdd[d] = (CostType)(hsumAdd[d] + pixDiff[x + d] * scale);
            }
            if (y > 0) {
              const CostType* hsumSub =
                  hsumBuf.data() + (std::max(y - SH2 - 1, 0) % hsumBufNRows) * costBufSize;
              const CostType* Cprev = !fullDP || y == 0 ? C : C - costBufSize;
              for (int x = D; x < width1 * D; x += D) {
                const CostType* pixAdd = pixDiff.data() + std::min(x + SW2 * D, (width1 - 1) * D);
                const CostType* pixSub = pixDiff.data() + std::max(x - (SW2 + 1) * D, 0);
                for (int d = 0; d < D; d++) {
                  const int hv = hsumAdd[x + d] = (CostType)(hsumAdd[x - D + d] + pixAdd[d] - pixSub[d]);
                  C[x + d] = (CostType)(Cprev[x + d] + hv - hsumSub[x + d]);
                }
              }
            } else {
              for (int x = D; x < width1 * D; x += D) {
                const CostType* pixAdd = pixDiff.data() + std::min(x + SW2 * D, (width1 - 1) * D);
                const CostType* pixSub = pixDiff.data() + std::max(x - (SW2 + 1) * D, 0);
                for (int d = 0; d < D; d++)
                  hsumAdd[x + d] = (CostType)(hsumAdd[x - D + d] + pixAdd[d] - pixSub[d]);
              }
            }
          }
          if (y == 0) {
            const int scale = k == 0 ? SH2 + 1 : 1;
            for (int x = 0; x < width1 * D; x++) C[x] = (CostType)(C[x] + hsumAdd[x] * scale);
          }
        }
        for (int k = 0; k < width1 * D; k++) S[k] = 0;
      }
      // clear the left and the right borders of the current row's buffers
      std::fill(Lr[0] - NRD2 - 8, Lr[0] - 8, (CostType)0);
      std::fill(Lr[0] + (size_t)width1 * NRD2 - 8, Lr[0] + (size_t)width1 * NRD2 - 8 + NRD2, (CostType)0);
      std::fill(minLr[0] - NR2, minLr[0], (CostType)0);
      std::fill(minLr[0] + (size_t)width1 * NR2, minLr[0] + (size_t)width1 * NR2 + NR2, (CostType)0);
      // L_r(p,d) = C(p,d) + min(L_r(p-r,d), L_r(p-r,d-1)+P1, L_r(p-r,d+1)+P1, min_k L_r(p-r,k)+P2)
      //            - min_k L_r(p-r,k)     for r = (-dx,0), (-1,-dy), (0,-dy), (1,-dy)
      for (int x = x1; x != x2; x += dx) {
        const int xm = x * NR2, xd = xm * D2;
        const int delta0 = minLr[0][xm - dx * NR2] + P2, delta1 = minLr[1][xm - NR2 + 1] + P2;
        const int delta2 = minLr[1][xm + 2] + P2, delta3 = minLr[1][xm + NR2 + 3] + P2;
        CostType* Lr_p0 = Lr[0] + xd - dx * NRD2;
        CostType* Lr_p1 = Lr[1] + xd - NRD2 + D2;
        CostType* Lr_p2 = Lr[1] + xd + D2 * 2;
        CostType* Lr_p3 = Lr[1] + xd + NRD2 + D2 * 3;
        Lr_p0[-1] = Lr_p0[D] = Lr_p1[-1] = Lr_p1[D] = Lr_p2[-1] = Lr_p2[D] = Lr_p3[-1] = Lr_p3[D] = MAX_COST;
        CostType* Lr_p = Lr[0] + xd;
        const CostType* Cp = C + (size_t)x * D;
        CostType* Sp = S + (size_t)x * D;
        int minL0 = MAX_COST, minL1 = MAX_COST, minL2 = MAX_COST, minL3 = MAX_COST;
        for (int d = 0; d < D; d++) {
          const int Cpd = Cp[d];
          const int L0 = Cpd + std::min((int)Lr_p0[d], std::min(Lr_p0[d - 1] + P1, std::min(Lr_p0[d + 1] + P1, delta0))) - delta0;
          const int L1 = Cpd + std::min((int)Lr_p1[d], std::min(Lr_p1[d - 1] + P1, std::min(Lr_p1[d + 1] + P1, delta1))) - delta1;
          const int L2 = Cpd + std::min((int)Lr_p2[d], std::min(Lr_p2[d - 1] + P1, std::min(Lr_p2[d + 1] + P1, delta2))) - delta2;
          const int L3 = Cpd + std::min((int)Lr_p3[d], std::min(Lr_p3[d - 1] + P1, std::min(Lr_p3[d + 1] + P1, delta3))) - delta3;
          Lr_p[d] = (CostType)L0;
          minL0 = std::min(minL0, L0);
          Lr_p[d + D2] = (CostType)L1;
          minL1 = std::min(minL1, L1);
          Lr_p[d + D2 * 2] = (CostType)L2;
          minL2 = std::min(minL2, L2);
          Lr_p[d + D2 * 3] = (CostType)L3;
          minL3 = std::min(minL3, L3);
          Sp[d] = sat_cost(Sp[d] + L0 + L1 + L2 + L3);
        }
        minLr[0][xm] = (CostType)minL0;
        minLr[0][xm + 1] = (CostType)minL1;
        minLr[0][xm + 2] = (CostType)minL2;
        minLr[0][xm + 3] = (CostType)minL3;
      }
      if (pass == npasses) {
        for (int x = 0; x < width; x++) {
          disp1ptr[x] = disp2ptr[x] = (DispType)INVALID_DISP_SCALED;
          disp2cost[x] = MAX_COST;
        }
        for (int x = width1 - 1; x >= 0; x--) {
          CostType* Sp = S + (size_t)x * D;
          int minS = MAX_COST, bestDisp = -1, d;
          if (npasses == 1) {
            const int xm = x * NR2, xd = xm * D2;
            int minL0 = MAX_COST;
            const int delta0 = minLr[0][xm + NR2] + P2;
            CostType* Lr_p0 = Lr[0] + xd + NRD2;
            Lr_p0[-1] = Lr_p0[D] = MAX_COST;
            CostType* Lr_p = Lr[0] + xd;
            const CostType* Cp = C + (size_t)x * D;
            for (d = 0; d < D; d++) {
              const int L0 = Cp[d] + std::min((int)Lr_p0[d], std::min(Lr_p0[d - 1] + P1, std::min(Lr_p0[d + 1] + P1, delta0))) - delta0;
              Lr_p[d] = (CostType)L0;
              minL0 = std::min(minL0, L0);
              const int Sval = Sp[d] = sat_cost(Sp[d] + L0);
              if (Sval < minS) {
                minS = Sval;
                bestDisp = d;
              }
            }
            minLr[0][xm] = (CostType)minL0;
          } else {
            for (d = 0; d < D; d++) {
              const int Sval = Sp[d];
              if (Sval < minS) {
                minS = Sval;
                bestDisp = d;
              }
            }
          }
          for (d = 0; d < D; d++)
            if (Sp[d] * (100 - uniquenessRatio) < minS * 100 && std::abs(bestDisp - d) > 1) break;
          if (d < D) continue;
          d = bestDisp;
          const int _x2 = x + minX1 - d - minD;
          if (disp2cost[_x2] > minS) {
            disp2cost[_x2] = (CostType)minS;
            disp2ptr[_x2] = (DispType)(d + minD);
          }
          if (0 < d && d < D - 1) {
            // parabola through (d-1, Sp[d-1]), (d, Sp[d]), (d+1, Sp[d+1])
            const int denom2 = std::max(Sp[d - 1] + Sp[d + 1] - 2 * Sp[d], 1);
            d = d * DISP_SCALE + ((Sp[d - 1] - Sp[d + 1]) * DISP_SCALE + denom2) / (denom2 * 2);
          } else
            d *= DISP_SCALE;
          disp1ptr[x + minX1] = (DispType)(d + minD * DISP_SCALE);
        }
        for (int x = minX1; x < maxX1; x++) {
          // left-right check against both roundings of the disparity
          const int d1 = disp1ptr[x];
          if (d1 == INVALID_DISP_SCALED) continue;
          const int _d = d1 >> DISP_SHIFT, d_ = (d1 + DISP_SCALE - 1) >> DISP_SHIFT;
          const int _x = x - _d, x_ = x - d_;
          if (0 <= _x && _x < width && disp2ptr[_x] >= minD && std::abs(disp2ptr[_x] - _d) > disp12MaxDiff &&
              0 <= x_ && x_ < width && disp2ptr[x_] >= minD && std::abs(disp2ptr[x_] - d_) > disp12MaxDiff)
            disp1ptr[x] = (DispType)INVALID_DISP_SCALED;
        }
      }
      std::swap(Lr[0], Lr[1]);
      std::swap(minLr[0], minLr[1]);
    }
  }
  if (fullDP && g_sgbm_debug_C) std::memcpy(g_sgbm_debug_C, Cbuf.data(), CSBufSize * sizeof(CostType));
  if (fullDP && g_sgbm_debug_S) std::memcpy(g_sgbm_debug_S, Sbuf.data(), CSBufSize * sizeof(CostType));
}

// median of a 16-bit neighbourhood (any selection algorithm gives OpenCV's sorting-network result)
template <int N>
inline short median_n(short* v) {
  std::nth_element(v, v + N / 2, v + N);
  return v[N / 2];
}
}  // namespace

// cv::medianBlur(src CV_16S, dst, ksize 3|5): replicated borders, works on a copy when in place.
void medianBlur16s(const short* src, int w, int h, size_t sstride, short* dst, size_t dstride, int ksize) {
  std::vector<short> copy((size_t)w * h);
  for (int y = 0; y < h; y++) std::memcpy(&copy[(size_t)y * w], src + (size_t)y * sstride, sizeof(short) * w);
  const int r = ksize / 2;
  short v[25];
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int n = 0;
      for (int j = -r; j <= r; j++) {
        const int yy = std::min(std::max(y + j, 0), h - 1);
        for (int i = -r; i <= r; i++) v[n++] = copy[(size_t)yy * w + std::min(std::max(x + i, 0), w - 1)];
      }
      dst[(size_t)y * dstride + x] = ksize == 3 ? median_n<9>(v) : median_n<25>(v);
    }
}

// cv::filterSpeckles(img CV_16S, newVal, maxSpeckleSize, maxDiff): 4-connected flood fill in scan order.
void filterSpeckles16s(short* img, int width, int height, size_t step, int newVal, int maxSpeckleSize,
                       int maxDiff) {
  const size_t npixels = (size_t)width * height;
  std::vector<int> labels(npixels, 0);
  std::vector<uint8_t> rtype(npixels + 1, 0);
  std::vector<int> wx(npixels), wy(npixels);
  int curlabel = 0;
  for (int i = 0; i < height; i++) {
    short* ds = img + (size_t)i * step;
    int* ls = labels.data() + (size_t)width * i;
    for (int j = 0; j < width; j++) {
      if (ds[j] == newVal) continue;
      if (ls[j]) {
        if (rtype[ls[j]]) ds[j] = (short)newVal;   // small region: zero out the disparity
        continue;
      }
      size_t top = 0;
      int px = j, py = i, count = 0;
      curlabel++;
      ls[j] = curlabel;
      for (;;) {
        count++;
        const short* dpp = img + (size_t)py * step + px;
        const int dp = *dpp;
        int* lpp = labels.data() + (size_t)width * py + px;
        if (py < height - 1 && !lpp[+width] && dpp[+(long)step] != newVal && std::abs(dp - dpp[+(long)step]) <= maxDiff) {
          lpp[+width] = curlabel;
          wx[top] = px; wy[top++] = py + 1;
        }
        if (py > 0 && !lpp[-width] && dpp[-(long)step] != newVal && std::abs(dp - dpp[-(long)step]) <= maxDiff) {
          lpp[-width] = curlabel;
          wx[top] = px; wy[top++] = py - 1;
        }
        if (px < width - 1 && !lpp[+1] && dpp[+1] != newVal && std::abs(dp - dpp[+1]) <= maxDiff) {
          lpp[+1] = curlabel;
          wx[top] = px + 1; wy[top++] = py;
        }
        if (px > 0 && !lpp[-1] && dpp[-1] != newVal && std::abs(dp - dpp[-1]) <= maxDiff) {
          lpp[-1] = curlabel;
          wx[top] = px - 1; wy[top++] = py;
        }
        if (top == 0) break;
        --top;
        px = wx[top]; py = wy[top];
      }
      if (count <= maxSpeckleSize) {
        rtype[ls[j]] = 1;
        ds[j] = (short)newVal;
      } else
        rtype[ls[j]] = 0;
    }
  }
}

// cv::StereoSGBM::compute: computeDisparitySGBM, medianBlur(3), filterSpeckles (stereosgbm.cpp
// StereoSGBMImpl::compute; MODE_SGBM_3WAY / MODE_HH4 are not used by the reference).
void stereoSGBM_compute(const uint8_t* left, const uint8_t* right, int w, int h, size_t stride,
                        const StereoSGBMParams& p, short* disp, size_t dstride) {
  computeDisparitySGBM(left, right, w, h, stride, p, disp, dstride);
  medianBlur16s(disp, w, h, dstride, disp, dstride, 3);
  if (p.speckleWindowSize > 0)
    filterSpeckles16s(disp, w, h, dstride, (p.minDisparity - 1) * DISP_SCALE, p.speckleWindowSize,
                      DISP_SCALE * p.speckleRange);
}

// ---------------------------------------------------------------------------------------------
// cv::StereoBM (stereobm.cpp, PREFILTER_XSOBEL, CV_16S output)
// ---------------------------------------------------------------------------------------------
namespace {
void prefilterXSobel(const uint8_t* src, int w, int h, size_t sstep, uint8_t* dst, int ftzero) {
  const int OFS = 256 * 4, TABSZ = OFS * 2 + 256;
  std::vector<uint8_t> tab(TABSZ);
  for (int x = 0; x < TABSZ; x++)
    tab[x] = (uint8_t)(x - OFS < -ftzero ? 0 : x - OFS > ftzero ? ftzero * 2 : x - OFS + ftzero);
  const uint8_t val0 = tab[0 + OFS];
  int y;
  for (y = 0; y < h - 1; y += 2) {
    const uint8_t* srow1 = src + (size_t)y * sstep;
    const uint8_t* srow0 = y > 0 ? srow1 - sstep : h > 1 ? srow1 + sstep : srow1;
    const uint8_t* srow2 = y < h - 1 ? srow1 + sstep : h > 1 ? srow1 - sstep : srow1;
    const uint8_t* srow3 = y < h - 2 ? srow1 + sstep * 2 : srow1;
    uint8_t* dptr0 = dst + (size_t)y * w;
    uint8_t* dptr1 = dptr0 + w;
    dptr0[0] = dptr0[w - 1] = dptr1[0] = dptr1[w - 1] = val0;
    for (int x = 1; x < w - 1; x++) {
      const int d0 = srow0[x + 1] - srow0[x - 1], d1 = srow1[x + 1] - srow1[x - 1],
                d2 = srow2[x + 1] - srow2[x - 1], d3 = srow3[x + 1] - srow3[x - 1];
      dptr0[x] = tab[d0 + d1 * 2 + d2 + OFS];
      dptr1[x] = tab[d1 + d2 * 2 + d3 + OFS];
    }
  }
  for (; y < h; y++) std::fill(dst + (size_t)y * w, dst + (size_t)y * w + w, val0);
}

struct Rect {
  int x, y, width, height;
  bool empty() const { return width <= 0 || height <= 0; }
};
Rect intersect(const Rect& a, const Rect& b) {
  const int x1 = std::max(a.x, b.x), y1 = std::max(a.y, b.y);
  const int x2 = std::min(a.x + a.width, b.x + b.width), y2 = std::min(a.y + a.height, b.y + b.height);
  if (x2 <= x1 || y2 <= y1) return Rect{0, 0, 0, 0};
  return Rect{x1, y1, x2 - x1, y2 - y1};
}
// cv::getValidDisparityROI
Rect getValidDisparityROI(Rect roi1, Rect roi2, int minDisparity, int numberOfDisparities, int blockSize) {
  const int SW2 = blockSize / 2;
  const int minD = minDisparity, maxD = minDisparity + numberOfDisparities - 1;
  const int xmin = std::max(roi1.x, roi2.x + maxD) + SW2;
  const int xmax = std::min(roi1.x + roi1.width, roi2.x + roi2.width - minD) - SW2;
  const int ymin = std::max(roi1.y, roi2.y) + SW2;
  const int ymax = std::min(roi1.y + roi1.height, roi2.y + roi2.height) - SW2;
  Rect r{xmin, ymin, xmax - xmin, ymax - ymin};
  return r.width > 0 && r.height > 0 ? r : Rect{0, 0, 0, 0};
}

// stereobm.cpp findStereoCorrespondenceBM (generic path) on rows [row0, row1) of the pre-filtered
// images; _dy0/_dy1 = rows available above / below the band.
void findStereoCorrespondenceBM(const uint8_t* left, const uint8_t* right, int width, int height,
                                size_t sstep, short* disp, size_t dstep, const StereoBMParams& state,
                                int _dy0, int _dy1) {
  const int wsz = state.blockSize, wsz2 = wsz / 2;
  const int dy0 = std::min(_dy0, wsz2 + 1), dy1 = std::min(_dy1, wsz2 + 1);
  const int ndisp = state.numDisparities, mindisp = state.minDisparity;
  const int lofs = std::max(ndisp - 1 + mindisp, 0), rofs = -std::min(ndisp - 1 + mindisp, 0);
  const int width1 = width - rofs - ndisp + 1;
  const int ftzero = state.preFilterCap;
  const int textureThreshold = state.textureThreshold, uniquenessRatio = state.uniquenessRatio;
  const short FILTERED = (short)((mindisp - 1) << DISP_SHIFT);
  const uint8_t *lptr0 = left + lofs, *rptr0 = right + rofs;
  const int cstep = (height + dy0 + dy1) * ndisp;
  uint8_t tab[256];
  for (int x = 0; x < 256; x++) tab[x] = (uint8_t)std::abs(x - ftzero);

  std::vector<int> sadStore(ndisp + 2);
  int* sad = sadStore.data() + 1;
  std::vector<int> hsadStore((size_t)(height + dy0 + dy1) * ndisp, 0);
  int* hsad0 = hsadStore.data() + (size_t)dy0 * ndisp;
  std::vector<int> htextStore(height + wsz + 2, 0);
  int* htext = htextStore.data() + wsz2 + 1;
  std::vector<uint8_t> cbufStore((size_t)cstep * (wsz + 1));
  uint8_t* cbuf0 = cbufStore.data() + (size_t)dy0 * ndisp;

  for (int x = -wsz2 - 1; x < wsz2; x++) {
    int* hsad = hsad0 - dy0 * ndisp;
    uint8_t* cbuf = cbuf0 + (size_t)(x + wsz2 + 1) * cstep - dy0 * ndisp;
    const uint8_t* lptr = lptr0 + std::min(std::max(x, -lofs), width - lofs - 1) - (long)dy0 * (long)sstep;
    const uint8_t* rptr = rptr0 + std::min(std::max(x, -rofs), width - rofs - ndisp) - (long)dy0 * (long)sstep;
    for (int y = -dy0; y < height + dy1; y++, hsad += ndisp, cbuf += ndisp, lptr += sstep, rptr += sstep) {
      const int lval = lptr[0];
      for (int d = 0; d < ndisp; d++) {
        const int diff = std::abs(lval - rptr[d]);
        cbuf[d] = (uint8_t)diff;
        hsad[d] = hsad[d] + diff;
      }
      htext[y] += tab[lval];
    }
  }
  for (int y = 0; y < height; y++) {
    for (int x = 0; x < lofs; x++) disp[y * dstep + x] = FILTERED;
    for (int x = lofs + width1; x < width; x++) disp[y * dstep + x] = FILTERED;
  }
  short* dptr = disp + lofs;
  for (int x = 0; x < width1; x++, dptr++) {
    const int x0 = x - wsz2 - 1, x1 = x + wsz2;
    const uint8_t* cbuf_sub = cbuf0 + (size_t)((x0 + wsz2 + 1) % (wsz + 1)) * cstep - dy0 * ndisp;
    uint8_t* cbuf = cbuf0 + (size_t)((x1 + wsz2 + 1) % (wsz + 1)) * cstep - dy0 * ndisp;
    int* hsad = hsad0 - dy0 * ndisp;
    const uint8_t* lptr_sub = lptr0 + std::min(std::max(x0, -lofs), width - 1 - lofs) - (long)dy0 * (long)sstep;
    const uint8_t* lptr = lptr0 + std::min(std::max(x1, -lofs), width - 1 - lofs) - (long)dy0 * (long)sstep;
    const uint8_t* rptr = rptr0 + std::min(std::max(x1, -rofs), width - ndisp - rofs) - (long)dy0 * (long)sstep;
    for (int y = -dy0; y < height + dy1;
         y++, cbuf += ndisp, cbuf_sub += ndisp, hsad += ndisp, lptr += sstep, lptr_sub += sstep, rptr += sstep) {
      const int lval = lptr[0];
      for (int d = 0; d < ndisp; d++) {
        const int diff = std::abs(lval - rptr[d]);
        cbuf[d] = (uint8_t)diff;
        hsad[d] = hsad[d] + diff - cbuf_sub[d];
      }
      htext[y] += tab[lval] - tab[lptr_sub[0]];
    }
    for (int y = dy1; y <= wsz2; y++) htext[height + y] = htext[height + dy1 - 1];
    for (int y = -wsz2 - 1; y < -dy0; y++) htext[y] = htext[-dy0];
    for (int d = 0; d < ndisp; d++) sad[d] = hsad0[d - ndisp * dy0] * (wsz2 + 2 - dy0);
    hsad = hsad0 + (1 - dy0) * ndisp;
    for (int y = 1 - dy0; y < wsz2; y++, hsad += ndisp)
      for (int d = 0; d < ndisp; d++) sad[d] = sad[d] + hsad[d];
    int tsum = 0;
    for (int y = -wsz2 - 1; y < wsz2; y++) tsum += htext[y];
    for (int y = 0; y < height; y++) {
      int minsad = INT_MAX, mind = -1;
      hsad = hsad0 + std::min(y + wsz2, height + dy1 - 1) * ndisp;
      const int* hsad_sub = hsad0 + std::max(y - wsz2 - 1, -dy0) * ndisp;
      for (int d = 0; d < ndisp; d++) {
        const int currsad = sad[d] + hsad[d] - hsad_sub[d];
        sad[d] = currsad;
        if (currsad < minsad) {
          minsad = currsad;
          mind = d;
        }
      }
      tsum += htext[y + wsz2] - htext[y - wsz2 - 1];
      if (tsum < textureThreshold) {
        if (lofs + x < width) dptr[y * dstep] = FILTERED;
        continue;
      }
      if (uniquenessRatio > 0) {
        const int thresh = minsad + (minsad * uniquenessRatio / 100);
        int d;
        for (d = 0; d < ndisp; d++)
          if ((d < mind - 1 || d > mind + 1) && sad[d] <= thresh) break;
        if (d < ndisp) {
          if (lofs + x < width) dptr[y * dstep] = FILTERED;
          continue;
        }
      }
      sad[-1] = sad[1];
      sad[ndisp] = sad[ndisp - 2];
      const int p = sad[mind + 1], n = sad[mind - 1];
      const int dd = p + n - 2 * sad[mind] + std::abs(p - n);
      // With minDisparity > 0 OpenCV's x loop runs minDisparity columns past the end of the row (lofs + width1
      // = width + minDisparity): the writes land in the first columns of the next row, which the invoker
      // overwrites with FILTERED afterwards (roi.x > maxD) — except below the last row of the band, where
      // they are a stray write.  The restatement drops those out-of-row writes.
      if (lofs + x >= width) continue;
      dptr[y * dstep] = (short)(((ndisp - mind - 1 + mindisp) * 256 + (dd != 0 ? (p - n) * 256 / dd : 0) + 15) >> 4);
    }
  }
}
}  // namespace

// cv::StereoBM::compute (CV_16S): prefilter, block matching inside the valid-disparity rectangle,
// FILTERED elsewhere, filterSpeckles.  disp12MaxDiff >= 0 (validateDisparity) is not restated.
void stereoBM_compute(const uint8_t* left0, const uint8_t* right0, int width, int height, size_t stride,
                      const StereoBMParams& params, short* disp, size_t dstride) {
  const int FILTERED = (params.minDisparity - 1) << DISP_SHIFT;
  const int mindisp = params.minDisparity, ndisp = params.numDisparities;
  const int lofs = std::max(ndisp - 1 + mindisp, 0), rofs = -std::min(ndisp - 1 + mindisp, 0);
  const int width1 = width - rofs - ndisp + 1;
  auto fill_rows = [&](int r0, int r1, int c0, int c1) {
    for (int y = r0; y < r1; y++)
      for (int x = c0; x < c1; x++) disp[(size_t)y * dstride + x] = (short)FILTERED;
  };
  if (lofs >= width || rofs >= width || width1 < 1) {
    fill_rows(0, height, 0, width);
    return;
  }
  std::vector<uint8_t> left((size_t)width * height), right((size_t)width * height);
  prefilterXSobel(left0, width, height, stride, left.data(), params.preFilterCap);
  prefilterXSobel(right0, width, height, stride, right.data(), params.preFilterCap);
  const Rect full{0, 0, width, height};
  const Rect R1{params.roi1[0], params.roi1[1], params.roi1[2], params.roi1[3]};
  const Rect R2{params.roi2[0], params.roi2[1], params.roi2[2], params.roi2[3]};
  const Rect valid = getValidDisparityROI(!R1.empty() ? R1 : full, !R2.empty() ? R2 : full, params.minDisparity,
                                          params.numDisparities, params.blockSize);
  // FindStereoCorrespInvoker with one stripe (the result does not depend on the striping: bands see
  // min(rows available, wsz2+1) real rows above and below)
  const Rect roi = intersect(valid, full);
  if (roi.height == 0) {
    // (OpenCV returns from the stripe body without writing: the output keeps whatever create() left;
    // the restatement defines it as FILTERED)
    fill_rows(0, height, 0, width);
    return;
  }
  const int row0 = roi.y, row1 = roi.y + roi.height;
  fill_rows(0, row0, 0, width);
  fill_rows(row1, height, 0, width);
  findStereoCorrespondenceBM(left.data() + (size_t)row0 * width, right.data() + (size_t)row0 * width, width,
                             row1 - row0, width, disp + (size_t)row0 * dstride, dstride, params, row0,
                             height - row1);
  if (roi.x > 0) fill_rows(row0, row1, 0, roi.x);
  if (roi.x + roi.width < width) fill_rows(row0, row1, roi.x + roi.width, width);
  if (params.speckleRange >= 0 && params.speckleWindowSize > 0)
    filterSpeckles16s(disp, width, height, dstride, FILTERED, params.speckleWindowSize, params.speckleRange);
}

// cv::reprojectImageTo3D(disparity CV_32F, _3dImage CV_32FC3, Q, handleMissingValues, CV_32F)
// (calib3d/src/calibration.cpp).  Reference: StereoCamera.cpp:193-194.
// (optimize("O2"): g++ 11 -O3 drops the double -> float -> double round trip of the Vec3d -> Vec3f
// conversion below when it vectorises the loop, which changes results by one float ulp)
__attribute__((optimize("O2"))) void reprojectImageTo3D(const float* disparity, int w, int h, size_t dstride,
                                                        const double Q[16], bool handleMissingValues,
                                                        float* xyz /* h*w*3 */) {
  const float bigZ = 10000.f;
  double minDisparity = 3.402823466e+38;   // FLT_MAX
  if (handleMissingValues) {
    minDisparity = disparity[0];
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) minDisparity = std::min(minDisparity, (double)disparity[(size_t)y * dstride + x]);
  }
  for (int y = 0; y < h; y++) {
    const float* sptr = disparity + (size_t)y * dstride;
    float* dptr = xyz + (size_t)y * w * 3;
    for (int x = 0; x < w; x++) {
      const double d = sptr[x];
      const double v[4] = {(double)x, (double)y, d, 1.0};
      double hom[4];
      for (int i = 0; i < 4; i++) {   // Matx44d * Vec4d
        double s = 0;
        for (int k = 0; k < 4; k++) s += Q[i * 4 + k] * v[k];
        hom[i] = s;
      }
      // Vec3f = Vec3d(hom); Vec3f /= hom[3]  (Vec /= double multiplies by 1./alpha in double)
      const double ialpha = 1. / hom[3];
      for (int i = 0; i < 3; i++) dptr[x * 3 + i] = (float)((float)hom[i] * ialpha);
      if (std::fabs(d - minDisparity) <= 1.192092896e-07) dptr[x * 3 + 2] = bigZ;
    }
  }
}

}  // namespace ocv
