// TEST INFRASTRUCTURE — see ocv.hpp.  Restatement of OpenCV 4.2 video/lkpyramid.cpp:
// buildOpticalFlowPyramid, calcSharrDeriv and LKTrackerInvoker::operator() in the
// x86-64 build (CV_SSE2 accumulation order: four float lanes per accumulator,
// horizontally added at the end; integer window/derivative arithmetic is exact).
// Reference call site: src/frontend/Tracker.cpp:137-146.
#include <algorithm>
#include <cfloat>

#include "ocv.hpp"

namespace ocv {

int buildPyramid(const uint8_t* img, int w, int h, size_t stride, int win, int maxLevel,
                 Pyramid& pyr) {
  pyr.img.clear();
  pyr.w.clear();
  pyr.h.clear();
  int sw = w, sh = h;
  for (int level = 0; level <= maxLevel; ++level) {
    std::vector<uint8_t> cur((size_t)sw * sh);
    if (level == 0) {
      for (int y = 0; y < h; y++) std::memcpy(&cur[(size_t)y * w], img + (size_t)y * stride, w);
    } else {
      pyrDown(pyr.img[level - 1].data(), pyr.w[level - 1], pyr.h[level - 1], pyr.w[level - 1],
              cur.data(), sw, sh, sw);
    }
    pyr.img.push_back(std::move(cur));
    pyr.w.push_back(sw);
    pyr.h.push_back(sh);
    sw = (sw + 1) / 2;
    sh = (sh + 1) / 2;
    if (sw <= win || sh <= win) return level;
  }
  return maxLevel;
}

namespace {

// level image extended by BORDER_REFLECT_101 (copyMakeBorder in
// buildOpticalFlowPyramid) and its Scharr derivative extended by zeros
// (copyMakeBorder(..., BORDER_CONSTANT) in calcOpticalFlowPyrLK).
struct Level {
  int w, h, pad;
  std::vector<uint8_t> I;   // (w+2pad) x (h+2pad)
  std::vector<short> dI;    // (w+2pad) x (h+2pad) x 2, zero outside
  int stepI() const { return w + 2 * pad; }
  const uint8_t* iptr() const { return I.data() + (size_t)pad * stepI() + pad; }
  const short* dptr() const { return dI.data() + ((size_t)pad * stepI() + pad) * 2; }
};

void make_level(const uint8_t* img, int w, int h, int pad, bool with_deriv, Level& L) {
  L.w = w;
  L.h = h;
  L.pad = pad;
  const int W = w + 2 * pad, H = h + 2 * pad;
  L.I.resize((size_t)W * H);
  for (int y = 0; y < H; y++) {
    int sy = reflect101(y - pad, h);
    for (int x = 0; x < W; x++) L.I[(size_t)y * W + x] = img[(size_t)sy * w + reflect101(x - pad, w)];
  }
  if (!with_deriv) return;
  L.dI.assign((size_t)W * H * 2, 0);
  // calcSharrDeriv on the unpadded image
  std::vector<int> trow0(w + 2), trow1(w + 2);
  for (int y = 0; y < h; y++) {
    const uint8_t* srow0 = img + (size_t)(y > 0 ? y - 1 : h > 1 ? 1 : 0) * w;
    const uint8_t* srow1 = img + (size_t)y * w;
    const uint8_t* srow2 = img + (size_t)(y < h - 1 ? y + 1 : h > 1 ? h - 2 : 0) * w;
    int* t0 = trow0.data() + 1;
    int* t1 = trow1.data() + 1;
    for (int x = 0; x < w; x++) {
      t0[x] = (short)((srow0[x] + srow2[x]) * 3 + srow1[x] * 10);
      t1[x] = (short)(srow2[x] - srow0[x]);
    }
    int x0 = (w > 1 ? 1 : 0), x1 = (w > 1 ? w - 2 : 0);
    t0[-1] = t0[x0];
    t0[w] = t0[x1];
    t1[-1] = t1[x0];
    t1[w] = t1[x1];
    short* drow = L.dI.data() + (((size_t)(y + pad)) * W + pad) * 2;
    for (int x = 0; x < w; x++) {
      drow[x * 2] = (short)(t0[x + 1] - t0[x - 1]);
      drow[x * 2 + 1] = (short)((t1[x + 1] + t1[x - 1]) * 3 + t1[x] * 10);
    }
  }
}

inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

}  // namespace

int calcOpticalFlowPyrLK(const uint8_t* prev, const uint8_t* next, int w, int h, size_t stride,
                         const Point2f* prevPts, Point2f* nextPts, int npoints,
                         uint8_t* status, float* err, int win, int maxLevelIn, int maxIter,
                         double epsIn, bool use_initial_flow, double minEigThresholdD) {
  Pyramid ppyr, npyr;
  int maxLevel = buildPyramid(prev, w, h, stride, win, maxLevelIn, ppyr);
  maxLevel = std::min(maxLevel, buildPyramid(next, w, h, stride, win, maxLevel, npyr));
  // TermCriteria clamping (SparsePyrLKOpticalFlowImpl::calc)
  maxIter = std::min(std::max(maxIter, 0), 100);
  double epsilon = std::min(std::max(epsIn, 0.), 10.);
  epsilon *= epsilon;
  const float minEigThreshold = (float)minEigThresholdD;

  for (int i = 0; i < npoints; i++) {
    status[i] = 1;
    if (err) err[i] = 0;
  }
  if (npoints == 0) return maxLevel;

  const int W_BITS = 14, W_BITS1 = 14;
  const float FLT_SCALE = 1.f / (1 << 20);
  const float halfWin = (win - 1) * 0.5f;
  std::vector<short> IWinBuf((size_t)win * win), dIWinBuf((size_t)win * win * 2);

  for (int level = maxLevel; level >= 0; level--) {
    Level LI, LJ;
    make_level(ppyr.img[level].data(), ppyr.w[level], ppyr.h[level], win, true, LI);
    make_level(npyr.img[level].data(), npyr.w[level], npyr.h[level], win, false, LJ);
    const int cols = LI.w, rows = LI.h;
    const int stepI = LI.stepI(), stepJ = LJ.stepI(), dstep = LI.stepI() * 2;
    const uint8_t* Ibase = LI.iptr();
    const uint8_t* Jbase = LJ.iptr();
    const short* dbase = LI.dptr();

    for (int ptidx = 0; ptidx < npoints; ptidx++) {
      const float lscale = (float)(1. / (1 << level));
      Point2f prevPt = {prevPts[ptidx].x * lscale, prevPts[ptidx].y * lscale};
      Point2f nextPt;
      if (level == maxLevel) {
        if (use_initial_flow)
          nextPt = {nextPts[ptidx].x * lscale, nextPts[ptidx].y * lscale};
        else
          nextPt = prevPt;
      } else
        nextPt = {nextPts[ptidx].x * 2.f, nextPts[ptidx].y * 2.f};
      nextPts[ptidx] = nextPt;

      prevPt.x -= halfWin;
      prevPt.y -= halfWin;
      int ipx = cvFloorf(prevPt.x), ipy = cvFloorf(prevPt.y);
      if (ipx < -win || ipx >= cols || ipy < -win || ipy >= rows) {
        if (level == 0) {
          status[ptidx] = 0;
          if (err) err[ptidx] = 0;
        }
        continue;
      }
      float a = prevPt.x - ipx, b = prevPt.y - ipy;
      int iw00 = cvRoundf((1.f - a) * (1.f - b) * (1 << W_BITS));
      int iw01 = cvRoundf(a * (1.f - b) * (1 << W_BITS));
      int iw10 = cvRoundf((1.f - a) * b * (1 << W_BITS));
      int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;

      float iA11 = 0, iA12 = 0, iA22 = 0;
      float qA11[4] = {0, 0, 0, 0}, qA12[4] = {0, 0, 0, 0}, qA22[4] = {0, 0, 0, 0};
      for (int y = 0; y < win; y++) {
        const uint8_t* src = Ibase + (ptrdiff_t)(y + ipy) * stepI + ipx;
        const short* dsrc = dbase + (ptrdiff_t)(y + ipy) * dstep + ipx * 2;
        short* Iptr = &IWinBuf[(size_t)y * win];
        short* dIptr = &dIWinBuf[(size_t)y * win * 2];
        int x = 0;
        for (; x <= win - 4; x += 4) {  // SSE2 body: 4 pixels, lane l = pixel x+l
          for (int l = 0; l < 4; l++) {
            const uint8_t* s = src + x + l;
            const short* d = dsrc + (x + l) * 2;
            int ival = (s[0] * iw00 + s[1] * iw01 + s[stepI] * iw10 + s[stepI + 1] * iw11 +
                        (1 << (W_BITS1 - 5 - 1))) >> (W_BITS1 - 5);
            int ixval = (d[0] * iw00 + d[2] * iw01 + d[dstep] * iw10 + d[dstep + 2] * iw11 +
                         (1 << (W_BITS1 - 1))) >> W_BITS1;
            int iyval = (d[1] * iw00 + d[3] * iw01 + d[dstep + 1] * iw10 + d[dstep + 3] * iw11 +
                         (1 << (W_BITS1 - 1))) >> W_BITS1;
            Iptr[x + l] = sat_short(ival);
            short sx = sat_short(ixval), sy = sat_short(iyval);
            dIptr[(x + l) * 2] = sx;
            dIptr[(x + l) * 2 + 1] = sy;
            float fx = (float)sx, fy = (float)sy;
            qA22[l] = qA22[l] + fy * fy;
            qA12[l] = qA12[l] + fx * fy;
            qA11[l] = qA11[l] + fx * fx;
          }
        }
        for (; x < win; x++) {  // scalar tail
          const uint8_t* s = src + x;
          const short* d = dsrc + x * 2;
          int ival = descale(s[0] * iw00 + s[1] * iw01 + s[stepI] * iw10 + s[stepI + 1] * iw11,
                             W_BITS1 - 5);
          int ixval = descale(d[0] * iw00 + d[2] * iw01 + d[dstep] * iw10 + d[dstep + 2] * iw11,
                              W_BITS1);
          int iyval = descale(
              d[1] * iw00 + d[3] * iw01 + d[dstep + 1] * iw10 + d[dstep + 3] * iw11, W_BITS1);
          Iptr[x] = (short)ival;
          dIptr[x * 2] = (short)ixval;
          dIptr[x * 2 + 1] = (short)iyval;
          iA11 += (float)(ixval * ixval);
          iA12 += (float)(ixval * iyval);
          iA22 += (float)(iyval * iyval);
        }
      }
      iA11 += qA11[0] + qA11[1] + qA11[2] + qA11[3];
      iA12 += qA12[0] + qA12[1] + qA12[2] + qA12[3];
      iA22 += qA22[0] + qA22[1] + qA22[2] + qA22[3];

      float A11 = iA11 * FLT_SCALE, A12 = iA12 * FLT_SCALE, A22 = iA22 * FLT_SCALE;
      float D = A11 * A22 - A12 * A12;
      float minEig = (A22 + A11 - std::sqrt((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                     (2 * win * win);
      if (minEig < minEigThreshold || D < FLT_EPSILON) {
        if (level == 0) status[ptidx] = 0;
        continue;
      }
      D = 1.f / D;

      nextPt.x -= halfWin;
      nextPt.y -= halfWin;
      Point2f prevDelta = {0, 0};

      for (int j = 0; j < maxIter; j++) {
        int inx = cvFloorf(nextPt.x), iny = cvFloorf(nextPt.y);
        if (inx < -win || inx >= LJ.w || iny < -win || iny >= LJ.h) {
          if (level == 0) status[ptidx] = 0;
          break;
        }
        a = nextPt.x - inx;
        b = nextPt.y - iny;
        iw00 = cvRoundf((1.f - a) * (1.f - b) * (1 << W_BITS));
        iw01 = cvRoundf(a * (1.f - b) * (1 << W_BITS));
        iw10 = cvRoundf((1.f - a) * b * (1 << W_BITS));
        iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        float ib1 = 0, ib2 = 0;
        float qb0[4] = {0, 0, 0, 0}, qb1[4] = {0, 0, 0, 0};
        for (int y = 0; y < win; y++) {
          const uint8_t* Jptr = Jbase + (ptrdiff_t)(y + iny) * stepJ + inx;
          const short* Iptr = &IWinBuf[(size_t)y * win];
          const short* dIptr = &dIWinBuf[(size_t)y * win * 2];
          int x = 0;
          for (; x <= win - 8; x += 8) {  // SSE2 body: 8 pixels
            int diff[8];
            for (int l = 0; l < 8; l++) {
              const uint8_t* s = Jptr + x + l;
              int t = (s[0] * iw00 + s[1] * iw01 + s[stepJ] * iw10 + s[stepJ + 1] * iw11 +
                       (1 << (W_BITS1 - 5 - 1))) >> (W_BITS1 - 5);
              diff[l] = sat_short(sat_short(t) - Iptr[x + l]);  // _mm_subs_epi16
            }
            const short* dp = dIptr + x * 2;
            // madd pairs: (0,4) (1,5) -> qb0 ; (2,6) (3,7) -> qb1 ; lanes {Ix,Iy,Ix,Iy}
            int m00 = diff[0] * dp[0] + diff[4] * dp[8];
            int m01 = diff[0] * dp[1] + diff[4] * dp[9];
            int m02 = diff[1] * dp[2] + diff[5] * dp[10];
            int m03 = diff[1] * dp[3] + diff[5] * dp[11];
            int m10 = diff[2] * dp[4] + diff[6] * dp[12];
            int m11 = diff[2] * dp[5] + diff[6] * dp[13];
            int m12 = diff[3] * dp[6] + diff[7] * dp[14];
            int m13 = diff[3] * dp[7] + diff[7] * dp[15];
            qb0[0] = qb0[0] + (float)m00;
            qb0[1] = qb0[1] + (float)m01;
            qb0[2] = qb0[2] + (float)m02;
            qb0[3] = qb0[3] + (float)m03;
            qb1[0] = qb1[0] + (float)m10;
            qb1[1] = qb1[1] + (float)m11;
            qb1[2] = qb1[2] + (float)m12;
            qb1[3] = qb1[3] + (float)m13;
          }
          for (; x < win; x++) {
            const uint8_t* s = Jptr + x;
            int diff = descale(s[0] * iw00 + s[1] * iw01 + s[stepJ] * iw10 + s[stepJ + 1] * iw11,
                               W_BITS1 - 5) - Iptr[x];
            ib1 += (float)(diff * dIptr[x * 2]);
            ib2 += (float)(diff * dIptr[x * 2 + 1]);
          }
        }
        float bbuf[4];
        for (int l = 0; l < 4; l++) bbuf[l] = qb0[l] + qb1[l];
        ib1 += bbuf[0] + bbuf[2];
        ib2 += bbuf[1] + bbuf[3];

        float b1 = ib1 * FLT_SCALE, b2 = ib2 * FLT_SCALE;
        Point2f delta = {(float)((A12 * b2 - A22 * b1) * D), (float)((A12 * b1 - A11 * b2) * D)};
        nextPt.x += delta.x;
        nextPt.y += delta.y;
        nextPts[ptidx] = {nextPt.x + halfWin, nextPt.y + halfWin};
        if ((double)delta.x * delta.x + (double)delta.y * delta.y <= epsilon) break;
        if (j > 0 && std::abs(delta.x + prevDelta.x) < 0.01 &&
            std::abs(delta.y + prevDelta.y) < 0.01) {
          nextPts[ptidx].x -= delta.x * 0.5f;
          nextPts[ptidx].y -= delta.y * 0.5f;
          break;
        }
        prevDelta = delta;
      }

      if (status[ptidx] && err && level == 0) {
        Point2f nextPoint = {nextPts[ptidx].x - halfWin, nextPts[ptidx].y - halfWin};
        int inx = cvFloorf(nextPoint.x), iny = cvFloorf(nextPoint.y);
        if (inx < -win || inx >= LJ.w || iny < -win || iny >= LJ.h) {
          status[ptidx] = 0;
          continue;
        }
        float aa = nextPoint.x - inx, bb = nextPoint.y - iny;
        iw00 = cvRoundf((1.f - aa) * (1.f - bb) * (1 << W_BITS));
        iw01 = cvRoundf(aa * (1.f - bb) * (1 << W_BITS));
        iw10 = cvRoundf((1.f - aa) * bb * (1 << W_BITS));
        iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        float errval = 0.f;
        for (int y = 0; y < win; y++) {
          const uint8_t* Jptr = Jbase + (ptrdiff_t)(y + iny) * stepJ + inx;
          const short* Iptr = &IWinBuf[(size_t)y * win];
          for (int x = 0; x < win; x++) {
            const uint8_t* s = Jptr + x;
            int diff = descale(s[0] * iw00 + s[1] * iw01 + s[stepJ] * iw10 + s[stepJ + 1] * iw11,
                               W_BITS1 - 5) - Iptr[x];
            errval += std::abs((float)diff);
          }
        }
        err[ptidx] = errval * 1.f / (32 * win * win);
      }
    }
  }
  return maxLevel;
}

}  // namespace ocv
