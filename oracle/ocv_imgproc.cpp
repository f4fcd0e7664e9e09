// TEST INFRASTRUCTURE — see ocv.hpp.  Restatement of OpenCV 4.2 imgproc routines
// (imgwarp.cpp: remap/remapBilinear; corner.cpp: cornerEigenValsVecs,
// calcMinEigenVal; deriv.cpp+filter.simd.hpp: Sobel via RowFilter<uchar,float> +
// SymmColumnSmallFilter<float>; box_filter: RowSum<float,double> +
// ColumnSum<double,float>; featureselect.cpp: goodFeaturesToTrack; drawing.cpp:
// Circle; cornersubpix.cpp + samplers.cpp: cornerSubPix / getRectSubPix_8u32f;
// pyramids.cpp: pyrDown; templmatch.cpp: matchTemplate TM_SQDIFF).
#include <algorithm>
#include <cfloat>

#include "ocv.hpp"

namespace ocv {

// ---------------------------------------------------------------------------
// cv::remap, INTER_LINEAR, BORDER_REPLICATE, CV_8UC1, maps CV_32FC1 x2.
// INTER_BITS = 5, INTER_REMAP_COEF_BITS = 15.  The 32x32 bilinear weight table
// is w = {(32-ax)(32-ay), ax(32-ay), (32-ax)ay, ax*ay} * 32 (sums to 2^15; the
// (0,0) entry is {32767,0,0,1} after saturation+fix-up, which yields the same
// output pixel as {32768,0,0,0} for 8-bit data).
// ---------------------------------------------------------------------------
static inline int clipi(int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; }

void remap_linear_replicate(const uint8_t* src, int sw, int sh, size_t sstride, uint8_t* dst,
                            int dw, int dh, size_t dstride, const float* map_x,
                            const float* map_y) {
  const unsigned width1 = std::max(sw - 1, 0), height1 = std::max(sh - 1, 0);
  for (int y = 0; y < dh; y++) {
    const float* sX = map_x + (size_t)y * dw;
    const float* sY = map_y + (size_t)y * dw;
    uint8_t* D = dst + (size_t)y * dstride;
    for (int x = 0; x < dw; x++) {
      int sxf = cvRoundf(sX[x] * 32);
      int syf = cvRoundf(sY[x] * 32);
      int ax = sxf & 31, ay = syf & 31;
      int sx = sat_short(sxf >> 5), sy = sat_short(syf >> 5);
      int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32,
          w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
      if (ax == 0 && ay == 0) {  // saturate_cast<short>(32768) + isum fix-up
        w00 = 32767;
        w11 = 1;
      }
      int v0, v1, v2, v3;
      if ((unsigned)sx < width1 && (unsigned)sy < height1) {
        const uint8_t* S = src + (size_t)sy * sstride + sx;
        v0 = S[0];
        v1 = S[1];
        v2 = S[sstride];
        v3 = S[sstride + 1];
      } else {
        int sx0 = clipi(sx, 0, sw), sx1 = clipi(sx + 1, 0, sw);
        int sy0 = clipi(sy, 0, sh), sy1 = clipi(sy + 1, 0, sh);
        v0 = src[(size_t)sy0 * sstride + sx0];
        v1 = src[(size_t)sy0 * sstride + sx1];
        v2 = src[(size_t)sy1 * sstride + sx0];
        v3 = src[(size_t)sy1 * sstride + sx1];
      }
      D[x] = sat_u8((v0 * w00 + v1 * w01 + v2 * w10 + v3 * w11 + (1 << 14)) >> 15);
    }
  }
}

// ---------------------------------------------------------------------------
// cv::cornerMinEigenVal / cv::cornerHarris (ksize 3, BORDER_REFLECT_101), float32: corner.cpp cornerEigenValsVecs
// with op_type MINEIGENVAL / HARRIS.  Both share Sobel -> (dx*dx, dx*dy, dy*dy) -> unnormalised box filter; they
// differ in the last line (calcMinEigenVal / calcHarris).
// calcHarris (corner.cpp): the covariance image is continuous, so OpenCV folds it into ONE row of w*h pixels and runs
// its 4-wide float lanes over it -- v_a*v_c - v_b*v_b - (float)k*(v_a+v_c)*(v_a+v_c), every operation in float -- and
// the (w*h) % 4 pixels at the very end of the image go through the scalar tail, whose k is a double:
// (float)(a*c - b*b - k*(a + c)*(a + c)).  (An AVX build takes 8 lanes first; the tail is the same w*h % 4 pixels.)
// ---------------------------------------------------------------------------
static void cornerEigenVals(const uint8_t* src, int w, int h, size_t stride, int block_size, bool harris, double k,
                            float* eig) {
  double scale = (double)(1 << 2) * block_size;  // aperture 3
  scale *= 255.0;
  scale = 1.0 / scale;
  const float f1 = 1.0f * (float)scale + 0.0f;  // kernel [1 2 1] *= scale (cvt in float)
  const float f0 = 2.0f * (float)scale + 0.0f;

  // Row pass (RowFilter<uchar,float,RowNoVec>): accumulate taps left to right.
  //  Rdx = [-1 0 1] row-derivative, Tdy = [f1 f0 f1] row-smoothing.
  std::vector<float> Rdx((size_t)w * h), Tdy((size_t)w * h);
  for (int y = 0; y < h; y++) {
    const uint8_t* S = src + (size_t)y * stride;
    for (int x = 0; x < w; x++) {
      int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
      float s0 = -1.0f * S[xm];
      s0 += 0.0f * S[x];
      s0 += 1.0f * S[xp];
      Rdx[(size_t)y * w + x] = s0;
      float t0 = f1 * S[xm];
      t0 += f0 * S[x];
      t0 += f1 * S[xp];
      Tdy[(size_t)y * w + x] = t0;
    }
  }
  // Column pass (SymmColumnSmallFilter<Cast<float,float>>), delta = 0.
  std::vector<float> cov((size_t)w * h * 3);
  for (int y = 0; y < h; y++) {
    int ym = reflect101(y - 1, h), yp = reflect101(y + 1, h);
    const float* R0 = &Rdx[(size_t)ym * w];
    const float* R1 = &Rdx[(size_t)y * w];
    const float* R2 = &Rdx[(size_t)yp * w];
    const float* T0 = &Tdy[(size_t)ym * w];
    const float* T2 = &Tdy[(size_t)yp * w];
    float* C = &cov[(size_t)y * w * 3];
    for (int x = 0; x < w; x++) {
      float dx = (R0[x] + R2[x]) * f1 + R1[x] * f0 + 0.0f;
      float dy = T2[x] - T0[x] + 0.0f;
      C[x * 3] = dx * dx;
      C[x * 3 + 1] = dx * dy;
      C[x * 3 + 2] = dy * dy;
    }
  }
  // boxFilter(cov, block x block, normalize=false, REFLECT_101): double sums.
  const int anchor = block_size / 2;
  std::vector<double> rs((size_t)w * h * 3);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++)
      for (int c = 0; c < 3; c++) {
        double s = 0;
        for (int k = 0; k < block_size; k++) {
          int xx = reflect101(x - anchor + k, w);
          s += (double)cov[((size_t)y * w + xx) * 3 + c];
        }
        rs[((size_t)y * w + x) * 3 + c] = s;
      }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      float cv3[3];
      for (int c = 0; c < 3; c++) {
        double s = 0;
        for (int k = 0; k < block_size; k++) {
          int yy = reflect101(y - anchor + k, h);
          s += rs[((size_t)yy * w + x) * 3 + c];
        }
        cv3[c] = (float)s;
      }
      if (!harris) {
        float a = cv3[0] * 0.5f, b = cv3[1], c = cv3[2] * 0.5f;
        eig[(size_t)y * w + x] = (float)((a + c) - std::sqrt((a - c) * (a - c) + b * b));
      } else {
        const float a = cv3[0], b = cv3[1], c = cv3[2];
        const size_t j = (size_t)y * w + x, n = (size_t)w * h;
        if (j < n - n % 4) {
          const float kf = (float)k;
          const float ac_bb = a * c - b * b;
          const float ac = a + c;
          eig[j] = ac_bb - kf * ac * ac;
        } else {
          eig[j] = (float)(a * c - b * b - k * (a + c) * (a + c));
        }
      }
    }
}
void cornerMinEigenVal(const uint8_t* src, int w, int h, size_t stride, int block_size, float* eig) {
  cornerEigenVals(src, w, h, stride, block_size, false, 0.0, eig);
}
void cornerHarris(const uint8_t* src, int w, int h, size_t stride, int block_size, double k, float* dst) {
  cornerEigenVals(src, w, h, stride, block_size, true, k, dst);
}

// ---------------------------------------------------------------------------
// cv::goodFeaturesToTrack
// ---------------------------------------------------------------------------
void goodFeaturesToTrack(const uint8_t* img, int w, int h, size_t stride, const uint8_t* mask,
                         size_t mask_stride, int maxCorners, double qualityLevel,
                         double minDistance, int blockSize, std::vector<Point2f>& corners,
                         std::vector<float>* quality, bool useHarrisDetector, double harrisK) {
  corners.clear();
  if (quality) quality->clear();
  std::vector<float> eig((size_t)w * h);
  if (useHarrisDetector)   // featureselect.cpp: cornerHarris(image, eig, blockSize, gradientSize, harrisK)
    cornerHarris(img, w, h, stride, blockSize, harrisK, eig.data());
  else
    cornerMinEigenVal(img, w, h, stride, blockSize, eig.data());

  // minMaxLoc(eig, 0, &maxVal, 0, 0, mask)
  double maxVal = 0;
  bool found = false;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      if (mask && !mask[(size_t)y * mask_stride + x]) continue;
      float v = eig[(size_t)y * w + x];
      if (!found || v > maxVal) {
        maxVal = v;
        found = true;
      }
    }
  // threshold(eig, eig, maxVal*qualityLevel, 0, THRESH_TOZERO)
  const float thr = (float)(maxVal * qualityLevel);
  for (size_t i = 0; i < eig.size(); i++) eig[i] = eig[i] > thr ? eig[i] : 0.f;

  // dilate 3x3 + collect local maxima in the interior
  std::vector<const float*> tmpCorners;
  for (int y = 1; y < h - 1; y++) {
    const float* e = &eig[(size_t)y * w];
    for (int x = 1; x < w - 1; x++) {
      float val = e[x];
      if (val == 0) continue;
      if (mask && !mask[(size_t)y * mask_stride + x]) continue;
      float m = val;
      for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) m = std::max(m, e[dy * w + x + dx]);
      if (val == m) tmpCorners.push_back(e + x);
    }
  }
  if (tmpCorners.empty()) return;
  // greaterThanPtr: value descending, ties -> higher address first
  std::sort(tmpCorners.begin(), tmpCorners.end(), [](const float* a, const float* b) {
    return (*a > *b) ? true : (*a < *b) ? false : (a > b);
  });

  size_t total = tmpCorners.size(), ncorners = 0;
  if (minDistance >= 1) {
    const int cell_size = cvRound(minDistance);
    const int grid_width = (w + cell_size - 1) / cell_size;
    const int grid_height = (h + cell_size - 1) / cell_size;
    std::vector<std::vector<Point2f>> grid((size_t)grid_width * grid_height);
    minDistance *= minDistance;
    for (size_t i = 0; i < total; i++) {
      int ofs = (int)(tmpCorners[i] - eig.data());
      int y = ofs / w, x = ofs - y * w;
      bool good = true;
      int x_cell = x / cell_size, y_cell = y / cell_size;
      int x1 = std::max(0, x_cell - 1), y1 = std::max(0, y_cell - 1);
      int x2 = std::min(grid_width - 1, x_cell + 1), y2 = std::min(grid_height - 1, y_cell + 1);
      for (int yy = y1; yy <= y2 && good; yy++)
        for (int xx = x1; xx <= x2 && good; xx++) {
          const std::vector<Point2f>& m = grid[(size_t)yy * grid_width + xx];
          for (size_t j = 0; j < m.size(); j++) {
            float dx = x - m[j].x, dy = y - m[j].y;
            if (dx * dx + dy * dy < minDistance) {
              good = false;
              break;
            }
          }
        }
      if (good) {
        grid[(size_t)y_cell * grid_width + x_cell].push_back(Point2f{(float)x, (float)y});
        corners.push_back(Point2f{(float)x, (float)y});
        if (quality) quality->push_back(*tmpCorners[i]);
        ++ncorners;
        if (maxCorners > 0 && (int)ncorners == maxCorners) break;
      }
    }
  } else {
    for (size_t i = 0; i < total; i++) {
      int ofs = (int)(tmpCorners[i] - eig.data());
      int y = ofs / w, x = ofs - y * w;
      corners.push_back(Point2f{(float)x, (float)y});
      if (quality) quality->push_back(*tmpCorners[i]);
      ++ncorners;
      if (maxCorners > 0 && (int)ncorners == maxCorners) break;
    }
  }
}

// ---------------------------------------------------------------------------
// cv::FAST(img, keypoints, threshold, nonmaxSuppression, TYPE_9_16) -- features2d/src/fast.cpp FAST_t<16> and
// fast_score.cpp cornerScore<16> (the scalar statements; the SIMD paths compute the same integers).
// A pixel is a corner when 9 contiguous pixels of the 16-pixel Bresenham circle of radius 3 are all darker than
// v - t or all brighter than v + t; rows 3 .. h-4, columns 3 .. w-4.  Its score is the largest threshold for which it
// still is one, minus 1 -- max over the 16 arcs of the arc's smallest one-sided difference -- stored as a byte; with
// non-maximum suppression a corner survives when its score is strictly greater than its 8 neighbours' (0 for
// non-corners).  Keypoints leave in raster order (row, then column), response = score.
// ---------------------------------------------------------------------------
static const int kFastOffsets16[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                                          {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

static int fastCornerScore16(const uint8_t* ptr, const int pixel[25], int threshold) {
  const int K = 8, N = K * 3 + 1;
  int k, v = ptr[0];
  short d[N];
  for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
  int a0 = threshold;
  for (k = 0; k < 16; k += 2) {
    int a = std::min((int)d[k + 1], (int)d[k + 2]);
    a = std::min(a, (int)d[k + 3]);
    if (a <= a0) continue;
    a = std::min(a, (int)d[k + 4]);
    a = std::min(a, (int)d[k + 5]);
    a = std::min(a, (int)d[k + 6]);
    a = std::min(a, (int)d[k + 7]);
    a = std::min(a, (int)d[k + 8]);
    a0 = std::max(a0, std::min(a, (int)d[k]));
    a0 = std::max(a0, std::min(a, (int)d[k + 9]));
  }
  int b0 = -a0;
  for (k = 0; k < 16; k += 2) {
    int b = std::max((int)d[k + 1], (int)d[k + 2]);
    b = std::max(b, (int)d[k + 3]);
    b = std::max(b, (int)d[k + 4]);
    b = std::max(b, (int)d[k + 5]);
    if (b >= b0) continue;
    b = std::max(b, (int)d[k + 6]);
    b = std::max(b, (int)d[k + 7]);
    b = std::max(b, (int)d[k + 8]);
    b0 = std::min(b0, std::max(b, (int)d[k]));
    b0 = std::min(b0, std::max(b, (int)d[k + 9]));
  }
  threshold = -b0 - 1;
  return threshold;
}

void FAST_9_16(const uint8_t* img, int w, int h, size_t stride, int threshold, bool nonmax_suppression,
               std::vector<FastKeyPoint>& keypoints) {
  const int K = 8, N = 16 + K + 1;
  int i, j, k, pixel[25];
  for (k = 0; k < 16; k++) pixel[k] = kFastOffsets16[k][0] + kFastOffsets16[k][1] * (int)stride;
  for (; k < 25; k++) pixel[k] = pixel[k - 16];
  keypoints.clear();
  threshold = std::min(std::max(threshold, 0), 255);
  uint8_t threshold_tab[512];
  for (i = -255; i <= 255; i++) threshold_tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
  std::vector<uint8_t> bufv((size_t)w * 3, 0);
  std::vector<int> cpv((size_t)(w + 1) * 3, 0);
  uint8_t* buf[3] = {bufv.data(), bufv.data() + w, bufv.data() + 2 * w};
  int* cpbuf[3] = {cpv.data(), cpv.data() + (w + 1), cpv.data() + 2 * (w + 1)};
  for (i = 3; i < h - 2; i++) {
    const uint8_t* ptr = img + (size_t)i * stride + 3;
    uint8_t* curr = buf[(i - 3) % 3];
    int* cornerpos = cpbuf[(i - 3) % 3] + 1;
    std::fill(curr, curr + w, (uint8_t)0);
    int ncorners = 0;
    if (i < h - 3) {
      for (j = 3; j < w - 3; j++, ptr++) {
        int v = ptr[0];
        const uint8_t* tab = &threshold_tab[0] - v + 255;
        int d = tab[ptr[pixel[0]]] | tab[ptr[pixel[8]]];
        if (d == 0) continue;
        d &= tab[ptr[pixel[2]]] | tab[ptr[pixel[10]]];
        d &= tab[ptr[pixel[4]]] | tab[ptr[pixel[12]]];
        d &= tab[ptr[pixel[6]]] | tab[ptr[pixel[14]]];
        if (d == 0) continue;
        d &= tab[ptr[pixel[1]]] | tab[ptr[pixel[9]]];
        d &= tab[ptr[pixel[3]]] | tab[ptr[pixel[11]]];
        d &= tab[ptr[pixel[5]]] | tab[ptr[pixel[13]]];
        d &= tab[ptr[pixel[7]]] | tab[ptr[pixel[15]]];
        if (d & 1) {
          int vt = v - threshold, count = 0;
          for (k = 0; k < N; k++) {
            int x = ptr[pixel[k]];
            if (x < vt) {
              if (++count > K) {
                cornerpos[ncorners++] = j;
                if (nonmax_suppression) curr[j] = (uint8_t)fastCornerScore16(ptr, pixel, threshold);
                break;
              }
            } else
              count = 0;
          }
        }
        if (d & 2) {
          int vt = v + threshold, count = 0;
          for (k = 0; k < N; k++) {
            int x = ptr[pixel[k]];
            if (x > vt) {
              if (++count > K) {
                cornerpos[ncorners++] = j;
                if (nonmax_suppression) curr[j] = (uint8_t)fastCornerScore16(ptr, pixel, threshold);
                break;
              }
            } else
              count = 0;
          }
        }
      }
    }
    cornerpos[-1] = ncorners;
    if (i == 3) continue;
    const uint8_t* prev = buf[(i - 4 + 3) % 3];
    const uint8_t* pprev = buf[(i - 5 + 3) % 3];
    cornerpos = cpbuf[(i - 4 + 3) % 3] + 1;
    ncorners = cornerpos[-1];
    for (k = 0; k < ncorners; k++) {
      j = cornerpos[k];
      int score = prev[j];
      if (!nonmax_suppression ||
          (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
           score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1])) {
        keypoints.push_back(FastKeyPoint{(float)j, (float)(i - 1), (float)score});
      }
    }
  }
}

// cv::FastFeatureDetector::detect(image, keypoints, mask): FAST + KeyPointsFilter::runByPixelsMask
void fastDetect(const uint8_t* img, int w, int h, size_t stride, const uint8_t* mask, size_t mask_stride,
                int threshold, bool nonmax, std::vector<FastKeyPoint>& keypoints) {
  if (w <= 0 || h <= 0) {
    keypoints.clear();
    return;
  }
  FAST_9_16(img, w, h, stride, threshold, nonmax, keypoints);
  if (!mask) return;
  std::vector<FastKeyPoint> kept;
  for (const FastKeyPoint& kp : keypoints)   // MaskPredicate: mask.at<uchar>((int)(pt.y + 0.5f), (int)(pt.x + 0.5f)) == 0
    if (mask[(size_t)(int)(kp.y + 0.5f) * mask_stride + (int)(kp.x + 0.5f)] != 0) kept.push_back(kp);
  keypoints.swap(kept);
}

// ---------------------------------------------------------------------------
// cv::circle(..., FILLED) -> drawing.cpp Circle(img, center, radius, color, fill=1)
// ---------------------------------------------------------------------------
static inline void hline(uint8_t* row, int x1, int x2, uint8_t color) {
  for (int x = x1; x <= x2; x++) row[x] = color;
}

void circle_filled(uint8_t* img, int w, int h, size_t stride, int cx, int cy, int radius,
                   uint8_t color) {
  int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
  const int inside = cx >= radius && cx < w - radius && cy >= radius && cy < h - radius;
  while (dx >= dy) {
    int mask;
    int y11 = cy - dy, y12 = cy + dy, y21 = cy - dx, y22 = cy + dx;
    int x11 = cx - dx, x12 = cx + dx, x21 = cx - dy, x22 = cx + dy;
    if (inside) {
      hline(img + (size_t)y11 * stride, x11, x12, color);
      hline(img + (size_t)y12 * stride, x11, x12, color);
      hline(img + (size_t)y21 * stride, x21, x22, color);
      hline(img + (size_t)y22 * stride, x21, x22, color);
    } else if (x11 < w && x12 >= 0 && y21 < h && y22 >= 0) {
      x11 = std::max(x11, 0);
      x12 = std::min(x12, w - 1);
      if ((unsigned)y11 < (unsigned)h) hline(img + (size_t)y11 * stride, x11, x12, color);
      if ((unsigned)y12 < (unsigned)h) hline(img + (size_t)y12 * stride, x11, x12, color);
      if (x21 < w && x22 >= 0) {
        x21 = std::max(x21, 0);
        x22 = std::min(x22, w - 1);
        if ((unsigned)y21 < (unsigned)h) hline(img + (size_t)y21 * stride, x21, x22, color);
        if ((unsigned)y22 < (unsigned)h) hline(img + (size_t)y22 * stride, x21, x22, color);
      }
    }
    dy++;
    err += plus;
    plus += 2;
    mask = (err <= 0) - 1;
    err -= minus & mask;
    dx += mask;
    minus -= mask & 2;
  }
}

// ---------------------------------------------------------------------------
// cv::getRectSubPix(src u8 -> f32) : getRectSubPix_8u32f / getRectSubPix_Cn_
// ---------------------------------------------------------------------------
static void getRectSubPix_8u32f(const uint8_t* src, size_t step, int sw, int sh, float* dst,
                                int win_w, int win_h, Point2f center0) {
  Point2f center = center0;
  center.x -= (win_w - 1) * 0.5f;
  center.y -= (win_h - 1) * 0.5f;
  int ipx = cvFloorf(center.x), ipy = cvFloorf(center.y);
  if (0 <= ipx && ipx + win_w < sw && 0 <= ipy && ipy + win_h < sh && win_w > 0 && win_h > 0) {
    float a = center.x - ipx;
    float b = center.y - ipy;
    a = std::max(a, 0.0001f);
    float a12 = a * (1.f - b);
    float a22 = a * b;
    float b1 = 1.f - b;
    float b2 = b;
    double s = (1. - a) / a;
    const uint8_t* S = src + (size_t)ipy * step + ipx;
    for (int i = 0; i < win_h; i++, S += step, dst += win_w) {
      float prev = (1 - a) * (b1 * S[0] + b2 * S[step]);
      for (int j = 0; j < win_w; j++) {
        float t = a12 * S[j + 1] + a22 * S[j + 1 + step];
        dst[j] = prev + t;
        prev = (float)(t * s);
      }
    }
    return;
  }
  // generic path with replicated border (getRectSubPix_Cn_<uchar,float,float>)
  float a = center.x - ipx, b = center.y - ipy;
  float a11 = (1.f - a) * (1.f - b), a12 = a * (1.f - b), a21 = (1.f - a) * b, a22 = a * b;
  float b1 = 1.f - b, b2 = b;
  // adjustRect
  int rx, rw, ry, rh;
  if (ipx >= 0)
    rx = 0;
  else {
    rx = -ipx;
    if (rx > win_w) rx = win_w;
  }
  if (ipx < sw - win_w)
    rw = win_w;
  else {
    rw = sw - ipx - 1;
    if (rw < 0) rw = 0;
  }
  if (ipy >= 0)
    ry = 0;
  else
    ry = -ipy;
  if (ipy < sh - win_h)
    rh = win_h;
  else {
    rh = sh - ipy - 1;
    if (rh < 0) rh = 0;
  }
  // image row of the running `src` pointer
  int row = ipy >= 0 ? ipy : 0;
  if (ipy >= sh - win_h && sh - ipy - 1 < 0) row += sh - ipy - 1;
  auto col = [&](int j) {  // image column addressed by src[j]
    int c = ipx + j;
    return c < 0 ? 0 : (c > sw - 1 ? sw - 1 : c);
  };
  for (int i = 0; i < win_h; i++, dst += win_w) {
    int row2 = row + 1;
    if (i < ry || i >= rh) row2 = row;
    int r0 = std::min(std::max(row, 0), sh - 1), r1 = std::min(std::max(row2, 0), sh - 1);
    const uint8_t* S = src + (size_t)r0 * step;
    const uint8_t* S2 = src + (size_t)r1 * step;
    float s0 = S[col(rx)] * b1 + S2[col(rx)] * b2;
    for (int j = 0; j < rx; j++) dst[j] = s0;
    s0 = S[col(rw)] * b1 + S2[col(rw)] * b2;
    for (int j = rw; j < win_w; j++) dst[j] = s0;
    for (int j = rx; j < rw; j++) {
      float v = S[col(j)] * a11 + S[col(j + 1)] * a12 + S2[col(j)] * a21 + S2[col(j + 1)] * a22;
      dst[j] = v;
    }
    if (i < rh) row = row2;
  }
}

void cornerSubPix(const uint8_t* img, int w, int h, size_t stride, Point2f* corners, int count,
                  int win, int zero_zone, int max_iters_in, double eps_in) {
  const int MAX_ITERS = 100;
  const int win_w = win * 2 + 1, win_h = win * 2 + 1;
  const int max_iters = std::min(std::max(max_iters_in, 1), MAX_ITERS);
  double eps = std::max(eps_in, 0.);
  eps *= eps;
  if (count == 0) return;
  std::vector<float> mask((size_t)win_w * win_h), subpix_buf((size_t)(win_w + 2) * (win_h + 2));
  for (int i = 0; i < win_h; i++) {
    float y = (float)(i - win) / win;
    float vy = std::exp(-y * y);
    for (int j = 0; j < win_w; j++) {
      float x = (float)(j - win) / win;
      mask[i * win_w + j] = (float)(vy * std::exp(-x * x));
    }
  }
  if (zero_zone >= 0 && zero_zone * 2 + 1 < win_w && zero_zone * 2 + 1 < win_h) {
    for (int i = win - zero_zone; i <= win + zero_zone; i++)
      for (int j = win - zero_zone; j <= win + zero_zone; j++) mask[i * win_w + j] = 0;
  }
  for (int pt_i = 0; pt_i < count; pt_i++) {
    Point2f cT = corners[pt_i], cI = cT;
    int iter = 0;
    double err = 0;
    do {
      Point2f cI2;
      double a = 0, b = 0, c = 0, bb1 = 0, bb2 = 0;
      getRectSubPix_8u32f(img, stride, w, h, subpix_buf.data(), win_w + 2, win_h + 2, cI);
      const float* subpix = &subpix_buf[(win_w + 2) + 1];
      for (int i = 0, k = 0; i < win_h; i++, subpix += win_w + 2) {
        double py = i - win;
        for (int j = 0; j < win_w; j++, k++) {
          double m = mask[k];
          double tgx = subpix[j + 1] - subpix[j - 1];
          double tgy = subpix[j + win_w + 2] - subpix[j - win_w - 2];
          double gxx = tgx * tgx * m;
          double gxy = tgx * tgy * m;
          double gyy = tgy * tgy * m;
          double px = j - win;
          a += gxx;
          b += gxy;
          c += gyy;
          bb1 += gxx * px + gxy * py;
          bb2 += gxy * px + gyy * py;
        }
      }
      double det = a * c - b * b;
      if (std::fabs(det) <= DBL_EPSILON * DBL_EPSILON) break;
      double scale = 1.0 / det;
      cI2.x = (float)(cI.x + c * scale * bb1 - b * scale * bb2);
      cI2.y = (float)(cI.y - b * scale * bb1 + a * scale * bb2);
      err = (cI2.x - cI.x) * (cI2.x - cI.x) + (cI2.y - cI.y) * (cI2.y - cI.y);
      cI = cI2;
      if (cI.x < 0 || cI.x >= w || cI.y < 0 || cI.y >= h) break;
    } while (++iter < max_iters && err > eps);
    if (std::fabs(cI.x - cT.x) > win || std::fabs(cI.y - cT.y) > win) cI = cT;
    corners[pt_i] = cI;
  }
}

// ---------------------------------------------------------------------------
// cv::equalizeHist (imgproc/src/histogram.cpp): 256-bin histogram, LUT = saturate_cast<uchar>(
// cumulative count above the first occupied bin * (255.f / (total - hist[first]))), float arithmetic
void equalizeHist(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, size_t dstride) {
  int hist[256] = {0}, lut[256] = {0};
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) hist[src[(size_t)y * sstride + x]]++;
  int i = 0;
  while (!hist[i]) ++i;
  const int total = w * h;
  if (hist[i] == total) {
    for (int y = 0; y < h; y++) std::memset(dst + (size_t)y * dstride, i, w);
    return;
  }
  const float scale = (256 - 1.f) / (total - hist[i]);
  int sum = 0;
  for (lut[i++] = 0; i < 256; ++i) {
    sum += hist[i];
    int v = cvRoundf(sum * scale);
    lut[i] = v < 0 ? 0 : (v > 255 ? 255 : v);
  }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) dst[(size_t)y * dstride + x] = (uint8_t)lut[src[(size_t)y * sstride + x]];
}

// cv::pyrDown (u8, 5x5 [1 4 6 4 1]/16, REFLECT_101, (sum+128)>>8)
// ---------------------------------------------------------------------------
void pyrDown(const uint8_t* src, int sw, int sh, size_t sstride, uint8_t* dst, int dw, int dh,
             size_t dstride) {
  std::vector<int> rowbuf((size_t)5 * dw);
  for (int y = 0; y < dh; y++) {
    for (int k = 0; k < 5; k++) {
      int sy = reflect101(y * 2 - 2 + k, sh);
      const uint8_t* S = src + (size_t)sy * sstride;
      int* row = &rowbuf[(size_t)k * dw];
      for (int x = 0; x < dw; x++) {
        int x0 = reflect101(x * 2 - 2, sw), x1 = reflect101(x * 2 - 1, sw),
            x2 = reflect101(x * 2, sw), x3 = reflect101(x * 2 + 1, sw),
            x4 = reflect101(x * 2 + 2, sw);
        row[x] = S[x2] * 6 + (S[x1] + S[x3]) * 4 + S[x0] + S[x4];
      }
    }
    const int *r0 = &rowbuf[0], *r1 = &rowbuf[dw], *r2 = &rowbuf[2 * (size_t)dw],
              *r3 = &rowbuf[3 * (size_t)dw], *r4 = &rowbuf[4 * (size_t)dw];
    uint8_t* D = dst + (size_t)y * dstride;
    for (int x = 0; x < dw; x++)
      D[x] = (uint8_t)((r2[x] * 6 + (r1[x] + r3[x]) * 4 + r0[x] + r4[x] + 128) >> 8);
  }
}

// ---------------------------------------------------------------------------
// cv::matchTemplate TM_SQDIFF (exact integer value)
// ---------------------------------------------------------------------------
void matchTemplateSqdiff(const uint8_t* img, int iw, int ih, size_t istride,
                         const uint8_t* templ, int tw, int th, size_t tstride,
                         std::vector<int64_t>& result) {
  const int rw = iw - tw + 1, rh = ih - th + 1;
  result.assign((size_t)std::max(rw, 0) * std::max(rh, 0), 0);
  for (int y = 0; y < rh; y++)
    for (int x = 0; x < rw; x++) {
      int64_t s = 0;
      for (int j = 0; j < th; j++) {
        const uint8_t* I = img + (size_t)(y + j) * istride + x;
        const uint8_t* T = templ + (size_t)j * tstride;
        int acc = 0;
        for (int i = 0; i < tw; i++) {
          int d = (int)I[i] - (int)T[i];
          acc += d * d;
        }
        s += acc;
      }
      result[(size_t)y * rw + x] = s;
    }
}

}  // namespace ocv
