// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
// CPU restatement of the OpenCV 4.2 routines Kimera-VIO's stereo front-end
// calls (OpenCV is an un-vendored, unpinned dependency of the reference:
// CMakeLists.txt:31; 4.2.0 in Dockerfile_20_04:1,23-24).  OpenCV's sources are
// NOT in /root/reference and no OpenCV build exists in this container, so each
// routine restates the published algorithm of the generic (non-IPP, SSE2
// baseline, no FMA) C++ path; the reference call site that fixes the arguments
// is cited next to each declaration.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may use anything in oracle/.
//
// PARITY PINNING: pinned against the reference's own known-answer tests where
// they exist (tests/test_oracle_kat.py: detector counts 393/400/300/20/200/140,
// chessboard 63 corners, stereo baseline 0.110078, 849/900 shifted-image stereo
// matches, synthetic-pair corners/depth).  cv::calcOpticalFlowPyrLK,
// cv::cornerSubPix, cv::remap pixel values and the matchTemplate score have NO
// numeric pin in the reference's tests ("parity unpinned").
#pragma once
#include <stddef.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace ocv {

// cvRound: round-half-to-even (SSE2 cvtsd2si / cvtss2si).
static inline int cvRound(double v) { return (int)std::lrint(v); }
static inline int cvRoundf(float v) { return (int)std::lrintf(v); }
static inline int cvFloor(double v) {
  int i = (int)v;
  return i - (i > v);
}
static inline int cvFloorf(float v) {
  int i = (int)v;
  return i - ((float)i > v);
}
static inline int cvCeil(double v) {
  int i = (int)v;
  return i + (i < v);
}
static inline short sat_short(int v) {
  return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v);
}
static inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

// cv::borderInterpolate(p, len, BORDER_REFLECT_101)
static inline int reflect101(int p, int len) {
  if ((unsigned)p < (unsigned)len) return p;
  if (len == 1) return 0;
  do {
    if (p < 0)
      p = -p;
    else
      p = 2 * len - 2 - p;
  } while ((unsigned)p >= (unsigned)len);
  return p;
}

struct Point2f {
  float x, y;
};

// ---- calib3d ------------------------------------------------------------
// cv::Rodrigues both directions (matrix->vector without the SVD
// re-orthonormalisation step; inputs here are rotations to ~1e-12).
void rodrigues_vec_to_mat(const double r[3], double R[9]);
void rodrigues_mat_to_vec(const double R[9], double r[3]);

// cvUndistortPointsInternal: cv::undistortPoints(src, dst, K, D, R, P) with the
// default criteria (COUNT, 5).  K row-major 3x3; D = k1 k2 p1 p2 [k3 k4 k5 k6];
// R may be null (identity); P (3x4 row-major) may be null.
// Reference call sites: UndistorterRectifier.cpp:42-47.
void undistortPoints(const Point2f* src, Point2f* dst, int n, const double K[9],
                     const double* D, int nD, const double* R /*9 or null*/,
                     const double* P /*12 or null*/);

// cv::stereoRectify(K1,D1,K2,D2,size,R,T,R1,R2,P1,P2,Q,flags=ZERO_DISPARITY,
// alpha,newImageSize=(0,0),roi1,roi2).  Reference: StereoCamera.cpp:329-348.
void stereoRectify(const double K1[9], const double* D1, int nD1, const double K2[9],
                   const double* D2, int nD2, int width, int height, const double R[9],
                   const double T[3], double alpha, bool zero_disparity, double R1[9],
                   double R2[9], double P1[12], double P2[12], double Q[16], int roi1[4],
                   int roi2[4]);

// cv::fisheye (calib3d/src/fisheye.cpp, OpenCV 4.2): the equidistant model the reference selects for
// distortion_model: equidistant (UndistorterRectifier.cpp:49-57,260-272, StereoCamera.cpp:350-366).
// D: 4 coefficients.  The reference's own tests of this model are DISABLED_ (tests/
// testUndistortRectifier.cpp:222-330): parity unpinned.
namespace fisheye {
void undistortPoints(const Point2f* src, Point2f* dst, int n, const double K[9], const double D[4],
                     const double* R /*9|null*/, const double* P /*12|null*/);
void undistortPointsD(const double* src_xy, double* dst_xy, int n, const double K[9], const double D[4],
                      const double* R);  // CV_64FC2 input (estimateNewCameraMatrixForUndistortRectify)
void estimateNewCameraMatrixForUndistortRectify(const double K[9], const double D[4], int w, int h,
                                                const double R[9], double newK[9]);
void stereoRectify(const double K1[9], const double D1[4], const double K2[9], const double D2[4], int w,
                   int h, const double R[9], const double T[3], double R1[9], double R2[9],
                   double P1[12], double P2[12], double Q[16]);  // CALIB_ZERO_DISPARITY, balance 0
void initUndistortRectifyMap(const double K[9], const double D[4], const double R[9], const double P[12],
                             int w, int h, float* map_x, float* map_y);
}  // namespace fisheye

// cv::initUndistortRectifyMap(K, D, R, P, size, CV_32FC1, map1, map2).
// Reference: UndistorterRectifier.cpp:248-258.
void initUndistortRectifyMap(const double K[9], const double* D, int nD, const double R[9],
                             const double P[12], int width, int height, float* map_x,
                             float* map_y);

// ---- imgproc --------------------------------------------------------------
// cv::remap(src u8, dst, mapx f32, mapy f32, INTER_LINEAR, BORDER_REPLICATE).
// Reference: UndistorterRectifier.cpp:121-127.
void remap_linear_replicate(const uint8_t* src, int sw, int sh, size_t sstride, uint8_t* dst,
                            int dw, int dh, size_t dstride, const float* map_x,
                            const float* map_y);

// cv::cornerMinEigenVal(src u8, dst f32, blockSize, ksize=3, BORDER_DEFAULT).
void cornerMinEigenVal(const uint8_t* src, int w, int h, size_t stride, int block_size,
                       float* eig);
// cv::cornerHarris(src u8, dst f32, blockSize, ksize=3, k, BORDER_DEFAULT) (FeatureDetector.cpp:78-79:
// use_harris_corner_detector_, k_)
void cornerHarris(const uint8_t* src, int w, int h, size_t stride, int block_size, double k, float* dst);

// cv::goodFeaturesToTrack(img, corners, maxCorners, quality, minDistance, mask,
// blockSize, gradientSize=3, useHarrisDetector, k) as reached through
// cv::GFTTDetector::detect.  Reference: FeatureDetector.cpp:73-80,170.
// mask may be null.  Output: integer-valued corners, quality-descending.
void goodFeaturesToTrack(const uint8_t* img, int w, int h, size_t stride, const uint8_t* mask,
                         size_t mask_stride, int maxCorners, double qualityLevel,
                         double minDistance, int blockSize, std::vector<Point2f>& corners,
                         std::vector<float>* quality = nullptr, bool useHarrisDetector = false,
                         double harrisK = 0.04);

// cv::FAST(..., TYPE_9_16) and cv::FastFeatureDetector::detect(img, keypoints, mask) (FeatureDetector.cpp:35-40:
// cv::FastFeatureDetector::create(fast_thresh_, nonmaxSuppression = true)): keypoints in raster order,
// response = corner score.
struct FastKeyPoint {
  float x, y, response;
};
void FAST_9_16(const uint8_t* img, int w, int h, size_t stride, int threshold, bool nonmax_suppression,
               std::vector<FastKeyPoint>& keypoints);
void fastDetect(const uint8_t* img, int w, int h, size_t stride, const uint8_t* mask, size_t mask_stride,
                int threshold, bool nonmax, std::vector<FastKeyPoint>& keypoints);

// cv::circle(img, Point(center), radius, color, FILLED, LINE_8, 0) on a u8 image.
// Reference: FeatureDetector.cpp:196-201.
void circle_filled(uint8_t* img, int w, int h, size_t stride, int cx, int cy, int radius,
                   uint8_t color);

// cv::cornerSubPix(img, corners, Size(win,win), Size(zz,zz),
// TermCriteria(COUNT+EPS, max_iters, eps)).  Reference: FeatureDetector.cpp:288-292.
void cornerSubPix(const uint8_t* img, int w, int h, size_t stride, Point2f* corners, int n,
                  int win, int zero_zone, int max_iters, double eps);

// cv::equalizeHist(src u8, dst) (reference call: UtilsOpenCV.cpp:398-401, equalizeImage: 1)
void equalizeHist(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, size_t dstride);

// cv::pyrDown(src u8, dst, Size((w+1)/2,(h+1)/2), BORDER_DEFAULT).
void pyrDown(const uint8_t* src, int sw, int sh, size_t sstride, uint8_t* dst, int dw, int dh,
             size_t dstride);

// cv::matchTemplate(image, templ, result, TM_SQDIFF) — exact integer SSD
// (OpenCV's DFT path carries ~1e-7 relative float noise around these values).
// result: (iw-tw+1) x (ih-th+1) int64.
void matchTemplateSqdiff(const uint8_t* img, int iw, int ih, size_t istride,
                         const uint8_t* templ, int tw, int th, size_t tstride,
                         std::vector<int64_t>& result);

// ---- calib3d: dense stereo (ocv_stereo.cpp) -----------------------------------
// Reference: StereoMatcher.cpp:32-121 (DenseStereoParams, StereoMatchingParams.h:39-58).  PARITY
// UNPINNED (no numeric test in the reference).
enum { STEREO_SGBM_MODE_SGBM = 0, STEREO_SGBM_MODE_HH = 1 };
struct StereoSGBMParams {
  int minDisparity, numDisparities, blockSize, P1, P2, disp12MaxDiff, preFilterCap, uniquenessRatio,
      speckleWindowSize, speckleRange, mode;
};
struct StereoBMParams {   // PREFILTER_XSOBEL, CV_16S output, disp12MaxDiff < 0
  int preFilterCap, blockSize, minDisparity, numDisparities, textureThreshold, uniquenessRatio,
      speckleRange, speckleWindowSize;
  int roi1[4], roi2[4];   // x, y, width, height (all zero: whole image)
};
// cv::StereoSGBM::compute -> CV_16S disparity with 4 fractional bits ((minDisparity-1)*16 = invalid)
void stereoSGBM_compute(const uint8_t* left, const uint8_t* right, int w, int h, size_t stride,
                        const StereoSGBMParams& p, short* disp, size_t dstride);
// cv::StereoBM::compute -> CV_16S
void stereoBM_compute(const uint8_t* left, const uint8_t* right, int w, int h, size_t stride,
                      const StereoBMParams& p, short* disp, size_t dstride);
void medianBlur16s(const short* src, int w, int h, size_t sstride, short* dst, size_t dstride, int ksize);
void filterSpeckles16s(short* img, int w, int h, size_t step, int newVal, int maxSpeckleSize, int maxDiff);
void reprojectImageTo3D(const float* disparity, int w, int h, size_t dstride, const double Q[16],
                        bool handleMissingValues, float* xyz);

// ---- video ------------------------------------------------------------------
// cv::calcOpticalFlowPyrLK(prev, next, prevPts, nextPts, status, err,
// Size(win,win), maxLevel, TermCriteria(COUNT+EPS, maxIter, eps),
// OPTFLOW_USE_INITIAL_FLOW (if use_initial_flow), minEigThreshold), SSE2
// accumulation order.  Reference: Tracker.cpp:137-146.
// Returns the maxLevel actually used.
int calcOpticalFlowPyrLK(const uint8_t* prev, const uint8_t* next, int w, int h, size_t stride,
                         const Point2f* prevPts, Point2f* nextPts, int n, uint8_t* status,
                         float* err, int win, int maxLevel, int maxIter, double eps,
                         bool use_initial_flow, double minEigThreshold);

// buildOpticalFlowPyramid's level images (unpadded), level 0 = copy of img.
struct Pyramid {
  std::vector<std::vector<uint8_t>> img;
  std::vector<int> w, h;
};
int buildPyramid(const uint8_t* img, int w, int h, size_t stride, int win, int maxLevel,
                 Pyramid& pyr);

}  // namespace ocv
