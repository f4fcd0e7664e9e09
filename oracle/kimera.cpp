// TEST INFRASTRUCTURE — see kimera.hpp.  Restatement of the reference-owned
// front-end logic; file:line citations are relative to /root/reference.
#include "kimera.hpp"
#include <cfloat>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>

namespace kimera {

void camera_matrix(const kvfe_camera_params& c, double K[9]) {
  // CameraParams::convertIntrinsicsVectorToMatrix (src/frontend/CameraParams.cpp)
  const double k[9] = {c.intrinsics[0], 0, c.intrinsics[2], 0, c.intrinsics[1],
                       c.intrinsics[3], 0, 0, 1};
  std::memcpy(K, k, sizeof(k));
}

static void mat3_mul(const double A[9], const double B[9], double C[9]) {
  double t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[k * 3 + j];
      t[i * 3 + j] = s;
    }
  std::memcpy(C, t, sizeof(t));
}

// --------------------------------------------------------------------------
// StereoCamera::StereoCamera / computeRectificationParameters
// (src/frontend/StereoCamera.cpp:34-94,292-379)
// --------------------------------------------------------------------------
void StereoCamera::init(const kvfe_camera_params& l, const kvfe_camera_params& r) {
  left = l;
  right = r;
  w = l.width;
  h = l.height;
  camera_matrix(l, K1);
  camera_matrix(r, K2);
  // camL_Pose_camR = body_Pose_camL.between(body_Pose_camR)  (StereoCamera.cpp:311-312)
  const double* TL = l.body_pose_cam;
  const double* TR = r.body_pose_cam;
  double RL[9], RR[9], tL[3], tR[3];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) {
      RL[i * 3 + j] = TL[i * 4 + j];
      RR[i * 3 + j] = TR[i * 4 + j];
    }
    tL[i] = TL[i * 4 + 3];
    tR[i] = TR[i * 4 + 3];
  }
  double RLt[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) RLt[i * 3 + j] = RL[j * 3 + i];
  double Rrel[9], trel[3], d[3] = {tR[0] - tL[0], tR[1] - tL[1], tR[2] - tL[2]};
  mat3_mul(RLt, RR, Rrel);
  for (int i = 0; i < 3; i++) trel[i] = RLt[i * 3] * d[0] + RLt[i * 3 + 1] * d[1] + RLt[i * 3 + 2] * d[2];
  // OpenCV convention is the inverse pose (StereoCamera.cpp:314-319)
  double Rinv[9], tinv[3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Rinv[i * 3 + j] = Rrel[j * 3 + i];
  for (int i = 0; i < 3; i++)
    tinv[i] = -(Rinv[i * 3] * trel[0] + Rinv[i * 3 + 1] * trel[1] + Rinv[i * 3 + 2] * trel[2]);

  const bool fisheye = l.distortion_model == KVFE_DIST_EQUIDISTANT;  // StereoCamera.cpp:324-366
  if (fisheye) {
    std::memset(&rect, 0, sizeof(rect));
    ocv::fisheye::stereoRectify(K1, l.distortion, K2, r.distortion, w, h, Rinv, tinv, rect.R1, rect.R2,
                                rect.P1, rect.P2, rect.Q);
  } else {
    ocv::stereoRectify(K1, l.distortion, l.n_distortion, K2, r.distortion, r.n_distortion, w, h,
                       Rinv, tinv, /*alpha=*/0, /*zero_disparity=*/true, rect.R1, rect.R2, rect.P1,
                       rect.P2, rect.Q, rect.roi1, rect.roi2);
  }
  rect.baseline = 1.0 / rect.Q[3 * 4 + 2];  // StereoCamera.cpp:70-72
  for (int c = 0; c < 2; c++) {
    map_x[c].resize((size_t)w * h);
    map_y[c].resize((size_t)w * h);
    const kvfe_camera_params& cp = c == 0 ? l : r;
    if (fisheye)
      ocv::fisheye::initUndistortRectifyMap(c == 0 ? K1 : K2, cp.distortion, c == 0 ? rect.R1 : rect.R2,
                                            c == 0 ? rect.P1 : rect.P2, w, h, map_x[c].data(),
                                            map_y[c].data());
    else
      ocv::initUndistortRectifyMap(c == 0 ? K1 : K2, cp.distortion, cp.n_distortion,
                                   c == 0 ? rect.R1 : rect.R2, c == 0 ? rect.P1 : rect.P2, w, h,
                                   map_x[c].data(), map_y[c].data());
  }
}

void StereoCamera::initMono(const kvfe_camera_params& c) {
  left = right = c;
  w = c.width;
  h = c.height;
  camera_matrix(c, K1);
  camera_matrix(c, K2);
  std::memset(&rect, 0, sizeof(rect));
  for (int i = 0; i < 3; i++) {
    rect.R1[i * 4] = rect.R2[i * 4] = 1.0;
    for (int j = 0; j < 3; j++) rect.P1[i * 4 + j] = rect.P2[i * 4 + j] = K1[i * 3 + j];
  }
  for (int cam = 0; cam < 2; cam++) {
    map_x[cam].resize((size_t)w * h);
    map_y[cam].resize((size_t)w * h);
    if (c.distortion_model == KVFE_DIST_EQUIDISTANT)
      ocv::fisheye::initUndistortRectifyMap(K1, c.distortion, rect.R1, rect.P1, w, h, map_x[cam].data(),
                                            map_y[cam].data());
    else
      ocv::initUndistortRectifyMap(K1, c.distortion, c.n_distortion, rect.R1, rect.P1, w, h,
                                   map_x[cam].data(), map_y[cam].data());
  }
}

void StereoCamera::undistortRectifyImage(int cam, const uint8_t* src, size_t stride,
                                         uint8_t* dst) const {
  ocv::remap_linear_replicate(src, w, h, stride, dst, w, h, w, map_x[cam].data(),
                              map_y[cam].data());
}

void StereoCamera::undistortRectifyKeypoints(int cam, const Point2f* in, int n, bool useR,
                                             bool useP, Point2f* out) const {
  const kvfe_camera_params& cp = cam == 0 ? left : right;
  if (cp.distortion_model == KVFE_DIST_EQUIDISTANT) {  // UndistorterRectifier.cpp:49-57
    ocv::fisheye::undistortPoints(in, out, n, cam == 0 ? K1 : K2, cp.distortion,
                                  useR ? (cam == 0 ? rect.R1 : rect.R2) : nullptr,
                                  useP ? (cam == 0 ? rect.P1 : rect.P2) : nullptr);
    return;
  }
  ocv::undistortPoints(in, out, n, cam == 0 ? K1 : K2, cp.distortion,
                       cp.distortion_model == KVFE_DIST_RADTAN ? cp.n_distortion : 0,
                       useR ? (cam == 0 ? rect.R1 : rect.R2) : nullptr,
                       useP ? (cam == 0 ? rect.P1 : rect.P2) : nullptr);
}

void StereoCamera::getBearingVector(int cam, Point2f kp, double versor[3]) const {
  Point2f u;
  undistortRectifyKeypoints(cam, &kp, 1, true, false, &u);
  double x = u.x, y = u.y, z = 1.0;
  double n = std::sqrt(x * x + (y * y + z * z));  // Eigen Vector3d::normalized()
  versor[0] = x / n;
  versor[1] = y / n;
  versor[2] = z / n;
}

bool cropToSize(Point2f* px, int w, int h) {  // UtilsOpenCV.cpp:215-235
  bool cropped = false;
  float max_width = (float)(w - 1);
  if (px->x > max_width) {
    px->x = max_width;
    cropped = true;
  } else if (px->x < 0.0f) {
    px->x = 0.0f;
    cropped = true;
  }
  float max_height = (float)(h - 1);
  if (px->y > max_height) {
    px->y = max_height;
    cropped = true;
  } else if (px->y < 0.0f) {
    px->y = 0.0f;
    cropped = true;
  }
  return cropped;
}

bool roundAndCropToSize(Point2f* px, int w, int h) {  // UtilsOpenCV.cpp:242-247
  px->x = std::round(px->x);
  px->y = std::round(px->y);
  return cropToSize(px, w, h);
}

// UndistorterRectifier::checkUndistortedRectifiedLeftKeypoints (UndistorterRectifier.cpp:138-211) of camera `cam`
void StereoCamera::checkUndistortedRectifiedKeypoints(int cam, const std::vector<Point2f>& kps,
                                                      const std::vector<Point2f>& und, float pixel_tol,
                                                      std::vector<StatusKeypoint>& out) const {
  out.clear();
  out.reserve(kps.size());
  for (size_t i = 0; i < und.size(); i++) {
    Point2f distorted_kp = kps[i];
    Point2f undistorted_kp = und[i];
    bool cropped = cropToSize(&undistorted_kp, w, h);
    int ry = (int)std::round(undistorted_kp.y), rx = (int)std::round(undistorted_kp.x);
    float ex = map_x[cam][(size_t)ry * w + rx];
    float ey = map_y[cam][(size_t)ry * w + rx];
    if (cropped) {
      out.push_back({KVFE_KP_NO_LEFT_RECT, undistorted_kp});
    } else if (std::fabs(distorted_kp.x - ex) > pixel_tol ||
               std::fabs(distorted_kp.y - ey) > pixel_tol) {
      out.push_back({KVFE_KP_NO_LEFT_RECT, undistorted_kp});
    } else {
      out.push_back({KVFE_KP_VALID, undistorted_kp});
    }
  }
}

void StereoCamera::undistortRectifyLeftKeypoints(const std::vector<Point2f>& kps,
                                                 std::vector<StatusKeypoint>& out) const {
  std::vector<Point2f> und(kps.size());
  undistortRectifyKeypoints(0, kps.data(), (int)kps.size(), true, true, und.data());
  checkUndistortedRectifiedKeypoints(0, kps, und, 2.0f, out);
}

// UndistorterRectifier::distortUnrectifyKeypoints (UndistorterRectifier.cpp:213-228) of camera `cam`
void StereoCamera::distortUnrectifyKeypoints(int cam, const std::vector<StatusKeypoint>& rectk,
                                             std::vector<Point2f>& out) const {
  out.clear();
  out.reserve(rectk.size());
  for (const StatusKeypoint& sk : rectk) {
    if (sk.status == KVFE_KP_VALID) {
      // (upstream: map_x_.at<float>(round(y), round(x)), unchecked in release builds -- a VALID keypoint that cornerSubPix
      // left outside the image reads out of bounds there; defined here, as on the device, as the nearest pixel's entry)
      int ry = std::min(std::max((int)std::round(sk.kp.y), 0), h - 1), rx = std::min(std::max((int)std::round(sk.kp.x), 0), w - 1);
      out.push_back({map_x[cam][(size_t)ry * w + rx], map_y[cam][(size_t)ry * w + rx]});
    } else {
      out.push_back({0.0f, 0.0f});
    }
  }
}

void StereoCamera::distortUnrectifyRightKeypoints(const std::vector<StatusKeypoint>& rectk,
                                                  std::vector<Point2f>& out) const {
  distortUnrectifyKeypoints(1, rectk, out);
}

// --------------------------------------------------------------------------
// cv::sortIdx on all-equal keys (NonMaximumSuppression.cpp:50-60)
// libstdc++ std::sort = introsort(threshold 16, median-of-3 moved to first,
// unguarded partition) + final insertion sort; comparator is always false.
// --------------------------------------------------------------------------
static void introsort_loop_equal(int* first, int* last) {
  while (last - first > 16) {
    int* mid = first + (last - first) / 2;
    // __move_median_to_first(first, first+1, mid, last-1): all comparisons false
    std::swap(*first, *mid);
    // __unguarded_partition(first+1, last, pivot=first)
    int* lo = first + 1;
    int* hi = last;
    for (;;) {
      --hi;
      if (!(lo < hi)) break;
      std::swap(*lo, *hi);
      ++lo;
    }
    int* cut = lo;
    introsort_loop_equal(cut, last);
    last = cut;
  }
}

void sortidx_descending_equal_keys(int n, int policy, std::vector<int>& idx) {
  idx.resize(n);
  for (int i = 0; i < n; i++) idx[i] = i;
  if (policy == KVFE_SORTIDX_STABLE || n == 0) return;
  introsort_loop_equal(idx.data(), idx.data() + n);  // final insertion sort: no-op
  for (int j = 0; j < n / 2; j++) std::swap(idx[j], idx[n - 1 - j]);  // SORT_DESCENDING
}

// --------------------------------------------------------------------------
// ANMS (NonMaximumSuppression.cpp:33-169, anms/anms.cpp:37-49)
// --------------------------------------------------------------------------
// anms::Sdc / KdTree / RangeTree / Ssc (src/frontend/feature-detector/anms/anms.cpp:83-436,
// Bailo et al.): binary search on the suppression radius / width until the greedy sweep keeps
// between K(1-tol) and K(1+tol) keypoints, tol = 0.1f (FeatureDetector.cpp:226).  The spatial index
// of each variant (grid of covered cells, nanoflann k-d tree, u16 range tree) only answers "which
// later keypoints does an accepted keypoint exclude"; that relation is restated directly:
//   SDC       cells of side c = 0.25 r / sqrt(2): sqrt(drow^2 + dcol^2) <= r / c     (anms.cpp:83-162)
//   KdTree    integer pixels, nanoflann radiusSearch(r*r): dx^2 + dy^2 <  r^2         (:164-252)
//   RangeTree u16 pixels, inclusive square [x-w, x+w] x [y-w, y+w]                    (:254-335)
//   SSC       cells of side c = (double)(w / 2) (integer division): |drow|,|dcol| <= floor(w / c) (:337-436)
// Keypoints are integer valued here (GFTT corners before cornerSubPix), so the float -> int / u16
// conversions of the reference are exact.  SSC with w < 2 divides by c = 0 in the reference (the
// process dies on the vector allocation that follows); here the search stops and returns the
// previous sweep.
static void anmsBinarySearch(const std::vector<Point2f>& kp, int numRetPoints, int cols, int rows,
                             int type, std::vector<Point2f>& out) {
  const float tolerance = 0.1f;
  const int n = (int)kp.size();
  int low, high;
  if (type == KVFE_ANMS_SDC) {
    low = 1;
    high = cols;
  } else {
    int exp1 = rows + cols + 2 * numRetPoints;
    long long exp2 = ((long long)4 * cols + (long long)4 * numRetPoints +
                      (long long)4 * rows * numRetPoints + (long long)rows * rows +
                      (long long)cols * cols - (long long)2 * rows * cols +
                      (long long)4 * rows * cols * numRetPoints);
    double exp3 = std::sqrt((double)exp2);
    double exp4 = numRetPoints - 1;
    double sol1 = -std::round((exp1 + exp3) / exp4);
    double sol2 = -std::round((exp1 - exp3) / exp4);
    high = (int)((sol1 > sol2) ? sol1 : sol2);
    low = (int)std::floor(std::sqrt((double)kp.size() / numRetPoints));
  }
  std::vector<int> xi(n), yi(n);
  for (int i = 0; i < n; i++) {
    xi[i] = (int)kp[i].x;
    yi[i] = (int)kp[i].y;
  }
  bool complete = false;
  unsigned int K = numRetPoints;
  unsigned int Kmin = (unsigned int)std::round(K - (K * tolerance));
  unsigned int Kmax = (unsigned int)std::round(K + (K * tolerance));
  std::vector<int> ResultVec, result;
  int r, prev = -1;
  std::vector<int> row(n), col(n);
  while (!complete) {
    std::vector<bool> Included(n, true);
    r = low + (high - low) / 2;
    if (r == prev || low > high) {
      ResultVec = result;
      break;
    }
    double c = 0, reach = 0;
    if (type == KVFE_ANMS_SDC) {
      c = 0.25 * r / std::sqrt(2);
      reach = ((double)r) / c;
    } else if (type == KVFE_ANMS_SSC) {
      c = r / 2;  // integer division, as in the reference
      if (c == 0) {
        ResultVec = result;
        break;
      }
      reach = std::floor(r / c);
    }
    if (type == KVFE_ANMS_SDC || type == KVFE_ANMS_SSC)
      for (int i = 0; i < n; i++) {
        row[i] = (int)std::floor(kp[i].y / c);
        col[i] = (int)std::floor(kp[i].x / c);
      }
    result.clear();
    for (int i = 0; i < n; ++i) {
      if (!Included[i]) continue;
      Included[i] = false;
      result.push_back(i);
      for (int j = i + 1; j < n; j++) {
        if (!Included[j]) continue;
        bool covered;
        switch (type) {
          case KVFE_ANMS_SDC: {
            const int dr = row[j] - row[i], dc = col[j] - col[i];
            covered = std::sqrt((double)(dr * dr + dc * dc)) <= reach;
          } break;
          case KVFE_ANMS_KDTREE: {
            const int dx = xi[j] - xi[i], dy = yi[j] - yi[i];
            covered = dx * dx + dy * dy < r * r;
          } break;
          case KVFE_ANMS_RANGETREE:
            covered = std::abs(xi[j] - xi[i]) <= r && std::abs(yi[j] - yi[i]) <= r;
            break;
          default: {  // SSC
            const int dr = row[j] - row[i], dc = col[j] - col[i];
            covered = std::abs(dr) <= (int)reach && std::abs(dc) <= (int)reach;
          }
        }
        if (covered) Included[j] = false;
      }
    }
    if (result.size() >= Kmin && result.size() <= Kmax) {
      ResultVec = result;
      complete = true;
    } else if (result.size() < Kmin)
      high = r - 1;
    else
      low = r + 1;
    if (type != KVFE_ANMS_SDC) prev = r;  // Sdc never updates prevradius (anms.cpp:83-162)
  }
  for (int i : ResultVec) out.push_back(kp[i]);
}

bool suppressNonMax(const std::vector<Point2f>& keyPoints, int numRetPoints, int cols, int rows,
                    const kvfe_detector_params& p, std::vector<Point2f>& out, const std::vector<float>* responses) {
  out.clear();
  if (keyPoints.empty()) return true;  // "No keypoints for non-max suppression..."
  std::vector<int> Indx;
  if (!responses) {   // GFTT keypoints: response 0 (OpenCV 4.2), all keys equal
    sortidx_descending_equal_keys((int)keyPoints.size(), p.sortidx_policy, Indx);
  } else {
    // NonMaximumSuppression.cpp:50-56: responseVector.push_back(keyPoints[i].response) truncates to int;
    // cv::sortIdx(SORT_DESCENDING) = std::sort(idx, LessThanIdx) -- the REAL std::sort of the host's libstdc++, whose
    // tie order is part of the reference's result -- followed by a reversal (core/src/matrix_sort.cpp sortIdx_)
    const int n = (int)keyPoints.size();
    std::vector<int> key(n);
    for (int i = 0; i < n; i++) key[i] = (int)(*responses)[i];
    Indx.resize(n);
    for (int i = 0; i < n; i++) Indx[i] = i;
    if (p.sortidx_policy == KVFE_SORTIDX_STABLE) {
      std::stable_sort(Indx.begin(), Indx.end(), [&](int a, int b) { return key[a] > key[b]; });
    } else {
      std::sort(Indx.begin(), Indx.end(), [&](int a, int b) { return key[a] < key[b]; });
      for (int j = 0; j < n / 2; j++) std::swap(Indx[j], Indx[n - 1 - j]);
    }
  }
  std::vector<Point2f> keyPointsSorted(keyPoints.size());
  for (size_t i = 0; i < keyPoints.size(); i++) keyPointsSorted[i] = keyPoints[Indx[i]];

  switch (p.non_max_suppression_type) {
    case KVFE_ANMS_TOPN: {  // anms::TopN on the UNSORTED keypoints (NonMaximumSuppression.cpp:67)
      if ((size_t)numRetPoints > keyPoints.size()) {
        out = keyPoints;
        return true;
      }
      for (int i = 0; i < numRetPoints; i++) out.push_back(keyPoints[i]);
      return true;
    }
    case KVFE_ANMS_BROWN: {  // anms::BrownANMS on the UNSORTED keypoints (anms.cpp:51-81, NonMaximumSuppression.cpp:74)
      if ((size_t)numRetPoints > keyPoints.size()) {
        out = keyPoints;
        return true;
      }
      std::vector<std::pair<float, int>> results;
      results.push_back(std::make_pair(FLT_MAX, 0));
      for (unsigned int i = 1; i < keyPoints.size(); ++i) {
        float minDist = FLT_MAX;
        for (unsigned int j = 0; j < i; ++j) {
          float exp1 = (keyPoints[j].x - keyPoints[i].x);
          float exp2 = (keyPoints[j].y - keyPoints[i].y);
          float curDist = std::sqrt(exp1 * exp1 + exp2 * exp2);
          minDist = std::min(curDist, minDist);
        }
        results.push_back(std::make_pair(minDist, (int)i));
      }
      // the REAL std::sort of the host's libstdc++ (the reference's tie order is whatever this does)
      std::sort(results.begin(), results.end(),
                [](const std::pair<float, int>& left, const std::pair<float, int>& right) { return left.first > right.first; });
      for (int i = 0; i < numRetPoints; ++i) out.push_back(keyPoints[results[i].second]);
      return true;
    }
    case KVFE_ANMS_BINNING: {  // NonMaximumSuppression.cpp:125-169
      if ((size_t)numRetPoints > keyPointsSorted.size()) {
        out = keyPointsSorted;
        return true;
      }
      const int hb = p.nr_horizontal_bins, vb = p.nr_vertical_bins;
      float binRowSize = float(rows) / float(vb);
      float binColSize = float(cols) / float(hb);
      double masksum = 0;
      for (int i = 0; i < hb * vb; i++) masksum += p.binning_mask[i];
      float nrActiveBins = (float)masksum;
      const int numRetPointsPerBin = (int)std::round(float(numRetPoints) / float(nrActiveBins));
      std::vector<double> nrKptsInBin((size_t)hb * vb, 0.0);
      for (size_t i = 0; i < keyPointsSorted.size(); i++) {
        const size_t binRowInd = (size_t)(keyPointsSorted[i].y / binRowSize);
        const size_t binColInd = (size_t)(keyPointsSorted[i].x / binColSize);
        if (binRowInd >= (size_t)vb || binColInd >= (size_t)hb) continue;  // (Eigen OOB in ref)
        if (p.binning_mask[binRowInd * hb + binColInd] == 1 &&
            nrKptsInBin[binRowInd * hb + binColInd] < numRetPointsPerBin) {
          out.push_back(keyPointsSorted[i]);
          nrKptsInBin[binRowInd * hb + binColInd] += 1;
        }
      }
      return true;
    }
    case KVFE_ANMS_SDC:
    case KVFE_ANMS_KDTREE:
    case KVFE_ANMS_RANGETREE:
    case KVFE_ANMS_SSC:
      // numRetPoints < 2 (the frame already holds maxFeaturesPerFrame - 1 or more keypoints): KdTree, RangeTree
      // and Ssc divide by numRetPoints - 1 / numRetPoints and convert an infinite double to int (anms.cpp:
      // 175-189 and the like) -- undefined behaviour upstream; defined here as "no new corners".  Sdc has no
      // such term and is restated for every numRetPoints.
      if (numRetPoints < 2 && p.non_max_suppression_type != KVFE_ANMS_SDC) return true;
      anmsBinarySearch(keyPointsSorted, numRetPoints, cols, rows, p.non_max_suppression_type, out);
      return true;
    default:
      return false;
  }
}

// --------------------------------------------------------------------------
// FeatureDetector::featureDetection(const Frame&, need) (FeatureDetector.cpp:174-299)
// --------------------------------------------------------------------------
bool featureDetection(const uint8_t* img, int w, int h, size_t stride,
                      const std::vector<Point2f>& tracked, int need_n_corners,
                      const kvfe_detector_params& p, std::vector<Point2f>& new_corners,
                      std::vector<Point2f>* raw_gftt, const uint8_t* detection_mask) {
  std::vector<uint8_t> mask((size_t)w * h, 255);   // FeatureDetector.cpp:186-190
  if (detection_mask) std::memcpy(mask.data(), detection_mask, mask.size());
  for (const Point2f& kp : tracked)
    ocv::circle_filled(mask.data(), w, h, w, ocv::cvRoundf(kp.x), ocv::cvRoundf(kp.y),
                       p.min_distance, 0);
  std::vector<Point2f> keypoints;
  std::vector<float> responses;
  const bool fast = p.feature_detector_type == KVFE_DET_FAST;
  if (fast) {   // FeatureDetector.cpp:35-40: cv::FastFeatureDetector::create(fast_thresh_, true)->detect(img, keypoints, mask)
    std::vector<ocv::FastKeyPoint> kps;
    ocv::fastDetect(img, w, h, stride, mask.data(), w, p.fast_thresh, true, kps);
    for (const ocv::FastKeyPoint& k : kps) {
      keypoints.push_back(Point2f{k.x, k.y});
      responses.push_back(k.response);
    }
  } else {
    ocv::goodFeaturesToTrack(img, w, h, stride, mask.data(), w, p.max_nr_keypoints_before_anms,
                             p.quality_level, (double)p.min_distance, p.block_size, keypoints, nullptr,
                             p.use_harris_detector != 0, p.k);
  }
  if (raw_gftt) *raw_gftt = keypoints;
  std::vector<Point2f> max_keypoints = keypoints;
  if (p.enable_non_max_suppression) {
    if (!suppressNonMax(keypoints, need_n_corners, w, h, p, max_keypoints, fast ? &responses : nullptr)) return false;
  }
  new_corners = max_keypoints;
  if (!new_corners.empty() && p.enable_subpixel_corner_refinement) {
    ocv::cornerSubPix(img, w, h, stride, new_corners.data(), (int)new_corners.size(),
                      p.subpix_window_size, p.subpix_zero_zone, p.subpix_max_iters,
                      p.subpix_epsilon);
  }
  return true;
}

// --------------------------------------------------------------------------
// OpticalFlowPredictor (optical-flow/OpticalFlowPredictor.cpp:27-33,70-126)
// --------------------------------------------------------------------------
static double quaternion_w(const double m[9]) {  // Eigen::Quaterniond(Matrix3d).w()
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    return 0.5 * t;
  }
  int i = 0;
  if (m[4] > m[0]) i = 1;
  if (m[8] > m[i * 4]) i = 2;
  int j = (i + 1) % 3, k = (j + 1) % 3;
  t = std::sqrt(m[i * 4] - m[j * 4] - m[k * 4] + 1.0);
  t = 0.5 / t;
  return (m[k * 3 + j] - m[j * 3 + k]) * t;
}

static void matx33f_mul(const float a[9], const float b[9], float c[9]) {
  float t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      float s = 0;
      for (int k = 0; k < 3; k++) s += a[i * 3 + k] * b[k * 3 + j];
      t[i * 3 + j] = s;
    }
  std::memcpy(c, t, sizeof(t));
}

static void matx33f_inv(const float a[9], float b[9]) {  // Matx_FastInvOp<float,3,3>
  auto A = [&](int r, int c) { return a[r * 3 + c]; };
  float d = A(0, 0) * (A(1, 1) * A(2, 2) - A(2, 1) * A(1, 2)) -
            A(0, 1) * (A(1, 0) * A(2, 2) - A(2, 0) * A(1, 2)) +
            A(0, 2) * (A(1, 0) * A(2, 1) - A(2, 0) * A(1, 1));
  d = 1 / d;
  b[0] = (A(1, 1) * A(2, 2) - A(1, 2) * A(2, 1)) * d;
  b[1] = (A(0, 2) * A(2, 1) - A(0, 1) * A(2, 2)) * d;
  b[2] = (A(0, 1) * A(1, 2) - A(0, 2) * A(1, 1)) * d;
  b[3] = (A(1, 2) * A(2, 0) - A(1, 0) * A(2, 2)) * d;
  b[4] = (A(0, 0) * A(2, 2) - A(0, 2) * A(2, 0)) * d;
  b[5] = (A(0, 2) * A(1, 0) - A(0, 0) * A(1, 2)) * d;
  b[6] = (A(1, 0) * A(2, 1) - A(1, 1) * A(2, 0)) * d;
  b[7] = (A(0, 1) * A(2, 0) - A(0, 0) * A(2, 1)) * d;
  b[8] = (A(0, 0) * A(1, 1) - A(0, 1) * A(1, 0)) * d;
}

void predictSparseFlow(int type, const double K[9], int w, int h, const Point2f* prev, int n,
                       const double R_in[9], Point2f* next) {
  if (type == KVFE_FLOW_NO_PREDICTION) {
    for (int i = 0; i < n; i++) next[i] = prev[i];
    return;
  }
  static constexpr double kSmallRotationTol = 1e-4;
  if (std::abs(1.0 - std::abs(quaternion_w(R_in))) < kSmallRotationTol) {
    for (int i = 0; i < n; i++) next[i] = prev[i];
    return;
  }
  float Kf[9], Kinv[9], Rt[9], KR[9], H[9];
  for (int i = 0; i < 9; i++) Kf[i] = (float)K[i];
  matx33f_inv(Kf, Kinv);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Rt[i * 3 + j] = (float)R_in[j * 3 + i];
  matx33f_mul(Kf, Rt, KR);
  matx33f_mul(KR, Kinv, H);
  for (int i = 0; i < n; i++) {
    const Point2f prev_kpt = prev[i];
    const float p1[3] = {prev_kpt.x, prev_kpt.y, 1.0f};
    float p2[3];
    for (int r = 0; r < 3; r++) {
      float s = 0;
      for (int k = 0; k < 3; k++) s += H[r * 3 + k] * p1[k];
      p2[r] = s;
    }
    Point2f new_kpt;
    if (p2[2] > 0.0f)
      new_kpt = {p2[0] / p2[2], p2[1] / p2[2]};
    else
      new_kpt = prev_kpt;
    // cv::Rect2f(0,0,w,h).contains(new_kpt)
    if (0.0f <= new_kpt.x && new_kpt.x < (float)w && 0.0f <= new_kpt.y && new_kpt.y < (float)h)
      next[i] = new_kpt;
    else
      next[i] = prev_kpt;
  }
}

// --------------------------------------------------------------------------
// StereoMatcher (StereoMatcher.cpp:196-483)
// --------------------------------------------------------------------------
static void searchRightKeypointEpipolar(const uint8_t* left_rect, Point2f left_kp,
                                        const uint8_t* right_rect, int w, int h, size_t stride,
                                        int stripe_cols, int stripe_rows,
                                        const kvfe_stereo_params& p, StatusKeypoint* out,
                                        double* score) {
  int rounded_x = (int)std::round(left_kp.x);
  int rounded_y = (int)std::round(left_kp.y);
  int temp_corner_y = rounded_y - (p.templ_rows - 1) / 2;
  if (temp_corner_y < 0 || temp_corner_y + p.templ_rows > h - 1) {
    *score = -1.0;
    *out = {KVFE_KP_NO_RIGHT_RECT, {0.0f, 0.0f}};
    return;
  }
  int offset_temp = 0;
  int temp_corner_x = rounded_x - (p.templ_cols - 1) / 2;
  if (temp_corner_x < 0) {
    offset_temp = temp_corner_x;
    temp_corner_x = 0;
  }
  if (temp_corner_x + p.templ_cols > w - 1) {
    offset_temp = (temp_corner_x + p.templ_cols) - (w - 1);
    temp_corner_x -= offset_temp;
  }
  int stripe_corner_y = rounded_y - (stripe_rows - 1) / 2;
  if (stripe_corner_y < 0 || stripe_corner_y + stripe_rows > h - 1) {
    *score = -1.0;
    *out = {KVFE_KP_NO_RIGHT_RECT, {0.0f, 0.0f}};
    return;
  }
  int offset_stripe = 0;
  int stripe_corner_x = rounded_x + (p.templ_cols - 1) / 2 - stripe_cols;
  if (stripe_corner_x + stripe_cols > w - 1) {
    offset_stripe = (stripe_corner_x + stripe_cols) - (w - 1);
    stripe_corner_x -= offset_stripe;
  }
  if (stripe_corner_x < 0) stripe_corner_x = 0;

  std::vector<int64_t> result;
  ocv::matchTemplateSqdiff(right_rect + (size_t)stripe_corner_y * stride + stripe_corner_x,
                           stripe_cols, stripe_rows, stride,
                           left_rect + (size_t)temp_corner_y * stride + temp_corner_x,
                           p.templ_cols, p.templ_rows, stride, result);
  const int rw = stripe_cols - p.templ_cols + 1, rh = stripe_rows - p.templ_rows + 1;
  // normalize(0,1,MINMAX) + minMaxLoc: first minimum in row-major order; the
  // normalised minimum is 0 (also for a flat result).
  // kvfe_stereo_params.ssd_tie_policy: KVFE_SSD_TIE_EXACT compares the exact integers; KVFE_SSD_TIE_F32 compares
  // them rounded to float32 (the CV_32F result matrix of cv::matchTemplate; int64 -> float converts with round to
  // nearest even), first minimum in row-major order either way.
  int bx = 0, by = 0;
  if (p.ssd_tie_policy == KVFE_SSD_TIE_F32) {
    float bestf = std::numeric_limits<float>::infinity();
    for (int y = 0; y < rh; y++)
      for (int x = 0; x < rw; x++) {
        const float v = (float)result[(size_t)y * rw + x];
        if (v < bestf) {
          bestf = v;
          bx = x;
          by = y;
        }
      }
  } else {
    int64_t best = std::numeric_limits<int64_t>::max();
    for (int y = 0; y < rh; y++)
      for (int x = 0; x < rw; x++)
        if (result[(size_t)y * rw + x] < best) {
          best = result[(size_t)y * rw + x];
          bx = x;
          by = y;
        }
  }
  double min_val = 0.0;
  int mx = bx + stripe_corner_x + (p.templ_cols - 1) / 2 + offset_temp;
  int my = by + stripe_corner_y + (p.templ_rows - 1) / 2;
  Point2f match_px = {(float)mx, (float)my};
  if (p.subpixel_refinement) {  // StereoMatcher.cpp:404-413
    ocv::cornerSubPix(right_rect, w, h, stride, &match_px, 1, 10, -1, 40, 0.001);
  }
  *score = min_val;
  if (min_val < p.tolerance_template_matching)
    *out = {KVFE_KP_VALID, match_px};
  else
    *out = {KVFE_KP_NO_RIGHT_RECT, match_px};
}

void getRightKeypointsRectified(const uint8_t* left_rect, const uint8_t* right_rect, int w, int h,
                                size_t stride, const std::vector<StatusKeypoint>& left,
                                double fx, double baseline, const kvfe_stereo_params& p,
                                std::vector<StatusKeypoint>& right, std::vector<double>* scores) {
  right.clear();
  right.reserve(left.size());
  if (scores) scores->clear();
  int stripe_rows = p.templ_rows + p.stripe_extra_rows;
  int stripe_cols = (int)std::round(fx * baseline / p.min_point_dist) + p.templ_cols + 4;
  if (stripe_cols % 2 != 1) stripe_cols += 1;
  if (stripe_cols > w) stripe_cols = w;
  for (const StatusKeypoint& lk : left) {
    if (lk.status != KVFE_KP_VALID) {
      right.push_back({lk.status, {0.0f, 0.0f}});
      if (scores) scores->push_back(-1.0);
      continue;
    }
    StatusKeypoint cand;
    double score;
    searchRightKeypointEpipolar(left_rect, lk.kp, right_rect, w, h, stride, stripe_cols,
                                stripe_rows, p, &cand, &score);
    right.push_back(cand);
    if (scores) scores->push_back(score);
  }
}

void getDepthFromRectifiedMatches(std::vector<StatusKeypoint>& left,
                                  std::vector<StatusKeypoint>& right, double fx, double baseline,
                                  const kvfe_stereo_params& p, std::vector<double>& depths) {
  depths.clear();
  double fx_b = fx * baseline;
  for (size_t i = 0; i < left.size(); i++) {
    if (left[i].status == KVFE_KP_VALID && right[i].status == KVFE_KP_VALID) {
      double disparity = left[i].kp.x - right[i].kp.x;
      if (disparity >= 0.0) {
        double depth = fx_b / disparity;
        if (depth < p.min_point_dist || depth > p.max_point_dist) {
          right[i].status = KVFE_KP_NO_DEPTH;
          depths.push_back(0.0);
        } else
          depths.push_back(depth);
      } else {
        right[i].status = KVFE_KP_NO_DEPTH;
        depths.push_back(0.0);
      }
    } else {
      if (left[i].status != KVFE_KP_VALID && right[i].status != left[i].status)
        right[i].status = left[i].status;
      depths.push_back(0.0);
    }
  }
}

void sparseStereoReconstruction(const StereoCamera& cam, const kvfe_stereo_params& p,
                                StereoFrame& sf) {
  const int w = cam.w, h = cam.h;
  sf.left_rect.resize((size_t)w * h);
  sf.right_rect.resize((size_t)w * h);
  cam.undistortRectifyImage(0, sf.left.img.data(), w, sf.left_rect.data());
  cam.undistortRectifyImage(1, sf.right_img.data(), w, sf.right_rect.data());
  cam.undistortRectifyLeftKeypoints(sf.left.keypoints, sf.left_kp_rect);
  getRightKeypointsRectified(sf.left_rect.data(), sf.right_rect.data(), w, h, w, sf.left_kp_rect,
                             cam.fx(), cam.baseline(), p, sf.right_kp_rect, nullptr);
  getDepthFromRectifiedMatches(sf.left_kp_rect, sf.right_kp_rect, cam.fx(), cam.baseline(), p,
                               sf.depth);
  cam.distortUnrectifyRightKeypoints(sf.right_kp_rect, sf.right_kp);
  sf.kp3d.assign(sf.right_kp_rect.size() * 3, 0.0);
  for (size_t i = 0; i < sf.right_kp_rect.size(); i++) {
    if (sf.right_kp_rect[i].status == KVFE_KP_VALID) {
      const double* v = &sf.left.versors[i * 3];
      for (int c = 0; c < 3; c++) sf.kp3d[i * 3 + c] = v[c] * sf.depth[i] / v[2];
    }
  }
}

// --------------------------------------------------------------------------
// Frontend
// --------------------------------------------------------------------------
void Frontend::init(const kvfe_camera_params& l, const kvfe_camera_params& r,
                    const kvfe_frontend_params& fp, bool mono_frontend) {
  mono = mono_frontend;
  if (mono)
    cam.initMono(l);
  else
    cam.init(l, r);
  p = fp;
  lmk_id = 0;
  frame_count = 0;
  initialized = false;
  tracker_status = TrackerStatusSummary();
}

// FeatureDetector::featureDetection(Frame*, R) (FeatureDetector.cpp:94-163)
void Frontend::featureDetectionFrame(Frame& f, int* n_detected, const uint8_t* detection_mask) {
  int n_existing = 0;
  std::vector<Point2f> tracked;
  for (size_t i = 0; i < f.landmarks.size(); ++i) {
    if (f.landmarks[i] != -1) {
      ++n_existing;
      tracked.push_back(f.keypoints[i]);
    }
    f.landmarks_age[i]++;
  }
  int need = std::max(p.detector.max_features_per_frame - n_existing, 0);
  std::vector<Point2f> corners;
  featureDetection(f.img.data(), f.w, f.h, f.w, tracked, need, p.detector, corners, nullptr, detection_mask);
  for (const Point2f& c : corners) {
    f.landmarks.push_back(lmk_id);
    f.landmarks_age.push_back(1);
    f.keypoints.push_back(c);
    double v[3];
    cam.getBearingVector(0, c, v);
    f.versors.insert(f.versors.end(), v, v + 3);
    ++lmk_id;
  }
  if (n_detected) *n_detected = (int)corners.size();
}

// Tracker::featureTracking (Tracker.cpp:92-211)
void Frontend::featureTracking(Frame& ref, Frame& cur, const double ref_R_cur[9]) {
  std::vector<Point2f> px_ref;
  std::vector<size_t> idx_valid;
  for (size_t i = 0; i < ref.keypoints.size(); ++i)
    if (ref.landmarks[i] != -1) {
      px_ref.push_back(ref.keypoints[i]);
      idx_valid.push_back(i);
    }
  std::vector<Point2f> px_cur(px_ref.size());
  // predictor built from the ORIGINAL left K and image size (Tracker.cpp:65-69)
  predictSparseFlow(p.tracker.optical_flow_predictor_type, cam.K1, cam.w, cam.h, px_ref.data(),
                    (int)px_ref.size(), ref_R_cur, px_cur.data());
  std::vector<uint8_t> status(px_ref.size());
  std::vector<float> error(px_ref.size());
  ocv::calcOpticalFlowPyrLK(ref.img.data(), cur.img.data(), cam.w, cam.h, cam.w, px_ref.data(),
                            px_cur.data(), (int)px_ref.size(), status.data(), error.data(),
                            p.tracker.klt_win_size, p.tracker.klt_max_level,
                            p.tracker.klt_max_iter, p.tracker.klt_eps, true, 1e-4);
  for (size_t i = 0; i < idx_valid.size(); ++i) {
    const size_t idx = idx_valid[i];
    const int32_t lmk_age = ref.landmarks_age[idx];
    const int64_t id = ref.landmarks[idx];
    if (!status[i] || lmk_age > p.tracker.max_feature_track_age) {
      ref.landmarks[idx] = -1;
      continue;
    }
    cur.landmarks.push_back(id);
    cur.landmarks_age.push_back(lmk_age);
    cur.keypoints.push_back(px_cur[i]);
    double v[3];
    cam.getBearingVector(0, px_cur[i], v);
    cur.versors.insert(cur.versors.end(), v, v + 3);
  }
}

// VisionImuFrontend::shouldBeKeyframe (VisionImuFrontend.cpp:175-232) with
// Tracker::findMatchingKeypoints / computeMedianDisparity (Tracker.cpp:919-1018);
// kfTrackingStatus_mono_ (of the last keyframe) can only be LOW_DISPARITY when useRANSAC = 1.
bool Frontend::shouldBeKeyframe(const Frame& frame, const Frame& frame_lkf) const {
  const int64_t kf_diff_ns = frame.timestamp - frame_lkf.timestamp;
  size_t nr_valid_features = 0;
  for (int64_t l : frame.landmarks)
    if (l != -1) nr_valid_features++;
  const bool min_time_elapsed = (double)kf_diff_ns >= p.min_intra_keyframe_time_ns;
  const bool max_time_elapsed = (double)kf_diff_ns >= p.max_intra_keyframe_time_ns;
  const bool nr_features_low = nr_valid_features <= (size_t)p.min_number_features;

  std::map<int64_t, size_t> ref_map;
  for (size_t i = 0; i < frame_lkf.landmarks.size(); ++i)
    if (frame_lkf.landmarks[i] != -1) ref_map[frame_lkf.landmarks[i]] = i;
  std::vector<double> disparity_sq;
  for (size_t i = 0; i < frame.landmarks.size(); ++i) {
    if (frame.landmarks[i] == -1) continue;
    auto it = ref_map.find(frame.landmarks[i]);
    if (it == ref_map.end()) continue;
    float dx = frame.keypoints[i].x - frame_lkf.keypoints[it->second].x;
    float dy = frame.keypoints[i].y - frame_lkf.keypoints[it->second].y;
    double px_dist = dx * dx + dy * dy;
    disparity_sq.push_back(px_dist);
  }
  double disparity = 0.0;
  if (!disparity_sq.empty()) {
    const size_t center = disparity_sq.size() / 2;
    std::nth_element(disparity_sq.begin(), disparity_sq.begin() + center, disparity_sq.end());
    disparity = std::sqrt(disparity_sq[center]);
  }
  const bool is_disparity_low = disparity < p.tracker.disparity_threshold;
  const bool disparity_low_first_time =
      is_disparity_low && !(tracker_status.mono == KVFE_TRACKING_LOW_DISPARITY);
  const bool enough_disparity = !is_disparity_low;
  const bool max_disparity_reached = disparity > p.max_disparity_since_lkf;
  const bool disparity_flipped = ((enough_disparity || disparity_low_first_time) && min_time_elapsed);
  return max_time_elapsed || max_disparity_reached || disparity_flipped || nr_features_low ||
         frame.isKeyframe;
}

// getSmartStereoMeasurements (StereoVisionImuFrontend.cpp:485-531)
void smartStereoMeasurements(const StereoFrame& sf, bool use_stereo_tracking, std::vector<int64_t>& meas_lmk,
                             std::vector<double>& meas_uLuRv) {
  meas_lmk.clear();
  meas_uLuRv.clear();
  for (size_t i = 0; i < sf.left.landmarks.size(); ++i) {
    if (sf.left.landmarks[i] == -1) continue;
    double uL = sf.left_kp_rect[i].kp.x, v = sf.left_kp_rect[i].kp.y;
    double uR = std::numeric_limits<double>::quiet_NaN();
    if (use_stereo_tracking && sf.right_kp_rect[i].status == KVFE_KP_VALID)
      uR = sf.right_kp_rect[i].kp.x;
    meas_lmk.push_back(sf.left.landmarks[i]);
    meas_uLuRv.push_back(uL);
    meas_uLuRv.push_back(uR);
    meas_uLuRv.push_back(v);
  }
}
void Frontend::getSmartStereoMeasurements(const StereoFrame& sf) {
  smartStereoMeasurements(sf, p.use_stereo_tracking != 0, meas_lmk, meas_uLuRv);
}

void Frontend::process(const uint8_t* left, const uint8_t* right, size_t stride,
                       const kvfe_frame_input& in) {
  const int w = cam.w, h = cam.h;
  k = StereoFrame();
  k.left.id = frame_count;
  k.left.timestamp = in.timestamp_ns;
  k.left.isKeyframe = in.force_keyframe != 0;
  k.left.w = w;
  k.left.h = h;
  k.left.img.resize((size_t)w * h);
  k.right_img.resize((size_t)w * h);
  for (int y = 0; y < h; y++) {
    std::memcpy(&k.left.img[(size_t)y * w], left + (size_t)y * stride, w);
    if (!rgbd) std::memcpy(&k.right_img[(size_t)y * w], right + (size_t)y * stride, w);
  }
  if (p.stereo.equalize_image && !rgbd) {  // UtilsOpenCV::ReadAndConvertToGrayScale(name, equalize) on both views
    ocv::equalizeHist(k.left.img.data(), w, h, w, k.left.img.data(), w);
    ocv::equalizeHist(k.right_img.data(), w, h, w, k.right_img.data(), w);
  }
  meas_lmk.clear();
  meas_uLuRv.clear();
  if (rgbd) {
    processRgbd(in, right, stride);
    return;
  }
  if (mono) {
    processMono(in);
    return;
  }

  if (!initialized) {  // processFirstStereoFrame (StereoVisionImuFrontend.cpp:245-276)
    k.left.isKeyframe = true;
    featureDetectionFrame(k.left, &k.n_detected);
    if (!k.left.keypoints.empty()) sparseStereoReconstruction(cam, p.stereo, k);
    km1 = k;
    lkf = k;
    km1_is_lkf = true;
    ++frame_count;
    initialized = true;
    last_is_keyframe = true;
    for (int i = 0; i < 9; i++) keyframe_R_ref_frame[i] = (i % 4 == 0) ? 1.0 : 0.0;
    return;
  }

  // processStereoFrame (StereoVisionImuFrontend.cpp:283-481), useRANSAC = 0
  double RrefT[9], ref_R_cur[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) RrefT[i * 3 + j] = keyframe_R_ref_frame[j * 3 + i];
  mat3_mul(RrefT, in.keyframe_R_cur_frame, ref_R_cur);
  featureTracking(km1.left, k.left, ref_R_cur);
  if (km1_is_lkf) lkf.left.landmarks = km1.left.landmarks;  // same object in the reference
  k.n_tracked = (int)k.left.keypoints.size();

  if (k.left.keypoints.empty()) {  // :313-323
    featureDetectionFrame(k.left, &k.n_detected);
    km1 = k;
    km1_is_lkf = false;
    ++frame_count;
    last_is_keyframe = false;
    return;
  }
  const bool new_keyframe = shouldBeKeyframe(k.left, lkf.left);
  if (new_keyframe) {
    // StereoVisionImuFrontend.cpp:349-412
    tracker_status.mono = KVFE_TRACKING_INVALID;
    tracker_status.stereo = KVFE_TRACKING_INVALID;
    if (p.use_ransac) {
      outlierRejectionMono(in.keyframe_R_cur_frame, lkf.left, k.left);
      sparseStereoReconstruction(cam, p.stereo, k);
      if (p.use_stereo_tracking) {
        outlierRejectionStereo(in.keyframe_R_cur_frame, lkf, k);
      } else {
        tracker_status.stereo = KVFE_TRACKING_INVALID;
      }
      if (p.use_pnp_tracking) {   // :389-399
        outlierRejectionPnP(k);
      } else {
        tracker_status.pnp = KVFE_TRACKING_INVALID;
        const double I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        std::memcpy(tracker_status.W_T_k_pnp, I, sizeof(I));
      }
    } else {
      tracker_status.mono = KVFE_TRACKING_DISABLED;
      tracker_status.stereo = KVFE_TRACKING_DISABLED;
    }
    k.left.isKeyframe = true;
    featureDetectionFrame(k.left, &k.n_detected);
    sparseStereoReconstruction(cam, p.stereo, k);
    lkf = k;
    getSmartStereoMeasurements(k);
    for (int i = 0; i < 9; i++) keyframe_R_ref_frame[i] = (i % 4 == 0) ? 1.0 : 0.0;
  } else {
    k.left.isKeyframe = false;
    std::memcpy(keyframe_R_ref_frame, in.keyframe_R_cur_frame, sizeof(keyframe_R_ref_frame));
  }
  last_is_keyframe = new_keyframe;
  km1 = k;
  km1_is_lkf = new_keyframe;
  ++frame_count;
}

// MonoVisionImuFrontend::processFirstFrame / processFrame (src/frontend/MonoVisionImuFrontend.cpp:
// 196-335) + getSmartMonoMeasurements (:340-369); `k` already holds the new image
void Frontend::processMono(const kvfe_frame_input& in) {
  auto undistortKeypoints = [&]() {  // Camera::undistortKeypoints (Camera.cpp:110-133)
    cam.undistortRectifyLeftKeypoints(k.left.keypoints, k.left_kp_rect);
    k.right_kp_rect.assign(k.left_kp_rect.size(), StatusKeypoint{0, {0.f, 0.f}});
    k.depth.assign(k.left_kp_rect.size(), 0.0);
    k.right_kp.assign(k.left_kp_rect.size(), Point2f{0.f, 0.f});
    k.kp3d.assign(k.left_kp_rect.size() * 3, 0.0);
  };
  if (!initialized) {
    k.left.isKeyframe = true;
    featureDetectionFrame(k.left, &k.n_detected);
    undistortKeypoints();
    km1 = k;
    lkf = k;
    km1_is_lkf = true;
    ++frame_count;
    initialized = true;
    last_is_keyframe = true;
    for (int i = 0; i < 9; i++) keyframe_R_ref_frame[i] = (i % 4 == 0) ? 1.0 : 0.0;
    return;
  }
  double RrefT[9], ref_R_cur[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) RrefT[i * 3 + j] = keyframe_R_ref_frame[j * 3 + i];
  mat3_mul(RrefT, in.keyframe_R_cur_frame, ref_R_cur);
  featureTracking(km1.left, k.left, ref_R_cur);
  if (km1_is_lkf) lkf.left.landmarks = km1.left.landmarks;
  k.n_tracked = (int)k.left.keypoints.size();
  tracker_status.mono = KVFE_TRACKING_INVALID;
  tracker_status.stereo = KVFE_TRACKING_DISABLED;
  const bool new_keyframe = shouldBeKeyframe(k.left, lkf.left);
  if (new_keyframe) {
    if (p.use_ransac)
      outlierRejectionMono(in.keyframe_R_cur_frame, lkf.left, k.left);
    else
      tracker_status.mono = KVFE_TRACKING_DISABLED;
    k.left.isKeyframe = true;
    featureDetectionFrame(k.left, &k.n_detected);
    undistortKeypoints();
    lkf = k;
    for (size_t i = 0; i < k.left.landmarks.size(); ++i) {  // getSmartMonoMeasurements
      if (k.left.landmarks[i] == -1) continue;
      meas_lmk.push_back(k.left.landmarks[i]);
      meas_uLuRv.push_back((double)k.left_kp_rect[i].kp.x);
      meas_uLuRv.push_back(std::numeric_limits<double>::quiet_NaN());
      meas_uLuRv.push_back((double)k.left_kp_rect[i].kp.y);
    }
    for (int i = 0; i < 9; i++) keyframe_R_ref_frame[i] = (i % 4 == 0) ? 1.0 : 0.0;
  } else {
    k.left.isKeyframe = false;
    std::memcpy(keyframe_R_ref_frame, in.keyframe_R_cur_frame, sizeof(keyframe_R_ref_frame));
  }
  last_is_keyframe = new_keyframe;
  km1 = k;
  km1_is_lkf = new_keyframe;
  ++frame_count;
}

// ---------------------------------------------------------------------------------------------
// RgbdVisionImuFrontend (src/frontend/RgbdVisionImuFrontend.cpp:184-431)
// ---------------------------------------------------------------------------------------------
void Frontend::initRgbd(const kvfe_camera_params& c, const kvfe_frontend_params& fp,
                        const kvfe_depth_params& dp) {
  init(c, c, fp, true);   // RgbdCamera is a Camera: undistortion with R = I, P = K like the mono front-end
  rgbd = true;
  depth_params = dp;
  // RgbdCamera::getFakeStereoCalib (RgbdCamera.cpp:92-100): Cal3_S2Stereo(fx, fy, skew, px, py, virtual baseline)
  cam.rect.baseline = (double)dp.virtual_baseline;
}

// DepthFrame::getDetectionMask (DepthFrame.cpp:76-96): cv::inRange(depth, min, max)
void depthDetectionMask(const kvfe_depth_params& dp, const void* depth, int w, int h, size_t stride,
                        std::vector<uint8_t>& mask) {
  mask.assign((size_t)w * h, 0);
  const float mn = dp.min_depth * 1.0f / dp.depth_to_meters;
  const float mx = dp.max_depth * 1.0f / dp.depth_to_meters;
  if (dp.depth_type == KVFE_DEPTH_F32) {
    const float* d = static_cast<const float*>(depth);
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) {
        const float v = d[(size_t)y * stride + x];
        mask[(size_t)y * w + x] = (mn <= v && v <= mx) ? 255 : 0;
      }
  } else {
    const uint16_t lo = static_cast<uint16_t>(mn), hi = static_cast<uint16_t>(mx);
    const uint16_t* d = static_cast<const uint16_t*>(depth);
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) {
        const uint16_t v = d[(size_t)y * stride + x];
        mask[(size_t)y * w + x] = (lo <= v && v <= hi) ? 255 : 0;
      }
  }
}
void Frontend::depthDetectionMask(const void* depth, size_t stride, std::vector<uint8_t>& mask) const {
  kimera::depthDetectionMask(depth_params, depth, cam.w, cam.h, stride, mask);
}

// DepthFrame::getDepthAtPoint (DepthFrame.cpp:40-74); (w, h) = size of the depth image
float depthAtPoint(const kvfe_depth_params& dp, const void* depth, int w, int h, size_t stride, Point2f pt) {
  const int x = static_cast<int>(pt.x), y = static_cast<int>(pt.y);
  float d = std::numeric_limits<float>::quiet_NaN();
  if (x < 0 || x >= w || y < 0 || y >= h) return d;
  if (dp.depth_type == KVFE_DEPTH_F32)
    d = static_cast<const float*>(depth)[(size_t)y * stride + x];
  else
    d = static_cast<const uint16_t*>(depth)[(size_t)y * stride + x];
  d *= dp.depth_to_meters;
  if (d < dp.min_depth) return std::numeric_limits<float>::quiet_NaN();
  return d;
}

// RgbdFrame::fillStereoFrame (RgbdFrame.cpp:48-115) + DepthFrame::getDepthAtPoint (DepthFrame.cpp:40-74)
void Frontend::fillStereoFrame(StereoFrame& sf, const void* depth, size_t stride, int dw, int dh) const {
  const int w = cam.w;
  if (dw <= 0) dw = cam.w;
  if (dh <= 0) dh = cam.h;
  const size_t n = sf.left_kp_rect.size();
  const double fx_b = cam.left.intrinsics[0] * depth_params.virtual_baseline;
  sf.right_kp_rect.assign(n, StatusKeypoint{0, {0.f, 0.f}});
  sf.depth.assign(n, 0.0);
  sf.kp3d.assign(n * 3, 0.0);
  sf.right_kp.assign(n, Point2f{0.f, 0.f});
  for (size_t i = 0; i < n; ++i) {
    const StatusKeypoint& lk = sf.left_kp_rect[i];
    if (lk.status != KVFE_KP_VALID) {
      sf.right_kp_rect[i].status = lk.status;
      continue;
    }
    const float d = depthAtPoint(depth_params, depth, dw, dh, stride, sf.left.keypoints[i]);
    if (!std::isfinite(d)) {
      sf.right_kp_rect[i].status = KVFE_KP_NO_DEPTH;
      continue;
    }
    const float disparity = (float)(fx_b / d);
    const float uR = lk.kp.x - disparity;
    if (uR < 0.0f) {
      sf.right_kp_rect[i].status = KVFE_KP_NO_DEPTH;
      continue;
    }
    sf.right_kp_rect[i] = StatusKeypoint{KVFE_KP_VALID, {uR, lk.kp.y}};
    sf.depth[i] = d;
    const double* v = &sf.left.versors[3 * i];
    for (int c = 0; c < 3; c++) sf.kp3d[3 * i + c] = v[c] * (double)d / v[2];
  }
  // RgbdCamera::distortKeypoints -> UndistorterRectifier::distortUnrectifyKeypoints (:213-228)
  for (size_t i = 0; i < n; ++i) {
    if (sf.right_kp_rect[i].status != KVFE_KP_VALID) continue;
    const Point2f px = sf.right_kp_rect[i].kp;
    const int ry = std::min(std::max((int)std::round(px.y), 0), cam.h - 1), rx = std::min(std::max((int)std::round(px.x), 0), cam.w - 1);
    sf.right_kp[i] = Point2f{cam.map_x[0][(size_t)ry * w + rx], cam.map_y[0][(size_t)ry * w + rx]};
  }
}

void Frontend::processRgbd(const kvfe_frame_input& in, const void* depth, size_t stride) {
  std::vector<uint8_t> mask;
  auto undistortKeypoints = [&]() {  // Camera::undistortKeypoints (Camera.cpp:110-133)
    cam.undistortRectifyLeftKeypoints(k.left.keypoints, k.left_kp_rect);
  };
  if (!initialized) {   // processFirstFrame (:184-217)
    k.left.isKeyframe = true;
    depthDetectionMask(depth, stride, mask);
    featureDetectionFrame(k.left, &k.n_detected, mask.data());
    undistortKeypoints();
    fillStereoFrame(k, depth, stride);
    km1 = k;
    lkf = k;
    km1_is_lkf = true;
    ++frame_count;
    initialized = true;
    last_is_keyframe = true;
    for (int i = 0; i < 9; i++) keyframe_R_ref_frame[i] = (i % 4 == 0) ? 1.0 : 0.0;
    return;
  }
  // processFrame (:236-298)
  double RrefT[9], ref_R_cur[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) RrefT[i * 3 + j] = keyframe_R_ref_frame[j * 3 + i];
  mat3_mul(RrefT, in.keyframe_R_cur_frame, ref_R_cur);
  featureTracking(km1.left, k.left, ref_R_cur);
  if (km1_is_lkf) lkf.left.landmarks = km1.left.landmarks;
  k.n_tracked = (int)k.left.keypoints.size();
  undistortKeypoints();
  // the stereo tables of a non-keyframe stay those of the constructor (RgbdFrame::getStereoFrame): empty
  k.right_kp_rect.assign(k.left_kp_rect.size(), StatusKeypoint{0, {0.f, 0.f}});
  k.depth.assign(k.left_kp_rect.size(), 0.0);
  k.right_kp.assign(k.left_kp_rect.size(), Point2f{0.f, 0.f});
  k.kp3d.assign(k.left_kp_rect.size() * 3, 0.0);
  tracker_status.mono = KVFE_TRACKING_INVALID;
  tracker_status.stereo = KVFE_TRACKING_INVALID;
  const bool new_keyframe = shouldBeKeyframe(k.left, lkf.left);
  if (new_keyframe) {   // handleKeyframe (:300-368)
    if (p.use_ransac) {
      outlierRejectionMono(in.keyframe_R_cur_frame, lkf.left, k.left);
      fillStereoFrame(k, depth, stride);
      if (p.use_stereo_tracking)
        outlierRejectionStereo(in.keyframe_R_cur_frame, lkf, k);
      else
        tracker_status.stereo = KVFE_TRACKING_INVALID;
      if (p.use_pnp_tracking) {   // RgbdVisionImuFrontend.cpp:328-334
        outlierRejectionPnP(k);
      } else {
        tracker_status.pnp = KVFE_TRACKING_INVALID;
        const double I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        std::memcpy(tracker_status.W_T_k_pnp, I, sizeof(I));
      }
    } else {
      tracker_status.mono = KVFE_TRACKING_DISABLED;
      tracker_status.stereo = KVFE_TRACKING_DISABLED;
      tracker_status.pnp = KVFE_TRACKING_DISABLED;   // :345
    }
    k.left.isKeyframe = true;
    depthDetectionMask(depth, stride, mask);
    featureDetectionFrame(k.left, &k.n_detected, mask.data());
    undistortKeypoints();
    fillStereoFrame(k, depth, stride);
    // fillSmartStereoMeasurements (:370-399)
    for (size_t i = 0; i < k.left.landmarks.size(); ++i) {
      if (k.left.landmarks[i] == -1) continue;
      meas_lmk.push_back(k.left.landmarks[i]);
      meas_uLuRv.push_back((double)k.left_kp_rect[i].kp.x);
      meas_uLuRv.push_back(k.right_kp_rect[i].status == KVFE_KP_VALID
                               ? (double)k.right_kp_rect[i].kp.x
                               : std::numeric_limits<double>::quiet_NaN());
      meas_uLuRv.push_back((double)k.left_kp_rect[i].kp.y);
    }
    lkf = k;
    for (int i = 0; i < 9; i++) keyframe_R_ref_frame[i] = (i % 4 == 0) ? 1.0 : 0.0;
  } else {
    k.left.isKeyframe = false;
    std::memcpy(keyframe_R_ref_frame, in.keyframe_R_cur_frame, sizeof(keyframe_R_ref_frame));
  }
  last_is_keyframe = new_keyframe;
  km1 = k;
  km1_is_lkf = new_keyframe;
  ++frame_count;
}

// StereoMatcher::denseStereoReconstruction (StereoMatcher.cpp:32-121)
void denseStereoReconstruction(const kvfe_dense_stereo_params& dp, const int roi1[4], const int roi2[4],
                               const uint8_t* left_rect, const uint8_t* right_rect, int w, int h,
                               size_t stride, short* disp) {
  if (dp.use_sgbm) {   // :47-65
    ocv::StereoSGBMParams p{dp.min_disparity, dp.num_disparities, dp.sad_window_size, dp.p1, dp.p2,
                            dp.disp_12_max_diff, dp.pre_filter_cap, dp.uniqueness_ratio,
                            dp.speckle_window_size, dp.speckle_range,
                            dp.use_mode_hh ? ocv::STEREO_SGBM_MODE_HH : ocv::STEREO_SGBM_MODE_SGBM};
    ocv::stereoSGBM_compute(left_rect, right_rect, w, h, stride, p, disp, w);
  } else {             // :66-90 (cv::StereoBM::create(numDisparities, blockSize) + setters)
    ocv::StereoBMParams p;
    p.preFilterCap = dp.pre_filter_cap;
    p.blockSize = dp.sad_window_size;
    p.minDisparity = dp.min_disparity;
    p.numDisparities = dp.num_disparities;
    p.textureThreshold = dp.texture_threshold;
    p.uniquenessRatio = dp.uniqueness_ratio;
    p.speckleRange = dp.speckle_range;
    p.speckleWindowSize = dp.speckle_window_size;
    const bool have = roi1 && roi2 && roi1[2] > 0 && roi1[3] > 0 && roi2[2] > 0 && roi2[3] > 0;   // :81-86
    for (int i = 0; i < 4; i++) {
      p.roi1[i] = have ? roi1[i] : 0;
      p.roi2[i] = have ? roi2[i] : 0;
    }
    ocv::stereoBM_compute(left_rect, right_rect, w, h, stride, p, disp, w);
  }
  if (dp.median_blur_disparity) ocv::medianBlur16s(disp, w, h, w, disp, w, 5);   // :104-107
}

}  // namespace kimera
