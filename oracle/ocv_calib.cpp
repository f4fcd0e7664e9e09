// TEST INFRASTRUCTURE — see ocv.hpp.  Restatement of OpenCV 4.2 calib3d routines
// (calibration.cpp: cvRodrigues2, cvStereoRectify, icvGetRectangles,
// cvProjectPoints2; undistort.dispatch.cpp: cvUndistortPointsInternal,
// initUndistortRectifyMap) used by the reference at
// src/frontend/StereoCamera.cpp:329-348 and
// src/frontend/UndistorterRectifier.cpp:42-47,248-258.
#include <algorithm>
#include <cfloat>

#include "ocv.hpp"

namespace ocv {

static const double CV_PI_ = 3.1415926535897932384626433832795;


static void matmul3(const double A[9], const double B[9], double C[9]) {
  double t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[k * 3 + j];
      t[i * 3 + j] = s;
    }
  std::memcpy(C, t, sizeof(t));
}

// cv::invert(3x3 double, DECOMP_LU) special case: cofactor formula.
static void invert3(const double S[9], double D[9]) {
  auto s = [&](int r, int c) { return S[r * 3 + c]; };
  double d = s(0, 0) * (s(1, 1) * s(2, 2) - s(1, 2) * s(2, 1)) -
             s(0, 1) * (s(1, 0) * s(2, 2) - s(1, 2) * s(2, 0)) +
             s(0, 2) * (s(1, 0) * s(2, 1) - s(1, 1) * s(2, 0));
  d = 1. / d;
  double t[9];
  t[0] = (s(1, 1) * s(2, 2) - s(1, 2) * s(2, 1)) * d;
  t[1] = (s(0, 2) * s(2, 1) - s(0, 1) * s(2, 2)) * d;
  t[2] = (s(0, 1) * s(1, 2) - s(0, 2) * s(1, 1)) * d;
  t[3] = (s(1, 2) * s(2, 0) - s(1, 0) * s(2, 2)) * d;
  t[4] = (s(0, 0) * s(2, 2) - s(0, 2) * s(2, 0)) * d;
  t[5] = (s(0, 2) * s(1, 0) - s(0, 0) * s(1, 2)) * d;
  t[6] = (s(1, 0) * s(2, 1) - s(1, 1) * s(2, 0)) * d;
  t[7] = (s(0, 1) * s(2, 0) - s(0, 0) * s(2, 1)) * d;
  t[8] = (s(0, 0) * s(1, 1) - s(0, 1) * s(1, 0)) * d;
  std::memcpy(D, t, sizeof(t));
}

void rodrigues_vec_to_mat(const double r_in[3], double R[9]) {
  double rx = r_in[0], ry = r_in[1], rz = r_in[2];
  double theta = std::sqrt(rx * rx + ry * ry + rz * rz);
  if (theta < DBL_EPSILON) {
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1. : 0.;
    return;
  }
  double c = std::cos(theta), s = std::sin(theta), c1 = 1. - c;
  double itheta = theta ? 1. / theta : 0.;
  rx *= itheta;
  ry *= itheta;
  rz *= itheta;
  const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry,
                         ry * rz, rx * rz, ry * rz, rz * rz};
  const double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
  for (int i = 0; i < 9; i++) {
    double eye = (i % 4 == 0) ? 1. : 0.;
    R[i] = c * eye + c1 * rrt[i] + s * r_x[i];
  }
}

void rodrigues_mat_to_vec(const double R[9], double r[3]) {
  // (OpenCV first replaces R by U*Vt of its SVD; skipped: |R^T R - I| ~ 1e-12.)
  double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
  double s = std::sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
  double c = (R[0] + R[4] + R[8] - 1) * 0.5;
  c = c > 1. ? 1. : c < -1. ? -1. : c;
  double theta = std::acos(c);
  if (s < 1e-5) {
    if (c > 0) {
      r[0] = r[1] = r[2] = 0;
    } else {
      double t;
      t = (R[0] + 1) * 0.5;
      rx = std::sqrt(std::max(t, 0.));
      t = (R[4] + 1) * 0.5;
      ry = std::sqrt(std::max(t, 0.)) * (R[1] < 0 ? -1. : 1.);
      t = (R[8] + 1) * 0.5;
      rz = std::sqrt(std::max(t, 0.)) * (R[2] < 0 ? -1. : 1.);
      if (std::fabs(rx) < std::fabs(ry) && std::fabs(rx) < std::fabs(rz) &&
          (R[5] > 0) != (ry * rz > 0))
        rz = -rz;
      theta /= std::sqrt(rx * rx + ry * ry + rz * rz);
      r[0] = rx * theta;
      r[1] = ry * theta;
      r[2] = rz * theta;
    }
  } else {
    double vth = 1 / (2 * s);
    vth *= theta;
    r[0] = rx * vth;
    r[1] = ry * vth;
    r[2] = rz * vth;
  }
}

void undistortPoints(const Point2f* src, Point2f* dst, int n, const double K[9],
                     const double* D, int nD, const double* R, const double* P) {
  double k[14] = {0};
  for (int i = 0; i < nD && i < 14; i++) k[i] = D[i];
  double RR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (R) std::memcpy(RR, R, sizeof(RR));
  if (P) {
    double PP[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]};
    matmul3(PP, RR, RR);
  }
  const double fx = K[0], fy = K[4], ifx = 1. / fx, ify = 1. / fy, cx = K[2], cy = K[5];
  for (int i = 0; i < n; i++) {
    double x = src[i].x, y = src[i].y;
    const double u = x, v = y;
    x = (x - cx) * ifx;
    y = (y - cy) * ify;
    if (D && nD > 0) {
      // tilt model absent (tauX = tauY = 0): invMatTilt = I.
      const double x0 = x, y0 = y;
      for (int j = 0; j < 5; j++) {
        double r2 = x * x + y * y;
        double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) /
                        (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        if (icdist < 0) {  // test: undistortPoints.regression_14583
          x = (u - cx) * ifx;
          y = (v - cy) * ify;
          break;
        }
        double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
        double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
      }
    }
    double xx = RR[0] * x + RR[1] * y + RR[2];
    double yy = RR[3] * x + RR[4] * y + RR[5];
    double ww = 1. / (RR[6] * x + RR[7] * y + RR[8]);
    x = xx * ww;
    y = yy * ww;
    dst[i].x = (float)x;
    dst[i].y = (float)y;
  }
}

// 0 = OpenCV 4.2 (default); 1 = OpenCV <= 3.4.0 behaviour (image corners at nx/ny, fc_new =
// min of the k1-adjusted focals), kept only to cross-check the restatement against rectification
// constants published for EuRoC by third parties (tests/test_oracle_kat.py).
int g_stereo_rectify_variant = 0;

struct Rectf {
  float x, y, width, height;
};

// icvGetRectangles (calibration.cpp)
static void getRectangles(const double K[9], const double* D, int nD, const double R[9],
                          const double P[12], int w, int h, Rectf& inner, Rectf& outer) {
  const int N = 9;
  Point2f pts[N * N];
  int k = 0;
  for (int y = 0; y < N; y++)
    for (int x = 0; x < N; x++) {
      if (g_stereo_rectify_variant == 0) {
        pts[k].x = (float)x * (w - 1) / (N - 1);
        pts[k].y = (float)y * (h - 1) / (N - 1);
      } else {
        pts[k].x = (float)x * w / (N - 1);
        pts[k].y = (float)y * h / (N - 1);
      }
      k++;
    }
  undistortPoints(pts, pts, N * N, K, D, nD, R, P);
  float iX0 = -FLT_MAX, iX1 = FLT_MAX, iY0 = -FLT_MAX, iY1 = FLT_MAX;
  float oX0 = FLT_MAX, oX1 = -FLT_MAX, oY0 = FLT_MAX, oY1 = -FLT_MAX;
  k = 0;
  for (int y = 0; y < N; y++)
    for (int x = 0; x < N; x++) {
      Point2f p = pts[k++];
      oX0 = std::min(oX0, p.x);
      oX1 = std::max(oX1, p.x);
      oY0 = std::min(oY0, p.y);
      oY1 = std::max(oY1, p.y);
      if (x == 0) iX0 = std::max(iX0, p.x);
      if (x == N - 1) iX1 = std::min(iX1, p.x);
      if (y == 0) iY0 = std::max(iY0, p.y);
      if (y == N - 1) iY1 = std::min(iY1, p.y);
    }
  inner = Rectf{iX0, iY0, iX1 - iX0, iY1 - iY0};
  outer = Rectf{oX0, oY0, oX1 - oX0, oY1 - oY0};
}

static void intersect_roi(int r[4], int W, int H) {
  int x1 = std::max(r[0], 0), y1 = std::max(r[1], 0);
  int x2 = std::min(r[0] + r[2], W), y2 = std::min(r[1] + r[3], H);
  r[0] = x1;
  r[1] = y1;
  r[2] = x2 - x1;
  r[3] = y2 - y1;
  if (r[2] <= 0 || r[3] <= 0) r[0] = r[1] = r[2] = r[3] = 0;
}

void stereoRectify(const double K1[9], const double* D1, int nD1, const double K2[9],
                   const double* D2, int nD2, int width, int height, const double Rin[9],
                   const double T[3], double alpha, bool zero_disparity, double R1[9],
                   double R2[9], double P1[12], double P2[12], double Q[16], int roi1[4],
                   int roi2[4]) {
  double om[3], r_r[9], t[3], uu[3] = {0, 0, 0}, ww[3], wR[9], Ri[9];
  const double nx = width, ny = height;

  rodrigues_mat_to_vec(Rin, om);
  for (int i = 0; i < 3; i++) om[i] *= -0.5;  // average rotation
  rodrigues_vec_to_mat(om, r_r);
  for (int i = 0; i < 3; i++) t[i] = r_r[i * 3] * T[0] + r_r[i * 3 + 1] * T[1] + r_r[i * 3 + 2] * T[2];

  const int idx = std::fabs(t[0]) > std::fabs(t[1]) ? 0 : 1;
  const double c = t[idx], nt = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
  uu[idx] = c > 0 ? 1 : -1;

  // global Z rotation
  ww[0] = t[1] * uu[2] - t[2] * uu[1];
  ww[1] = t[2] * uu[0] - t[0] * uu[2];
  ww[2] = t[0] * uu[1] - t[1] * uu[0];
  double nw = std::sqrt(ww[0] * ww[0] + ww[1] * ww[1] + ww[2] * ww[2]);
  if (nw > 0.0) {
    double sc = std::acos(std::fabs(c) / nt) / nw;
    for (int i = 0; i < 3; i++) ww[i] *= sc;
  }
  rodrigues_vec_to_mat(ww, wR);

  // R1 = wR * r_r^T ; R2 = wR * r_r
  double r_rT[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r_rT[i * 3 + j] = r_r[j * 3 + i];
  matmul3(wR, r_rT, Ri);
  std::memcpy(R1, Ri, sizeof(Ri));
  matmul3(wR, r_r, Ri);
  std::memcpy(R2, Ri, sizeof(Ri));
  for (int i = 0; i < 3; i++) t[i] = Ri[i * 3] * T[0] + Ri[i * 3 + 1] * T[1] + Ri[i * 3 + 2] * T[2];

  // new focal length: mean of the two (same image size => ratio = 0.5)
  const double ratio = 0.5;
  double fc_new = (K1[(idx ^ 1) * 4] + K2[(idx ^ 1) * 4]) * ratio;
  double cc_new[2][2];
  const int old = g_stereo_rectify_variant;
  if (old) {
    fc_new = DBL_MAX;
    for (int k = 0; k < 2; k++) {
      const double* A = k == 0 ? K1 : K2;
      const double* Dk = k == 0 ? D1 : D2;
      double dk1 = (Dk && (k == 0 ? nD1 : nD2) > 0) ? Dk[0] : 0;
      double fc = A[(idx ^ 1) * 4];
      if (dk1 < 0) fc *= 1 + dk1 * (nx * nx + ny * ny) / (4 * fc * fc);
      fc_new = std::min(fc_new, fc);
    }
  }

  for (int k = 0; k < 2; k++) {
    const double* A = k == 0 ? K1 : K2;
    const double* Dk = k == 0 ? D1 : D2;
    const int nDk = k == 0 ? nD1 : nD2;
    Point2f pts[4];
    for (int i = 0; i < 4; i++) {
      int j = (i < 2) ? 0 : 1;
      pts[i].x = (float)((i % 2) * (old ? nx : nx - 1));
      pts[i].y = (float)(j * (old ? ny : ny - 1));
    }
    undistortPoints(pts, pts, 4, A, Dk, nDk, nullptr, nullptr);
    // cvProjectPoints2(pts_3, R_k (3x3 -> rvec -> 3x3), 0, A_tmp(fc_new, cc=0), no dist)
    double rv[3], Rk[9];
    rodrigues_mat_to_vec(k == 0 ? R1 : R2, rv);
    rodrigues_vec_to_mat(rv, Rk);
    double sx = 0, sy = 0;
    for (int i = 0; i < 4; i++) {
      double X = pts[i].x, Y = pts[i].y, Z = 1.0f;
      double x = Rk[0] * X + Rk[1] * Y + Rk[2] * Z + 0;
      double y = Rk[3] * X + Rk[4] * Y + Rk[5] * Z + 0;
      double z = Rk[6] * X + Rk[7] * Y + Rk[8] * Z + 0;
      z = z ? 1. / z : 1;
      x *= z;
      y *= z;
      float mx = (float)(x * fc_new + 0.0), my = (float)(y * fc_new + 0.0);
      sx += mx;
      sy += my;
    }
    cc_new[k][0] = (old ? nx : nx - 1) / 2 - sx / 4;
    cc_new[k][1] = (old ? ny : ny - 1) / 2 - sy / 4;
  }

  if (zero_disparity) {
    cc_new[0][0] = cc_new[1][0] = (cc_new[0][0] + cc_new[1][0]) * 0.5;
    cc_new[0][1] = cc_new[1][1] = (cc_new[0][1] + cc_new[1][1]) * 0.5;
  } else if (idx == 0)
    cc_new[0][1] = cc_new[1][1] = (cc_new[0][1] + cc_new[1][1]) * 0.5;
  else
    cc_new[0][0] = cc_new[1][0] = (cc_new[0][0] + cc_new[1][0]) * 0.5;

  double pp[12] = {0};
  pp[0] = pp[5] = fc_new;
  pp[2] = cc_new[0][0];
  pp[6] = cc_new[0][1];
  pp[10] = 1;
  std::memcpy(P1, pp, sizeof(pp));
  pp[2] = cc_new[1][0];
  pp[6] = cc_new[1][1];
  pp[idx * 4 + 3] = t[idx] * fc_new;  // baseline * focal length
  std::memcpy(P2, pp, sizeof(pp));

  alpha = std::min(alpha, 1.);
  Rectf inner1, inner2, outer1, outer2;
  getRectangles(K1, D1, nD1, R1, P1, width, height, inner1, outer1);
  getRectangles(K2, D2, nD2, R2, P2, width, height, inner2, outer2);

  {
    const double W = width, H = height;  // newImgSize == imageSize
    double cx1_0 = cc_new[0][0], cy1_0 = cc_new[0][1];
    double cx2_0 = cc_new[1][0], cy2_0 = cc_new[1][1];
    double cx1 = W * cx1_0 / width, cy1 = H * cy1_0 / height;
    double cx2 = W * cx2_0 / width, cy2 = H * cy2_0 / height;
    double s = 1.;
    if (alpha >= 0) {
      double s0 = std::max(std::max(std::max((double)cx1 / (cx1_0 - inner1.x),
                                             (double)cy1 / (cy1_0 - inner1.y)),
                                    (double)(W - cx1) / (inner1.x + inner1.width - cx1_0)),
                           (double)(H - cy1) / (inner1.y + inner1.height - cy1_0));
      s0 = std::max(std::max(std::max(std::max((double)cx2 / (cx2_0 - inner2.x),
                                               (double)cy2 / (cy2_0 - inner2.y)),
                                      (double)(W - cx2) / (inner2.x + inner2.width - cx2_0)),
                             (double)(H - cy2) / (inner2.y + inner2.height - cy2_0)),
                    s0);
      double s1 = std::min(std::min(std::min((double)cx1 / (cx1_0 - outer1.x),
                                             (double)cy1 / (cy1_0 - outer1.y)),
                                    (double)(W - cx1) / (outer1.x + outer1.width - cx1_0)),
                           (double)(H - cy1) / (outer1.y + outer1.height - cy1_0));
      s1 = std::min(std::min(std::min(std::min((double)cx2 / (cx2_0 - outer2.x),
                                               (double)cy2 / (cy2_0 - outer2.y)),
                                      (double)(W - cx2) / (outer2.x + outer2.width - cx2_0)),
                             (double)(H - cy2) / (outer2.y + outer2.height - cy2_0)),
                    s1);
      s = s0 * (1 - alpha) + s1 * alpha;
    }
    fc_new *= s;
    cc_new[0][0] = cx1;
    cc_new[0][1] = cy1;
    cc_new[1][0] = cx2;
    cc_new[1][1] = cy2;
    P1[0] = fc_new;
    P1[5] = fc_new;
    P1[2] = cx1;
    P1[6] = cy1;
    P2[0] = fc_new;
    P2[5] = fc_new;
    P2[2] = cx2;
    P2[6] = cy2;
    P2[idx * 4 + 3] = s * P2[idx * 4 + 3];
    if (roi1) {
      roi1[0] = cvCeil((inner1.x - cx1_0) * s + cx1);
      roi1[1] = cvCeil((inner1.y - cy1_0) * s + cy1);
      roi1[2] = cvFloor(inner1.width * s);
      roi1[3] = cvFloor(inner1.height * s);
      intersect_roi(roi1, width, height);
    }
    if (roi2) {
      roi2[0] = cvCeil((inner2.x - cx2_0) * s + cx2);
      roi2[1] = cvCeil((inner2.y - cy2_0) * s + cy2);
      roi2[2] = cvFloor(inner2.width * s);
      roi2[3] = cvFloor(inner2.height * s);
      intersect_roi(roi2, width, height);
    }
  }

  if (Q) {
    const double q[16] = {1, 0, 0, -cc_new[0][0],
                          0, 1, 0, -cc_new[0][1],
                          0, 0, 0, fc_new,
                          0, 0, -1. / t[idx],
                          (idx == 0 ? cc_new[0][0] - cc_new[1][0] : cc_new[0][1] - cc_new[1][1]) / t[idx]};
    std::memcpy(Q, q, sizeof(q));
  }
}

void initUndistortRectifyMap(const double K[9], const double* D, int nD, const double R[9],
                             const double P[12], int width, int height, float* map_x,
                             float* map_y) {
  double k[14] = {0};
  for (int i = 0; i < nD && i < 14; i++) k[i] = D[i];
  const double k1 = k[0], k2 = k[1], p1 = k[2], p2 = k[3], k3 = k[4], k4 = k[5], k5 = k[6],
               k6 = k[7], s1 = k[8], s2 = k[9], s3 = k[10], s4 = k[11];
  double Ar[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]};
  double ArR[9], ir[9];
  matmul3(Ar, R, ArR);
  invert3(ArR, ir);
  const double u0 = K[2], v0 = K[5], fx = K[0], fy = K[4];
  for (int i = 0; i < height; i++) {
    float* m1f = map_x + (size_t)i * width;
    float* m2f = map_y + (size_t)i * width;
    double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
    for (int j = 0; j < width; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
      double w = 1. / _w, x = _x * w, y = _y * w;
      double x2 = x * x, y2 = y * y;
      double r2 = x2 + y2, _2xy = 2 * x * y;
      double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
      double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + s1 * r2 + s2 * r2 * r2);
      double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + s3 * r2 + s4 * r2 * r2);
      // matTilt = I
      double vt0 = 1 * xd + 0 * yd + 0 * 1, vt1 = 0 * xd + 1 * yd + 0 * 1, vt2 = 0 * xd + 0 * yd + 1 * 1;
      double invProj = vt2 ? 1. / vt2 : 1;
      double u = fx * invProj * vt0 + u0;
      double v = fy * invProj * vt1 + v0;
      m1f[j] = (float)u;
      m2f[j] = (float)v;
    }
  }
}

// ---------------------------------------------------------------------------
// cv::fisheye (equidistant model)
// ---------------------------------------------------------------------------
namespace fisheye {

static void undistort_one(double px, double py, const double K[9], const double D[4], const double RR[9],
                          double* ox, double* oy) {
  const double f0 = K[0], f1 = K[4], c0 = K[2], c1 = K[5];
  const double pw0 = (px - c0) / f0, pw1 = (py - c1) / f1;
  double scale = 1.0;
  double theta_d = std::sqrt(pw0 * pw0 + pw1 * pw1);
  theta_d = std::min(std::max(-CV_PI_ / 2., theta_d), CV_PI_ / 2.);
  if (theta_d > 1e-8) {
    double theta = theta_d;
    const double EPS = 1e-8;
    for (int j = 0; j < 10; j++) {
      double theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta6 * theta2;
      double k0_theta2 = D[0] * theta2, k1_theta4 = D[1] * theta4, k2_theta6 = D[2] * theta6,
             k3_theta8 = D[3] * theta8;
      double theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d) /
                         (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
      theta = theta - theta_fix;
      if (std::fabs(theta_fix) < EPS) break;
    }
    scale = std::tan(theta) / theta_d;
  }
  const double pu0 = pw0 * scale, pu1 = pw1 * scale;
  const double pr0 = RR[0] * pu0 + RR[1] * pu1 + RR[2] * 1.0;
  const double pr1 = RR[3] * pu0 + RR[4] * pu1 + RR[5] * 1.0;
  const double pr2 = RR[6] * pu0 + RR[7] * pu1 + RR[8] * 1.0;
  *ox = pr0 / pr2;
  *oy = pr1 / pr2;
}

static void make_RR(const double* R, const double* P, double RR[9]) {
  const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  std::memcpy(RR, R ? R : I, sizeof(I));
  if (P) {
    double PP[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]};
    matmul3(PP, RR, RR);
  }
}

void undistortPoints(const Point2f* src, Point2f* dst, int n, const double K[9], const double D[4],
                     const double* R, const double* P) {
  double RR[9];
  make_RR(R, P, RR);
  for (int i = 0; i < n; i++) {
    double x, y;
    undistort_one((double)src[i].x, (double)src[i].y, K, D, RR, &x, &y);
    dst[i].x = (float)x;
    dst[i].y = (float)y;
  }
}

void undistortPointsD(const double* src, double* dst, int n, const double K[9], const double D[4],
                      const double* R) {
  double RR[9];
  make_RR(R, nullptr, RR);
  for (int i = 0; i < n; i++) undistort_one(src[2 * i], src[2 * i + 1], K, D, RR, &dst[2 * i], &dst[2 * i + 1]);
}

void estimateNewCameraMatrixForUndistortRectify(const double K[9], const double D[4], int w, int h,
                                                const double R[9], double newK[9]) {
  double pts[8] = {(double)(w / 2), 0, (double)w, (double)(h / 2), (double)(w / 2), (double)h, 0, (double)(h / 2)};
  undistortPointsD(pts, pts, 4, K, D, R);
  double cn[2] = {(pts[0] + pts[2] + pts[4] + pts[6]) / 4.0, (pts[1] + pts[3] + pts[5] + pts[7]) / 4.0};
  const double aspect_ratio = K[0] / K[4];
  cn[0] *= aspect_ratio;  // sic
  for (int i = 0; i < 4; i++) pts[2 * i + 1] *= aspect_ratio;
  double minx = DBL_MAX, miny = DBL_MAX, maxx = -DBL_MAX, maxy = -DBL_MAX;
  for (int i = 0; i < 4; i++) {
    miny = std::min(miny, pts[2 * i + 1]);
    maxy = std::max(maxy, pts[2 * i + 1]);
    minx = std::min(minx, pts[2 * i]);
    maxx = std::max(maxx, pts[2 * i]);
  }
  double f1 = w * 0.5 / (cn[0] - minx);
  double f2 = w * 0.5 / (maxx - cn[0]);
  double f3 = h * 0.5 * aspect_ratio / (cn[1] - miny);
  double f4 = h * 0.5 * aspect_ratio / (maxy - cn[1]);
  double fmin = std::min(f1, std::min(f2, std::min(f3, f4)));
  double fmax = std::max(f1, std::max(f2, std::max(f3, f4)));
  const double balance = 0.0, fov_scale = 1.0;
  double f = balance * fmin + (1.0 - balance) * fmax;
  f *= fov_scale > 0 ? 1.0 / fov_scale : 1.0;
  double new_f[2] = {f, f};
  double new_c[2] = {-cn[0] * f + w * 0.5, -cn[1] * f + (h * aspect_ratio) * 0.5};
  new_f[1] /= aspect_ratio;
  new_c[1] /= aspect_ratio;
  const double out[9] = {new_f[0], 0, new_c[0], 0, new_f[1], new_c[1], 0, 0, 1};
  std::memcpy(newK, out, sizeof(out));
}

void stereoRectify(const double K1[9], const double D1[4], const double K2[9], const double D2[4], int w,
                   int h, const double R[9], const double T[3], double R1[9], double R2[9],
                   double P1[12], double P2[12], double Q[16]) {
  double rvec[3];
  rodrigues_mat_to_vec(R, rvec);  // Affine3d(rmat).rvec() (its SVD re-orthogonalisation is omitted)
  for (double& v : rvec) v *= -0.5;
  double r_r[9];
  rodrigues_vec_to_mat(rvec, r_r);
  double t[3];
  for (int i = 0; i < 3; i++) t[i] = r_r[i * 3] * T[0] + r_r[i * 3 + 1] * T[1] + r_r[i * 3 + 2] * T[2];
  const double uu[3] = {t[0] > 0 ? 1.0 : -1.0, 0, 0};
  double ww[3] = {t[1] * uu[2] - t[2] * uu[1], t[2] * uu[0] - t[0] * uu[2], t[0] * uu[1] - t[1] * uu[0]};
  const double nw = std::sqrt(ww[0] * ww[0] + ww[1] * ww[1] + ww[2] * ww[2]);
  const double nt = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
  if (nw > 0.0) {
    const double sc = std::acos(std::fabs(t[0]) / nt) / nw;
    for (double& v : ww) v *= sc;
  }
  double wr[9], r_rt[9];
  rodrigues_vec_to_mat(ww, wr);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r_rt[i * 3 + j] = r_r[j * 3 + i];
  matmul3(wr, r_rt, R1);
  matmul3(wr, r_r, R2);
  double tnew[3];
  for (int i = 0; i < 3; i++) tnew[i] = R2[i * 3] * T[0] + R2[i * 3 + 1] * T[1] + R2[i * 3 + 2] * T[2];
  double newK1[9], newK2[9];
  estimateNewCameraMatrixForUndistortRectify(K1, D1, w, h, R1, newK1);
  estimateNewCameraMatrixForUndistortRectify(K2, D2, w, h, R2, newK2);
  const double fc_new = std::min(newK1[4], newK2[4]);
  double cc0[2] = {newK1[2], newK1[5]}, cc1[2] = {newK2[2], newK2[5]};
  for (int i = 0; i < 2; i++) cc0[i] = cc1[i] = (cc0[i] + cc1[i]) * 0.5;  // CALIB_ZERO_DISPARITY
  const double p1[12] = {fc_new, 0, cc0[0], 0, 0, fc_new, cc0[1], 0, 0, 0, 1, 0};
  const double p2[12] = {fc_new, 0, cc1[0], tnew[0] * fc_new, 0, fc_new, cc1[1], 0, 0, 0, 1, 0};
  const double q[16] = {1, 0, 0, -cc0[0], 0, 1, 0, -cc0[1], 0, 0, 0, fc_new,
                        0, 0, -1. / tnew[0], (cc0[0] - cc1[0]) / tnew[0]};
  std::memcpy(P1, p1, sizeof(p1));
  std::memcpy(P2, p2, sizeof(p2));
  std::memcpy(Q, q, sizeof(q));
}

void initUndistortRectifyMap(const double K[9], const double D[4], const double R[9], const double P[12],
                             int w, int h, float* map_x, float* map_y) {
  const double f0 = K[0], f1 = K[4], c0 = K[2], c1 = K[5];
  double PP[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]};
  double PR[9], iR[9];
  matmul3(PP, R, PR);
  invert3(PR, iR);  // cv::invert(DECOMP_SVD) in the reference; closed form here
  for (int i = 0; i < h; ++i) {
    float* m1f = map_x + (size_t)i * w;
    float* m2f = map_y + (size_t)i * w;
    double _x = i * iR[1] + iR[2], _y = i * iR[4] + iR[5], _w = i * iR[7] + iR[8];
    for (int j = 0; j < w; ++j) {
      double x = _x / _w, y = _y / _w;
      double r = std::sqrt(x * x + y * y);
      double theta = std::atan(r);
      double theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
      double theta_d = theta * (1 + D[0] * theta2 + D[1] * theta4 + D[2] * theta6 + D[3] * theta8);
      double scale = (r == 0) ? 1.0 : theta_d / r;
      double u = f0 * x * scale + c0;
      double v = f1 * y * scale + c1;
      m1f[j] = (float)u;
      m2f[j] = (float)v;
      _x += iR[0];
      _y += iR[3];
      _w += iR[6];
    }
  }
}

}  // namespace fisheye

}  // namespace ocv
