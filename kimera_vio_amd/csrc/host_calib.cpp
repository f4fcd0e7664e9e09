// See host_calib.hpp.  Double-precision init-time math, evaluated in the same
// operation order as OpenCV 4.2's generic C++ path so that rectification
// constants and maps agree with the reference CPU front-end to the last bit
// wherever IEEE-754 allows (no FMA contraction: built with -ffp-contract=off).
#include "host_calib.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

namespace kvfe {

M3 mul(const M3& a, const M3& b) {
  M3 o;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += a(r, k) * b(k, c);
      o(r, c) = s;
    }
  return o;
}

V3 mul(const M3& a, const V3& x) {
  V3 o;
  for (int r = 0; r < 3; r++) o.v[r] = a(r, 0) * x.v[0] + a(r, 1) * x.v[1] + a(r, 2) * x.v[2];
  return o;
}

M3 inv3(const M3& a) {
  // cofactor form used by cv::invert for 3x3
  double d = a(0, 0) * (a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1)) -
             a(0, 1) * (a(1, 0) * a(2, 2) - a(1, 2) * a(2, 0)) +
             a(0, 2) * (a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0));
  d = 1. / d;
  M3 o;
  o(0, 0) = (a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1)) * d;
  o(0, 1) = (a(0, 2) * a(2, 1) - a(0, 1) * a(2, 2)) * d;
  o(0, 2) = (a(0, 1) * a(1, 2) - a(0, 2) * a(1, 1)) * d;
  o(1, 0) = (a(1, 2) * a(2, 0) - a(1, 0) * a(2, 2)) * d;
  o(1, 1) = (a(0, 0) * a(2, 2) - a(0, 2) * a(2, 0)) * d;
  o(1, 2) = (a(0, 2) * a(1, 0) - a(0, 0) * a(1, 2)) * d;
  o(2, 0) = (a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0)) * d;
  o(2, 1) = (a(0, 1) * a(2, 0) - a(0, 0) * a(2, 1)) * d;
  o(2, 2) = (a(0, 0) * a(1, 1) - a(0, 1) * a(1, 0)) * d;
  return o;
}

M3 rodrigues(const V3& rv) {
  double x = rv.v[0], y = rv.v[1], z = rv.v[2];
  const double theta = std::sqrt(x * x + y * y + z * z);
  if (theta < DBL_EPSILON) return M3::eye();
  const double c = std::cos(theta), s = std::sin(theta), c1 = 1. - c;
  const double it = theta ? 1. / theta : 0.;
  x *= it;
  y *= it;
  z *= it;
  const double rrt[9] = {x * x, x * y, x * z, x * y, y * y, y * z, x * z, y * z, z * z};
  const double rx[9] = {0, -z, y, z, 0, -x, -y, x, 0};
  M3 R;
  for (int i = 0; i < 9; i++) R.m[i] = c * (i % 4 == 0 ? 1. : 0.) + c1 * rrt[i] + s * rx[i];
  return R;
}

V3 rodrigues_inv(const M3& R) {
  double x = R(2, 1) - R(1, 2), y = R(0, 2) - R(2, 0), z = R(1, 0) - R(0, 1);
  const double s = std::sqrt((x * x + y * y + z * z) * 0.25);
  double c = (R(0, 0) + R(1, 1) + R(2, 2) - 1) * 0.5;
  c = c > 1. ? 1. : c < -1. ? -1. : c;
  double theta = std::acos(c);
  V3 r;
  if (s < 1e-5) {
    if (c > 0) return V3{{0, 0, 0}};
    double t = (R(0, 0) + 1) * 0.5;
    x = std::sqrt(std::max(t, 0.));
    t = (R(1, 1) + 1) * 0.5;
    y = std::sqrt(std::max(t, 0.)) * (R(0, 1) < 0 ? -1. : 1.);
    t = (R(2, 2) + 1) * 0.5;
    z = std::sqrt(std::max(t, 0.)) * (R(0, 2) < 0 ? -1. : 1.);
    if (std::fabs(x) < std::fabs(y) && std::fabs(x) < std::fabs(z) && (R(1, 2) > 0) != (y * z > 0))
      z = -z;
    theta /= std::sqrt(x * x + y * y + z * z);
    r = V3{{x * theta, y * theta, z * theta}};
  } else {
    double vth = 1 / (2 * s);
    vth *= theta;
    r = V3{{x * vth, y * vth, z * vth}};
  }
  return r;
}

M3 camera_matrix(const kvfe_camera_params& c) {
  return M3{{c.intrinsics[0], 0, c.intrinsics[2], 0, c.intrinsics[1], c.intrinsics[3], 0, 0, 1}};
}

UndistortCtx make_undistort_ctx(const kvfe_camera_params& cam, const double* R, const double* P) {
  UndistortCtx u;
  u.fx = cam.intrinsics[0];
  u.fy = cam.intrinsics[1];
  u.cx = cam.intrinsics[2];
  u.cy = cam.intrinsics[3];
  u.ifx = 1. / u.fx;
  u.ify = 1. / u.fy;
  for (int i = 0; i < 8; i++) u.k[i] = i < cam.n_distortion ? cam.distortion[i] : 0.0;
  u.has_dist = (cam.distortion_model == KVFE_DIST_RADTAN && cam.n_distortion > 0) ? 1
               : cam.distortion_model == KVFE_DIST_EQUIDISTANT                      ? 2
                                                                                    : 0;
  M3 RR = M3::eye();
  if (R) std::memcpy(RR.m, R, sizeof(RR.m));
  if (P) {
    M3 PP{{P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]}};
    RR = mul(PP, RR);
  }
  u.RR = RR;
  return u;
}

// cv::fisheye::undistortPoints (calib3d/src/fisheye.cpp, OpenCV 4.2) for one pixel
static void fisheye_undistort_point(const UndistortCtx& u, float x_in, float y_in, float* x_out,
                                    float* y_out) {
  const double* k = u.k;
  const double pw0 = ((double)x_in - u.cx) / u.fx, pw1 = ((double)y_in - u.cy) / u.fy;
  double scale = 1.0;
  double theta_d = std::sqrt(pw0 * pw0 + pw1 * pw1);
  theta_d = std::min(std::max(-3.14159265358979323846 / 2., theta_d), 3.14159265358979323846 / 2.);
  if (theta_d > 1e-8) {
    double theta = theta_d;
    const double EPS = 1e-8;
    for (int j = 0; j < 10; j++) {
      const double theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2,
                   theta8 = theta6 * theta2;
      const double k0_theta2 = k[0] * theta2, k1_theta4 = k[1] * theta4, k2_theta6 = k[2] * theta6,
                   k3_theta8 = k[3] * theta8;
      const double theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d) /
                               (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
      theta = theta - theta_fix;
      if (std::fabs(theta_fix) < EPS) break;
    }
    scale = std::tan(theta) / theta_d;
  }
  const double pu0 = pw0 * scale, pu1 = pw1 * scale;
  const M3& RR = u.RR;
  const double pr0 = RR(0, 0) * pu0 + RR(0, 1) * pu1 + RR(0, 2) * 1.0;
  const double pr1 = RR(1, 0) * pu0 + RR(1, 1) * pu1 + RR(1, 2) * 1.0;
  const double pr2 = RR(2, 0) * pu0 + RR(2, 1) * pu1 + RR(2, 2) * 1.0;
  *x_out = (float)(pr0 / pr2);
  *y_out = (float)(pr1 / pr2);
}

void undistort_point(const UndistortCtx& u, float x_in, float y_in, float* x_out, float* y_out) {
  if (u.has_dist == 2) {
    fisheye_undistort_point(u, x_in, y_in, x_out, y_out);
    return;
  }
  const double* k = u.k;  // k1 k2 p1 p2 k3 k4 k5 k6
  double x = x_in, y = y_in;
  const double px = x, py = y;
  x = (x - u.cx) * u.ifx;
  y = (y - u.cy) * u.ify;
  if (u.has_dist == 1) {
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
      const double r2 = x * x + y * y;
      const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) /
                            (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
      if (icdist < 0) {
        x = (px - u.cx) * u.ifx;
        y = (py - u.cy) * u.ify;
        break;
      }
      const double dX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + 0 * r2 + 0 * r2 * r2;
      const double dY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + 0 * r2 + 0 * r2 * r2;
      x = (x0 - dX) * icdist;
      y = (y0 - dY) * icdist;
    }
  }
  const M3& RR = u.RR;
  const double xx = RR(0, 0) * x + RR(0, 1) * y + RR(0, 2);
  const double yy = RR(1, 0) * x + RR(1, 1) * y + RR(1, 2);
  const double ww = 1. / (RR(2, 0) * x + RR(2, 1) * y + RR(2, 2));
  *x_out = (float)(xx * ww);
  *y_out = (float)(yy * ww);
}

namespace {

struct RectF {
  float x, y, w, h;
};

void valid_rectangles(const kvfe_camera_params& cam, const double R[9], const double P[12],
                      RectF* inner, RectF* outer) {
  const int N = 9;
  const UndistortCtx u = make_undistort_ctx(cam, R, P);
  float iX0 = -FLT_MAX, iX1 = FLT_MAX, iY0 = -FLT_MAX, iY1 = FLT_MAX;
  float oX0 = FLT_MAX, oX1 = -FLT_MAX, oY0 = FLT_MAX, oY1 = -FLT_MAX;
  for (int gy = 0; gy < N; gy++)
    for (int gx = 0; gx < N; gx++) {
      float px = (float)gx * (cam.width - 1) / (N - 1);
      float py = (float)gy * (cam.height - 1) / (N - 1);
      float ux, uy;
      undistort_point(u, px, py, &ux, &uy);
      oX0 = std::min(oX0, ux);
      oX1 = std::max(oX1, ux);
      oY0 = std::min(oY0, uy);
      oY1 = std::max(oY1, uy);
      if (gx == 0) iX0 = std::max(iX0, ux);
      if (gx == N - 1) iX1 = std::min(iX1, ux);
      if (gy == 0) iY0 = std::max(iY0, uy);
      if (gy == N - 1) iY1 = std::min(iY1, uy);
    }
  *inner = RectF{iX0, iY0, iX1 - iX0, iY1 - iY0};
  *outer = RectF{oX0, oY0, oX1 - oX0, oY1 - oY0};
}

inline int iceil(double v) {
  int i = (int)v;
  return i + (i < v);
}
inline int ifloor(double v) {
  int i = (int)v;
  return i - (i > v);
}

void clip_roi(int r[4], int W, int H) {
  int x1 = std::max(r[0], 0), y1 = std::max(r[1], 0);
  int x2 = std::min(r[0] + r[2], W), y2 = std::min(r[1] + r[3], H);
  r[0] = x1;
  r[1] = y1;
  r[2] = x2 - x1;
  r[3] = y2 - y1;
  if (r[2] <= 0 || r[3] <= 0) r[0] = r[1] = r[2] = r[3] = 0;
}

}  // namespace

namespace {

// cv::fisheye::estimateNewCameraMatrixForUndistortRectify(K, D, size, R, P, balance = 0,
// new_size = (), fov_scale = 1) (calib3d/src/fisheye.cpp)
M3 fisheye_estimate_new_camera_matrix(const kvfe_camera_params& cam, const M3& K, const M3& R) {
  const int w = cam.width, h = cam.height;
  const UndistortCtx u = make_undistort_ctx(cam, R.m, nullptr);
  // the four edge mid-points, undistorted in float64 (the function works on a CV_64FC2 array)
  const double pts[4][2] = {{(double)(w / 2), 0.0}, {(double)w, (double)(h / 2)}, {(double)(w / 2), (double)h},
                            {0.0, (double)(h / 2)}};
  double q[4][2];
  for (int i = 0; i < 4; i++) {
    // float64 path of fisheye::undistortPoints: same arithmetic as fisheye_undistort_point without
    // the float rounding of the input / output
    const double* k = u.k;
    const double pw0 = (pts[i][0] - u.cx) / u.fx, pw1 = (pts[i][1] - u.cy) / u.fy;
    double scale = 1.0;
    double theta_d = std::sqrt(pw0 * pw0 + pw1 * pw1);
    theta_d = std::min(std::max(-3.14159265358979323846 / 2., theta_d), 3.14159265358979323846 / 2.);
    if (theta_d > 1e-8) {
      double theta = theta_d;
      for (int j = 0; j < 10; j++) {
        const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t6 * t2;
        const double a = k[0] * t2, b = k[1] * t4, c = k[2] * t6, d = k[3] * t8;
        const double fix = (theta * (1 + a + b + c + d) - theta_d) / (1 + 3 * a + 5 * b + 7 * c + 9 * d);
        theta = theta - fix;
        if (std::fabs(fix) < 1e-8) break;
      }
      scale = std::tan(theta) / theta_d;
    }
    const double pu0 = pw0 * scale, pu1 = pw1 * scale;
    const double pr0 = u.RR(0, 0) * pu0 + u.RR(0, 1) * pu1 + u.RR(0, 2);
    const double pr1 = u.RR(1, 0) * pu0 + u.RR(1, 1) * pu1 + u.RR(1, 2);
    const double pr2 = u.RR(2, 0) * pu0 + u.RR(2, 1) * pu1 + u.RR(2, 2);
    q[i][0] = pr0 / pr2;
    q[i][1] = pr1 / pr2;
  }
  double cn[2] = {(q[0][0] + q[1][0] + q[2][0] + q[3][0]) / 4.0, (q[0][1] + q[1][1] + q[2][1] + q[3][1]) / 4.0};
  const double aspect_ratio = K(0, 0) / K(1, 1);
  cn[0] *= aspect_ratio;  // sic: "convert to identity ratio" scales cn[0] and the points' y
  for (int i = 0; i < 4; i++) q[i][1] *= aspect_ratio;
  double minx = DBL_MAX, miny = DBL_MAX, maxx = -DBL_MAX, maxy = -DBL_MAX;
  for (int i = 0; i < 4; i++) {
    miny = std::min(miny, q[i][1]);
    maxy = std::max(maxy, q[i][1]);
    minx = std::min(minx, q[i][0]);
    maxx = std::max(maxx, q[i][0]);
  }
  const double f1 = w * 0.5 / (cn[0] - minx), f2 = w * 0.5 / (maxx - cn[0]);
  const double f3 = h * 0.5 * aspect_ratio / (cn[1] - miny), f4 = h * 0.5 * aspect_ratio / (maxy - cn[1]);
  const double fmin = std::min(f1, std::min(f2, std::min(f3, f4)));
  const double fmax = std::max(f1, std::max(f2, std::max(f3, f4)));
  const double balance = 0.0;
  double f = balance * fmin + (1.0 - balance) * fmax;
  f *= 1.0;  // fov_scale = 1
  double new_f[2] = {f, f};
  double new_c[2] = {-cn[0] * f + w * 0.5, -cn[1] * f + (h * aspect_ratio) * 0.5};
  new_f[1] /= aspect_ratio;
  new_c[1] /= aspect_ratio;
  return M3{{new_f[0], 0, new_c[0], 0, new_f[1], new_c[1], 0, 0, 1}};
}

// cv::fisheye::stereoRectify(K1, D1, K2, D2, size, R, T, R1, R2, P1, P2, Q, CALIB_ZERO_DISPARITY)
// (StereoCamera.cpp:350-366; balance 0, fov_scale 1, newImageSize = imageSize)
kvfe_status fisheye_stereo_rectify(const kvfe_camera_params& L, const kvfe_camera_params& Rc, const M3& K1,
                                   const M3& K2, const M3& Rcv, const V3& Tcv, kvfe_rectification* out) {
  V3 rvec = rodrigues_inv(Rcv);  // Affine3d(rmat).rvec()
  for (double& v : rvec.v) v *= -0.5;
  const M3 r_r = rodrigues(rvec);
  const V3 t = mul(r_r, Tcv);
  const V3 uu{{t.v[0] > 0 ? 1.0 : -1.0, 0, 0}};
  V3 ww{{t.v[1] * uu.v[2] - t.v[2] * uu.v[1], t.v[2] * uu.v[0] - t.v[0] * uu.v[2],
         t.v[0] * uu.v[1] - t.v[1] * uu.v[0]}};
  const double nw = std::sqrt(ww.v[0] * ww.v[0] + ww.v[1] * ww.v[1] + ww.v[2] * ww.v[2]);
  const double nt = std::sqrt(t.v[0] * t.v[0] + t.v[1] * t.v[1] + t.v[2] * t.v[2]);
  if (nw > 0.0) {
    const double sc = std::acos(std::fabs(t.v[0]) / nt) / nw;
    for (double& v : ww.v) v *= sc;
  }
  const M3 wr = rodrigues(ww);
  const M3 ri1 = mul(wr, r_r.t());
  const M3 ri2 = mul(wr, r_r);
  const V3 tnew = mul(ri2, Tcv);
  const M3 newK1 = fisheye_estimate_new_camera_matrix(L, K1, ri1);
  const M3 newK2 = fisheye_estimate_new_camera_matrix(Rc, K2, ri2);
  const double fc_new = std::min(newK1(1, 1), newK2(1, 1));
  double cc0[2] = {newK1(0, 2), newK1(1, 2)}, cc1[2] = {newK2(0, 2), newK2(1, 2)};
  // CALIB_ZERO_DISPARITY: both principal points become their average
  for (int i = 0; i < 2; i++) cc0[i] = cc1[i] = (cc0[i] + cc1[i]) * 0.5;
  std::memset(out, 0, sizeof(*out));
  std::memcpy(out->R1, ri1.m, sizeof(out->R1));
  std::memcpy(out->R2, ri2.m, sizeof(out->R2));
  const double P1[12] = {fc_new, 0, cc0[0], 0, 0, fc_new, cc0[1], 0, 0, 0, 1, 0};
  const double P2[12] = {fc_new, 0, cc1[0], tnew.v[0] * fc_new, 0, fc_new, cc1[1], 0, 0, 0, 1, 0};
  std::memcpy(out->P1, P1, sizeof(P1));
  std::memcpy(out->P2, P2, sizeof(P2));
  const double Q[16] = {1, 0, 0, -cc0[0], 0, 1, 0, -cc0[1], 0, 0, 0, fc_new,
                        0, 0, -1. / tnew.v[0], (cc0[0] - cc1[0]) / tnew.v[0]};
  std::memcpy(out->Q, Q, sizeof(Q));
  // (cv::fisheye::stereoRectify reports no valid-pixel ROIs)
  if (out->Q[14] == 0.0) return KVFE_ERR_INVALID_ARG;
  out->baseline = 1.0 / out->Q[14];
  if (!(out->baseline > 0.0)) return KVFE_ERR_INVALID_ARG;
  return KVFE_OK;
}

}  // namespace

kvfe_status stereo_rectify(const kvfe_camera_params& L, const kvfe_camera_params& Rc,
                           kvfe_rectification* out) {
  if (L.width != Rc.width || L.height != Rc.height || L.width <= 0 || L.height <= 0)
    return KVFE_ERR_INVALID_ARG;
  // StereoCamera::computeRectificationParameters switches on the LEFT camera's model
  // (StereoCamera.cpp:324-378); a mixed pair is not a configuration the reference can express
  if ((L.distortion_model == KVFE_DIST_EQUIDISTANT) != (Rc.distortion_model == KVFE_DIST_EQUIDISTANT))
    return KVFE_ERR_UNSUPPORTED;
  const bool fisheye = L.distortion_model == KVFE_DIST_EQUIDISTANT;
  const double nx = L.width, ny = L.height;
  const M3 K1 = camera_matrix(L), K2 = camera_matrix(Rc);

  // camL_Pose_camR = body_Pose_camL^-1 * body_Pose_camR, then inverted for OpenCV
  // (StereoCamera.cpp:311-319)
  M3 RL, RR;
  V3 tL, tR;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) {
      RL(r, c) = L.body_pose_cam[r * 4 + c];
      RR(r, c) = Rc.body_pose_cam[r * 4 + c];
    }
    tL.v[r] = L.body_pose_cam[r * 4 + 3];
    tR.v[r] = Rc.body_pose_cam[r * 4 + 3];
  }
  const M3 Rrel = mul(RL.t(), RR);
  const V3 trel = mul(RL.t(), V3{{tR.v[0] - tL.v[0], tR.v[1] - tL.v[1], tR.v[2] - tL.v[2]}});
  const M3 Rcv = Rrel.t();
  V3 Tcv = mul(Rcv, trel);
  for (double& v : Tcv.v) v = -v;

  if (fisheye) return fisheye_stereo_rectify(L, Rc, K1, K2, Rcv, Tcv, out);

  // Bouguet: split the rotation, align the baseline with the x (or y) axis.
  V3 om = rodrigues_inv(Rcv);
  for (double& v : om.v) v *= -0.5;
  const M3 r_r = rodrigues(om);
  V3 t = mul(r_r, Tcv);
  const int idx = std::fabs(t.v[0]) > std::fabs(t.v[1]) ? 0 : 1;
  const double c = t.v[idx], nt = std::sqrt(t.v[0] * t.v[0] + t.v[1] * t.v[1] + t.v[2] * t.v[2]);
  if (!(nt > 0.0)) return KVFE_ERR_INVALID_ARG;
  V3 uu{{0, 0, 0}};
  uu.v[idx] = c > 0 ? 1 : -1;
  V3 ww{{t.v[1] * uu.v[2] - t.v[2] * uu.v[1], t.v[2] * uu.v[0] - t.v[0] * uu.v[2],
         t.v[0] * uu.v[1] - t.v[1] * uu.v[0]}};
  const double nw = std::sqrt(ww.v[0] * ww.v[0] + ww.v[1] * ww.v[1] + ww.v[2] * ww.v[2]);
  if (nw > 0.0) {
    const double sc = std::acos(std::fabs(c) / nt) / nw;
    for (double& v : ww.v) v *= sc;
  }
  const M3 wR = rodrigues(ww);
  const M3 R1 = mul(wR, r_r.t());
  const M3 R2 = mul(wR, r_r);
  t = mul(R2, Tcv);

  double fc_new = (K1(idx ^ 1, idx ^ 1) + K2(idx ^ 1, idx ^ 1)) * 0.5;
  double cc[2][2];
  for (int k = 0; k < 2; k++) {
    const kvfe_camera_params& cam = k == 0 ? L : Rc;
    const UndistortCtx u = make_undistort_ctx(cam, nullptr, nullptr);
    // cvProjectPoints2 converts the 3x3 rotation to a vector and back
    const M3 Rk = rodrigues(rodrigues_inv(k == 0 ? R1 : R2));
    double sx = 0, sy = 0;
    for (int i = 0; i < 4; i++) {
      float px = (float)((i % 2) * (nx - 1)), py = (float)((i < 2 ? 0 : 1) * (ny - 1));
      float ux, uy;
      undistort_point(u, px, py, &ux, &uy);
      const double X = ux, Y = uy, Z = 1.0f;
      double x = Rk(0, 0) * X + Rk(0, 1) * Y + Rk(0, 2) * Z + 0;
      double y = Rk(1, 0) * X + Rk(1, 1) * Y + Rk(1, 2) * Z + 0;
      double z = Rk(2, 0) * X + Rk(2, 1) * Y + Rk(2, 2) * Z + 0;
      z = z ? 1. / z : 1;
      x *= z;
      y *= z;
      sx += (float)(x * fc_new + 0.0);
      sy += (float)(y * fc_new + 0.0);
    }
    cc[k][0] = (nx - 1) / 2 - sx / 4;
    cc[k][1] = (ny - 1) / 2 - sy / 4;
  }
  // CALIB_ZERO_DISPARITY
  cc[0][0] = cc[1][0] = (cc[0][0] + cc[1][0]) * 0.5;
  cc[0][1] = cc[1][1] = (cc[0][1] + cc[1][1]) * 0.5;

  double P1[12] = {0}, P2[12] = {0};
  P1[0] = P1[5] = fc_new;
  P1[2] = cc[0][0];
  P1[6] = cc[0][1];
  P1[10] = 1;
  std::memcpy(P2, P1, sizeof(P1));
  P2[2] = cc[1][0];
  P2[6] = cc[1][1];
  P2[idx * 4 + 3] = t.v[idx] * fc_new;

  RectF in1, in2, out1, out2;
  valid_rectangles(L, R1.m, P1, &in1, &out1);
  valid_rectangles(Rc, R2.m, P2, &in2, &out2);

  // alpha = 0 (StereoCamera.cpp:326): scale so that only valid pixels remain
  const double W = nx, H = ny;
  const double cx1_0 = cc[0][0], cy1_0 = cc[0][1], cx2_0 = cc[1][0], cy2_0 = cc[1][1];
  const double cx1 = W * cx1_0 / nx, cy1 = H * cy1_0 / ny, cx2 = W * cx2_0 / nx,
               cy2 = H * cy2_0 / ny;
  double s0 = std::max(std::max(std::max(cx1 / (cx1_0 - in1.x), cy1 / (cy1_0 - in1.y)),
                                (W - cx1) / (in1.x + in1.w - cx1_0)),
                       (H - cy1) / (in1.y + in1.h - cy1_0));
  s0 = std::max(std::max(std::max(std::max(cx2 / (cx2_0 - in2.x), cy2 / (cy2_0 - in2.y)),
                                  (W - cx2) / (in2.x + in2.w - cx2_0)),
                         (H - cy2) / (in2.y + in2.h - cy2_0)),
                s0);
  double s1 = std::min(std::min(std::min(cx1 / (cx1_0 - out1.x), cy1 / (cy1_0 - out1.y)),
                                (W - cx1) / (out1.x + out1.w - cx1_0)),
                       (H - cy1) / (out1.y + out1.h - cy1_0));
  s1 = std::min(std::min(std::min(std::min(cx2 / (cx2_0 - out2.x), cy2 / (cy2_0 - out2.y)),
                                  (W - cx2) / (out2.x + out2.w - cx2_0)),
                         (H - cy2) / (out2.y + out2.h - cy2_0)),
                s1);
  const double alpha = 0.0;
  const double s = s0 * (1 - alpha) + s1 * alpha;
  fc_new *= s;
  P1[0] = P1[5] = fc_new;
  P1[2] = cx1;
  P1[6] = cy1;
  P2[0] = P2[5] = fc_new;
  P2[2] = cx2;
  P2[6] = cy2;
  P2[idx * 4 + 3] = s * P2[idx * 4 + 3];

  out->roi1[0] = iceil((in1.x - cx1_0) * s + cx1);
  out->roi1[1] = iceil((in1.y - cy1_0) * s + cy1);
  out->roi1[2] = ifloor(in1.w * s);
  out->roi1[3] = ifloor(in1.h * s);
  clip_roi(out->roi1, L.width, L.height);
  out->roi2[0] = iceil((in2.x - cx2_0) * s + cx2);
  out->roi2[1] = iceil((in2.y - cy2_0) * s + cy2);
  out->roi2[2] = ifloor(in2.w * s);
  out->roi2[3] = ifloor(in2.h * s);
  clip_roi(out->roi2, L.width, L.height);

  std::memcpy(out->R1, R1.m, sizeof(out->R1));
  std::memcpy(out->R2, R2.m, sizeof(out->R2));
  std::memcpy(out->P1, P1, sizeof(P1));
  std::memcpy(out->P2, P2, sizeof(P2));
  const double q[16] = {1, 0, 0, -cx1, 0, 1, 0, -cy1, 0, 0, 0, fc_new, 0, 0, -1. / t.v[idx],
                        (idx == 0 ? cx1 - cx2 : cy1 - cy2) / t.v[idx]};
  std::memcpy(out->Q, q, sizeof(q));
  if (out->Q[14] == 0.0) return KVFE_ERR_INVALID_ARG;  // CHECK_NE(Q(3,2), 0)
  out->baseline = 1.0 / out->Q[14];                    // StereoCamera.cpp:70-72
  if (!(out->baseline > 0.0)) return KVFE_ERR_INVALID_ARG;
  return KVFE_OK;
}

kvfe_status init_undistort_rectify_map(const kvfe_camera_params& cam, const double R[9],
                                       const double P[12], float* map_x, float* map_y) {
  if (cam.distortion_model == KVFE_DIST_EQUIDISTANT) {
    // cv::fisheye::initUndistortRectifyMap (calib3d/src/fisheye.cpp); the reference's inverse is
    // cv::invert(DECOMP_SVD), here the closed-form 3x3 inverse (agrees to rounding)
    M3 Rm;
    std::memcpy(Rm.m, R, sizeof(Rm.m));
    const M3 PP{{P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]}};
    const M3 iR = inv3(mul(PP, Rm));
    const double* kk = cam.distortion;
    const double f0 = cam.intrinsics[0], f1 = cam.intrinsics[1], c0 = cam.intrinsics[2], c1 = cam.intrinsics[3];
    for (int i = 0; i < cam.height; ++i) {
      float* m1f = map_x + (size_t)i * cam.width;
      float* m2f = map_y + (size_t)i * cam.width;
      double _x = i * iR(0, 1) + iR(0, 2), _y = i * iR(1, 1) + iR(1, 2), _w = i * iR(2, 1) + iR(2, 2);
      for (int j = 0; j < cam.width; ++j) {
        const double x = _x / _w, y = _y / _w;
        const double r = std::sqrt(x * x + y * y);
        const double theta = std::atan(r);
        const double theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2,
                     theta8 = theta4 * theta4;
        const double theta_d = theta * (1 + kk[0] * theta2 + kk[1] * theta4 + kk[2] * theta6 + kk[3] * theta8);
        const double scale = (r == 0) ? 1.0 : theta_d / r;
        const double u = f0 * x * scale + c0;
        const double v = f1 * y * scale + c1;
        m1f[j] = (float)u;
        m2f[j] = (float)v;
        _x += iR(0, 0);
        _y += iR(1, 0);
        _w += iR(2, 0);
      }
    }
    return KVFE_OK;
  }
  double k[8];
  for (int i = 0; i < 8; i++)
    k[i] = (cam.distortion_model == KVFE_DIST_RADTAN && i < cam.n_distortion) ? cam.distortion[i] : 0;
  const double k1 = k[0], k2 = k[1], p1 = k[2], p2 = k[3], k3 = k[4], k4 = k[5], k5 = k[6],
               k6 = k[7];
  M3 Rm;
  std::memcpy(Rm.m, R, sizeof(Rm.m));
  const M3 Ar{{P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]}};
  const M3 iR = inv3(mul(Ar, Rm));
  const double* ir = iR.m;
  const double u0 = cam.intrinsics[2], v0 = cam.intrinsics[3], fx = cam.intrinsics[0],
               fy = cam.intrinsics[1];
  for (int i = 0; i < cam.height; i++) {
    float* m1 = map_x + (size_t)i * cam.width;
    float* m2 = map_y + (size_t)i * cam.width;
    double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
    for (int j = 0; j < cam.width; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
      const double w = 1. / _w, x = _x * w, y = _y * w;
      const double x2 = x * x, y2 = y * y;
      const double r2 = x2 + y2, _2xy = 2 * x * y;
      const double kr =
          (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
      const double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + 0 * r2 + 0 * r2 * r2);
      const double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + 0 * r2 + 0 * r2 * r2);
      // tilt matrix = I:  vecTilt = (xd, yd, 1)
      const double t0 = 1 * xd + 0 * yd + 0 * 1, t1 = 0 * xd + 1 * yd + 0 * 1,
                   t2 = 0 * xd + 0 * yd + 1 * 1;
      const double invProj = t2 ? 1. / t2 : 1;
      m1[j] = (float)(fx * invProj * t0 + u0);
      m2[j] = (float)(fy * invProj * t1 + v0);
    }
  }
  return KVFE_OK;
}

std::vector<int> circle_half_widths(int radius) {
  std::vector<int> hw(std::max(radius, 0) + 1, -1);
  // Bresenham/midpoint iteration of cv::Circle(fill): rows cy +- dy get half-width dx,
  // rows cy +- dx get half-width dy.
  int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
  while (dx >= dy) {
    hw[dy] = std::max(hw[dy], dx);
    hw[dx] = std::max(hw[dx], dy);
    dy++;
    err += plus;
    plus += 2;
    const int mask = (err <= 0) - 1;
    err -= minus & mask;
    dx += mask;
    minus -= mask & 2;
  }
  return hw;
}

namespace {
void introsort_equal(uint16_t* first, uint16_t* last) {
  // libstdc++ std::__introsort_loop specialised for a comparator that is always false
  // (all keys equal): median-of-three moves `mid` to `first`, the unguarded partition
  // swaps symmetric pairs of [first+1, last) and cuts in the middle.
  while (last - first > 16) {
    uint16_t* mid = first + (last - first) / 2;
    std::swap(*first, *mid);
    uint16_t* lo = first + 1;
    uint16_t* hi = last;
    for (;;) {
      --hi;
      if (!(lo < hi)) break;
      std::swap(*lo, *hi);
      ++lo;
    }
    introsort_equal(lo, last);
    last = lo;
  }
}
}  // namespace

void sortidx_permutation(int n, int policy, uint16_t* out) {
  for (int i = 0; i < n; i++) out[i] = (uint16_t)i;
  if (policy == KVFE_SORTIDX_STABLE || n <= 1) return;
  introsort_equal(out, out + n);
  std::reverse(out, out + n);
}

}  // namespace kvfe
