// cv::undistortPoints for one pixel on the device, float64, same operation order as the host
// implementation in host_calib.cpp (reference call sites: UndistorterRectifier.cpp:42-47 for
// rectified keypoints, :83-88 for bearing vectors).  Included inside namespace kvfe.

__device__ __forceinline__ void undistort_point_d(const UndistortDev& u, float x_in, float y_in,
                                                  double* xo, double* yo) {
  const double* k = u.k;  // k1 k2 p1 p2 k3 k4 k5 k6
  if (u.has_dist == 2) {
    // cv::fisheye::undistortPoints (equidistant model): Newton iterations on theta, then tan.
    // tan() is the one libm function on this path: its last bit may differ from the host's.
    const double pw0 = ((double)x_in - u.cx) / u.fx, pw1 = ((double)y_in - u.cy) / u.fy;
    double scale = 1.0;
    double theta_d = sqrt(pw0 * pw0 + pw1 * pw1);
    theta_d = fmin(fmax(-3.14159265358979323846 / 2., theta_d), 3.14159265358979323846 / 2.);
    if (theta_d > 1e-8) {
      double theta = theta_d;
      for (int j = 0; j < 10; j++) {
        const double theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2,
                     theta8 = theta6 * theta2;
        const double k0_theta2 = k[0] * theta2, k1_theta4 = k[1] * theta4, k2_theta6 = k[2] * theta6,
                     k3_theta8 = k[3] * theta8;
        const double theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d) /
                                 (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
        theta = theta - theta_fix;
        if (fabs(theta_fix) < 1e-8) break;
      }
      scale = tan(theta) / theta_d;
    }
    const double pu0 = pw0 * scale, pu1 = pw1 * scale;
    const double* RR = u.RR;
    const double pr0 = RR[0] * pu0 + RR[1] * pu1 + RR[2] * 1.0;
    const double pr1 = RR[3] * pu0 + RR[4] * pu1 + RR[5] * 1.0;
    const double pr2 = RR[6] * pu0 + RR[7] * pu1 + RR[8] * 1.0;
    *xo = pr0 / pr2;
    *yo = pr1 / pr2;
    return;
  }
  double x = (double)x_in, y = (double)y_in;
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
  const double* RR = u.RR;
  const double xx = RR[0] * x + RR[1] * y + RR[2];
  const double yy = RR[3] * x + RR[4] * y + RR[5];
  const double ww = 1. / (RR[6] * x + RR[7] * y + RR[8]);
  *xo = xx * ww;
  *yo = yy * ww;
}

__device__ __forceinline__ void undistort_point_dev(const UndistortDev& u, float x_in, float y_in,
                                                    float* xo, float* yo) {
  double x, y;
  undistort_point_d(u, x_in, y_in, &x, &y);
  *xo = (float)x;
  *yo = (float)y;
}

// UndistorterRectifier::GetBearingVector: undistort (R, no P) -> float -> (x, y, 1) normalised
__device__ __forceinline__ void bearing_vector(const UndistortDev& u, float x_in, float y_in,
                                               double* v) {
  float fx, fy;
  undistort_point_dev(u, x_in, y_in, &fx, &fy);
  const double x = (double)fx, y = (double)fy, z = 1.0;
  const double n = sqrt(x * x + (y * y + z * z));
  v[0] = x / n;
  v[1] = y / n;
  v[2] = z / n;
}
