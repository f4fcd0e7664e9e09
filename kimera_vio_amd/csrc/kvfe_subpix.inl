// cv::cornerSubPix for one corner by one wavefront (included inside namespace kvfe by
// k_detect.hip and k_stereo.hip).  reference call sites: FeatureDetector.cpp:288-292,
// StereoMatcher.cpp:404-413.  The five normal-equation sums are accumulated in float64 in
// OpenCV's row-major order (lanes 0-4 each own one accumulator) => bit-identical to the CPU path.
__device__ __forceinline__ int cv_floorf(float v) { return (int)floorf(v); }

// cv::getRectSubPix(u8 -> f32) of a (n x n) window centred at c, into LDS `dst` (row stride n)
static __device__ void rect_subpix_8u32f(const unsigned char* __restrict__ img, size_t step, int W, int H,
                                  float cx, float cy, int n, float* dst, int lane) {
  const float ccx = cx - (n - 1) * 0.5f, ccy = cy - (n - 1) * 0.5f;
  const int ipx = cv_floorf(ccx), ipy = cv_floorf(ccy);
  if (0 <= ipx && ipx + n < W && 0 <= ipy && ipy + n < H) {
    float a = ccx - ipx;
    const float b = ccy - ipy;
    a = fmaxf(a, 0.0001f);
    const float a12 = a * (1.f - b), a22 = a * b, b1 = 1.f - b, b2 = b;
    const double sc = (1. - a) / a;
    const unsigned char* S = img + (size_t)ipy * step + ipx;
    for (int e = lane; e < n * n; e += 64) {
      const int i = e / n, j = e - i * n;
      const unsigned char* R = S + (size_t)i * step;
      float prev;
      if (j == 0) {
        prev = (1 - a) * (b1 * R[0] + b2 * R[step]);
      } else {
        const float tp = a12 * R[j] + a22 * R[j + step];
        prev = (float)(tp * sc);
      }
      const float t = a12 * R[j + 1] + a22 * R[j + 1 + step];
      dst[e] = prev + t;
    }
  } else {
    const float a = ccx - ipx, b = ccy - ipy;
    const float a11 = (1.f - a) * (1.f - b), a12 = a * (1.f - b), a21 = (1.f - a) * b, a22 = a * b;
    const float b1 = 1.f - b, b2 = b;
    int rx, rw, ry, rh;
    if (ipx >= 0)
      rx = 0;
    else {
      rx = -ipx;
      if (rx > n) rx = n;
    }
    if (ipx < W - n)
      rw = n;
    else {
      rw = W - ipx - 1;
      if (rw < 0) rw = 0;
    }
    ry = ipy >= 0 ? 0 : -ipy;
    if (ipy < H - n)
      rh = n;
    else {
      rh = H - ipy - 1;
      if (rh < 0) rh = 0;
    }
    const int row0 = ipy >= 0 ? ipy : 0;
    for (int e = lane; e < n * n; e += 64) {
      const int i = e / n, j = e - i * n;
      // running source row after i window rows: advances once per row with r.y <= row < r.height
      int adv = min(i, rh) - min(i, ry);
      if (adv < 0) adv = 0;
      int row = row0 + adv;
      int row2 = (i < ry || i >= rh) ? row : row + 1;
      row = min(max(row, 0), H - 1);
      row2 = min(max(row2, 0), H - 1);
      const unsigned char* S = img + (size_t)row * step;
      const unsigned char* S2 = img + (size_t)row2 * step;
      auto col = [&](int jj) {
        const int c = ipx + jj;
        return c < 0 ? 0 : (c > W - 1 ? W - 1 : c);
      };
      float v;
      if (j < rx) {
        v = S[col(rx)] * b1 + S2[col(rx)] * b2;
      } else if (j >= rw) {
        v = S[col(rw)] * b1 + S2[col(rw)] * b2;
      } else {
        v = S[col(j)] * a11 + S[col(j + 1)] * a12 + S2[col(j)] * a21 + S2[col(j + 1)] * a22;
      }
      dst[e] = v;
    }
  }
}

// refines one corner; all 64 lanes of the wave call it with identical arguments.
// LDS: patch (2w+3)^2 floats, terms 5*(2w+1)^2 doubles.
static __device__ float2 corner_subpix_wave(const unsigned char* __restrict__ img, size_t step, int W,
                                     int H, float2 cT, int win, int max_iters, double eps2,
                                     const float* __restrict__ mask, float* patch, double* terms,
                                     int lane) {
  const int ww = 2 * win + 1, pw = ww + 2, nt = ww * ww;
  float2 cI = cT;
  int iter = 0;
  double err = 0;
  do {
    rect_subpix_8u32f(img, step, W, H, cI.x, cI.y, pw, patch, lane);
    __syncthreads();
    for (int k = lane; k < nt; k += 64) {
      const int i = k / ww, j = k - i * ww;
      const float* sp = patch + (i + 1) * pw + (j + 1);
      const double m = (double)mask[k];
      const double tgx = (double)(sp[1] - sp[-1]);
      const double tgy = (double)(sp[pw] - sp[-pw]);
      const double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
      const double px = (double)(j - win), py = (double)(i - win);
      terms[k] = gxx;
      terms[nt + k] = gxy;
      terms[2 * nt + k] = gyy;
      terms[3 * nt + k] = gxx * px + gxy * py;
      terms[4 * nt + k] = gxy * px + gyy * py;
    }
    __syncthreads();
    double acc = 0;
    if (lane < 5) {
      const double* t = terms + lane * nt;
      for (int k = 0; k < nt; k++) acc += t[k];
    }
    const double a = __shfl(acc, 0), b = __shfl(acc, 1), c = __shfl(acc, 2), bb1 = __shfl(acc, 3),
                 bb2 = __shfl(acc, 4);
    __syncthreads();
    const double det = a * c - b * b;
    if (fabs(det) <= 2.220446049250313e-16 * 2.220446049250313e-16) break;
    const double scale = 1.0 / det;
    float2 cI2;
    cI2.x = (float)(cI.x + c * scale * bb1 - b * scale * bb2);
    cI2.y = (float)(cI.y - b * scale * bb1 + a * scale * bb2);
    err = (double)((cI2.x - cI.x) * (cI2.x - cI.x) + (cI2.y - cI.y) * (cI2.y - cI.y));
    cI = cI2;
    if (cI.x < 0 || cI.x >= W || cI.y < 0 || cI.y >= H) break;
  } while (++iter < max_iters && err > eps2);
  if (fabsf(cI.x - cT.x) > win || fabsf(cI.y - cT.y) > win) cI = cT;
  return cI;
}

