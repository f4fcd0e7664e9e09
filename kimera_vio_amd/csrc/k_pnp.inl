// Tracker::pnp (src/frontend/Tracker.cpp:1122-1288) for pnp_algorithm_ = EPNP (3) and KneipP3P (1):
//   runRansac<ProblemPnP> (Tracker.h:247-296) = opengv::sac::Ransac<AbsolutePoseSacProblem(adapter, ALG)>::computeModel
//   EPNP : sample size 6, model = absolute_pose::epnp(adapter, sample)
//   KNEIP: sample size 4, model = the solution of absolute_pose::p3p_kneip(first three) the fourth one reprojects best
//   model = world_T_camera, distance = 1 - f . normalize(R^T (p - t))
// + the status rule of VisionImuFrontend::outlierRejectionPnP (VisionImuFrontend.cpp:146-173).
//
// (included at the end of k_ransac.hip: shares its scan / reduction helpers, rs_svd3 and the sampler table)
//
// One 256-thread workgroup per problem.  The sample stream does not depend on the results, so EP_BATCH hypotheses are
// drawn ahead, solved in parallel -- one lane per hypothesis, spread over the four wavefronts, every work array in an
// LDS slot because a private array with run-time indices would live in scratch memory --, scored by the whole block
// over all correspondences, and then replayed in order through the adaptive stopping rule of Ransac::computeModel.
// EPnP itself is, operation for operation, the oracle's (oracle/opengv_epnp.inl): PCA control points, barycentric
// coordinates, M^T M accumulated row by row, cyclic Jacobi of the 12x12 matrix, L_6x10 / rho, the three beta
// approximations, five Gauss-Newton steps with the authors' Householder solver, Horn / Arun alignment with rs_svd3,
// reprojection error, best of N = 1..3.  All float64, no FMA contraction.

constexpr int EP_BATCH = 8;
constexpr int EP_N = 6;   // AbsolutePoseSacProblem::getSampleSize() for EPNP

struct EpSlot {
  double a[144];      // M^T M, destroyed by the Jacobi iteration
  double v[144];      // eigenvectors (columns)
  double d[12], b[12], z[12];
  double l[60], rho[6];
  double pws[3 * EP_N], us[2 * EP_N], alphas[4 * EP_N], pcs[3 * EP_N];
  double cws[12], ccs[12];
  double qa[30], qb[6], qx[5], qa1[8], qa2[8];   // qr_solve
  double c3[9], v3[9], d3[3], b3[3], z3[3];      // 3x3 covariance of the world points
  double betas[4];
  double Rs[3][9], ts[3][3], rep[3];
  int signs[EP_N];
};

__device__ __forceinline__ double ep_d3(const double* x, const double* y) { return (x[0] * y[0] + x[1] * y[1]) + x[2] * y[2]; }

// cyclic Jacobi for a symmetric n x n matrix; v: columns = eigenvectors, d: eigenvalues, sorted descending
__device__ void ep_jacobi(double* a, int n, double* v, double* d, double* b, double* z) {
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) v[i * n + j] = (i == j) ? 1.0 : 0.0;
    b[i] = d[i] = a[i * n + i];
    z[i] = 0.0;
  }
  for (int it = 1; it <= 50; it++) {
    double sm = 0.0;
    for (int p = 0; p < n - 1; p++)
      for (int q = p + 1; q < n; q++) sm += fabs(a[p * n + q]);
    if (sm == 0.0) break;
    const double tresh = it < 4 ? 0.2 * sm / (double)(n * n) : 0.0;
    for (int p = 0; p < n - 1; p++)
      for (int q = p + 1; q < n; q++) {
        const double g = 100.0 * fabs(a[p * n + q]);
        if (it > 4 && fabs(d[p]) + g == fabs(d[p]) && fabs(d[q]) + g == fabs(d[q])) {
          a[p * n + q] = 0.0;
        } else if (fabs(a[p * n + q]) > tresh) {
          double h = d[q] - d[p];
          double t;
          if (fabs(h) + g == fabs(h)) {
            t = a[p * n + q] / h;
          } else {
            const double theta = 0.5 * h / a[p * n + q];
            t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
            if (theta < 0.0) t = -t;
          }
          const double c = 1.0 / sqrt(1.0 + t * t);
          const double s = t * c;
          const double tau = s / (1.0 + c);
          h = t * a[p * n + q];
          z[p] -= h;
          z[q] += h;
          d[p] -= h;
          d[q] += h;
          a[p * n + q] = 0.0;
#define KVFE_EP_ROT(m, i1, j1, i2, j2)                         \
  {                                                            \
    const double gg = m[(i1) * n + (j1)], hh = m[(i2) * n + (j2)]; \
    m[(i1) * n + (j1)] = gg - s * (hh + gg * tau);             \
    m[(i2) * n + (j2)] = hh + s * (gg - hh * tau);             \
  }
          for (int j = 0; j < p; j++) KVFE_EP_ROT(a, j, p, j, q)
          for (int j = p + 1; j < q; j++) KVFE_EP_ROT(a, p, j, j, q)
          for (int j = q + 1; j < n; j++) KVFE_EP_ROT(a, p, j, q, j)
          for (int j = 0; j < n; j++) KVFE_EP_ROT(v, j, p, j, q)
#undef KVFE_EP_ROT
        }
      }
    for (int i = 0; i < n; i++) {
      b[i] += z[i];
      d[i] = b[i];
      z[i] = 0.0;
    }
  }
  for (int i = 0; i < n - 1; i++) {
    int k = i;
    double p = d[i];
    for (int j = i + 1; j < n; j++)
      if (d[j] > p) {
        k = j;
        p = d[j];
      }
    if (k != i) {
      d[k] = d[i];
      d[i] = p;
      for (int j = 0; j < n; j++) {
        const double t = v[j * n + i];
        v[j * n + i] = v[j * n + k];
        v[j * n + k] = t;
      }
    }
  }
}

// Householder least squares of the EPnP reference code; A: nr x nc row-major (destroyed), b: nr (destroyed)
__device__ bool ep_qr_solve(double* A, double* b, double* X, double* A1, double* A2, int nr, int nc) {
  for (int k = 0; k < nc; k++) {
    const int kk = k * nc + k;
    double eta = fabs(A[kk]);
    for (int i = k + 1; i < nr; i++) {   // (rows k .. nr-2, as the published code reads them)
      const double elt = fabs(A[kk + (i - k - 1) * nc]);
      if (eta < elt) eta = elt;
    }
    if (eta == 0) {
      A1[k] = A2[k] = 0.0;
      return false;
    }
    double sum = 0.0;
    const double inv_eta = 1. / eta;
    for (int i = k; i < nr; i++) {
      const int ik = i * nc + k;
      A[ik] *= inv_eta;
      sum += A[ik] * A[ik];
    }
    double sigma = sqrt(sum);
    if (A[kk] < 0) sigma = -sigma;
    A[kk] += sigma;
    A1[k] = sigma * A[kk];
    A2[k] = -eta * sigma;
    for (int j = k + 1; j < nc; j++) {
      double sm = 0;
      for (int i = k; i < nr; i++) sm += A[i * nc + k] * A[i * nc + j];
      const double tau = sm / A1[k];
      for (int i = k; i < nr; i++) A[i * nc + j] -= tau * A[i * nc + k];
    }
  }
  for (int j = 0; j < nc; j++) {   // b <- Qt b
    double tau = 0;
    for (int i = j; i < nr; i++) tau += A[i * nc + j] * b[i];
    tau /= A1[j];
    for (int i = j; i < nr; i++) b[i] -= tau * A[i * nc + j];
  }
  X[nc - 1] = b[nc - 1] / A2[nc - 1];   // X = R-1 b
  for (int i = nc - 2; i >= 0; i--) {
    double sum = 0;
    for (int j = i + 1; j < nc; j++) sum += A[i * nc + j] * X[j];
    X[i] = (b[i] - sum) / A2[i];
  }
  return true;
}

__device__ void ep_inv3(const double* m, double* r) {   // Eigen::Matrix3d::inverse() (cofactors)
  const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
  const double det = (m[0] * c00 + m[1] * c01) + m[2] * c02;
  const double id = 1.0 / det;
  r[0] = c00 * id;
  r[1] = (m[2] * m[7] - m[1] * m[8]) * id;
  r[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  r[3] = c01 * id;
  r[4] = (m[0] * m[8] - m[2] * m[6]) * id;
  r[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  r[6] = c02 * id;
  r[7] = (m[1] * m[6] - m[0] * m[7]) * id;
  r[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

// least squares of L(:, cols) x = rho
__device__ void ep_solve_subset(EpSlot& W, const int* cols, int nc) {
  for (int i = 0; i < 6; i++) {
    for (int c = 0; c < nc; c++) W.qa[i * nc + c] = W.l[10 * i + cols[c]];
    W.qb[i] = W.rho[i];
  }
  if (!ep_qr_solve(W.qa, W.qb, W.qx, W.qa1, W.qa2, 6, nc))
    for (int c = 0; c < nc; c++) W.qx[c] = 0.0;
}

__device__ void ep_gauss_newton(EpSlot& W) {
  double* b = W.betas;
  for (int k = 0; k < 5; k++) {
    for (int i = 0; i < 6; i++) {
      const double* rowL = W.l + i * 10;
      double* rowA = W.qa + i * 4;
      rowA[0] = 2 * rowL[0] * b[0] + rowL[1] * b[1] + rowL[3] * b[2] + rowL[6] * b[3];
      rowA[1] = rowL[1] * b[0] + 2 * rowL[2] * b[1] + rowL[4] * b[2] + rowL[7] * b[3];
      rowA[2] = rowL[3] * b[0] + rowL[4] * b[1] + 2 * rowL[5] * b[2] + rowL[8] * b[3];
      rowA[3] = rowL[6] * b[0] + rowL[7] * b[1] + rowL[8] * b[2] + 2 * rowL[9] * b[3];
      W.qb[i] = W.rho[i] - (rowL[0] * b[0] * b[0] + rowL[1] * b[0] * b[1] + rowL[2] * b[1] * b[1] +
                            rowL[3] * b[0] * b[2] + rowL[4] * b[1] * b[2] + rowL[5] * b[2] * b[2] +
                            rowL[6] * b[0] * b[3] + rowL[7] * b[1] * b[3] + rowL[8] * b[2] * b[3] +
                            rowL[9] * b[3] * b[3]);
    }
    if (!ep_qr_solve(W.qa, W.qb, W.qx, W.qa1, W.qa2, 6, 4)) return;
    for (int i = 0; i < 4; i++) b[i] += W.qx[i];
  }
}

// compute_ccs, compute_pcs, solve_for_sign, estimate_R_and_t, reprojection_error for the betas of the slot
__device__ double ep_compute_R_and_t(EpSlot& W, double* R /*[9]*/, double* t /*[3]*/) {
  constexpr int n = EP_N;
  for (int i = 0; i < 12; i++) W.ccs[i] = 0.0;
  for (int i = 0; i < 4; i++) {
    // row 11 - i of Ut = column 11 - i of v
    for (int j = 0; j < 4; j++)
      for (int k = 0; k < 3; k++) W.ccs[3 * j + k] += W.betas[i] * W.v[(3 * j + k) * 12 + (11 - i)];
  }
  for (int i = 0; i < n; i++) {
    const double* al = W.alphas + 4 * i;
    for (int j = 0; j < 3; j++)
      W.pcs[3 * i + j] = al[0] * W.ccs[j] + al[1] * W.ccs[3 + j] + al[2] * W.ccs[6 + j] + al[3] * W.ccs[9 + j];
  }
  if ((W.pcs[2] < 0.0 && W.signs[0] > 0) || (W.pcs[2] > 0.0 && W.signs[0] < 0)) {
    for (int i = 0; i < 12; i++) W.ccs[i] = -W.ccs[i];
    for (int i = 0; i < 3 * n; i++) W.pcs[i] = -W.pcs[i];
  }
  double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
  for (int i = 0; i < n; i++)
    for (int j = 0; j < 3; j++) {
      pc0[j] += W.pcs[3 * i + j];
      pw0[j] += W.pws[3 * i + j];
    }
  for (int j = 0; j < 3; j++) {
    pc0[j] /= n;
    pw0[j] /= n;
  }
  double abt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; i++)
    for (int j = 0; j < 3; j++) {
      abt[3 * j] += (W.pcs[3 * i + j] - pc0[j]) * (W.pws[3 * i] - pw0[0]);
      abt[3 * j + 1] += (W.pcs[3 * i + j] - pc0[j]) * (W.pws[3 * i + 1] - pw0[1]);
      abt[3 * j + 2] += (W.pcs[3 * i + j] - pc0[j]) * (W.pws[3 * i + 2] - pw0[2]);
    }
  double U[9], S[3], V[9];
  rs_svd3(abt, U, S, V);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R[3 * i + j] = ep_d3(U + 3 * i, V + 3 * j);
  const double det = R[0] * R[4] * R[8] + R[1] * R[5] * R[6] + R[2] * R[3] * R[7] - R[2] * R[4] * R[6] -
                     R[1] * R[3] * R[8] - R[0] * R[5] * R[7];
  if (det < 0) {
    R[6] = -R[6];
    R[7] = -R[7];
    R[8] = -R[8];
  }
  t[0] = pc0[0] - ep_d3(R, pw0);
  t[1] = pc0[1] - ep_d3(R + 3, pw0);
  t[2] = pc0[2] - ep_d3(R + 6, pw0);
  double sum2 = 0.0;
  for (int i = 0; i < n; i++) {
    const double* pw = W.pws + 3 * i;
    const double Xc = ep_d3(R, pw) + t[0];
    const double Yc = ep_d3(R + 3, pw) + t[1];
    const double inv_Zc = 1.0 / (ep_d3(R + 6, pw) + t[2]);
    const double ue = Xc * inv_Zc, ve = Yc * inv_Zc;
    const double u = W.us[2 * i], v = W.us[2 * i + 1];
    sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
  }
  return sum2 / n;
}

// absolute_pose::epnp over the six correspondences sel[0..6): model = world_T_camera [R^T | -R^T t] (3x4 row-major)
__device__ void ep_solve(const double* f, const double* p, const int* sel, EpSlot& W, double* model) {
  constexpr int n = EP_N;
  for (int i = 0; i < n; i++) {
    const double* pi = p + 3 * (size_t)sel[i];
    const double* fi = f + 3 * (size_t)sel[i];
    W.pws[3 * i] = pi[0];
    W.pws[3 * i + 1] = pi[1];
    W.pws[3 * i + 2] = pi[2];
    W.us[2 * i] = fi[0] / fi[2];
    W.us[2 * i + 1] = fi[1] / fi[2];
    W.signs[i] = fi[2] > 0.0 ? 1 : -1;
  }
  // choose_control_points
  W.cws[0] = W.cws[1] = W.cws[2] = 0;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < 3; j++) W.cws[j] += W.pws[3 * i + j];
  for (int j = 0; j < 3; j++) W.cws[j] /= n;
  for (int i = 0; i < 9; i++) W.c3[i] = 0.0;
  for (int i = 0; i < n; i++) {
    double dlt[3];
    for (int j = 0; j < 3; j++) dlt[j] = W.pws[3 * i + j] - W.cws[j];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) W.c3[r * 3 + c] += dlt[r] * dlt[c];
  }
  ep_jacobi(W.c3, 3, W.v3, W.d3, W.b3, W.z3);
  for (int i = 1; i < 4; i++) {
    const double k = sqrt(W.d3[i - 1] / n);
    for (int j = 0; j < 3; j++) W.cws[3 * i + j] = W.cws[j] + k * W.v3[j * 3 + (i - 1)];
  }
  // compute_barycentric_coordinates
  {
    double cc[9], ci[9];
    for (int i = 0; i < 3; i++)
      for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = W.cws[3 * j + i] - W.cws[i];
    ep_inv3(cc, ci);
    for (int i = 0; i < n; i++) {
      const double* pi = W.pws + 3 * i;
      double* al = W.alphas + 4 * i;
      for (int j = 0; j < 3; j++)
        al[1 + j] = ci[3 * j] * (pi[0] - W.cws[0]) + ci[3 * j + 1] * (pi[1] - W.cws[1]) +
                    ci[3 * j + 2] * (pi[2] - W.cws[2]);
      al[0] = 1.0 - al[1] - al[2] - al[3];
    }
  }
  // M^T M, row by row
  for (int i = 0; i < 144; i++) W.a[i] = 0.0;
  for (int i = 0; i < n; i++) {
    const double* as = W.alphas + 4 * i;
    double* m1 = W.qa;        // (the solver's scratch is free here)
    double* m2 = W.qa + 12;
    for (int j = 0; j < 4; j++) {
      m1[3 * j] = as[j];
      m1[3 * j + 1] = 0.0;
      m1[3 * j + 2] = as[j] * (0.0 - W.us[2 * i]);
      m2[3 * j] = 0.0;
      m2[3 * j + 1] = as[j];
      m2[3 * j + 2] = as[j] * (0.0 - W.us[2 * i + 1]);
    }
    for (int r = 0; r < 12; r++)
      for (int c = 0; c < 12; c++) {
        W.a[r * 12 + c] += m1[r] * m1[c];
        W.a[r * 12 + c] += m2[r] * m2[c];
      }
  }
  ep_jacobi(W.a, 12, W.v, W.d, W.b, W.z);
  // compute_L_6x10: v[k] = row 11 - k of Ut = column 11 - k of W.v
  for (int i = 0; i < 6; i++) {
    // pair (a, b) number i of (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
    const int pa = i < 3 ? 0 : (i < 5 ? 1 : 2);
    const int pb = i < 3 ? i + 1 : (i < 5 ? i - 1 : 3);
    double dv[4][3];
    for (int k = 0; k < 4; k++)
      for (int c = 0; c < 3; c++)
        dv[k][c] = W.v[(3 * pa + c) * 12 + (11 - k)] - W.v[(3 * pb + c) * 12 + (11 - k)];
    double* row = W.l + 10 * i;
    row[0] = ep_d3(dv[0], dv[0]);
    row[1] = 2.0 * ep_d3(dv[0], dv[1]);
    row[2] = ep_d3(dv[1], dv[1]);
    row[3] = 2.0 * ep_d3(dv[0], dv[2]);
    row[4] = 2.0 * ep_d3(dv[1], dv[2]);
    row[5] = ep_d3(dv[2], dv[2]);
    row[6] = 2.0 * ep_d3(dv[0], dv[3]);
    row[7] = 2.0 * ep_d3(dv[1], dv[3]);
    row[8] = 2.0 * ep_d3(dv[2], dv[3]);
    row[9] = ep_d3(dv[3], dv[3]);
  }
  // compute_rho
  {
    auto dist2 = [&](int x, int y) {
      const double* p1 = W.cws + 3 * x;
      const double* p2 = W.cws + 3 * y;
      return (p1[0] - p2[0]) * (p1[0] - p2[0]) + (p1[1] - p2[1]) * (p1[1] - p2[1]) + (p1[2] - p2[2]) * (p1[2] - p2[2]);
    };
    W.rho[0] = dist2(0, 1);
    W.rho[1] = dist2(0, 2);
    W.rho[2] = dist2(0, 3);
    W.rho[3] = dist2(1, 2);
    W.rho[4] = dist2(1, 3);
    W.rho[5] = dist2(2, 3);
  }
  double* betas = W.betas;
  // N = 1: [B11 B12 B13 B14]
  {
    const int cols[4] = {0, 1, 3, 6};
    ep_solve_subset(W, cols, 4);
    const double* b4 = W.qx;
    if (b4[0] < 0) {
      betas[0] = sqrt(-b4[0]);
      betas[1] = -b4[1] / betas[0];
      betas[2] = -b4[2] / betas[0];
      betas[3] = -b4[3] / betas[0];
    } else {
      betas[0] = sqrt(b4[0]);
      betas[1] = b4[1] / betas[0];
      betas[2] = b4[2] / betas[0];
      betas[3] = b4[3] / betas[0];
    }
    ep_gauss_newton(W);
    W.rep[0] = ep_compute_R_and_t(W, W.Rs[0], W.ts[0]);
  }
  // N = 2: [B11 B12 B22]
  {
    const int cols[3] = {0, 1, 2};
    ep_solve_subset(W, cols, 3);
    const double* b3 = W.qx;
    if (b3[0] < 0) {
      betas[0] = sqrt(-b3[0]);
      betas[1] = (b3[2] < 0) ? sqrt(-b3[2]) : 0.0;
    } else {
      betas[0] = sqrt(b3[0]);
      betas[1] = (b3[2] > 0) ? sqrt(b3[2]) : 0.0;
    }
    if (b3[1] < 0) betas[0] = -betas[0];
    betas[2] = 0.0;
    betas[3] = 0.0;
    ep_gauss_newton(W);
    W.rep[1] = ep_compute_R_and_t(W, W.Rs[1], W.ts[1]);
  }
  // N = 3: [B11 B12 B22 B13 B23]
  {
    const int cols[5] = {0, 1, 2, 3, 4};
    ep_solve_subset(W, cols, 5);
    const double* b5 = W.qx;
    if (b5[0] < 0) {
      betas[0] = sqrt(-b5[0]);
      betas[1] = (b5[2] < 0) ? sqrt(-b5[2]) : 0.0;
    } else {
      betas[0] = sqrt(b5[0]);
      betas[1] = (b5[2] > 0) ? sqrt(b5[2]) : 0.0;
    }
    if (b5[1] < 0) betas[0] = -betas[0];
    betas[2] = b5[3] / betas[0];
    betas[3] = 0.0;
    ep_gauss_newton(W);
    W.rep[2] = ep_compute_R_and_t(W, W.Rs[2], W.ts[2]);
  }
  int N = 0;
  if (W.rep[1] < W.rep[0]) N = 1;
  if (W.rep[2] < W.rep[N]) N = 2;
  const double* R = W.Rs[N];
  const double* t = W.ts[N];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) model[r * 4 + c] = R[c * 3 + r];   // rotation.transposeInPlace()
  for (int r = 0; r < 3; r++)                                       // translation = -rotation * translation
    model[r * 4 + 3] = -((model[r * 4] * t[0] + model[r * 4 + 1] * t[1]) + model[r * 4 + 2] * t[2]);
}

// ---------------------------------------------------------------------------------------------------------------
// AbsolutePoseSacProblem(adapter, KNEIP) (pnp_algorithm: 1, params/KinectAzure): Kneip's P3P on three correspondences,
// the fourth one picks among the up to four solutions.  Operation for operation the oracle's (oracle/opengv_epnp.inl),
// including its closed-form quartic with complex square / cube roots built from +, -, *, / and sqrt only.
// ---------------------------------------------------------------------------------------------------------------
struct KnCx {
  double re, im;
};
__device__ __forceinline__ KnCx kn_cx(double a, double b = 0.0) { return KnCx{a, b}; }
__device__ __forceinline__ KnCx kn_add(KnCx a, KnCx b) { return KnCx{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ KnCx kn_sub(KnCx a, KnCx b) { return KnCx{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ KnCx kn_scale(KnCx a, double s) { return KnCx{a.re * s, a.im * s}; }
__device__ __forceinline__ KnCx kn_div(KnCx a, KnCx b) {
  const double d = b.re * b.re + b.im * b.im;
  return KnCx{(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
}
__device__ KnCx kn_sqrt(KnCx z) {
  if (z.re == 0.0 && z.im == 0.0) return KnCx{0.0, 0.0};
  const double r = sqrt(z.re * z.re + z.im * z.im);
  const double t = sqrt((r + fabs(z.re)) / 2.0);
  if (z.re >= 0.0) return KnCx{t, z.im / (2.0 * t)};
  return KnCx{fabs(z.im) / (2.0 * t), z.im < 0.0 ? -t : t};
}
__device__ double kn_cbrt_pos(double m) {
  int e;
  const double f = frexp(m, &e);
  int k = e / 3;
  if (3 * k < e) k++;
  const double g = ldexp(f, e - 3 * k);
  double t = 1.0;
  for (int it = 0; it < 100; it++) {
    const double tn = t - (t * t * t - g) / (3.0 * t * t);
    if (tn >= t) break;
    t = tn;
  }
  return ldexp(t, k);
}
__device__ KnCx kn_cbrt(KnCx z) {
  const double m = sqrt(z.re * z.re + z.im * z.im);
  if (m == 0.0) return KnCx{0.0, 0.0};
  const double c = z.re / m;
  double x = 1.0;
  for (int it = 0; it < 100; it++) {
    const double xn = x - (4.0 * x * x * x - 3.0 * x - c) / (12.0 * x * x - 3.0);
    if (xn >= x) break;
    x = xn;
  }
  double s2 = 1.0 - x * x;
  if (s2 < 0.0) s2 = 0.0;
  const double sn = sqrt(s2);
  const double rm = kn_cbrt_pos(m);
  return KnCx{rm * x, z.im < 0.0 ? -(rm * sn) : rm * sn};
}
__device__ void kn_o4_roots(const double* p, double* roots) {
  const double A = p[0], B = p[1], C = p[2], D = p[3], E = p[4];
  const double A_pw2 = A * A, B_pw2 = B * B, A_pw3 = A_pw2 * A, B_pw3 = B_pw2 * B, A_pw4 = A_pw3 * A,
               B_pw4 = B_pw3 * B;
  const double alpha = -3 * B_pw2 / (8 * A_pw2) + C / A;
  const double beta = B_pw3 / (8 * A_pw3) - B * C / (2 * A_pw2) + D / A;
  const double gamma = -3 * B_pw4 / (256 * A_pw4) + B_pw2 * C / (16 * A_pw3) - B * D / (4 * A_pw2) + E / A;
  const double alpha_pw2 = alpha * alpha, alpha_pw3 = alpha_pw2 * alpha;
  const double P = -alpha_pw2 / 12 - gamma;
  const double Q = -alpha_pw3 / 108 + alpha * gamma / 3 - beta * beta / 8;
  const KnCx R = kn_add(kn_cx(-Q / 2.0), kn_sqrt(kn_cx(Q * Q / 4.0 + P * P * P / 27.0)));
  const KnCx U = kn_cbrt(R);
  KnCx y;
  if (U.re == 0) {
    const double q3 = Q == 0.0 ? 0.0 : (Q > 0 ? kn_cbrt_pos(Q) : -kn_cbrt_pos(-Q));
    y = kn_cx(-5.0 * alpha / 6.0 - q3);
  } else {
    y = kn_add(kn_sub(kn_cx(-5.0 * alpha / 6.0), kn_div(kn_cx(P), kn_scale(U, 3.0))), U);
  }
  const KnCx w = kn_sqrt(kn_add(kn_cx(alpha), kn_scale(y, 2.0)));
  const KnCx base = kn_add(kn_cx(3.0 * alpha), kn_scale(y, 2.0));
  const KnCx bw = kn_div(kn_cx(2.0 * beta), w);
  const KnCx s1 = kn_sqrt(kn_scale(kn_add(base, bw), -1.0));
  const KnCx s2 = kn_sqrt(kn_scale(kn_sub(base, bw), -1.0));
  const double sh = -B / (4.0 * A);
  roots[0] = sh + 0.5 * (w.re + s1.re);
  roots[1] = sh + 0.5 * (w.re - s1.re);
  roots[2] = sh + 0.5 * (-w.re + s2.re);
  roots[3] = sh + 0.5 * (-w.re - s2.re);
}
__device__ __forceinline__ double kn_nrm3(const double* a) { return sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]); }
__device__ __forceinline__ double ep_distance(const double* model, const double* bearing, const double* point);

// p3p_kneip_main on sel[0..3) + the choice by sel[3]; false when no solution scores (collinear points, NaN)
__device__ bool kn_solve(const double* f, const double* p, const int* sel, double* model) {
  double P1[3], P2[3], P3[3], f1[3], f2[3], f3[3];
  for (int c = 0; c < 3; c++) {
    P1[c] = p[3 * (size_t)sel[0] + c];
    P2[c] = p[3 * (size_t)sel[1] + c];
    P3[c] = p[3 * (size_t)sel[2] + c];
    f1[c] = f[3 * (size_t)sel[0] + c];
    f2[c] = f[3 * (size_t)sel[1] + c];
    f3[c] = f[3 * (size_t)sel[2] + c];
  }
  double temp1[3], temp2[3], cr[3];
  for (int c = 0; c < 3; c++) {
    temp1[c] = P2[c] - P1[c];
    temp2[c] = P3[c] - P1[c];
  }
  rs_cross3(temp1, temp2, cr);
  if (kn_nrm3(cr) == 0) return false;
  double T[9];
  auto frame = [&](const double* a, const double* b) {
    double e3[3], e2[3];
    rs_cross3(a, b, e3);
    const double n = kn_nrm3(e3);
    for (int c = 0; c < 3; c++) e3[c] = e3[c] / n;
    rs_cross3(e3, a, e2);
    for (int c = 0; c < 3; c++) {
      T[c] = a[c];
      T[3 + c] = e2[c];
      T[6 + c] = e3[c];
    }
  };
  frame(f1, f2);
  double f3t[3];
  for (int r = 0; r < 3; r++) f3t[r] = ep_d3(T + 3 * r, f3);
  if (f3t[2] > 0) {
    for (int c = 0; c < 3; c++) {
      const double t = f1[c];
      f1[c] = f2[c];
      f2[c] = t;
      const double q = P1[c];
      P1[c] = P2[c];
      P2[c] = q;
    }
    frame(f1, f2);
    for (int r = 0; r < 3; r++) f3t[r] = ep_d3(T + 3 * r, f3);
  }
  double n1[3], n2[3], n3[3], d31[3], N[9];
  for (int c = 0; c < 3; c++) {
    n1[c] = P2[c] - P1[c];
    d31[c] = P3[c] - P1[c];
  }
  const double nn1 = kn_nrm3(n1);
  for (int c = 0; c < 3; c++) n1[c] = n1[c] / nn1;
  rs_cross3(n1, d31, n3);
  const double nn3 = kn_nrm3(n3);
  for (int c = 0; c < 3; c++) n3[c] = n3[c] / nn3;
  rs_cross3(n3, n1, n2);
  for (int c = 0; c < 3; c++) {
    N[c] = n1[c];
    N[3 + c] = n2[c];
    N[6 + c] = n3[c];
  }
  double P3n[3];
  for (int r = 0; r < 3; r++) P3n[r] = ep_d3(N + 3 * r, d31);
  const double d_12 = kn_nrm3(temp1);
  const double f_1 = f3t[0] / f3t[2], f_2 = f3t[1] / f3t[2], p_1 = P3n[0], p_2 = P3n[1];
  const double cos_beta = ep_d3(f1, f2);
  double b = 1 / (1 - cos_beta * cos_beta) - 1;
  if (cos_beta < 0)
    b = -sqrt(b);
  else
    b = sqrt(b);
  const double f_1_pw2 = f_1 * f_1, f_2_pw2 = f_2 * f_2, p_1_pw2 = p_1 * p_1, p_1_pw3 = p_1_pw2 * p_1,
               p_1_pw4 = p_1_pw3 * p_1, p_2_pw2 = p_2 * p_2, p_2_pw3 = p_2_pw2 * p_2, p_2_pw4 = p_2_pw3 * p_2,
               d_12_pw2 = d_12 * d_12, b_pw2 = b * b;
  double factors[5];
  factors[0] = -f_2_pw2 * p_2_pw4 - p_2_pw4 * f_1_pw2 - p_2_pw4;
  factors[1] = 2 * p_2_pw3 * d_12 * b + 2 * f_2_pw2 * p_2_pw3 * d_12 * b - 2 * f_2 * p_2_pw3 * f_1 * d_12;
  factors[2] = -f_2_pw2 * p_2_pw2 * p_1_pw2 - f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2 - f_2_pw2 * p_2_pw2 * d_12_pw2 +
               f_2_pw2 * p_2_pw4 + p_2_pw4 * f_1_pw2 + 2 * p_1 * p_2_pw2 * d_12 +
               2 * f_1 * f_2 * p_1 * p_2_pw2 * d_12 * b - p_2_pw2 * p_1_pw2 * f_1_pw2 +
               2 * p_1 * p_2_pw2 * f_2_pw2 * d_12 - p_2_pw2 * d_12_pw2 * b_pw2 - 2 * p_1_pw2 * p_2_pw2;
  factors[3] = 2 * p_1_pw2 * p_2 * d_12 * b + 2 * f_2 * p_2_pw3 * f_1 * d_12 - 2 * f_2_pw2 * p_2_pw3 * d_12 * b -
               2 * p_1 * p_2 * d_12_pw2 * b;
  factors[4] = -2 * f_2 * p_2_pw2 * f_1 * p_1 * d_12 * b + f_2_pw2 * p_2_pw2 * d_12_pw2 + 2 * p_1_pw3 * d_12 -
               p_1_pw2 * d_12_pw2 + f_2_pw2 * p_2_pw2 * p_1_pw2 - p_1_pw4 - 2 * f_2_pw2 * p_2_pw2 * p_1 * d_12 +
               p_2_pw2 * f_1_pw2 * p_1_pw2 + f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2;
  double roots[4];
  kn_o4_roots(factors, roots);
  double minScore = 1000000.0;
  bool have = false;
  const double* f4 = f + 3 * (size_t)sel[3];
  const double* p4 = p + 3 * (size_t)sel[3];
  for (int i = 0; i < 4; i++) {
    const double cot_alpha =
        (-f_1 * p_1 / f_2 - roots[i] * p_2 + d_12 * b) / (-f_1 * roots[i] * p_2 / f_2 + p_1 - d_12);
    const double cos_theta = roots[i];
    const double sin_theta = sqrt(1 - roots[i] * roots[i]);
    const double sin_alpha = sqrt(1 / (cot_alpha * cot_alpha + 1));
    double cos_alpha = sqrt(1 - sin_alpha * sin_alpha);
    if (cot_alpha < 0) cos_alpha = -cos_alpha;
    const double k = sin_alpha * b + cos_alpha;
    const double Cv[3] = {d_12 * cos_alpha * k, cos_theta * d_12 * sin_alpha * k, sin_theta * d_12 * sin_alpha * k};
    double sol[12];
    for (int r = 0; r < 3; r++) sol[4 * r + 3] = P1[r] + ((N[r] * Cv[0] + N[3 + r] * Cv[1]) + N[6 + r] * Cv[2]);
    const double Rm[9] = {-cos_alpha, -sin_alpha * cos_theta, -sin_alpha * sin_theta,
                          sin_alpha,  -cos_alpha * cos_theta, -cos_alpha * sin_theta,
                          0.0,        -sin_theta,             cos_theta};
    double NtRt[9];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) NtRt[3 * r + c] = (N[r] * Rm[3 * c] + N[3 + r] * Rm[3 * c + 1]) + N[6 + r] * Rm[3 * c + 2];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++)
        sol[4 * r + c] = (NtRt[3 * r] * T[c] + NtRt[3 * r + 1] * T[3 + c]) + NtRt[3 * r + 2] * T[6 + c];
    const double score = ep_distance(sol, f4, p4);
    if (score < minScore) {
      minScore = score;
      have = true;
      for (int q = 0; q < 12; q++) model[q] = sol[q];
    }
  }
  return have;
}

// AbsolutePoseSacProblem::getSelectedDistancesToModel for one correspondence
__device__ __forceinline__ double ep_distance(const double* model, const double* bearing, const double* point) {
  const double dlt[3] = {point[0] - model[3], point[1] - model[7], point[2] - model[11]};
  double r[3];
  for (int c = 0; c < 3; c++) r[c] = (model[c] * dlt[0] + model[4 + c] * dlt[1]) + model[8 + c] * dlt[2];
  const double nrm = sqrt((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
  for (int c = 0; c < 3; c++) r[c] = r[c] / nrm;
  return 1.0 - ((r[0] * bearing[0] + r[1] * bearing[1]) + r[2] * bearing[2]);
}

// Tracker::pnp over n correspondences, executed by the whole 256-thread workgroup.  out_status: outlierRejectionPnP's
// status; out_counts: [n_inliers, iterations, success]; out_pose 3x4; inliers ascending.  shuffled: [n] ints of LDS.
template <int ALG>   // Pose3d2dAlgorithm: 3 = EPNP (6 points per sample), 1 = KneipP3P (3 + 1)
__device__ void ep_ransac_block(const KParams& P, const Tables& T, const double* f, const double* p, int n,
                                double threshold, int min_inliers, int* shuffled, int* inliers, int* out_status,
                                double* out_pose, int* out_counts) {
  constexpr int SS = ALG == 3 ? EP_N : 4;   // AbsolutePoseSacProblem::getSampleSize()
  __shared__ int wave_tot[RS_T / 64];
  __shared__ int sh_sel[EP_BATCH][EP_N];
  __shared__ int sh_cnt[EP_BATCH];
  __shared__ int sh_ok[EP_BATCH];
  __shared__ double sh_models[EP_BATCH][12];
  __shared__ double sh_best[12];
  __shared__ EpSlot sh_slots[EP_BATCH];
  __shared__ int sh_state[2];
  const int tid = threadIdx.x;
  for (int i = tid; i < n; i += RS_T) shuffled[i] = i;
  __syncthreads();
  int iterations = 0, best = -INT_MAX, draw = 0;
  const unsigned max_skip = (unsigned)P.ransac_max_iters * 10u;
  unsigned skipped = 0;   // (computeModelCoefficients of EPNP never fails; P3P does for collinear points)
  double k = 1.0;
  bool have_model = false, done = false;
  if (n < SS) {
    iterations = INT_MAX;   // getSamples: not enough correspondences -> computeModel returns false
    done = true;
  }
  while (!done) {
    if (tid == 0) {   // drawIndexSample for the next EP_BATCH hypotheses (the shuffle persists)
      for (int h = 0; h < EP_BATCH; h++) {
        for (int i = 0; i < SS; ++i) {
          const int r = T.ransac_rnd[min(draw + SS * h + i, T.n_ransac_rnd - 1)];
          const int j = i + (int)((unsigned)r % (unsigned)(n - i));
          const int tmp = shuffled[i];
          shuffled[i] = shuffled[j];
          shuffled[j] = tmp;
        }
        for (int i = 0; i < SS; i++) sh_sel[h][i] = shuffled[i];
      }
    }
    draw += SS * EP_BATCH;
    __syncthreads();
    {
      static_assert(EP_BATCH * 32 == RS_T, "32 lanes per hypothesis: two solver lanes per wavefront");
      const int h = tid >> 5;
      if ((tid & 31) == 0) {
        if (ALG == 3) {
          ep_solve(f, p, sh_sel[h], sh_slots[h], sh_models[h]);
          sh_ok[h] = 1;
        } else {
          sh_ok[h] = kn_solve(f, p, sh_sel[h], sh_models[h]) ? 1 : 0;
        }
      }
    }
    __syncthreads();
    for (int h = 0; h < EP_BATCH; h++) {   // countWithinDistance
      if (!sh_ok[h]) continue;
      double M[12];
      for (int i = 0; i < 12; i++) M[i] = sh_models[h][i];
      int cnt = 0;
      for (int i = tid; i < n; i += RS_T)
        if (ep_distance(M, f + 3 * (size_t)i, p + 3 * (size_t)i) < threshold) cnt++;
      cnt = rs_block_sum(cnt, wave_tot);
      if (tid == 0) sh_cnt[h] = cnt;
    }
    __syncthreads();
    for (int h = 0; h < EP_BATCH && !done; h++) {   // replay of Ransac::computeModel
      if (!((double)iterations < k && skipped < max_skip)) {
        done = true;
        break;
      }
      if (!sh_ok[h]) {
        ++skipped;
        continue;
      }
      const int cnt = sh_cnt[h];
      if (cnt > best) {
        best = cnt;
        have_model = true;
        if (tid < 12) sh_best[tid] = sh_models[h][tid];
        const double w = (double)best / (double)n;
        double p_no_outliers = 1.0 - pow(w, (double)SS);
        p_no_outliers = fmax(2.220446049250313e-16, p_no_outliers);
        p_no_outliers = fmin(1.0 - 2.220446049250313e-16, p_no_outliers);
        k = log(1.0 - P.ransac_probability) / log(p_no_outliers);
      }
      ++iterations;
      if (iterations > P.ransac_max_iters) done = true;
    }
    if (!((double)iterations < k && skipped < max_skip)) done = true;
    __syncthreads();
  }
  bool success = have_model;
  int n_in = 0;
  if (have_model) {
    __syncthreads();
    double M[12];
    for (int i = 0; i < 12; i++) M[i] = sh_best[i];
    if (tid == 0) sh_state[0] = 0;
    __syncthreads();
    for (int base = 0; base < n; base += RS_T) {   // selectWithinDistance
      const int i = base + tid;
      const bool in = i < n && ep_distance(M, f + 3 * (size_t)i, p + 3 * (size_t)i) < threshold;
      int tot;
      const int pos = rs_scan(in ? 1 : 0, wave_tot, &tot);
      const int off = sh_state[0];
      if (in) inliers[off + pos] = i;
      __syncthreads();
      if (tid == 0) sh_state[0] = off + tot;
      __syncthreads();
    }
    n_in = sh_state[0];
    if (iterations >= P.ransac_max_iters && n_in == 0) {   // Tracker.h:270-273
      success = false;
      n_in = 0;
    }
  }
  if (tid == 0) {
    out_counts[0] = n_in;
    out_counts[1] = iterations;
    out_counts[2] = success ? 1 : 0;
    out_status[0] = (success && n_in > min_inliers) ? TRK_VALID : TRK_FEW_MATCHES;
    for (int i = 0; i < 12; i++) out_pose[i] = success ? sh_best[i] : ((i % 5 == 0) ? 1.0 : 0.0);
  }
}

template <int ALG>
__global__ __launch_bounds__(RS_T) void pnp_ransac_kernel(KParams P, Tables T, const double* f, const double* p, int n,
                                                          double threshold, int min_inliers, int* inliers,
                                                          int* out_status, double* out_pose, int* out_counts) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  ep_ransac_block<ALG>(P, T, f, p, n, threshold, min_inliers, reinterpret_cast<int*>(lds_raw), inliers, out_status,
                       out_pose, out_counts);
}

// VisionImuFrontend::outlierRejectionPnP on the keyframes of the step (StereoVisionImuFrontend.cpp:389-399,
// RgbdVisionImuFrontend.cpp:328-341): Tracker::pnp(const StereoFrame&) (Tracker.cpp:1064-1120) gathers, in keypoint order,
// the keypoints with a VALID rectified left keypoint whose landmark id is in the map of Tracker::updateMap (binary
// search over the sorted ids) -- bearing vector = keypoints_3d_[i] as upstream -- and runs the RANSAC above.
// One workgroup per stream; LDS: shuffled [kcap] int.
template <int ALG>
__global__ __launch_bounds__(RS_T) void pnp_frontend_kernel(KParams P, Tables T, FrameTab K, StereoTab ST, StreamState S,
                                                            RansacScratch RS) {
  const int s = blockIdx.x, tid = threadIdx.x;
  const int flags = S.flags[s];
  if (!(flags & FLAG_KEYFRAME) || (flags & FLAG_FIRST)) return;
  if (!P.use_ransac) {   // RgbdVisionImuFrontend.cpp:342-346 sets DISABLED; the stereo front-end leaves the field alone
    if (P.rgbd && tid == 0) S.pnp_status[s] = TRK_DISABLED;
    return;
  }
  if (!P.use_pnp) return;   // (status INVALID, pose identity: what the fields hold from the start)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ int wave_tot_g[RS_T / 64];
  __shared__ int sh_total;
  const size_t so = (size_t)s * P.kcap;
  const int cnt = S.n_tracked[s];   // frame k holds the tracked keypoints only at this point
  const long long* ids = S.map_ids + (size_t)s * P.map_cap;
  const double* xyz = S.map_xyz + (size_t)s * P.map_cap * 3;
  const int nm = S.map_n[s];
  double* f = RS.f_ref + so * 3;
  double* p = RS.f_cur + so * 3;
  // removeOutliersStereo (Tracker.cpp:763-767, 886-917) zeroes keypoints_3d_ of the stereo outliers in frame k before
  // PnP looks at it; the step never materialises that (the second sparseStereoReconstruction of the keyframe recomputes
  // every entry), so the outliers are flagged here from the match list and the inlier list of the stereo rejection
  unsigned char* outl = lds_raw;   // [kcap] flags (the area is re-initialised as `shuffled` afterwards)
  for (int i = tid; i < cnt; i += RS_T) outl[i] = 0;
  __syncthreads();
  if (P.use_stereo_tracking) {
    const bool voting = P.ransac_1pt_stereo && !rs_rot_is_identity(S.kf_R_cur + (size_t)s * 9);
    const int st_stereo = S.trk_status[2 * (size_t)s + 1];
    const int* cn = S.trk_counts + 6 * (size_t)s;
    int nmat = -1, n_in = 0;
    if (voting) {   // geometricOutlierRejection3d3dGivenRotation: outliers removed whatever the status
      nmat = RS.n_matches[s];
      n_in = st_stereo != TRK_INVALID ? cn[4] : 0;
    } else if (st_stereo != TRK_INVALID) {   // 3-point problem: only after a successful RANSAC
      nmat = cn[3];
      n_in = cn[4];
    }
    const int2* matches = RS.matches + so;
    const int* inl = RS.inliers + so;   // ascending match indices
    for (int m = tid; m < nmat; m += RS_T) {
      int lo = 0, hi = n_in;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (inl[mid] < m) lo = mid + 1;
        else hi = mid;
      }
      if (!(lo < n_in && inl[lo] == m)) outl[matches[m].y] = 1;
    }
    __syncthreads();
  }
  if (tid == 0) sh_total = 0;
  __syncthreads();
  for (int base = 0; base < cnt; base += RS_T) {
    const int i = base + tid;
    int hit = -1;
    if (i < cnt && ST.left_status[so + i] == 0 /* KeypointStatus::VALID */) {
      const long long id = K.lmk[so + i];
      if (id != -1) {
        int lo = 0, hi = nm;   // first index with ids[idx] >= id
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (ids[mid] < id) lo = mid + 1;
          else hi = mid;
        }
        if (lo < nm && ids[lo] == id) hit = lo;
      }
    }
    int tot;
    const int pos = rs_scan(hit >= 0 ? 1 : 0, wave_tot_g, &tot);
    const int off = sh_total;
    if (hit >= 0) {
      for (int c = 0; c < 3; c++) {
        f[3 * (size_t)(off + pos) + c] = outl[i] ? 0.0 : ST.kp3d[(so + i) * 3 + c];
        p[3 * (size_t)(off + pos) + c] = xyz[3 * (size_t)hit + c];
      }
    }
    __syncthreads();
    if (tid == 0) sh_total = off + tot;
    __syncthreads();
  }
  const int n = sh_total;
  __syncthreads();
  if (n == 0) {   // "No 2D-3D correspondences found for 2D-3D RANSAC...": Pose3(), no inliers, failure
    if (tid == 0) {
      S.pnp_status[s] = TRK_FEW_MATCHES;
      for (int i = 0; i < 3; i++) S.pnp_counts[3 * (size_t)s + i] = 0;
      for (int i = 0; i < 12; i++) S.pnp_pose[12 * (size_t)s + i] = (i % 5 == 0) ? 1.0 : 0.0;
    }
    return;
  }
  ep_ransac_block<ALG>(P, T, f, p, n, P.pnp_threshold, P.pnp_min_inliers, reinterpret_cast<int*>(lds_raw),
                       RS.inliers + so, S.pnp_status + s, S.pnp_pose + 12 * (size_t)s, S.pnp_counts + 3 * (size_t)s);
}

void launch_pnp_frontend(const KParams& P, const Tables& T, const FrameTab& k, const StereoTab& ST, const StreamState& S,
                         const RansacScratch& RS, hipStream_t st) {
  const size_t lds = sizeof(int) * (size_t)P.kcap;
  if (P.pnp_alg == 1)
    hipLaunchKernelGGL(pnp_frontend_kernel<1>, dim3(P.B), dim3(RS_T), lds, st, P, T, k, ST, S, RS);
  else
    hipLaunchKernelGGL(pnp_frontend_kernel<3>, dim3(P.B), dim3(RS_T), lds, st, P, T, k, ST, S, RS);
}

void launch_pnp(const KParams& P, const Tables& T, int algorithm, const double* f, const double* p, int n,
                double threshold, int min_inliers, int* inliers, int* out_status, double* out_pose, int* out_counts,
                hipStream_t st) {
  const size_t lds = sizeof(int) * (size_t)(n > 0 ? n : 1);
  if (algorithm == 1)
    hipLaunchKernelGGL(pnp_ransac_kernel<1>, dim3(1), dim3(RS_T), lds, st, P, T, f, p, n, threshold, min_inliers,
                       inliers, out_status, out_pose, out_counts);
  else
    hipLaunchKernelGGL(pnp_ransac_kernel<3>, dim3(1), dim3(RS_T), lds, st, P, T, f, p, n, threshold, min_inliers,
                       inliers, out_status, out_pose, out_counts);
}
