// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
// CPU restatement of the part of OpenGV behind Kimera-VIO's PnP tracking (Tracker::pnp, src/frontend/Tracker.cpp:1122-1288,
// pnp_algorithm: 3 = EPNP in every shipped parameter set but params/KinectAzure):
//   opengv::sac_problems::absolute_pose::AbsolutePoseSacProblem(adapter, EPNP)
//     getSampleSize() = 6, computeModelCoefficients = absolute_pose::epnp(adapter, indices),
//     getSelectedDistancesToModel = 1 - f . normalize(R^T (p - t))
//   opengv::absolute_pose::epnp -> modules::Epnp: the EPnP algorithm of Lepetit, Moreno-Noguer and Fua ("EPnP: An
//     accurate O(n) solution to the PnP problem", IJCV 2009) in the authors' published reference implementation, which
//     OpenGV carries with two changes: the observation is a bearing vector (u = x/z, v = y/z, unit focal length, zero
//     principal point) and its sign (z > 0 or not) decides the sign of the camera-frame control points.
// OpenGV is an un-vendored third-party dependency (fork marcusabate/opengv, no pin, Dockerfile_20_04:55); its sources
// are not in /root/reference.  What is restated is the published algorithm, step by step (control points from the PCA of
// the world points, barycentric coordinates, M^T M, its four smallest eigenvectors, L_6x10 / rho, the three beta
// approximations, five Gauss-Newton steps with the authors' Householder solver, Horn/Arun alignment, reprojection
// error, best of N = 1..3).  PARITY UNPINNED at the bit level: the dense linear algebra (symmetric eigen-decomposition
// of M^T M and of the 3x3 covariance, least squares) is a cyclic Jacobi / Householder QR written here, not Eigen's
// JacobiSVD.  It is anchored on the reference's own known-answer test for this call site, tests/testTracker.cpp:
// 1613-1800 (PnPTracking: 22 landmarks seen from a known pose + 3 outliers -> 22 inliers, pose within 1e-5), see
// tests/test_oracle_pnp.py.
// (included by opengv_re.cpp inside its unnamed namespace: uses svd3 and the Problem interface defined there)

inline double d3(const double* a, const double* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

// cyclic Jacobi for a symmetric n x n matrix (row-major, destroyed); V: columns = eigenvectors; w: eigenvalues.
// Sorted by descending eigenvalue afterwards (singular values of a positive semi-definite matrix, U = V).
void jacobi_eig_sym(double* a, int n, double* v, double* d) {
  std::vector<double> b(n), z(n);
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) v[i * n + j] = (i == j) ? 1.0 : 0.0;
    b[i] = d[i] = a[i * n + i];
    z[i] = 0.0;
  }
  for (int it = 1; it <= 50; it++) {
    double sm = 0.0;
    for (int p = 0; p < n - 1; p++)
      for (int q = p + 1; q < n; q++) sm += std::fabs(a[p * n + q]);
    if (sm == 0.0) break;
    const double tresh = it < 4 ? 0.2 * sm / (double)(n * n) : 0.0;
    for (int p = 0; p < n - 1; p++)
      for (int q = p + 1; q < n; q++) {
        const double g = 100.0 * std::fabs(a[p * n + q]);
        if (it > 4 && std::fabs(d[p]) + g == std::fabs(d[p]) && std::fabs(d[q]) + g == std::fabs(d[q])) {
          a[p * n + q] = 0.0;
        } else if (std::fabs(a[p * n + q]) > tresh) {
          double h = d[q] - d[p];
          double t;
          if (std::fabs(h) + g == std::fabs(h)) {
            t = a[p * n + q] / h;
          } else {
            const double theta = 0.5 * h / a[p * n + q];
            t = 1.0 / (std::fabs(theta) + std::sqrt(1.0 + theta * theta));
            if (theta < 0.0) t = -t;
          }
          const double c = 1.0 / std::sqrt(1.0 + t * t);
          const double s = t * c;
          const double tau = s / (1.0 + c);
          h = t * a[p * n + q];
          z[p] -= h;
          z[q] += h;
          d[p] -= h;
          d[q] += h;
          a[p * n + q] = 0.0;
          auto rot = [&](double* m, int i1, int j1, int i2, int j2) {
            const double gg = m[i1 * n + j1], hh = m[i2 * n + j2];
            m[i1 * n + j1] = gg - s * (hh + gg * tau);
            m[i2 * n + j2] = hh + s * (gg - hh * tau);
          };
          for (int j = 0; j < p; j++) rot(a, j, p, j, q);
          for (int j = p + 1; j < q; j++) rot(a, p, j, j, q);
          for (int j = q + 1; j < n; j++) rot(a, p, j, q, j);
          for (int j = 0; j < n; j++) rot(v, j, p, j, q);
        }
      }
    for (int i = 0; i < n; i++) {
      b[i] += z[i];
      d[i] = b[i];
      z[i] = 0.0;
    }
  }
  // descending order: selection sort, first maximum wins
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
      for (int j = 0; j < n; j++) std::swap(v[j * n + i], v[j * n + k]);
    }
  }
}

// Householder least squares of the EPnP reference code (qr_solve), A: nr x nc row-major (destroyed), b: nr (destroyed)
bool qr_solve(double* A, double* b, double* X, int nr, int nc) {
  double A1[8], A2[8];
  double* ppAkk = A;
  for (int k = 0; k < nc; k++) {
    double* ppAik = ppAkk;
    double eta = std::fabs(*ppAik);
    for (int i = k + 1; i < nr; i++) {
      const double elt = std::fabs(*ppAik);
      if (eta < elt) eta = elt;
      ppAik += nc;
    }
    if (eta == 0) {
      A1[k] = A2[k] = 0.0;
      return false;  // "A is singular, this shouldn't happen"
    }
    ppAik = ppAkk;
    double sum = 0.0;
    const double inv_eta = 1. / eta;
    for (int i = k; i < nr; i++) {
      *ppAik *= inv_eta;
      sum += *ppAik * *ppAik;
      ppAik += nc;
    }
    double sigma = std::sqrt(sum);
    if (*ppAkk < 0) sigma = -sigma;
    *ppAkk += sigma;
    A1[k] = sigma * *ppAkk;
    A2[k] = -eta * sigma;
    for (int j = k + 1; j < nc; j++) {
      double* pp = ppAkk;
      double sm = 0;
      for (int i = k; i < nr; i++) {
        sm += *pp * pp[j - k];
        pp += nc;
      }
      const double tau = sm / A1[k];
      pp = ppAkk;
      for (int i = k; i < nr; i++) {
        pp[j - k] -= tau * *pp;
        pp += nc;
      }
    }
    ppAkk += nc + 1;
  }
  // b <- Qt b
  double* ppAjj = A;
  for (int j = 0; j < nc; j++) {
    double* ppAij = ppAjj;
    double tau = 0;
    for (int i = j; i < nr; i++) {
      tau += *ppAij * b[i];
      ppAij += nc;
    }
    tau /= A1[j];
    ppAij = ppAjj;
    for (int i = j; i < nr; i++) {
      b[i] -= tau * *ppAij;
      ppAij += nc;
    }
    ppAjj += nc + 1;
  }
  // X = R-1 b
  X[nc - 1] = b[nc - 1] / A2[nc - 1];
  for (int i = nc - 2; i >= 0; i--) {
    double* ppAij = A + i * nc + (i + 1);
    double sum = 0;
    for (int j = i + 1; j < nc; j++) {
      sum += *ppAij * X[j];
      ppAij++;
    }
    X[i] = (b[i] - sum) / A2[i];
  }
  return true;
}

// Eigen::Matrix3d::inverse() (cofactors)
void inv3(const double* m, double* r) {
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

struct Epnp {
  int n = 0;
  std::vector<double> pws, us, alphas, pcs;
  std::vector<int> signs;
  double cws[4][3], ccs[4][3];

  void add(const double* p, const double* f) {
    pws.insert(pws.end(), p, p + 3);
    us.push_back(f[0] / f[2]);
    us.push_back(f[1] / f[2]);
    signs.push_back(f[2] > 0.0 ? 1 : -1);
    n++;
  }

  void choose_control_points() {
    cws[0][0] = cws[0][1] = cws[0][2] = 0;
    for (int i = 0; i < n; i++)
      for (int j = 0; j < 3; j++) cws[0][j] += pws[3 * i + j];
    for (int j = 0; j < 3; j++) cws[0][j] /= n;
    double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // PW0^T PW0
    for (int i = 0; i < n; i++) {
      double dlt[3];
      for (int j = 0; j < 3; j++) dlt[j] = pws[3 * i + j] - cws[0][j];
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) C[r * 3 + c] += dlt[r] * dlt[c];
    }
    double V[9], dc[3];
    jacobi_eig_sym(C, 3, V, dc);
    for (int i = 1; i < 4; i++) {
      const double k = std::sqrt(dc[i - 1] / n);
      for (int j = 0; j < 3; j++) cws[i][j] = cws[0][j] + k * V[j * 3 + (i - 1)];  // UCt row i-1 = eigenvector i-1
    }
  }

  void compute_barycentric_coordinates() {
    double cc[9], ci[9];
    for (int i = 0; i < 3; i++)
      for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = cws[j][i] - cws[0][i];
    inv3(cc, ci);
    alphas.assign(4 * n, 0.0);
    for (int i = 0; i < n; i++) {
      const double* pi = &pws[3 * i];
      double* a = &alphas[4 * i];
      for (int j = 0; j < 3; j++)
        a[1 + j] = ci[3 * j] * (pi[0] - cws[0][0]) + ci[3 * j + 1] * (pi[1] - cws[0][1]) +
                   ci[3 * j + 2] * (pi[2] - cws[0][2]);
      a[0] = 1.0 - a[1] - a[2] - a[3];
    }
  }

  void compute_ccs(const double* betas, const double* ut) {
    for (int i = 0; i < 4; i++) ccs[i][0] = ccs[i][1] = ccs[i][2] = 0.0;
    for (int i = 0; i < 4; i++) {
      const double* v = ut + 12 * (11 - i);
      for (int j = 0; j < 4; j++)
        for (int k = 0; k < 3; k++) ccs[j][k] += betas[i] * v[3 * j + k];
    }
  }

  void compute_pcs() {
    pcs.assign(3 * n, 0.0);
    for (int i = 0; i < n; i++) {
      const double* a = &alphas[4 * i];
      for (int j = 0; j < 3; j++)
        pcs[3 * i + j] = a[0] * ccs[0][j] + a[1] * ccs[1][j] + a[2] * ccs[2][j] + a[3] * ccs[3][j];
    }
  }

  void solve_for_sign() {
    if ((pcs[2] < 0.0 && signs[0] > 0) || (pcs[2] > 0.0 && signs[0] < 0)) {
      for (int i = 0; i < 4; i++)
        for (int j = 0; j < 3; j++) ccs[i][j] = -ccs[i][j];
      for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) pcs[3 * i + j] = -pcs[3 * i + j];
    }
  }

  void estimate_R_and_t(double R[3][3], double t[3]) {
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
    for (int i = 0; i < n; i++)
      for (int j = 0; j < 3; j++) {
        pc0[j] += pcs[3 * i + j];
        pw0[j] += pws[3 * i + j];
      }
    for (int j = 0; j < 3; j++) {
      pc0[j] /= n;
      pw0[j] /= n;
    }
    double abt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; i++)
      for (int j = 0; j < 3; j++) {
        abt[3 * j] += (pcs[3 * i + j] - pc0[j]) * (pws[3 * i] - pw0[0]);
        abt[3 * j + 1] += (pcs[3 * i + j] - pc0[j]) * (pws[3 * i + 1] - pw0[1]);
        abt[3 * j + 2] += (pcs[3 * i + j] - pc0[j]) * (pws[3 * i + 2] - pw0[2]);
      }
    double U[9], S[3], V[9];
    svd3(abt, U, S, V);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) R[i][j] = d3(U + 3 * i, V + 3 * j);
    const double det = R[0][0] * R[1][1] * R[2][2] + R[0][1] * R[1][2] * R[2][0] + R[0][2] * R[1][0] * R[2][1] -
                       R[0][2] * R[1][1] * R[2][0] - R[0][1] * R[1][0] * R[2][2] - R[0][0] * R[1][2] * R[2][1];
    if (det < 0) {
      R[2][0] = -R[2][0];
      R[2][1] = -R[2][1];
      R[2][2] = -R[2][2];
    }
    t[0] = pc0[0] - d3(R[0], pw0);
    t[1] = pc0[1] - d3(R[1], pw0);
    t[2] = pc0[2] - d3(R[2], pw0);
  }

  double reprojection_error(const double R[3][3], const double t[3]) {
    double sum2 = 0.0;
    for (int i = 0; i < n; i++) {
      const double* pw = &pws[3 * i];
      const double Xc = d3(R[0], pw) + t[0];
      const double Yc = d3(R[1], pw) + t[1];
      const double inv_Zc = 1.0 / (d3(R[2], pw) + t[2]);
      const double ue = Xc * inv_Zc, ve = Yc * inv_Zc;
      const double u = us[2 * i], v = us[2 * i + 1];
      sum2 += std::sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
    }
    return sum2 / n;
  }

  double compute_R_and_t(const double* ut, const double* betas, double R[3][3], double t[3]) {
    compute_ccs(betas, ut);
    compute_pcs();
    solve_for_sign();
    estimate_R_and_t(R, t);
    return reprojection_error(R, t);
  }

  static void compute_L_6x10(const double* ut, double* l) {
    const double* v[4] = {ut + 12 * 11, ut + 12 * 10, ut + 12 * 9, ut + 12 * 8};
    double dv[4][6][3];
    for (int i = 0; i < 4; i++) {
      int a = 0, b = 1;
      for (int j = 0; j < 6; j++) {
        dv[i][j][0] = v[i][3 * a] - v[i][3 * b];
        dv[i][j][1] = v[i][3 * a + 1] - v[i][3 * b + 1];
        dv[i][j][2] = v[i][3 * a + 2] - v[i][3 * b + 2];
        b++;
        if (b > 3) {
          a++;
          b = a + 1;
        }
      }
    }
    for (int i = 0; i < 6; i++) {
      double* row = l + 10 * i;
      row[0] = d3(dv[0][i], dv[0][i]);
      row[1] = 2.0 * d3(dv[0][i], dv[1][i]);
      row[2] = d3(dv[1][i], dv[1][i]);
      row[3] = 2.0 * d3(dv[0][i], dv[2][i]);
      row[4] = 2.0 * d3(dv[1][i], dv[2][i]);
      row[5] = d3(dv[2][i], dv[2][i]);
      row[6] = 2.0 * d3(dv[0][i], dv[3][i]);
      row[7] = 2.0 * d3(dv[1][i], dv[3][i]);
      row[8] = 2.0 * d3(dv[2][i], dv[3][i]);
      row[9] = d3(dv[3][i], dv[3][i]);
    }
  }

  static double dist2(const double* p1, const double* p2) {
    return (p1[0] - p2[0]) * (p1[0] - p2[0]) + (p1[1] - p2[1]) * (p1[1] - p2[1]) + (p1[2] - p2[2]) * (p1[2] - p2[2]);
  }

  void compute_rho(double* rho) {
    rho[0] = dist2(cws[0], cws[1]);
    rho[1] = dist2(cws[0], cws[2]);
    rho[2] = dist2(cws[0], cws[3]);
    rho[3] = dist2(cws[1], cws[2]);
    rho[4] = dist2(cws[1], cws[3]);
    rho[5] = dist2(cws[2], cws[3]);
  }

  // betas10 = [B11 B12 B22 B13 B23 B33 B14 B24 B34 B44]; least squares on a column subset
  static void solve_subset(const double* l, const double* rho, const int* cols, int nc, double* x) {
    double A[6 * 5], b[6];
    for (int i = 0; i < 6; i++) {
      for (int c = 0; c < nc; c++) A[i * nc + c] = l[10 * i + cols[c]];
      b[i] = rho[i];
    }
    if (!qr_solve(A, b, x, 6, nc))
      for (int c = 0; c < nc; c++) x[c] = 0.0;
  }

  static void find_betas_approx_1(const double* l, const double* rho, double* betas) {  // [B11 B12 B13 B14]
    const int cols[4] = {0, 1, 3, 6};
    double b4[4];
    solve_subset(l, rho, cols, 4, b4);
    if (b4[0] < 0) {
      betas[0] = std::sqrt(-b4[0]);
      betas[1] = -b4[1] / betas[0];
      betas[2] = -b4[2] / betas[0];
      betas[3] = -b4[3] / betas[0];
    } else {
      betas[0] = std::sqrt(b4[0]);
      betas[1] = b4[1] / betas[0];
      betas[2] = b4[2] / betas[0];
      betas[3] = b4[3] / betas[0];
    }
  }
  static void find_betas_approx_2(const double* l, const double* rho, double* betas) {  // [B11 B12 B22]
    const int cols[3] = {0, 1, 2};
    double b3[3];
    solve_subset(l, rho, cols, 3, b3);
    if (b3[0] < 0) {
      betas[0] = std::sqrt(-b3[0]);
      betas[1] = (b3[2] < 0) ? std::sqrt(-b3[2]) : 0.0;
    } else {
      betas[0] = std::sqrt(b3[0]);
      betas[1] = (b3[2] > 0) ? std::sqrt(b3[2]) : 0.0;
    }
    if (b3[1] < 0) betas[0] = -betas[0];
    betas[2] = 0.0;
    betas[3] = 0.0;
  }
  static void find_betas_approx_3(const double* l, const double* rho, double* betas) {  // [B11 B12 B22 B13 B23]
    const int cols[5] = {0, 1, 2, 3, 4};
    double b5[5];
    solve_subset(l, rho, cols, 5, b5);
    if (b5[0] < 0) {
      betas[0] = std::sqrt(-b5[0]);
      betas[1] = (b5[2] < 0) ? std::sqrt(-b5[2]) : 0.0;
    } else {
      betas[0] = std::sqrt(b5[0]);
      betas[1] = (b5[2] > 0) ? std::sqrt(b5[2]) : 0.0;
    }
    if (b5[1] < 0) betas[0] = -betas[0];
    betas[2] = b5[3] / betas[0];
    betas[3] = 0.0;
  }

  static void gauss_newton(const double* l, const double* rho, double* b) {
    for (int k = 0; k < 5; k++) {
      double A[6 * 4], B[6], X[4];
      for (int i = 0; i < 6; i++) {
        const double* rowL = l + i * 10;
        double* rowA = A + i * 4;
        rowA[0] = 2 * rowL[0] * b[0] + rowL[1] * b[1] + rowL[3] * b[2] + rowL[6] * b[3];
        rowA[1] = rowL[1] * b[0] + 2 * rowL[2] * b[1] + rowL[4] * b[2] + rowL[7] * b[3];
        rowA[2] = rowL[3] * b[0] + rowL[4] * b[1] + 2 * rowL[5] * b[2] + rowL[8] * b[3];
        rowA[3] = rowL[6] * b[0] + rowL[7] * b[1] + rowL[8] * b[2] + 2 * rowL[9] * b[3];
        B[i] = rho[i] - (rowL[0] * b[0] * b[0] + rowL[1] * b[0] * b[1] + rowL[2] * b[1] * b[1] +
                         rowL[3] * b[0] * b[2] + rowL[4] * b[1] * b[2] + rowL[5] * b[2] * b[2] +
                         rowL[6] * b[0] * b[3] + rowL[7] * b[1] * b[3] + rowL[8] * b[2] * b[3] +
                         rowL[9] * b[3] * b[3]);
      }
      if (!qr_solve(A, B, X, 6, 4)) return;
      for (int i = 0; i < 4; i++) b[i] += X[i];
    }
  }

  double compute_pose(double R[3][3], double t[3]) {
    choose_control_points();
    compute_barycentric_coordinates();
    // M^T M accumulated row by row (fu = fv = 1, uc = vc = 0)
    double MtM[144];
    std::memset(MtM, 0, sizeof(MtM));
    for (int i = 0; i < n; i++) {
      const double* as = &alphas[4 * i];
      double m1[12], m2[12];
      for (int j = 0; j < 4; j++) {
        m1[3 * j] = as[j];
        m1[3 * j + 1] = 0.0;
        m1[3 * j + 2] = as[j] * (0.0 - us[2 * i]);
        m2[3 * j] = 0.0;
        m2[3 * j + 1] = as[j];
        m2[3 * j + 2] = as[j] * (0.0 - us[2 * i + 1]);
      }
      for (int r = 0; r < 12; r++)
        for (int c = 0; c < 12; c++) {
          MtM[r * 12 + c] += m1[r] * m1[c];
          MtM[r * 12 + c] += m2[r] * m2[c];
        }
    }
    double V[144], D[12], ut[144];
    jacobi_eig_sym(MtM, 12, V, D);
    for (int i = 0; i < 12; i++)
      for (int j = 0; j < 12; j++) ut[i * 12 + j] = V[j * 12 + i];
    double l[60], rho[6];
    compute_L_6x10(ut, l);
    compute_rho(rho);
    double Betas[4][4], rep[4], Rs[4][3][3], ts[4][3];
    find_betas_approx_1(l, rho, Betas[1]);
    gauss_newton(l, rho, Betas[1]);
    rep[1] = compute_R_and_t(ut, Betas[1], Rs[1], ts[1]);
    find_betas_approx_2(l, rho, Betas[2]);
    gauss_newton(l, rho, Betas[2]);
    rep[2] = compute_R_and_t(ut, Betas[2], Rs[2], ts[2]);
    find_betas_approx_3(l, rho, Betas[3]);
    gauss_newton(l, rho, Betas[3]);
    rep[3] = compute_R_and_t(ut, Betas[3], Rs[3], ts[3]);
    int N = 1;
    if (rep[2] < rep[1]) N = 2;
    if (rep[3] < rep[N]) N = 3;
    std::memcpy(R, Rs[N], sizeof(double) * 9);
    std::memcpy(t, ts[N], sizeof(double) * 3);
    return rep[N];
  }
};

// absolute_pose::epnp(adapter, indices): world-from-camera transformation [R^T | -R^T t], row-major 3x4
void epnp_transformation(const double* bearings, const double* points, const int* idx, int n, double model[12]) {
  Epnp pnp;
  for (int i = 0; i < n; i++) pnp.add(points + 3 * idx[i], bearings + 3 * idx[i]);
  double R[3][3], t[3];
  pnp.compute_pose(R, t);
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) model[r * 4 + c] = R[c][r];  // rotation.transposeInPlace()
  }
  for (int r = 0; r < 3; r++)  // translation = -rotation * translation
    model[r * 4 + 3] = -((model[r * 4] * t[0] + model[r * 4 + 1] * t[1]) + model[r * 4 + 2] * t[2]);
}


// AbsolutePoseSacProblem(adapter, EPNP)
struct AbsolutePoseEpnp : Problem {
  const double *f, *p;  // n x 3 bearing vectors (camera frame), n x 3 points (world frame)
  int n;
  int sampleSize() const override { return 6; }
  int size() const override { return n; }
  bool computeModelCoefficients(const std::vector<int>& s, double* model) const override {
    epnp_transformation(f, p, s.data(), (int)s.size(), model);
    return true;
  }
  // getSelectedDistancesToModel: 1 - f . normalize(R^T (p - t))
  double distance(const double* model, const double* /*aux*/, int i) const override {
    const double* point = p + 3 * i;
    const double* bearing = f + 3 * i;
    const double dlt[3] = {point[0] - model[3], point[1] - model[7], point[2] - model[11]};
    double r[3];
    for (int c = 0; c < 3; c++) r[c] = (model[c] * dlt[0] + model[4 + c] * dlt[1]) + model[8 + c] * dlt[2];
    const double nrm = std::sqrt((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
    for (int c = 0; c < 3; c++) r[c] = r[c] / nrm;
    return 1.0 - ((r[0] * bearing[0] + r[1] * bearing[1]) + r[2] * bearing[2]);
  }
};
