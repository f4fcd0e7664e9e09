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

// ---------------------------------------------------------------------------------------------------------------
// AbsolutePoseSacProblem(adapter, KNEIP) (pnp_algorithm: 1, params/KinectAzure): sample size 4 = three correspondences
// for absolute_pose::p3p_kneip (Kneip, Scaramuzza, Siegwart, "A novel parametrization of the perspective-three-point
// problem for a direct computation of absolute camera position and orientation", CVPR 2011, the authors' published
// p3p code as carried by OpenGV: intermediate camera and world frames, the quartic in cos(theta), closed-form roots,
// back-substitution) + a fourth one that picks among the up to four solutions by reprojection.
// One deliberate difference: OpenGV's math::o4_roots evaluates Ferrari's formulas with std::complex pow / sqrt
// (libm transcendental functions); here the complex square and cube roots are computed with +, -, *, / and sqrt only
// (fixed Newton iterations), so that this restatement and the device code agree bit for bit.  Any cube root of R is a
// valid choice in Ferrari's method, the four roots are the same up to rounding.
// ---------------------------------------------------------------------------------------------------------------
struct Cx {
  double re, im;
};
inline Cx cx(double a, double b = 0.0) { return Cx{a, b}; }
inline Cx cx_add(Cx a, Cx b) { return Cx{a.re + b.re, a.im + b.im}; }
inline Cx cx_sub(Cx a, Cx b) { return Cx{a.re - b.re, a.im - b.im}; }
inline Cx cx_mul(Cx a, Cx b) { return Cx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
inline Cx cx_scale(Cx a, double s) { return Cx{a.re * s, a.im * s}; }
inline Cx cx_div(Cx a, Cx b) {
  const double d = b.re * b.re + b.im * b.im;
  return Cx{(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
}
inline Cx cx_sqrt(Cx z) {   // principal square root
  if (z.re == 0.0 && z.im == 0.0) return Cx{0.0, 0.0};
  const double r = std::sqrt(z.re * z.re + z.im * z.im);
  const double t = std::sqrt((r + std::fabs(z.re)) / 2.0);
  if (z.re >= 0.0) return Cx{t, z.im / (2.0 * t)};
  return Cx{std::fabs(z.im) / (2.0 * t), z.im < 0.0 ? -t : t};
}
// real cube root of m > 0: m = g 2^(3k) with g in [1/8, 1): Newton on t^3 = g from t = 1 (monotone from above, stops
// at the first step that does not decrease), result t 2^k
inline double cbrt_pos(double m) {
  int e;
  const double f = std::frexp(m, &e);   // m = f 2^e, f in [0.5, 1)
  int k = e / 3;
  if (3 * k < e) k++;                    // k = ceil(e / 3)
  const double g = std::ldexp(f, e - 3 * k);
  double t = 1.0;
  for (int it = 0; it < 100; it++) {
    const double tn = t - (t * t * t - g) / (3.0 * t * t);
    if (tn >= t) break;
    t = tn;
  }
  return std::ldexp(t, k);
}
// a cube root of z (the one with argument arg(z) / 3): |z|^(1/3) (cos(phi / 3) + i sin(phi / 3)); cos(phi / 3) is the
// root in [1/2, 1] of 4 x^3 - 3 x = cos(phi) (Newton from x = 1, monotone), the sign of the sine follows im(z)
inline Cx cx_cbrt(Cx z) {
  const double m = std::sqrt(z.re * z.re + z.im * z.im);
  if (m == 0.0) return Cx{0.0, 0.0};
  const double c = z.re / m;
  double x = 1.0;
  for (int it = 0; it < 100; it++) {
    const double xn = x - (4.0 * x * x * x - 3.0 * x - c) / (12.0 * x * x - 3.0);
    if (xn >= x) break;
    x = xn;
  }
  double s2 = 1.0 - x * x;
  if (s2 < 0.0) s2 = 0.0;
  const double s = std::sqrt(s2);
  const double rm = cbrt_pos(m);
  return Cx{rm * x, z.im < 0.0 ? -(rm * s) : rm * s};
}

// math::o4_roots: real parts of the four roots of p[0] x^4 + p[1] x^3 + p[2] x^2 + p[3] x + p[4] (Ferrari)
void o4_roots(const double* p, double* roots) {
  const double A = p[0], B = p[1], C = p[2], D = p[3], E = p[4];
  const double A_pw2 = A * A, B_pw2 = B * B, A_pw3 = A_pw2 * A, B_pw3 = B_pw2 * B, A_pw4 = A_pw3 * A,
               B_pw4 = B_pw3 * B;
  const double alpha = -3 * B_pw2 / (8 * A_pw2) + C / A;
  const double beta = B_pw3 / (8 * A_pw3) - B * C / (2 * A_pw2) + D / A;
  const double gamma = -3 * B_pw4 / (256 * A_pw4) + B_pw2 * C / (16 * A_pw3) - B * D / (4 * A_pw2) + E / A;
  const double alpha_pw2 = alpha * alpha, alpha_pw3 = alpha_pw2 * alpha;
  const double P = -alpha_pw2 / 12 - gamma;
  const double Q = -alpha_pw3 / 108 + alpha * gamma / 3 - beta * beta / 8;
  const Cx R = cx_add(cx(-Q / 2.0), cx_sqrt(cx(Q * Q / 4.0 + P * P * P / 27.0)));
  const Cx U = cx_cbrt(R);
  Cx y;
  if (U.re == 0) {
    const double q3 = Q == 0.0 ? 0.0 : (Q > 0 ? cbrt_pos(Q) : -cbrt_pos(-Q));
    y = cx(-5.0 * alpha / 6.0 - q3);
  } else {
    y = cx_add(cx_sub(cx(-5.0 * alpha / 6.0), cx_div(cx(P), cx_scale(U, 3.0))), U);
  }
  const Cx w = cx_sqrt(cx_add(cx(alpha), cx_scale(y, 2.0)));
  const Cx base = cx_add(cx(3.0 * alpha), cx_scale(y, 2.0));
  const Cx bw = cx_div(cx(2.0 * beta), w);
  const Cx s1 = cx_sqrt(cx_scale(cx_add(base, bw), -1.0));
  const Cx s2 = cx_sqrt(cx_scale(cx_sub(base, bw), -1.0));
  const double sh = -B / (4.0 * A);
  roots[0] = sh + 0.5 * (w.re + s1.re);
  roots[1] = sh + 0.5 * (w.re - s1.re);
  roots[2] = sh + 0.5 * (-w.re + s2.re);
  roots[3] = sh + 0.5 * (-w.re - s2.re);
}

inline void x3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
inline double nrm3(const double* a) { return std::sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]); }

// absolute_pose::modules::p3p_kneip_main: up to four world_T_camera solutions [R | C] (3x4 row-major); returns their number
int p3p_kneip(const double* f, const double* p, const int* idx, double sol[4][12]) {
  double P1[3], P2[3], P3[3], f1[3], f2[3], f3[3];
  for (int c = 0; c < 3; c++) {
    P1[c] = p[3 * idx[0] + c];
    P2[c] = p[3 * idx[1] + c];
    P3[c] = p[3 * idx[2] + c];
    f1[c] = f[3 * idx[0] + c];
    f2[c] = f[3 * idx[1] + c];
    f3[c] = f[3 * idx[2] + c];
  }
  double temp1[3], temp2[3], cr[3];
  for (int c = 0; c < 3; c++) {
    temp1[c] = P2[c] - P1[c];
    temp2[c] = P3[c] - P1[c];
  }
  x3(temp1, temp2, cr);
  if (nrm3(cr) == 0) return 0;   // collinear world points
  double T[9];
  auto frame = [&](const double* a, const double* b) {   // rows e1 = a, e2 = e3 x e1, e3 = a x b normalised
    double e3[3], e2[3];
    x3(a, b, e3);
    const double n = nrm3(e3);
    for (int c = 0; c < 3; c++) e3[c] = e3[c] / n;
    x3(e3, a, e2);
    for (int c = 0; c < 3; c++) {
      T[c] = a[c];
      T[3 + c] = e2[c];
      T[6 + c] = e3[c];
    }
  };
  frame(f1, f2);
  double f3t[3];
  for (int r = 0; r < 3; r++) f3t[r] = d3(T + 3 * r, f3);
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
    for (int r = 0; r < 3; r++) f3t[r] = d3(T + 3 * r, f3);
  }
  double n1[3], n2[3], n3[3], d31[3], N[9];
  for (int c = 0; c < 3; c++) {
    n1[c] = P2[c] - P1[c];
    d31[c] = P3[c] - P1[c];
  }
  const double nn1 = nrm3(n1);
  for (int c = 0; c < 3; c++) n1[c] = n1[c] / nn1;
  x3(n1, d31, n3);
  const double nn3 = nrm3(n3);
  for (int c = 0; c < 3; c++) n3[c] = n3[c] / nn3;
  x3(n3, n1, n2);
  for (int c = 0; c < 3; c++) {
    N[c] = n1[c];
    N[3 + c] = n2[c];
    N[6 + c] = n3[c];
  }
  double P3n[3];
  for (int r = 0; r < 3; r++) P3n[r] = d3(N + 3 * r, d31);
  const double d_12 = nrm3(temp1);
  const double f_1 = f3t[0] / f3t[2], f_2 = f3t[1] / f3t[2], p_1 = P3n[0], p_2 = P3n[1];
  const double cos_beta = d3(f1, f2);
  double b = 1 / (1 - cos_beta * cos_beta) - 1;
  if (cos_beta < 0)
    b = -std::sqrt(b);
  else
    b = std::sqrt(b);
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
  o4_roots(factors, roots);
  for (int i = 0; i < 4; i++) {
    const double cot_alpha =
        (-f_1 * p_1 / f_2 - roots[i] * p_2 + d_12 * b) / (-f_1 * roots[i] * p_2 / f_2 + p_1 - d_12);
    const double cos_theta = roots[i];
    const double sin_theta = std::sqrt(1 - roots[i] * roots[i]);
    const double sin_alpha = std::sqrt(1 / (cot_alpha * cot_alpha + 1));
    double cos_alpha = std::sqrt(1 - sin_alpha * sin_alpha);
    if (cot_alpha < 0) cos_alpha = -cos_alpha;
    const double k = sin_alpha * b + cos_alpha;
    const double Cv[3] = {d_12 * cos_alpha * k, cos_theta * d_12 * sin_alpha * k, sin_theta * d_12 * sin_alpha * k};
    double Cw[3];
    for (int r = 0; r < 3; r++) Cw[r] = P1[r] + ((N[r] * Cv[0] + N[3 + r] * Cv[1]) + N[6 + r] * Cv[2]);   // P1 + N^T C
    const double Rm[9] = {-cos_alpha, -sin_alpha * cos_theta, -sin_alpha * sin_theta,
                          sin_alpha,  -cos_alpha * cos_theta, -cos_alpha * sin_theta,
                          0.0,        -sin_theta,             cos_theta};
    double NtRt[9];   // N^T R^T
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) NtRt[3 * r + c] = (N[r] * Rm[3 * c] + N[3 + r] * Rm[3 * c + 1]) + N[6 + r] * Rm[3 * c + 2];
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++)
        sol[i][4 * r + c] = (NtRt[3 * r] * T[c] + NtRt[3 * r + 1] * T[3 + c]) + NtRt[3 * r + 2] * T[6 + c];
      sol[i][4 * r + 3] = Cw[r];
    }
  }
  return 4;
}

// AbsolutePoseSacProblem(adapter, KNEIP)
struct AbsolutePoseKneip : Problem {
  const double *f, *p;
  int n;
  int sampleSize() const override { return 4; }
  int size() const override { return n; }
  bool computeModelCoefficients(const std::vector<int>& s, double* model) const override {
    double sol[4][12];
    const int ns = p3p_kneip(f, p, s.data(), sol);
    // the fourth correspondence picks the solution: smallest 1 - f4 . normalize(R^T (p4 - t)) (NaN never wins)
    double minScore = 1000000.0;
    int minIndex = -1;
    AbsolutePoseEpnp scorer;
    scorer.f = f;
    scorer.p = p;
    scorer.n = n;
    for (int i = 0; i < ns; i++) {
      const double score = scorer.distance(sol[i], nullptr, s[3]);
      if (score < minScore) {
        minScore = score;
        minIndex = i;
      }
    }
    if (minIndex == -1) return false;
    std::memcpy(model, sol[minIndex], sizeof(double) * 12);
    return true;
  }
  double distance(const double* model, const double* aux, int i) const override {
    AbsolutePoseEpnp scorer;
    scorer.f = f;
    scorer.p = p;
    scorer.n = n;
    return scorer.distance(model, aux, i);
  }
};
