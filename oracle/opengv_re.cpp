// TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See opengv_re.hpp for scope and provenance.
#include "opengv_re.hpp"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <limits>

namespace opengv_re {

// ---- std::mt19937 ------------------------------------------------------------------------------
void Mt19937::seed(uint32_t s) {
  mt[0] = s;
  for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
  idx = 624;
}
uint32_t Mt19937::next() {
  if (idx >= 624) {
    for (int k = 0; k < 624; k++) {
      const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
      mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    idx = 0;
  }
  uint32_t y = mt[idx++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

namespace {

// SampleConsensusProblem::rnd(): (*rng_gen_)() with rng_dist_ = uniform_int_distribution<>(0, INT_MAX)
// and rng_alg_.seed(12345u) (randomSeed = false <=> ransac_randomize: 0)
struct Rnd {
  Mt19937 eng;
  int policy;
  explicit Rnd(int pol) : policy(pol) { eng.seed(12345u); }
  int operator()() {
    if (policy == RNG_LIBSTDCXX_11) return (int)(eng.next() >> 1);
    uint32_t r;
    do r = eng.next();
    while (r >= 0x80000000u);
    return (int)r;
  }
};

// Eigen fixed-size semantics used below: 3-term dot = (a0*b0 + a1*b1) + a2*b2
inline double dot3(const double* a, const double* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
inline void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
inline void matvec3(const double* R, const double* v, double* o) {
  for (int r = 0; r < 3; r++) o[r] = (R[r * 3] * v[0] + R[r * 3 + 1] * v[1]) + R[r * 3 + 2] * v[2];
}

// A sample-consensus problem: sample size, model fit, per-correspondence distances
struct Problem {
  virtual ~Problem() {}
  virtual int sampleSize() const = 0;
  virtual int size() const = 0;
  virtual bool computeModelCoefficients(const std::vector<int>& sample, double* model) const = 0;
  virtual double distance(const double* model, const double* aux, int idx) const = 0;
  virtual void prepare(const double* /*model*/, double* /*aux*/) const {}
};

// opengv::sac::Ransac<P>::computeModel + SampleConsensusProblem::getSamples/drawIndexSample
RansacResult run_ransac(const Problem& prob, double threshold, int max_iterations,
                        double probability, int rng_policy) {
  RansacResult res;
  const int N = prob.size();
  const int ss = prob.sampleSize();
  Rnd rnd(rng_policy);
  std::vector<int> shuffled(N);  // setUniformIndices(N): indices_ = shuffled_indices_ = 0..N-1
  for (int i = 0; i < N; i++) shuffled[i] = i;
  const int max_sample_checks = 10;

  int iterations = 0;
  int n_best_inliers_count = -INT_MAX;
  double k = 1.0;
  std::vector<int> selection;
  double model[12], aux[16];
  unsigned skipped_count = 0;
  const unsigned max_skip = (unsigned)max_iterations * 10;

  while (iterations < k && skipped_count < max_skip) {
    // getSamples(iterations, selection)
    if (N < ss) {
      selection.clear();
      iterations = std::numeric_limits<int>::max();  // getSamples: "one of these will make it stop"
    } else {
      selection.resize(ss);
      bool good = false;
      for (int iter = 0; iter < max_sample_checks; ++iter) {
        // drawIndexSample
        for (int i = 0; i < ss; ++i)
          std::swap(shuffled[i], shuffled[i + (rnd() % (N - i))]);
        std::copy(shuffled.begin(), shuffled.begin() + ss, selection.begin());
        good = true;  // isSampleGood: not overridden by the problems used here
        if (good) break;
      }
      if (!good) selection.clear();
    }
    if (selection.empty()) break;
    if (!prob.computeModelCoefficients(selection, model)) {
      ++skipped_count;
      continue;
    }
    // countWithinDistance
    prob.prepare(model, aux);
    int n_inliers_count = 0;
    for (int i = 0; i < N; i++)
      if (prob.distance(model, aux, i) < threshold) n_inliers_count++;
    if (n_inliers_count > n_best_inliers_count) {
      n_best_inliers_count = n_inliers_count;
      res.model = selection;
      std::memcpy(res.coeff, model, sizeof(model));
      double w = (double)n_best_inliers_count / (double)N;
      double p_no_outliers = 1.0 - std::pow(w, (double)selection.size());
      p_no_outliers = (std::max)(std::numeric_limits<double>::epsilon(), p_no_outliers);
      p_no_outliers = (std::min)(1.0 - std::numeric_limits<double>::epsilon(), p_no_outliers);
      k = std::log(1.0 - probability) / std::log(p_no_outliers);
    }
    ++iterations;
    if (iterations > max_iterations) break;
  }
  res.iterations = iterations;
  if (res.model.empty()) {
    res.success = false;
    return res;
  }
  // selectWithinDistance(model_coefficients_, threshold_, inliers_)
  prob.prepare(res.coeff, aux);
  for (int i = 0; i < N; i++)
    if (prob.distance(res.coeff, aux, i) < threshold) res.inliers.push_back(i);
  res.success = true;
  return res;
}

// ---- TranslationOnlySacProblem -------------------------------------------------------------------
struct TranslationOnly : Problem {
  const double *f1, *f2, *R12;
  int n;
  int sampleSize() const override { return 2; }
  int size() const override { return n; }
  // relative_pose::twopt(adapter, unrotate = true, i0, i1); model = [R12 | t]
  bool computeModelCoefficients(const std::vector<int>& s, double* model) const override {
    const double* a1 = f1 + 3 * s[0];
    const double* b1 = f1 + 3 * s[1];
    double a2[3], b2[3];
    matvec3(R12, f2 + 3 * s[0], a2);
    matvec3(R12, f2 + 3 * s[1], b2);
    double n1[3], n2[3], t[3];
    cross3(a1, a2, n1);
    cross3(b1, b2, n2);
    cross3(n1, n2, t);
    const double nrm = std::sqrt(dot3(t, t));
    for (int i = 0; i < 3; i++) t[i] = t[i] / nrm;
    const double flow[3] = {a1[0] - a2[0], a1[1] - a2[1], a1[2] - a2[2]};
    if (dot3(flow, t) < 0)
      for (int i = 0; i < 3; i++) t[i] = -t[i];
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) model[r * 4 + c] = R12[r * 3 + c];
      model[r * 4 + 3] = t[r];
    }
    return true;
  }
  // aux = inverse solution [R^T | -R^T t] (3x4)
  void prepare(const double* model, double* aux) const override {
    double Rt[9], t[3] = {model[3], model[7], model[11]}, it[3];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) Rt[r * 3 + c] = model[c * 4 + r];
    matvec3(Rt, t, it);
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) aux[r * 4 + c] = Rt[r * 3 + c];
      aux[r * 4 + 3] = -it[r];
    }
  }
  // getSelectedDistancesToModel: triangulate2, reproject into both views, 1 - cos on each
  double distance(const double* model, const double* aux, int i) const override {
    const double R[9] = {model[0], model[1], model[2], model[4], model[5], model[6],
                         model[8], model[9], model[10]};
    const double t[3] = {model[3], model[7], model[11]};
    const double* a = f1 + 3 * i;
    const double* b = f2 + 3 * i;
    double bu[3];
    matvec3(R, b, bu);
    const double b0 = dot3(t, a), b1 = dot3(t, bu);
    const double A00 = dot3(a, a), A10 = dot3(a, bu), A01 = -A10, A11 = -dot3(bu, bu);
    const double det = A00 * A11 - A10 * A01;
    const double invdet = 1.0 / det;
    const double i00 = A11 * invdet, i10 = -A10 * invdet, i01 = -A01 * invdet, i11 = A00 * invdet;
    const double l0 = i00 * b0 + i01 * b1, l1 = i10 * b0 + i11 * b1;
    double p[3];
    for (int c = 0; c < 3; c++) {
      const double xm = l0 * a[c];
      const double xn = t[c] + l1 * bu[c];
      p[c] = (xm + xn) / 2;
    }
    double r1[3], r2[3];
    const double n1 = std::sqrt(dot3(p, p));
    for (int c = 0; c < 3; c++) r1[c] = p[c] / n1;
    for (int r = 0; r < 3; r++)
      r2[r] = ((aux[r * 4] * p[0] + aux[r * 4 + 1] * p[1]) + aux[r * 4 + 2] * p[2]) + aux[r * 4 + 3] * 1.0;
    const double n2 = std::sqrt(dot3(r2, r2));
    for (int c = 0; c < 3; c++) r2[c] = r2[c] / n2;
    const double e1 = 1.0 - dot3(a, r1);
    const double e2 = 1.0 - dot3(b, r2);
    return e1 + e2;
  }
};

// ---- PointCloudSacProblem (3-point Arun) -----------------------------------------------------------
// Eigen::JacobiSVD restated as a cyclic two-sided Jacobi on H^T H (V) + U = H V / sigma; the
// rotation only has to agree with Eigen's to rounding (parity with tolerance, see header).
void svd3(const double H[9], double U[9], double S[3], double V[9]) {
  double A[9];
  std::memcpy(A, H, sizeof(A));
  for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  // one-sided Jacobi (Hestenes): rotate column pairs of A until orthogonal
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int r = 0; r < 3; r++) {
          alpha += A[r * 3 + p] * A[r * 3 + p];
          beta += A[r * 3 + q] * A[r * 3 + q];
          gamma += A[r * 3 + p] * A[r * 3 + q];
        }
        off = std::max(off, std::fabs(gamma) / std::sqrt(std::max(alpha * beta, 1e-300)));
        if (std::fabs(gamma) <= 1e-300) continue;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double tt = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + tt * tt), s = c * tt;
        for (int r = 0; r < 3; r++) {
          const double ap = A[r * 3 + p], aq = A[r * 3 + q];
          A[r * 3 + p] = c * ap - s * aq;
          A[r * 3 + q] = s * ap + c * aq;
          const double vp = V[r * 3 + p], vq = V[r * 3 + q];
          V[r * 3 + p] = c * vp - s * vq;
          V[r * 3 + q] = s * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  for (int c = 0; c < 3; c++) S[c] = std::sqrt(A[c] * A[c] + A[3 + c] * A[3 + c] + A[6 + c] * A[6 + c]);
  // sort singular values descending (Eigen's convention)
  int order[3] = {0, 1, 2};
  for (int i = 1; i < 3; i++)   // stable insertion sort (the device kernel does the same)
    for (int j = i; j > 0 && S[order[j]] > S[order[j - 1]]; j--) std::swap(order[j], order[j - 1]);
  double A2[9], V2[9], S2[3];
  for (int c = 0; c < 3; c++) {
    S2[c] = S[order[c]];
    for (int r = 0; r < 3; r++) {
      A2[r * 3 + c] = A[r * 3 + order[c]];
      V2[r * 3 + c] = V[r * 3 + order[c]];
    }
  }
  std::memcpy(V, V2, sizeof(V2));
  std::memcpy(S, S2, sizeof(S2));
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) U[r * 3 + c] = S[c] > 1e-300 ? A2[r * 3 + c] / S[c] : 0.0;
  // complete U for (near-)zero singular values: third column = cross of the first two
  if (S[2] <= 1e-12 * std::max(S[0], 1e-300)) {
    double u0[3] = {U[0], U[3], U[6]}, u1[3] = {U[1], U[4], U[7]}, u2[3];
    cross3(u0, u1, u2);
    const double n = std::sqrt(dot3(u2, u2));
    if (n > 0)
      for (int r = 0; r < 3; r++) U[r * 3 + 2] = u2[r] / n;
  }
}

double det3(const double* M) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
         M[2] * (M[3] * M[7] - M[4] * M[6]);
}

struct PointCloud : Problem {
  const double *p1, *p2;
  int n;
  int sampleSize() const override { return 3; }
  int size() const override { return n; }
  // point_cloud::threept_arun = arun_complete over the sample
  bool computeModelCoefficients(const std::vector<int>& s, double* model) const override {
    double c1[3] = {0, 0, 0}, c2[3] = {0, 0, 0};
    for (int i : s)
      for (int c = 0; c < 3; c++) {
        c1[c] += p1[3 * i + c];
        c2[c] += p2[3 * i + c];
      }
    for (int c = 0; c < 3; c++) {
      c1[c] = c1[c] / (double)s.size();
      c2[c] = c2[c] / (double)s.size();
    }
    double H[9] = {0};
    for (int i : s) {
      double f[3], fp[3];
      for (int c = 0; c < 3; c++) {
        f[c] = p1[3 * i + c] - c1[c];
        fp[c] = p2[3 * i + c] - c2[c];
      }
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) H[r * 3 + c] += fp[r] * f[c];
    }
    double U[9], S[3], V[9], R[9];
    svd3(H, U, S, V);
    auto mulVUt = [&](const double* Vm) {
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
          R[r * 3 + c] = (Vm[r * 3] * U[c * 3] + Vm[r * 3 + 1] * U[c * 3 + 1]) + Vm[r * 3 + 2] * U[c * 3 + 2];
    };
    mulVUt(V);
    if (det3(R) < 0) {
      double Vp[9];
      std::memcpy(Vp, V, sizeof(Vp));
      for (int r = 0; r < 3; r++) Vp[r * 3 + 2] = -Vp[r * 3 + 2];
      mulVUt(Vp);
    }
    double Rc2[3];
    matvec3(R, c2, Rc2);
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) model[r * 4 + c] = R[r * 3 + c];
      model[r * 4 + 3] = c1[r] - Rc2[r];
    }
    return true;
  }
  double distance(const double* model, const double*, int i) const override {
    const double R[9] = {model[0], model[1], model[2], model[4], model[5], model[6],
                         model[8], model[9], model[10]};
    double q[3];
    matvec3(R, p2 + 3 * i, q);
    double d[3];
    for (int c = 0; c < 3; c++) d[c] = p1[3 * i + c] - (q[c] + model[c * 4 + 3]);
    return std::sqrt(dot3(d, d));
  }
};

// ---- CentralRelativePoseSacProblem(NISTER): 5-point problem ---------------------------------------
// relative_pose::fivept_nister (Nister, "An efficient solution to the five-point relative pose problem",
// PAMI 2004): null space of the 5 epipolar constraints, the ten cubic constraints on E = xA + yB + zC + D,
// Gauss-Jordan elimination in Nister's monomial order, the 3x3 polynomial matrix B(z), its determinant (degree
// 10), real roots by Sturm sequences.  OpenGV's own arithmetic (Eigen::JacobiSVD for the null space, its
// expansion of the determinant) is not reproduced: the essential matrices are the same up to rounding and up
// to the scale that the choice of null-space basis fixes (the translation norm of the model follows that scale
// in OpenGV too, i.e. it carries no information).  PARITY UNPINNED at the bit level; inlier decisions pinned by
// the scenes of tests/testTracker.cpp:704-802.
// linear polynomial (x, y, z, 1 coefficients) times linear -> quadratic [x2 y2 z2 xy xz yz x y z 1]
static inline void lin_mul(const double* a, const double* b, double* q) {
  q[0] = a[0] * b[0];
  q[1] = a[1] * b[1];
  q[2] = a[2] * b[2];
  q[3] = a[0] * b[1] + a[1] * b[0];
  q[4] = a[0] * b[2] + a[2] * b[0];
  q[5] = a[1] * b[2] + a[2] * b[1];
  q[6] = a[0] * b[3] + a[3] * b[0];
  q[7] = a[1] * b[3] + a[3] * b[1];
  q[8] = a[2] * b[3] + a[3] * b[2];
  q[9] = a[3] * b[3];
}
// quadratic times linear -> cubic in Nister's monomial order
// [x3 y3 x2y xy2 x2z x2 y2z y2 xyz xy | xz2 xz x yz2 yz y z3 z2 z 1]; out += sign * product
static inline void quadlin_acc(const double* q, const double* l, double sign, double* out) {
  out[0] += sign * (q[0] * l[0]);
  out[1] += sign * (q[1] * l[1]);
  out[2] += sign * (q[0] * l[1] + q[3] * l[0]);
  out[3] += sign * (q[1] * l[0] + q[3] * l[1]);
  out[4] += sign * (q[0] * l[2] + q[4] * l[0]);
  out[5] += sign * (q[0] * l[3] + q[6] * l[0]);
  out[6] += sign * (q[1] * l[2] + q[5] * l[1]);
  out[7] += sign * (q[1] * l[3] + q[7] * l[1]);
  out[8] += sign * ((q[3] * l[2] + q[4] * l[1]) + q[5] * l[0]);
  out[9] += sign * ((q[3] * l[3] + q[6] * l[1]) + q[7] * l[0]);
  out[10] += sign * (q[2] * l[0] + q[4] * l[2]);
  out[11] += sign * ((q[4] * l[3] + q[6] * l[2]) + q[8] * l[0]);
  out[12] += sign * (q[6] * l[3] + q[9] * l[0]);
  out[13] += sign * (q[2] * l[1] + q[5] * l[2]);
  out[14] += sign * ((q[5] * l[3] + q[7] * l[2]) + q[8] * l[1]);
  out[15] += sign * (q[7] * l[3] + q[9] * l[1]);
  out[16] += sign * (q[2] * l[2]);
  out[17] += sign * (q[2] * l[3] + q[8] * l[2]);
  out[18] += sign * (q[8] * l[3] + q[9] * l[2]);
  out[19] += sign * (q[9] * l[3]);
}

// four orthonormal null vectors of the 5 x 9 system Q: Gauss-Jordan with complete pivoting, the free columns in
// ascending order, modified Gram-Schmidt
static bool nullspace_5x9(double Q[5][9], double N[4][9]) {
  int pc[5];
  bool used[9] = {false, false, false, false, false, false, false, false, false};
  for (int s = 0; s < 5; s++) {
    int br = s, bc = -1;
    double bv = -1.0;
    for (int r = s; r < 5; r++)
      for (int c = 0; c < 9; c++) {
        if (used[c]) continue;
        const double v = std::fabs(Q[r][c]);
        if (v > bv) {
          bv = v;
          br = r;
          bc = c;
        }
      }
    if (!(bv > 1e-300)) return false;
    if (br != s)
      for (int c = 0; c < 9; c++) std::swap(Q[br][c], Q[s][c]);
    used[bc] = true;
    pc[s] = bc;
    const double inv = 1.0 / Q[s][bc];
    for (int c = 0; c < 9; c++) Q[s][c] *= inv;
    for (int r = 0; r < 5; r++) {
      if (r == s) continue;
      const double f = Q[r][bc];
      if (f == 0.0) continue;
      for (int c = 0; c < 9; c++) Q[r][c] -= f * Q[s][c];
    }
  }
  int j = 0;
  for (int fc = 0; fc < 9; fc++) {
    if (used[fc]) continue;
    for (int c = 0; c < 9; c++) N[j][c] = 0.0;
    N[j][fc] = 1.0;
    for (int s = 0; s < 5; s++) N[j][pc[s]] = -Q[s][fc];
    j++;
  }
  for (int a = 0; a < 4; a++) {
    for (int b = 0; b < a; b++) {
      double d = 0;
      for (int c = 0; c < 9; c++) d += N[a][c] * N[b][c];
      for (int c = 0; c < 9; c++) N[a][c] -= d * N[b][c];
    }
    double nn = 0;
    for (int c = 0; c < 9; c++) nn += N[a][c] * N[a][c];
    nn = std::sqrt(nn);
    if (!(nn > 1e-300)) return false;
    for (int c = 0; c < 9; c++) N[a][c] = N[a][c] / nn;
  }
  return true;
}

// univariate polynomials, coefficient c[k] of z^k
struct Poly1 {
  double c[12];
  int deg;
};
static Poly1 p1_make(int deg) {
  Poly1 r;
  for (double& v : r.c) v = 0.0;
  r.deg = deg;
  return r;
}
static Poly1 p1_mul(const Poly1& a, const Poly1& b) {
  Poly1 r = p1_make(a.deg + b.deg);
  for (int i = 0; i <= a.deg; i++)
    for (int j = 0; j <= b.deg; j++) r.c[i + j] += a.c[i] * b.c[j];
  return r;
}
static Poly1 p1_sub(const Poly1& a, const Poly1& b) {
  Poly1 r = p1_make(std::max(a.deg, b.deg));
  for (int i = 0; i <= a.deg; i++) r.c[i] += a.c[i];
  for (int i = 0; i <= b.deg; i++) r.c[i] -= b.c[i];
  return r;
}
static double p1_eval(const Poly1& a, double z) {
  double v = 0;
  for (int i = a.deg; i >= 0; i--) v = v * z + a.c[i];
  return v;
}
static void p1_trim(Poly1& a) {
  double m = 0;
  for (int i = 0; i <= a.deg; i++) m = std::max(m, std::fabs(a.c[i]));
  while (a.deg > 0 && std::fabs(a.c[a.deg]) <= 1e-14 * m) a.deg--;
}
// real roots by a Sturm chain (remainders rescaled by positive factors) + bisection
static int p1_real_roots(Poly1 p, double* roots /* up to 10 */) {
  p1_trim(p);
  if (p.deg < 1) return 0;
  Poly1 chain[12];
  int nc = 0;
  chain[nc++] = p;
  Poly1 d = p1_make(p.deg - 1);
  for (int i = 1; i <= p.deg; i++) d.c[i - 1] = i * p.c[i];
  chain[nc++] = d;
  while (chain[nc - 1].deg > 0 && nc < 12) {
    Poly1 a = chain[nc - 2];
    const Poly1& b = chain[nc - 1];
    for (int i = a.deg; i >= b.deg; i--) {   // a <- a mod b
      const double f = a.c[i] / b.c[b.deg];
      for (int j = 0; j <= b.deg; j++) a.c[i - b.deg + j] -= f * b.c[j];
      a.c[i] = 0.0;
    }
    a.deg = std::max(b.deg - 1, 0);
    double m = 0;
    for (int i = 0; i <= a.deg; i++) m = std::max(m, std::fabs(a.c[i]));
    if (m == 0) break;
    for (int i = 0; i <= a.deg; i++) a.c[i] = -a.c[i] / m;
    while (a.deg > 0 && std::fabs(a.c[a.deg]) <= 1e-13) a.deg--;
    chain[nc++] = a;
  }
  auto changes = [&](double z) {
    int n = 0, prev = 0;
    for (int i = 0; i < nc; i++) {
      const double v = p1_eval(chain[i], z);
      const int sgn = v > 0 ? 1 : v < 0 ? -1 : 0;
      if (sgn != 0) {
        if (prev != 0 && sgn != prev) n++;
        prev = sgn;
      }
    }
    return n;
  };
  double bound = 0;
  for (int i = 0; i < p.deg; i++) bound = std::max(bound, std::fabs(p.c[i] / p.c[p.deg]));
  bound += 1.0;
  int nroots = 0;
  struct Iv {
    double a, b;
    int na, nb;
  };
  Iv stack[64];
  int sp = 0;
  stack[sp++] = Iv{-bound, bound, changes(-bound), changes(bound)};
  while (sp > 0 && nroots < 10) {
    const Iv iv = stack[--sp];
    const int cnt = iv.na - iv.nb;
    if (cnt <= 0) continue;
    const double mid = 0.5 * (iv.a + iv.b);
    if (cnt == 1 || iv.b - iv.a < 1e-13 * std::max(1.0, std::fabs(mid))) {
      double a = iv.a, b = iv.b;
      double fa = p1_eval(p, a);
      for (int it = 0; it < 200 && b - a > 1e-16 * std::max(1.0, std::fabs(a) + std::fabs(b)); it++) {
        const double m = 0.5 * (a + b);
        if (m <= a || m >= b) break;
        const double fm = p1_eval(p, m);
        if (cnt == 1 && ((fa < 0) != (fm < 0))) {
          b = m;
        } else if (cnt == 1) {
          a = m;
          fa = fm;
        } else
          break;
      }
      roots[nroots++] = 0.5 * (a + b);
      continue;
    }
    const int nm = changes(mid);
    if (sp + 2 <= 64) {
      stack[sp++] = Iv{mid, iv.b, nm, iv.nb};
      stack[sp++] = Iv{iv.a, mid, iv.na, nm};
    }
  }
  std::sort(roots, roots + nroots);
  return nroots;
}

// relative_pose::fivept_nister(adapter, indices): essential matrices E (row-major 3x3, f1^T E f2 = 0), up to 10
static int fivept_nister(const double* f1, const double* f2, const int* idx5, double E_out[][9]) {
  double Q[5][9];
  for (int i = 0; i < 5; i++) {
    const double* f = f1 + 3 * idx5[i];
    const double* fp = f2 + 3 * idx5[i];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) Q[i][3 * r + c] = f[c] * fp[r];
  }
  // E = x A + y B + z C + D with four orthonormal null vectors
  double N[4][9];
  if (!nullspace_5x9(Q, N)) return 0;
  double El[3][3][4];   // entry (r, c) as a linear polynomial
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++)
      for (int k = 0; k < 4; k++) El[r][c][k] = N[k][3 * r + c];
  double M[10][20];
  for (int r = 0; r < 10; r++)
    for (int c = 0; c < 20; c++) M[r][c] = 0.0;
  {   // det(E)
    double m0[10], m1[10], m2[10], ta[10], tb[10];
    lin_mul(El[1][1], El[2][2], ta);
    lin_mul(El[1][2], El[2][1], tb);
    for (int i = 0; i < 10; i++) m0[i] = ta[i] - tb[i];
    lin_mul(El[1][0], El[2][2], ta);
    lin_mul(El[1][2], El[2][0], tb);
    for (int i = 0; i < 10; i++) m1[i] = ta[i] - tb[i];
    lin_mul(El[1][0], El[2][1], ta);
    lin_mul(El[1][1], El[2][0], tb);
    for (int i = 0; i < 10; i++) m2[i] = ta[i] - tb[i];
    quadlin_acc(m0, El[0][0], 1.0, M[0]);
    quadlin_acc(m1, El[0][1], -1.0, M[0]);
    quadlin_acc(m2, El[0][2], 1.0, M[0]);
  }
  {   // 2 E E^T E - trace(E E^T) E
    double EEt[3][3][10], tr[10], t[10];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) {
        lin_mul(El[r][0], El[c][0], EEt[r][c]);
        lin_mul(El[r][1], El[c][1], t);
        for (int i = 0; i < 10; i++) EEt[r][c][i] += t[i];
        lin_mul(El[r][2], El[c][2], t);
        for (int i = 0; i < 10; i++) EEt[r][c][i] += t[i];
      }
    for (int i = 0; i < 10; i++) tr[i] = (EEt[0][0][i] + EEt[1][1][i]) + EEt[2][2][i];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) {
        double* row = M[1 + 3 * r + c];
        for (int k = 0; k < 3; k++) quadlin_acc(EEt[r][k], El[k][c], 2.0, row);
        quadlin_acc(tr, El[r][c], -1.0, row);
      }
  }
  // Gauss-Jordan with partial pivoting on the first ten columns
  for (int col = 0; col < 10; col++) {
    int piv = col;
    for (int r = col + 1; r < 10; r++)
      if (std::fabs(M[r][col]) > std::fabs(M[piv][col])) piv = r;
    if (std::fabs(M[piv][col]) < 1e-300) return 0;
    if (piv != col)
      for (int c = 0; c < 20; c++) std::swap(M[piv][c], M[col][c]);
    const double inv = 1.0 / M[col][col];
    for (int c = 0; c < 20; c++) M[col][c] *= inv;
    for (int r = 0; r < 10; r++) {
      if (r == col) continue;
      const double f = M[r][col];
      if (f == 0.0) continue;
      for (int c = 0; c < 20; c++) M[r][c] -= f * M[col][c];
    }
  }
  // rows <k> = <e> - z<f>, <l> = <g> - z<h>, <m> = <i> - z<j>: B(z) [x y 1]^T = 0
  Poly1 B[3][3];
  for (int q = 0; q < 3; q++) {
    const double* e = M[4 + 2 * q] + 10;
    const double* f = M[5 + 2 * q] + 10;
    for (int v = 0; v < 2; v++) {   // x and y columns: degree 3
      Poly1 b = p1_make(3);
      b.c[0] = e[3 * v + 2];
      b.c[1] = e[3 * v + 1] - f[3 * v + 2];
      b.c[2] = e[3 * v] - f[3 * v + 1];
      b.c[3] = -f[3 * v];
      B[q][v] = b;
    }
    Poly1 b = p1_make(4);
    b.c[0] = e[9];
    b.c[1] = e[8] - f[9];
    b.c[2] = e[7] - f[8];
    b.c[3] = e[6] - f[7];
    b.c[4] = -f[6];
    B[q][2] = b;
  }
  const Poly1 det = p1_sub(
      p1_sub(p1_mul(B[0][0], p1_sub(p1_mul(B[1][1], B[2][2]), p1_mul(B[1][2], B[2][1]))),
             p1_mul(B[0][1], p1_sub(p1_mul(B[1][0], B[2][2]), p1_mul(B[1][2], B[2][0])))),
      p1_mul(p1_mul(B[0][2], p1_sub(p1_mul(B[1][1], B[2][0]), p1_mul(B[1][0], B[2][1]))), [] {
        Poly1 one = p1_make(0);
        one.c[0] = 1.0;
        return one;
      }()));
  double roots[10];
  const int nr = p1_real_roots(det, roots);
  int ne = 0;
  for (int k = 0; k < nr; k++) {
    const double z = roots[k];
    double b[3][3];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) b[r][c] = p1_eval(B[r][c], z);
    // [x y]: the best conditioned pair of rows
    double bestdet = 0;
    int r0 = 0, r1 = 1;
    static const int pairs[3][2] = {{0, 1}, {0, 2}, {1, 2}};
    for (const auto& pr : pairs) {
      const double dd = b[pr[0]][0] * b[pr[1]][1] - b[pr[0]][1] * b[pr[1]][0];
      if (std::fabs(dd) > std::fabs(bestdet)) {
        bestdet = dd;
        r0 = pr[0];
        r1 = pr[1];
      }
    }
    if (bestdet == 0) continue;
    const double x = (-b[r0][2] * b[r1][1] + b[r1][2] * b[r0][1]) / bestdet;
    const double y = (-b[r0][0] * b[r1][2] + b[r1][0] * b[r0][2]) / bestdet;
    // the 9-vector is column-major in OpenGV (Eigen): E(i, j) = e[i + 3 j], i.e. f1^T E f2 = 0, E = [t12]x R12
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        const int a = i + 3 * j;
        E_out[ne][3 * i + j] = ((x * N[0][a] + y * N[1][a]) + z * N[2][a]) + N[3][a];
      }
    ne++;
  }
  return ne;
}

struct CentralRelativePose : Problem {   // CentralRelativePoseSacProblem(adapter, NISTER)
  const double *f1, *f2;
  int n;
  int sampleSize() const override { return 5 + 3; }
  int size() const override { return n; }
  static void inverse_tf(const double* T, double* inv) {
    double Rt[9], t[3] = {T[3], T[7], T[11]}, it[3];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) Rt[r * 3 + c] = T[c * 4 + r];
    matvec3(Rt, t, it);
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) inv[r * 4 + c] = Rt[r * 3 + c];
      inv[r * 4 + 3] = -it[r];
    }
  }
  // the four [R | t] decompositions of an essential matrix (CentralRelativePoseSacProblem.cpp, NISTER case)
  static void decompose(const double* E, double T[4][12]) {
    double U[9], S[3], V[9];
    svd3(E, U, S, V);
    const double W[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1}, Wt[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
    auto uwvt = [&](const double* Wm, double* R) {
      double UW[9];
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) UW[r * 3 + c] = (U[r * 3] * Wm[c] + U[r * 3 + 1] * Wm[3 + c]) + U[r * 3 + 2] * Wm[6 + c];
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) R[r * 3 + c] = (UW[r * 3] * V[c * 3] + UW[r * 3 + 1] * V[c * 3 + 1]) + UW[r * 3 + 2] * V[c * 3 + 2];
      if (det3(R) < 0)
        for (int i = 0; i < 9; i++) R[i] = -R[i];
    };
    double Ra[9], Rb[9];
    uwvt(W, Ra);
    uwvt(Wt, Rb);
    const double scale = S[0];
    const double ta[3] = {scale * U[2], scale * U[5], scale * U[8]};
    const double* Rs[4] = {Ra, Rb, Ra, Rb};
    const double sg[4] = {1, 1, -1, -1};
    for (int j = 0; j < 4; j++)
      for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) T[j][r * 4 + c] = Rs[j][r * 3 + c];
        T[j][r * 4 + 3] = sg[j] * ta[r];
      }
  }
  double reproj(const double* T, const double* inv, int i) const {   // same terms as TranslationOnly::distance
    TranslationOnly tmp;
    tmp.f1 = f1;
    tmp.f2 = f2;
    tmp.R12 = nullptr;
    tmp.n = n;
    return tmp.distance(T, inv, i);
  }
  bool computeModelCoefficients(const std::vector<int>& s, double* model) const override {
    double Es[10][9];
    const int ne = fivept_nister(f1, f2, s.data(), Es);
    double bestQuality = 1000000.0;
    int best_i = -1, best_j = -1;
    for (int i = 0; i < ne; i++) {
      double T[4][12];
      decompose(Es[i], T);
      for (int j = 0; j < 4; j++) {
        double inv[12];
        inverse_tf(T[j], inv);
        double quality = 0.0;
        for (int k = 0; k < sampleSize(); k++) quality += reproj(T[j], inv, s[k]);
        if (quality < bestQuality) {
          bestQuality = quality;
          best_i = i;
          best_j = j;
        }
      }
    }
    if (best_i == -1) return false;
    double T[4][12];
    decompose(Es[best_i], T);
    std::memcpy(model, T[best_j], sizeof(double) * 12);
    return true;
  }
  void prepare(const double* model, double* aux) const override { inverse_tf(model, aux); }
  double distance(const double* model, const double* aux, int i) const override { return reproj(model, aux, i); }
};

#include "opengv_epnp.inl"

}  // namespace

RansacResult ransac_absolute_pose_epnp(const double* bearings, const double* points, int n, double threshold,
                                       int max_iterations, double probability, int rng_policy) {
  AbsolutePoseEpnp p;
  p.f = bearings;
  p.p = points;
  p.n = n;
  return run_ransac(p, threshold, max_iterations, probability, rng_policy);
}

RansacResult ransac_absolute_pose_kneip(const double* bearings, const double* points, int n, double threshold,
                                        int max_iterations, double probability, int rng_policy) {
  AbsolutePoseKneip p;
  p.f = bearings;
  p.p = points;
  p.n = n;
  return run_ransac(p, threshold, max_iterations, probability, rng_policy);
}

int p3p_kneip_solutions(const double* bearings, const double* points, const int* idx3, double* sol /* 4 x 12 */) {
  double s[4][12];
  const int n = p3p_kneip(bearings, points, idx3, s);
  std::memcpy(sol, s, sizeof(double) * 12 * n);
  return n;
}

void quartic_roots(const double* p5, double* roots4) { o4_roots(p5, roots4); }

int epnp(const double* bearings, const double* points, const int* idx, int n, double model[12]) {
  if (n < 4) return 0;
  epnp_transformation(bearings, points, idx, n, model);
  return 1;
}

RansacResult ransac_translation_only(const double* f1, const double* f2, int n, const double R12[9],
                                     double threshold, int max_iterations, double probability,
                                     int rng_policy) {
  TranslationOnly p;
  p.f1 = f1;
  p.f2 = f2;
  p.R12 = R12;
  p.n = n;
  return run_ransac(p, threshold, max_iterations, probability, rng_policy);
}

RansacResult ransac_point_cloud(const double* p1, const double* p2, int n, double threshold,
                                int max_iterations, double probability, int rng_policy) {
  PointCloud p;
  p.p1 = p1;
  p.p2 = p2;
  p.n = n;
  return run_ransac(p, threshold, max_iterations, probability, rng_policy);
}

RansacResult ransac_central_relative_pose_nister(const double* f1, const double* f2, int n, double threshold,
                                                 int max_iterations, double probability, int rng_policy) {
  CentralRelativePose p;
  p.f1 = f1;
  p.f2 = f2;
  p.n = n;
  return run_ransac(p, threshold, max_iterations, probability, rng_policy);
}

int fivept_nister_essentials(const double* f1, const double* f2, const int* idx5, double* E_out /* 10 x 9 */) {
  double Es[10][9];
  const int n = fivept_nister(f1, f2, idx5, Es);
  std::memcpy(E_out, Es, sizeof(double) * 9 * n);
  return n;
}

}  // namespace opengv_re
