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

}  // namespace

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

}  // namespace opengv_re
