// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
// CPU restatement of the parts of OpenGV that Kimera-VIO's geometric outlier rejection calls
// (Tracker::runRansac, include/kimera-vio/frontend/Tracker.h:247-296, with the problem types of
// include/kimera-vio/frontend/Tracker-definitions.h:43-61).
//
// OpenGV is an un-vendored third-party dependency of the reference (fork marcusabate/opengv,
// no version pin: Dockerfile_20_04:55); its sources are not in /root/reference.  What is restated
// here is its published algorithm:
//   opengv::sac::Ransac<P>::computeModel           (opengv/sac/implementation/Ransac.hpp)
//   opengv::sac::SampleConsensusProblem<M>         (getSamples / drawIndexSample / rnd,
//                                                   opengv/sac/implementation/SampleConsensusProblem.hpp)
//   sac_problems::relative_pose::TranslationOnlySacProblem  (2-point, rotation known)
//   relative_pose::twopt, triangulation::triangulate2
//   sac_problems::point_cloud::PointCloudSacProblem + point_cloud::threept_arun (3-point Arun)
// PARITY UNPINNED at the bit level (no OpenGV build is reachable here).  It is anchored on the
// reference's own known-answer tests for these call sites (tests/testTracker.cpp:704-1186: every
// synthetic inlier kept, every synthetic outlier rejected, translation recovered) — see
// tests/test_oracle_kat.py.
#pragma once
#include <cstdint>
#include <vector>

namespace opengv_re {

// std::uniform_int_distribution<int>(0, INT_MAX) over std::mt19937 is implementation defined:
//   RNG_LIBSTDCXX_PRE11: libstdc++ <= 10 (GCC 9.4 of the reference's Ubuntu 20.04 image):
//                        "downscaling" = redraw while the 32-bit output is >= 2^31
//   RNG_LIBSTDCXX_11   : libstdc++ >= 11 (Lemire's method) = 32-bit output >> 1, never redraws
enum { RNG_LIBSTDCXX_PRE11 = 0, RNG_LIBSTDCXX_11 = 1 };

struct Mt19937 {  // std::mt19937 (ISO C++ [rand.eng.mers], bit-exact by definition)
  uint32_t mt[624];
  int idx;
  void seed(uint32_t s);
  uint32_t next();
};

struct RansacResult {
  bool success = false;       // Ransac::computeModel() return value
  int iterations = 0;         // Ransac::iterations_
  std::vector<int> model;     // Ransac::model_ (sample of the best model)
  double coeff[12] = {0};     // Ransac::model_coefficients_ (3x4 [R|t], row-major)
  std::vector<int> inliers;   // Ransac::inliers_
};

// opengv::sac::Ransac<TranslationOnlySacProblem>::computeModel with
// TranslationOnlySacProblem(adapter(f1, f2) + setR12(R12), randomSeed = false)
// f1, f2: n x 3 bearing vectors (reference / current frame)
RansacResult ransac_translation_only(const double* f1, const double* f2, int n, const double R12[9],
                                     double threshold, int max_iterations, double probability,
                                     int rng_policy);

// opengv::sac::Ransac<PointCloudSacProblem>::computeModel (3-point Arun), p1, p2: n x 3 points
RansacResult ransac_point_cloud(const double* p1, const double* p2, int n, double threshold,
                                int max_iterations, double probability, int rng_policy);

// opengv::sac::Ransac<CentralRelativePoseSacProblem(NISTER)>::computeModel: 5-point problem, sample size 5 + 3
RansacResult ransac_central_relative_pose_nister(const double* f1, const double* f2, int n, double threshold,
                                                 int max_iterations, double probability, int rng_policy);

// opengv::sac::Ransac<AbsolutePoseSacProblem(EPNP)>::computeModel (Tracker::pnp, pnp_algorithm 3): sample size 6,
// bearings: n x 3 camera-frame bearing vectors, points: n x 3 world points; coeff = world_T_camera [R | t]
RansacResult ransac_absolute_pose_epnp(const double* bearings, const double* points, int n, double threshold,
                                       int max_iterations, double probability, int rng_policy);
// opengv::sac::Ransac<AbsolutePoseSacProblem(KNEIP)>::computeModel (pnp_algorithm 1): sample size 4 = Kneip's P3P on
// three correspondences, the fourth picks among its solutions
RansacResult ransac_absolute_pose_kneip(const double* bearings, const double* points, int n, double threshold,
                                        int max_iterations, double probability, int rng_policy);
// absolute_pose::p3p_kneip on idx3[0..3) (test hook): up to four world_T_camera 3x4 row-major; math::o4_roots
int p3p_kneip_solutions(const double* bearings, const double* points, const int* idx3, double* sol);
void quartic_roots(const double* p5, double* roots4);
// absolute_pose::epnp(adapter, indices) on idx[0..n) (test hook), model = world_T_camera 3x4 row-major
int epnp(const double* bearings, const double* points, const int* idx, int n, double model[12]);

// relative_pose::fivept_nister on five correspondences (test hook): up to 10 essential matrices, row-major
int fivept_nister_essentials(const double* f1, const double* f2, const int* idx5, double* E_out);

}  // namespace opengv_re
