// Minimal stand-in for <gtsam/geometry/Rot3.h> (gtsam::Rot3::matrix(), gtsam::Vector3 with operator()):
// TEST INFRASTRUCTURE for tests/cpp/shim_check.cpp, not part of the product.
#pragma once
namespace gtsam {
struct Matrix3 {
  double m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double operator()(int i, int j) const { return m[3 * i + j]; }
  double& operator()(int i, int j) { return m[3 * i + j]; }
};
struct Vector3 {
  double v[3] = {0, 0, 0};
  double operator()(int i) const { return v[i]; }
  double& operator()(int i) { return v[i]; }
};
struct Rot3 {
  Matrix3 R;
  Rot3() = default;
  explicit Rot3(const Matrix3& M) : R(M) {}
  Matrix3 matrix() const { return R; }
};
using Point3 = Vector3;
struct Pose3 {   // stand-in for <gtsam/geometry/Pose3.h>
  Rot3 R;
  Point3 t;
  Pose3() = default;
  Pose3(const Rot3& r, const Point3& tt) : R(r), t(tt) {}
  const Rot3& rotation() const { return R; }
  const Point3& translation() const { return t; }
};
}  // namespace gtsam
