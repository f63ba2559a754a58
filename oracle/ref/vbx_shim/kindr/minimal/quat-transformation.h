// Stand-in for minkindr's QuatTransformationTemplate as voxblox's integrators use it: getPosition() and
// transformation * point.  TEST INFRASTRUCTURE ONLY.
//
// The rotation is held as a unit quaternion (w, x, y, z) and applied as Eigen 3.3's QuaternionBase::_transformVector
// does (minkindr quat-transformation-inl.h:159-162 -> rotation-quaternion-inl.h -> Eigen Quaternion.h):
//   uv = 2 * (q.vec x v);  result = v + q.w * uv + q.vec x uv,   then + position
// with the sums evaluated left to right, as the expression template does for fixed-size vectors of three.  The
// conversion of a pose MATRIX into the quaternion (the kindr constructor TsdfServer::insertPointCloud calls) is not
// part of this stand-in: the wrappers hand the quaternion over (from the restatement in oracle/tsdf_voxblox.c), so
// what the compiled reference sources pin is the integrator arithmetic, not that conversion.
#pragma once
#include <Eigen/Core>
namespace kindr {
namespace minimal {
template <class Scalar>
struct RotationQuaternionTemplate {};
template <class Scalar>
struct QuatTransformationTemplate {
  typedef Eigen::Matrix<Scalar, 4, 4> TransformationMatrix;
  typedef Eigen::Matrix<Scalar, 3, 1> Vec3;
  Scalar q[4];   // w, x, y, z
  Vec3 t;
  QuatTransformationTemplate() : q{1, 0, 0, 0}, t() {}
  QuatTransformationTemplate(const Scalar* quat_wxyz, const Scalar* position)
      : q{quat_wxyz[0], quat_wxyz[1], quat_wxyz[2], quat_wxyz[3]}, t(position[0], position[1], position[2]) {}
  const Vec3& getPosition() const { return t; }
  Vec3 operator*(const Vec3& v) const {
    const Scalar w = q[0], x = q[1], y = q[2], z = q[3];
    Scalar u0 = y * v.z() - z * v.y(), u1 = z * v.x() - x * v.z(), u2 = x * v.y() - y * v.x();
    u0 += u0; u1 += u1; u2 += u2;
    const Scalar c0 = y * u2 - z * u1, c1 = z * u0 - x * u2, c2 = x * u1 - y * u0;
    return Vec3(((v.x() + w * u0) + c0) + t.x(), ((v.y() + w * u1) + c1) + t.y(), ((v.z() + w * u2) + c2) + t.z());
  }
};
}  // namespace minimal
}  // namespace kindr
