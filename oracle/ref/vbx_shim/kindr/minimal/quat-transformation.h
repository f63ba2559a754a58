// Type names only: voxblox's common.h typedefs kindr's transformation, the sources compiled into
// oracle/_ref/libvoxblox_ref.so never use it.  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <Eigen/Core>
namespace kindr {
namespace minimal {
template <class Scalar>
struct RotationQuaternionTemplate {};
template <class Scalar>
struct QuatTransformationTemplate {
  typedef Eigen::Matrix<Scalar, 4, 4> TransformationMatrix;
};
}  // namespace minimal
}  // namespace kindr
