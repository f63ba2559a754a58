// Declaration-only stand-in for open_chisel's MarchingCubes.h, placed ahead of the reference's include directory
// when its MarchingCubes.cpp — which DEFINES the triangle table and the edge index pairs — is compiled into
// oracle/_ref/libchisel_ref.so (oracle/ref/Makefile): the real header's inline members need far more of Eigen than
// the stand-in of oracle/ref/eigen_shim provides, the two tables need none.  TEST INFRASTRUCTURE ONLY.
#pragma once
namespace chisel {
class MarchingCubes {
 public:
  static const int triangleTable[256][16];
  static const int edgeIndexPairs[12][2];
  MarchingCubes();
  virtual ~MarchingCubes();
};
}  // namespace chisel
