// C entry points over the reference's OWN open_chisel sources — Raycast.cpp, DistVoxel, ColorVoxel,
// QuadraticTruncator, ConstantWeighter — compiled where they lie under /root/reference against the Eigen stand-in of
// oracle/ref/eigen_shim (oracle/ref/Makefile -> oracle/_ref/libchisel_ref.so).  tests/test_oracle_pinned.py checks
// the restatement in oracle/tsdf_chisel.c against these on random and adversarial inputs: the traversal's control
// flow and tie rules, the voxel update arithmetic and the truncation / weight formulas then come from the
// reference's source, not from a reading of it.
// TEST INFRASTRUCTURE ONLY: nothing under plvs_amd/ may call into this file.
#include <climits>
#include <cstdint>
#include <memory>

#include <open_chisel/ColorVoxel.h>
#include <open_chisel/DistVoxel.h>
#include <open_chisel/geometry/Geometry.h>
#include <open_chisel/geometry/Raycast.h>
#include <open_chisel/truncation/Truncator.h>   // (QuadraticTruncator.h relies on its includer for the base class)
#include <open_chisel/truncation/QuadraticTruncator.h>
#include <open_chisel/weighting/ConstantWeighter.h>

// Chisel.cpp:447, `const float diag = 2.0f * sqrt(3.0f) * resolution;`, as that translation unit sees it: inside
// namespace chisel, after <open_chisel/Chisel.h> (Eigen: <cmath>, never <math.h>), <iostream>, <vector>,
// <unordered_map>.  With only <cmath> in the include chain the unqualified call finds the C library's
// ::sqrt(double) — libstdc++ adds the float overload to the global namespace only when <math.h> itself is
// included — so the product is formed in double and rounded once; oracle/tsdf_chisel.c assumes exactly that.
#include <iostream>
#include <unordered_map>
#include <vector>
namespace chisel {
static_assert(sizeof(sqrt(3.0f)) == sizeof(double), "unqualified sqrt(3.0f) binds to ::sqrt(double) here");
static float truncation_floor(float resolution) {
  const float diag = 2.0f * sqrt(3.0f) * resolution;
  return diag;
}
}  // namespace chisel

extern "C" {

float ref_chisel_diag(float resolution) { return chisel::truncation_floor(resolution); }

// Raycast(start, end, minVal, maxVal, &voxels) with the bounds Chisel.cpp:447-448 passes; returns the number of
// voxels (the first `cap` are written to out as x, y, z).
int ref_chisel_raycast(const float* start, const float* end, int32_t* out, int cap) {
  const chisel::Point3 minVal(-INT_MAX, -INT_MAX, -INT_MAX), maxVal(INT_MAX, INT_MAX, INT_MAX);
  chisel::Point3List voxels;
  Raycast(chisel::Vec3(start[0], start[1], start[2]), chisel::Vec3(end[0], end[1], end[2]), minVal, maxVal, &voxels);
  const int n = (int)voxels.size();
  for (int i = 0; i < n && i < cap; ++i) {
    out[3 * i + 0] = voxels[i].x();
    out[3 * i + 1] = voxels[i].y();
    out[3 * i + 2] = voxels[i].z();
  }
  return n;
}

// DistVoxel::Integrate on a voxel holding (sdf, weight).
void ref_chisel_dist_integrate(float* sdf, float* weight, float dist_update, float weight_update) {
  chisel::DistVoxel v;
  v.SetSDF(*sdf);
  v.SetWeight(*weight);
  v.Integrate(dist_update, weight_update);
  *sdf = v.GetSDF();
  *weight = v.GetWeight();
}

// ColorVoxel::IntegrateSimple on a voxel holding rgbw[0..3].
void ref_chisel_colour_integrate_simple(uint8_t* rgbw, uint8_t r, uint8_t g, uint8_t b, uint8_t weight_update) {
  chisel::ColorVoxel v;
  v.SetRed(rgbw[0]);
  v.SetGreen(rgbw[1]);
  v.SetBlue(rgbw[2]);
  v.SetWeight(rgbw[3]);
  v.IntegrateSimple(r, g, b, weight_update);
  rgbw[0] = v.GetRed();
  rgbw[1] = v.GetGreen();
  rgbw[2] = v.GetBlue();
  rgbw[3] = v.GetWeight();
}

// ColorVoxel::Integrate (the flavour IntegrateWorldPointCloudWithNormals calls) on a voxel holding rgbw[0..3].
void ref_chisel_colour_integrate(uint8_t* rgbw, uint8_t r, uint8_t g, uint8_t b, uint8_t weight_update) {
  chisel::ColorVoxel v;
  v.SetRed(rgbw[0]);
  v.SetGreen(rgbw[1]);
  v.SetBlue(rgbw[2]);
  v.SetWeight(rgbw[3]);
  v.Integrate(r, g, b, weight_update);
  rgbw[0] = v.GetRed();
  rgbw[1] = v.GetGreen();
  rgbw[2] = v.GetBlue();
  rgbw[3] = v.GetWeight();
}

float ref_chisel_truncation(float quadratic, float linear, float constant, float scale, float reading) {
  return chisel::QuadraticTruncator(quadratic, linear, constant, scale).GetTruncationDistance(reading);
}

float ref_chisel_weight(float weight, float surface_dist, float truncation) {
  return chisel::ConstantWeighter(weight).GetWeight(surface_dist, truncation);
}

}  // extern "C"
