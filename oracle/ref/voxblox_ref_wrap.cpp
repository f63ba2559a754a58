// C entry points over voxblox's OWN sources — integrator_utils.cc (RayCaster, ThreadSafeIndex and its bit-reversal
// table) and the inline helpers of common.h / block_hash.h — compiled where they lie under /root/reference against
// the stand-ins of oracle/ref/vbx_shim (Eigen, glog, kindr type names) -> oracle/_ref/libvoxblox_ref.so.
// tests/test_oracle_pinned.py checks the restatement in oracle/tsdf_voxblox.c against these.
// TEST INFRASTRUCTURE ONLY: nothing under plvs_amd/ may call into this file.
#include <cstdint>
#include <cstring>

#include "voxblox/core/block_hash.h"
#include "voxblox/core/common.h"
#include "voxblox/integrator/integrator_utils.h"

extern "C" {

// RayCaster(origin, point_G, is_clearing_ray, voxel_carving_enabled, max_ray_length_m, voxel_size_inv,
// truncation_distance) as SimpleTsdfIntegrator::integrateFunction builds it (tsdf_integrator.cc:296-299), walked
// with nextRayIndex; returns the number of voxels.
int ref_voxblox_raycast(const float* origin, const float* point_G, int is_clearing, int carving, float max_ray,
                        float voxel_size_inv, float truncation, int32_t* out, int cap) {
  voxblox::RayCaster ray_caster(voxblox::Point(origin[0], origin[1], origin[2]),
                                voxblox::Point(point_G[0], point_G[1], point_G[2]), is_clearing != 0, carving != 0,
                                max_ray, voxel_size_inv, truncation);
  voxblox::AnyIndex idx;
  int n = 0;
  while (ray_caster.nextRayIndex(&idx)) {
    if (n < cap) {
      out[3 * n] = idx.x();
      out[3 * n + 1] = idx.y();
      out[3 * n + 2] = idx.z();
    }
    ++n;
  }
  return n;
}

// The order ThreadSafeIndex hands out the n points of a cloud (one thread).
void ref_voxblox_mixed_order(int n, int64_t* out) {
  voxblox::ThreadSafeIndex index_getter((size_t)n);
  size_t idx;
  int k = 0;
  while (index_getter.getNextIndex(&idx)) out[k++] = (int64_t)idx;
}

uint32_t ref_voxblox_blend(uint32_t c1, float w1, uint32_t c2, float w2) {
  voxblox::Color a, b;
  std::memcpy(&a, &c1, 4);
  std::memcpy(&b, &c2, 4);
  const voxblox::Color r = voxblox::Color::blendTwoColors(a, w1, b, w2);
  uint32_t out;
  std::memcpy(&out, &r, 4);
  return out;
}

// Block index, local voxel index (16 voxels per side) and the block map's hash of a global voxel index.
void ref_voxblox_indices(const int32_t* g, int32_t* block, int32_t* local, uint64_t* hash) {
  const voxblox::AnyIndex gi(g[0], g[1], g[2]);
  const voxblox::FloatingPoint voxels_per_side_inv = 1.0 / 16;   // tsdf_integrator.cc:19
  const voxblox::BlockIndex b = voxblox::getBlockIndexFromGlobalVoxelIndex(gi, voxels_per_side_inv);
  const voxblox::VoxelIndex l = voxblox::getLocalFromGlobalVoxelIndex(gi, 16);
  for (int k = 0; k < 3; ++k) {
    block[k] = b[k];
    local[k] = l[k];
  }
  *hash = (uint64_t)voxblox::AnyIndexHash()(b);
}

// The order in which the reference's own map type — AnyIndexHashMapType<T>::type, what MergedTsdfIntegrator::bundleRays
// fills and integrateVoxels walks (tsdf_integrator.cc:337-341, :448-470) — hands back keys inserted in the given
// sequence (operator[] semantics).  order_out: first-insertion positions of the distinct keys, in iteration order.
int ref_voxblox_bundle_order(const int32_t* g, int n, int32_t* order_out) {
  voxblox::AnyIndexHashMapType<int32_t>::type m;
  for (int i = 0; i < n; ++i) {
    const voxblox::AnyIndex k(g[3 * i], g[3 * i + 1], g[3 * i + 2]);
    if (m.find(k) == m.end()) m[k] = i;
  }
  int c = 0;
  for (const auto& kv : m) order_out[c++] = kv.second;
  return c;
}

}  // extern "C"
