// oracle/tsdf_voxblox_merged.cpp — CPU restatement of voxblox's MergedTsdfIntegrator::integratePointCloud
// (Thirdparty/voxblox/src/integrator/tsdf_integrator.cc:329-492) with integrator_threads = 1, the bundling part.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under plvs_amd/ may call into this file.
//
//   bundleRays      :361-391   points in ThreadSafeIndex's mixed order, grouped by the voxel T_G_C * point_C ends in:
//                              voxel_map[voxel_index].push_back(point_idx) (clear_map for clearing rays).  The maps are
//                              AnyIndexHashMapType<AlignedVector<size_t>>::type = std::unordered_map with AnyIndexHash
//                              (core/block_hash.h:15-34); integrateVoxels (:448-470) walks them with begin() / ++it.
//   integrateVoxel  :393-446   the bundle's points folded into one: merged_point_C = (merged * W + p * w) / (W + w),
//                              Color::blendTwoColors, W += w (only the first point of a clearing bundle); one ray.
// The order of the bundles IS the iteration order of a std::unordered_map filled in that sequence: the map below is one,
// with the reference's hash (pinned against the reference's own container type compiled into oracle/_ref:
// tests/test_oracle_pinned.py), so it iterates as the reference's does on the same libstdc++.  The ray and the voxel
// update of each bundle are oracle_voxblox_integrate_bundles (oracle/tsdf_voxblox.c).
#include <array>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

struct oracle_voxblox;
extern "C" {
int oracle_voxblox_point_kind(const oracle_voxblox* o, const float* pC);
void oracle_voxblox_point_voxel(const oracle_voxblox* o, const float* Twc, const float* pC, int32_t* g);
float oracle_voxblox_point_weight(const float* pC);
uint32_t oracle_voxblox_blend(uint32_t c1, float w1, uint32_t c2, float w2);
void oracle_voxblox_mixed_order(int n, int64_t* out);
void oracle_voxblox_integrate_bundles(oracle_voxblox* o, const float* merged_C, const uint32_t* colours, const float* weights,
                                      const uint8_t* clearing, int n, const float* Twc);
}

namespace {

using Key = std::array<int32_t, 3>;
struct AnyIndexHash {   // core/block_hash.h:15-26
  std::size_t operator()(const Key& k) const {
    return (static_cast<unsigned int>(k[0]) * std::size_t(73856093) ^ k[1] * std::size_t(19349663) ^ k[2] * std::size_t(83492791));
  }
};
using BundleMap = std::unordered_map<Key, std::vector<size_t>, AnyIndexHash>;

}  // namespace

extern "C" {

// The order in which a std::unordered_map<index, ...> with AnyIndexHash hands back n keys inserted in the given
// sequence (duplicates allowed: operator[] semantics).  order_out: the first-insertion positions of the distinct keys,
// in iteration order; returns their number.
int oracle_voxblox_bundle_order(const int32_t* g, int n, int32_t* order_out) {
  std::unordered_map<Key, int32_t, AnyIndexHash> m;
  for (int i = 0; i < n; ++i) {
    const Key k = {g[3 * i], g[3 * i + 1], g[3 * i + 2]};
    if (m.find(k) == m.end()) m[k] = i;
  }
  int c = 0;
  for (const auto& kv : m) order_out[c++] = kv.second;
  return c;
}

// xyz: n x 3 camera-frame points, rgba: n x 4.  Optionally reports the bundles (for tests): *nbundles, and if
// bundle_first_point != nullptr the first point of each bundle in integration order (capacity n).
void oracle_voxblox_integrate_merged(oracle_voxblox* o, const float* xyz, const uint8_t* rgba, int n, const float* Twc,
                                     int* nbundles, int32_t* bundle_first_point) {
  BundleMap voxel_map, clear_map;
  std::vector<int64_t> order((size_t)(n > 0 ? n : 1));
  oracle_voxblox_mixed_order(n, order.data());
  for (int s = 0; s < n; ++s) {                                         // bundleRays
    const size_t pt = (size_t)order[(size_t)s];
    const float* pC = xyz + 3 * pt;
    const int kind = oracle_voxblox_point_kind(o, pC);
    if (kind == 0) continue;
    Key k;
    oracle_voxblox_point_voxel(o, Twc, pC, k.data());
    (kind == 2 ? clear_map : voxel_map)[k].push_back(pt);
  }
  std::vector<float> merged, weights;
  std::vector<uint32_t> colours;
  std::vector<uint8_t> clearing;
  int nb = 0;
  for (int pass = 0; pass < 2; ++pass) {                                // integrateRays(false, ...) then (true, ...)
    const BundleMap& map = pass ? clear_map : voxel_map;
    for (const auto& kv : map) {                                        // integrateVoxels: begin(), ++it
      if (kv.second.empty()) continue;
      uint32_t merged_color = 0;                                        // Color(): 0, 0, 0, 0
      float mp[3] = {0.f, 0.f, 0.f};
      float merged_weight = 0.0f;
      for (const size_t pt : kv.second) {
        const float* pC = xyz + 3 * pt;
        uint32_t color;
        std::memcpy(&color, rgba + 4 * pt, 4);
        const float w = oracle_voxblox_point_weight(pC);
        for (int k = 0; k < 3; ++k) mp[k] = (mp[k] * merged_weight + pC[k] * w) / (merged_weight + w);
        merged_color = oracle_voxblox_blend(merged_color, merged_weight, color, w);
        merged_weight += w;
        if (pass) break;                                                // only take first point when clearing
      }
      merged.insert(merged.end(), mp, mp + 3);
      weights.push_back(merged_weight);
      colours.push_back(merged_color);
      clearing.push_back((uint8_t)pass);
      if (bundle_first_point != nullptr) bundle_first_point[nb] = (int32_t)kv.second.front();
      ++nb;
    }
  }
  if (nbundles != nullptr) *nbundles = nb;
  oracle_voxblox_integrate_bundles(o, merged.data(), colours.data(), weights.data(), clearing.data(), nb, Twc);
}

}  // extern "C"
