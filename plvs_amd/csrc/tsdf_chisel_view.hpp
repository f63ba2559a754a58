// Read-only view of a chisel map for the other translation units of the library (meshing).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "tsdf_directory.hpp"

struct plvs_tsdf_chisel;

namespace plvs {
namespace tsdf {

struct ChiselMapView {
  float resolution = 0.f;
  Directory dir{};
  const float* sdf = nullptr;       // [slot * 4096 + (z * 16 + y) * 16 + x]
  const float* weight = nullptr;
  const uint32_t* kfid = nullptr;
  const uint32_t* rgbw = nullptr;   // r | g << 8 | b << 16 | colour weight << 24
  int num_chunks = 0;
  int shard_count = 1;
  // a slot for state another translation unit keeps with the map (the meshing scratch buffers):
  // *ext is freed with (*ext_free)(*ext) when the map is destroyed
  void** ext = nullptr;
  void (**ext_free)(void*) = nullptr;
};

// False if the handle is unusable (null / poisoned by an earlier error).
bool chisel_map_view(plvs_tsdf_chisel* h, ChiselMapView* v);

}  // namespace tsdf
}  // namespace plvs
