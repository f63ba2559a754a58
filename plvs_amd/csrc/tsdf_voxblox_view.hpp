// Read-only view of a voxblox map for the other translation units of the library (meshing).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "tsdf_directory.hpp"

struct plvs_tsdf_voxblox;

namespace plvs {
namespace vbx {

struct VoxbloxMapView {
  float voxel_size = 0.f;
  float voxel_size_inv = 0.f;       // Block::voxel_size_inv_ = 1.0 / voxel_size_ (core/block.h:29)
  plvs::tsdf::Directory dir{};
  const float* distance = nullptr;  // [slot * 4096 + x + 16 * (y + 16 * z)]
  const float* weight = nullptr;
  const uint32_t* rgba = nullptr;   // r | g << 8 | b << 16 | a << 24
  int num_blocks = 0;
  int visible_blocks = 0;           // own blocks the layer shows (< num_blocks while world-cloud blocks wait: see plvs_hip_tsdf_voxblox_set_deferred_world_blocks)
  int shard_count = 1;
  // sharded maps: copies of other ranks' blocks brought in for meshing (plvs_hip_tsdf_voxblox_halo_*): id -> pool slot
  // past num_blocks; keys null until a halo has been imported
  plvs::tsdf::Directory ghost{};
  // state another translation unit keeps with the map (the meshing scratch buffers): *ext is freed with
  // (*ext_free)(*ext) when the map is destroyed
  void** ext = nullptr;
  void (**ext_free)(void*) = nullptr;
};

// False if the handle is unusable (null / poisoned by an earlier error).
bool voxblox_map_view(plvs_tsdf_voxblox* h, VoxbloxMapView* v);

}  // namespace vbx
}  // namespace plvs
