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
  int shard_rank = 0;
  // sharded maps (shard_count > 1): copies of other ranks' chunks brought in for meshing (plvs_hip_tsdf_chisel_halo_*).
  // ghost.keys is null until a halo has been imported; a ghost entry's slot is a pool slot past num_chunks, or
  // kGhostAbsent for a chunk its owner does not have.  A look-up of a foreign chunk that is in neither directory is
  // recorded in the miss set (ids, deduplicated through miss_keys) and answered "does not exist" for this pass.
  Directory ghost{};
  unsigned long long* miss_keys = nullptr;
  uint32_t miss_mask = 0;
  int32_t* miss_ids = nullptr;      // miss_cap x 3
  uint32_t* miss_count = nullptr;
  uint32_t miss_cap = 0;
  // a slot for state another translation unit keeps with the map (the meshing scratch buffers):
  // *ext is freed with (*ext_free)(*ext) when the map is destroyed
  void** ext = nullptr;
  void (**ext_free)(void*) = nullptr;
};

constexpr int kGhostAbsent = -2;

// False if the handle is unusable (null / poisoned by an earlier error).  On a sharded map this also readies the
// miss set (allocates it on first use, empties it).
bool chisel_map_view(plvs_tsdf_chisel* h, ChiselMapView* v);

}  // namespace tsdf
}  // namespace plvs
