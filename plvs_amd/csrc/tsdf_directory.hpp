// Device-resident directory of 16^3 voxel blocks shared by the TSDF back ends:
// an open-addressing hash from the packed 3x21-bit block id to the slot of the
// block in the voxel pool, plus the inverse table slot -> id.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace plvs {
namespace tsdf {

constexpr unsigned long long kEmptyKey = ~0ull;
constexpr int kCoordBias = 1 << 20;  // block ids must lie in [-2^20, 2^20)

enum ErrBits : uint32_t {
  kErrPoolFull = 1u,
  kErrCoordRange = 2u,
  kErrDirectoryMiss = 4u,
};

struct Directory {
  unsigned long long* keys;  // packed block id or kEmptyKey
  int32_t* slots;            // pool slot of the entry
  int32_t* slot_ids;         // slot -> block id (3 ints)
  uint32_t mask;             // capacity - 1
  int32_t max_blocks;
};

__device__ __forceinline__ bool pack_block(int x, int y, int z, unsigned long long* key) {
  const unsigned ux = (unsigned)(x + kCoordBias), uy = (unsigned)(y + kCoordBias),
                 uz = (unsigned)(z + kCoordBias);
  if ((ux | uy | uz) >> 21) return false;
  *key = ((unsigned long long)ux << 42) | ((unsigned long long)uy << 21) | (unsigned long long)uz;
  return true;
}

// The spatial hash both references use for their block maps (ChunkHasher,
// open_chisel ChunkManager.h:42-54; AnyIndexHash, voxblox block_hash.h:15-26).
__device__ __forceinline__ uint32_t dir_hash(int x, int y, int z, uint32_t mask) {
  const uint64_t h = ((uint64_t)(int64_t)x * 73856093ull) ^ ((uint64_t)(int64_t)y * 19349663ull) ^
                     ((uint64_t)(int64_t)z * 83492791ull);
  return (uint32_t)h & mask;
}

// Insert-if-absent; the slot of a freshly inserted block becomes visible to
// other threads only after the kernel boundary (the count pass never needs it).
__device__ inline void dir_insert(const Directory& d, int x, int y, int z, int32_t* num_blocks,
                                  uint32_t* err) {
  unsigned long long key;
  if (!pack_block(x, y, z, &key)) {
    atomicOr(err, kErrCoordRange);
    return;
  }
  uint32_t h = dir_hash(x, y, z, d.mask);
  for (uint32_t probe = 0; probe <= d.mask; ++probe) {
    unsigned long long cur = d.keys[h];
    if (cur == key) return;
    if (cur == kEmptyKey) {
      cur = atomicCAS(&d.keys[h], kEmptyKey, key);
      if (cur == kEmptyKey) {
        const int slot = atomicAdd(num_blocks, 1);
        if (slot < d.max_blocks) {
          d.slots[h] = slot;
          d.slot_ids[3 * slot + 0] = x;
          d.slot_ids[3 * slot + 1] = y;
          d.slot_ids[3 * slot + 2] = z;
        } else {
          atomicOr(err, kErrPoolFull);
        }
        return;
      }
      if (cur == key) return;
    }
    h = (h + 1) & d.mask;
  }
  atomicOr(err, kErrPoolFull);
}

// Find-or-insert with the slot returned; usable by concurrent workgroups of ONE kernel.  A block that
// another thread is just inserting is waited for (its slot arrives a few instructions after its key) —
// in a SECOND phase, entered only after every lane of the wave has finished its own insertions: lanes
// of one wave run their divergent paths one after the other, so a lane that spun for another wave's
// slot before its neighbour lane had published the slot THAT wave is spinning for would deadlock.
__device__ inline int dir_find_or_insert(const Directory& d, int x, int y, int z, int32_t* num_blocks,
                                         uint32_t* err) {
  unsigned long long key;
  if (!pack_block(x, y, z, &key)) {
    atomicOr(err, kErrCoordRange);
    return -1;
  }
  uint32_t h = dir_hash(x, y, z, d.mask);
  int slot = -1;
  bool pending = false, full = true;
  // ---- phase 1: find the entry or create it (no waiting)
  for (uint32_t probe = 0; probe <= d.mask; ++probe) {
    // plain (cached) loads first: a block inserted by an earlier kernel is found without leaving the L2.  A
    // stale line can only look emptier than the truth (an entry goes from empty to its final value once),
    // and then the read-modify-write path decides: RMW atomics are performed at the device's coherence
    // point, unlike loads, which another XCD's L2 may serve from a stale line.
    unsigned long long cur = d.keys[h];
    if (cur == key) {
      const int cached = d.slots[h];
      if (cached >= 0) {
        slot = cached;
        full = false;
        break;
      }
    } else if (cur != kEmptyKey) {   // another block's entry (an entry never changes once written): next probe, and
      h = (h + 1) & d.mask;          // no read-modify-write on a word that thousands of threads pass over
      continue;
    }
    cur = atomicCAS(&d.keys[h], kEmptyKey, key);
    if (cur == kEmptyKey) {   // inserted here: allocate and publish the slot
      const int s = atomicAdd(num_blocks, 1);
      if (s < d.max_blocks) {
        d.slot_ids[3 * s + 0] = x;
        d.slot_ids[3 * s + 1] = y;
        d.slot_ids[3 * s + 2] = z;
        atomicExch(&d.slots[h], s);
        slot = s;
      } else {
        atomicOr(err, kErrPoolFull);
        atomicExch(&d.slots[h], -2);
      }
      full = false;
      break;
    }
    if (cur == key) {
      pending = true;
      full = false;
      break;
    }
    h = (h + 1) & d.mask;
  }
  if (full) atomicOr(err, kErrPoolFull);
  // ---- phase 2: the slot of an entry somebody else created (slots are -1 while pending, -2 / >= 0 when final)
  if (pending) {
    int spins = 0;
    do {
      slot = atomicMax(&d.slots[h], -2);   // an RMW read
    } while (slot == -1 && ++spins < (1 << 24));
    if (slot == -1) atomicOr(err, kErrDirectoryMiss);
    if (slot < 0) slot = -1;
  }
  return slot;
}

__device__ inline int dir_find(const Directory& d, int x, int y, int z) {
  unsigned long long key;
  if (!pack_block(x, y, z, &key)) return -1;
  uint32_t h = dir_hash(x, y, z, d.mask);
  for (uint32_t probe = 0; probe <= d.mask; ++probe) {
    const unsigned long long cur = d.keys[h];
    if (cur == key) return d.slots[h];
    if (cur == kEmptyKey) return -1;
    h = (h + 1) & d.mask;
  }
  return -1;
}

// cloud index of global point i: largest c with offsets[c] <= i.
__device__ __forceinline__ int cloud_of(const int32_t* __restrict__ offsets, int nclouds, int i) {
  int lo = 0, hi = nclouds - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (offsets[mid] <= i) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// The tiles of the single-walk integrate: `rays` consecutive points of ONE cloud — a cloud's last tile is shorter, so no
// tile straddles two clouds (two poses): until round 4 tiles were cut from the concatenated stream and the one tile per
// cloud boundary took the general kernel, ~100 us of latency behind every batch.  The device copy of a call's offsets
// carries the table right behind them: offsets[0 .. nclouds], then tile_base[0 .. nclouds] (first tile of cloud c;
// tile_base[nclouds] = the tiles of the call).
struct TileSpan {
  int cloud;
  uint32_t first, nrays;   // first point (index into the call's stream), rays of the tile
};
__device__ __forceinline__ TileSpan tile_span(const int32_t* __restrict__ offsets, int nclouds, uint32_t gtile, uint32_t rays) {
  const int32_t* tb = offsets + nclouds + 1;
  TileSpan t;
  t.cloud = cloud_of(tb, nclouds, (int)gtile);   // (the LAST cloud whose base is <= gtile: empty clouds share a base with their successor)
  t.first = (uint32_t)offsets[t.cloud] + (gtile - (uint32_t)tb[t.cloud]) * rays;
  const uint32_t left = (uint32_t)offsets[t.cloud + 1] - t.first;
  t.nrays = left < rays ? left : rays;
  return t;
}
// The same from the two per-tile tables a caller may put behind the tile table (tile_first[0 .. ntiles), tile_cloud[0 ..
// ntiles)): three dependent loads instead of the eight of the search.
__device__ __forceinline__ TileSpan tile_span_tables(const int32_t* __restrict__ offsets, int nclouds, uint32_t gtile, uint32_t rays) {
  const int32_t* tb = offsets + nclouds + 1;
  const int32_t* tf = tb + nclouds + 1;
  const uint32_t nt = (uint32_t)tb[nclouds];
  TileSpan t;
  t.first = (uint32_t)tf[gtile];
  t.cloud = tf[nt + gtile];
  const uint32_t left = (uint32_t)offsets[t.cloud + 1] - t.first;
  t.nrays = left < rays ? left : rays;
  return t;
}
// host: both[0 .. nclouds] = offsets, both[nclouds + 1 .. 2 nclouds + 1] = tile_base; returns the tiles of the call
static inline uint32_t fill_tile_table(const int32_t* offsets, int nclouds, int32_t* both, uint32_t rays) {
  uint32_t t = 0;
  for (int c = 0; c <= nclouds; ++c) both[c] = offsets[c];
  for (int c = 0; c < nclouds; ++c) {
    both[nclouds + 1 + c] = (int32_t)t;
    t += ((uint32_t)(offsets[c + 1] - offsets[c]) + rays - 1u) / rays;
  }
  both[2 * nclouds + 1] = (int32_t)t;
  return t;
}

}  // namespace tsdf
}  // namespace plvs
