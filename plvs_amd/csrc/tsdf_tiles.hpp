// Pieces of the tile pipeline shared by the TSDF back ends (tsdf_chisel.hip, tsdf_voxblox.hip):
//   * the LDS phases of a tile kernel once the visits of the tile sit in LDS — group by voxel
//     key (hash table), stable radix sort of the (group, slot) tags, run heads — and the
//     decoupled look-back that numbers the runs across tiles;
//   * the run pipeline behind it: run_counts, mark_blocks, gather_runs (runs -> voxel order),
//     voxel_heads (voxel list + updated blocks).
// See the stage description in tsdf_chisel.hip.  Everything lives in an anonymous namespace:
// each translation unit instantiates its own copy.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "tsdf_directory.hpp"

namespace {

using plvs::tsdf::Directory;
using plvs::tsdf::dir_find;
using plvs::tsdf::dir_hash;
using plvs::tsdf::kEmptyKey;
using plvs::tsdf::pack_block;

constexpr int kTileSlots = 4096;
constexpr int kTileThreads = 512;
constexpr int kTileItems = kTileSlots / kTileThreads;   // per thread in the LDS phases
constexpr uint32_t kTileEmpty = 0xFFFFFFFFu;
constexpr int kTileChunkCache = 64;
constexpr int kTileCloudCache = 64;

__global__ void mark_tiles(const uint32_t* __restrict__ voff /* n + 1 */, int n,
                           uint32_t* __restrict__ tile_first) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t o = voff[i], e = voff[i + 1];
  // the point owning the first slot of a tile
  for (uint32_t t = (o + kTileSlots - 1) / kTileSlots; (unsigned long long)t * kTileSlots < e; ++t) tile_first[t] = (uint32_t)i;
}

// One pass of a stable LSD radix sort of the kTileSlots tags of a tile in LDS (6-bit digit at
// `shift`).  Wave w owns the contiguous span [w*512, (w+1)*512); equal digits are ranked inside
// a wave-row with ballots (lane order = slot order), a running count per (wave, digit) carries
// over the rows, and one scan over the 8 x 64 counts places the spans.
constexpr int kTileRadixBits = 6;
constexpr int kTileRadix = 1 << kTileRadixBits;
__device__ __forceinline__ void tile_radix_pass(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                int shift, uint32_t (*wave_hist)[kTileRadix], int tid) {
  constexpr int kWavesT = kTileThreads / 64;
  const int lane = tid & 63, wid = tid >> 6;
  wave_hist[wid][lane] = 0;   // 64 digits, 64 lanes
  __syncthreads();
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  uint32_t v[kTileItems], rank[kTileItems];
  volatile uint32_t* my_hist = wave_hist[wid];
#pragma unroll
  for (int it = 0; it < kTileItems; ++it) {
    v[it] = src[wid * (kTileItems * 64) + it * 64 + lane];
    const uint32_t d = (v[it] >> shift) & (kTileRadix - 1);
    unsigned long long peers = ~0ull;
#pragma unroll
    for (int b = 0; b < kTileRadixBits; ++b) {
      const unsigned long long bal = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? bal : ~bal;
    }
    const uint32_t before = (uint32_t)__popcll(peers & lt_mask);
    const uint32_t base = my_hist[d];
    rank[it] = base + before;
    // the lowest peer publishes the new running count; LDS operations of one wave execute in
    // order, so every peer has read `base` before this store lands
    if (before == 0) my_hist[d] = base + (uint32_t)__popcll(peers);
  }
  __syncthreads();
  // exclusive scan over (digit, wave) in that order: thread = one (digit, wave) cell
  {
    const int d = tid / kWavesT, w = tid % kWavesT;   // 512 cells
    const uint32_t c = wave_hist[w][d];
    uint32_t inc = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)inc, off);
      if (lane >= off) inc += up;
    }
    __shared__ uint32_t part[kWavesT];
    if (lane == 63) part[wid] = inc;
    __syncthreads();
    uint32_t basev = 0;
#pragma unroll
    for (int q = 0; q < kWavesT; ++q) basev += (q < wid) ? part[q] : 0u;
    wave_hist[w][d] = basev + inc - c;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kTileItems; ++it) {
    const uint32_t d = (v[it] >> shift) & (kTileRadix - 1);
    dst[wave_hist[wid][d] + rank[it]] = v[it];
  }
  __syncthreads();
}

// dir_find through a small per-tile cache in LDS (a tile meets a handful of chunks, every ray
// of it asks for them again).  An entry whose slot is still pending, or a full cache, falls
// back to the directory.
__device__ __forceinline__ int tile_find_chunk(const Directory& dir, unsigned long long* ckey, int32_t* cslot,
                                               int x, int y, int z) {
  unsigned long long key;
  if (!pack_block(x, y, z, &key)) return -1;
  uint32_t h = dir_hash(x, y, z, kTileChunkCache - 1);
  for (int probe = 0; probe < kTileChunkCache; ++probe) {
    unsigned long long cur = ckey[h];
    if (cur == kEmptyKey) {
      const int slot = dir_find(dir, x, y, z);
      cur = atomicCAS(&ckey[h], kEmptyKey, key);
      if (cur == kEmptyKey) {
        cslot[h] = slot;
        return slot;
      }
      if (cur == key) return slot;
    } else if (cur == key) {
      const int slot = cslot[h];
      return slot != -2 ? slot : dir_find(dir, x, y, z);
    }
    h = (h + 1) & (kTileChunkCache - 1);
  }
  return dir_find(dir, x, y, z);
}

__device__ __forceinline__ unsigned long long ld_state(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_state(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Phases 2-4 of a tile kernel.  In: skey[s] = voxel key of the visit in slot s (s < n).  Out:
// bufA = the (group << 12 | slot) tags sorted by group (slot order inside a group), bufB
// reinterpreted as uint16_t hp[]: hp[g] = position of run g, hp[ngroups] = n.  Returns ngroups.
__device__ __forceinline__ uint32_t tile_group_sort_heads(const uint32_t* skey, uint32_t* bufA, uint32_t* bufB,
                                                          uint32_t (*wave_hist)[kTileRadix], uint32_t* wsum,
                                                          uint32_t n, int tid) {
  uint32_t* const gtab = bufB;   // representative slot of the voxel hashed to this entry
  const int lane = tid & 63, wid = tid >> 6;
  // ---- phase 2: group by voxel key.  The hash table gives every voxel of the tile one entry;
  // a visit is tagged with the entry of its voxel.
#pragma unroll
  for (int k = 0; k < kTileItems; ++k) {
    const uint32_t s = tid + k * kTileThreads;
    uint32_t tag = kTileEmpty;   // beyond the tile's last slot: sorts behind everything
    if (s < n) {
      const uint32_t key = skey[s];
      uint32_t h = (key * 2654435761u) >> 20;
      for (;;) {
        uint32_t e = gtab[h];
        if (e == kTileEmpty) {
          e = atomicCAS(&gtab[h], kTileEmpty, s);   // s: the entry's representative visit
          if (e == kTileEmpty) break;
        }
        if (skey[e] == key) break;
        h = (h + 1) & (kTileSlots - 1);
      }
      tag = (h << 12) | s;
    }
    bufA[s] = tag;
  }
  __syncthreads();

  // ---- phase 3: stable LSD radix sort of the tags by table entry (2 x 6 bits): visits of one
  // voxel end up contiguous and in slot (= point) order.  bufB overlays the hash table.
  tile_radix_pass(bufA, bufB, 12, wave_hist, tid);
  tile_radix_pass(bufB, bufA, 18, wave_hist, tid);

  // ---- phase 4: group heads: positions hp[g] of the runs, their number
  uint32_t acc = 0;
  uint32_t flags = 0;
#pragma unroll
  for (int k = 0; k < kTileItems; ++k) {
    const uint32_t j = tid * kTileItems + k;   // this thread's contiguous positions
    const uint32_t cur = bufA[j];
    const uint32_t prev = j ? bufA[j - 1] : kTileEmpty;
    const bool head = j < n && (j == 0 || (cur >> 12) != (prev >> 12));
    flags |= head ? (1u << k) : 0u;
    acc += head ? 1u : 0u;
  }
  uint32_t inc = acc;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)inc, off);
    if (lane >= off) inc += up;
  }
  if (lane == 63) wsum[wid] = inc;
  __syncthreads();
  uint32_t wbase = 0, ngroups = 0;
#pragma unroll
  for (int w = 0; w < kTileThreads / 64; ++w) {
    const uint32_t v = wsum[w];
    if (w < wid) wbase += v;
    ngroups += v;
  }
  uint16_t* const hp = reinterpret_cast<uint16_t*>(bufB);
  {
    uint32_t g = wbase + inc - acc;
#pragma unroll
    for (int k = 0; k < kTileItems; ++k)
      if (flags & (1u << k)) hp[g++] = (uint16_t)(tid * kTileItems + k);
  }
  if (tid == 0) hp[ngroups] = (uint16_t)n;
  return ngroups;
}

// The tile's place in the run numbering: wave 0 looks back over 64 predecessors at a time
// (decoupled look-back; tile ids are tickets, so a tile only waits for tiles that already
// started).  The tile's own count must have been published (status 1) before.
__device__ __forceinline__ void tile_lookback(uint32_t t, uint32_t ngroups, uint32_t ntiles,
                                              unsigned long long* __restrict__ tile_state, uint32_t* sh_base,
                                              uint32_t* num_desc, int tid) {
  const int lane = tid & 63;
  if ((tid >> 6) != 0) return;
  unsigned long long base = 0;
  if (t > 0) {
    for (long long hi = (long long)t - 1; hi >= 0; hi -= 64) {
      const long long p = hi - lane;   // lane 0 = nearest predecessor
      unsigned long long st = 2ull << 62;   // before the first tile: an empty prefix
      if (p >= 0) {
        do { st = ld_state(&tile_state[p]); } while ((st >> 62) == 0);
      }
      const unsigned long long is_prefix = __ballot((st >> 62) == 2);
      const int stop = __ffsll((long long)is_prefix) - 1;   // nearest tile with a known prefix
      unsigned long long v = (stop < 0 || lane <= stop) ? (st & ((1ull << 62) - 1)) : 0ull;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += (unsigned long long)__shfl_xor((long long)v, off);
      base += v;
      if (stop >= 0) break;
    }
  }
  if (lane == 0) {
    st_state(&tile_state[t], (2ull << 62) | (base + ngroups));
    *sh_base = (uint32_t)base;
    if (t + 1 == ntiles) *num_desc = (uint32_t)(base + ngroups);
  }
}

// lengths of the runs in sorted order (input of the scan that places them)
__global__ void run_counts(const unsigned long long* __restrict__ sorted_val, uint32_t nd,
                           uint32_t* __restrict__ cnts) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < nd) cnts[j] = (uint32_t)(sorted_val[j] >> 32);
}

// the run holding the first record of every block of kGatherSpan output records
constexpr int kGatherThreads = 256;
constexpr int kGatherSpan = 2048;
__global__ void mark_blocks(const uint32_t* __restrict__ dst, uint32_t nd, uint32_t V,
                            uint32_t* __restrict__ block_first) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nd) return;
  const uint32_t a = dst[j], e = (j + 1 < nd) ? dst[j + 1] : V;
  for (uint32_t b = (a + kGatherSpan - 1) / kGatherSpan; (unsigned long long)b * kGatherSpan < e; ++b) block_first[b] = j;
}

// Stage 5.  Runs in voxel order (stable: tile order inside a voxel) -> the records of
// every voxel contiguous and in point order, which is what the chain kernels walk:
//   rec[r]   = (w_u * u, +-w_u)   negative on the LAST record of a voxel
//   rec_c[r] = r | g<<8 | b<<16   (only if the caller wants the colours in voxel order too)
// Output-centric: a block owns kGatherSpan consecutive output records, so its stores are
// fully coalesced; the runs that cover the span are looked up once (their first positions
// are marked in LDS and a max-scan hands every output record its run), the loads follow the
// runs (contiguous pieces of a tile).  Also sets the keyframe id of the voxel (SetKfid: the
// last update of the call wins).  The voxel list itself comes from voxel_heads.
__global__ __launch_bounds__(kGatherThreads) void gather_runs(
    const uint32_t* __restrict__ skeys, const unsigned long long* __restrict__ sorted_val, uint32_t nd,
    const uint32_t* __restrict__ last_pt, const uint32_t* __restrict__ dst,
    const uint32_t* __restrict__ block_first, uint32_t nblocks, uint32_t V,
    const float2* __restrict__ rec_t, const uint32_t* __restrict__ recc_t, float2* __restrict__ rec,
    uint32_t* __restrict__ rec_c, const uint32_t* __restrict__ kfid, uint32_t* __restrict__ vkfid) {
  constexpr int kItems = kGatherSpan / kGatherThreads;   // 8
  constexpr int kMaxRuns = kGatherSpan + 1;
  __shared__ uint16_t id[kGatherSpan];          // run (local index + 1) of every output record
  __shared__ uint32_t delta[kMaxRuns + 3];      // source position - output position of the run
  __shared__ uint32_t endl[kMaxRuns + 3];       // end of the run | bit 31: it closes its voxel
  __shared__ uint32_t wtot[kGatherThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t b = blockIdx.x;
  const uint32_t B0 = b * kGatherSpan;
  const uint32_t nrec = min((uint32_t)kGatherSpan, V - B0);
  const uint32_t j_lo = block_first[b];
  const uint32_t j_hi = (b + 1 < nblocks) ? block_first[b + 1] : nd - 1;
  const uint32_t nruns = j_hi - j_lo + 1;
#pragma unroll
  for (int k = 0; k < kItems; ++k) id[tid + k * kGatherThreads] = 0;
  __syncthreads();

  // ---- the runs of the span; the run that closes a voxel also sets its keyframe id
  for (uint32_t jl = tid; jl < nruns; jl += kGatherThreads) {
    const uint32_t j = j_lo + jl;
    const uint32_t key = skeys[j];
    const uint32_t next = (j + 1 < nd) ? skeys[j + 1] : ~key;
    const unsigned long long d = sorted_val[j];
    const uint32_t from = (uint32_t)d, len = (uint32_t)(d >> 32);
    const uint32_t a = dst[j];
    const bool closes = key != next;
    delta[jl] = from - a;
    endl[jl] = (a + len) | (closes ? 0x80000000u : 0u);
    if (a < B0 + nrec) id[max(a, B0) - B0] = (uint16_t)(jl + 1);
    if (vkfid != nullptr && closes && a >= B0 && a < B0 + nrec) vkfid[key] = kfid ? kfid[last_pt[from]] : 0u;
  }
  __syncthreads();   // id / delta / endl complete
  // ---- every output record learns its run: inclusive max-scan of the marks
  uint32_t loc[kItems], run_max = 0;
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    run_max = max(run_max, (uint32_t)id[tid * kItems + k]);
    loc[k] = run_max;
  }
  uint32_t sc = run_max;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)sc, off);
    if (lane >= off) sc = max(sc, up);
  }
  if (lane == 63) wtot[wid] = sc;
  __syncthreads();
  uint32_t before = 0;
#pragma unroll
  for (int w = 0; w < kGatherThreads / 64; ++w) before = (w < wid) ? max(before, wtot[w]) : before;
  const uint32_t up1 = (uint32_t)__shfl_up((int)sc, 1);
  before = max(before, lane ? up1 : 0u);
#pragma unroll
  for (int k = 0; k < kItems; ++k) id[tid * kItems + k] = (uint16_t)max(loc[k], before);
  __syncthreads();
  // ---- copy
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const uint32_t q = tid + k * kGatherThreads;
    if (q < nrec) {
      const uint32_t jl = (uint32_t)id[q] - 1u;
      const uint32_t r = B0 + q;
      const uint32_t src = r + delta[jl];
      const uint32_t e = endl[jl];
      float2 v = rec_t[src];
      if ((e >> 31) && (e & 0x7FFFFFFFu) == r + 1) v.y = -v.y;
      rec[r] = v;
      if (rec_c != nullptr) rec_c[r] = recc_t[src];
    }
  }
}

constexpr int kHeadTiles = 16;   // 4096 runs per block: few same-address atomics
template <typename CountersT>
__global__ __launch_bounds__(256) void voxel_heads(
    const uint32_t* __restrict__ skeys, uint32_t nd, uint32_t* __restrict__ vj0,
    uint32_t* __restrict__ updated_slots, CountersT* __restrict__ ctr, const uint32_t* __restrict__ nd_dev = nullptr) {
  // (nd_dev: the array holds min(nd, *nd_dev) runs — a launch sized by a bound on their number)
  if (nd_dev != nullptr) nd = min(nd, *nd_dev);
  __shared__ uint32_t wtot[4];
  __shared__ uint32_t block_base[2];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const uint32_t first = blockIdx.x * (256 * kHeadTiles) + threadIdx.x * kHeadTiles;   // this thread's runs
  uint32_t hmask = 0, cmask = 0;
#pragma unroll
  for (int k = 0; k < kHeadTiles; ++k) {
    const uint32_t j = first + k;
    if (j < nd) {
      const uint32_t key = skeys[j];
      const uint32_t prev = j ? skeys[j - 1] : ~key;
      hmask |= (key != prev) ? (1u << k) : 0u;
      cmask |= (j == 0 || (key >> 12) != (prev >> 12)) ? (1u << k) : 0u;
    }
  }
  const uint32_t mine = (uint32_t)__popc(hmask) | ((uint32_t)__popc(cmask) << 16);
  uint32_t inc = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)inc, off);
    if (lane >= off) inc += up;
  }
  if (lane == 63) wtot[wid] = inc;
  __syncthreads();
  uint32_t wbase = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const uint32_t v = wtot[w];
    if (w < wid) wbase += v;
    total += v;
  }
  if (threadIdx.x < 2) {
    const uint32_t c = threadIdx.x == 0 ? (total & 0xFFFFu) : (total >> 16);
    block_base[threadIdx.x] = c ? atomicAdd(threadIdx.x == 0 ? &ctr->num_heads : &ctr->num_updated, c) : 0u;
  }
  __syncthreads();
  const uint32_t excl = wbase + inc - mine;
  uint32_t at_h = block_base[0] + (excl & 0xFFFFu), at_c = block_base[1] + (excl >> 16);
#pragma unroll
  for (int k = 0; k < kHeadTiles; ++k) {
    if (hmask & (1u << k)) vj0[at_h++] = first + k;
    if (cmask & (1u << k)) updated_slots[at_c++] = skeys[first + k] >> 12;
  }
}

}  // namespace
