// TSDF integrate for the open_chisel back end (PointCloudMapChisel::InsertCloud
// -> Chisel::IntegratePointCloudWidthDepth, point-cloud part).
//
// Data layout in HBM (per handle):
//   voxel pool      four planes sdf / weight / kfid / rgbw, each
//                   max_chunks * 4096 dwords; chunk slot s owns words
//                   [s*4096, (s+1)*4096) of every plane (64 KiB per chunk).
//   chunk directory open-addressing hash (2*max_chunks, power of two) from the
//                   packed 3x21-bit chunk id to the pool slot, plus slot -> id.
//   per call        per voxel visit the update operands (w_u*u, w_u: 8 bytes) and the
//                   colour (4 bytes), written once grouped per tile and once in voxel
//                   order; per tile-local run of visits to one voxel a 20-byte descriptor.
//
// The reference integrates points strictly in order, and both the running
// weighted mean (f32) and the truncating u8 colour mean are order dependent.
// The device path keeps that order exactly:
//   1. ray_count   one thread per point walks its Amanatides-Woo ray and counts
//                  the voxels that take an update; first-touch chunks are
//                  inserted into the directory.
//   2. scan        exclusive scan of the counts = visit offsets (point order).
//   3. ray_tiles   the visit slots are cut into tiles of 4096; a workgroup re-walks the
//                  rays of its tile, keeps the visits in LDS, groups them by voxel (point
//                  order inside a group) and writes the update operands (w_u*u, w_u) and
//                  colours grouped that way, plus one descriptor per group ("run").
//   4. sort_runs   stable radix sort of the run descriptors by voxel key: per voxel its
//                  runs in tile (= point) order.  Runs, not visits, are sorted.
//   5. gather_runs copies the runs into voxel order -> per voxel its records contiguous
//                  and in point order; compacts voxel heads and updated chunks.
//   6. chain       one thread per voxel folds its records sequentially in registers (the
//                  f32 weighted mean; the truncating u8 colour mean on a second stream):
//                  each voxel is read and written once per call.
// Results are bit-identical to the sequential CPU loop.  This is the ordered mode (order_free = 0); the
// order-free mode (sdf / weight within a stated float tolerance, kfid and colour exact) is the single-walk
// pipeline of tsdf_walk.hpp.
#include <vector>

#include "common.hpp"
#include "device_utils.hpp"
#include "tsdf_chisel_core.hpp"
#include "tsdf_directory.hpp"
#include "tsdf_chisel_view.hpp"
#include "tsdf_tiles.hpp"
#include "tsdf_walk.hpp"
#include "tsdf_shard.hpp"

using namespace plvs;
using namespace plvs::chisel;
using namespace plvs::tsdf;

namespace {

constexpr int kNumStages = 6;
const char* const kStageNames[kNumStages] = {"ray_count", "scan", "ray_tiles", "sort_runs",
                                             "gather_runs", "chain_runs"};

#ifndef PLVS_CHAIN_PROBE
#define PLVS_CHAIN_PROBE 0
#endif
#ifndef PLVS_TILE_PROBE
#define PLVS_TILE_PROBE 0
#endif
#if PLVS_TILE_PROBE
#define TILE_PROBE(i)                                                          \
  __builtin_amdgcn_sched_barrier(0);                                           \
  if (threadIdx.x == 0) { const unsigned long long now_ = clock64(); atomicAdd(&ctr->tprobe[i], now_ - tp_); tp_ = now_; } \
  __builtin_amdgcn_sched_barrier(0);
#else
#define TILE_PROBE(i)
#endif

struct Counters {           // device-side, read back once per call
  uint32_t total_visits;
  int32_t num_chunks;
  uint32_t err;
  uint32_t num_heads;
  uint32_t num_updated;
  uint32_t max_run;
  uint32_t num_desc;
#if PLVS_CHAIN_PROBE
  unsigned long long probe[6];   // instrumentation build only
#endif
#if PLVS_TILE_PROBE
  unsigned long long tprobe[8];   // instrumentation build only
#endif
};

__global__ void pose_prep(const float* __restrict__ Twc, int nclouds, Pose* __restrict__ poses) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < nclouds) make_pose(Twc + 12 * c, &poses[c]);
}

// The order-free call's whole prologue in one launch: poses, the cloud offsets (read from the pinned host copy),
// zeroed counters and per-chunk segment counts.  (Seven small commands — copy, kernel, five fills — cost 40 us of
// queueing in front of a 55 us walk of one keyframe.)
__global__ void walk_prologue(const float* __restrict__ Twc, int nclouds, Pose* __restrict__ poses,
                              const int32_t* __restrict__ host_offsets, int32_t* __restrict__ offsets,
                              WalkCounters* __restrict__ wctr, Counters* __restrict__ ctr, uint32_t* __restrict__ chunk_nseg,
                              int nseg, int ntile_first = 0) {
  const int i0 = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  for (int c = i0; c < nclouds; c += stride) make_pose(Twc + 12 * c, &poses[c]);
  // (offsets + tile table, and behind them the first point of every tile: what the colour fold reads per run)
  for (int c = i0; c < 2 * (nclouds + 1) + ntile_first; c += stride) offsets[c] = host_offsets[c];
  for (int k = i0; k < nseg; k += stride) chunk_nseg[k] = 0u;
  uint32_t* w = reinterpret_cast<uint32_t*>(wctr);
  for (int k = i0; k < (int)(2 * sizeof(WalkCounters) / sizeof(uint32_t)); k += stride) w[k] = 0u;
  if (i0 == 0) {   // reset the per-call counters, keep num_chunks
    ctr->total_visits = 0;
    ctr->err = 0;
    ctr->num_heads = 0;
    ctr->num_updated = 0;
    ctr->max_run = 0;
    ctr->num_desc = 0;
  }
}

// Counters -> their pinned host copies by a kernel's stores: a small device-to-host copy command costs tens of
// microseconds of queueing, a store through the host-mapped pointer a few.
// host_seq (the order-free pipeline's reads): after the counters a sequence number, stored with system scope — the host polls that
// word in pinned memory instead of sleeping on the stream (wait_published: a wake-up is 20-30 us, on the critical path of a
// long call's colour chain and a fifth of a one-key-frame call)
__global__ void publish_counters(const WalkCounters* __restrict__ wctr, const Counters* __restrict__ ctr,
                                 WalkCounters* __restrict__ host_wctr, Counters* __restrict__ host_ctr,
                                 uint32_t* __restrict__ host_seq = nullptr, uint32_t seq = 0u) {
  if (wctr) {
    const uint32_t* a = reinterpret_cast<const uint32_t*>(wctr);
    uint32_t* b = reinterpret_cast<uint32_t*>(host_wctr);
    for (int k = threadIdx.x; k < (int)(2 * sizeof(WalkCounters) / sizeof(uint32_t)); k += blockDim.x) b[k] = a[k];
  }
  if (ctr) {
    const uint32_t* c = reinterpret_cast<const uint32_t*>(ctr);
    uint32_t* d = reinterpret_cast<uint32_t*>(host_ctr);
    for (int k = threadIdx.x; k < (int)(sizeof(Counters) / sizeof(uint32_t)); k += blockDim.x) d[k] = c[k];
  }
  __threadfence_system();
  if (host_seq != nullptr) {
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// The start of shard_apply in one launch: counters of the call cleared, the received totals in place (what a
// handful of small memsets / copies would do, each a runtime call of its own).
__global__ void shard_apply_begin(Counters* ctr, WalkCounters* wctr, int32_t* xcount_sat, uint32_t* __restrict__ chunk_nseg,
                                  uint32_t max_chunks, uint32_t total_seg, uint32_t total_runs) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  for (uint32_t c = i; c < max_chunks; c += gridDim.x * blockDim.x) chunk_nseg[c] = 0u;
  if (i == 0) {
    ctr->total_visits = 0u;
    ctr->err = 0u; ctr->num_heads = 0u; ctr->num_updated = 0u; ctr->max_run = 0u; ctr->num_desc = 0u;
    wctr[0] = WalkCounters{};
    wctr[1] = WalkCounters{};
    wctr[0].seg_top = total_seg;
    wctr[1].num_desc = total_runs;
    *xcount_sat = 0;
  }
}

// ... and any few words the same way (up to three ranges per launch).
__global__ void publish_words(const uint32_t* __restrict__ a, uint32_t* __restrict__ ha, int na, const uint32_t* __restrict__ b,
                              uint32_t* __restrict__ hb, int nb, const uint32_t* __restrict__ c, uint32_t* __restrict__ hc, int nc) {
  for (int k = threadIdx.x; k < na; k += blockDim.x) ha[k] = a[k];
  for (int k = threadIdx.x; k < nb; k += blockDim.x) hb[k] = b[k];
  for (int k = threadIdx.x; k < nc; k += blockDim.x) hc[k] = c[k];
  __threadfence_system();
}

__global__ void pool_init(float* __restrict__ sdf, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t j = i; j < n; j += stride) sdf[j] = 99999.0f;
}

// Stage 1: count the updating visits of each point and insert first-touch chunks.
// kNormals: the world-cloud-with-normals flavour (make_ray_normal / resolve_visit_normal), `normals` n x 3.
template <bool kNormals>
__global__ __launch_bounds__(256) void ray_count(
    Params P, const float* __restrict__ xyz, const float* __restrict__ normals, int npoints,
    const int32_t* __restrict__ offsets, int nclouds, const Pose* __restrict__ poses, Directory dir,
    Counters* __restrict__ ctr, uint32_t* __restrict__ counts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npoints) return;
  const Pose pose = poses[cloud_of(offsets, nclouds, i)];
  Ray ray;
  RayN aux;
  uint32_t n = 0;
  bool walk = true;
  if (kNormals)
    make_ray_normal(P, pose, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], normals[3 * (size_t)i],
                    normals[3 * (size_t)i + 1], normals[3 * (size_t)i + 2], &ray, &aux);
  else
    walk = make_ray(P, pose, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], &ray);
  if (walk && !ray_in_coord_range(ray)) {
    // beyond the range in which the integer chunk addressing equals the reference's float
    // lookup: fail loudly instead of diverging
    atomicOr(&ctr->err, kErrCoordRange);
    walk = false;
  }
  // on a shard most rays cannot reach a chunk of this rank: skip their set-up and walk (with two
  // ranks nearly every ray still can, the test would only cost)
  if (walk && P.shard_count > 2 && !walk_may_touch_owned(P, ray)) walk = false;
  if (walk) {
    RayCursor cur;
    OwnerCache owner;
    ray_begin(ray, &cur);
    int vx, vy, vz;
    int lcx = 0, lcy = 0, lcz = 0;  // last chunk seen by this ray
    bool have_last = false;
    // one DDA step per trip for every lane (an early `continue` on rejected steps makes the compiler
    // nest a skip loop in which the lanes of a wave wait for each other's rejected stretches)
    while (ray_next(&cur, &vx, &vy, &vz)) {
      Visit v;
      const bool ok = kNormals ? resolve_visit_normal(P, aux, ray, vx, vy, vz, &v, &owner)
                               : resolve_visit(P, pose, ray, vx, vy, vz, &v, &owner);
      if (ok && (!have_last || v.cx != lcx || v.cy != lcy || v.cz != lcz)) {
        lcx = v.cx; lcy = v.cy; lcz = v.cz;
        have_last = true;
        dir_insert(dir, lcx, lcy, lcz, &ctr->num_chunks, &ctr->err);
      }
      n += ok ? 1u : 0u;
    }
  }
  counts[i] = n;
}

// ------------------------------------------------------------------ tiles
// Stage 3.  The visit slots of the call (point order, dense: offsets = scan of the counts)
// are cut into tiles of kTileSlots.  One workgroup per tile re-walks the rays of its points,
// keeps the tile's visits in LDS, groups them by voxel and writes
//   * the update operands (w_u*u, w_u) and the colour of every visit, grouped by voxel
//     and, inside a group, in point order (tile-local "runs"), fully coalesced;
//   * one descriptor per run: voxel key (slot*4096 + voxel), position, length and the
//     point of its last visit.
// Only the descriptors (one per run, not one per visit) go through the global sort.
//
// Grouping: an LDS hash table keyed by the voxel key gives every visit the table entry of
// its voxel; a stable LDS radix sort of (entry, slot) tags by entry (12 bits, two passes)
// makes the visits of a voxel contiguous and keeps them in slot (= point) order — its cost
// does not depend on how many visits a voxel collects.  The order of the groups inside
// a tile is irrelevant: a tile holds at most one run per voxel, and runs of different tiles
// keep their tile order through the stable global sort.
//
// Run descriptors are numbered across tiles by a decoupled look-back over tile_state (tile
// ids are tickets, so a tile only ever waits for tiles that already started).
struct TileOut {
  float2* vis;          // [V] per visit, in slot order: (u, point index as bits) — kept out of LDS so that
                        // three tiles fit a CU
  float2* rec_t;        // [V] operands, tile-grouped
  uint32_t* recc_t;     // [V] colours, tile-grouped
  uint32_t* dkey;       // run descriptors: voxel key,
  unsigned long long* dval;   // value array of the sort: position in rec_t | length << 32,
  uint32_t* last_pt;    // [V], sparse: at a run's position, the point of its last visit
};

template <bool kNormals>
__global__ __launch_bounds__(kTileThreads, 6) void ray_tiles(
    Params P, const float* __restrict__ xyz, const float* __restrict__ normals, const uint8_t* __restrict__ rgb, int npoints,
    const int32_t* __restrict__ offsets, int nclouds, const Pose* __restrict__ poses, Directory dir,
    Counters* __restrict__ ctr, const uint32_t* __restrict__ voff, uint32_t V,
    const uint32_t* __restrict__ tile_first, uint32_t ntiles, uint32_t* __restrict__ ticket,
    unsigned long long* __restrict__ tile_state, const uint32_t* __restrict__ rgbw, TileOut out) {
  __shared__ uint32_t skey[kTileSlots];     // voxel key of the visit in slot s
  __shared__ uint32_t bufA[kTileSlots];     // tags: group table entry << 12 | slot; sorted in place
  __shared__ uint32_t bufB[kTileSlots];     // the group hash table, then the sort's second buffer, then run heads
  __shared__ uint32_t wave_hist[kTileThreads / 64][kTileRadix];
  uint32_t* const gtab = bufB;              // representative slot of the voxel hashed to this entry
  __shared__ uint32_t wsum[kTileThreads / 64];
  __shared__ uint32_t sh_tile, sh_base;
  __shared__ unsigned long long ckey[kTileChunkCache];   // chunk id -> pool slot, the chunks this tile meets
  __shared__ int32_t cslot[kTileChunkCache];
  __shared__ int32_t cl_off[kTileCloudCache + 1];          // cloud offsets around the tile
  const int tid = threadIdx.x;

#if PLVS_TILE_PROBE
  unsigned long long tp_ = clock64();
#endif
  if (tid == 0) {
    sh_tile = atomicAdd(ticket, 1u);
  }
#pragma unroll
  for (int k = 0; k < kTileItems; ++k) gtab[tid + k * kTileThreads] = kTileEmpty;
  if (tid < kTileChunkCache) {
    ckey[tid] = kEmptyKey;
    cslot[tid] = -2;
  }
  __syncthreads();
  const uint32_t t = sh_tile;
  const uint32_t slot0 = t * kTileSlots;
  const uint32_t n = min((uint32_t)kTileSlots, V - slot0);
  const uint32_t first = tile_first[t];
  const uint32_t last = (t + 1 < ntiles) ? tile_first[t + 1] : (uint32_t)(npoints - 1);

  TILE_PROBE(0)
  // the clouds the tile's points belong to: offsets of up to kTileCloudCache of them in LDS
  const int cloud0 = cloud_of(offsets, nclouds, (int)first);
  const int ncl = min(nclouds - cloud0, kTileCloudCache);
  if (tid <= ncl) cl_off[tid] = offsets[cloud0 + tid];
  __syncthreads();

  // ---- phase 1a: the points of [first, last] that have visits in this tile, compacted (any order:
  // a visit's slot comes from voff).  On a shard most points of the range have none — their chunks
  // belong to other ranks, whole keyframes can look at chunks of other ranks only — and a wave
  // would otherwise walk 64 rays for the few lanes that do.  Stretches without visits are jumped
  // over by bisection on voff (uniform control flow).
  uint32_t* const tile_rays = bufA;   // free until phase 2; a tile holds <= kTileSlots such points
  if (tid == 0) sh_base = 0;
  __syncthreads();
  for (uint32_t base = first; base <= last;) {
    const uint32_t vb = voff[base];
    if (vb >= slot0 + n) break;                          // the rest belongs to later tiles
    const uint32_t end = min(base + (uint32_t)kTileThreads, last + 1);
    if (voff[end] == vb) {                               // nothing in [base, end)
      uint32_t lo = end, hi = last + 1;                  // voff[lo] == vb throughout
      if (voff[hi] == vb) break;
      while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (voff[mid] > vb) hi = mid; else lo = mid;
      }
      base = lo;                                         // point lo is the next one with visits
      continue;
    }
    const uint32_t i = base + tid;
    bool has = false;
    if (i < end) {
      const uint32_t o = voff[i], e = voff[i + 1];
      has = !(e == o || e <= slot0 || o >= slot0 + n);
    }
    const unsigned long long m = __ballot(has);
    uint32_t wbase = 0;
    if ((tid & 63) == 0 && m != 0ull) wbase = atomicAdd(&sh_base, (uint32_t)__popcll(m));
    wbase = __shfl(wbase, 0);
    if (has) tile_rays[wbase + __popcll(m & ((1ull << (tid & 63)) - 1ull))] = i;
    base = end;
  }
  __syncthreads();
  const uint32_t nrays = sh_base;
  __syncthreads();   // sh_base is reused by the look-back

  // ---- phase 1b: the visits of this tile, in slot (= point, then ray) order
  for (uint32_t r = tid; r < nrays; r += kTileThreads) {
    const uint32_t i = tile_rays[r];
    const uint32_t o = voff[i], e = voff[i + 1];
    const uint32_t n_lo = (o < slot0) ? slot0 - o : 0u;       // visits before it belong to the previous tile
    const uint32_t n_hi = min(e, slot0 + n) - o;              // visits from it on to the next one
    int cl = 0;
    while (cl + 1 < ncl && (int)i >= cl_off[cl + 1]) ++cl;
    if ((int)i >= cl_off[ncl]) cl = cloud_of(offsets, nclouds, (int)i) - cloud0;   // beyond the cached clouds
    const Pose pose = poses[cloud0 + cl];
    Ray ray;
    RayN aux;
    if (kNormals)
      make_ray_normal(P, pose, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], normals[3 * (size_t)i],
                      normals[3 * (size_t)i + 1], normals[3 * (size_t)i + 2], &ray, &aux);
    else if (!make_ray(P, pose, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], &ray))
      continue;
    RayCursor cur;
    OwnerCache owner;
    ray_begin(ray, &cur);
    int vx, vy, vz;
    int lcx = 0, lcy = 0, lcz = 0, lslot = -1;
    bool have_last = false;
    uint32_t nv = 0;
    while (nv < n_hi && ray_next(&cur, &vx, &vy, &vz)) {
      Visit v;
      const bool ok = kNormals ? resolve_visit_normal(P, aux, ray, vx, vy, vz, &v, &owner)
                               : resolve_visit(P, pose, ray, vx, vy, vz, &v, &owner);   // no early continue: see ray_count
      if (ok && nv >= n_lo) {
        if (!have_last || v.cx != lcx || v.cy != lcy || v.cz != lcz) {
          lcx = v.cx; lcy = v.cy; lcz = v.cz;
          have_last = true;
          lslot = tile_find_chunk(dir, ckey, cslot, lcx, lcy, lcz);
          if (lslot < 0) atomicOr(&ctr->err, kErrDirectoryMiss);
        }
        const uint32_t s = o + nv - slot0;
        skey[s] = (uint32_t)max(lslot, 0) * (uint32_t)kChunkVox + (uint32_t)v.vid;
        out.vis[slot0 + s] = make_float2(v.u, __uint_as_float(i));
      }
      nv += ok ? 1u : 0u;
    }
    if (nv < n_hi) atomicOr(&ctr->err, kErrDirectoryMiss);   // the count pass saw more visits: cannot happen
  }
  __syncthreads();

  TILE_PROBE(1)
  // ---- phases 2-4: group the visits by voxel (LDS hash table), stable LDS radix sort of the
  // (group, slot) tags, run heads (tsdf_tiles.hpp)
  const uint32_t ngroups = tile_group_sort_heads(skey, bufA, bufB, wave_hist, wsum, n, tid);
  uint16_t* const hp = reinterpret_cast<uint16_t*>(bufB);   // positions of the runs; the sorted tags are in bufA
  TILE_PROBE(3)
  // publish this tile's run count for the tiles behind it
  if (tid == 0 && t > 0) st_state(&tile_state[t], (1ull << 62) | ngroups);
  __syncthreads();   // hp complete

  TILE_PROBE(5)
  // ---- phase 5: operands out, in sorted order
#pragma unroll
  for (int k = 0; k < kTileItems; ++k) {
    const uint32_t j = tid + k * kTileThreads;
    if (j < n) {
      const uint32_t s = bufA[j] & 0xFFFu;
      const float2 vv = out.vis[slot0 + s];
      const size_t p = __float_as_uint(vv.y);
      const float tr = kNormals ? 4 * P.resolution : truncation_of(P, xyz[3 * p + 2]);
      const float wu = P.weight / (2.0f * tr);
      out.rec_t[slot0 + j] = make_float2(wu * vv.x, wu);
      out.recc_t[slot0 + j] = colour_roundtrip(rgb[3 * p + 0]) | (colour_roundtrip(rgb[3 * p + 1]) << 8) |
                              (colour_roundtrip(rgb[3 * p + 2]) << 16);
    }
  }

  // the tile's place in the run numbering (decoupled look-back, wave 0)
  tile_lookback(t, ngroups, ntiles, tile_state, &sh_base, &ctr->num_desc, tid);
  __syncthreads();

  // ---- phase 6: run descriptors out
  const uint32_t dbase = sh_base;
  for (uint32_t g = tid; g < ngroups; g += kTileThreads) {
    const uint32_t p0 = hp[g], p1 = hp[g + 1];
    const uint32_t d = dbase + g;
    out.dkey[d] = skey[bufA[p0] & 0xFFFu];
    out.dval[d] = (unsigned long long)(slot0 + p0) | ((unsigned long long)(p1 - p0) << 32);
    out.last_pt[slot0 + p0] = __float_as_uint(out.vis[slot0 + (bufA[p1 - 1] & 0xFFFu)].y);
  }
  TILE_PROBE(6)
}

// The truncating u8 colour mean of the ordered mode: one thread per voxel whose colour weight is below
// 254 folds its visits one by one through the sorted runs, exactly as the reference does, until the
// weight reaches 254 (at most 254 steps in the life of a voxel).
// kDivide: ColorVoxel::Integrate (a true division; the world-cloud-with-normals flavour) instead of IntegrateSimple.
template <bool kDivide>
__global__ __launch_bounds__(256) void fold_colours(
    const uint32_t* __restrict__ skeys, const unsigned long long* __restrict__ sval, uint32_t nd,
    const uint32_t* __restrict__ vj0, const uint32_t* __restrict__ recc_t, uint32_t* __restrict__ rgbw,
    const Counters* __restrict__ ctr) {
  // 1 / (1 + weight), the factor of ColorVoxel::IntegrateSimple, for every weight it can see
  __shared__ float inv_tab[256];
  inv_tab[threadIdx.x] = 1.f / (float)(1u + (uint32_t)threadIdx.x);
  __syncthreads();
  constexpr int kTurn = 16;   // visits per turn
  constexpr int kRuns = 8;    // run descriptors looked at per turn
  const uint32_t nvox = ctr->num_heads;
  for (uint32_t v0 = blockIdx.x * blockDim.x; v0 < nvox; v0 += gridDim.x * blockDim.x) {
    const uint32_t v = v0 + threadIdx.x;
    uint32_t key = 0, col = 254u << 24, jj = 0;
    if (v < nvox) {
      jj = vj0[v];
      key = skeys[jj];
      col = rgbw[key];
    }
    const bool fresh = v < nvox && (col >> 24) < 254u;
    // One flat loop for the whole wave.  A turn takes up to kTurn visits, across up to kRuns
    // consecutive runs of the voxel (runs are short: ~10 visits); the run descriptors of the next
    // turn are requested while the colours of this one are in flight — the fold is cheap, the
    // latency of a load per visit is not.
    bool active = fresh;
    uint32_t pos = 0;   // visits of run jj already folded
    unsigned long long d[kRuns];
    uint32_t dk[kRuns];
#pragma unroll
    for (int q = 0; q < kRuns; ++q) {
      const uint32_t jq = min(jj + q, nd - 1);
      d[q] = active ? sval[jq] : 0ull;
      dk[q] = active ? skeys[jq] : ~key;
    }
    while (__ballot(active) != 0ull) {
      // the runs at hand: which of them belong to the voxel, where each starts in the turn's
      // visit sequence (prefix of the remaining lengths)
      uint32_t start[kRuns + 1];   // visit index (within the turn's sequence) at which run q begins
      uint32_t nvalid = 0;          // leading runs of the voxel among the descriptors
      start[0] = 0;
#pragma unroll
      for (int q = 0; q < kRuns; ++q) {
        const bool mine = nvalid == (uint32_t)q && (jj + q < nd) && dk[q] == key;
        nvalid += mine ? 1u : 0u;
        const uint32_t len = mine ? (uint32_t)(d[q] >> 32) - (q == 0 ? pos : 0u) : 0u;
        start[q + 1] = start[q] + len;
      }
      const bool voxel_ends = nvalid < (uint32_t)kRuns;   // the voxel's runs end within the descriptors at hand
      const uint32_t avail = start[kRuns];
      const uint32_t taken = min(avail, (uint32_t)kTurn);
      // addresses of this turn's visits
      uint32_t addr[kTurn];
#pragma unroll
      for (int e = 0; e < kTurn; ++e) {
        uint32_t a0 = (uint32_t)d[0] + pos + (uint32_t)e;
#pragma unroll
        for (int q = 1; q < kRuns; ++q) a0 = ((uint32_t)e >= start[q]) ? (uint32_t)d[q] + ((uint32_t)e - start[q]) : a0;
        addr[e] = a0;
      }
      // where the turn stops: the run holding visit number `taken` (or past the last one)
      uint32_t run = 0;
#pragma unroll
      for (int q = 1; q <= kRuns; ++q) run += (taken >= start[q]) ? 1u : 0u;   // runs fully consumed
      const uint32_t p = (run < (uint32_t)kRuns) ? taken - start[run] + (run == 0 ? pos : 0u) : 0u;
      uint32_t c[kTurn];
#pragma unroll
      for (int e = 0; e < kTurn; ++e) c[e] = (active && (uint32_t)e < taken) ? recc_t[addr[e]] : 0u;
      // where the next turn starts, and its descriptors
      // (run / p after the loop: p may equal the count of run `run`; the skip at the top handles it)
      const uint32_t jj_next = jj + min(run, (uint32_t)kRuns);
      const uint32_t pos_next = (run < (uint32_t)kRuns) ? p : 0u;
      unsigned long long dn[kRuns];
      uint32_t dkn[kRuns];
#pragma unroll
      for (int q = 0; q < kRuns; ++q) {
        const uint32_t jq = min(jj_next + q, nd - 1);
        dn[q] = active ? sval[jq] : 0ull;
        dkn[q] = active && (jj_next + q < nd) ? skeys[jq] : ~key;
      }
      if (active) {
#pragma unroll
        for (int e = 0; e < kTurn; ++e) {
          const uint32_t cw = col >> 24;
          if ((uint32_t)e < taken && cw < 254u) {   // ColorVoxel::IntegrateSimple, visit by visit
            uint32_t red, green, blue;
            if (kDivide) {   // ColorVoxel::Integrate (ColorVoxel.h:68-89); Saturate cannot bind: a mean of bytes
              const float den = (float)(cw + 1u);
              red = (uint32_t)(uint8_t)((float)(cw * (col & 255u) + (c[e] & 255u)) / den);
              green = (uint32_t)(uint8_t)((float)(cw * ((col >> 8) & 255u) + ((c[e] >> 8) & 255u)) / den);
              blue = (uint32_t)(uint8_t)((float)(cw * ((col >> 16) & 255u) + ((c[e] >> 16) & 255u)) / den);
            } else {
              const float inv = inv_tab[cw];
              red = (uint32_t)(uint8_t)((float)(cw * (col & 255u) + (c[e] & 255u)) * inv);
              green = (uint32_t)(uint8_t)((float)(cw * ((col >> 8) & 255u) + ((c[e] >> 8) & 255u)) * inv);
              blue = (uint32_t)(uint8_t)((float)(cw * ((col >> 16) & 255u) + ((c[e] >> 16) & 255u)) * inv);
            }
            col = red | (green << 8) | (blue << 16) | ((cw + 1u) << 24);
          }
        }
        if ((col >> 24) >= 254u || (voxel_ends && taken == avail) || taken == 0) active = false;
      }
      jj = jj_next;
      pos = pos_next;
#pragma unroll
      for (int q = 0; q < kRuns; ++q) { d[q] = dn[q]; dk[q] = dkn[q]; }
    }
    if (fresh) rgbw[key] = col;
  }
}

// The order-dependent part: one thread per voxel run, 64 runs per wave, eight records per
// run and pass.  A single wave issues about one instruction every four cycles, and the
// longest run of the call is a serial chain, so the kernel is built to keep the
// instructions per step low and every wait off that chain:
//  * Loads: a lane walking its own run touches 64 different cache lines per load
//    instruction.  Here the wave fetches a pass cooperatively — four lanes read the eight
//    consecutive records (64 B) of one run, sixteen runs per load instruction — and hands
//    the records to their lanes through LDS (XOR-swizzled 16-byte units).  Four passes are
//    in flight in registers; the LDS hop is pipelined one pass deep (four buffers, no
//    barrier: one wave, and LDS operations of a wave execute in order).
//  * Arithmetic: w_k = w_{k-1} + wu_k does not depend on the running sdf, so the weights and
//    their reciprocals of the NEXT pass are computed beside the sdf recurrence of the
//    current pass; the recurrence itself is dist_update_rcp (mul, add, mul, fma, fma).
//    v_rcp_f32 plus one Newton step gives the correctly rounded reciprocal for every
//    binary32 significand on gfx950 (plvs_hip_selftest_rcp checks all 2^23 of them).
//  * A lane whose pass contains the end of its run (negative weight = last record), or an
//    operand outside the exact range of the reciprocal form, redoes that pass step by step.
//    Nothing is loaded there: the keyframe id of the voxel is written by gather_runs,
//    the longest run is reduced once per wave.
constexpr int kChainBatch = 8;
constexpr int kChainSets = 4;

__device__ __forceinline__ float rcp_rn(float b) {
  const float y0 = __builtin_amdgcn_rcpf(b);
  const float e = fmaf(-b, y0, 1.0f);
  return fmaf(e, y0, y0);
}

struct __attribute__((packed, aligned(8))) RecPair {   // two consecutive float2 records
  float x0, y0, x1, y1;
};

// 16-byte unit u (records 2u, 2u+1) of a run inside a staging buffer
__device__ __forceinline__ int stage_unit(int run, int u) { return run * 4 + ((u ^ (run >> 1)) & 3); }

__global__ __launch_bounds__(64) void chain_runs(
    const uint32_t* __restrict__ vj0, const uint32_t* __restrict__ skeys, const uint32_t* __restrict__ dst,
    uint32_t nrec, const float2* __restrict__ rec, Counters* __restrict__ ctr, float* __restrict__ sdf,
    float* __restrict__ weight) {
  __shared__ float4 stage[kChainSets][64 * 4];
  const int l = threadIdx.x;
  const uint32_t nheads = ctr->num_heads;
  const uint32_t last_pair = nrec - 1;   // the record buffer holds at least nrec + 1 records
  // the grid is an upper bound (the run count is only known on the device): surplus waves
  // leave at once, and a wave takes further groups of 64 runs if the grid was capped
  for (uint32_t group = blockIdx.x; group * 64u < nheads; group += gridDim.x) {
    const uint32_t h = group * 64u + (uint32_t)l;
    bool live = h < nheads;
    const uint32_t j0 = live ? vj0[h] : 0u;            // first run of the voxel
    const uint32_t r0 = live ? dst[j0] : 0u;           // its first record
    const size_t a = live ? (size_t)skeys[j0] : 0;     // slot*4096 + vid
    float s = live ? sdf[a] : 0.0f;
    float w = live ? weight[a] : 1.0f;
    uint32_t my_len = 0;
    // load i serves runs 16 i .. 16 i + 15; this lane fetches records 2q, 2q+1 (q = lane & 3)
    // of run 16 i + (lane >> 2)
    uint32_t base[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) base[i] = (uint32_t)__shfl((int)r0, 16 * i + (l >> 2)) + 2u * (uint32_t)(l & 3);
    const char* const rec_bytes = reinterpret_cast<const char*>(rec);

    RecPair G[kChainSets][4];
    auto fetch = [&](uint32_t pass, RecPair (&g)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        g[i] = *reinterpret_cast<const RecPair*>(rec_bytes + (min(base[i] + pass * kChainBatch, last_pair) << 3));
    };
    auto to_stage = [&](int buf, const RecPair (&g)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        stage[buf][stage_unit(16 * i + (l >> 2), l & 3)] = make_float4(g[i].x0, g[i].y0, g[i].x1, g[i].y1);
    };
    auto from_stage = [&](int buf, float4 (&R)[4]) {
#pragma unroll
      for (int u = 0; u < 4; ++u) R[u] = stage[buf][stage_unit(l, u)];
    };
    // weights, reciprocals and end markers of a pass, from the weight the run has before it
    struct Prepared {
      float x[kChainBatch], wn[kChainBatch], y[kChainBatch], wu[kChainBatch];
      bool plain;   // the pass holds the end of the run, or a weight outside the exact range
    };
    auto prepare = [&](const float4 (&R)[4], float w_in, Prepared& P) {
      uint32_t signs = 0;
      float wk = w_in;
#pragma unroll
      for (int k = 0; k < kChainBatch; ++k) {
        const float4 t = R[k >> 1];
        P.x[k] = (k & 1) ? t.z : t.x;
        P.wu[k] = (k & 1) ? t.w : t.y;
        signs |= __float_as_uint(P.wu[k]);
        wk = fabsf(P.wu[k]) + wk;
        P.wn[k] = wk;
        P.y[k] = rcp_rn(wk);
      }
      // the weights grow along the pass: the first and the last bound them all
      P.plain = ((signs >> 31) != 0) | !(P.wn[0] >= 0x1p-20f) | !(P.wn[kChainBatch - 1] <= 0x1p40f);
    };

    static_assert(kChainSets == 4, "the rotation below is written for four register sets / buffers");
    fetch(0, G[0]);
    fetch(1, G[1]);
    fetch(2, G[2]);
    fetch(3, G[3]);
    to_stage(0, G[0]);
    fetch(4, G[0]);
    to_stage(1, G[1]);
    fetch(5, G[1]);
    float4 R[4];
    Prepared cur, nxt;
    from_stage(0, R);
    prepare(R, w, cur);

    uint32_t pass = 0;
    // Pass p: records of pass p+2 go to LDS (and their registers are refilled with pass p+6),
    // pass p+1 is read from LDS and prepared, the recurrence of pass p runs.
#define PLVS_CHAIN_PASS(J, CUR, NXT)                                                                      \
  {                                                                                               \
    PLVS_PROBE(0)                                                                                 \
    from_stage(((J) + 1) & 3, R);                                                                 \
    to_stage(((J) + 2) & 3, G[((J) + 2) & 3]);                                                    \
    fetch(pass + 2 + kChainSets, G[((J) + 2) & 3]);                                               \
    PLVS_PROBE(1)                                                                                 \
    float s_fast = s, w_fast = w, amin = 0x1p0f, amax = 0x1p0f;                                   \
    _Pragma("unroll") for (int k = 0; k < kChainBatch; ++k)                                       \
        dist_update_rcp(s_fast, w_fast, CUR.x[k], CUR.wn[k], CUR.y[k], amin, amax);               \
    PLVS_PROBE(2)                                                                                 \
    prepare(R, CUR.wn[kChainBatch - 1], NXT);                                                     \
    PLVS_PROBE(3)                                                                                 \
    const bool redo = live && (CUR.plain || !(amin >= 0x1p-60f) || !(amax <= 0x1p60f));           \
    if (__ballot(redo) != 0ull && redo) {                                                         \
      bool fin = false;                                                                           \
      _Pragma("unroll") for (int k = 0; k < kChainBatch; ++k) {                                   \
        if (!fin) {                                                                               \
          float s2 = s, w2 = w, mn = 0x1p0f, mx = 0x1p0f;                                         \
          dist_update_rcp(s2, w2, CUR.x[k], CUR.wn[k], CUR.y[k], mn, mx);                         \
          if ((mn >= 0x1p-60f) && (mx <= 0x1p60f) && (CUR.wn[k] >= 0x1p-20f) &&                   \
              (CUR.wn[k] <= 0x1p40f)) {                                                           \
            s = s2;                                                                               \
            w = w2;                                                                               \
          } else {                                                                                \
            dist_update(s, w, CUR.x[k], fabsf(CUR.wu[k]));                                        \
          }                                                                                       \
          if (CUR.wu[k] < 0.0f) {                                                                 \
            fin = true;                                                                           \
            my_len = pass * kChainBatch + (uint32_t)k + 1u;                                       \
          }                                                                                       \
        }                                                                                         \
      }                                                                                           \
      if (fin) {                                                                                  \
        sdf[a] = s;                                                                               \
        weight[a] = w;                                                                            \
        live = false;                                                                             \
      }                                                                                           \
    } else {                                                                                      \
      s = s_fast;                                                                                 \
      w = w_fast;                                                                                 \
    }                                                                                             \
    ++pass;                                                                                       \
    PLVS_PROBE(4)                                                                                 \
    if (__ballot(live) == 0ull) break;                                                            \
  }
#if PLVS_CHAIN_PROBE
    unsigned long long pt[5] = {0, 0, 0, 0, 0}, pacc[4] = {0, 0, 0, 0};
    const unsigned long long p_begin = clock64(), p_wall = wall_clock64();
#define PLVS_PROBE(i)                                                                \
  __builtin_amdgcn_sched_barrier(0);                                                 \
  pt[i] = clock64();                                                                 \
  __builtin_amdgcn_sched_barrier(0);                                                 \
  if ((i) > 0) pacc[(i) - 1] += pt[i] - pt[(i) - 1];
#else
#define PLVS_PROBE(i)
#endif
    for (;;) {
      PLVS_CHAIN_PASS(0, cur, nxt)
      PLVS_CHAIN_PASS(1, nxt, cur)
      PLVS_CHAIN_PASS(2, cur, nxt)
      PLVS_CHAIN_PASS(3, nxt, cur)
    }
#undef PLVS_CHAIN_PASS
#if PLVS_CHAIN_PROBE
    if (l == 0) {
      const unsigned long long key = (unsigned long long)pass << 40;
      atomicMax(&ctr->probe[0], key | (clock64() - p_begin));
      atomicMax(&ctr->probe[1], key | (wall_clock64() - p_wall));
      for (int i = 0; i < 4; ++i) atomicMax(&ctr->probe[2 + i], key | pacc[i]);
    }
#endif
    // longest run of the call = the serial-latency floor of this stage (reported in the stats)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) my_len = max(my_len, (uint32_t)__shfl_xor((int)my_len, off));
    if (l == 0 && my_len > ctr->max_run) atomicMax(&ctr->max_run, my_len);
  }
}

// ------------------------------------------------------------------ carving (T7)
// Chisel::IntegratePointCloudWidthDepth, the part before the point cloud (Chisel.cpp:394-438):
// every existing chunk the camera frustum "intersects" (ChunkManager::GetChunkIDsIntersecting,
// ChunkManager.cpp:241-271 + Frustum::Intersects, Frustum.cpp:40-78) goes through
// ProjectionIntegrator::CarveWithDepth (ProjectionIntegrator.h:271-338): a known voxel whose
// centre projects onto the depth image, lies more than truncation + carvingDist in front of
// the measured surface and has sdf < 1e-5 is Reset().  Block-centric and order free: one
// thread per voxel of every allocated chunk.  The frustum (six planes, bounding box) is built
// on the host exactly as the reference does (carve_frustum below).
struct CarveCamera {
  float R[9], t[3];            // camera -> world
  float fx, fy, cx, cy;
  float width, height;         // as floats (IsPointOnImage compares float coordinates with them)
  int iwidth;
  float plane_n[6][3], plane_d[6];   // far, near, top, bottom, left, right
  int lo[3], hi[3];            // candidate chunk ids, inclusive (minID - 1 .. maxID + 1)
  float carving_dist;
};

__global__ __launch_bounds__(256) void carve_chunks(Params P, CarveCamera C, const float* __restrict__ depth,
                                                    const int32_t* __restrict__ slot_ids, int num_chunks,
                                                    float* __restrict__ sdf, float* __restrict__ weight,
                                                    uint32_t* __restrict__ vkfid, uint32_t* __restrict__ carved) {
  const int slot = blockIdx.x >> 4;   // 16 blocks of 256 voxels per chunk
  if (slot >= num_chunks) return;
  const int id[3] = {slot_ids[3 * slot], slot_ids[3 * slot + 1], slot_ids[3 * slot + 2]};
  // ---- is the chunk on the reference's list?
  for (int k = 0; k < 3; ++k)
    if (id[k] < C.lo[k] || id[k] > C.hi[k]) return;
  const float bmin[3] = {(float)(id[0] * 16) * P.resolution, (float)(id[1] * 16) * P.resolution,
                         (float)(id[2] * 16) * P.resolution};
  const float ext = 16.0f * P.resolution;
  bool hit = false;
  for (int p = 0; p < 6 && !hit; ++p) {
    float v[3];
    for (int k = 0; k < 3; ++k) v[k] = (C.plane_n[p][k] < 0.0f) ? bmin[k] : bmin[k] + ext;
    hit = sum3(v[0] * C.plane_n[p][0], v[1] * C.plane_n[p][1], v[2] * C.plane_n[p][2]) + C.plane_d[p] > 0.0f;
  }
  if (!hit) return;
  // ---- CarveWithDepth for this thread's voxel
  const int i = ((blockIdx.x & 15) << 8) | threadIdx.x;
  const size_t a = (size_t)slot * kChunkVox + (size_t)i;
  bool updated = false;
  if (!((double)weight[a] <= 1e-15)) {
    const int lx = i & 15, ly = (i >> 4) & 15, lz = i >> 8;
    const float cen[3] = {((float)lx * P.resolution + P.half_voxel) + bmin[0],
                          ((float)ly * P.resolution + P.half_voxel) + bmin[1],
                          ((float)lz * P.resolution + P.half_voxel) + bmin[2]};
    const float d0 = cen[0] - C.t[0], d1 = cen[1] - C.t[1], d2 = cen[2] - C.t[2];
    float pc[3];
    for (int r = 0; r < 3; ++r) pc[r] = sum3(C.R[r] * d0, C.R[3 + r] * d1, C.R[6 + r] * d2);   // Rcw = R^T
    const float inv_z = 1.0f / pc[2];
    const float u = C.fx * pc[0] * inv_z + C.cx, v = C.fy * pc[1] * inv_z + C.cy;
    if (!(pc[2] < 0) && (u >= 0 && v >= 0 && u < C.width && v < C.height)) {
      const float d = depth[(size_t)(int)v * (size_t)C.iwidth + (size_t)(int)u];
      if (!isnan(d)) {
        const float trunc = truncation_of(P, d);
        if (d - pc[2] > trunc + C.carving_dist && (double)sdf[a] < 1e-5) {
          sdf[a] = 99999.0f;   // DistVoxel::Reset
          weight[a] = 0.0f;
          vkfid[a] = 0u;
          updated = true;
        }
      }
    }
  }
  if (__syncthreads_or(updated) && threadIdx.x == 0) carved[slot] = 1u;
}

__global__ void carve_collect(const uint32_t* __restrict__ carved, int num_chunks, uint32_t* __restrict__ list,
                              Counters* __restrict__ ctr) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < num_chunks && carved[s]) list[atomicAdd(&ctr->num_updated, 1u)] = (uint32_t)s;
}

// Self-test of the device-wide stable radix sort (device_utils.hip) at sizes on both sides of its two scatter paths:
// pseudo-random keys below 2^bits (many equal keys when bits is small), values = original positions.
__global__ void selftest_sort_fill(uint32_t* __restrict__ keys, uint32_t* __restrict__ v32, unsigned long long* __restrict__ v64,
                                   uint32_t n, int bits, uint32_t seed) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t x = (i + 1u) * 2654435761u + seed;
  x ^= x << 13; x ^= x >> 17; x ^= x << 5;
  x ^= x << 13; x ^= x >> 17; x ^= x << 5;
  keys[i] = bits >= 32 ? x : (x & ((1u << bits) - 1u));
  if (v32) v32[i] = i;
  if (v64) v64[i] = ((unsigned long long)~i << 32) | i;
}
// mismatches[0]: order / stability violations, [1]: a pair whose key is not the key its value's position held
__global__ void selftest_sort_check(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ keys,
                                    const uint32_t* __restrict__ v32, const unsigned long long* __restrict__ v64, uint32_t n,
                                    int bit_lo, int bit_hi, uint32_t* __restrict__ mismatches) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t mask = (bit_hi - bit_lo >= 32 ? 0xFFFFFFFFu : ((1u << (bit_hi - bit_lo)) - 1u)) << bit_lo;
  const uint32_t pos = v32 ? v32[i] : (uint32_t)v64[i];
  if (pos >= n || keys_in[pos] != keys[i] || (v64 && (uint32_t)(v64[i] >> 32) != ~pos)) atomicAdd(&mismatches[1], 1u);
  if (i + 1 < n) {
    const uint32_t a = keys[i] & mask, b = keys[i + 1] & mask;
    const uint32_t pn = v32 ? v32[i + 1] : (uint32_t)v64[i + 1];
    if (a > b || (a == b && pos >= pn)) atomicAdd(&mismatches[0], 1u);
  }
}

// Hardware assumption of chain_runs, checked exhaustively: rcp_rn(b) == RN(1/b) for every
// significand at the given exponent.
__global__ void selftest_rcp_kernel(int exponent, uint32_t* __restrict__ mismatches) {
  const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
  const float b = __uint_as_float(((uint32_t)(exponent + 127) << 23) | m);
  if (__float_as_uint(rcp_rn(b)) != __float_as_uint(1.0f / b)) atomicAdd(mismatches, 1u);
}

}  // namespace

struct ChiselDeformState;   // tsdf_chisel_deform.hpp (included at the end of this file)

struct plvs_tsdf_chisel {
  plvs_tsdf_chisel_params prm;
  Params P;
  Directory dir;
  float* sdf = nullptr;
  float* weight = nullptr;
  uint32_t* kfid = nullptr;
  uint32_t* rgbw = nullptr;
  Counters* d_ctr = nullptr;
  Counters* h_ctr = nullptr;  // pinned
  int num_chunks = 0;         // host mirror
  bool poisoned = false;
  // per-call scratch
  DevBuf<uint32_t> counts, heads, updated, scratch;
  DevBuf<float2> rec, rec_t;         // operands in voxel order / grouped per tile
  DevBuf<uint32_t> recc_t;           // colours, grouped per tile (folded through the sorted runs)
  DevBuf<uint32_t> dkey0, dkey1, run_cnt, run_dst, last_pt;   // run descriptors
  DevBuf<unsigned long long> didx0, didx1;
  DevBuf<uint32_t> tile_first, block_first;
  DevBuf<unsigned long long> tile_state;   // [0]: ticket, [1..]: look-back state per tile
  DevBuf<Pose> poses;
  ChiselDeformState* dfm = nullptr;    // Chisel::Deform: the reference's chunk-map order, kept once enable_deform is on
  void* ext = nullptr;                 // see ChiselMapView::ext
  void (*ext_free)(void*) = nullptr;
  // halo of a sharded map (meshing): ghost copies of other ranks' chunks in the pool slots past num_chunks
  Directory gdir{};                    // id -> ghost slot / kGhostAbsent (allocated by the first import)
  int ghost_count = 0;                 // ghost chunks (pool slots taken) since the last halo_clear
  long long ghost_entries = 0;         //   and directory entries ("absent" answers included)
  DevBuf<uint32_t> halo_row;           // payload row of each request (prefix of the found flags)
  unsigned long long* miss_keys = nullptr;   // the chunks the last meshing pass looked for and did not have
  int32_t* miss_ids = nullptr;
  uint32_t* miss_count = nullptr;
  uint32_t miss_cap = 0, miss_mask = 0;
  DevBuf<int32_t> offsets;
  // host-flavour staging
  DevBuf<float> st_xyz, st_Twc, st_nrm;
  DevBuf<uint8_t> st_rgb;
  DevBuf<uint32_t> st_kfid;
  DevBuf<uint32_t> st_pos, st_scan, st_off;   // depth-image entry of an ordered / sharded / deforming handle: the clouds' scan
  plvs_tsdf_stats stats{};
  uint32_t last_updated = 0;
  // queued key-frame clouds (plvs_hip_tsdf_chisel_queue / _flush): uploaded, not yet integrated
  DevBuf<float> q_xyz;
  DevBuf<uint8_t> q_rgb;
  DevBuf<uint32_t> q_kfid;
  std::vector<int32_t> q_offsets;   // [clouds + 1] once anything is queued
  std::vector<float> q_Twc;         // 12 per cloud
  bool q_kfid_given = false;
  // single-walk pipeline (tsdf_walk.hpp)
  WalkCounters* d_wctr = nullptr;   // [2]: the call's counters, the colour pass's voxel list
  WalkCounters* h_wctr = nullptr;   // pinned
  uint32_t* h_seq = nullptr;        // pinned, coherent: the sequence number of the last publish_counters that has landed
  uint32_t seq_next = 0;
  double wait_ema_us[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};       // how long the host's last waits for the published counters took (wait_published)
  DevBuf<uint4> w_rec, w_seg, w_sorted_seg;
  // a long call's runs chunk by chunk (runs_count ... parts_place): the segments' run descriptors and the runs of the
  // same row in the block's earlier tiles; runs per (row, block); chunk slot -> place among the updated; region, runs and
  // first part of every row; the parts' rows and histograms
  DevBuf<uint4> w_rseg, w_rpre;
  DevBuf<uint32_t> w_run_matrix, w_active_idx, w_item_base, w_item_cnt, w_item_part0, w_part_item, w_phist, w_row_heads, w_row_tot;
  hipEvent_t ev_zero = nullptr;  // the run matrix is zero (side stream -> caller's stream)
  hipEvent_t ev_seg = nullptr;   // the updated chunks are listed (seg_scan; caller's stream -> side stream)
  int last_chain = 0;            // (developer trace) the last call's colour chain: 0 on its own counts, 1 predicted, 2 collected
  bool last_chain_skipped = false;   //   ... and whether it had to be repeated
  DevBuf<uint32_t> w_chunk_nseg, w_chunk_off, w_chunk_fill, w_active_off, w_masks, w_dummy, w_seg_cnt, w_tile_visits, w_deferred;
  DevBuf<uint32_t> w_part_off, w_multi_idx;          // apply stage: parts of the updated chunks
  DevBuf<long long> pa_wuu;                          //   accumulators of the chunks applied in parts (zero between calls)
  DevBuf<unsigned long long> pa_w;
  DevBuf<uint32_t> pa_last, pa_cnt, pa_done;
  uint32_t multi_cap = 0;
  uint32_t part_segs = kPartSegs, part_min = kPartMin;   // (plvs_hip_tsdf_chisel_set_apply_parts)
  bool third_pass = false;                                // a 4096-entry pass behind the 1024- and the 2048-entry one (the call before needed it)
  bool walk_small = false, walk_small_used = false;      // first-pass table of the order-free walk: 1024 entries instead of 2048
  DevBuf<uint32_t> w_runkey, w_run_cnt, w_run_off, w_val0, w_val1;   // runs: per-tile regions of 2^run_r1_log2 slots
  int32_t* h_offsets = nullptr;      // pinned copy of the call's cloud offsets
  size_t h_offsets_cap = 0;
  // run slots per tile (log2): 2048 from the start — a tile of the 2048-entry walk can need 1792, and growing the regions
  // later means repeating a call and re-allocating its largest buffers (ntiles << r1_log2 masks of 64 B) in the middle of
  // a job; a tile of the 4096-entry walk that needs more still grows them once
  uint32_t run_r1_log2 = 11;
  float scale_u = 1.f, scale_w = 1.f;   // fixed-point scales of the order-free accumulators (powers of two)
  int stage_set = 0;                    // which pipeline the stage times belong to
  // ray-sharded multi-GPU integrate (tsdf_shard.hpp)
  Directory xdir{};                  // the walk directory: every chunk the rank's tiles have crossed (ids only)
  int32_t* d_xcount = nullptr;
  uint32_t* x_sat = nullptr;         //   + one bit per voxel: its owner has reported the colour saturated
  DevBuf<uint32_t> sh_ctl;
  uint32_t* h_sh_ctl = nullptr;      // pinned [320]
  uint32_t* h_sh_off = nullptr;      // pinned [128 + 132]: region offsets on their way to the device (pack | apply)
  DevBuf<uint4> sh_seg_reg, sh_rec_reg;
  DevBuf<uint32_t> sh_nrec, sh_owner, sh_slot_owner, sh_src_off, sh_run_ctr, sh_vkey, sh_sat;
  DevBuf<uint4> sh_run_first;   // per run of a sharded walk: the first wire record's spans + its number of records
  uint32_t sh_nt = 0, sh_runs = 0, sh_nsat = 0;
  DevBuf<int32_t> sh_wait;   // saturated voxels not yet announced ({chunk x, y, z, voxel}): [sh_wait_first, + sh_wait_count)
  uint32_t sh_wait_first = 0, sh_wait_count = 0;
  long long* h_sh_counts = nullptr;  // pinned
  int sh_n = 0, sh_nclouds = 0;      // the call in flight (shard_walk -> shard_pack -> shard_apply)
  uint32_t sh_ntiles = 0;            // tiles of the whole point stream
  std::vector<int32_t> sh_tiletab;   // the call's offsets + tile table (host copy)
  int sh_phase = 0;
  plvs_tsdf_stats sh_stats{};
  hipStream_t side = nullptr;   // second stream for the colour chain
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool small_runs_known = false;   // the runs of the last small call (integrate_walk_acc launches the next one's colour chain on them)
  uint32_t small_runs_last = 0, small_tiles_last = 1;
  // optional per-stage timing (HIP events on the caller's stream)
  bool profiling = false;
  hipEvent_t ev[kNumStages + 1] = {};
  double stage_ms[kNumStages] = {};
  int64_t prof_calls = 0;
};

template <typename T>
static hipError_t grow_keep(DevBuf<T>& b, size_t used, size_t want) {   // reserve() that keeps the first `used` elements
  if (want <= b.cap) return hipSuccess;
  DevBuf<T> nb;
  hipError_t e = nb.reserve(std::max(want, 2 * b.cap));
  if (e != hipSuccess) return e;
  if (used) e = hipMemcpy(nb.p, b.p, used * sizeof(T), hipMemcpyDeviceToDevice);
  if (e != hipSuccess) { nb.release(); return e; }
  b.release();
  b = nb;
  return hipSuccess;
}

extern "C" int plvs_hip_tsdf_chisel_flush(plvs_tsdf_chisel* h);
#define PLVS_FLUSH_QUEUE(h)                                        \
  do {                                                             \
    if ((h) && !(h)->q_offsets.empty()) {                          \
      const int rc_flush_ = plvs_hip_tsdf_chisel_flush(h);         \
      if (rc_flush_ != PLVS_OK) return rc_flush_;                  \
    }                                                              \
  } while (0)

// Chisel::Deform support (tsdf_chisel_deform.hpp)
static void deform_state_clear(plvs_tsdf_chisel* h);
static void deform_state_free(plvs_tsdf_chisel* h);
static void deform_note_created(plvs_tsdf_chisel* h, int cx, int cy, int cz);
static int deform_track_begin(plvs_tsdf_chisel* h, const float* d_xyz, const float* d_normals, int n, int nclouds,
                              const float* d_Twc, hipStream_t s);
static int deform_track_end(plvs_tsdf_chisel* h, hipStream_t s);

// The reference's cloud of every depth image of a call, for the handles that take point streams (ordered / sharded /
// deform-tracking): cell = (image, grid pixel) in raster order; mark -> exclusive scan -> emit.
__global__ __launch_bounds__(256) void grid_cloud_mark(GridSrc g, int nclouds, uint32_t* __restrict__ flag) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, ngrid = (size_t)g.gw * g.gh;
  if (i >= ngrid * (size_t)nclouds) return;
  const uint32_t c = (uint32_t)(i / ngrid), r = (uint32_t)(i - (size_t)c * ngrid), m = r / g.gw, n = r - m * g.gw;
  const float d = g.depth[(size_t)c * g.image_stride + (size_t)(m * g.step) * g.pitch + n * g.step];
  flag[i] = (((double)d > g.min_depth) && ((double)d < g.max_depth)) ? 1u : 0u;   // src/PointCloudMapping.cc:967
}
__global__ __launch_bounds__(256) void grid_cloud_emit(GridSrc g, int nclouds, const uint32_t* __restrict__ pos /* cells + 1 */,
                                                       const uint8_t* __restrict__ bgr, const uint32_t* __restrict__ kfid_of_image,
                                                       float* __restrict__ xyz, uint8_t* __restrict__ rgb,
                                                       uint32_t* __restrict__ kfid, uint32_t* __restrict__ offsets) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, ngrid = (size_t)g.gw * g.gh, cells = ngrid * (size_t)nclouds;
  if (i > cells) return;
  if (i == cells || i % ngrid == 0) offsets[i / ngrid] = pos[i];
  if (i == cells || pos[i + 1] == pos[i]) return;
  const uint32_t c = (uint32_t)(i / ngrid), r = (uint32_t)(i - (size_t)c * ngrid), m = r / g.gw, n = r - m * g.gw;
  const float d = g.depth[(size_t)c * g.image_stride + (size_t)(m * g.step) * g.pitch + n * g.step];
  const float2 cam = reinterpret_cast<const float2*>(g.cam)[r];
  const size_t o = pos[i];
  xyz[3 * o] = cam.x * d;        // :973-975
  xyz[3 * o + 1] = cam.y * d;
  xyz[3 * o + 2] = d;
  const uint8_t* px = bgr + (size_t)c * g.bgr_image_stride + (size_t)(m * g.step) * g.bgr_pitch + (size_t)(n * g.step) * 3u;
  rgb[3 * o] = px[0];            // :978-980: the point's r, g, b members take bytes 0, 1, 2 of the pixel
  rgb[3 * o + 1] = px[1];
  rgb[3 * o + 2] = px[2];
  kfid[o] = kfid_of_image ? kfid_of_image[c] : 0u;
}

static int read_counters(plvs_tsdf_chisel* h, hipStream_t s) {
  hipLaunchKernelGGL(publish_counters, dim3(1), dim3(64), 0, s, (const WalkCounters*)nullptr, h->d_ctr, (WalkCounters*)nullptr,
                     h->h_ctr);
  PLVS_KERNEL_CHECK();
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  return PLVS_OK;
}

// ------------------------------------------------------------------ single-walk pipeline (tsdf_walk.hpp)
constexpr int kWalkStages = 4;
constexpr unsigned kDeferGrid = 1024;   // workgroups of the general walk over the deferred tiles (it loops over the list)
// The lean walk comes with two table sizes (tsdf_walk.hpp, FastShared): 2048 entries at two tiles per CU for a call that
// fills the device, 4096 entries at one tile per CU for the tiles that overflowed 2048 — a dozen in a hundred on an office
// scene with points up to 5 m away — and for every tile of a call of at most kSmallCallTiles tiles (a few key frames: one
// tile per CU is all there is to run, and a deferral costs such a call a second walk's latency).  A tile owns
// kRecStride records (the larger table's limit) in the record buffer.
constexpr int kFastEntriesSmall = 1024, kFastEntries = 2048, kFastEntriesBig = 4096;
constexpr uint32_t kRecStride = kFastEntriesBig * 7 / 8;
constexpr uint32_t kSmallCallTiles = 320;
constexpr uint32_t kPredictTiles = 4096;   // calls up to this size (~25 key frames) launch their colour chain on the previous call's sizes
constexpr unsigned kListGrid = 512;      // workgroups of the large-table pass over the first list (it loops)
static_assert(kRecStride == (uint32_t)kWalkLimit, "a tile's record region holds a flush of the largest table");
const char* const kWalkStageNames[kWalkStages] = {"walk_tiles", "sort_segments", "apply_chunks", "fold_colours"};

// The host's wait for a publish_counters launch that carried sequence number `seq` on stream q: it polls the word for up to
// PLVS_TSDF_SPIN_US microseconds (default 3000; 0 = never), then sleeps on the stream.  Everything enqueued on q before that
// launch has completed when the word arrives (stream order), so this stands for hipStreamSynchronize(q) as far as the
// pipeline's own buffers and the caller's inputs are concerned.
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__) || defined(__arm__)
  __asm__ __volatile__("yield");
#endif
}
// The host's cost: the calling thread SLEEPS through most of the wait it expects (the handle remembers how long its last waits
// of this kind took: a 100-key-frame step's last wait is ~0.9 ms, a one-key-frame call's ~0.1 ms) and polls only for the
// rest — at most PLVS_TSDF_SPIN_US microseconds (default 400; 0: never poll) before it falls back to hipStreamSynchronize —
// so a SLAM thread next to it loses a core for a few hundred microseconds per call at worst, not for the length of the call.
// kind: 0 the end of a call, 1 the read in front of its colour chain; size_class: calls of a few key frames, of tens, of a hundred
// (their waits differ by an order of magnitude, and a SLAM system alternates them)
// Time between two stage events of a call the host has read the counters of.  The read polls a word the call's last kernel
// stores to pinned memory (wait_published) and can be ahead of the runtime's own book-keeping of the events recorded in
// front of that kernel: hipEventElapsedTime then says "not ready" (once in two thousand calls, measured) — the events are
// waited for and asked again.
static hipError_t stage_elapsed(float* ms, hipEvent_t a, hipEvent_t b) {
  hipError_t e = hipEventElapsedTime(ms, a, b);
  if (e == hipErrorNotReady) {
    (void)hipGetLastError();
    e = hipEventSynchronize(a);
    if (e == hipSuccess) e = hipEventSynchronize(b);
    if (e == hipSuccess) e = hipEventElapsedTime(ms, a, b);
  }
  return e;
}

static int wait_published(plvs_tsdf_chisel* h, uint32_t seq, hipStream_t q, int kind = 0, int size_class = 0) {
  double& ema = h->wait_ema_us[kind][size_class];
  static const int spin_us = plvs::env_int("PLVS_TSDF_SPIN_US", 400, 0, 10000000);
  if (spin_us > 0 && h->h_seq != nullptr) {
    const volatile uint32_t* const word = h->h_seq;
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    auto elapsed_us = [&]() {
      clock_gettime(CLOCK_MONOTONIC, &t1);
      return (double)(t1.tv_sec - t0.tv_sec) * 1e6 + (double)(t1.tv_nsec - t0.tv_nsec) * 1e-3;
    };
    bool done = *word == seq;
    if (!done && ema > 200.0) {   // most of an expected long wait is slept, not polled (timer slack: ~60 us)
      timespec nap;
      // (... of a wait of a whole long call — a chain queued without a read of the walk's counters — 70 %: its length follows
      // the view, +-20 % from call to call, and a nap that overshoots is paid in full)
      const double us = std::min(std::min(ema - 120.0, 0.7 * ema), 5000.0);
      nap.tv_sec = 0;
      nap.tv_nsec = (long)(us * 1e3);
      nanosleep(&nap, nullptr);
      done = *word == seq;
    }
    const double spin_from = done ? 0.0 : elapsed_us();
    for (uint32_t spins = 0; !done; ++spins) {
      done = *word == seq;
      if (done) break;
      cpu_relax();
      if ((spins & 255u) == 255u && elapsed_us() - spin_from > (double)spin_us) break;
    }
    if (done) {
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      ema = 0.75 * ema + 0.25 * elapsed_us();
      return PLVS_OK;
    }
    ema = 0.75 * ema + 0.25 * (elapsed_us() + 200.0);   // (longer than expected: sleep longer next time)
  }
  // (the runtime has not observed the stream's completion after a polled read; nothing below relies on it: buffers are
  // re-used in stream order, and hipFree — DevBuf::reserve — synchronises the device itself)
  PLVS_HIP_TRY(hipStreamSynchronize(q));
  return PLVS_OK;
}

static int read_walk_counters(plvs_tsdf_chisel* h, hipStream_t s, int size_class = 0) {
  const uint32_t seq = ++h->seq_next;
  hipLaunchKernelGGL(publish_counters, dim3(1), dim3(64), 0, s, h->d_wctr, h->d_ctr, h->h_wctr, h->h_ctr, h->h_seq, seq);
  PLVS_KERNEL_CHECK();
  return wait_published(h, seq, s, 0, size_class);
}

static int walk_fail(plvs_tsdf_chisel* h, uint32_t err) {
  h->poisoned = true;
  plvs::set_error("tsdf_chisel integrate: %s%s(err=%u)",
                  (err & kErrPoolFull) ? "chunk pool full (raise max_chunks) " : "",
                  (err & kErrCoordRange) ? "voxel coordinates beyond +-2^20 (outside the supported map extent) " : "", err);
  return PLVS_ERR_CAPACITY;
}

// Stable sort of the D runs walk_tiles left in the per-tile regions by voxel key: per voxel its runs
// in tile (= point) order; the value carried is the run's slot (tile = slot >> r1_log2, mask at slot * 8).
static int sort_runs(plvs_tsdf_chisel* h, uint32_t D, uint32_t ntiles, int num_chunks, hipStream_t s,
                     const uint32_t** skeys, const uint32_t** sval, const RunGuard* guard = nullptr) {
  // guard (a chain launched before the host knows the call's runs): D and num_chunks are BOUNDS, compact_runs pads the pairs
  // up to D and leaves the verdict in *guard->skip
  PLVS_HIP_TRY(h->dkey0.reserve(D));
  PLVS_HIP_TRY(h->dkey1.reserve(D));
  PLVS_HIP_TRY(h->w_val0.reserve(D));
  PLVS_HIP_TRY(h->w_val1.reserve(D));
  PLVS_HIP_TRY(h->w_run_off.reserve((size_t)ntiles + 1));
  PLVS_HIP_TRY(h->scratch.reserve(std::max(radix_scratch_words(D), scan_scratch_words(ntiles))));
  // (the caller has scanned run_cnt into w_run_off)
  int key_bits = 12;
  // (the count after this call's insertions; + 1 under a guard: its padding keys, all ones, must not be a voxel's)
  while ((1ll << (key_bits - 12)) < (long long)num_chunks + (guard ? 1 : 0)) ++key_bits;
  // (the plain sort of a known number of pairs: its status words are zeroed by the compaction, on the side — a launch of its
  // own otherwise, 25 us in front of the chain)
  const size_t zero_words = guard ? 0 : radix_sort_zero_words(D, 0, key_bits);
  RunGuard g = guard ? *guard : RunGuard{0xFFFFFFFFu, nullptr, nullptr, 0, nullptr, nullptr, 0u, nullptr, 0u};
  g.zero = zero_words ? h->scratch.p : nullptr;
  g.zero_words = (uint32_t)zero_words;
  hipLaunchKernelGGL(compact_runs, dim3(ceil_div(ntiles, 4) + (guard ? (guard->pad ? std::min<unsigned>(64u, ceil_div((size_t)D, 1024)) : 1u) : 0u)),
                     dim3(256), 0, s, h->w_runkey.p, h->w_run_cnt.p, h->w_run_off.p, ntiles, h->run_r1_log2, h->dkey0.p,
                     h->w_val0.p, g);
  bool second = false;
  if (guard && !guard->pad)   // the bound is loose: the sort takes the number of pairs from the device
    PLVS_HIP_TRY(radix_sort_pairs_bound(h->dkey0.p, h->w_val0.p, h->dkey1.p, h->w_val1.p, D, guard->total, 0, key_bits,
                                        h->scratch.p, s, &second));
  else if (zero_words)
    PLVS_HIP_TRY(radix_sort_pairs_zeroed(h->dkey0.p, h->w_val0.p, h->dkey1.p, h->w_val1.p, D, 0, key_bits, h->scratch.p, s,
                                         &second));
  else
    PLVS_HIP_TRY(radix_sort_pairs(h->dkey0.p, h->w_val0.p, h->dkey1.p, h->w_val1.p, D, 0, key_bits, h->scratch.p, s,
                                  &second));
  *skeys = second ? h->dkey1.p : h->dkey0.p;
  *sval = second ? h->w_val1.p : h->w_val0.p;
  return PLVS_OK;
}

// Accumulators for `chunks` chunks applied in parts (apply_chunks leaves them zero).
static int ensure_part_acc(plvs_tsdf_chisel* h, uint32_t chunks) {
  if (chunks <= h->multi_cap) return PLVS_OK;
  const size_t nv = (size_t)chunks * kChunkVox, nd = (size_t)chunks * kSlabs;
  h->pa_wuu.release(); h->pa_w.release(); h->pa_last.release(); h->pa_cnt.release(); h->pa_done.release();
  h->multi_cap = 0;
  PLVS_HIP_TRY(h->pa_wuu.reserve(nv));
  PLVS_HIP_TRY(h->pa_w.reserve(nv));
  PLVS_HIP_TRY(h->pa_last.reserve(nv));
  PLVS_HIP_TRY(h->pa_cnt.reserve(nv));
  PLVS_HIP_TRY(h->pa_done.reserve(nd));
  PLVS_HIP_TRY(hipMemset(h->pa_wuu.p, 0, nv * sizeof(long long)));
  PLVS_HIP_TRY(hipMemset(h->pa_w.p, 0, nv * sizeof(unsigned long long)));
  PLVS_HIP_TRY(hipMemset(h->pa_last.p, 0, nv * sizeof(uint32_t)));
  PLVS_HIP_TRY(hipMemset(h->pa_cnt.p, 0, nv * sizeof(uint32_t)));
  PLVS_HIP_TRY(hipMemset(h->pa_done.p, 0, nd * sizeof(uint32_t)));
  h->multi_cap = chunks;
  return PLVS_OK;
}

// Order-free mode: walk_tiles -> segment sort -> apply_chunks (+ the colour fold when the call met voxels
// whose colour weight is below 254).
// gsrc (plvs_hip_tsdf_chisel_integrate_depth_batch_dev): the clouds are depth images — tiles are 32 x 16 blocks of grid
// pixels (GridSrc, tsdf_walk.hpp), d_xyz is null, d_rgb = the colour images, d_kfid = one id per image, offsets =
// nclouds + 1 zeros (nothing reads them).
static int integrate_walk_acc(plvs_tsdf_chisel* h, const float* d_xyz, const uint8_t* d_rgb, const uint32_t* d_kfid,
                              int n, int nclouds, const int32_t* offsets, const float* d_Twc, hipStream_t s,
                              const GridSrc* gsrc = nullptr) {
  const int max_chunks = h->prm.max_chunks;
  const GridSrc grid = gsrc ? *gsrc : GridSrc{};
  // tiles: 512 consecutive points of one cloud (tsdf_directory.hpp)
  size_t tiles_of_call = 0;
  for (int c = 0; c < nclouds; ++c) tiles_of_call += ((size_t)(offsets[c + 1] - offsets[c]) + kWalkRays - 1) / kWalkRays;
  const size_t grid_tiles = gsrc ? (size_t)nclouds * grid.ntx * grid.nty : 0;
  if (gsrc) PLVS_REQUIRE(grid_tiles < 0x7FFFFFFFull, "too many images in one call");
  constexpr size_t kGridWords = (sizeof(GridSrc) + 3) / 4;
  static_assert(sizeof(GridSrc) % 4 == 0 && alignof(GridSrc) <= 8, "GridSrc travels as words behind the offsets");
  const size_t table_words = 2 * ((size_t)nclouds + 1) + 2 * tiles_of_call + (gsrc ? kGridWords : 0);
  if (h->h_offsets_cap < table_words) {   // pinned copy of the offsets + tile table + tile starts: the prologue kernel reads it
    if (h->h_offsets) (void)hipHostFree(h->h_offsets);
    h->h_offsets = nullptr;
    h->h_offsets_cap = 0;
    PLVS_HIP_TRY(hipHostMalloc((void**)&h->h_offsets, (2 * table_words + 64) * sizeof(int32_t)));
    h->h_offsets_cap = 2 * table_words + 64;
  }
  const uint32_t ntiles = gsrc ? (uint32_t)grid_tiles : plvs::tsdf::fill_tile_table(offsets, nclouds, h->h_offsets, kWalkRays);
  if (gsrc) {   // (zeros: the table is unused; the walk's copy of the grid description travels behind it)
    plvs::tsdf::fill_tile_table(offsets, nclouds, h->h_offsets, kWalkRays);
    memcpy(h->h_offsets + 2 * ((size_t)nclouds + 1), &grid, sizeof(GridSrc));
  }
  {   // the first point of every tile (the colour fold would otherwise search the cloud table once per RUN)
    int32_t* tf = h->h_offsets + 2 * ((size_t)nclouds + 1);
    size_t t = 0;
    for (int c = 0; c < nclouds; ++c)
      for (int32_t p = offsets[c]; p < offsets[c + 1]; p += kWalkRays) {
        tf[ntiles + t] = c;   // (and its cloud: the walk's tiles read both instead of searching: tile_span_tables)
        tf[t++] = p;
      }
  }
  PLVS_HIP_TRY(h->offsets.reserve(table_words));
  PLVS_HIP_TRY(h->tile_state.reserve((size_t)ntiles + 1));
  PLVS_HIP_TRY(h->w_chunk_nseg.reserve((size_t)max_chunks));
  PLVS_HIP_TRY(h->w_chunk_off.reserve((size_t)max_chunks + 1));
  PLVS_HIP_TRY(h->w_chunk_fill.reserve((size_t)max_chunks));
  PLVS_HIP_TRY(h->w_active_off.reserve((size_t)max_chunks + 1));
  PLVS_HIP_TRY(h->updated.reserve((size_t)max_chunks + 1));
  PLVS_HIP_TRY(h->w_seg_cnt.reserve(ntiles));
  PLVS_HIP_TRY(h->w_tile_visits.reserve(ntiles));
  PLVS_HIP_TRY(h->w_deferred.reserve(3 * (size_t)ntiles));   // (three lists: one behind each lean pass)
  // every tile owns kRecStride records / kWalkChunks segments; the spill area behind them grows on demand
  const size_t rec_own = (size_t)ntiles * kRecStride, seg_own = (size_t)ntiles * kWalkChunks;
  if (rec_own + (1 << 16) >= 0xFFFFFFFFull) {
    plvs::set_error("tsdf_chisel integrate: %d points in one call exceed the record index range (split the batch)", n);
    return PLVS_ERR_CAPACITY;
  }
  size_t rec_spill = std::max<size_t>(h->w_rec.cap > rec_own ? h->w_rec.cap - rec_own : 0, (size_t)1 << 16);
  size_t seg_spill = std::max<size_t>(h->w_seg.cap / 2 > seg_own ? h->w_seg.cap / 2 - seg_own : 0, (size_t)1 << 12);
  PLVS_HIP_TRY(h->w_run_cnt.reserve(ntiles));
  PLVS_HIP_TRY(h->w_part_off.reserve((size_t)max_chunks + 1));
  PLVS_HIP_TRY(h->w_multi_idx.reserve((size_t)max_chunks + 1));
  {
    int rc = ensure_part_acc(h, std::min<uint32_t>((uint32_t)max_chunks, 64u));
    if (rc != PLVS_OK) return rc;
  }
  PLVS_HIP_TRY(h->dkey0.reserve(kSmallRuns));
  PLVS_HIP_TRY(h->w_val0.reserve(kSmallRuns));
  PLVS_HIP_TRY(h->heads.reserve(kSmallRuns));
  PLVS_HIP_TRY(h->w_run_off.reserve((size_t)ntiles + 1));
  PLVS_HIP_TRY(h->scratch.reserve(scan_scratch_words(ntiles)));
  h->stage_set = 1;
  const int size_class = ntiles <= kSmallCallTiles ? 0 : (ntiles <= kPredictTiles ? 1 : 2);
  const int chunks_before = h->num_chunks;
  timespec trace_t0;   // (developer trace: the call's time on the host's clock)
  clock_gettime(CLOCK_MONOTONIC, &trace_t0);
#define STAGE_MARK(i) \
  do { if (h->profiling) PLVS_HIP_TRY(hipEventRecord(h->ev[i], s)); } while (0)
  for (int attempt = 0;; ++attempt) {
    // (the per-tile regions are sized by the call's tiles: a stream of calls of varying length would otherwise re-allocate
    // these — the largest buffers of the handle, hundreds of MB: a hipFree + hipMalloc of that size costs milliseconds —
    // every time a call is a little longer than any before; they grow to TWICE what a call needs instead)
    if (h->w_rec.cap < rec_own + rec_spill) PLVS_HIP_TRY(h->w_rec.reserve(2 * rec_own + rec_spill));
    if (h->w_seg.cap < 2 * (seg_own + seg_spill)) PLVS_HIP_TRY(h->w_seg.reserve(2 * (2 * seg_own + seg_spill)));
    PLVS_HIP_TRY(h->w_sorted_seg.reserve(h->w_seg.cap));
    // (a long call's runs may be collected chunk by chunk instead of sorted: collect_chain, below)
    static const int collect_mode = plvs::env_int("PLVS_TSDF_COLLECT", 1, 0, 2);   // (developer switch: 0 never, 2 every call — tests)
    // (the colour chain of this attempt — the reasons are where the chains are queued, below: `predicted` = on the sizes of
    // the call before, general chain; a long call over new ground: collected)
    const size_t expect_runs = h->small_runs_known
        ? (size_t)((double)h->small_runs_last * (double)ntiles / (double)std::max(1u, h->small_tiles_last)) : ~(size_t)0;
    constexpr size_t kPredictRuns = 200000;
    constexpr uint32_t kCollectMinRuns = 65536;
    static const bool predict_long = plvs::env_int("PLVS_TSDF_PREDICT_LONG", 0, 0, 1) != 0;   // (developer switch)
    const bool predicted = h->small_runs_known && attempt == 0 && collect_mode != 2 &&
                           (predict_long || ntiles <= kPredictTiles || (expect_runs <= kPredictRuns && collect_mode == 0));
    // (with the collected chain a long call is never `predicted`: that chain is queued without the call's counts just as
    // well, and the few runs of a saturated map's rim cost it seven short kernels instead of the sort's ten — steady state
    // 0.631 -> 0.618 ms)
    const bool collect_ready = collect_mode != 0 && (ntiles > kPredictTiles || collect_mode == 2) && !predicted;
    // (rows of the run matrix: twice the chunks the call before updated — more than that and the general chain takes over)
    // (a power of two: the matrix is re-allocated when a stream's calls update twice the chunks, not a few more each time)
    size_t collect_row_chunks = 256;
    while (collect_row_chunks < 2 * (size_t)h->last_updated + 64) collect_row_chunks *= 2;
    // (developer switch, tests: at most this many chunks' rows — calls that update more repeat their chain, the general one)
    static const int max_row_chunks = plvs::env_int("PLVS_TSDF_COLLECT_MAX_ROWS", 0, 0, 1 << 20);
    if (max_row_chunks > 0) collect_row_chunks = std::min<size_t>(collect_row_chunks, (size_t)max_row_chunks);
    const uint32_t collect_rows = (uint32_t)std::min<size_t>((size_t)max_chunks, collect_row_chunks) * kSlabs;
    const uint32_t collect_blocks = (uint32_t)ceil_div(seg_own, kSegSpan);
    // (the most runs its buffers hold: four times the call before scaled to this call's tiles — the stream's counts go 4.5 M,
    // 1.4 M, 1.6 M, 1.1 M, 2.6 M —, 8 M at least, never more than the tiles' run slots)
    const uint32_t collect_bound = (uint32_t)std::min<size_t>(
        (size_t)ntiles << h->run_r1_log2,
        std::max<size_t>((size_t)8 << 20, h->small_runs_known ? (size_t)(4.0 * (double)h->small_runs_last * (double)ntiles /
                                                                         (double)std::max(1u, h->small_tiles_last)) : 0));
    if (collect_ready) {
      PLVS_HIP_TRY(h->w_rseg.reserve(h->w_seg.cap));
      PLVS_HIP_TRY(h->w_rpre.reserve(seg_own));
      PLVS_HIP_TRY(h->w_active_idx.reserve((size_t)max_chunks));
      PLVS_HIP_TRY(h->w_run_matrix.reserve((size_t)collect_rows * collect_blocks));
      PLVS_HIP_TRY(h->w_item_base.reserve(collect_rows));
      PLVS_HIP_TRY(h->w_item_cnt.reserve(collect_rows));
      PLVS_HIP_TRY(h->w_item_part0.reserve(collect_rows));
      PLVS_HIP_TRY(h->w_row_heads.reserve(collect_rows));
      PLVS_HIP_TRY(h->w_row_tot.reserve((size_t)collect_rows * kSlabVox));
      const size_t parts_cap = (size_t)collect_bound / kCollectPart + collect_rows + 1;
      PLVS_HIP_TRY(h->dkey0.reserve(collect_bound));
      PLVS_HIP_TRY(h->dkey1.reserve(collect_bound));
      PLVS_HIP_TRY(h->w_val0.reserve(collect_bound));
      PLVS_HIP_TRY(h->w_val1.reserve(collect_bound));
      PLVS_HIP_TRY(h->heads.reserve(collect_bound));
      PLVS_HIP_TRY(h->w_part_item.reserve(parts_cap));
      PLVS_HIP_TRY(h->w_phist.reserve(parts_cap * kSlabVox));
      // (zero while the walk runs: runs_count writes the cells that hold runs)
      PLVS_HIP_TRY(hipMemsetAsync(h->w_run_matrix.p, 0, (size_t)collect_rows * collect_blocks * sizeof(uint32_t), h->side));
      PLVS_HIP_TRY(hipEventRecord(h->ev_zero, h->side));   // (runs_count, on the caller's stream, waits for it: long over by then)
    }
    if (h->w_runkey.cap < ((size_t)ntiles << h->run_r1_log2)) {
      PLVS_HIP_TRY(h->w_runkey.reserve((size_t)2 * ntiles << h->run_r1_log2));
      PLVS_HIP_TRY(h->w_masks.reserve(((size_t)2 * ntiles << h->run_r1_log2) * kMaskWords));
    }
    PLVS_HIP_TRY(h->w_masks.reserve(((size_t)ntiles << h->run_r1_log2) * kMaskWords));
    hipLaunchKernelGGL(walk_prologue, dim3(ceil_div((size_t)std::max(max_chunks, nclouds + 1), 256)), dim3(256), 0, s, d_Twc,
                       nclouds, h->poses.p, (const int32_t*)h->h_offsets, h->offsets.p, h->d_wctr, h->d_ctr, h->w_chunk_nseg.p,
                       max_chunks, gsrc ? (int)kGridWords : (int)(2 * ntiles));
    STAGE_MARK(0);
    const GridSrc* const d_grid = gsrc ? reinterpret_cast<const GridSrc*>(h->offsets.p + 2 * ((size_t)nclouds + 1)) : nullptr;
    static const bool count_in_walk = plvs::env_int("PLVS_SEG_COUNT_IN_WALK", 1, 0, 1) != 0;   // (developer switch)
    AccOut out{h->w_rec.p, (uint32_t)std::min<size_t>(rec_own + rec_spill, 0xFFFFFFFFu), h->w_seg.p,
               (uint32_t)std::min<size_t>(seg_own + seg_spill, 0xFFFFFFFFu), h->w_seg_cnt.p, h->w_tile_visits.p,
               count_in_walk ? h->w_chunk_nseg.p : nullptr};
    RunOut runs{h->w_runkey.p, h->w_masks.p, h->w_run_cnt.p, h->run_r1_log2, collect_ready ? h->w_rseg.p : nullptr};
    // the common case of a tile alone in a lean kernel; what it defers (tiles over several clouds, table overflows,
    // the owner-filtered walk of a sharded handle) is walked by the general kernel from the list
    uint32_t* const list_a = h->w_deferred.p;
    uint32_t* const list_b = h->w_deferred.p + ntiles;
    uint32_t* const list_c = h->w_deferred.p + 2 * (size_t)ntiles;
    const uint32_t* last_list = list_a;
    const uint32_t* last_count = &h->d_wctr->ndeferred;
    bool second_small = false, third_pass = false;
#define PLVS_LAUNCH_WALK_FAST(E, GRID, TILES, LIST, NLIST, DEFERRED, NDEFERRED)                                              \
  hipLaunchKernelGGL((walk_fast<E, GRID>), dim3(TILES), dim3(kWalkRays), 0, s, h->P, h->scale_u, h->scale_w, d_xyz, n,          \
                     h->offsets.p, nclouds, h->poses.p, h->dir, &h->d_ctr->num_chunks, h->d_wctr, h->rgbw,                     \
                     (const uint32_t*)nullptr, out, runs, TileMap{1u, 0u, 1u}, kRecStride, (const uint32_t*)(LIST),            \
                     (const uint32_t*)(NLIST), DEFERRED, NDEFERRED, d_grid)
#define PLVS_WALK_FAST(E, TILES, LIST, NLIST, DEFERRED, NDEFERRED)                                  \
  do {                                                                                              \
    if (gsrc) PLVS_LAUNCH_WALK_FAST(E, true, TILES, LIST, NLIST, DEFERRED, NDEFERRED);              \
    else PLVS_LAUNCH_WALK_FAST(E, false, TILES, LIST, NLIST, DEFERRED, NDEFERRED);                  \
  } while (0)
    if (ntiles <= kSmallCallTiles) {
      PLVS_WALK_FAST(kFastEntriesBig, ntiles, nullptr, nullptr, list_a, &h->d_wctr->ndeferred);
    } else {
      // The table of the first pass follows the scene: tiles of near surfaces (a small room, a desk) hold 300-600 voxels
      // and a 1024-entry table lets THREE of them share a CU (6 waves per SIMD: 0.49 against 0.65 ms for the 100 key
      // frames of the saturated room); tiles of walls 3-5 m away hold 800-2000 and would nearly all overflow it.  The
      // counters of the call before decide (walk_small): the 2048-entry kernel counts the tiles a 1024-entry table would
      // not have held, the 1024-entry kernel's deferred list says when it stops paying.
      {   // developer switch: PLVS_WALK_SMALL = 0 / 1 forces the first pass's table
        static const int force = plvs::env_int("PLVS_WALK_SMALL", -1, -1, 1);
        if (force >= 0) h->walk_small = force != 0;
      }
      h->walk_small_used = h->walk_small;
      if (h->walk_small) PLVS_WALK_FAST(kFastEntriesSmall, ntiles, nullptr, nullptr, list_a, &h->d_wctr->ndeferred);
      else PLVS_WALK_FAST(kFastEntries, ntiles, nullptr, nullptr, list_a, &h->d_wctr->ndeferred);
      // the tiles that overflowed the first pass: a 1024-entry first pass hands them to the 2048-entry kernel (two tiles per CU
      // instead of one: nearly all of them fit it), a 2048-entry first pass to the 4096-entry one
      if (h->walk_small) {
        PLVS_WALK_FAST(kFastEntries, std::min<unsigned>(ntiles, 2 * kListGrid), list_a, &h->d_wctr->ndeferred, list_b,
                       &h->d_wctr->ndeferred2);
        // ... and what overflows that one too (a wall 5 m away seen at a slant) to the 4096-entry kernel rather than to
        // walk_tiles — a call none of whose tiles reaches walk_tiles can have its runs collected chunk by chunk (below) instead
        // of sorted — when the call before had such tiles: a launch that finds an empty list costs the stream 8 us
        third_pass = h->third_pass;
        if (third_pass)
          PLVS_WALK_FAST(kFastEntriesBig, std::min<unsigned>(ntiles, kListGrid), list_b, &h->d_wctr->ndeferred2, list_c,
                         &h->d_wctr->ndeferred3);
      } else
        PLVS_WALK_FAST(kFastEntriesBig, std::min<unsigned>(ntiles, kListGrid), list_a, &h->d_wctr->ndeferred, list_b,
                       &h->d_wctr->ndeferred2);
      second_small = h->walk_small && !third_pass;   // (the last lean pass had 2048 entries)
#undef PLVS_WALK_FAST
#undef PLVS_LAUNCH_WALK_FAST
      last_list = third_pass ? list_c : list_b;
      last_count = third_pass ? &h->d_wctr->ndeferred3 : &h->d_wctr->ndeferred2;
    }
    hipLaunchKernelGGL((walk_tiles<true, true>), dim3(kDeferGrid), dim3(kWalkRays), 0, s, h->P, h->scale_u, h->scale_w, d_xyz, n,
                       h->offsets.p, nclouds, h->poses.p, h->dir, &h->d_ctr->num_chunks, h->d_wctr, h->rgbw,
                       (const uint32_t*)nullptr, out, runs, TileMap{1u, 0u, 1u}, (uint32_t)ntiles, last_list, last_count,
                       kRecStride, second_small ? 1u : 2u,
                       d_grid);   // (what is flagged overflowed a 4096-entry table: two pieces at once; a 2048-entry one: it goes whole)
    // A small call (a few key frames: PointCloudMapping::UpdateMap's batches) launches its colour chain on the sizes of
    // the small call before it instead of waiting for its own (below), and keeps the chain — a dozen dependent launches,
    // the longer of the two branches — on the caller's stream: the segment sort and the apply stage go to the side stream
    // and are long over when the chain ends.  (A branch on another stream starts ~20 us after the event it waits for and is
    // joined ~20 us after it ends.)
    // (round 5) ... and so does a LONG call once the map has saturated: the call before left at most a few hundred runs (what
    // few rays reach at the rim of the map), the chain is then three short launches on a bound of kSmallRuns, and the host's
    // read of the run count — scan, publish, a wake-up: 0.06 ms behind a 0.09 ms segment sort + apply — was what a steady-state
    // step ended with.  A long call over new ground (millions of runs, twice or half the call before) keeps its own count.
    // (expect_runs, predicted: computed at the top of the attempt — what the call reserves depends on them)
    // (a moderate number — a saturated map's rim: tens of thousands — is sorted on a bound a quarter above the expectation)
    // (end of round 5, measured and left OFF: PLVS_TSDF_PREDICT_LONG=1) every call whose predecessor left a count could do so:
    // beyond a moderate number with a LOOSE bound — three times the expectation, a million at least: the stream's counts go
    // 4.5 M, 1.4 M, 1.6 M, 1.1 M, 2.6 M ... — and a sort that takes the number of pairs from the device
    // (radix_sort_pairs_bound: launches sized by the bound, surplus tiles leave at once), so that a loose bound costs empty
    // workgroups, not sorted padding, and the chain is queued behind the walk without a host read.  On the stream: GPU time
    // per step 0.969 -> 0.965 ms, wall time 1.05 -> 1.07 (a bound that fails costs the chain twice): the chain's length is its
    // kernels' (fold 0.13, three passes 0.15, compaction, heads), not the host's read.
    // (a long call over new ground, not the handle's first: its runs are collected chunk by chunk, below; it has nothing for
    // the side stream before its counting stages are over — ev_seg — and goes without the event behind the walk: an event
    // between two kernels of a stream costs ~8 us)
    const bool collect_fast = collect_ready && h->small_runs_known && attempt == 0 && !predicted;
    if (!collect_fast) PLVS_HIP_TRY(hipEventRecord(h->ev_fork, s));
    STAGE_MARK(1);
    // When that bound is the small one (<= kSmallRuns: ONE sorting launch), nothing is worth a second stream: a branch on
    // another stream starts ~20 us after the event it waits for and is joined ~20 us after it ends — more than the chain
    // itself.  Segment sort, apply, sort_runs_small, fold follow each other on the caller's stream; seg_scan, which sums the
    // tiles' run counts anyway, leaves the total where the colour side reads it (no scan of the counts either).
    const bool serial_small = predicted && expect_runs <= kSmallRuns / 2;
    const hipStream_t q_apply = (predicted && !serial_small) ? h->side : s, q_colour = predicted ? s : h->side;
#define STAGE_MARK_ON(i, q) \
  do { if (h->profiling) PLVS_HIP_TRY(hipEventRecord(h->ev[i], q)); } while (0)
    uint32_t collect_seq = 0;   // (the sequence number rows_place publishes the walk's counters under)
    auto segments_and_apply = [&]() -> int {
    const unsigned seg_blocks = ceil_div(seg_own + seg_spill, kSegSpan);
    if (!count_in_walk)
      hipLaunchKernelGGL(seg_pass<false>, dim3(seg_blocks), dim3(256), 0, q_apply, h->w_seg.p, out.seg_cap, ntiles,
                         h->w_seg_cnt.p, h->w_chunk_nseg.p, h->w_chunk_off.p, h->w_chunk_fill.p, h->w_sorted_seg.p,
                         h->d_wctr);
    hipLaunchKernelGGL(seg_scan, dim3(1), dim3(1024), 0, q_apply, h->w_chunk_nseg.p, h->w_chunk_off.p, h->w_chunk_fill.p,
                       h->updated.p, h->w_active_off.p, h->d_wctr, &h->d_ctr->num_chunks, max_chunks,
                       h->w_tile_visits.p, h->w_run_cnt.p, ntiles, h->w_part_off.p, h->w_multi_idx.p, h->multi_cap,
                       h->part_segs, h->part_min, collect_ready ? h->w_active_idx.p : (uint32_t*)nullptr);
    if (collect_ready) {
      // the colour side's counting stages here, in front of the segment sort: short kernels that would otherwise start
      // beside the apply stage's first thousand workgroups and wait for their slots (40 us each, measured)
      PLVS_HIP_TRY(hipStreamWaitEvent(q_apply, h->ev_zero, 0));
      hipLaunchKernelGGL(runs_count, dim3(collect_blocks), dim3(kSegSpan), 0, q_apply, h->w_seg.p, h->w_rseg.p, ntiles, h->w_seg_cnt.p,
                         h->w_active_idx.p, collect_rows, collect_blocks, h->w_run_matrix.p, h->w_rpre.p, h->d_wctr, last_count);
      hipLaunchKernelGGL(runs_rowscan, dim3(std::min<uint32_t>(ceil_div(collect_rows, 4), 1024u)), dim3(256), 0, q_apply,
                         h->w_run_matrix.p, collect_rows, collect_blocks, h->d_wctr, h->w_item_cnt.p);
      hipLaunchKernelGGL(rows_place, dim3(1), dim3(1024), 0, q_apply, h->w_item_cnt.p, collect_rows, collect_bound,
                         (uint32_t)std::min<size_t>((size_t)collect_bound / kCollectPart + collect_rows + 1, 0xFFFFFFFFu), h->d_wctr,
                         h->w_item_base.p, h->w_item_part0.p, h->w_part_item.p, reinterpret_cast<const uint32_t*>(h->d_ctr),
                         reinterpret_cast<uint32_t*>(h->h_wctr), reinterpret_cast<uint32_t*>(h->h_ctr),
                         (uint32_t)(sizeof(Counters) / sizeof(uint32_t)), collect_fast ? h->h_seq : (uint32_t*)nullptr,
                         collect_fast ? (collect_seq = ++h->seq_next) : 0u);
      PLVS_HIP_TRY(hipEventRecord(h->ev_seg, q_apply));
    }
    // (a long call: 4096 descriptor slots — 64 tiles — per workgroup instead of 1024: a quarter of the workgroups, and what
    // the kernel waits for is their atomics on the counters of a hundred-odd chunks)
    static const bool wide_span = plvs::env_int("PLVS_SEG_SPAN_WIDE", 1, 0, 1) != 0;   // (developer switch)
    if (wide_span && ntiles > kPredictTiles)
      hipLaunchKernelGGL((seg_pass<true, kSegSpanLong>), dim3(ceil_div(seg_own + seg_spill, kSegSpanLong)), dim3(256), 0, q_apply, h->w_seg.p,
                         out.seg_cap, ntiles, h->w_seg_cnt.p, h->w_chunk_nseg.p, h->w_chunk_off.p, h->w_chunk_fill.p,
                         h->w_sorted_seg.p, h->d_wctr);
    else
    hipLaunchKernelGGL(seg_pass<true>, dim3(seg_blocks), dim3(256), 0, q_apply, h->w_seg.p, out.seg_cap, ntiles,
                       h->w_seg_cnt.p, h->w_chunk_nseg.p, h->w_chunk_off.p, h->w_chunk_fill.p, h->w_sorted_seg.p,
                       h->d_wctr);
    STAGE_MARK_ON(2, q_apply);
    hipLaunchKernelGGL((apply_chunks<false, false>), dim3(4096), dim3(kApplyThreads), 0, q_apply, h->w_sorted_seg.p, h->updated.p,
                       h->w_active_off.p, h->w_part_off.p, h->w_multi_idx.p, h->part_segs,
                       PartAcc{h->pa_wuu.p, h->pa_w.p, h->pa_last.p, h->pa_cnt.p, h->pa_done.p}, h->w_rec.p,
                       1.0 / (double)h->scale_u, 1.0 / (double)h->scale_w, d_kfid, h->sdf, h->weight, h->kfid, h->d_wctr,
                       EmitOut{}, gsrc ? grid.key_bits : 0u);
    PLVS_KERNEL_CHECK();
    STAGE_MARK_ON(3, q_apply);
    return PLVS_OK;
    };
    // (the host issues the critical branch first: a one-key-frame walk is over before a dozen launches have been made)
    if (!predicted) {
      int rc = segments_and_apply();
      if (rc != PLVS_OK) return rc;
    }
    // ---- colour fold: the truncating u8 mean is order dependent -> through the sorted runs of the voxels
    // whose colour weight is below 254.  It only needs the runs the walk left, so it runs on another stream than
    // the segment sort and the apply stage (both short, latency-bound kernels).
    uint32_t* const side_ctr = &h->d_wctr[1].num_desc;   // the run count, for the side stream's kernels
    // the chain behind the scan of the run counts, for D runs (or a bound on them: guard) on stream q
    // (developer switch: the most runs the one-launch sort takes, 0 = never; one workgroup sorts ~1 000 pairs per microsecond
    // and pass, the general chain costs ~60 us of launches before it does anything)
    static const uint32_t medium_max = (uint32_t)plvs::env_int("PLVS_TSDF_MEDIUM_SORT", 16384, 0, (int)kMediumRuns);
    // (... and of few tiles: its listing of the runs is one workgroup's loop over the tiles — 15 000 tiles with a run each, the rim
    // of a saturated map in a 100-key-frame call, take it longer than the general chain's launches: steady state 0.62 -> 0.67 ms)
    const bool medium_sort = medium_max != 0 && ntiles <= 2048u;
    auto launch_fold = [&](uint32_t D, const uint32_t* skeys, const uint32_t* sval, hipStream_t q, const uint32_t* skip) -> int {
      const RunSrc rsrc{h->w_masks.p, (uint32_t)kMaskWords, h->run_r1_log2, TileMap{1u, 0u}, h->offsets.p, nclouds,
                        reinterpret_cast<const uint32_t*>(h->offsets.p) + 2 * ((size_t)nclouds + 1)};
      if (gsrc)
        hipLaunchKernelGGL(fold_colours_masks<true>, dim3(std::min<size_t>(ceil_div(D, kFoldWaves), 8192)),
                           dim3(64 * kFoldWaves), 0, q, skeys, sval, side_ctr, rsrc, h->heads.p, d_rgb,
                           h->rgbw, &h->d_wctr[1].num_heads, (uint32_t*)nullptr, (uint32_t*)nullptr, skip, grid);
      else
        hipLaunchKernelGGL(fold_colours_masks<false>, dim3(std::min<size_t>(ceil_div(D, kFoldWaves), 8192)),
                           dim3(64 * kFoldWaves), 0, q, skeys, sval, side_ctr, rsrc, h->heads.p, d_rgb,
                           h->rgbw, &h->d_wctr[1].num_heads, (uint32_t*)nullptr, (uint32_t*)nullptr, skip, grid);
      PLVS_KERNEL_CHECK();
      return PLVS_OK;
    };
    auto colour_chain = [&](uint32_t D, int chunks, hipStream_t q, const RunGuard* guard) -> int {
      const uint32_t* skeys = h->dkey0.p;
      const uint32_t* sval = h->w_val0.p;
      if (D <= kSmallRuns) {
        hipLaunchKernelGGL(sort_runs_small, dim3(1), dim3(1024), 0, q, h->w_runkey.p, h->w_run_cnt.p, ntiles,
                           h->run_r1_log2, h->d_wctr + 1, h->dkey0.p, h->w_val0.p, h->heads.p, guard ? guard->skip : (uint32_t*)nullptr);
      } else if (medium_sort && D <= medium_max) {
        // one workgroup, one launch: the runs listed, sorted and their voxels' first runs found (sort_runs_medium)
        PLVS_HIP_TRY(h->dkey0.reserve(D));
        PLVS_HIP_TRY(h->dkey1.reserve(D));
        PLVS_HIP_TRY(h->w_val0.reserve(D));
        PLVS_HIP_TRY(h->w_val1.reserve(D));
        PLVS_HIP_TRY(h->heads.reserve(D));
        int key_bits = 12;
        while ((1ll << (key_bits - 12)) < (long long)chunks) ++key_bits;
        const bool ten = key_bits <= 20;   // (two passes of ten bits instead of three of eight)
        const int passes = ten ? 2 : (key_bits + 7) / 8;
#define PLVS_SORT_MEDIUM(BITS)                                                                                                    \
  hipLaunchKernelGGL(sort_runs_medium<BITS>, dim3(1), dim3(1024), 0, q, h->w_runkey.p, h->w_run_cnt.p, ntiles, h->run_r1_log2,   \
                     h->d_wctr + 1, h->dkey0.p, h->w_val0.p, h->dkey1.p, h->w_val1.p, h->w_run_off.p, h->heads.p, passes,         \
                     guard ? guard->limit : 0xFFFFFFFFu, &h->d_ctr->num_chunks, guard ? guard->chunk_limit : 0,                   \
                     guard ? guard->skip : (uint32_t*)nullptr)
        if (ten) PLVS_SORT_MEDIUM(10);
        else PLVS_SORT_MEDIUM(8);
#undef PLVS_SORT_MEDIUM
        skeys = (passes & 1) ? h->dkey1.p : h->dkey0.p;
        sval = (passes & 1) ? h->w_val1.p : h->w_val0.p;
      } else {
        int rc = sort_runs(h, D, ntiles, chunks, q, &skeys, &sval, guard);
        if (rc != PLVS_OK) return rc;
        PLVS_HIP_TRY(h->heads.reserve(D));
        PLVS_HIP_TRY(h->w_dummy.reserve((size_t)max_chunks + 1));
        hipLaunchKernelGGL(voxel_heads, dim3(ceil_div(D, 256 * kHeadTiles)), dim3(256), 0, q, skeys, D,
                           h->heads.p, h->w_dummy.p, h->d_wctr + 1, guard ? (const uint32_t*)side_ctr : (const uint32_t*)nullptr);
      }
      return launch_fold(D, skeys, sval, q, guard ? (const uint32_t*)guard->skip : (const uint32_t*)nullptr);
    };
    // the runs of a long call whose tiles all went through walk_fast, chunk by chunk (runs_count ... parts_place): behind the
    // list of the updated chunks (ev_seg), no pass that sorts all runs.  What it cannot take sets `skip` — the fold then
    // leaves at once and the general chain runs once the call's counters are read (as for a predicted chain whose bounds
    // did not hold).
    auto collect_chain = [&](uint32_t D, hipStream_t q) -> int {
      const size_t parts_cap = (size_t)collect_bound / kCollectPart + collect_rows + 1;
      PLVS_HIP_TRY(hipStreamWaitEvent(q, h->ev_seg, 0));
      hipLaunchKernelGGL(runs_scatter, dim3(collect_blocks * (kSegSpan / 256)), dim3(256), 0, q, h->w_seg.p, h->w_rseg.p, ntiles, h->w_seg_cnt.p,
                         h->w_active_idx.p, collect_rows, collect_blocks, h->w_run_matrix.p, h->w_rpre.p, h->w_item_base.p,
                         h->d_wctr, h->w_val0.p);
      const unsigned part_grid = (unsigned)std::min<size_t>(parts_cap, 4096);
      hipLaunchKernelGGL(parts_count, dim3(part_grid), dim3(256), 0, q, h->w_part_item.p, h->w_item_part0.p, h->w_item_base.p,
                         h->w_item_cnt.p, h->w_runkey.p, h->d_wctr, h->w_val0.p, h->dkey0.p, h->w_phist.p);
      hipLaunchKernelGGL(rows_heads, dim3(std::min<uint32_t>(ceil_div(collect_rows, 4), 1024u)), dim3(256), 0, q, h->w_item_part0.p,
                         h->w_item_cnt.p, h->w_phist.p, h->d_wctr, collect_rows, h->w_row_heads.p, h->w_row_tot.p);
      hipLaunchKernelGGL(parts_place, dim3(part_grid), dim3(512), 0, q, h->w_part_item.p, h->w_item_part0.p, h->w_item_base.p,
                         h->w_item_cnt.p, h->w_phist.p, h->d_wctr, h->dkey0.p, h->w_val0.p, h->dkey1.p, h->w_val1.p, h->heads.p,
                         h->w_row_heads.p, collect_rows, h->w_row_tot.p);
      return launch_fold(D, h->dkey1.p, h->w_val1.p, q, &h->d_wctr[1].skip);
    };
    // (predicted: a bound that does not hold costs the chain a second time — the fold of the first skips itself: compact_runs)
    uint32_t run_bound = 0;
    int chunk_bound = 0;
    bool collected = false, scanned = false;
    if (serial_small) {
      int rc = segments_and_apply();
      if (rc != PLVS_OK) return rc;
      run_bound = kSmallRuns;
      chunk_bound = std::min(max_chunks, std::max(2 * chunks_before, chunks_before + 256));
      const RunGuard guard{run_bound, side_ctr, &h->d_ctr->num_chunks, chunk_bound, &h->d_wctr[0].err, &h->d_wctr[1].skip, 1u, nullptr, 0u};
      rc = colour_chain(run_bound, chunk_bound, s, &guard);
      if (rc != PLVS_OK) return rc;
    } else {
      if (!predicted && !collect_fast) PLVS_HIP_TRY(hipStreamWaitEvent(h->side, h->ev_fork, 0));
      if (predicted) {
        const size_t slots = (size_t)ntiles << h->run_r1_log2;
        // (the call before scaled to this call's tiles — calls of one and of five key frames may alternate —, a quarter more)
        const size_t expect = (size_t)((double)h->small_runs_last * (double)ntiles / (double)std::max(1u, h->small_tiles_last));
        // (up to kPredictRuns: a tight bound, padded — two or three short launches of the plain sort; beyond: the loose one)
        const bool loose = predict_long && expect > kPredictRuns;
        run_bound = expect <= kSmallRuns / 2 ? kSmallRuns
                    : loose ? (uint32_t)std::min<size_t>(slots, (std::max<size_t>(3 * expect, (size_t)1 << 20) + 4095) / 4096 * 4096)
                            : (uint32_t)std::min<size_t>(slots, (expect * 5 / 4 + 8191) / 4096 * 4096);
        // (a moderate expectation: the one-launch sort on its full capacity — a bound that costs nothing; it sums the tiles'
        // run counts itself, the general chain wants them scanned)
        if (medium_sort && run_bound > kSmallRuns && expect * 5 / 4 + 1024 <= medium_max) run_bound = medium_max;
        if (run_bound > medium_max) {
          scanned = true;
          PLVS_HIP_TRY(exclusive_scan_u32(h->w_run_cnt.p, h->w_run_off.p, ntiles, side_ctr, h->scratch.p, q_colour));
        }
        chunk_bound = std::min(max_chunks, std::max(2 * chunks_before, chunks_before + 256));
        const RunGuard guard{run_bound, side_ctr, &h->d_ctr->num_chunks, chunk_bound, &h->d_wctr[0].err, &h->d_wctr[1].skip,
                             loose ? 0u : 1u, nullptr, 0u};
        int rc = colour_chain(run_bound, chunk_bound, q_colour, &guard);
        if (rc != PLVS_OK) return rc;
        PLVS_HIP_TRY(hipStreamWaitEvent(h->side, h->ev_fork, 0));
        rc = segments_and_apply();
        if (rc != PLVS_OK) return rc;
      } else if (collect_fast) {
        // a long call over new ground, not the handle's first: the runs chunk by chunk, queued behind the walk without a
        // read of its counters — the buffers hold four times the call before, the kernels decide themselves whether the
        // call is theirs (runs_count, rows_place: `skip`)
        // The walk's counters are published all the same — in front of the chain on the side stream, which waits for the
        // counting stages anyway — and read while the chain is queued: a tile or two of one call in twenty reach walk_tiles (a
        // wall seen at a slant: more chunks than a tile's cache holds), the chain's kernels then leave at once (runs_count:
        // `skip`) and the general chain is queued behind them now, not after the call's last kernel.
        const uint32_t seq = collect_seq;   // (published by rows_place)
        collected = true;
        int rc = collect_chain(collect_bound, h->side);
        if (rc != PLVS_OK) return rc;
        {
          int rcw = wait_published(h, seq, h->side, 1, size_class);
          if (rcw != PLVS_OK) return rcw;
        }
        const uint32_t left_to_walk_tiles =
            reinterpret_cast<const uint32_t*>(h->h_wctr)[last_count - reinterpret_cast<const uint32_t*>(h->d_wctr)];
        if (h->h_wctr[0].err == 0 && (left_to_walk_tiles != 0u || h->h_wctr[0].seg_top != 0u)) {
          collected = false;
          scanned = true;
          PLVS_HIP_TRY(exclusive_scan_u32(h->w_run_cnt.p, h->w_run_off.p, ntiles, side_ctr, h->scratch.p, h->side));
          // (the run count: what seg_scan leaves is not there yet — the tiles' counts are: summed here)
          const uint32_t seq2 = ++h->seq_next;
          hipLaunchKernelGGL(publish_counters, dim3(1), dim3(64), 0, h->side, h->d_wctr, h->d_ctr, h->h_wctr, h->h_ctr, h->h_seq, seq2);
          int rcw = wait_published(h, seq2, h->side, 2, size_class);   // (kind 2: a short wait of its own expectation)
          if (rcw != PLVS_OK) return rcw;
          const uint32_t D = h->h_wctr[1].num_desc;
          if (D > 0) {
            rc = colour_chain(D, h->h_ctr->num_chunks, h->side, nullptr);
            if (rc != PLVS_OK) return rc;
          }
        }
      } else {
        scanned = true;
        PLVS_HIP_TRY(exclusive_scan_u32(h->w_run_cnt.p, h->w_run_off.p, ntiles, side_ctr, h->scratch.p, q_colour));
        const uint32_t seq = ++h->seq_next;
        hipLaunchKernelGGL(publish_counters, dim3(1), dim3(64), 0, h->side, h->d_wctr, h->d_ctr, h->h_wctr, h->h_ctr, h->h_seq, seq);
        {   // the walk is over; segment sort and apply are queued behind it
          int rcw = wait_published(h, seq, h->side, 1, size_class);
          if (rcw != PLVS_OK) return rcw;
        }
        const uint32_t D = h->h_wctr[1].num_desc;
        // (collected: no tile was left to walk_tiles — its runs are not grouped by chunk — and enough runs to pay two launches)
        const uint32_t left_to_walk_tiles =
            reinterpret_cast<const uint32_t*>(h->h_wctr)[last_count - reinterpret_cast<const uint32_t*>(h->d_wctr)];
        collected = collect_ready && left_to_walk_tiles == 0u && h->h_wctr[0].seg_top == 0u &&
                    (D > kCollectMinRuns || collect_mode == 2);
        if (h->h_wctr[0].err == 0 && D > 0) {
          int rc = collected ? collect_chain(D, h->side) : colour_chain(D, h->h_ctr->num_chunks, h->side, nullptr);
          if (rc != PLVS_OK) return rc;
        }
      }
      PLVS_HIP_TRY(hipEventRecord(h->ev_join, h->side));
      PLVS_HIP_TRY(hipStreamWaitEvent(s, h->ev_join, 0));
    }
#undef STAGE_MARK_ON
    STAGE_MARK(4);
    int rc = read_walk_counters(h, s, size_class);
    if (rc != PLVS_OK) return rc;
    const uint32_t err = h->h_wctr->err;
    if (err & ~kErrScratch) return walk_fail(h, err);
    if (err & kErrScratch) {   // the map is untouched (apply_chunks left at once, no colours folded): grow and repeat
      if (attempt >= 8) return walk_fail(h, err);
      rec_spill = std::max<size_t>(rec_spill, (size_t)h->h_wctr->rec_top * 2);
      seg_spill = std::max<size_t>(seg_spill, (size_t)h->h_wctr->seg_top * 2);
      while ((1u << h->run_r1_log2) < h->h_wctr->run_need) ++h->run_r1_log2;
      if (((size_t)ntiles << h->run_r1_log2) >= 0xFFFFFFFFull) return walk_fail(h, err);
      continue;
    }
    h->last_chain = collected ? 2 : (predicted ? 1 : 0);
    h->last_chain_skipped = (predicted || collected) && h->h_wctr[1].skip != 0u;
    if ((predicted || collected) && h->h_wctr[1].skip != 0u) {   // the bounds did not hold: the chain once more, with the call's numbers
      const uint32_t D = h->h_wctr[1].num_desc;
      if (!scanned)   // (its chain had no use for the offsets of the tiles' runs: the compaction of the long form has)
        PLVS_HIP_TRY(exclusive_scan_u32(h->w_run_cnt.p, h->w_run_off.p, ntiles, side_ctr, h->scratch.p, s));
      PLVS_HIP_TRY(hipMemsetAsync(&h->d_wctr[1].num_heads, 0, sizeof(uint32_t), s));
      PLVS_HIP_TRY(hipMemsetAsync(&h->d_wctr[1].num_updated, 0, sizeof(uint32_t), s));
      int rc2 = colour_chain(D, h->h_ctr->num_chunks, s, nullptr);
      if (rc2 != PLVS_OK) return rc2;
      rc2 = read_walk_counters(h, s, size_class);
      if (rc2 != PLVS_OK) return rc2;
    }
    h->small_runs_known = true;      // (the runs of the last call, whatever its length)
    h->small_runs_last = h->h_wctr[1].num_desc;
    h->small_tiles_last = ntiles;
    break;
  }
  const WalkCounters& c = *h->h_wctr;
  h->num_chunks = h->h_ctr->num_chunks;
  if (ntiles > kSmallCallTiles) {   // the first pass's table for the next call of this size class
    // (round 5: what overflows the small table goes to the 2048-entry kernel at two tiles per CU — a fifth of the tiles there
    // still costs less than a 2048-entry first pass for all of them: 0.51 + 0.1 against 0.8 ms on the office stream, where
    // the 6 % / 2 % thresholds of round 4 had every other step fall back to the large table)
    if (h->walk_small_used) {
      h->third_pass = c.ndeferred2 != 0u;   // (tiles overflowed the 2048-entry table too: the next call has a 4096-entry pass)
      if ((size_t)c.ndeferred * 4 > ntiles) h->walk_small = false;    // more than a quarter of the tiles overflowed the small table
    } else if ((size_t)c.over_small * 6 <= ntiles) {
      h->walk_small = true;                                             // at most a sixth would
    }
  }
  {   // developer trace of the call's counters (PLVS_HIP_TSDF_TRACE=1)
    static const bool trace = plvs::env_int("PLVS_HIP_TSDF_TRACE", 0, 0, 1) != 0;
    if (trace) {
      timespec t1;
      clock_gettime(CLOCK_MONOTONIC, &t1);
      fprintf(stderr, "[tsdf_chisel] %.0f us ", (double)(t1.tv_sec - trace_t0.tv_sec) * 1e6 + (double)(t1.tv_nsec - trace_t0.tv_nsec) * 1e-3);
    }
    if (trace)
      fprintf(stderr, "[tsdf_chisel] tiles %u deferred %u split %u visits %llu runs %u updated %u parts %u multi %u "
              "rec_top %u seg_top %u voxels %u max_run %u chunks %d chain %d%s\n", ntiles, c.ndeferred * 1000u + c.ndeferred2 + c.ndeferred3 * 1000000u, c.split_tiles,
              (unsigned long long)c.total_visits, h->h_wctr[1].num_desc, c.num_updated, c.num_parts, c.num_multi, c.rec_top,
              c.seg_top, c.num_heads, c.max_run, h->num_chunks, h->last_chain, h->last_chain_skipped ? " REPEATED" : "");
  }
  h->stats.visits = (int64_t)c.total_visits;
  h->stats.new_chunks = h->num_chunks - chunks_before;
  h->stats.updated_chunks = (int32_t)c.num_updated;
  h->stats.voxels = (int32_t)c.num_heads;
  h->stats.max_run = (int32_t)c.max_run;
  h->last_updated = c.num_updated;
  // Part accumulators for the next call: a chunk beyond them is applied in ONE part — never wrong, but on a stream of new
  // views the busy chunks of a call are not those of the call before, and a single 1 500-segment item then is the
  // whole stage (0.4 ms).  Room for twice the chunks this call updated (98 KB each), grown geometrically.
  {
    const uint32_t want = std::min<uint32_t>((uint32_t)max_chunks, std::max(c.num_multi + c.num_multi / 2, 2u * c.num_updated));
    if (want > h->multi_cap) {
      int rc = ensure_part_acc(h, std::min<uint32_t>((uint32_t)max_chunks, std::max(want, 2u * h->multi_cap)));
      if (rc != PLVS_OK) return rc;
    }
  }
  float ms[4] = {0.f, 0.f, 0.f, 0.f};   // the last one: what the colour fold adds behind the apply stage
  if (h->profiling)
    for (int i = 0; i < 4; ++i) PLVS_HIP_TRY(stage_elapsed(&ms[i], h->ev[i], h->ev[i + 1]));
#undef STAGE_MARK
  if (h->profiling) {
    for (int i = 0; i < 4; ++i) h->stage_ms[i] += ms[i];
    h->prof_calls++;
  }
  return PLVS_OK;
}

extern "C" {

int plvs_hip_selftest_rcp(int exponent, uint32_t* mismatches) {
  PLVS_REQUIRE(mismatches && exponent > -126 && exponent < 127, "bad argument");
  uint32_t* d = nullptr;
  PLVS_HIP_TRY(hipMalloc((void**)&d, sizeof(uint32_t)));
  PLVS_HIP_TRY(hipMemset(d, 0, sizeof(uint32_t)));
  hipLaunchKernelGGL(selftest_rcp_kernel, dim3((1u << 23) / 256), dim3(256), 0, nullptr, exponent, d);
  hipError_t e = hipMemcpy(mismatches, d, sizeof(uint32_t), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  PLVS_HIP_TRY(e);
  return PLVS_OK;
}

int plvs_hip_selftest_radix_sort(uint32_t n, int bit_lo, int bit_hi, int wide_values, uint32_t seed, uint32_t* mismatches2) {
  PLVS_REQUIRE(mismatches2 && n > 0 && bit_lo >= 0 && bit_hi > bit_lo && bit_hi <= 32, "bad argument");
  plvs::DevBuf<uint32_t> k_in, k0, k1, v0, v1, scratch, bad;
  plvs::DevBuf<unsigned long long> w0, w1;
  // wide_values = 2: the sort launched on a BOUND of the number of pairs (radix_sort_pairs_bound): the arrays are sized for
  // n + n / 2 + 4097 pairs, the first n are filled, the count sits in a device word
  const bool bound_mode = wide_values == 2;
  if (bound_mode) wide_values = 0;
  const uint32_t n_alloc = bound_mode ? n + n / 2 + 4097u : n;
  PLVS_HIP_TRY(k_in.reserve(n_alloc));
  PLVS_HIP_TRY(k0.reserve(n_alloc));
  PLVS_HIP_TRY(k1.reserve(n_alloc));
  if (wide_values) {
    PLVS_HIP_TRY(w0.reserve(n));
    PLVS_HIP_TRY(w1.reserve(n));
  } else {
    PLVS_HIP_TRY(v0.reserve(n_alloc));
    PLVS_HIP_TRY(v1.reserve(n_alloc));
  }
  PLVS_HIP_TRY(scratch.reserve(radix_scratch_words(n_alloc) + 1));
  PLVS_HIP_TRY(bad.reserve(2));
  PLVS_HIP_TRY(hipMemset(bad.p, 0, 2 * sizeof(uint32_t)));
  const dim3 grid(ceil_div((size_t)n, 256)), block(256);
  hipLaunchKernelGGL(selftest_sort_fill, grid, block, 0, nullptr, k0.p, v0.p, w0.p, n, bit_hi, seed);
  PLVS_HIP_TRY(hipMemcpyAsync(k_in.p, k0.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, nullptr));
  bool second = false;
  if (bound_mode) {
    PLVS_HIP_TRY(hipMemsetAsync(k0.p + n, 0x5A, (size_t)(n_alloc - n) * sizeof(uint32_t), nullptr));   // (what lies behind the pairs is not sorted in)
    uint32_t* d_n = scratch.p + radix_scratch_words(n_alloc);
    PLVS_HIP_TRY(hipMemcpyAsync(d_n, &n, sizeof(uint32_t), hipMemcpyHostToDevice, nullptr));
    PLVS_HIP_TRY(hipStreamSynchronize(nullptr));
    PLVS_HIP_TRY(radix_sort_pairs_bound(k0.p, v0.p, k1.p, v1.p, n_alloc, d_n, bit_lo, bit_hi, scratch.p, nullptr, &second));
  } else if (wide_values) PLVS_HIP_TRY(radix_sort_pairs_u64(k0.p, w0.p, k1.p, w1.p, n, bit_lo, bit_hi, scratch.p, nullptr, &second));
  else PLVS_HIP_TRY(radix_sort_pairs(k0.p, v0.p, k1.p, v1.p, n, bit_lo, bit_hi, scratch.p, nullptr, &second));
  hipLaunchKernelGGL(selftest_sort_check, grid, block, 0, nullptr, k_in.p, second ? k1.p : k0.p,
                     wide_values ? (const uint32_t*)nullptr : (second ? v1.p : v0.p),
                     wide_values ? (second ? w1.p : w0.p) : (const unsigned long long*)nullptr, n, bit_lo, bit_hi, bad.p);
  hipError_t e = hipMemcpy(mismatches2, bad.p, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost);
  k_in.release(); k0.release(); k1.release(); v0.release(); v1.release(); w0.release(); w1.release(); scratch.release(); bad.release();
  PLVS_HIP_TRY(e);
  return PLVS_OK;
}

#ifdef PLVS_WALK_PROF
// developer build only (make PROF=1): the phase clocks of walk_tiles, summed over the tiles since the last reset
int plvs_hip_debug_walk_prof(unsigned long long* out16, int reset) {
  if (out16) PLVS_HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_walk_prof), 16 * sizeof(unsigned long long)));
  if (reset) {
    unsigned long long z[16] = {};
    PLVS_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_walk_prof), z, sizeof z));
  }
  return PLVS_OK;
}
#endif

int plvs_hip_selftest_walk_math(uint32_t seed, uint32_t* mismatches_sqrt_div) {
  PLVS_REQUIRE(mismatches_sqrt_div, "null argument");
  uint32_t* d = nullptr;
  PLVS_HIP_TRY(hipMalloc((void**)&d, 2 * sizeof(uint32_t)));
  PLVS_HIP_TRY(hipMemset(d, 0, 2 * sizeof(uint32_t)));
  hipLaunchKernelGGL(selftest_walk_math_kernel, dim3(16384), dim3(256), 0, nullptr, seed, d);
  hipError_t e = hipMemcpy(mismatches_sqrt_div, d, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  PLVS_HIP_TRY(e);
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_default_params(float resolution, plvs_tsdf_chisel_params* p) {
  PLVS_REQUIRE(p, "params is null");
  PLVS_REQUIRE(resolution > 0.0f, "resolution must be positive");
  p->resolution = resolution;
  p->trunc_quad = 0.0019f;      // ChiselServer.cpp:56-59
  p->trunc_linear = -0.00152f;
  p->trunc_const = 0.001504f;
  p->trunc_scale = 6.0f;
  p->weight = 1.0f;             // ChiselServer.cpp:60 (uint16_t weight = 1)
  p->max_chunks = 32768;        // 2 GiB of voxel pool
  p->shard_rank = 0;
  p->shard_count = 1;
  p->order_free = 0;
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_create(const plvs_tsdf_chisel_params* p, plvs_tsdf_chisel** out) {
  PLVS_REQUIRE(p && out, "null argument");
  PLVS_REQUIRE(p->resolution > 0.0f, "resolution must be positive");
  PLVS_REQUIRE(p->max_chunks > 0 && p->max_chunks <= (1 << 20), "max_chunks must be in (0, 2^20]");
  PLVS_REQUIRE(p->shard_count <= 1 || (p->shard_rank >= 0 && p->shard_rank < p->shard_count),
               "shard_rank out of range");
  plvs_tsdf_chisel* h = new plvs_tsdf_chisel();
  h->prm = *p;
  Params& P = h->P;
  P.resolution = p->resolution;
  P.round_to_voxel = 1.0f / p->resolution;                                   // Chisel.cpp:444
  P.half_voxel = p->resolution * 0.5f;                                       // ChunkManager.cpp:68
  P.rounding = 1.0f / ((float)16 * p->resolution);                           // ChunkManager.cpp:91
  P.diag = (float)(2.0 * std::sqrt((double)3.0f) * (double)p->resolution);   // Chisel.cpp:447
  P.tq = p->trunc_quad; P.tl = p->trunc_linear; P.tc = p->trunc_const; P.ts = p->trunc_scale;
  P.weight = p->weight;
  P.shard_rank = p->shard_rank;
  P.shard_count = p->shard_count < 1 ? 1 : p->shard_count;

  uint32_t cap = 1024;
  while (cap < 2u * (uint32_t)p->max_chunks) cap <<= 1;
  h->dir.mask = cap - 1;
  h->dir.max_blocks = p->max_chunks;
  const size_t nvox = (size_t)p->max_chunks * kChunkVox;
#define CREATE_TRY(call)                                                        \
  do {                                                                          \
    hipError_t _e = (call);                                                     \
    if (_e != hipSuccess) {                                                     \
      plvs::set_error("%s failed: %s", #call, hipGetErrorString(_e));          \
      plvs_hip_tsdf_chisel_destroy(h);                                          \
      return PLVS_ERR_HIP;                                                      \
    }                                                                           \
  } while (0)
  CREATE_TRY(hipMalloc((void**)&h->dir.keys, (size_t)cap * sizeof(unsigned long long)));
  CREATE_TRY(hipMalloc((void**)&h->dir.slots, (size_t)cap * sizeof(int32_t)));
  CREATE_TRY(hipMalloc((void**)&h->dir.slot_ids, (size_t)p->max_chunks * 3 * sizeof(int32_t)));
  CREATE_TRY(hipMalloc((void**)&h->sdf, nvox * sizeof(float)));
  CREATE_TRY(hipMalloc((void**)&h->weight, nvox * sizeof(float)));
  CREATE_TRY(hipMalloc((void**)&h->kfid, nvox * sizeof(uint32_t)));
  CREATE_TRY(hipMalloc((void**)&h->rgbw, nvox * sizeof(uint32_t)));
  CREATE_TRY(hipMalloc((void**)&h->d_ctr, sizeof(Counters)));
  CREATE_TRY(hipHostMalloc((void**)&h->h_ctr, sizeof(Counters), hipHostMallocCoherent));   // (read behind a polled word: wait_published)
  CREATE_TRY(hipMalloc((void**)&h->d_wctr, 2 * sizeof(WalkCounters)));
  CREATE_TRY(hipHostMalloc((void**)&h->h_wctr, 2 * sizeof(WalkCounters), hipHostMallocCoherent));
  CREATE_TRY(hipHostMalloc((void**)&h->h_seq, 64, hipHostMallocCoherent));
  h->h_seq[0] = 0u;
  {
    // Fixed-point scales of the order-free accumulators: a tile adds at most kWalkRays terms per voxel,
    // |w_u u| < weight / 2 and w_u <= weight / (2 diag); the largest powers of two that keep a tile's
    // sums inside 31 bits (one bit of headroom).
    const double wu_max = (double)p->weight / (2.0 * (double)P.diag), wuu_max = 0.5 * (double)p->weight;
    h->scale_u = (float)std::exp2(std::floor(std::log2(1073741824.0 / (kWalkRays * wuu_max))));
    h->scale_w = (float)std::exp2(std::floor(std::log2(1073741824.0 / (kWalkRays * wu_max))));
  }
  {
    // The side stream carries a long call's colour chain — the longer of the two branches behind the walk, a row of short
    // kernels — beside the apply stage's thousands of workgroups on the caller's stream: at the highest priority its
    // workgroups are dispatched ahead of the apply stage's queue instead of behind it.  (developer switch)
    int least = 0, greatest = 0;
    CREATE_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    const bool prio = plvs::env_int("PLVS_TSDF_SIDE_PRIORITY", 1, 0, 1) != 0;
    CREATE_TRY(hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, prio ? greatest : least));
    // (developer switches: the apply stage's parts, as plvs_hip_tsdf_chisel_set_apply_parts sets them)
    h->part_segs = (uint32_t)plvs::env_int("PLVS_APPLY_PART_SEGS", (int)kPartSegs, 1, 1 << 20);
    h->part_min = (uint32_t)plvs::env_int("PLVS_APPLY_PART_MIN", (int)kPartMin, 1, 1 << 20);
  }
  CREATE_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
  CREATE_TRY(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
  CREATE_TRY(hipEventCreateWithFlags(&h->ev_seg, hipEventDisableTiming));
  CREATE_TRY(hipEventCreateWithFlags(&h->ev_zero, hipEventDisableTiming));
#undef CREATE_TRY
  *out = h;
  int rc = plvs_hip_tsdf_chisel_clear(h);
  if (rc != PLVS_OK) {
    plvs_hip_tsdf_chisel_destroy(h);
    *out = nullptr;
  }
  return rc;
}

int plvs_hip_tsdf_chisel_destroy(plvs_tsdf_chisel* h) {
  if (!h) return PLVS_OK;
  if (h->ext != nullptr && h->ext_free != nullptr) h->ext_free(h->ext);
  deform_state_free(h);
  (void)hipFree(h->dir.keys);
  (void)hipFree(h->dir.slots);
  (void)hipFree(h->dir.slot_ids);
  (void)hipFree(h->sdf);
  (void)hipFree(h->weight);
  (void)hipFree(h->kfid);
  (void)hipFree(h->rgbw);
  (void)hipFree(h->d_ctr);
  if (h->h_ctr) (void)hipHostFree(h->h_ctr);
  (void)hipFree(h->d_wctr);
  if (h->h_wctr) (void)hipHostFree(h->h_wctr);
  if (h->h_seq) (void)hipHostFree(h->h_seq);
  if (h->h_offsets) (void)hipHostFree(h->h_offsets);
  (void)hipFree(h->gdir.keys);
  (void)hipFree(h->gdir.slots);
  (void)hipFree(h->miss_keys);
  (void)hipFree(h->miss_ids);
  (void)hipFree(h->miss_count);
  (void)hipFree(h->xdir.keys);
  (void)hipFree(h->xdir.slots);
  (void)hipFree(h->xdir.slot_ids);
  (void)hipFree(h->d_xcount);
  (void)hipFree(h->x_sat);
  if (h->h_sh_counts) (void)hipHostFree(h->h_sh_counts);
  if (h->h_sh_ctl) (void)hipHostFree(h->h_sh_ctl);
  if (h->h_sh_off) (void)hipHostFree(h->h_sh_off);
  h->q_xyz.release(); h->q_rgb.release(); h->q_kfid.release();
  h->w_rec.release(); h->w_seg.release(); h->w_sorted_seg.release(); h->w_chunk_nseg.release();
  h->w_chunk_off.release(); h->w_chunk_fill.release(); h->w_active_off.release(); h->w_masks.release();
  h->w_dummy.release(); h->w_runkey.release(); h->w_run_cnt.release(); h->w_run_off.release(); h->w_val0.release();
  h->w_val1.release(); h->w_seg_cnt.release(); h->w_tile_visits.release(); h->w_deferred.release(); h->w_part_off.release(); h->w_multi_idx.release();
  h->pa_wuu.release(); h->pa_w.release(); h->pa_last.release(); h->pa_cnt.release(); h->pa_done.release();
  h->sh_nrec.release(); h->sh_owner.release(); h->halo_row.release();
  h->sh_ctl.release(); h->sh_seg_reg.release(); h->sh_rec_reg.release(); h->sh_src_off.release(); h->sh_slot_owner.release(); h->sh_run_ctr.release(); h->sh_vkey.release(); h->sh_sat.release(); h->sh_wait.release(); h->sh_run_first.release();
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->ev_seg) (void)hipEventDestroy(h->ev_seg);
  if (h->ev_zero) (void)hipEventDestroy(h->ev_zero);
  h->w_rseg.release(); h->w_rpre.release(); h->w_run_matrix.release(); h->w_active_idx.release(); h->w_item_base.release();
  h->w_item_cnt.release(); h->w_item_part0.release(); h->w_part_item.release(); h->w_phist.release();
  h->w_row_heads.release(); h->w_row_tot.release();
  if (h->side) (void)hipStreamDestroy(h->side);
  for (int i = 0; i <= kNumStages; ++i)
    if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
  h->counts.release();
  h->rec.release(); h->rec_t.release(); h->recc_t.release();
  h->dkey0.release(); h->dkey1.release(); h->didx0.release(); h->didx1.release();
  h->last_pt.release(); h->run_cnt.release(); h->run_dst.release();
  h->tile_first.release(); h->block_first.release(); h->tile_state.release();
  h->heads.release(); h->updated.release(); h->scratch.release(); h->poses.release();
  h->offsets.release(); h->st_xyz.release(); h->st_Twc.release(); h->st_nrm.release(); h->st_rgb.release();
  h->st_kfid.release();
  h->st_pos.release(); h->st_scan.release(); h->st_off.release();
  delete h;
  return PLVS_OK;
}

static int shard_state_clear(plvs_tsdf_chisel* h);
static int halo_drop(plvs_tsdf_chisel* h, hipStream_t s);

int plvs_hip_tsdf_chisel_clear(plvs_tsdf_chisel* h) {
  if (h) {   // (queued clouds belong to the map that is dropped)
    h->q_offsets.clear();
    h->q_Twc.clear();
  }
  PLVS_REQUIRE(h, "null handle");
  {
    int rc = halo_drop(h, nullptr);
    if (rc != PLVS_OK) return rc;
  }
  {
    int rc = shard_state_clear(h);   // (the walk directory of the ray-sharded integrate, if in use)
    if (rc != PLVS_OK) return rc;
  }
  const size_t cap = (size_t)h->dir.mask + 1;
  const size_t nvox = (size_t)h->prm.max_chunks * kChunkVox;
  PLVS_HIP_TRY(hipMemset(h->dir.keys, 0xFF, cap * sizeof(unsigned long long)));
  PLVS_HIP_TRY(hipMemset(h->dir.slots, 0xFF, cap * sizeof(int32_t)));
  PLVS_HIP_TRY(hipMemset(h->weight, 0, nvox * sizeof(float)));
  PLVS_HIP_TRY(hipMemset(h->kfid, 0, nvox * sizeof(uint32_t)));
  PLVS_HIP_TRY(hipMemset(h->rgbw, 0, nvox * sizeof(uint32_t)));
  PLVS_HIP_TRY(hipMemset(h->d_ctr, 0, sizeof(Counters)));
  hipLaunchKernelGGL(pool_init, dim3(2048), dim3(256), 0, nullptr, h->sdf, nvox);
  PLVS_KERNEL_CHECK();
  PLVS_HIP_TRY(hipDeviceSynchronize());
  h->num_chunks = 0;
  h->small_runs_known = false;   // (the first small call on the empty map reads its own run count)
  h->poisoned = false;
  h->stats = plvs_tsdf_stats{};
  h->last_updated = 0;
  deform_state_clear(h);
  return PLVS_OK;
}

}  // extern "C"

// d_normals != nullptr: the world-cloud-with-normals flavour (Chisel::IntegrateWorldPointCloudWithNormals), always
// through the ordered pipeline.
static int integrate_batch_core(plvs_tsdf_chisel* h, const float* d_xyz, const uint8_t* d_rgb, const uint32_t* d_kfid,
                                const int32_t* offsets, int nclouds, const float* d_Twc, void* stream,
                                const float* d_normals);

static int integrate_batch_impl(plvs_tsdf_chisel* h, const float* d_xyz, const uint8_t* d_rgb, const uint32_t* d_kfid,
                                const int32_t* offsets, int nclouds, const float* d_Twc, void* stream,
                                const float* d_normals) {
  PLVS_REQUIRE(h, "null handle");
  // a map with deform enabled keeps the reference's chunk-map order: the call's visits are replayed first
  const bool track = h->dfm != nullptr && !h->poisoned && offsets && nclouds >= 1 && offsets[nclouds] - offsets[0] > 0 && d_xyz &&
                     d_Twc;
  if (track) {
    int rc = halo_drop(h, static_cast<hipStream_t>(stream));
    if (rc == PLVS_OK) rc = deform_track_begin(h, d_xyz, d_normals, offsets[nclouds] - offsets[0], nclouds, d_Twc, static_cast<hipStream_t>(stream));
    if (rc != PLVS_OK) return rc;
  }
  int rc = integrate_batch_core(h, d_xyz, d_rgb, d_kfid, offsets, nclouds, d_Twc, stream, d_normals);
  if (track && rc == PLVS_OK) rc = deform_track_end(h, static_cast<hipStream_t>(stream));
  return rc;
}

static int integrate_batch_core(plvs_tsdf_chisel* h, const float* d_xyz, const uint8_t* d_rgb, const uint32_t* d_kfid,
                                const int32_t* offsets, int nclouds, const float* d_Twc, void* stream,
                                const float* d_normals) {
  PLVS_REQUIRE(h, "null handle");
  PLVS_REQUIRE(!h->poisoned, "handle is in a failed state (clear it)");
  PLVS_REQUIRE(offsets && nclouds >= 0, "bad offsets");
  hipStream_t s = static_cast<hipStream_t>(stream);
  h->stats = plvs_tsdf_stats{};
  h->last_updated = 0;
  if (nclouds == 0) return PLVS_OK;
  const int n = offsets[nclouds] - offsets[0];
  PLVS_REQUIRE(offsets[0] == 0 && n >= 0, "offsets must start at 0 and be non-decreasing");
  for (int c = 0; c < nclouds; ++c) PLVS_REQUIRE(offsets[c + 1] >= offsets[c], "offsets must be non-decreasing");
  h->stats.points = n;
  if (n == 0) return PLVS_OK;
  PLVS_REQUIRE(d_xyz && d_rgb && d_Twc, "null device pointer");
  {
    int rc = halo_drop(h, s);   // new chunks go into the pool slots a meshing halo may still occupy
    if (rc != PLVS_OK) return rc;
  }

  PLVS_HIP_TRY(h->offsets.reserve(2 * ((size_t)nclouds + 1)));
  PLVS_HIP_TRY(h->poses.reserve((size_t)nclouds));
  if (h->prm.order_free != 0 && d_normals == nullptr) return integrate_walk_acc(h, d_xyz, d_rgb, d_kfid, n, nclouds, offsets, d_Twc, s);
  PLVS_HIP_TRY(h->counts.reserve((size_t)n + 1));
  PLVS_HIP_TRY(h->scratch.reserve(scan_scratch_words((size_t)n)));
  PLVS_HIP_TRY(hipMemcpyAsync(h->offsets.p, offsets, ((size_t)nclouds + 1) * sizeof(int32_t),
                              hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(pose_prep, dim3(ceil_div((size_t)nclouds, 64)), dim3(64), 0, s, d_Twc, nclouds,
                     h->poses.p);
  // reset the per-call counters, keep num_chunks
  PLVS_HIP_TRY(hipMemsetAsync(&h->d_ctr->total_visits, 0, sizeof(uint32_t), s));
  PLVS_HIP_TRY(hipMemsetAsync(&h->d_ctr->err, 0, 5 * sizeof(uint32_t), s));

  h->stage_set = 0;

#define STAGE_MARK(i) \
  do { if (h->profiling) PLVS_HIP_TRY(hipEventRecord(h->ev[i], s)); } while (0)
  STAGE_MARK(0);
  if (d_normals != nullptr)
    hipLaunchKernelGGL(ray_count<true>, dim3(ceil_div((size_t)n, 256)), dim3(256), 0, s, h->P, d_xyz, d_normals, n,
                       h->offsets.p, nclouds, h->poses.p, h->dir, h->d_ctr, h->counts.p);
  else
    hipLaunchKernelGGL(ray_count<false>, dim3(ceil_div((size_t)n, 256)), dim3(256), 0, s, h->P, d_xyz, d_normals, n,
                       h->offsets.p, nclouds, h->poses.p, h->dir, h->d_ctr, h->counts.p);
  PLVS_KERNEL_CHECK();
  STAGE_MARK(1);
  // counts -> visit offsets (n + 1 entries: the total closes the list)
  PLVS_HIP_TRY(exclusive_scan_u32(h->counts.p, h->counts.p, (size_t)n, &h->d_ctr->total_visits,
                                  h->scratch.p, s));
  PLVS_HIP_TRY(hipMemcpyAsync(h->counts.p + n, &h->d_ctr->total_visits, sizeof(uint32_t),
                              hipMemcpyDeviceToDevice, s));
  STAGE_MARK(2);
  int rc = read_counters(h, s);
  if (rc != PLVS_OK) return rc;
  if (h->h_ctr->err) {
    h->poisoned = true;
    plvs::set_error("tsdf_chisel integrate: %s%s",
                    (h->h_ctr->err & kErrPoolFull) ? "chunk pool full (raise max_chunks) " : "",
                    (h->h_ctr->err & kErrCoordRange) ? "voxel coordinates beyond +-2^20 (outside the supported map extent) " : "");
    return PLVS_ERR_CAPACITY;
  }
  const uint32_t V = h->h_ctr->total_visits;
  const int chunks_before = h->num_chunks;
  h->num_chunks = h->h_ctr->num_chunks;
  h->stats.visits = V;
  h->stats.new_chunks = h->num_chunks - chunks_before;
  if (V == 0) return PLVS_OK;
  if (V >= (1u << 29)) {
    plvs::set_error("tsdf_chisel integrate: %u voxel visits in one call exceed the 2^29 limit (split the batch)", V);
    return PLVS_ERR_CAPACITY;
  }

  const uint32_t ntiles = ceil_div(V, kTileSlots);
  PLVS_HIP_TRY(h->dkey0.reserve(V));
  PLVS_HIP_TRY(h->tile_first.reserve(ntiles));
  PLVS_HIP_TRY(h->tile_state.reserve((size_t)ntiles + 1));
  PLVS_HIP_TRY(h->updated.reserve((size_t)h->num_chunks + 1));
  PLVS_HIP_TRY(h->rec_t.reserve(V));
  PLVS_HIP_TRY(h->recc_t.reserve(V));
  PLVS_HIP_TRY(h->rec.reserve((size_t)V + 2));   // chain_runs reads record pairs
  PLVS_HIP_TRY(h->didx0.reserve(V));
  PLVS_HIP_TRY(h->last_pt.reserve(V));
  PLVS_HIP_TRY(h->block_first.reserve(ceil_div(V, kGatherSpan)));
  float ms_a[2] = {0.f, 0.f};
  if (h->profiling) {  // stages 0,1 are complete (the counter read synchronised)
    PLVS_HIP_TRY(stage_elapsed(&ms_a[0], h->ev[0], h->ev[1]));
    PLVS_HIP_TRY(stage_elapsed(&ms_a[1], h->ev[1], h->ev[2]));
  }
  STAGE_MARK(2);
  PLVS_HIP_TRY(hipMemsetAsync(h->tile_state.p, 0, ((size_t)ntiles + 1) * sizeof(unsigned long long), s));
  hipLaunchKernelGGL(mark_tiles, dim3(ceil_div((size_t)n, 256)), dim3(256), 0, s, h->counts.p, n,
                     h->tile_first.p);
  {
    // per-visit (u, point): in the buffer the gather fills later
    TileOut out{h->rec.p, h->rec_t.p, h->recc_t.p, h->dkey0.p, h->didx0.p, h->last_pt.p};
    if (d_normals != nullptr)
      hipLaunchKernelGGL(ray_tiles<true>, dim3(ntiles), dim3(kTileThreads), 0, s, h->P, d_xyz, d_normals, d_rgb, n,
                         h->offsets.p, nclouds, h->poses.p, h->dir, h->d_ctr, h->counts.p, V, h->tile_first.p, ntiles,
                         reinterpret_cast<uint32_t*>(h->tile_state.p), h->tile_state.p + 1, h->rgbw, out);
    else
      hipLaunchKernelGGL(ray_tiles<false>, dim3(ntiles), dim3(kTileThreads), 0, s, h->P, d_xyz, d_normals, d_rgb, n,
                         h->offsets.p, nclouds, h->poses.p, h->dir, h->d_ctr, h->counts.p, V, h->tile_first.p, ntiles,
                         reinterpret_cast<uint32_t*>(h->tile_state.p), h->tile_state.p + 1, h->rgbw, out);
  }
  PLVS_KERNEL_CHECK();
  STAGE_MARK(3);
  rc = read_counters(h, s);   // the number of runs sizes the sort
  if (rc != PLVS_OK) return rc;
  if (h->h_ctr->err) {
    h->poisoned = true;
    plvs::set_error("tsdf_chisel integrate: internal directory miss (err=%u)", h->h_ctr->err);
    return PLVS_ERR_CAPACITY;
  }
  const uint32_t D = h->h_ctr->num_desc;
  float ms_b = 0.f;
  if (h->profiling) PLVS_HIP_TRY(stage_elapsed(&ms_b, h->ev[2], h->ev[3]));
  PLVS_HIP_TRY(h->dkey1.reserve(D));
  PLVS_HIP_TRY(h->scratch.reserve(radix_scratch_words(D)));
  int key_bits = 12;
  while ((1ll << (key_bits - 12)) < (long long)h->num_chunks) ++key_bits;
  bool second = false;
  {
    PLVS_HIP_TRY(h->didx1.reserve(D));
    PLVS_HIP_TRY(h->run_cnt.reserve(D));
    PLVS_HIP_TRY(h->run_dst.reserve(D));
    STAGE_MARK(3);
    PLVS_HIP_TRY(radix_sort_pairs_u64(h->dkey0.p, h->didx0.p, h->dkey1.p, h->didx1.p, D, 0, key_bits,
                                  h->scratch.p, s, &second));
    const uint32_t* skeys = second ? h->dkey1.p : h->dkey0.p;
    const unsigned long long* sidx = second ? h->didx1.p : h->didx0.p;
    STAGE_MARK(4);
    PLVS_HIP_TRY(h->heads.reserve(D));
    hipLaunchKernelGGL(voxel_heads, dim3(ceil_div(D, 256 * kHeadTiles)), dim3(256), 0, s, skeys, D, h->heads.p,
                       h->updated.p, h->d_ctr);
    // The colour fold (truncating u8 mean, exact: fold_colours) reads the tile-ordered colours through
    // the sorted runs and touches only rgbw, so it runs on a second stream beside the gather; the
    // distance chain then has the machine to itself.
    PLVS_HIP_TRY(hipEventRecord(h->ev_fork, s));
    PLVS_HIP_TRY(hipStreamWaitEvent(h->side, h->ev_fork, 0));
    if (d_normals != nullptr)
      hipLaunchKernelGGL(fold_colours<true>, dim3(std::min<size_t>(ceil_div(D, 256), 1024)), dim3(256), 0, h->side,
                         skeys, sidx, D, h->heads.p, h->recc_t.p, h->rgbw, h->d_ctr);
    else
      hipLaunchKernelGGL(fold_colours<false>, dim3(std::min<size_t>(ceil_div(D, 256), 1024)), dim3(256), 0, h->side,
                         skeys, sidx, D, h->heads.p, h->recc_t.p, h->rgbw, h->d_ctr);
    PLVS_HIP_TRY(hipEventRecord(h->ev_join, h->side));
    hipLaunchKernelGGL(run_counts, dim3(ceil_div(D, 256)), dim3(256), 0, s, sidx, D, h->run_cnt.p);
    PLVS_HIP_TRY(exclusive_scan_u32(h->run_cnt.p, h->run_dst.p, D, nullptr, h->scratch.p, s));
    const uint32_t nblocks = ceil_div(V, kGatherSpan);
    hipLaunchKernelGGL(mark_blocks, dim3(ceil_div(D, 256)), dim3(256), 0, s, h->run_dst.p, D, V, h->block_first.p);
    hipLaunchKernelGGL(gather_runs, dim3(nblocks), dim3(kGatherThreads), 0, s, skeys, sidx, D, h->last_pt.p,
                       h->run_dst.p, h->block_first.p, nblocks, V, h->rec_t.p, h->recc_t.p, h->rec.p,
                       (uint32_t*)nullptr, d_kfid, h->kfid);
    PLVS_KERNEL_CHECK();
    STAGE_MARK(5);
    // one thread per voxel; the grid is an upper bound of the voxel count, surplus waves exit
    // on the device-side count
    hipLaunchKernelGGL(chain_runs, dim3(std::min<size_t>(ceil_div(D, 64), 16384)), dim3(64), 0, s,
                       h->heads.p, skeys, h->run_dst.p, V, h->rec.p, h->d_ctr, h->sdf, h->weight);
    PLVS_HIP_TRY(hipStreamWaitEvent(s, h->ev_join, 0));
    PLVS_KERNEL_CHECK();
    STAGE_MARK(6);
  }
#undef STAGE_MARK
  rc = read_counters(h, s);
  if (rc != PLVS_OK) return rc;
  if (h->profiling) {
    h->stage_ms[0] += ms_a[0];
    h->stage_ms[1] += ms_a[1];
    h->stage_ms[2] += ms_b;
    for (int i = 3; i < kNumStages; ++i) {
      float ms = 0.f;
      PLVS_HIP_TRY(stage_elapsed(&ms, h->ev[i], h->ev[i + 1]));
      h->stage_ms[i] += ms;
    }
    h->prof_calls++;
  }
  if (h->h_ctr->err) {
    h->poisoned = true;
    plvs::set_error("tsdf_chisel integrate: internal directory miss (err=%u)", h->h_ctr->err);
    return PLVS_ERR_CAPACITY;
  }
  h->stats.updated_chunks = (int32_t)h->h_ctr->num_updated;
  h->stats.voxels = (int32_t)h->h_ctr->num_heads;
  h->stats.max_run = (int32_t)h->h_ctr->max_run;
#if PLVS_TILE_PROBE
  {
    const unsigned long long* q = h->h_ctr->tprobe;
    fprintf(stderr, "tile probe (cycles/tile): setup %llu raycast %llu group %llu radix %llu heads+lookback %llu out %llu | tiles %u runs %u\n",
            q[0] / ntiles, q[1] / ntiles, q[2] / ntiles, q[3] / ntiles, q[5] / ntiles, q[6] / ntiles, ntiles, D);
    (void)hipMemsetAsync(h->d_ctr->tprobe, 0, sizeof(h->d_ctr->tprobe), s);
  }
#endif
#if PLVS_CHAIN_PROBE
  {
    const unsigned long long m = (1ull << 40) - 1;
    const unsigned long long* q = h->h_ctr->probe;
    fprintf(stderr, "chain probe: passes %llu cycles %llu wall100MHz %llu | stage+fetch %llu chain %llu prepare %llu tail %llu\n",
            q[0] >> 40, q[0] & m, q[1] & m, q[2] & m, q[3] & m, q[4] & m, q[5] & m);
    (void)hipMemsetAsync(h->d_ctr->probe, 0, sizeof(h->d_ctr->probe), s);
  }
#endif
  h->last_updated = h->h_ctr->num_updated;
  return PLVS_OK;
}

extern "C" {

int plvs_hip_tsdf_chisel_integrate_batch_dev(plvs_tsdf_chisel* h, const float* d_xyz,
                                             const uint8_t* d_rgb, const uint32_t* d_kfid,
                                             const int32_t* offsets, int nclouds,
                                             const float* d_Twc, void* stream) {
  PLVS_FLUSH_QUEUE(h);
  return integrate_batch_impl(h, d_xyz, d_rgb, d_kfid, offsets, nclouds, d_Twc, stream, nullptr);
}

// ---- depth images straight into the map (round 5): GeneratePointCloudInCameraFrameBGRA + InsertCloud in one call.
// Order-free handles walk 32 x 16 blocks of grid pixels (GridSrc, tsdf_walk.hpp): the cloud is never written.  Ordered
// handles (and sharded ones) build the reference's clouds in scratch memory — raster-order compaction of the valid grid
// pixels — and take the ordinary batch path: bit for bit plvs_hip_cloudgen_generate_dev + integrate_batch_dev.
int plvs_hip_tsdf_chisel_integrate_depth_batch_dev(plvs_tsdf_chisel* h, const plvs_depth_batch* in, int nclouds,
                                                   const float* d_Twc, void* stream) {
  PLVS_FLUSH_QUEUE(h);
  PLVS_REQUIRE(h && in, "null argument");
  PLVS_REQUIRE(!h->poisoned, "handle is in a failed state (clear it)");
  PLVS_REQUIRE(nclouds >= 0, "bad image count");
  h->stats = plvs_tsdf_stats{};
  h->last_updated = 0;
  if (nclouds == 0) return PLVS_OK;
  PLVS_REQUIRE(in->d_depth && in->d_bgr && in->d_grid_points && d_Twc, "null device pointer");
  PLVS_REQUIRE(in->width > 0 && in->height > 0 && in->step > 0, "image size / step");
  PLVS_REQUIRE(in->depth_pitch >= in->width && in->bgr_pitch >= 3 * in->width, "row pitch smaller than a row");
  PLVS_REQUIRE(in->depth_image_stride >= (size_t)in->depth_pitch * (size_t)(in->height - 1) + (size_t)in->width &&
               in->bgr_image_stride >= (size_t)in->bgr_pitch * (size_t)(in->height - 1) + 3 * (size_t)in->width,
               "image stride smaller than an image");
  hipStream_t s = static_cast<hipStream_t>(stream);
  GridSrc g{};
  g.depth = in->d_depth;
  g.cam = in->d_grid_points;
  g.image_stride = in->depth_image_stride;
  g.pitch = (uint32_t)in->depth_pitch;
  g.step = (uint32_t)in->step;
  g.gw = (uint32_t)((in->width + in->step - 1) / in->step);
  g.gh = (uint32_t)((in->height + in->step - 1) / in->step);
  g.ntx = (g.gw + kGridTileW - 1) / kGridTileW;
  g.nty = (g.gh + kGridTileH - 1) / kGridTileH;
  g.inv_ntx = 1.0f / (float)g.ntx;
  g.inv_nty = 1.0f / (float)g.nty;
  g.key_bits = 1;
  while ((1ull << g.key_bits) < (unsigned long long)g.gw * g.gh) ++g.key_bits;
  g.min_depth = in->min_depth;
  g.max_depth = in->max_depth;
  g.bgr_image_stride = in->bgr_image_stride;
  g.bgr_pitch = (uint32_t)in->bgr_pitch;
  PLVS_REQUIRE(g.key_bits < 31 && (unsigned long long)nclouds <= (1ull << (32 - g.key_bits)),
               "too many images in one call for the order keys (split the batch)");
  const size_t ngrid = (size_t)g.gw * g.gh;
  h->stats.points = 0;
  {
    int rc = halo_drop(h, s);   // new chunks go into the pool slots a meshing halo may still occupy
    if (rc != PLVS_OK) return rc;
  }
  const bool walk2d = h->prm.order_free != 0 && std::max(1, h->prm.shard_count) == 1 && h->dfm == nullptr;
  if (walk2d) {
    std::vector<int32_t> zeros((size_t)nclouds + 1, 0);
    PLVS_HIP_TRY(h->offsets.reserve(2 * ((size_t)nclouds + 1)));
    PLVS_HIP_TRY(h->poses.reserve((size_t)nclouds));
    return integrate_walk_acc(h, nullptr, in->d_bgr, in->d_kfid, 0, nclouds, zeros.data(), d_Twc, s, &g);
  }
  // ---- the reference's clouds, in scratch memory
  const size_t cells = ngrid * (size_t)nclouds;
  PLVS_REQUIRE(cells < 0x7FFFFFFFull, "too many grid pixels in one call (split the batch)");
  // (the scratch of this path lives with the handle: allocated on the handle's device, released by destroy)
  plvs::DevBuf<uint32_t>&pos = h->st_pos, &scan_scratch = h->st_scan, &d_off = h->st_off;
  PLVS_HIP_TRY(pos.reserve(cells + 1));
  PLVS_HIP_TRY(scan_scratch.reserve(scan_scratch_words(cells)));
  PLVS_HIP_TRY(d_off.reserve((size_t)nclouds + 1));
  PLVS_HIP_TRY(h->st_xyz.reserve(cells * 3));
  PLVS_HIP_TRY(h->st_rgb.reserve(cells * 3));
  PLVS_HIP_TRY(h->st_kfid.reserve(cells));
  hipLaunchKernelGGL(grid_cloud_mark, dim3(ceil_div(cells, 256)), dim3(256), 0, s, g, nclouds, pos.p);
  PLVS_HIP_TRY(exclusive_scan_u32(pos.p, pos.p, cells, pos.p + cells, scan_scratch.p, s));
  hipLaunchKernelGGL(grid_cloud_emit, dim3(ceil_div(cells + 1, 256)), dim3(256), 0, s, g, nclouds, (const uint32_t*)pos.p, in->d_bgr,
                     in->d_kfid, h->st_xyz.p, h->st_rgb.p, h->st_kfid.p, d_off.p);
  PLVS_KERNEL_CHECK();
  std::vector<int32_t> offsets((size_t)nclouds + 1);
  PLVS_HIP_TRY(hipMemcpyAsync(offsets.data(), d_off.p, ((size_t)nclouds + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  return integrate_batch_impl(h, h->st_xyz.p, h->st_rgb.p, in->d_kfid ? h->st_kfid.p : nullptr, offsets.data(), nclouds, d_Twc,
                              stream, nullptr);
}

int plvs_hip_tsdf_chisel_integrate_world_normals_dev(plvs_tsdf_chisel* h, const float* d_xyz, const uint8_t* d_rgb,
                                                     const uint32_t* d_kfid, const float* d_normals, int n,
                                                     const float* d_Twc, void* stream) {
  PLVS_FLUSH_QUEUE(h);
  PLVS_REQUIRE(h && n >= 0, "bad arguments");
  PLVS_REQUIRE(n == 0 || d_normals, "null normals");
  PLVS_REQUIRE(std::max(1, h->prm.shard_count) == 1 || h->prm.order_free == 0,
               "a ray-sharded (order_free) map takes the world cloud on the rank that owns each chunk: use an ordered sharded handle");
  const int32_t offsets[2] = {0, n};
  return integrate_batch_impl(h, d_xyz, d_rgb, d_kfid, offsets, 1, d_Twc, stream, d_normals);
}

int plvs_hip_tsdf_chisel_integrate_world_normals(plvs_tsdf_chisel* h, const float* xyz, const uint8_t* rgb,
                                                 const uint32_t* kfid, const float* normals, int n, const float* Twc) {
  PLVS_FLUSH_QUEUE(h);
  PLVS_REQUIRE(h, "null handle");
  PLVS_REQUIRE(n >= 0 && Twc, "bad arguments");
  h->stats = plvs_tsdf_stats{};
  h->last_updated = 0;
  if (n == 0) return PLVS_OK;
  PLVS_REQUIRE(xyz && rgb && normals, "null input");
  PLVS_HIP_TRY(h->st_xyz.reserve((size_t)n * 3));
  PLVS_HIP_TRY(h->st_rgb.reserve((size_t)n * 3));
  PLVS_HIP_TRY(h->st_Twc.reserve(12));
  PLVS_HIP_TRY(h->st_nrm.reserve((size_t)n * 3));
  if (kfid) PLVS_HIP_TRY(h->st_kfid.reserve((size_t)n));
  PLVS_HIP_TRY(hipMemcpy(h->st_xyz.p, xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
  PLVS_HIP_TRY(hipMemcpy(h->st_rgb.p, rgb, (size_t)n * 3, hipMemcpyHostToDevice));
  PLVS_HIP_TRY(hipMemcpy(h->st_nrm.p, normals, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
  PLVS_HIP_TRY(hipMemcpy(h->st_Twc.p, Twc, 12 * sizeof(float), hipMemcpyHostToDevice));
  if (kfid) PLVS_HIP_TRY(hipMemcpy(h->st_kfid.p, kfid, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice));
  const int rc = plvs_hip_tsdf_chisel_integrate_world_normals_dev(h, h->st_xyz.p, h->st_rgb.p, kfid ? h->st_kfid.p : nullptr,
                                                                  h->st_nrm.p, n, h->st_Twc.p, nullptr);
  (void)hipDeviceSynchronize();
  return rc;
}

// ---- queued integration.  PLVS hands a key frame's cloud over with InsertCloud and reads the map only in UpdateMap, after
// at most five of them (src/PointCloudMapping.cc:540-552, 594-598).  _queue uploads the cloud and returns; _flush
// integrates everything queued in ONE call of the batch pipeline — the same result as integrating the clouds one by one
// (bit for bit in the ordered mode: the batch pipeline applies every update in point order across the clouds; within the
// stated tolerance in the order-free mode, where a voxel takes one update per CALL) at a fraction of the per-call
// launch chain.  Every entry point that reads or changes the map flushes first.
int plvs_hip_tsdf_chisel_queue(plvs_tsdf_chisel* h, const float* xyz, const uint8_t* rgb, const uint32_t* kfid, int n,
                               const float* Twc) {
  PLVS_REQUIRE(h, "null handle");
  PLVS_REQUIRE(n >= 0 && Twc, "bad arguments");
  if (n == 0) return PLVS_OK;
  PLVS_REQUIRE(xyz && rgb, "null cloud pointer");
  const bool first = h->q_offsets.empty();
  PLVS_REQUIRE(first || h->q_kfid_given == (kfid != nullptr), "queued clouds must all carry key-frame ids, or none");
  const size_t at = first ? 0 : (size_t)h->q_offsets.back();
  PLVS_REQUIRE(at + (size_t)n < 0x7FFFFFFFull, "too many queued points");
  PLVS_HIP_TRY(grow_keep(h->q_xyz, 3 * at, 3 * (at + (size_t)n)));
  PLVS_HIP_TRY(grow_keep(h->q_rgb, 3 * at, 3 * (at + (size_t)n)));
  PLVS_HIP_TRY(hipMemcpy(h->q_xyz.p + 3 * at, xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
  PLVS_HIP_TRY(hipMemcpy(h->q_rgb.p + 3 * at, rgb, (size_t)n * 3, hipMemcpyHostToDevice));
  if (kfid) {
    PLVS_HIP_TRY(grow_keep(h->q_kfid, at, at + (size_t)n));
    PLVS_HIP_TRY(hipMemcpy(h->q_kfid.p + at, kfid, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  if (first) h->q_offsets.push_back(0);
  h->q_kfid_given = kfid != nullptr;
  h->q_offsets.push_back((int32_t)(at + (size_t)n));
  h->q_Twc.insert(h->q_Twc.end(), Twc, Twc + 12);
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_queued(plvs_tsdf_chisel* h, int* nclouds) {
  PLVS_REQUIRE(h && nclouds, "null argument");
  *nclouds = h->q_offsets.empty() ? 0 : (int)h->q_offsets.size() - 1;
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_flush(plvs_tsdf_chisel* h) {
  PLVS_REQUIRE(h, "null handle");
  if (h->q_offsets.empty()) return PLVS_OK;
  const int nclouds = (int)h->q_offsets.size() - 1;
  std::vector<int32_t> offsets;
  std::vector<float> Twc;
  offsets.swap(h->q_offsets);   // (the queue is empty from here on: the integrate below flushes nothing)
  Twc.swap(h->q_Twc);
  PLVS_HIP_TRY(h->st_Twc.reserve((size_t)12 * nclouds));
  PLVS_HIP_TRY(hipMemcpy(h->st_Twc.p, Twc.data(), (size_t)12 * nclouds * sizeof(float), hipMemcpyHostToDevice));
  int rc = plvs_hip_tsdf_chisel_integrate_batch_dev(h, h->q_xyz.p, h->q_rgb.p, h->q_kfid_given ? h->q_kfid.p : nullptr,
                                                    offsets.data(), nclouds, h->st_Twc.p, nullptr);
  if (rc != PLVS_OK) {
    // The queue was taken before the batch ran (a reader that flushes must not flush again from inside it): its clouds are
    // gone.  Say so, and how many — the error surfaces from whichever call triggered the flush, possibly a reader.
    char own[400];
    snprintf(own, sizeof own, "%s", plvs::last_error_buf());
    plvs::set_error("%s — raised by the flush of %d queued key-frame cloud%s (plvs_hip_tsdf_chisel_queue): NONE of them was "
                    "integrated and they are dropped; queue them again after clearing / enlarging the map", own, nclouds,
                    nclouds == 1 ? "" : "s");
    return rc;
  }
  PLVS_HIP_TRY(hipDeviceSynchronize());
  return PLVS_OK;
}
int plvs_hip_tsdf_chisel_integrate(plvs_tsdf_chisel* h, const float* xyz, const uint8_t* rgb,
                                   const uint32_t* kfid, int n, const float* Twc) {
  PLVS_REQUIRE(h, "null handle");
  PLVS_REQUIRE(n >= 0 && Twc, "bad arguments");
  PLVS_FLUSH_QUEUE(h);
  if (n == 0) {
    h->stats = plvs_tsdf_stats{};
    h->last_updated = 0;
    return PLVS_OK;
  }
  PLVS_REQUIRE(xyz && rgb, "null cloud pointer");
  PLVS_HIP_TRY(h->st_xyz.reserve((size_t)n * 3));
  PLVS_HIP_TRY(h->st_rgb.reserve((size_t)n * 3));
  PLVS_HIP_TRY(h->st_Twc.reserve(12));
  PLVS_HIP_TRY(hipMemcpy(h->st_xyz.p, xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
  PLVS_HIP_TRY(hipMemcpy(h->st_rgb.p, rgb, (size_t)n * 3, hipMemcpyHostToDevice));
  PLVS_HIP_TRY(hipMemcpy(h->st_Twc.p, Twc, 12 * sizeof(float), hipMemcpyHostToDevice));
  const uint32_t* dk = nullptr;
  if (kfid) {
    PLVS_HIP_TRY(h->st_kfid.reserve((size_t)n));
    PLVS_HIP_TRY(hipMemcpy(h->st_kfid.p, kfid, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice));
    dk = h->st_kfid.p;
  }
  const int32_t offsets[2] = {0, n};
  int rc = plvs_hip_tsdf_chisel_integrate_batch_dev(h, h->st_xyz.p, h->st_rgb.p, dk, offsets, 1,
                                                    h->st_Twc.p, nullptr);
  if (rc != PLVS_OK) return rc;
  PLVS_HIP_TRY(hipDeviceSynchronize());
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_set_apply_parts(plvs_tsdf_chisel* h, int part_segments, int min_segments) {
  PLVS_REQUIRE(h && part_segments >= 1 && min_segments >= 1, "bad argument");
  h->part_segs = (uint32_t)part_segments;
  h->part_min = (uint32_t)min_segments;
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_set_profiling(plvs_tsdf_chisel* h, int enable) {
  PLVS_REQUIRE(h, "null handle");
  if (enable && !h->ev[0])
    for (int i = 0; i <= kNumStages; ++i) PLVS_HIP_TRY(hipEventCreate(&h->ev[i]));
  h->profiling = enable != 0;
  for (int i = 0; i < kNumStages; ++i) h->stage_ms[i] = 0.0;
  h->prof_calls = 0;
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_stage_ms(plvs_tsdf_chisel* h, double* ms, int cap, int* nstages,
                                  int64_t* calls) {
  PLVS_REQUIRE(h && nstages, "null argument");
  *nstages = h->stage_set == 1 ? kWalkStages : kNumStages;
  if (calls) *calls = h->prof_calls;
  for (int i = 0; i < kNumStages && i < cap; ++i) ms[i] = h->stage_ms[i];
  return PLVS_OK;
}

const char* plvs_hip_tsdf_chisel_stage_name(int i) {
  return (i >= 0 && i < kNumStages) ? kStageNames[i] : "";
}

const char* plvs_hip_tsdf_chisel_stage_name_of(plvs_tsdf_chisel* h, int i) {
  if (h == nullptr || i < 0) return "";
  if (h->stage_set == 1) return i < kWalkStages ? kWalkStageNames[i] : "";
  return i < kNumStages ? kStageNames[i] : "";
}

__global__ void gather_slot_ids(const uint32_t* __restrict__ slots, int n,
                                const int32_t* __restrict__ slot_ids, int32_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint32_t s = slots[i];
    out[3 * i] = slot_ids[3 * s];
    out[3 * i + 1] = slot_ids[3 * s + 1];
    out[3 * i + 2] = slot_ids[3 * s + 2];
  }
}

int plvs_hip_tsdf_chisel_updated_chunk_ids_dev(plvs_tsdf_chisel* h, int32_t* d_ids_xyz, int cap,
                                               int* n, void* stream) {
  PLVS_FLUSH_QUEUE(h);
  PLVS_REQUIRE(h && n, "null argument");
  *n = (int)h->last_updated;
  const int m = (int)h->last_updated < cap ? (int)h->last_updated : cap;
  if (m <= 0) return PLVS_OK;
  PLVS_REQUIRE(d_ids_xyz, "null output");
  hipLaunchKernelGGL(gather_slot_ids, dim3(ceil_div((size_t)m, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), h->updated.p, m, h->dir.slot_ids, d_ids_xyz);
  PLVS_KERNEL_CHECK();
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_last_stats(plvs_tsdf_chisel* h, plvs_tsdf_stats* s) {
  PLVS_FLUSH_QUEUE(h);
  PLVS_REQUIRE(h && s, "null argument");
  *s = h->stats;
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_num_chunks(plvs_tsdf_chisel* h, int* n) {
  PLVS_FLUSH_QUEUE(h);
  PLVS_REQUIRE(h && n, "null argument");
  *n = h->num_chunks;
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_chunk_ids(plvs_tsdf_chisel* h, int32_t* ids_xyz, int cap, int* n) {
  PLVS_FLUSH_QUEUE(h);
  PLVS_REQUIRE(h && n, "null argument");
  *n = h->num_chunks;
  const int m = h->num_chunks < cap ? h->num_chunks : cap;
  if (m > 0) {
    PLVS_REQUIRE(ids_xyz, "null output");
    PLVS_HIP_TRY(hipMemcpy(ids_xyz, h->dir.slot_ids, (size_t)m * 3 * sizeof(int32_t),
                           hipMemcpyDeviceToHost));
  }
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_updated_chunk_ids(plvs_tsdf_chisel* h, int32_t* ids_xyz, int cap, int* n) {
  PLVS_FLUSH_QUEUE(h);
  PLVS_REQUIRE(h && n, "null argument");
  *n = (int)h->last_updated;
  const int m = (int)h->last_updated < cap ? (int)h->last_updated : cap;
  if (m <= 0) return PLVS_OK;
  PLVS_REQUIRE(ids_xyz, "null output");
  // small lists: resolve slot -> id on the host
  uint32_t* slots = new uint32_t[h->last_updated];
  int32_t* all = new int32_t[(size_t)h->num_chunks * 3];
  hipError_t e1 = hipMemcpy(slots, h->updated.p, (size_t)h->last_updated * sizeof(uint32_t),
                            hipMemcpyDeviceToHost);
  hipError_t e2 = hipMemcpy(all, h->dir.slot_ids, (size_t)h->num_chunks * 3 * sizeof(int32_t),
                            hipMemcpyDeviceToHost);
  if (e1 == hipSuccess && e2 == hipSuccess)
    for (int i = 0; i < m; ++i) memcpy(ids_xyz + 3 * i, all + 3 * (size_t)slots[i], 3 * sizeof(int32_t));
  delete[] slots;
  delete[] all;
  PLVS_HIP_TRY(e1);
  PLVS_HIP_TRY(e2);
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_download_chunk(plvs_tsdf_chisel* h, int cx, int cy, int cz, float* sdf,
                                        float* weight, uint32_t* kfid, uint32_t* rgbw) {
  PLVS_FLUSH_QUEUE(h);
  PLVS_REQUIRE(h && sdf && weight && kfid && rgbw, "null argument");
  // linear search of the (host-copied) slot table; a download is a debug /
  // meshing hand-off, not part of the integrate path.
  int32_t* all = new int32_t[(size_t)(h->num_chunks > 0 ? h->num_chunks : 1) * 3];
  hipError_t e = hipSuccess;
  if (h->num_chunks > 0)
    e = hipMemcpy(all, h->dir.slot_ids, (size_t)h->num_chunks * 3 * sizeof(int32_t),
                  hipMemcpyDeviceToHost);
  int slot = -1;
  if (e == hipSuccess)
    for (int i = 0; i < h->num_chunks; ++i)
      if (all[3 * i] == cx && all[3 * i + 1] == cy && all[3 * i + 2] == cz) { slot = i; break; }
  delete[] all;
  PLVS_HIP_TRY(e);
  if (slot < 0) {
    plvs::set_error("chunk (%d,%d,%d) does not exist", cx, cy, cz);
    return PLVS_ERR_INVALID_ARG;
  }
  const size_t off = (size_t)slot * kChunkVox;
  PLVS_HIP_TRY(hipMemcpy(sdf, h->sdf + off, kChunkVox * sizeof(float), hipMemcpyDeviceToHost));
  PLVS_HIP_TRY(hipMemcpy(weight, h->weight + off, kChunkVox * sizeof(float), hipMemcpyDeviceToHost));
  PLVS_HIP_TRY(hipMemcpy(kfid, h->kfid + off, kChunkVox * sizeof(uint32_t), hipMemcpyDeviceToHost));
  PLVS_HIP_TRY(hipMemcpy(rgbw, h->rgbw + off, kChunkVox * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return PLVS_OK;
}

}  // extern "C"

// PinholeCamera::SetupFrustum -> Frustum::SetFromParams / SetFromVectors (PinholeCamera.cpp:55-59,
// Frustum.cpp:150-196), Plane(a, b, c) (Plane.cpp:46-54), Frustum::ComputeBoundingBox (:100-125),
// ChunkManager::GetChunkIDsIntersecting's id range (ChunkManager.cpp:248-257).  Literal, including
// fy handed over for fx, the plane distance of the unnormalised normal, and the double atan2 / tan.
static void carve_frustum(const Params& P, const float* Twc, float near_d, float far_d, float fy, float cy,
                          float width, float height, CarveCamera* C) {
  auto s3 = [](float a, float b, float c) { return a + (b + c); };
  float R[9], t[3];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) R[3 * i + j] = Twc[4 * i + j];
    t[i] = Twc[4 * i + 3];
  }
  float right[3], up[3], fwd[3];
  for (int i = 0; i < 3; ++i) { right[i] = R[3 * i]; up[i] = -R[3 * i + 1]; fwd[i] = R[3 * i + 2]; }
  const float fx = fy;
  const float aspect = (fx * width) / (fy * height);
  const float fov = (float)(atan2((double)cy, (double)fy) + atan2((double)(height - cy), (double)fy));
  const float tang = (float)tan((double)(fov / 2));
  const float hf = tang * far_d, wf = hf * aspect, hn = tang * near_d, wn = hn * aspect;
  float fc[3], nc[3], c[8][3];
  for (int i = 0; i < 3; ++i) { fc[i] = t[i] + fwd[i] * far_d; nc[i] = t[i] + fwd[i] * near_d; }
  float *ftl = c[0], *ftr = c[1], *fbl = c[2], *fbr = c[3], *nbr = c[4], *ntl = c[5], *ntr = c[6], *nbl = c[7];
  for (int i = 0; i < 3; ++i) {
    ftl[i] = fc[i] + (up[i] * hf) - (right[i] * wf);
    ftr[i] = fc[i] + (up[i] * hf) + (right[i] * wf);
    fbl[i] = fc[i] - (up[i] * hf) - (right[i] * wf);
    fbr[i] = fc[i] - (up[i] * hf) + (right[i] * wf);
    ntl[i] = nc[i] + (up[i] * hn) - (right[i] * wn);
    ntr[i] = nc[i] + (up[i] * hn) + (right[i] * wn);
    nbl[i] = nc[i] - (up[i] * hn) - (right[i] * wn);
    nbr[i] = nc[i] - (up[i] * hn) + (right[i] * wn);
  }
  auto plane = [&](int k, const float* a, const float* b, const float* cc) {
    float ab[3], ac[3], cr[3];
    for (int i = 0; i < 3; ++i) { ab[i] = b[i] - a[i]; ac[i] = cc[i] - a[i]; }
    cr[0] = ab[1] * ac[2] - ab[2] * ac[1];
    cr[1] = ab[2] * ac[0] - ab[0] * ac[2];
    cr[2] = ab[0] * ac[1] - ab[1] * ac[0];
    const float z = s3(cr[0] * cr[0], cr[1] * cr[1], cr[2] * cr[2]);
    for (int i = 0; i < 3; ++i) C->plane_n[k][i] = (z > 0.0f) ? cr[i] / std::sqrt(z) : cr[i];
    C->plane_d[k] = -s3(cr[0] * a[0], cr[1] * a[1], cr[2] * a[2]);
  };
  plane(0, ftr, ftl, fbr);   // far
  plane(1, nbl, ntl, nbr);   // near
  plane(2, ntl, ftl, ntr);   // top
  plane(3, nbr, fbl, nbl);   // bottom
  plane(4, ftl, ntl, fbl);   // left
  plane(5, ntr, ftr, nbr);   // right
  float lo[3], hi[3];
  for (int i = 0; i < 3; ++i) { lo[i] = 3.402823466e+38f; hi[i] = -3.402823466e+38f; }
  for (int k = 0; k < 8; ++k)
    for (int i = 0; i < 3; ++i) {
      lo[i] = (c[k][i] < lo[i]) ? c[k][i] : lo[i];
      hi[i] = (hi[i] < c[k][i]) ? c[k][i] : hi[i];
    }
  for (int i = 0; i < 3; ++i) {   // GetIDAt(min) - 1 .. GetIDAt(max) + 1 + 1
    C->lo[i] = (int)std::floor(lo[i] * P.rounding) - 1;
    C->hi[i] = (int)std::floor(hi[i] * P.rounding) + 1 + 1;
    for (int j = 0; j < 3; ++j) C->R[3 * i + j] = R[3 * i + j];
    C->t[i] = t[i];
  }
}

extern "C" int plvs_hip_tsdf_chisel_carve_dev(plvs_tsdf_chisel* h, const float* d_depth, int width, int height,
                                              float fx, float fy, float cx, float cy, float near_dist,
                                              float far_dist, const float* Twc, float carving_dist, void* stream,
                                              int* carved_chunks) {
  PLVS_FLUSH_QUEUE(h);
  PLVS_REQUIRE(h && Twc && carved_chunks, "null argument");
  PLVS_REQUIRE(!h->poisoned, "handle is in a failed state (clear it)");
  PLVS_REQUIRE(width > 0 && height > 0, "empty depth image");
  *carved_chunks = 0;
  h->stats = plvs_tsdf_stats{};
  h->last_updated = 0;
  if (h->num_chunks == 0) return PLVS_OK;
  PLVS_REQUIRE(d_depth, "null depth image");
  hipStream_t s = static_cast<hipStream_t>(stream);
  {   // carving changes owned voxels: ghost copies of them held for meshing (here or on peer ranks) are stale now
    int rc = halo_drop(h, s);
    if (rc != PLVS_OK) return rc;
  }
  CarveCamera C;
  carve_frustum(h->P, Twc, near_dist, far_dist, fy, cy, (float)width, (float)height, &C);
  C.fx = fx; C.fy = fy; C.cx = cx; C.cy = cy;
  C.width = (float)width; C.height = (float)height; C.iwidth = width;
  C.carving_dist = carving_dist;
  PLVS_HIP_TRY(h->scratch.reserve((size_t)h->num_chunks));
  PLVS_HIP_TRY(h->updated.reserve((size_t)h->num_chunks + 1));
  PLVS_HIP_TRY(hipMemsetAsync(h->scratch.p, 0, (size_t)h->num_chunks * sizeof(uint32_t), s));
  PLVS_HIP_TRY(hipMemsetAsync(&h->d_ctr->num_updated, 0, sizeof(uint32_t), s));
  hipLaunchKernelGGL(carve_chunks, dim3((unsigned)h->num_chunks * 16u), dim3(256), 0, s, h->P, C, d_depth,
                     h->dir.slot_ids, h->num_chunks, h->sdf, h->weight, h->kfid, h->scratch.p);
  hipLaunchKernelGGL(carve_collect, dim3(ceil_div((size_t)h->num_chunks, 256)), dim3(256), 0, s, h->scratch.p,
                     h->num_chunks, h->updated.p, h->d_ctr);
  PLVS_KERNEL_CHECK();
  int rc = read_counters(h, s);
  if (rc != PLVS_OK) return rc;
  h->last_updated = h->h_ctr->num_updated;   // updated_chunk_ids now lists the carved chunks (meshesToUpdate)
  h->stats.updated_chunks = (int32_t)h->last_updated;
  *carved_chunks = (int)h->last_updated;
  return PLVS_OK;
}

extern "C" int plvs_hip_tsdf_chisel_carve(plvs_tsdf_chisel* h, const float* depth, int width, int height, float fx,
                                          float fy, float cx, float cy, float near_dist, float far_dist,
                                          const float* Twc, float carving_dist, int* carved_chunks) {
  PLVS_REQUIRE(h && depth && width > 0 && height > 0, "bad arguments");
  PLVS_HIP_TRY(h->st_xyz.reserve((size_t)width * height));
  PLVS_HIP_TRY(hipMemcpy(h->st_xyz.p, depth, (size_t)width * height * sizeof(float), hipMemcpyHostToDevice));
  int rc = plvs_hip_tsdf_chisel_carve_dev(h, h->st_xyz.p, width, height, fx, fy, cx, cy, near_dist, far_dist, Twc,
                                          carving_dist, nullptr, carved_chunks);
  if (rc != PLVS_OK) return rc;
  PLVS_HIP_TRY(hipDeviceSynchronize());
  return PLVS_OK;
}

namespace plvs {
namespace tsdf {

bool chisel_map_view(plvs_tsdf_chisel* h, ChiselMapView* v) {
  if (h == nullptr || v == nullptr || h->poisoned) return false;
  v->resolution = h->P.resolution;
  v->dir = h->dir;
  v->sdf = h->sdf;
  v->weight = h->weight;
  v->kfid = h->kfid;
  v->rgbw = h->rgbw;
  v->num_chunks = h->num_chunks;
  v->shard_count = h->P.shard_count;
  v->shard_rank = h->P.shard_rank;
  v->ext = &h->ext;
  v->ext_free = &h->ext_free;
  if (h->P.shard_count > 1) {
    if (h->miss_keys == nullptr) {
      // every chunk of the whole map could be asked for, and many ids that exist nowhere (InterpolateColor's look-ups at
      // voxel indices used as metres reach ~20 chunk widths: thousands of distinct ids per call): at least 2^18 entries
      const size_t want = std::max<size_t>(std::min<size_t>((size_t)h->prm.max_chunks * (size_t)h->P.shard_count, (size_t)1 << 22),
                                           (size_t)1 << 18);
      size_t cap = 1024;
      while (cap < 2 * want) cap <<= 1;
      if (hipMalloc(&h->miss_keys, cap * sizeof(unsigned long long)) != hipSuccess) return false;
      if (hipMalloc(&h->miss_ids, want * 3 * sizeof(int32_t)) != hipSuccess) return false;
      if (hipMalloc(&h->miss_count, sizeof(uint32_t)) != hipSuccess) return false;
      h->miss_cap = (uint32_t)want;
      h->miss_mask = (uint32_t)(cap - 1);
    }
    if (hipMemsetAsync(h->miss_keys, 0xFF, ((size_t)h->miss_mask + 1) * sizeof(unsigned long long), nullptr) != hipSuccess) return false;
    if (hipMemsetAsync(h->miss_count, 0, sizeof(uint32_t), nullptr) != hipSuccess) return false;
    v->ghost = h->gdir;
    v->miss_keys = h->miss_keys;
    v->miss_mask = h->miss_mask;
    v->miss_ids = h->miss_ids;
    v->miss_count = h->miss_count;
    v->miss_cap = h->miss_cap;
  }
  return true;
}

}  // namespace tsdf
}  // namespace plvs

// One chunk id -> its pool slot, created if absent (plvs_hip_tsdf_chisel_upload_chunk).
__global__ void chunk_slot_of(Directory dir, int x, int y, int z, int32_t* __restrict__ num_chunks, uint32_t* __restrict__ err,
                              int32_t* __restrict__ slot_out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *slot_out = dir_find_or_insert(dir, x, y, z, num_chunks, err);
}

// ------------------------------------------------------------------ halo of a sharded map (meshing)
namespace {

constexpr int kHaloWords = 4 * kChunkVox;   // a chunk on the wire: sdf, weight, kfid, rgbw planes

// Which of the requested chunks this rank has.
__global__ void halo_lookup_chunks(Directory dir, const int32_t* __restrict__ ids, int n, uint32_t* __restrict__ found) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) found[i] = dir_find(dir, ids[3 * i], ids[3 * i + 1], ids[3 * i + 2]) >= 0 ? 1u : 0u;
}

// row[i] = number of found chunks before request i (one workgroup; request lists are a few thousand long).
__global__ __launch_bounds__(1024) void halo_rows(const uint32_t* __restrict__ found, int n, uint32_t* __restrict__ row) {
  __shared__ uint32_t s_part[1024];
  const int per = (n + 1023) / 1024;
  const int lo = min((int)threadIdx.x * per, n), hi = min(lo + per, n);
  uint32_t sum = 0;
  for (int i = lo; i < hi; ++i) sum += found[i] ? 1u : 0u;
  s_part[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const uint32_t add = threadIdx.x >= (unsigned)d ? s_part[threadIdx.x - d] : 0u;
    __syncthreads();
    s_part[threadIdx.x] += add;
    __syncthreads();
  }
  uint32_t run = s_part[threadIdx.x] - sum;
  for (int i = lo; i < hi; ++i) {
    row[i] = run;
    run += found[i] ? 1u : 0u;
  }
}

// The found chunks' planes, one payload row (kHaloWords) each, in request order.
__global__ __launch_bounds__(256) void halo_export_chunks(Directory dir, const float* __restrict__ sdf,
                                                          const float* __restrict__ weight, const uint32_t* __restrict__ kfid,
                                                          const uint32_t* __restrict__ rgbw, const int32_t* __restrict__ ids,
                                                          const uint32_t* __restrict__ found, const uint32_t* __restrict__ row,
                                                          uint32_t* __restrict__ payload) {
  const int i = blockIdx.x;
  if (!found[i]) return;
  __shared__ int s_slot;
  if (threadIdx.x == 0) s_slot = dir_find(dir, ids[3 * i], ids[3 * i + 1], ids[3 * i + 2]);
  __syncthreads();
  const int slot = s_slot;
  if (slot < 0) return;
  const size_t src = (size_t)slot * kChunkVox;
  uint4* dst = reinterpret_cast<uint4*>(payload + (size_t)row[i] * kHaloWords);
  const uint4* p0 = reinterpret_cast<const uint4*>(sdf + src);
  const uint4* p1 = reinterpret_cast<const uint4*>(weight + src);
  const uint4* p2 = reinterpret_cast<const uint4*>(kfid + src);
  const uint4* p3 = reinterpret_cast<const uint4*>(rgbw + src);
  for (int v = threadIdx.x; v < kChunkVox / 4; v += 256) {
    dst[v] = p0[v];
    dst[kChunkVox / 4 + v] = p1[v];
    dst[2 * (kChunkVox / 4) + v] = p2[v];
    dst[3 * (kChunkVox / 4) + v] = p3[v];
  }
}

// id -> ghost slot (base + its payload row), or kGhostAbsent for a chunk its owner does not have; an id already
// present keeps its entry.
__global__ void halo_insert(Directory g, const int32_t* __restrict__ ids, const uint32_t* __restrict__ found,
                            const uint32_t* __restrict__ row, int n, int base, uint32_t* __restrict__ err) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = ids[3 * i], y = ids[3 * i + 1], z = ids[3 * i + 2];
  unsigned long long key;
  if (!pack_block(x, y, z, &key)) {
    atomicOr(err, kErrCoordRange);
    return;
  }
  uint32_t hsh = dir_hash(x, y, z, g.mask);
  for (uint32_t probe = 0; probe <= g.mask; ++probe) {
    unsigned long long cur = g.keys[hsh];
    if (cur == key) return;
    if (cur == kEmptyKey) {
      cur = atomicCAS(&g.keys[hsh], kEmptyKey, key);
      if (cur == kEmptyKey) {
        g.slots[hsh] = found[i] ? base + (int)row[i] : plvs::tsdf::kGhostAbsent;
        return;
      }
      if (cur == key) return;
    }
    hsh = (hsh + 1) & g.mask;
  }
  atomicOr(err, kErrPoolFull);
}

__global__ __launch_bounds__(256) void halo_import_chunks(float* __restrict__ sdf, float* __restrict__ weight,
                                                          uint32_t* __restrict__ kfid, uint32_t* __restrict__ rgbw,
                                                          const uint32_t* __restrict__ found, const uint32_t* __restrict__ row,
                                                          const uint32_t* __restrict__ payload, int base) {
  const int i = blockIdx.x;
  if (!found[i]) return;
  const size_t dst = (size_t)(base + (int)row[i]) * kChunkVox;
  const uint4* src = reinterpret_cast<const uint4*>(payload + (size_t)row[i] * kHaloWords);
  uint4* p0 = reinterpret_cast<uint4*>(sdf + dst);
  uint4* p1 = reinterpret_cast<uint4*>(weight + dst);
  uint4* p2 = reinterpret_cast<uint4*>(kfid + dst);
  uint4* p3 = reinterpret_cast<uint4*>(rgbw + dst);
  for (int v = threadIdx.x; v < kChunkVox / 4; v += 256) {
    p0[v] = src[v];
    p1[v] = src[kChunkVox / 4 + v];
    p2[v] = src[2 * (kChunkVox / 4) + v];
    p3[v] = src[3 * (kChunkVox / 4) + v];
  }
}

// Ghost slots back to the state of a never-used pool slot (clear() leaves sdf 99999, everything else 0).
__global__ __launch_bounds__(256) void halo_reset_slots(float* __restrict__ sdf, float* __restrict__ weight,
                                                        uint32_t* __restrict__ kfid, uint32_t* __restrict__ rgbw, int base) {
  const size_t at = (size_t)(base + blockIdx.x) * kChunkVox;
  for (int v = threadIdx.x; v < kChunkVox; v += 256) {
    sdf[at + v] = 99999.0f;
    weight[at + v] = 0.f;
    kfid[at + v] = 0u;
    rgbw[at + v] = 0u;
  }
}

}  // namespace

// Drops the ghosts (the integrate calls allocate new chunks in the slots they occupy).
static int halo_drop(plvs_tsdf_chisel* h, hipStream_t s) {
  if (h->ghost_entries == 0) return PLVS_OK;
  if (h->ghost_count > 0) {
    hipLaunchKernelGGL(halo_reset_slots, dim3((unsigned)h->ghost_count), dim3(256), 0, s, h->sdf, h->weight, h->kfid, h->rgbw,
                       h->num_chunks);
    PLVS_KERNEL_CHECK();
  }
  PLVS_HIP_TRY(hipMemsetAsync(h->gdir.keys, 0xFF, ((size_t)h->gdir.mask + 1) * sizeof(unsigned long long), s));
  h->ghost_count = 0;
  h->ghost_entries = 0;
  return PLVS_OK;
}

extern "C" {

// Creates or REPLACES one chunk with the given voxel planes (host, 4096 each, id = (z * 16 + y) * 16 + x): the way a
// volume saved with download_chunk comes back, and what lets tests put analytic distance fields on the device.
int plvs_hip_tsdf_chisel_upload_chunk(plvs_tsdf_chisel* h, int cx, int cy, int cz, const float* sdf, const float* weight,
                                      const uint32_t* kfid, const uint32_t* rgbw) {
  PLVS_FLUSH_QUEUE(h);
  PLVS_REQUIRE(h && sdf && weight && kfid && rgbw, "null argument");
  PLVS_REQUIRE(!h->poisoned, "handle is in a failed state (clear it)");
  if (std::max(1, h->prm.shard_count) > 1)
    PLVS_REQUIRE(shard_of(chunk_hash(cx, cy, cz), h->prm.shard_count) == h->prm.shard_rank, "the chunk belongs to another rank");
  int rc = halo_drop(h, nullptr);
  if (rc != PLVS_OK) return rc;
  PLVS_HIP_TRY(hipMemset(&h->d_ctr->err, 0, sizeof(uint32_t)));
  int32_t* d_slot = reinterpret_cast<int32_t*>(&h->d_ctr->total_visits);   // (a counter no call is using now)
  hipLaunchKernelGGL(chunk_slot_of, dim3(1), dim3(64), 0, nullptr, h->dir, cx, cy, cz, &h->d_ctr->num_chunks, &h->d_ctr->err,
                     d_slot);
  PLVS_KERNEL_CHECK();
  rc = read_counters(h, nullptr);
  if (rc != PLVS_OK) return rc;
  if (h->h_ctr->err) {
    h->poisoned = true;
    plvs::set_error("upload_chunk: %s", (h->h_ctr->err & kErrPoolFull) ? "chunk pool full (raise max_chunks)" : "chunk id out of range");
    return PLVS_ERR_CAPACITY;
  }
  const int slot = (int)h->h_ctr->total_visits;
  h->num_chunks = h->h_ctr->num_chunks;
  if (h->dfm) deform_note_created(h, cx, cy, cz);
  const size_t off = (size_t)slot * kChunkVox;
  PLVS_HIP_TRY(hipMemcpy(h->sdf + off, sdf, kChunkVox * sizeof(float), hipMemcpyHostToDevice));
  PLVS_HIP_TRY(hipMemcpy(h->weight + off, weight, kChunkVox * sizeof(float), hipMemcpyHostToDevice));
  PLVS_HIP_TRY(hipMemcpy(h->kfid + off, kfid, kChunkVox * sizeof(uint32_t), hipMemcpyHostToDevice));
  PLVS_HIP_TRY(hipMemcpy(h->rgbw + off, rgbw, kChunkVox * sizeof(uint32_t), hipMemcpyHostToDevice));
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_halo_missing(plvs_tsdf_chisel* h, int32_t* ids_xyz, int cap, int* n) {
  PLVS_REQUIRE(h && n, "null argument");
  *n = 0;
  if (h->miss_count == nullptr) return PLVS_OK;
  uint32_t cnt = 0;
  PLVS_HIP_TRY(hipMemcpy(&cnt, h->miss_count, sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (cnt > h->miss_cap) {
    plvs::set_error("halo_missing: %u missing chunks exceed the list capacity %u", cnt, h->miss_cap);
    return PLVS_ERR_CAPACITY;
  }
  *n = (int)cnt;
  if (cnt == 0) return PLVS_OK;
  if ((int)cnt > cap) return PLVS_ERR_CAPACITY;
  PLVS_REQUIRE(ids_xyz, "null output");
  PLVS_HIP_TRY(hipMemcpy(ids_xyz, h->miss_ids, (size_t)cnt * 3 * sizeof(int32_t), hipMemcpyDeviceToHost));
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_halo_lookup(plvs_tsdf_chisel* h, const int32_t* d_ids_xyz, int n, uint32_t* d_found, void* stream) {
  PLVS_REQUIRE(h && !h->poisoned, "unusable handle");
  PLVS_REQUIRE(n >= 0, "negative size");
  if (n == 0) return PLVS_OK;
  PLVS_REQUIRE(d_ids_xyz && d_found, "null argument");
  hipLaunchKernelGGL(halo_lookup_chunks, dim3(ceil_div((size_t)n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), h->dir,
                     d_ids_xyz, n, d_found);
  PLVS_KERNEL_CHECK();
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_halo_export(plvs_tsdf_chisel* h, const int32_t* d_ids_xyz, const uint32_t* d_found, int n,
                                     uint32_t* d_payload, void* stream) {
  PLVS_REQUIRE(h && !h->poisoned, "unusable handle");
  PLVS_REQUIRE(n >= 0, "negative size");
  if (n == 0 || d_payload == nullptr) return PLVS_OK;   // (no payload buffer: the caller saw no flag set)
  PLVS_REQUIRE(d_ids_xyz && d_found, "null argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  PLVS_HIP_TRY(h->halo_row.reserve((size_t)n));
  hipLaunchKernelGGL(halo_rows, dim3(1), dim3(1024), 0, s, d_found, n, h->halo_row.p);
  hipLaunchKernelGGL(halo_export_chunks, dim3((unsigned)n), dim3(256), 0, s, h->dir, h->sdf, h->weight, h->kfid, h->rgbw,
                     d_ids_xyz, d_found, h->halo_row.p, d_payload);
  PLVS_KERNEL_CHECK();
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_halo_import(plvs_tsdf_chisel* h, const int32_t* d_ids_xyz, const uint32_t* d_found,
                                     const uint32_t* d_payload, int n, int nfound, void* stream) {
  PLVS_REQUIRE(h && !h->poisoned, "unusable handle");
  PLVS_REQUIRE(n >= 0 && nfound >= 0 && nfound <= n, "bad sizes");
  if (n == 0) return PLVS_OK;
  PLVS_REQUIRE(d_ids_xyz && d_found && (nfound == 0 || d_payload), "null argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if ((long long)h->num_chunks + h->ghost_count + nfound > (long long)h->prm.max_chunks) {
    plvs::set_error("halo_import: %d own + %d ghost + %d new chunks exceed the pool (%d)", h->num_chunks, h->ghost_count, nfound,
                    h->prm.max_chunks);
    return PLVS_ERR_CAPACITY;
  }
  if (h->gdir.keys == nullptr) {
    // entries are foreign chunks that exist (at most the pool's worth) and ids that exist nowhere (the colour look-up's
    // reach: thousands per call), kept until the next integrate call: twice the miss set's table
    size_t cap = 2 * std::max<size_t>((size_t)h->miss_mask + 1, (size_t)1 << 19);
    while (cap < 4 * (size_t)h->prm.max_chunks) cap <<= 1;
    // (both tables or neither: a half-built ghost directory would be taken for a complete one by the next call)
    unsigned long long* gkeys = nullptr;
    int32_t* gslots = nullptr;
    PLVS_HIP_TRY(hipMalloc(&gkeys, cap * sizeof(unsigned long long)));
    {
      const hipError_t e = hipMalloc(&gslots, cap * sizeof(int32_t));
      if (e != hipSuccess) {
        (void)hipFree(gkeys);
        PLVS_HIP_TRY(e);
      }
    }
    h->gdir.keys = gkeys;
    h->gdir.slots = gslots;
    h->gdir.slot_ids = nullptr;
    h->gdir.mask = (uint32_t)(cap - 1);
    h->gdir.max_blocks = h->prm.max_chunks;
    PLVS_HIP_TRY(hipMemsetAsync(h->gdir.keys, 0xFF, cap * sizeof(unsigned long long), s));
  }
  if ((size_t)h->ghost_entries + (size_t)n > ((size_t)h->gdir.mask + 1) / 2) {
    plvs::set_error("halo_import: %lld + %d entries exceed the ghost directory (halo_clear drops them)", h->ghost_entries, n);
    return PLVS_ERR_CAPACITY;
  }
  const int base = h->num_chunks + h->ghost_count;
  PLVS_HIP_TRY(h->halo_row.reserve((size_t)n));
  hipLaunchKernelGGL(halo_rows, dim3(1), dim3(1024), 0, s, d_found, n, h->halo_row.p);
  hipLaunchKernelGGL(halo_insert, dim3(ceil_div((size_t)n, 256)), dim3(256), 0, s, h->gdir, d_ids_xyz, d_found, h->halo_row.p, n,
                     base, &h->d_ctr->err);
  if (nfound > 0)
    hipLaunchKernelGGL(halo_import_chunks, dim3((unsigned)n), dim3(256), 0, s, h->sdf, h->weight, h->kfid, h->rgbw, d_found,
                       h->halo_row.p, d_payload, base);
  PLVS_KERNEL_CHECK();
  h->ghost_count += nfound;
  h->ghost_entries += n;
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_halo_clear(plvs_tsdf_chisel* h) {
  PLVS_REQUIRE(h, "null handle");
  int rc = halo_drop(h, nullptr);
  if (rc != PLVS_OK) return rc;
  PLVS_HIP_TRY(hipStreamSynchronize(nullptr));
  return PLVS_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ ray-sharded integrate (tsdf_shard.hpp)
static int shard_state_init(plvs_tsdf_chisel* h) {
  if (h->xdir.keys) return PLVS_OK;
  // the walk directory: every chunk of the whole map may pass through it (ids + 512 B of bits each)
  const size_t xmax = std::min<size_t>((size_t)h->prm.max_chunks * (size_t)std::max(1, h->prm.shard_count), (size_t)1 << 22);
  size_t cap = 1024;
  while (cap < 2 * xmax) cap <<= 1;
  h->xdir.mask = (uint32_t)(cap - 1);
  h->xdir.max_blocks = (int32_t)xmax;
  PLVS_HIP_TRY(hipMalloc((void**)&h->xdir.keys, cap * sizeof(unsigned long long)));
  PLVS_HIP_TRY(hipMalloc((void**)&h->xdir.slots, cap * sizeof(int32_t)));
  PLVS_HIP_TRY(hipMalloc((void**)&h->xdir.slot_ids, xmax * 3 * sizeof(int32_t)));
  PLVS_HIP_TRY(hipMalloc((void**)&h->x_sat, xmax * (kChunkVox / 32) * sizeof(uint32_t)));
  PLVS_HIP_TRY(hipMalloc((void**)&h->d_xcount, 4 * sizeof(int32_t)));   // [0] chunks, [1] error bits, [2] saturated this call
  PLVS_HIP_TRY(hipHostMalloc((void**)&h->h_sh_counts, ((size_t)3 * 64 + 2) * sizeof(long long)));
  PLVS_HIP_TRY(hipHostMalloc((void**)&h->h_sh_ctl, 320 * sizeof(uint32_t)));
  PLVS_HIP_TRY(hipHostMalloc((void**)&h->h_sh_off, (128 + 132) * sizeof(uint32_t)));
  PLVS_HIP_TRY(hipMemset(h->xdir.keys, 0xFF, cap * sizeof(unsigned long long)));
  PLVS_HIP_TRY(hipMemset(h->xdir.slots, 0xFF, cap * sizeof(int32_t)));
  PLVS_HIP_TRY(hipMemset(h->x_sat, 0, xmax * (kChunkVox / 32) * sizeof(uint32_t)));
  PLVS_HIP_TRY(hipMemset(h->d_xcount, 0, 4 * sizeof(int32_t)));
  return PLVS_OK;
}

static int shard_state_clear(plvs_tsdf_chisel* h) {
  if (!h->xdir.keys) return PLVS_OK;
  const size_t cap = (size_t)h->xdir.mask + 1;
  PLVS_HIP_TRY(hipMemset(h->xdir.keys, 0xFF, cap * sizeof(unsigned long long)));
  PLVS_HIP_TRY(hipMemset(h->xdir.slots, 0xFF, cap * sizeof(int32_t)));
  PLVS_HIP_TRY(hipMemset(h->x_sat, 0, (size_t)h->xdir.max_blocks * (kChunkVox / 32) * sizeof(uint32_t)));
  PLVS_HIP_TRY(hipMemset(h->d_xcount, 0, 4 * sizeof(int32_t)));
  h->sh_phase = 0;
  h->sh_nsat = 0;
  h->sh_wait_first = h->sh_wait_count = 0;
  return PLVS_OK;
}

extern "C" {

int plvs_hip_tsdf_chisel_shard_walk(plvs_tsdf_chisel* h, const float* d_xyz, const int32_t* offsets, int nclouds,
                                    const float* d_Twc, int64_t* send_counts, void* stream) {
  PLVS_REQUIRE(h && send_counts, "null argument");
  PLVS_REQUIRE(!h->poisoned, "handle is in a failed state (clear it)");
  PLVS_REQUIRE(h->prm.order_free != 0 && h->prm.shard_count >= 1 && h->prm.shard_count <= 64,
               "the ray-sharded integrate needs order_free = 1 and 1 <= shard_count <= 64");
  PLVS_REQUIRE(offsets && nclouds >= 0, "bad offsets");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int N = std::max(1, h->prm.shard_count), rank = N > 1 ? h->prm.shard_rank : 0;
  for (int p = 0; p < 3 * N; ++p) send_counts[p] = 0;
  h->sh_stats = plvs_tsdf_stats{};
  h->sh_phase = 0;
  const int n = nclouds > 0 ? offsets[nclouds] - offsets[0] : 0;
  PLVS_REQUIRE(nclouds == 0 || (offsets[0] == 0 && n >= 0), "offsets must start at 0 and be non-decreasing");
  for (int c = 0; c < nclouds; ++c) PLVS_REQUIRE(offsets[c + 1] >= offsets[c], "offsets must be non-decreasing");
  int rc = shard_state_init(h);
  if (rc != PLVS_OK) return rc;
  for (int p = 0; p < 3 * N; ++p) h->h_sh_counts[p] = 0;
  h->sh_n = n;
  h->sh_nclouds = nclouds;
  h->sh_tiletab.resize(2 * ((size_t)nclouds + 1));
  h->sh_ntiles = plvs::tsdf::fill_tile_table(offsets, nclouds, h->sh_tiletab.data(), kWalkRays);   // (tiles never straddle clouds)
  if (h->sh_ntiles >= (1u << kWireTileBits)) {
    plvs::set_error("tsdf_chisel shard_walk: %u tiles in one call exceed the wire format's tile index (split the batch)", h->sh_ntiles);
    return PLVS_ERR_CAPACITY;
  }
  h->sh_nt = 0;
  h->sh_runs = 0;
  h->sh_phase = 1;
  {   // the points of this rank's tiles
    int64_t own = 0;
    for (int c = 0; c < nclouds; ++c) {
      const uint32_t t0 = (uint32_t)h->sh_tiletab[(size_t)nclouds + 1 + c], t1 = (uint32_t)h->sh_tiletab[(size_t)nclouds + 2 + c];
      for (uint32_t t = t0; t < t1; ++t)
        if (t % (uint32_t)N == (uint32_t)rank)
          own += std::min<int64_t>(kWalkRays, (int64_t)(offsets[c + 1] - offsets[c]) - (int64_t)(t - t0) * kWalkRays);
    }
    h->sh_stats.points = own;
  }
  const uint32_t nt = h->sh_ntiles > (uint32_t)rank ? (h->sh_ntiles - (uint32_t)rank + (uint32_t)N - 1u) / (uint32_t)N : 0u;
  // (the tile table goes to the device even on a rank without tiles: the runs other ranks send it name tiles of the stream)
  PLVS_HIP_TRY(h->offsets.reserve(2 * ((size_t)nclouds + 1)));
  PLVS_HIP_TRY(hipMemcpyAsync(h->offsets.p, h->sh_tiletab.data(), 2 * ((size_t)nclouds + 1) * sizeof(int32_t),
                              hipMemcpyHostToDevice, s));
  if (nt == 0) return PLVS_OK;
  PLVS_REQUIRE(d_xyz && d_Twc, "null device pointer");
  h->sh_nt = nt;
  const size_t xmax = (size_t)h->xdir.max_blocks;
  PLVS_HIP_TRY(h->poses.reserve((size_t)nclouds));
  hipLaunchKernelGGL(pose_prep, dim3(ceil_div((size_t)nclouds, 64)), dim3(64), 0, s, d_Twc, nclouds, h->poses.p);
  PLVS_HIP_TRY(h->w_chunk_nseg.reserve(xmax));
  PLVS_HIP_TRY(h->w_chunk_off.reserve(xmax + 1));
  PLVS_HIP_TRY(h->w_chunk_fill.reserve(xmax));
  PLVS_HIP_TRY(h->w_active_off.reserve(xmax + 1));
  PLVS_HIP_TRY(h->updated.reserve(xmax + 1));
  PLVS_HIP_TRY(h->w_seg_cnt.reserve(nt));
  PLVS_HIP_TRY(h->w_tile_visits.reserve(nt));
  PLVS_HIP_TRY(h->w_deferred.reserve(nt));
  PLVS_HIP_TRY(h->w_run_cnt.reserve(nt));
  PLVS_HIP_TRY(h->w_run_off.reserve((size_t)nt + 1));
  PLVS_HIP_TRY(h->w_part_off.reserve(xmax + 1));
  PLVS_HIP_TRY(h->w_multi_idx.reserve(xmax + 1));
  PLVS_HIP_TRY(h->sh_nrec.reserve(xmax));
  PLVS_HIP_TRY(h->sh_owner.reserve(xmax));
  PLVS_HIP_TRY(h->sh_slot_owner.reserve(xmax));
  PLVS_HIP_TRY(h->sh_run_ctr.reserve((size_t)3 * 64));   // counts, bases, fill cursors per destination
  PLVS_HIP_TRY(h->sh_ctl.reserve(256 + 2));              // regions, fill cursors, region totals
  {
    int rc2 = ensure_part_acc(h, (uint32_t)std::min<size_t>(xmax, 64));
    if (rc2 != PLVS_OK) return rc2;
  }
  const size_t rec_own = (size_t)nt * kWalkLimit, seg_own = (size_t)nt * kWalkChunks;
  size_t rec_spill = std::max<size_t>(h->w_rec.cap > rec_own ? h->w_rec.cap - rec_own : 0, (size_t)1 << 16);
  size_t seg_spill = std::max<size_t>(h->w_seg.cap / 2 > seg_own ? h->w_seg.cap / 2 - seg_own : 0, (size_t)1 << 12);
  Params Pw = h->P;       // this rank walks its tiles through every chunk they cross
  Pw.shard_count = 1;
  Pw.shard_rank = 0;
  const TileMap tmap{(uint32_t)N, (uint32_t)rank};
  for (int attempt = 0;; ++attempt) {
    // (sized by the call's tiles, and the largest buffers of the handle — 64 B of masks per run slot, gigabytes: a stream of
    // calls of varying length would re-allocate them whenever a call is a little longer than any before, so they grow to
    // TWICE what a call needs, as in integrate_walk_acc)
    if (h->w_rec.cap < rec_own + rec_spill) PLVS_HIP_TRY(h->w_rec.reserve(2 * rec_own + rec_spill));
    if (h->w_seg.cap < 2 * (seg_own + seg_spill)) PLVS_HIP_TRY(h->w_seg.reserve(2 * (2 * seg_own + seg_spill)));
    PLVS_HIP_TRY(h->w_sorted_seg.reserve(h->w_seg.cap));
    const size_t run_slots = (size_t)nt << h->run_r1_log2;
    if (h->w_runkey.cap < run_slots) PLVS_HIP_TRY(h->w_runkey.reserve(2 * run_slots));
    if (h->w_masks.cap < run_slots * kMaskWords) PLVS_HIP_TRY(h->w_masks.reserve(2 * run_slots * kMaskWords));
    if (h->dkey0.cap < run_slots) PLVS_HIP_TRY(h->dkey0.reserve(2 * run_slots));   // (all a call's runs, whatever their number)
    if (h->sh_run_first.cap < run_slots) PLVS_HIP_TRY(h->sh_run_first.reserve(2 * run_slots));   // (first wire record per run)
    if (h->w_val0.cap < run_slots) PLVS_HIP_TRY(h->w_val0.reserve(2 * run_slots));
    PLVS_HIP_TRY(h->scratch.reserve(scan_scratch_words(nt)));
    PLVS_HIP_TRY(hipMemsetAsync(h->d_wctr, 0, 2 * sizeof(WalkCounters), s));
    PLVS_HIP_TRY(hipMemsetAsync(h->w_chunk_nseg.p, 0, xmax * sizeof(uint32_t), s));
    PLVS_HIP_TRY(hipMemsetAsync(h->sh_run_ctr.p, 0, 3 * 64 * sizeof(uint32_t), s));
    AccOut out{h->w_rec.p, (uint32_t)std::min<size_t>(rec_own + rec_spill, 0xFFFFFFFFu), h->w_seg.p,
               (uint32_t)std::min<size_t>(seg_own + seg_spill, 0xFFFFFFFFu), h->w_seg_cnt.p, h->w_tile_visits.p, nullptr};
    RunOut runs{h->w_runkey.p, h->w_masks.p, h->w_run_cnt.p, h->run_r1_log2};
    // (a chunk entered by an attempt that has to be repeated stays in the walk directory: harmless)
    hipLaunchKernelGGL(walk_fast<kFastEntries>, dim3(nt), dim3(kWalkRays), 0, s, Pw, h->scale_u, h->scale_w, d_xyz, n,
                       h->offsets.p, nclouds, h->poses.p, h->xdir, h->d_xcount, h->d_wctr, (const uint32_t*)nullptr,
                       (const uint32_t*)h->x_sat, out, runs, tmap, (uint32_t)kWalkLimit, (const uint32_t*)nullptr,
                       (const uint32_t*)nullptr, h->w_deferred.p, &h->d_wctr->ndeferred, (const GridSrc*)nullptr);
    hipLaunchKernelGGL((walk_tiles<true, true>), dim3(kDeferGrid), dim3(kWalkRays), 0, s, Pw, h->scale_u, h->scale_w, d_xyz, n,
                       h->offsets.p, nclouds, h->poses.p, h->xdir, h->d_xcount, h->d_wctr, (const uint32_t*)nullptr,
                       (const uint32_t*)h->x_sat, out, runs, tmap, (uint32_t)nt, (const uint32_t*)h->w_deferred.p,
                       (const uint32_t*)&h->d_wctr->ndeferred, (uint32_t)kWalkLimit,
                       1u, (const GridSrc*)nullptr);   // (flagged = overflowed 2048 entries: this kernel's table takes 3584, the tile goes whole)
    const unsigned seg_blocks = ceil_div(seg_own + seg_spill, kSegSpan);
    hipLaunchKernelGGL(seg_pass<false>, dim3(seg_blocks), dim3(256), 0, s, h->w_seg.p, out.seg_cap, nt, h->w_seg_cnt.p,
                       h->w_chunk_nseg.p, h->w_chunk_off.p, h->w_chunk_fill.p, h->w_sorted_seg.p, h->d_wctr);
    hipLaunchKernelGGL(seg_scan, dim3(1), dim3(1024), 0, s, h->w_chunk_nseg.p, h->w_chunk_off.p, h->w_chunk_fill.p,
                       h->updated.p, h->w_active_off.p, h->d_wctr, h->d_xcount, (int)xmax, h->w_tile_visits.p,
                       h->w_run_cnt.p, nt, h->w_part_off.p, h->w_multi_idx.p, h->multi_cap, h->part_segs, h->part_min);
    hipLaunchKernelGGL(seg_pass<true>, dim3(seg_blocks), dim3(256), 0, s, h->w_seg.p, out.seg_cap, nt, h->w_seg_cnt.p,
                       h->w_chunk_nseg.p, h->w_chunk_off.p, h->w_chunk_fill.p, h->w_sorted_seg.p, h->d_wctr);
    hipLaunchKernelGGL(shard_chunk_totals, dim3(1024), dim3(256), 0, s, h->w_sorted_seg.p, h->updated.p,
                       h->w_active_off.p, h->xdir.slot_ids, N, h->d_wctr, h->sh_nrec.p, h->sh_owner.p, h->sh_slot_owner.p);
    hipLaunchKernelGGL(shard_plan, dim3(1), dim3(1024), 0, s, h->sh_nrec.p, h->sh_owner.p, N, h->d_wctr, h->sh_ctl.p,
                       h->sh_ctl.p + 256);
    // the runs, densely, in tile order (seg_scan has left their number in num_desc), counted per destination
    PLVS_HIP_TRY(exclusive_scan_u32(h->w_run_cnt.p, h->w_run_off.p, nt, nullptr, h->scratch.p, s));
    hipLaunchKernelGGL(compact_runs, dim3(ceil_div(nt, 4)), dim3(256), 0, s, h->w_runkey.p, h->w_run_cnt.p,
                       h->w_run_off.p, nt, h->run_r1_log2, h->dkey0.p, h->w_val0.p,
                       RunGuard{0xFFFFFFFFu, nullptr, nullptr, 0, nullptr, nullptr, 0u, nullptr, 0u});
    hipLaunchKernelGGL(shard_run_count, dim3(2048), dim3(256), 0, s, h->dkey0.p, h->w_val0.p, &h->d_wctr[0].num_desc,
                       h->w_masks.p, h->sh_slot_owner.p, N, h->sh_run_ctr.p, h->sh_run_first.p, h->d_wctr);
    hipLaunchKernelGGL(shard_run_plan, dim3(1), dim3(64), 0, s, h->sh_run_ctr.p, N, h->d_wctr);
    PLVS_KERNEL_CHECK();
    // sizes of the send regions (and whether the walk has to be repeated)
    uint32_t* const h_plan = reinterpret_cast<uint32_t*>(h->h_sh_counts + 3 * 64);
    hipLaunchKernelGGL(publish_words, dim3(1), dim3(64), 0, s, (const uint32_t*)(h->sh_ctl.p + 256), h_plan, 2,
                       reinterpret_cast<const uint32_t*>(h->d_wctr), reinterpret_cast<uint32_t*>(h->h_wctr),
                       (int)(sizeof(WalkCounters) / sizeof(uint32_t)), (const uint32_t*)nullptr, (uint32_t*)nullptr, 0);
    PLVS_KERNEL_CHECK();
    PLVS_HIP_TRY(hipStreamSynchronize(s));
    const uint32_t err = h->h_wctr->err;
    if (err & kErrPoolFull) {
      h->poisoned = true;
      plvs::set_error("tsdf_chisel shard_walk: the walk directory is full (max_chunks x shard_count chunks)");
      return PLVS_ERR_CAPACITY;
    }
    if (err & ~kErrScratch) return walk_fail(h, err);
    if (err & kErrScratch) {
      if (attempt >= 8) return walk_fail(h, err);
      if (getenv("PLVS_DEBUG_SHARD"))   // (a repeated walk doubles the step: which scratch region was short)
        fprintf(stderr, "shard_walk repeats: %u tiles, records %u of %zu spill, segments %u of %zu spill, runs per tile %u of %u\n",
                nt, h->h_wctr->rec_top, rec_spill, h->h_wctr->seg_top, seg_spill, h->h_wctr->run_need, 1u << h->run_r1_log2);
      rec_spill = std::max<size_t>(rec_spill, (size_t)h->h_wctr->rec_top * 2);
      seg_spill = std::max<size_t>(seg_spill, (size_t)h->h_wctr->seg_top * 2);
      while ((1u << h->run_r1_log2) < h->h_wctr->run_need) ++h->run_r1_log2;
      if (((size_t)nt << h->run_r1_log2) >= 0xFFFFFFFFull) return walk_fail(h, err);
      continue;
    }
    // ---- this rank's own aggregation: one sum per touched voxel into the owner's send region
    PLVS_HIP_TRY(h->sh_seg_reg.reserve(2 * (size_t)h_plan[0] + 2));
    PLVS_HIP_TRY(h->sh_rec_reg.reserve(2 * (size_t)h_plan[1] + 2));
    uint32_t* const ctl = h->sh_ctl.p;
    hipLaunchKernelGGL((apply_chunks<false, true>), dim3(4096), dim3(kApplyThreads), 0, s, h->w_sorted_seg.p, h->updated.p,
                       h->w_active_off.p, h->w_part_off.p, h->w_multi_idx.p, h->part_segs,
                       PartAcc{h->pa_wuu.p, h->pa_w.p, h->pa_last.p, h->pa_cnt.p, h->pa_done.p}, h->w_rec.p, 0.0, 0.0,
                       (const uint32_t*)nullptr, (float*)nullptr, (float*)nullptr, (uint32_t*)nullptr, h->d_wctr,
                       EmitOut{h->xdir.slot_ids, h->sh_owner.p, ctl, ctl + 64, ctl + 128, ctl + 192, h->sh_seg_reg.p,
                               h->sh_rec_reg.p}, 0u);
    PLVS_KERNEL_CHECK();
    hipLaunchKernelGGL(publish_words, dim3(1), dim3(256), 0, s, (const uint32_t*)h->sh_ctl.p, h->h_sh_ctl, 256,
                       (const uint32_t*)h->sh_run_ctr.p, h->h_sh_ctl + 256, 64, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0);
    PLVS_KERNEL_CHECK();
    PLVS_HIP_TRY(hipStreamSynchronize(s));
    break;
  }
  for (int p = 0; p < N; ++p) {
    h->h_sh_counts[3 * p] = (long long)h->h_sh_ctl[128 + p];
    h->h_sh_counts[3 * p + 1] = (long long)h->h_sh_ctl[192 + p];
    h->h_sh_counts[3 * p + 2] = (long long)h->h_sh_ctl[256 + p];
  }
  if (h->h_wctr->num_multi > h->multi_cap) {
    rc = ensure_part_acc(h, (uint32_t)std::min<size_t>(xmax, (size_t)h->h_wctr->num_multi + h->h_wctr->num_multi / 2));
    if (rc != PLVS_OK) return rc;
  }
  h->sh_runs = h->h_wctr->num_desc;
  for (int p = 0; p < 3 * N; ++p) send_counts[p] = (int64_t)h->h_sh_counts[p];
  h->sh_stats.visits = (int64_t)h->h_wctr->total_visits;
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_shard_pack(plvs_tsdf_chisel* h, void* d_seg_dst, void* d_rec_dst, void* d_run_dst, void* stream) {
  PLVS_REQUIRE(h, "null handle");
  PLVS_REQUIRE(h->sh_phase == 1, "shard_pack follows shard_walk");
  hipStream_t s = static_cast<hipStream_t>(stream);
  h->sh_phase = 2;
  const int N = std::max(1, h->prm.shard_count);
  long long nseg = 0;
  for (int p = 0; p < N; ++p) nseg += h->h_sh_counts ? h->h_sh_counts[3 * p] : 0;
  if (nseg == 0) return PLVS_OK;
  PLVS_REQUIRE(d_seg_dst && d_rec_dst && (h->sh_runs == 0 || d_run_dst), "null send buffer");
  // (pinned staging: the copy is asynchronous and its source outlives this call; the previous step's copies have
  // executed — shard_apply ends with a synchronisation)
  uint32_t* const dst_off = h->h_sh_off;
  for (int p = 0; p < 128; ++p) dst_off[p] = 0;
  for (int p = 1; p < N; ++p) {
    dst_off[p] = dst_off[p - 1] + (uint32_t)h->h_sh_counts[3 * (p - 1)];
    dst_off[64 + p] = dst_off[64 + p - 1] + (uint32_t)h->h_sh_counts[3 * (p - 1) + 1];
  }
  PLVS_HIP_TRY(h->sh_src_off.reserve(128 + 132));
  PLVS_HIP_TRY(hipMemcpyAsync(h->sh_src_off.p, dst_off, 128 * sizeof(uint32_t), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(shard_copy_regions, dim3(512), dim3(256), 0, s, h->sh_seg_reg.p, h->sh_rec_reg.p, h->sh_ctl.p,
                     h->sh_src_off.p, N, static_cast<uint4*>(d_seg_dst), static_cast<uint4*>(d_rec_dst));
  if (h->sh_runs > 0)
    hipLaunchKernelGGL(shard_run_pack, dim3(std::min<size_t>(ceil_div((size_t)h->sh_runs, kRunSpan), 4096)), dim3(256), 0, s,
                       h->dkey0.p, h->w_val0.p, &h->d_wctr[0].num_desc, h->w_masks.p, h->sh_run_first.p, h->run_r1_log2,
                       TileMap{(uint32_t)std::max(1, h->prm.shard_count), h->prm.shard_count > 1 ? (uint32_t)h->prm.shard_rank : 0u},
                       h->xdir.slot_ids,
                       h->sh_slot_owner.p, h->sh_run_ctr.p + 64, h->sh_run_ctr.p + 128, static_cast<uint32_t*>(d_run_dst));
  PLVS_KERNEL_CHECK();
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_shard_apply(plvs_tsdf_chisel* h, const void* d_seg_src, const void* d_rec_src,
                                     const void* d_run_src, const int64_t* recv_counts, const uint8_t* d_rgb,
                                     const uint32_t* d_kfid, void* stream) {
  PLVS_REQUIRE(h && recv_counts, "null argument");
  PLVS_REQUIRE(!h->poisoned, "handle is in a failed state (clear it)");
  PLVS_REQUIRE(h->sh_phase == 2, "shard_apply follows shard_pack");
  hipStream_t s = static_cast<hipStream_t>(stream);
  {
    int rc = halo_drop(h, s);   // first-touch chunks go into the pool slots a meshing halo may still occupy
    if (rc != PLVS_OK) return rc;
  }
  h->sh_phase = 0;
  h->sh_nsat = 0;
  const int N = std::max(1, h->prm.shard_count);
  const int max_chunks = h->prm.max_chunks;
  PLVS_REQUIRE(h->h_sh_off != nullptr, "shard_apply follows shard_walk");
  uint32_t* const src_off = h->h_sh_off + 128;   // pinned, 2 (N + 1) <= 130 words
  size_t tseg = 0, trec = 0, trun = 0;
  for (int q = 0; q < N; ++q) {
    PLVS_REQUIRE(recv_counts[3 * q] >= 0 && recv_counts[3 * q + 1] >= 0 && recv_counts[3 * q + 2] >= 0, "negative receive count");
    src_off[q] = (uint32_t)tseg;
    src_off[N + 1 + q] = (uint32_t)trec;
    tseg += (size_t)recv_counts[3 * q];
    trec += (size_t)recv_counts[3 * q + 1];
    trun += (size_t)recv_counts[3 * q + 2];
  }
  src_off[N] = (uint32_t)tseg;
  src_off[2 * N + 1] = (uint32_t)trec;
  PLVS_REQUIRE(tseg < 0x7FFFFFFFull && trec < 0xFFFFFFFFull && trun < 0x7FFFFFFFull,
               "receive buffers beyond the index range (split the batch)");
  h->stats = h->sh_stats;
  h->last_updated = 0;
  h->stage_set = 1;
  if (tseg == 0) return PLVS_OK;
  PLVS_REQUIRE(d_seg_src && d_rec_src && d_rgb && (trun == 0 || d_run_src), "null device pointer");
  const uint32_t total = (uint32_t)tseg;
  PLVS_HIP_TRY(h->sh_src_off.reserve(128 + 132));
  PLVS_HIP_TRY(h->w_seg.reserve(2 * (size_t)total));
  PLVS_HIP_TRY(h->w_sorted_seg.reserve(2 * (size_t)total));
  PLVS_HIP_TRY(h->w_chunk_nseg.reserve((size_t)max_chunks));
  PLVS_HIP_TRY(h->w_chunk_off.reserve((size_t)max_chunks + 1));
  PLVS_HIP_TRY(h->w_chunk_fill.reserve((size_t)max_chunks));
  PLVS_HIP_TRY(h->w_active_off.reserve((size_t)max_chunks + 1));
  PLVS_HIP_TRY(h->updated.reserve((size_t)max_chunks + 1));
  PLVS_HIP_TRY(h->w_part_off.reserve((size_t)max_chunks + 1));
  PLVS_HIP_TRY(h->w_multi_idx.reserve((size_t)max_chunks + 1));
  {
    // every chunk applied in parts has more than kPartMin segments: the received total bounds their number
    int rc = ensure_part_acc(h, std::min<uint32_t>((uint32_t)max_chunks, total / std::max(1u, h->part_min) + 1u));
    if (rc != PLVS_OK) return rc;
  }
  const int chunks_before = h->num_chunks;
  uint32_t* const d_src_off = h->sh_src_off.p + 128;   // (apart from the words shard_pack's kernels may still be reading)
  PLVS_HIP_TRY(hipMemcpyAsync(d_src_off, src_off, 2 * ((size_t)N + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(shard_apply_begin, dim3(ceil_div((size_t)max_chunks, 1024)), dim3(256), 0, s, h->d_ctr, h->d_wctr,
                     h->d_xcount + 2, h->w_chunk_nseg.p, (uint32_t)max_chunks, total, (uint32_t)trun);
#define STAGE_MARK(i) \
  do { if (h->profiling) PLVS_HIP_TRY(hipEventRecord(h->ev[i], s)); } while (0)
  STAGE_MARK(0);
  STAGE_MARK(1);   // (the walk ran in shard_walk: its stage time stays 0 here)
  hipLaunchKernelGGL(shard_translate, dim3(ceil_div((size_t)total, 256)), dim3(256), 0, s,
                     static_cast<const uint4*>(d_seg_src), total, d_src_off, N, h->dir, &h->d_ctr->num_chunks,
                     &h->d_wctr[0].err, h->w_seg.p);
  const unsigned seg_blocks = ceil_div((size_t)total, kSegSpan);
  hipLaunchKernelGGL(seg_pass<false>, dim3(seg_blocks), dim3(256), 0, s, h->w_seg.p, total, 0u, (const uint32_t*)nullptr,
                     h->w_chunk_nseg.p, h->w_chunk_off.p, h->w_chunk_fill.p, h->w_sorted_seg.p, h->d_wctr);
  hipLaunchKernelGGL(seg_scan, dim3(1), dim3(1024), 0, s, h->w_chunk_nseg.p, h->w_chunk_off.p, h->w_chunk_fill.p,
                     h->updated.p, h->w_active_off.p, h->d_wctr, &h->d_ctr->num_chunks, max_chunks,
                     (const uint32_t*)nullptr, (const uint32_t*)nullptr, 0u, h->w_part_off.p, h->w_multi_idx.p, h->multi_cap,
                     h->part_segs, h->part_min);
  hipLaunchKernelGGL(seg_pass<true>, dim3(seg_blocks), dim3(256), 0, s, h->w_seg.p, total, 0u, (const uint32_t*)nullptr,
                     h->w_chunk_nseg.p, h->w_chunk_off.p, h->w_chunk_fill.p, h->w_sorted_seg.p, h->d_wctr);
  STAGE_MARK(2);
  hipLaunchKernelGGL((apply_chunks<true, false>), dim3(4096), dim3(kApplyThreads), 0, s, h->w_sorted_seg.p, h->updated.p,
                     h->w_active_off.p, h->w_part_off.p, h->w_multi_idx.p, h->part_segs,
                     PartAcc{h->pa_wuu.p, h->pa_w.p, h->pa_last.p, h->pa_cnt.p, h->pa_done.p},
                     static_cast<const uint4*>(d_rec_src), 1.0 / (double)h->scale_u, 1.0 / (double)h->scale_w, d_kfid,
                     h->sdf, h->weight, h->kfid, h->d_wctr, EmitOut{}, 0u);
  PLVS_KERNEL_CHECK();
  STAGE_MARK(3);
  // ---- colours: the received runs, by (voxel, tile)
  if (trun > 0) {
    const uint32_t R = (uint32_t)trun;
    const uint32_t* runs = static_cast<const uint32_t*>(d_run_src);
    PLVS_HIP_TRY(h->dkey0.reserve(R));
    PLVS_HIP_TRY(h->dkey1.reserve(R));
    PLVS_HIP_TRY(h->w_val0.reserve(R));
    PLVS_HIP_TRY(h->w_val1.reserve(R));
    PLVS_HIP_TRY(h->sh_vkey.reserve(R));
    PLVS_HIP_TRY(h->heads.reserve(R));
    PLVS_HIP_TRY(h->sh_sat.reserve(R));
    PLVS_HIP_TRY(h->w_dummy.reserve((size_t)max_chunks + 1));
    PLVS_HIP_TRY(h->scratch.reserve(radix_scratch_words(R)));
    uint32_t* const err = &h->d_wctr[0].err;
    hipLaunchKernelGGL(shard_run_translate, dim3(ceil_div((size_t)R, 256)), dim3(256), 0, s, runs, R, h->dir, err,
                       h->sh_vkey.p, h->dkey0.p, h->w_val0.p);
    int tile_bits = 1;
    while ((1ull << tile_bits) < (unsigned long long)h->sh_ntiles) ++tile_bits;
    bool second = false;
    PLVS_HIP_TRY(radix_sort_pairs(h->dkey0.p, h->w_val0.p, h->dkey1.p, h->w_val1.p, R, 0, tile_bits, h->scratch.p, s, &second));
    uint32_t* order = second ? h->w_val1.p : h->w_val0.p;
    uint32_t* other = second ? h->w_val0.p : h->w_val1.p;
    uint32_t* k_in = second ? h->dkey0.p : h->dkey1.p;   // the key buffer the tile sort has left free
    uint32_t* k_out = second ? h->dkey1.p : h->dkey0.p;
    hipLaunchKernelGGL(shard_gather_keys, dim3(ceil_div((size_t)R, 256)), dim3(256), 0, s, h->sh_vkey.p, order, R, k_in);
    // (a received descriptor can add one chunk at most: the chunks before the call + the descriptors bound the slots)
    const long long slot_bound = std::min<long long>(max_chunks, (long long)chunks_before + (long long)tseg);
    int key_bits = 12;
    while ((1ll << (key_bits - 12)) < slot_bound) ++key_bits;
    PLVS_HIP_TRY(radix_sort_pairs(k_in, order, k_out, other, R, 0, key_bits, h->scratch.p, s, &second));
    const uint32_t* skeys = second ? k_out : k_in;
    const uint32_t* sval = second ? other : order;
    hipLaunchKernelGGL(voxel_heads, dim3(ceil_div(R, 256 * kHeadTiles)), dim3(256), 0, s, skeys, R, h->heads.p,
                       h->w_dummy.p, h->d_wctr + 1);
    hipLaunchKernelGGL(fold_colours_masks<false>, dim3(std::min<size_t>(ceil_div(R, kFoldWaves), 8192)), dim3(64 * kFoldWaves), 0, s,
                       skeys, sval, &h->d_wctr[1].num_desc,
                       RunSrc{runs, kWireRun, 0u, TileMap{1u, 0u}, h->offsets.p, h->sh_nclouds, nullptr}, h->heads.p, d_rgb,
                       h->rgbw, &h->d_wctr[1].num_heads, h->sh_sat.p, reinterpret_cast<uint32_t*>(h->d_xcount + 2),
                       (const uint32_t*)nullptr, GridSrc{});
    PLVS_KERNEL_CHECK();
  }
  STAGE_MARK(4);
  hipLaunchKernelGGL(publish_words, dim3(1), dim3(64), 0, s, reinterpret_cast<const uint32_t*>(h->d_xcount + 2),
                     reinterpret_cast<uint32_t*>(h->h_sh_counts), 1, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0,
                     (const uint32_t*)nullptr, (uint32_t*)nullptr, 0);
  int rc = read_walk_counters(h, s);
  if (rc != PLVS_OK) return rc;
  if (h->h_wctr->err) return walk_fail(h, h->h_wctr->err);
  h->sh_nsat = (uint32_t)(*reinterpret_cast<int32_t*>(h->h_sh_counts));
  h->num_chunks = h->h_ctr->num_chunks;
  h->stats.new_chunks = h->num_chunks - chunks_before;
  h->stats.updated_chunks = (int32_t)h->h_wctr->num_updated;
  h->stats.voxels = (int32_t)h->h_wctr->num_heads;
  h->stats.max_run = (int32_t)h->h_wctr->max_run;
  h->last_updated = h->h_wctr->num_updated;
  if (h->profiling) {
    for (int i = 0; i < 4; ++i) {
      float ms = 0.f;
      PLVS_HIP_TRY(stage_elapsed(&ms, h->ev[i], h->ev[i + 1]));
      h->stage_ms[i] += ms;
    }
    h->prof_calls++;
  }
#undef STAGE_MARK
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_shard_saturated(plvs_tsdf_chisel* h, int32_t* d_voxels, int cap, int* n, void* stream) {
  PLVS_REQUIRE(h && n, "null argument");
  *n = (int)h->sh_nsat;
  if (h->sh_nsat == 0) return PLVS_OK;
  if (cap < (int)h->sh_nsat) {
    plvs::set_error("shard_saturated: %u voxels, room for %d", h->sh_nsat, cap);
    return PLVS_ERR_CAPACITY;
  }
  PLVS_REQUIRE(d_voxels, "null output");
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(shard_saturated_ids, dim3(ceil_div((size_t)h->sh_nsat, 256)), dim3(256), 0, s, h->sh_sat.p, h->sh_nsat,
                     h->dir.slot_ids, d_voxels);
  PLVS_KERNEL_CHECK();
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_shard_saturated_message(plvs_tsdf_chisel* h, int32_t* d_msg, int rows, void* stream) {
  PLVS_REQUIRE(h && d_msg && rows > 0, "bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (h->sh_nsat > 0) {   // the last shard_apply's voxels join the waiting list
    const size_t need = 4 * ((size_t)h->sh_wait_first + h->sh_wait_count + h->sh_nsat);
    if (need > h->sh_wait.cap) {
      DevBuf<int32_t> grown;
      PLVS_HIP_TRY(grown.reserve(need));
      if (h->sh_wait_count)
        PLVS_HIP_TRY(hipMemcpyAsync(grown.p, h->sh_wait.p + 4 * (size_t)h->sh_wait_first, 16 * (size_t)h->sh_wait_count,
                                    hipMemcpyDeviceToDevice, s));
      PLVS_HIP_TRY(hipStreamSynchronize(s));
      h->sh_wait.release();
      h->sh_wait = grown;
      h->sh_wait_first = 0;
    }
    hipLaunchKernelGGL(shard_saturated_ids, dim3(ceil_div((size_t)h->sh_nsat, 256)), dim3(256), 0, s, h->sh_sat.p, h->sh_nsat,
                       h->dir.slot_ids, h->sh_wait.p + 4 * ((size_t)h->sh_wait_first + h->sh_wait_count));
    h->sh_wait_count += h->sh_nsat;
    h->sh_nsat = 0;
  }
  const uint32_t k = std::min<uint32_t>(h->sh_wait_count, (uint32_t)rows);
  hipLaunchKernelGGL(shard_sat_message, dim3(std::max<unsigned>(1u, ceil_div((size_t)k, 256))), dim3(256), 0, s,
                     h->sh_wait.p ? h->sh_wait.p + 4 * (size_t)h->sh_wait_first : (const int32_t*)nullptr, k, (uint32_t)rows, d_msg);
  PLVS_KERNEL_CHECK();
  h->sh_wait_first += k;
  h->sh_wait_count -= k;
  if (h->sh_wait_count == 0) h->sh_wait_first = 0;
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_shard_note_gathered(plvs_tsdf_chisel* h, const int32_t* d_gathered, int nranks, int rows, void* stream) {
  PLVS_REQUIRE(h && d_gathered && nranks >= 1 && rows > 0, "bad argument");
  PLVS_REQUIRE(h->prm.order_free != 0 && h->prm.shard_count >= 1, "not a ray-sharded map");
  int rc = shard_state_init(h);
  if (rc != PLVS_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(shard_note_gathered, dim3(ceil_div((size_t)rows, 256), (unsigned)nranks), dim3(256), 0, s, d_gathered,
                     (uint32_t)rows, h->xdir, h->d_xcount, reinterpret_cast<uint32_t*>(h->d_xcount + 1), h->x_sat);
  PLVS_KERNEL_CHECK();
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_shard_note_saturated(plvs_tsdf_chisel* h, const int32_t* d_voxels, int n, void* stream) {
  PLVS_REQUIRE(h && n >= 0, "bad argument");
  PLVS_REQUIRE(h->prm.order_free != 0 && h->prm.shard_count >= 1, "not a ray-sharded map");
  if (n == 0) return PLVS_OK;
  PLVS_REQUIRE(d_voxels, "null list");
  int rc = shard_state_init(h);
  if (rc != PLVS_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(shard_note_saturated, dim3(ceil_div((size_t)n, 256)), dim3(256), 0, s, d_voxels, (uint32_t)n, h->xdir,
                     h->d_xcount, reinterpret_cast<uint32_t*>(h->d_xcount + 1), h->x_sat);
  PLVS_KERNEL_CHECK();
  return PLVS_OK;
}

}  // extern "C"

#include "tsdf_chisel_deform.hpp"
