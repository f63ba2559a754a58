// The single-walk integrate of the open_chisel back end: the order-free mode
// (plvs_tsdf_chisel_params.order_free = 1; included by tsdf_chisel.hip).
//
// One workgroup = one tile of kWalkRays consecutive points.  Every thread walks the Amanatides-Woo ray
// of its point ONCE (tsdf_chisel_core.hpp: the reference's arithmetic, bit for bit) and the visits of the
// tile meet in an LDS hash table keyed by the voxel coordinates:
//
//   walk_tiles  an entry accumulates sum(w_u * u), sum(w_u), the visit count and the last visiting ray of
//               its voxel — in FIXED POINT, so the sums do not depend on the order in which the lanes
//               arrive (deterministic, order-free).  No global memory is touched inside the ray loop.
//               At the end of the tile the entries are resolved to pool slots (first-touch chunks are
//               inserted here), grouped by (chunk, slab) and written as 16-byte records into the tile's
//               own record region; one segment descriptor per (tile, chunk).
//               The u8 colour mean of the reference truncates after every visit and freezes at weight
//               254: it stays EXACT.  Every ray logs the entries it visited (LDS); for the entries whose
//               voxel is still below 254 the log is turned into the BIT MASK of the tile's rays that
//               visit the voxel — a ray visits a voxel at most once, so ray order = point order = the
//               reference's update order — and one run (voxel key + mask) leaves the tile.
//   seg_pass / seg_scan           counting sort of the segment descriptors by chunk (LDS-aggregated).
//   apply_chunks                  "LDS-staged blocks": a workgroup owns a slab of one chunk, adds the
//               slab's records into 64-bit LDS accumulators and applies ONE update per voxel,
//                   sdf <- (W * sdf + sum w_u u) / (W + sum w_u),   W <- W + sum w_u,
//               kfid <- kfid of the last visiting point.
//   compact_runs, stable radix sort by voxel key, voxel_heads, fold_colours_masks
//               ColorVoxel::IntegrateSimple visit by visit through the sorted runs of the voxels below
//               weight 254 (a wave per voxel; nothing once a map has saturated).
//
// A tile whose visits do not fit the table is cut in halves (ray ranges), a single ray that does not fit
// is cut into windows of visits; both are rare and only cost a re-walk.  Nothing in the pipeline queues
// on one global counter: a returning device-scope atomic on one word costs ~11 ns, and 15 000 tiles
// taking a ticket, a record range and a run range each used to cost more than the walk itself.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "tsdf_chisel_core.hpp"
#include "tsdf_directory.hpp"
#include "tsdf_tiles.hpp"

namespace {

using namespace plvs::chisel;
using plvs::tsdf::cloud_of;
using plvs::tsdf::tile_span;
using plvs::tsdf::tile_span_tables;
using plvs::tsdf::TileSpan;
using plvs::tsdf::kCoordBias;
using plvs::tsdf::kErrCoordRange;
using plvs::tsdf::kErrDirectoryMiss;
using plvs::tsdf::kErrPoolFull;

constexpr int kWalkRays = 512;                  // rays (= threads) per tile
constexpr int kMaskWords = kWalkRays / 32;      // a run's ray mask
// LDS hash table entries of a tile.  2048 since round 4: at 5 cm the 512 rays of a tile whose points lie 3 - 5 m away
// touch 800 - 2000 distinct voxels (the rays of neighbouring pixels stop sharing voxels once the pixel footprint nears
// the voxel size); with 1024 entries half the tiles of a real office scene overflowed and took the slow general path
// (rounds 1-3 only ever measured a 6 x 4 m room, depths below 3 m: 300 - 600 voxels per tile).
// (this constant: the table of the GENERAL kernel, walk_tiles — one tile per CU; the lean kernel walk_fast<E> takes its
// table size as a template parameter: 2048 entries at two tiles per CU for the bulk of a long call, 4096 for the tiles
// that overflow that)
#ifndef PLVS_WALK_ENTRIES
#define PLVS_WALK_ENTRIES 4096
#endif
constexpr int kWalkEntries = PLVS_WALK_ENTRIES;
constexpr int kWalkLimit = kWalkEntries * 7 / 8;   // entries a (sub-)tile may use (buckets of four: probes stay short)
constexpr int kWalkWindow = kWalkLimit / 2;     // visits per window of a single over-long ray
constexpr int kWalkChunks = 64;                 // per-tile chunk cache
constexpr uint32_t kErrScratch = 8u;            // record / segment / run buffers too small: the host grows them and retries

// Phase clocks of walk_tiles (a developer build: make PROF=1 -> libplvs_hip_prof.so; thread 0 of a tile adds the shader
// cycles between the tile's barriers to g_walk_prof[phase]).  Compiled out of the product library.
#ifndef PLVS_WALK_SORT
#define PLVS_WALK_SORT 0    // walk_fast: 1 = the rays of a tile dealt to the waves by depth — parity holds, measured SLOWER
                            // (walk 0.857 -> 0.880 ms on the stream, 0.512 -> 0.550 in the room: three barriers, scattered point
                            // loads and neighbouring lanes no longer sharing table buckets cost more than the 14-19 % of loop
                            // iterations the homogeneous waves save); kept as a switch
#endif
#ifndef PLVS_WALK_EXP
#define PLVS_WALK_EXP 0     // timing experiments (developer builds; results are wrong with any bit set)
#endif
#ifdef PLVS_WALK_PROF
__device__ unsigned long long g_walk_prof[16];
__device__ unsigned long long g_apply_items[8192][4];   // per item of the LAST apply launch: cycles, segments << 32 | records of thread 0's group
#define WALK_PROF_BEGIN()                                         \
  long long prof_t = (long long)clock64();                        \
  unsigned long long prof_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}
#define WALK_PROF(k)                                               \
  do {                                                             \
    const long long now = (long long)clock64();                    \
    prof_acc[k] += (unsigned long long)(now - prof_t);             \
    prof_t = now;                                                  \
  } while (0)
#define WALK_PROF_END()                                            \
  do {                                                             \
    if (threadIdx.x == 0)                                          \
      for (int k = 0; k < 9; ++k) atomicAdd(&g_walk_prof[k], prof_acc[k]); \
  } while (0)
#else
#define WALK_PROF_BEGIN() do {} while (0)
#define WALK_PROF(k) do {} while (0)
#define WALK_PROF_END() do {} while (0)
#endif

struct WalkCounters {           // device-side, read back once per call
  unsigned long long total_visits;
  uint32_t err;
  uint32_t num_heads;           // voxels updated by the call
  uint32_t num_updated;         // chunks updated by the call
  uint32_t max_run;             // most visits of one voxel in the call
  uint32_t num_desc;            // runs of the call (sum of the per-tile counts)
  uint32_t run_need;            // most run slots a tile asked for (when they did not fit)
  uint32_t rec_top, seg_top;    // records / segments written (walk_acc)
  uint32_t ncold;               // tiles that met a voxel with colour weight < 254
  uint32_t split_tiles;         // tiles that had to be cut (table overflow)
  uint32_t num_parts;           // apply stage: sum over the updated chunks of their parts
  uint32_t num_multi;           //   chunks applied in more than one part
  uint32_t ndeferred;           // tiles walk_fast left to the next kernel (several clouds in the tile, table overflow)
  uint32_t ndeferred2;          //   and what the larger-table pass over that list left to walk_tiles
  uint32_t over_small;          // tiles of a 2048-entry first pass that a 1024-entry table would not have held
  uint32_t skip;                // colour side: a chain launched on predicted sizes found them too small (compact_runs)
  uint32_t collect_top;         // colour side: runs given a place by runs_rowscan (the regions of the (chunk, slab) rows)
  uint32_t collect_parts;       //   and the parts of kCollectPart runs these regions are sorted in
  uint32_t ndeferred3;          // what a third pass (4096 entries, behind a 1024- and a 2048-entry one) left to walk_tiles
};

// ------------------------------------------------------------------ the walk of one (sub-)tile
constexpr int kLogLen = 16;                      // visits per ray kept in the tile's visit log
constexpr int kSlabs = 8;                        // a chunk is applied in slabs of kSlabVox voxels
constexpr int kSlabVox = kChunkVox / kSlabs;

struct SubTile {          // rays [lo, hi) of the tile — all of one cloud —, visits [vlo, vhi) of each ray
  uint16_t lo, hi;
  uint32_t vlo, vhi;
  int32_t cloud;
};

// The voxel table of a (sub-)tile: kWalkEntries 32-bit keys in buckets of four (one ds_read_b128 per probe).
// A key = the voxel's coordinates relative to the (sub-)tile's origin, 10 bits per axis: the rays of a tile
// — consecutive points — stay within +-512 voxels of the first ray's start (a sub-tile whose rays do not is
// cut like one whose voxels do not fit the table; a single ray sets its origin at the first visit of its window).
constexpr uint32_t kKeyEmpty = 0xFFFFFFFFu;
constexpr int kBuckets = kWalkEntries / 4;
constexpr int kOriginBias = 512;

constexpr int log2_of(int v) { return v <= 1 ? 0 : 1 + log2_of(v / 2); }
struct WalkShared {       // LDS state of walk_tiles
  static constexpr int kEntries = kWalkEntries, kBucketCount = kWalkEntries / 4, kBucketShift = 32 - log2_of(kWalkEntries / 4);
  alignas(16) uint32_t ekey[kWalkEntries];
  int32_t org[3];                              // origin of the keys
  uint32_t cand[kWalkRays / 64];               // does the wave have a walking ray ...
  int32_t worg[kWalkRays / 64][3];             // ... and the origin its first one proposes
  uint32_t run_total, vis_total;               // runs emitted / visits walked by the tile so far
  uint32_t ccode[kWalkChunks];                 // chunk cache: chunk code (the chunk bits of a voxel key),
  int32_t cslot[kWalkChunks];                  //   pool slot,
  uint32_t ccnt[kWalkChunks * kSlabs];         //   entries of the (sub-)tile per (chunk, slab),
  uint16_t cbase[kWalkChunks * kSlabs];        //   first record of the group, relative to the flush's first record
  uint32_t rbase;                              // first record of the flush
  SubTile stack[16];   // depth: up to four pieces at once, 8 halvings of the ray range, the windows of one ray
  int sp;
  uint32_t nent;          // entries in use
  uint32_t overflow;      // the table is full: the (sub-)tile is cut
  uint32_t wsum[kWalkRays / 64];
  uint32_t next, nrays;   // rays of the tile not yet handed out as sub-tiles / rays of the tile
};
// LDS state of walk_fast<E>: WalkShared without the sub-tile stack, the table size a template parameter (E = 2048: two
// tiles per CU, the kernel of a long call; E = 4096: one tile per CU — the tiles that overflowed 2048 entries, and
// every tile of a call too short to fill the device, where the occupancy is immaterial and a deferral is not)
template <int E>
struct FastShared {
  static constexpr int kEntries = E, kBucketCount = E / 4, kBucketShift = 32 - log2_of(E / 4);
  static_assert((E & (E - 1)) == 0 && E >= 1024, "power-of-two table");
  alignas(16) uint32_t ekey[E];
  uint32_t cand[kWalkRays / 64];
  int32_t worg[kWalkRays / 64][3];
  uint32_t run_total, vis_total;
  uint32_t ccode[kWalkChunks];
  int32_t cslot[kWalkChunks];
  uint32_t ccnt[kWalkChunks * kSlabs];
  uint16_t cbase[kWalkChunks * kSlabs];
  uint32_t rcnt[kWalkChunks * kSlabs];    // runs of the (chunk, slab) group
  uint16_t rcbase[kWalkChunks * kSlabs];  // the group's first run among the tile's (every wave writes the same table)
  uint32_t any_cold;                      // a voxel of the tile needs a run
  uint32_t nent, overflow;
};


// Key of a voxel relative to the origin; false if it lies outside the 1024^3 box of the keys.
__device__ __forceinline__ bool rel_key(int vx, int vy, int vz, int ox, int oy, int oz, uint32_t* key) {
  const uint32_t dx = (uint32_t)(vx - ox), dy = (uint32_t)(vy - oy), dz = (uint32_t)(vz - oz);
  *key = dx | (dy << 10) | (dz << 20);
  return ((dx | dy | dz) >> 10) == 0u;
}
// The origin is a multiple of 16 on every axis (origin_of), so a key carries its chunk and its voxel id in fixed bit
// fields: bits 4..9 of each 10-bit coordinate = the chunk relative to the origin's chunk (the CHUNK CODE: the key with
// the voxel bits masked away — unique per chunk of the tile), bits 0..3 = the voxel inside the chunk.
__device__ __forceinline__ int origin_of(int v) { return (v - kOriginBias) & ~15; }
constexpr uint32_t kChunkCodeMask = 0x3F0FC3F0u;
__device__ __forceinline__ uint32_t chunk_code(uint32_t key) { return key & kChunkCodeMask; }
__device__ __forceinline__ uint32_t voxel_in_chunk(uint32_t key) {   // (z * 16 + y) * 16 + x  (Chunk.h:90-93)
  return (key & 15u) | ((key >> 6) & 0xF0u) | ((key >> 12) & 0xF00u);
}
__device__ __forceinline__ void chunk_of_code(uint32_t code, int ox, int oy, int oz, int* cx, int* cy, int* cz) {
  *cx = (ox >> 4) + (int)((code >> 4) & 63u);
  *cy = (oy >> 4) + (int)((code >> 14) & 63u);
  *cz = (oz >> 4) + (int)((code >> 24) & 63u);
}
// The table stores a voxel key MULTIPLIED by an odd constant ("table key", a bijection of the 32-bit words): its top
// eight bits are the home bucket, and since the walk moves the key by one of three constants per voxel step the
// table key moves by one of three constants too — the voxel loop carries the table key alone and never multiplies
// (v_mul_lo_u32 issues at a quarter of the rate of an add).  No voxel key maps to kKeyEmpty (0xF174D0AF would: keys
// have 30 bits).
constexpr uint32_t kKeyMul = 2654435761u, kKeyMulInv = 0x0E8B2F51u;
static_assert((uint32_t)(kKeyMul * kKeyMulInv) == 1u, "inverse of the table-key multiplier");
__device__ __forceinline__ uint32_t table_key(uint32_t key) { return key * kKeyMul; }
__device__ __forceinline__ uint32_t voxel_key(uint32_t tkey) { return tkey * kKeyMulInv; }
template <class SH>
__device__ __forceinline__ uint32_t key_bucket(uint32_t tkey) { return tkey >> SH::kBucketShift; }

// Inclusive prefix sum over the 64 lanes of a wave by DPP (row shifts inside the rows of 16, then the row broadcasts):
// six VALU instructions, no LDS traffic (a __shfl_up is a ds_bpermute_b32 through the LDS crossbar).
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x) {
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);   // row_shr:1
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);   // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);   // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);   // row_shr:8
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
  return x;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t x) {   // (a scalar: the last lane of the scan)
  return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(x), 63);
}

// position of `want` among the four keys of a bucket, -1 if absent.  All four comparisons feed the result, so the
// bucket is ONE ds_read_b128 (a short-circuit form makes the compiler read the first key alone and branch).
__device__ __forceinline__ int bucket_match(const uint4 k4, uint32_t want) {
  const bool c0 = k4.x == want, c1 = k4.y == want, c2 = k4.z == want, c3 = k4.w == want;
  const int j = c3 ? 3 : (c2 ? 2 : (c1 ? 1 : 0));
  return (c0 | c1 | c2 | c3) ? j : -1;
}

// Entry of the key in the table (inserted if absent); -1 when the table is full.  A key lives in the first
// bucket from its home bucket on that had a free slot when it came (entries are never removed), so a search
// ends at the first bucket that holds the key or a free slot.  ONE LDS round trip for a key that is there, two for
// a first touch (the bucket, the compare-and-swap): every wave step of the walk has a few lanes on this path and
// the step is as slow as its slowest lane.  Entries are not counted here — the flush counts them (entries beyond
// kWalkLimit: the (sub-)tile is cut, as when the table is full).
constexpr int kProbeCap = 24;
template <class SH>
__device__ __forceinline__ int table_find_or_insert(SH& S, uint32_t key /* a table key */) {
  uint32_t b = key_bucket<SH>(key);
  // (kProbeCap: a tile within its limit — three quarters of the table — never chains that far; a table on its way to
  // full does, and every probe is an LDS round trip: without the cap the rays of an overflowing tile spent hundreds of
  // probes each before they gave up, ~1 ms per tile on the office stream of bench.py)
  for (int probe = 0; probe < kProbeCap; ++probe) {
    const uint4 k4 = *reinterpret_cast<const uint4*>(&S.ekey[4 * b]);
    const int j = bucket_match(k4, key);
    if (j >= 0) return (int)(4 * b) + j;
    const int je = bucket_match(k4, kKeyEmpty);
    if (je >= 0) {
      const uint32_t old = atomicCAS(&S.ekey[4 * b + je], kKeyEmpty, key);
      if (old == kKeyEmpty || old == key) return (int)(4 * b) + je;
      continue;   // another voxel took the slot: look at this bucket again
    }
    b = (b + 1) & (SH::kBucketCount - 1);
  }
  S.overflow = 1u;
  return -1;
}
// Entry of a (table) key that is known to be in the table.
template <class SH>
__device__ __forceinline__ int table_find(const SH& S, uint32_t key) {
  uint32_t b = key_bucket<SH>(key);
  for (int probe = 0; probe < SH::kBucketCount; ++probe) {
    const int j = bucket_match(*reinterpret_cast<const uint4*>(&S.ekey[4 * b]), key);
    if (j >= 0) return (int)(4 * b) + j;
    b = (b + 1) & (SH::kBucketCount - 1);
  }
  return -1;
}

// sqrtf and the IEEE quotient without the range scaffolding the compiler wraps around them (denormal scaling,
// div_scale / div_fmas / div_fixup, class checks): 9 + 6 instructions instead of 16 + 12, for operands in the
// normal range — squared lengths and depths of metres here.  Same results bit for bit:
//   sqrt   v_sqrt_f32 (<= 1 ulp) corrected by the residuals at s - 1 ulp and s + 1 ulp — the core of the
//          compiler's own correctly rounded sequence;
//   a / b  y = RN(1 / b) (v_rcp_f32 + one Newton step: correctly rounded for every binary32 significand on
//          gfx950, plvs_hip_selftest_rcp), q = RN(a y), r = a - q b (exact in one fma), RN(q + r y) = RN(a / b)
//          (Markstein's correction step, as in dist_update_rcp).
// plvs_hip_selftest_walk_math compares both against sqrtf and `/` on the device over the operand ranges of the walk.
__device__ __forceinline__ float sqrt_rn_normal(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
  const float rm = fmaf(-sm, s, x), rp = fmaf(-sp, s, x);
  float r = (rm <= 0.0f) ? sm : s;
  r = (rp > 0.0f) ? sp : r;
  return r;
}
__device__ __forceinline__ float div_rn_normal(float a, float b) {
  const float y0 = __builtin_amdgcn_rcpf(b);
  const float y = fmaf(fmaf(-b, y0, 1.0f), y0, y0);
  const float q = a * y;
  return fmaf(fmaf(-q, b, a), y, q);
}
// signed_dist (tsdf_chisel_core.hpp) with the two above; operands outside their range (a voxel centre within
// 2^-40 m of the camera plane or centre) take the plain forms.
__device__ __forceinline__ float signed_dist_fast(const Pose& pose, float depth, float c0, float c1, float c2) {
  float cc[3];
  xform(pose.Ri, pose.ti, c0, c1, c2, cc);
  const float n2 = sqnorm3(cc[0], cc[1], cc[2]);
  const bool plain = !(n2 >= 0x1p-80f && n2 <= 0x1p80f) || !(fabsf(cc[2]) >= 0x1p-40f && fabsf(cc[2]) <= 0x1p40f);
#if (PLVS_WALK_EXP & 4)   // (timing experiment: raw v_sqrt / v_rcp, no correction steps)
  return __builtin_amdgcn_sqrtf(n2) * (depth * __builtin_amdgcn_rcpf(cc[2]) - 1);
#endif
  if (__builtin_expect(plain, 0)) return sqrtf(n2) * (depth / cc[2] - 1);
  return sqrt_rn_normal(n2) * (div_rn_normal(depth, cc[2]) - 1);
}
// ------------------------------------------------------------------ two-tier u and the lean walk
// TWO-TIER u.  The reference's u = |c_c| (z / c_c.z - 1) needs a correctly rounded square root and quotient and the
// reference's own rounding sequence of the camera transform only where it DECIDES something, i.e. where |u| is within
// the error bound of the truncation distance.  Everywhere else (all but ~1 voxel step in 10^3) the voxel takes u from
// 9 fused multiply-adds (voxel index -> camera frame in one affine map), v_sqrt_f32 and v_rcp_f32 raw:
//     |u_lean - u_reference| <= band   (lean_ray: 2^-19 of the coordinate magnitudes involved times the squared
//                                       obliquity of the ray, ~3x the bound derived in DESIGN.md §4.1;
//                                       plvs_hip_selftest_walk_lean measures the actual maximum on the device),
// so |u_lean| < tau - band  =>  inside,  |u_lean| >= tau + band  =>  outside, and in between the exact form decides.
// The order-free sums are toleranced (fixed point, 2^-22 m), the membership is not: it stays the reference's.
// EVERY walk of the order-free mode takes u this way (the lean walk below and the general walk_one), so a voxel's
// contribution does not depend on which of the two walked its ray.
__device__ __forceinline__ float uniform_f(float x) {   // a wave-uniform value -> a scalar register
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)));
}
struct LeanRay {            // voxel index -> camera frame: c_c = A v + b (A = Ri * resolution, b = ti + Ri * half voxel)
  float A[9], b[3];
  float tau_in, tau_out;    // truncation distance -/+ band
};
__device__ __forceinline__ void lean_ray(const Params& P, const Pose& pose, const Ray& ray, LeanRay* L) {
  // (the pose is uniform over the sub-tile: the map lives in scalar registers — a VALU instruction takes one of them
  // as an operand for free, and twelve vector registers stay available to the loop)
#pragma unroll
  for (int k = 0; k < 9; ++k) L->A[k] = uniform_f(pose.Ri[k] * P.resolution);
#pragma unroll
  for (int k = 0; k < 3; ++k)
    L->b[k] = uniform_f(fmaf(pose.Ri[3 * k], P.half_voxel, fmaf(pose.Ri[3 * k + 1], P.half_voxel, fmaf(pose.Ri[3 * k + 2], P.half_voxel, pose.ti[k]))));
  // the band: coordinate magnitudes the two forms round at (camera position, the ray's ends, its depth) and the
  // squared obliquity |p|^2 / p.z^2 of the ray (du / dc_c grows with it); all from the ray itself, so that every
  // walk of the same ray derives the same band
  const float mx = (ray.start[0] + ray.end[0]) * (0.5f * P.resolution) - pose.t[0];
  const float my = (ray.start[1] + ray.end[1]) * (0.5f * P.resolution) - pose.t[1];
  const float mz = (ray.start[2] + ray.end[2]) * (0.5f * P.resolution) - pose.t[2];
  const float obl2 = fmaxf(sqnorm3(mx, my, mz) / (ray.depth * ray.depth), 1.0f);
  float ext = 0.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k) ext = fmaxf(ext, fmaxf(fabsf(ray.start[k]), fabsf(ray.end[k])));
  const float m = fabsf(pose.ti[0]) + fabsf(pose.ti[1]) + fabsf(pose.ti[2]) + 3.0f * (ext + 4.0f) * P.resolution + ray.depth + 1.0f;
  const float band = m * obl2 * 0x1p-19f;
  L->tau_in = ray.truncation - band;
  L->tau_out = ray.truncation + band;
}
// u of voxel (fx, fy, fz) (floats holding integers) by the lean form
__device__ __forceinline__ float lean_u(const LeanRay& L, float depth, float fx, float fy, float fz) {
  const float c0 = fmaf(L.A[0], fx, fmaf(L.A[1], fy, fmaf(L.A[2], fz, L.b[0])));
  const float c1 = fmaf(L.A[3], fx, fmaf(L.A[4], fy, fmaf(L.A[5], fz, L.b[1])));
  const float c2 = fmaf(L.A[6], fx, fmaf(L.A[7], fy, fmaf(L.A[8], fz, L.b[2])));
  const float n2 = fmaf(c0, c0, fmaf(c1, c1, c2 * c2));
  return __builtin_amdgcn_sqrtf(n2) * ((depth - c2) * __builtin_amdgcn_rcpf(c2));
}
// Chisel.cpp:525-531: does voxel (fx, fy, fz) take the update?  *u = its signed distance (lean or exact, see above).
__device__ __forceinline__ bool two_tier_visit(const Params& P, const Pose& pose, const Ray& ray, const LeanRay& L, float fx,
                                               float fy, float fz, float* u) {
  *u = lean_u(L, ray.depth, fx, fy, fz);
  const float au = fabsf(*u);
  bool in = au < L.tau_in;
  if (__builtin_expect(!in && au < L.tau_out, 0)) {   // too close to call: the reference's own arithmetic
    const float c0 = fx * P.resolution + P.half_voxel, c1 = fy * P.resolution + P.half_voxel,
                c2 = fz * P.resolution + P.half_voxel;
    *u = signed_dist_fast(pose, ray.depth, c0, c1, c2);
    in = fabsf(*u) < ray.truncation;
  }
  return in;
}

// Walks the ray inside the window [vlo, vhi) of its visits; on_visit(k, vx, vy, vz, u), k = index of
// the visit inside the window, returns false to abandon the walk (table overflow).  Returns the number
// of accepted visits seen up to the point where the walk stopped (at most vhi).  The general walk: windows of a
// ray that does not fit the table, the owner filter of a sharded handle, the re-walk behind a full visit log.
template <class OnVisit>
__device__ __forceinline__ uint32_t walk_one(const Params& P, const Pose& pose, const Ray& ray, uint32_t vlo,
                                             uint32_t vhi, OnVisit&& on_visit) {
  RayCursor cur;
  OwnerCache owner;
  LeanRay L;
  lean_ray(P, pose, ray, &L);
  ray_begin(ray, &cur);
  int vx, vy, vz;
  uint32_t nv = 0;
  bool go = true;
  while (go && nv < vhi && ray_next(&cur, &vx, &vy, &vz)) {
    float u = 0.0f;
    const bool ok = chunk_owned(P, vx >> 4, vy >> 4, vz >> 4, &owner) &&
                    two_tier_visit(P, pose, ray, L, (float)vx, (float)vy, (float)vz, &u);
    if (ok && nv >= vlo) go = on_visit(nv - vlo, vx, vy, vz, u);
    nv += ok ? 1u : 0u;
  }
  return nv;
}

// The lean walk (the common case of walk_tiles: whole rays of a sub-tile of several rays, no owner filter).  The same
// voxels in the same order as walk_one — with ~half the instructions per voxel step: the position is carried as three
// floats (exact integers) and as the packed table key itself (one add per step: +-1, +-1 << 10 or +-1 << 20), so a
// step has no int -> float conversions, no key packing and no range test — the ray's whole box (start voxel +- its
// length) is checked against the key box once, before the walk (lean_fits).  The DDA state (tMax, tDelta, the distance
// test against maxDist, the tie rules) is the reference's, bit for bit.
__device__ __forceinline__ int lean_reach(const Ray& ray) {   // how far (in voxels, per axis) a walk can get from its start voxel
  const float maxDist = sqnorm3(ray.end[0] - ray.start[0], ray.end[1] - ray.start[1], ray.end[2] - ray.start[2]);
  return (int)(__builtin_amdgcn_sqrtf(maxDist)) + 3;
}
template <bool kAcc, bool kRuns, class SH>
__device__ __forceinline__ uint32_t walk_lean(const Params& P, const Pose& pose, const Ray& ray, SH& S, int ox, int oy,
                                              int oz, int tid, float wu_scaled, uint32_t q_w, int32_t* e_wuu,
                                              unsigned long long* e_wc, uint32_t* e_last, uint16_t* vlog, uint32_t rid) {
  RayCursor cur;
  ray_begin(ray, &cur);
  if (cur.done) return 0u;   // start voxel == end voxel: Raycast.cpp emits nothing
  LeanRay L;
  lean_ray(P, pose, ray, &L);
  // (table keys: see key_bucket)
  uint32_t key = table_key((uint32_t)(cur.x - ox) | ((uint32_t)(cur.y - oy) << 10) | ((uint32_t)(cur.z - oz) << 20));
  const uint32_t endkey = table_key((uint32_t)(cur.endX - ox) | ((uint32_t)(cur.endY - oy) << 10) | ((uint32_t)(cur.endZ - oz) << 20));
  const uint32_t kx = table_key((uint32_t)cur.stepX), ky = table_key((uint32_t)cur.stepY << 10), kz = table_key((uint32_t)cur.stepZ << 20);
  float fx = (float)cur.x, fy = (float)cur.y, fz = (float)cur.z;
  const float sfx = (float)cur.stepX, sfy = (float)cur.stepY, sfz = (float)cur.stepZ;
  float tmx = cur.tMaxX, tmy = cur.tMaxY, tmz = cur.tMaxZ;
  // an axis that never steps has tDelta = 0 / 0; it is never added in the reference (its tMax is +inf), here it is
  // multiplied by a zero mask: keep it finite
  const float tdx = cur.stepX ? cur.tDeltaX : 0.0f, tdy = cur.stepY ? cur.tDeltaY : 0.0f, tdz = cur.stepZ ? cur.tDeltaZ : 0.0f;
  // the visits so far, carried as the BYTE OFFSET of the ray's next slot in the visit log (visit j of ray r at
  // [j * kWalkRays + r], 16-bit entries): one register for the count and the address instead of two
  static_assert(kWalkRays * sizeof(uint16_t) == 1024, "the count is the offset >> 10");
  uint32_t vp = (uint32_t)tid * (uint32_t)sizeof(uint16_t);
  for (int guard = 0; guard < kRayStepGuard; ++guard) {   // (the reference loop is unbounded)
    // ---- Raycast.cpp:115-129 at the current voxel
    const float d = sqnorm3(fx - cur.sx, fy - cur.sy, fz - cur.sz);
    const bool stop = (d > cur.maxDist) || (key == endkey);
    float u;
    if (two_tier_visit(P, pose, ray, L, fx, fy, fz, &u)) {
      // the home bucket first, in straight-line code: the key is there (7 visits in 8) or one of its slots is free and
      // takes it (nearly all the rest); whatever else — the slot went to another voxel, the bucket is full — is the
      // general search's business
      int e;
      {
        const uint32_t b = key_bucket<SH>(key);
        const uint4 k4 = *reinterpret_cast<const uint4*>(&S.ekey[4 * b]);
        const int j = bucket_match(k4, key);
        if (j >= 0) {
          e = (int)(4 * b) + j;
        } else {
          const int je = bucket_match(k4, kKeyEmpty);
          uint32_t old = 0u;
          if (je >= 0) old = atomicCAS(&S.ekey[4 * b + je], kKeyEmpty, key);
          e = (je >= 0 && (old == kKeyEmpty || old == key)) ? (int)(4 * b) + je : table_find_or_insert(S, key);
        }
      }
      if (__builtin_expect(e < 0, 0)) break;   // the table is full: the (sub-)tile is cut
      if (kRuns && vp < (uint32_t)(kLogLen * kWalkRays * sizeof(uint16_t)))
        *reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(vlog) + vp) = (uint16_t)e;
      if (kAcc) {
        atomicAdd(&e_wuu[e], __float2int_rn(wu_scaled * u));
        atomicAdd(&e_wc[e], (1ull << 32) | (unsigned long long)q_w);
        atomicMax(&e_last[e], rid);   // (the ray's index in the tile: = tid unless the tile's rays were re-dealt)
      }
      vp += (uint32_t)(kWalkRays * sizeof(uint16_t));
    }
    if (stop) break;
    // ---- Raycast.cpp:131-180: the axis with the smallest tMax steps (the reference's comparisons and tie rules)
    const bool x_lt_y = tmx < tmy, x_lt_z = tmx < tmz, y_lt_z = tmy < tmz;
    const bool go_x = x_lt_y && x_lt_z, go_y = !x_lt_y && y_lt_z;
    const float mx = go_x ? 1.0f : 0.0f, my = go_y ? 1.0f : 0.0f, mz = (go_x || go_y) ? 0.0f : 1.0f;
    fx = fmaf(mx, sfx, fx);      // (exact: integers)
    fy = fmaf(my, sfy, fy);
    fz = fmaf(mz, sfz, fz);
    tmx = fmaf(mx, tdx, tmx);    // RN(tMax + tDelta) on the stepping axis, unchanged on the others
    tmy = fmaf(my, tdy, tmy);
    tmz = fmaf(mz, tdz, tmz);
    key += go_x ? kx : (go_y ? ky : kz);
  }
  return vp >> 10;
}

// ------------------------------------------------------------------ rays straight from depth images (round 5)
// plvs_hip_tsdf_chisel_integrate_depth_batch_dev: the call's "clouds" are depth images and the points are what
// PointCloudMapping::GeneratePointCloudInCameraFrameBGRA (src/PointCloudMapping.cc:957-996) would have pushed into the
// cloud — p = (gx d, gy d, d) for every pixel of the stride-`step` grid with min < d < max, in raster order — but they are
// never written: a tile is a 32 x 16 block of GRID PIXELS (kGridTileW x kGridTileH), thread r = pixel (r / 32, r % 32) of
// the block.  Rays of a 2-D block share their voxels: 2-3 x fewer table entries and records per tile than 512
// CONSECUTIVE points (a strip 1.6 rows high), and a chunk meets a quarter of the tiles.
// What needs the reference's point order still has it:
//   * the order of two visits of one voxel = (image, raster index of the grid pixel) — inside a tile that is the ray
//     number (bit order of a run's mask), between tiles of one band of 16 grid rows it interleaves row by row: the fold
//     (fold_colours_masks<true>) merges the runs of a band by mask word, one word = one grid row of a tile;
//   * a record's last visitor is the ORDER KEY image << key_bits | raster index (maxima compare like point indices);
//     apply_chunks takes the key-frame id from kfid[key >> key_bits].
// Tile t of the call: image t / (ntx nty), band (t / ntx) % nty, column block t % ntx.
constexpr int kGridTileW = 32, kGridTileH = kWalkRays / kGridTileW;
static_assert(kGridTileW == 32 && kGridTileH == kMaskWords, "one mask word per grid row of a tile");
struct GridSrc {
  const float* depth;        // nullptr: the call's points come from its point stream (xyz)
  const float* cam;          // matCamGridPoints_: (gx, gy) per grid pixel, gh x gw x 2
  unsigned long long image_stride;   // floats between two images
  uint32_t pitch;            // floats per image row
  uint32_t step, gw, gh;     // grid: pixel (m, n) of the grid is pixel (m step, n step) of the image
  uint32_t ntx, nty;         // tiles per band, bands per image
  uint32_t key_bits;         // bits of the raster index gh x gw in an order key
  double min_depth, max_depth;
  // the colour images (the fold reads a visit's colour where the reference's cloud would carry it: bytes 0, 1, 2 of the
  // pixel -> the point's r, g, b members, :978-980)
  unsigned long long bgr_image_stride;   // bytes between two images
  uint32_t bgr_pitch;                    // bytes per image row
  float inv_ntx, inv_nty;                // 1 / ntx, 1 / nty (fast_div: a tile index is split once per run in the fold)
};
struct GridTile {
  uint32_t cloud, row0, col0;   // image, first grid row / column of the tile
};
// x / d for x below 2^24 by the float reciprocal (one multiply and a fix-up instead of the ~40 instructions of a 32-bit
// division: the colour fold splits a tile index for every run it reads); the plain quotient beyond.
__device__ __forceinline__ uint32_t fast_div(uint32_t x, uint32_t d, float inv) {
  if (x >= (1u << 24)) return x / d;
  uint32_t q = (uint32_t)((float)x * inv);
  const int32_t r = (int32_t)(x - q * d);
  q = r < 0 ? q - 1u : ((uint32_t)r >= d ? q + 1u : q);
  return q;
}
// tile -> band (image and band of grid rows in one number: tiles are numbered band by band)
__device__ __forceinline__ uint32_t grid_band(const GridSrc& g, uint32_t gtile) { return fast_div(gtile, g.ntx, g.inv_ntx); }
__device__ __forceinline__ GridTile grid_tile_of_band(const GridSrc& g, uint32_t gtile, uint32_t band) {
  const uint32_t c = fast_div(band, g.nty, g.inv_nty);
  return GridTile{c, (band - c * g.nty) * (uint32_t)kGridTileH, (gtile - band * g.ntx) * (uint32_t)kGridTileW};
}
__device__ __forceinline__ GridTile grid_tile(const GridSrc& g, uint32_t gtile) {
  return grid_tile_of_band(g, gtile, grid_band(g, gtile));
}
// The point ray `rid` of the tile would be in the reference's cloud; false: the grid pixel does not exist or its depth
// is not inside (min, max) (compared in double, :967).
__device__ __forceinline__ bool grid_point(const GridSrc& g, const GridTile& t, uint32_t rid, float* x, float* y, float* z) {
  const uint32_t m = t.row0 + rid / (uint32_t)kGridTileW, n = t.col0 + rid % (uint32_t)kGridTileW;
  if (m >= g.gh || n >= g.gw) return false;
  const float d = g.depth[(size_t)t.cloud * g.image_stride + (size_t)(m * g.step) * g.pitch + n * g.step];
  if (!(((double)d > g.min_depth) && ((double)d < g.max_depth))) return false;
  const float2 c = reinterpret_cast<const float2*>(g.cam)[m * g.gw + n];
  *x = c.x * d;   // :973-975 (two float products)
  *y = c.y * d;
  *z = d;
  return true;
}
__device__ __forceinline__ uint32_t grid_order_key(const GridSrc& g, const GridTile& t, uint32_t rid) {
  return (t.cloud << g.key_bits) | ((t.row0 + rid / (uint32_t)kGridTileW) * g.gw + t.col0 + rid % (uint32_t)kGridTileW);
}

// Ray of point i under the pose of its cloud; false if the point casts no ray (or lies outside the
// supported extent).
__device__ __forceinline__ bool ray_of_point(const Params& P, const Pose& pose, float x, float y, float z, Ray* ray,
                                             uint32_t* err) {
  if (!make_ray(P, pose, x, y, z, ray)) return false;
  if (!ray_in_coord_range(*ray)) {
    atomicOr(err, kErrCoordRange);
    return false;
  }
  if (P.shard_count > 2 && !walk_may_touch_owned(P, *ray)) return false;
  return true;
}
// ... of ray `rid` of a tile: point first + rid of the stream, or the grid pixel of a depth-image tile
__device__ __forceinline__ bool tile_ray_src(const Params& P, const float* __restrict__ xyz, const GridSrc* __restrict__ g,
                                             const GridTile& gt, const Pose& pose, uint32_t first, uint32_t rid, Ray* ray,
                                             uint32_t* err) {
  float x, y, z;
  if (g) {   // (a pointer, not a kernel argument by value: its twenty words would live in scalar registers through the voxel loop)
    if (!grid_point(*g, gt, rid, &x, &y, &z)) return false;
  } else {
    const size_t i = (size_t)first + rid;
    x = xyz[3 * i]; y = xyz[3 * i + 1]; z = xyz[3 * i + 2];
  }
  return ray_of_point(P, pose, x, y, z, ray, err);
}
__device__ __forceinline__ bool tile_ray(const Params& P, const float* __restrict__ xyz, const Pose& pose, uint32_t i,
                                         Ray* ray, uint32_t* err) {
  if (!make_ray(P, pose, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], ray)) return false;
  if (!ray_in_coord_range(*ray)) {
    atomicOr(err, kErrCoordRange);
    return false;
  }
  if (P.shard_count > 2 && !walk_may_touch_owned(P, *ray)) return false;
  return true;
}

template <class SH>
__device__ __forceinline__ void subtile_reset(SH& S, int tid) {
#pragma unroll
  for (int k = 0; k < SH::kEntries / kWalkRays; ++k) S.ekey[tid + k * kWalkRays] = kKeyEmpty;
  if (tid < kWalkChunks) {
    S.ccode[tid] = kKeyEmpty;
    S.cslot[tid] = -1;
  }
#pragma unroll
  for (int k = 0; k < kWalkChunks * kSlabs / kWalkRays; ++k) S.ccnt[tid + k * kWalkRays] = 0;
  if (tid == 0) {
    S.nent = 0;
    S.overflow = 0;
  }
}

// After an overflow: cut the (sub-)tile (thread 0).  Rays first; a single ray is cut into windows of its
// visits (a window of kWalkWindow visits always fits).
__device__ __forceinline__ void subtile_split(WalkShared& S, const SubTile st) {
  if (st.hi - st.lo > 1) {
    const uint16_t mid = (uint16_t)((st.lo + st.hi) / 2);
    S.stack[S.sp++] = SubTile{mid, st.hi, st.vlo, st.vhi, st.cloud};
    S.stack[S.sp++] = SubTile{st.lo, mid, st.vlo, st.vhi, st.cloud};   // processed first: sub-tiles stay in point order
  } else {
    const uint32_t a = st.vlo, b = (st.vhi == 0xFFFFFFFFu) ? a + 2u * kWalkWindow : st.vhi;
    const uint32_t mid = (st.vhi == 0xFFFFFFFFu) ? a + kWalkWindow : a + (b - a) / 2;
    S.stack[S.sp++] = SubTile{st.lo, st.hi, mid, st.vhi, st.cloud};
    S.stack[S.sp++] = SubTile{st.lo, st.hi, a, mid, st.cloud};
  }
}

// Cache index of a chunk (by its code) among the (sub-)tile's chunks; -1: the cache is full.  *won: this call
// created the entry.
template <class SH>
__device__ __forceinline__ int chunk_cache_insert(SH& S, uint32_t code, bool* won) {
  uint32_t h = (code * 2654435761u) >> 26;
  static_assert(kWalkChunks == 64, "the hash yields 6 bits");
  *won = false;
  for (int probe = 0; probe < kWalkChunks; ++probe) {
    uint32_t cur = S.ccode[h];
    if (cur == kKeyEmpty) {
      cur = atomicCAS(&S.ccode[h], kKeyEmpty, code);
      if (cur == kKeyEmpty) {
        *won = true;
        return (int)h;
      }
    }
    if (cur == code) return (int)h;
    h = (h + 1) & (kWalkChunks - 1);
  }
  return -1;
}

// The home entry of a chunk in the directory, read without waiting: the walk's set-up issues these loads for the
// chunks its rays start and end in and looks at them only after the rest of the set-up arithmetic, so the flush
// finds the pool slots of (nearly) all its chunks already in the cache instead of paying two dependent global
// round trips with seven of the tile's eight waves idle.  A chunk that is not at its home entry, not in the map yet
// or just being inserted by another tile stays unresolved (-1) and takes dir_find_or_insert in the flush.
struct DirPeek {
  unsigned long long want, key;
  int32_t slot;
};
__device__ __forceinline__ DirPeek dir_peek(const Directory& d, int cx, int cy, int cz) {
  DirPeek p;
  p.key = kEmptyKey;
  p.slot = -1;
  if (!pack_block(cx, cy, cz, &p.want)) {
    p.want = 0;   // (never equals kEmptyKey)
    return p;
  }
  const uint32_t h = dir_hash(cx, cy, cz, d.mask);
  p.key = d.keys[h];
  p.slot = d.slots[h];
  return p;
}
__device__ __forceinline__ int dir_peek_slot(const DirPeek& p) { return (p.key == p.want && p.slot >= 0) ? p.slot : -1; }

// ------------------------------------------------------------------ walk_tiles
// Output layout of the order-free records: tile t owns records [t * kWalkLimit, (t + 1) * kWalkLimit)
// and segment descriptors [t * kWalkChunks, ...) for its first flush — no allocation, no same-address
// atomics (a returning device-scope atomic on one word costs ~11 ns: 30 000 tiles queueing for one
// counter would take longer than the walk).  Further flushes of a tile that had to be cut, and entries
// beyond the chunk cache, go to a spill area behind the per-tile regions through two counters (rare).
// A segment = the records of one (tile, chunk), grouped by slab; its descriptor is two uint4:
//   {slot, first record, records, 0}, {first record of slab 0..7 inside the segment, 8 x u16}.
struct AccOut {
  uint4* rec;                // {vid | count << 12, last point, sum w_u*u (fixed), sum w_u (fixed)}
  uint32_t rec_cap;          // ntiles * kWalkLimit + spill
  uint4* seg;                // two per segment
  uint32_t seg_cap;          // ntiles * kWalkChunks + spill (in segments)
  uint32_t* seg_cnt;         // [ntiles] segments in the tile's own region
  uint32_t* tile_visits;     // [ntiles]
  // (round 5) per pool slot: segments of the call, counted where they are written — one fire-and-forget atomic per
  // (tile, chunk), spread over the walk — instead of by a pass over every descriptor slot behind it (seg_pass<false>:
  // 46 us for the 960 000 slots of a 100-key-frame call, nine tenths of them empty).  nullptr: seg_pass<false> counts.
  uint32_t* chunk_nseg;
};
// Runs: one descriptor + 256-bit ray mask per (tile, voxel) that needs its visits in order.
// Tile t owns run slots [t << r1_log2, (t + 1) << r1_log2) and fills them from the front, in the order
// of its flushes (= point order); run_cnt[t] = how many.  No numbering across tiles inside the kernel
// (a look-back made every tile wait for its slowest neighbour): compact_runs lists (key, slot) densely
// in tile order for the stable sort by voxel key; the tile (slot >> r1_log2) and the mask follow from the slot.  A tile that needs more slots reports how many
// (run_need) and the host repeats the call with larger regions.
struct RunOut {
  uint32_t* dkey;                 // [ntiles << r1_log2] voxel key (slot * 4096 + voxel)
  uint32_t* masks;                // [(ntiles << r1_log2) * kMaskWords] rays of the tile that visit the voxel, bit r = ray r
  uint32_t* run_cnt;              // [ntiles]
  uint32_t r1_log2;
  // walk_fast only, may be null: the runs of a tile lie grouped by (chunk, slab), and beside segment descriptor sg
  // (AccOut::seg) stands where its chunk's runs are: [2 sg] = {first run slot, runs, 0, 0}, [2 sg + 1] = the first run of
  // each slab inside them (pack_suboffsets) — what runs_count / runs_scatter place a chunk's runs by without a sort of all runs
  uint4* rseg;
};

constexpr int kMaskCap = kWalkEntries / 4;   // ray masks built per round (LDS: the area of the accumulators)

__device__ __forceinline__ uint4 pack_suboffsets(const uint32_t* o) {
  return make_uint4(o[0] | (o[1] << 16), o[2] | (o[3] << 16), o[4] | (o[5] << 16), o[6] | (o[7] << 16));
}

// Which tile of the point stream a workgroup takes: its block index itself (the whole stream on one device) or
// every stride-th tile (a rank of the ray-sharded multi-GPU integrate).  The output regions are indexed by the
// block ("local tile").
struct TileMap {
  uint32_t stride, first;
  uint32_t tables;   // 1: the call's offsets carry tile_first / tile_cloud behind the tile table (tile_span_tables)
  __device__ __host__ uint32_t tile_of(uint32_t local) const { return local * stride + first; }
};

// Order-free accumulation (records + segments) and the runs of the voxels whose colour weight is below 254.
// `sat` = nullptr: the single-device integrate, the colour weight is read from rgbw.  Otherwise a rank's share
// of the rays in the ray-sharded integrate: dir = the rank's directory of every chunk it has walked through,
// sat = its bitmap of the voxels their owners have reported saturated (4096 bits per directory slot).
template <bool kAcc, bool kRuns>
__global__ __launch_bounds__(kWalkRays, kWalkEntries > 2048 ? 2 : (kWalkEntries > 1024 ? 4 : 6)) void walk_tiles(
    Params P, float scale_u, float scale_w, const float* __restrict__ xyz, int npoints,
    const int32_t* __restrict__ offsets, int nclouds, const Pose* __restrict__ poses, Directory dir,
    int32_t* __restrict__ num_chunks, WalkCounters* __restrict__ ctr, const uint32_t* __restrict__ rgbw,
    const uint32_t* __restrict__ sat, AccOut out, RunOut runs, TileMap tmap, uint32_t ntiles,
    const uint32_t* __restrict__ tile_list, const uint32_t* __restrict__ ntile_list, uint32_t rec_stride,
    uint32_t cut_pieces, const GridSrc* __restrict__ grid) {
  constexpr int kPer = kWalkEntries / kWalkRays;
  __shared__ WalkShared S;
  __shared__ uint32_t raw[kMaskCap * kMaskWords];              // accumulators during the walk, ray masks afterwards
  __shared__ uint16_t vlog[kLogLen * kWalkRays];        // entry of visit k of ray r at [k * kWalkRays + r]
  __shared__ uint16_t e_midx[kWalkEntries];           // mask index of the entry (0xFFFF: none)
  int32_t* const e_wuu = reinterpret_cast<int32_t*>(raw);                                 // sum of w_u * u, fixed point
  unsigned long long* const e_wc = reinterpret_cast<unsigned long long*>(raw + kWalkEntries);   // visits << 32 | sum of w_u
  uint32_t* const e_last = raw + 3 * kWalkEntries;                                        // last visiting ray
  static_assert(kMaskCap * kMaskWords >= 4 * kWalkEntries, "the accumulators overlay the mask area");
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // the tiles walk_fast deferred (tile_list) — or every tile of the call (tile_list = nullptr: blockIdx is the tile)
  const uint32_t nlist = tile_list ? min(*ntile_list, ntiles) : ntiles;
  for (uint32_t lb = blockIdx.x; lb < nlist; lb += gridDim.x) {
  // (bit 31 of a list entry: walk_fast saw the tile overflow its table — it is cut in two at once)
  const uint32_t tile = tile_list ? (tile_list[lb] & 0x7FFFFFFFu) : lb;   // the local tile: output regions
  const bool cut_at_once = tile_list && (tile_list[lb] >> 31) != 0u;
  const uint32_t gtile = tmap.tile_of(tile);             // its place in the point stream
  const GridTile gt = grid ? grid_tile(*grid, gtile) : GridTile{0u, 0u, 0u};
  const TileSpan span = grid ? TileSpan{(int)gt.cloud, 0u, (uint32_t)kWalkRays}   // (a block of grid pixels of one image)
                      : tmap.tables ? tile_span_tables(offsets, nclouds, gtile, kWalkRays)   // (512 points of one cloud:
                                    : tile_span(offsets, nclouds, gtile, kWalkRays);          //  tsdf_directory.hpp)
  const uint32_t first = span.first;
  __syncthreads();   // (the previous tile of this workgroup is done with the shared state)
  // Nearly every tile lies inside one cloud and fits its table: ONE sub-tile, known without a word of shared memory.
  // A barrier costs a tile about a microsecond (the slowest of eight waves, their memory operations drained) — the
  // common path below has five of them; the sub-tile stack (several clouds in the tile, a table overflow) costs three
  // more per sub-tile.
  const uint32_t nrays = span.nrays;
  const int cloud0 = span.cloud;
  const bool one_cloud = true;   // (by construction of the tiles; the cloud-by-cloud hand-out below stays for a caller-defined tiling)
  const bool single = one_cloud && !(cut_at_once && nrays > 1u);
  if (tid == 0) {
    S.sp = 0;
    S.next = one_cloud ? nrays : 0u;
    if (one_cloud && !single) {
      // cut_pieces (2 or 4: how far the table that overflowed exceeds this kernel's) equal ray ranges, the lowest on top
      // (sub-tiles stay in point order); a piece that still does not fit is halved as usual
      const uint32_t np = min(cut_pieces, nrays);
      for (uint32_t q = np; q-- > 0;)
        S.stack[S.sp++] = SubTile{(uint16_t)(nrays * q / np), (uint16_t)(nrays * (q + 1) / np), 0u, 0xFFFFFFFFu, cloud0};
    }
    S.nrays = nrays;
    S.run_total = 0;
    S.vis_total = 0;
  }
  uint32_t my_visits = 0;
  bool was_split = one_cloud && !single;
  int flushes = 0;
  bool first_pass = true;
  WALK_PROF_BEGIN();

  while (true) {
    SubTile st;
    if (first_pass && single) {
      st = SubTile{(uint16_t)0, (uint16_t)nrays, 0u, 0xFFFFFFFFu, cloud0};
    } else {
      __syncthreads();
      // A sub-tile holds rays of ONE cloud, so that its pose is uniform (scalar registers); the tile hands
      // out its rays cloud by cloud.
      if (tid == 0 && S.sp == 0 && S.next < S.nrays) {
        const int c = cloud_of(offsets, nclouds, (int)(first + S.next));
        const uint32_t end = min((uint32_t)(offsets[c + 1] - (int32_t)first), S.nrays);
        S.stack[S.sp++] = SubTile{(uint16_t)S.next, (uint16_t)end, 0u, 0xFFFFFFFFu, c};
        S.next = end;
      }
      __syncthreads();
      if (S.sp == 0) break;
      st = S.stack[S.sp - 1];
      __syncthreads();
      if (tid == 0) --S.sp;
    }
    first_pass = false;
    subtile_reset(S, tid);
#pragma unroll
    for (int k = 0; k < 4 * kPer; ++k) raw[tid + k * kWalkRays] = 0u;
    const Pose& pose = poses[__builtin_amdgcn_readfirstlane(st.cloud)];   // (scalar loads where it is used)
    Ray ray;
    const bool walks = tid >= st.lo && tid < st.hi && tile_ray_src(P, xyz, grid, gt, pose, first, (uint32_t)tid, &ray, &ctr->err);
    const float wu = walks ? P.weight / (2.0f * ray.truncation) : 0.0f;
    const float wu_scaled = wu * scale_u;
    const uint32_t q_w = (uint32_t)__float2int_rn(wu * scale_w);
    // origin of the voxel keys: below the start voxel of the first walking ray, on a chunk boundary (a lone ray sets
    // it at the first visit of its window instead: its windows may lie anywhere along a very long ray)
    const bool lone = st.hi - st.lo == 1;
    {
      const unsigned long long wm = __ballot(walks);
      if (lane == 0) S.cand[wid] = wm ? 1u : 0u;
      if (wm && lane == __ffsll((long long)wm) - 1)
        for (int k = 0; k < 3; ++k) S.worg[wid][k] = origin_of((int)floorf(ray.start[k]));
    }
    __syncthreads();
    int ox = 0, oy = 0, oz = 0;
    if (!lone) {   // the first wave that has a walking ray names the origin (every thread reads the same words)
      int w0 = 0;
#pragma unroll
      for (int w = kWalkRays / 64 - 1; w >= 0; --w) w0 = S.cand[w] ? w : w0;
      ox = S.worg[w0][0]; oy = S.worg[w0][1]; oz = S.worg[w0][2];
      if (tid == 0) { S.org[0] = ox; S.org[1] = oy; S.org[2] = oz; }   // (the flush reads S.org: a lone ray sets it later)
    }
    bool org_set = !lone;
    uint32_t nv = 0;
    // the lean walk: whole rays of a sub-tile of several rays on an unsharded map (everything but the re-walks
    // of a ray that did not fit the table and the owner-filtered walk of a sharded handle)
    const bool lean = !lone && st.vlo == 0u && st.vhi == 0xFFFFFFFFu && P.shard_count <= 1 && !(PLVS_WALK_EXP & 512);
    if (lean) {
      // ---- the chunks the ray starts and ends in -> chunk cache, their directory entries requested (DirPeek) ...
      bool fits = false, won_s = false, won_e = false;
      int ci_s = -1, ci_e = -1;
      uint32_t code_s = kKeyEmpty, code_e = kKeyEmpty;
      if (walks) {
        const int reach = lean_reach(ray);
        const int rx = (int)floorf(ray.start[0]) - ox, ry = (int)floorf(ray.start[1]) - oy, rz = (int)floorf(ray.start[2]) - oz;
        fits = min(min(rx, ry), rz) - reach >= 0 && max(max(rx, ry), rz) + reach <= 1023;
        if (fits) {
          code_s = chunk_code((uint32_t)rx | ((uint32_t)ry << 10) | ((uint32_t)rz << 20));
          code_e = chunk_code((uint32_t)((int)floorf(ray.end[0]) - ox) | ((uint32_t)((int)floorf(ray.end[1]) - oy) << 10) |
                              ((uint32_t)((int)floorf(ray.end[2]) - oz) << 20));
        } else {
          S.overflow = 1u;   // the rays of the sub-tile are too far apart: it is cut (a lone ray takes the general walk)
        }
      }
      {   // neighbouring rays share their chunks: only the first lane of a run of equal codes goes to the cache
        const uint32_t prev_s = (uint32_t)__shfl_up((int)code_s, 1), prev_e = (uint32_t)__shfl_up((int)code_e, 1);
        if (code_s != kKeyEmpty && (lane == 0 || code_s != prev_s)) ci_s = chunk_cache_insert(S, code_s, &won_s);
        if (code_e != kKeyEmpty && code_e != code_s && (lane == 0 || code_e != prev_e)) ci_e = chunk_cache_insert(S, code_e, &won_e);
      }
      DirPeek peek_s, peek_e;
      peek_s.want = peek_e.want = 0; peek_s.key = peek_e.key = kEmptyKey; peek_s.slot = peek_e.slot = -1;
      if (won_s) {
        int cx, cy, cz;
        chunk_of_code(code_s, ox, oy, oz, &cx, &cy, &cz);
        peek_s = dir_peek(dir, cx, cy, cz);
      }
      if (won_e) {
        int cx, cy, cz;
        chunk_of_code(code_e, ox, oy, oz, &cx, &cy, &cz);
        peek_e = dir_peek(dir, cx, cy, cz);
      }
      WALK_PROF(0);   // sub-tile set-up: table reset, ray of the point, key origin
      // ---- ... the walk (its set-up arithmetic hides the directory's latency; the slots land before the loop)
      if (fits && !(PLVS_WALK_EXP & 32)) {
        RayCursor cur;
        ray_begin(ray, &cur);   // (walk_lean's own ray_begin is this one: common subexpression)
        if (won_s && dir_peek_slot(peek_s) >= 0) S.cslot[ci_s] = dir_peek_slot(peek_s);
        if (won_e && dir_peek_slot(peek_e) >= 0) S.cslot[ci_e] = dir_peek_slot(peek_e);
        nv = walk_lean<kAcc, kRuns>(P, pose, ray, S, ox, oy, oz, tid, wu_scaled, q_w, e_wuu, e_wc, e_last, vlog, (uint32_t)tid);
      }
    } else if (walks && !(PLVS_WALK_EXP & 32)) {   // (bit 32, timing experiment: set-up and flush only)
      WALK_PROF(0);
      nv = walk_one(P, pose, ray, st.vlo, st.vhi, [&](uint32_t k, int vx, int vy, int vz, float u) {
        if (!org_set) {   // (one lane only)
          ox = origin_of(vx); oy = origin_of(vy); oz = origin_of(vz);
          S.org[0] = ox; S.org[1] = oy; S.org[2] = oz;
          org_set = true;
        }
        uint32_t key;
        if (!rel_key(vx, vy, vz, ox, oy, oz, &key)) {   // the rays of the sub-tile are too far apart: cut it
          S.overflow = 1u;
          return false;
        }
        const int e = table_find_or_insert(S, table_key(key));
        if (e < 0) return false;
        if (kRuns && k < (uint32_t)kLogLen) vlog[k * kWalkRays + tid] = (uint16_t)e;
        if (kAcc) {
          atomicAdd(&e_wuu[e], __float2int_rn(wu_scaled * u));
          atomicAdd(&e_wc[e], (1ull << 32) | (unsigned long long)q_w);
          atomicMax(&e_last[e], (uint32_t)tid);
        }
        return true;
      });
    }
    WALK_PROF(1);   // thread 0's own walk
#if (PLVS_WALK_EXP & 2048)
    __builtin_amdgcn_s_setprio(3);   // the flush: short dependent phases between barriers — ahead of the other tiles' walks
#endif
    __syncthreads();
    WALK_PROF(2);   // ... and the wait for the tile's longest ray
    uint32_t ekey[kPer];
    {   // entries in use (nobody counts them during the walk)
      uint32_t mine = 0;
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        ekey[k] = S.ekey[tid + k * kWalkRays];
        mine += ekey[k] != kKeyEmpty ? 1u : 0u;
        if (ekey[k] != kKeyEmpty) ekey[k] = voxel_key(ekey[k]);   // (the table holds table keys)
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) mine += (uint32_t)__shfl_xor((int)mine, off);
      if (lane == 0 && mine) atomicAdd(&S.nent, mine);
    }
    __syncthreads();
    const bool overflowed = S.overflow != 0 || S.nent > (uint32_t)kWalkLimit;
    if (overflowed) {
      __syncthreads();   // (everybody has read the flag before the stack changes)
      was_split = true;
      if (tid == 0) subtile_split(S, st);
      continue;
    }
    const uint32_t nmine = nv > st.vlo ? nv - st.vlo : 0u;   // this ray's visits in the (sub-)tile
    my_visits += nmine;
#if (PLVS_WALK_EXP & 16)   // (timing experiment: no flush)
    ++flushes;
    continue;
#endif

    // ---- entries -> chunks -> pool slots
    const int fox = S.org[0], foy = S.org[1], foz = S.org[2];   // (a lone ray set them during its walk)
    int ci[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      ci[k] = -2;
      if (ekey[k] != kKeyEmpty) {
        bool won;
        ci[k] = chunk_cache_insert(S, chunk_code(ekey[k]), &won);
        // (a chunk joins the map when one of its voxels takes an update — not because a ray started or ended in it)
        if (ci[k] >= 0 && S.cslot[ci[k]] == -1) S.cslot[ci[k]] = -3;   // wanted, not resolved yet
      }
    }
    bool unresolved = false;
#pragma unroll
    for (int k = 0; k < kPer; ++k) unresolved = unresolved || (ci[k] >= 0 && S.cslot[ci[k]] < 0);
    // (one barrier: the cache is complete — and is any chunk with entries still without its pool slot?  The lean walk
    // resolved the chunks its rays start and end in ahead of the loop; what can be left: chunks a ray only passes
    // through, chunks not at their home entry of the directory, and the first-touch chunks, inserted here)
    const bool any_unresolved = __syncthreads_or(unresolved ? 1 : 0) != 0;
    WALK_PROF(3);   // entries -> chunk cache
    if (any_unresolved) {
      if (tid < kWalkChunks && S.cslot[tid] == -3) {
        int cx, cy, cz;
        chunk_of_code(S.ccode[tid], fox, foy, foz, &cx, &cy, &cz);
        S.cslot[tid] = dir_find_or_insert(dir, cx, cy, cz, num_chunks, &ctr->err);
      }
      __syncthreads();
    }
    WALK_PROF(4);   // chunk cache -> directory
    uint32_t rank[kPer], vkey[kPer], elast[kPer];
    unsigned long long ewc[kPer];
    int slot_of[kPer];
    uint32_t need = 0, nneed = 0;   // entries of this thread that leave the tile as runs
    bool multi = false;             // ... one of them visited by more than one ray
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      rank[k] = 0;
      vkey[k] = 0;
      slot_of[k] = -1;
      elast[k] = 0;
      ewc[k] = 0;
      if (ci[k] == -2) continue;
      const uint32_t vid = voxel_in_chunk(ekey[k]);
      ewc[k] = e_wc[tid + k * kWalkRays];
      elast[k] = e_last[tid + k * kWalkRays];
      if (ci[k] >= 0) {
        slot_of[k] = S.cslot[ci[k]];
        if (slot_of[k] >= 0) rank[k] = atomicAdd(&S.ccnt[ci[k] * kSlabs + (int)(vid / kSlabVox)], 1u);
      } else {
        int cx, cy, cz;
        chunk_of_code(chunk_code(ekey[k]), fox, foy, foz, &cx, &cy, &cz);
        slot_of[k] = dir_find_or_insert(dir, cx, cy, cz, num_chunks, &ctr->err);
      }
      if (slot_of[k] >= 0) {
        vkey[k] = (uint32_t)slot_of[k] * (uint32_t)kChunkVox + vid;
#if (PLVS_WALK_EXP & 128)   // (timing experiment: no colour-weight loads)
        const bool cold = false;
#else
        const bool cold = sat ? ((sat[vkey[k] >> 5] >> (vkey[k] & 31u)) & 1u) == 0u : (rgbw[vkey[k]] >> 24) < 254u;
#endif
        if (kRuns && cold) {   // its colour still depends on the order of the visits
          need |= 1u << k;
          ++nneed;
          multi = multi || (uint32_t)(ewc[k] >> 32) > 1u;
        }
      }
    }
    // (one barrier: the ranks are complete, and does any voxel that needs a run have more than one visiting ray?)
    const bool any_multi = __syncthreads_or(multi ? 1 : 0) != 0;
    WALK_PROF(5);   // ranks, colour weights
    if (kAcc) {
      // ---- records: the (chunk, slab) groups are placed by a scan over the cache's 64 chunks, one lane per chunk.
      // First flush of the tile (its own record / segment regions, no allocation): EVERY wave runs the scan and writes
      // the same table of group bases — a wave reads back its own LDS writes in order, so nobody waits for wave 0
      // behind a barrier; wave 0 alone writes the segment descriptors.  Later flushes (a tile that was cut) take their
      // regions from the spill counters: wave 0 scans, the others wait.
      const bool own_region = flushes == 0;
      uint32_t rbase = tile * rec_stride;
      if (own_region || tid < 64) {
        const int cl = lane;   // the chunk of this lane
        uint32_t sub[kSlabs], c = 0;
#pragma unroll
        for (int s = 0; s < kSlabs; ++s) {
          sub[s] = c;
          c += S.ccnt[cl * kSlabs + s];
        }
        uint32_t inc = c, sinc = c ? 1u : 0u;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const uint32_t up = (uint32_t)__shfl_up((int)inc, off), sup = (uint32_t)__shfl_up((int)sinc, off);
          if (lane >= off) { inc += up; sinc += sup; }
        }
        const uint32_t tot = (uint32_t)__shfl((int)inc, 63), stot = (uint32_t)__shfl((int)sinc, 63);
        uint32_t sbase = tile * (uint32_t)kWalkChunks;
        if (!own_region) {
          if (tid == 0 && tot) {
            rbase = ntiles * rec_stride + atomicAdd(&ctr->rec_top, tot);
            sbase = ntiles * (uint32_t)kWalkChunks + atomicAdd(&ctr->seg_top, stot);
            if (rbase + tot > out.rec_cap || sbase + stot > out.seg_cap) atomicOr(&ctr->err, kErrScratch);
          }
          rbase = (uint32_t)__shfl((int)rbase, 0);
          sbase = (uint32_t)__shfl((int)sbase, 0);
          if (tid == 0) S.rbase = rbase;
        } else if (tid == 0) {
          out.seg_cnt[tile] = stot;
        }
#pragma unroll
        for (int s = 0; s < kSlabs; ++s) S.cbase[cl * kSlabs + s] = (uint16_t)(inc - c + sub[s]);
        if (c && tid < 64) {
          const uint32_t sg = sbase + sinc - 1u;
          if (sg < out.seg_cap) {
            out.seg[2 * (size_t)sg] = make_uint4((uint32_t)S.cslot[cl], rbase + inc - c, c, gtile);
            out.seg[2 * (size_t)sg + 1] = pack_suboffsets(sub);
            if (out.chunk_nseg) atomicAdd(&out.chunk_nseg[S.cslot[cl]], 1u);
          }
        }
      }
      if (!own_region) {
        __syncthreads();
        rbase = S.rbase;
      }
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        if (slot_of[k] < 0) continue;
        const int e = tid + k * kWalkRays;
        const uint32_t vid = vkey[k] % (uint32_t)kChunkVox;
        const uint4 r = make_uint4(vid | ((uint32_t)(ewc[k] >> 32) << 12),
                                   grid ? grid_order_key(*grid, gt, elast[k]) : first + elast[k], (uint32_t)e_wuu[e],
                                   (uint32_t)ewc[k]);
        uint32_t at;
        if (ci[k] >= 0) {
          at = rbase + S.cbase[ci[k] * kSlabs + (int)(vid / kSlabVox)] + rank[k];
        } else {   // beyond the chunk cache (scattered clouds): a segment of its own in the spill area
          at = ntiles * rec_stride + atomicAdd(&ctr->rec_top, 1u);
          const uint32_t sg = ntiles * (uint32_t)kWalkChunks + atomicAdd(&ctr->seg_top, 1u);
          if (at >= out.rec_cap || sg >= out.seg_cap) {
            atomicOr(&ctr->err, kErrScratch);
          } else {
            uint32_t sub[kSlabs];
#pragma unroll
            for (int s = 0; s < kSlabs; ++s) sub[s] = (uint32_t)s > vid / kSlabVox ? 1u : 0u;
            out.seg[2 * (size_t)sg] = make_uint4((uint32_t)slot_of[k], at, 1u, gtile);
            out.seg[2 * (size_t)sg + 1] = pack_suboffsets(sub);
            if (out.chunk_nseg) atomicAdd(&out.chunk_nseg[slot_of[k]], 1u);
          }
        }
        if (at < out.rec_cap) out.rec[at] = r;
      }
    }

    ++flushes;
    WALK_PROF(6);   // records
    if (!kRuns || (PLVS_WALK_EXP & 64)) {   // (bit 64, timing experiment: no runs)
      if (single && !was_split) break;
      continue;
    }
    // ---- runs: number the entries that need one (wave by wave)
    uint32_t inc = nneed;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)inc, off);
      if (lane >= off) inc += up;
    }
    const uint32_t wave_runs = (uint32_t)__shfl((int)inc, 63);
    if (!any_multi) {
      // every voxel that needs a run has ONE visiting ray — the steady state of a map whose colours have mostly
      // saturated: what is still below 254 is what few rays reach — and the accumulator named it (e_last): the
      // mask is that one bit, no pass over the visit logs.  The tile's run slots are handed out wave by wave (the
      // order of a tile's runs among themselves is immaterial: they are runs of different voxels), no barrier.
      uint32_t base = 0;
      if (lane == 0 && wave_runs) base = atomicAdd(&S.run_total, wave_runs);
      base = (uint32_t)__shfl((int)base, 0);
      const bool fits_runs = base + wave_runs <= (1u << runs.r1_log2);
      if (!fits_runs && lane == 0) {
        atomicOr(&ctr->err, kErrScratch);
        atomicMax(&ctr->run_need, base + wave_runs);   // (the largest of these is the tile's total)
      }
      uint32_t m = base + inc - nneed;
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        if (!(need & (1u << k)) || !fits_runs) continue;
        const size_t d = ((size_t)tile << runs.r1_log2) + m++;
        runs.dkey[d] = vkey[k];
        uint4* dst = reinterpret_cast<uint4*>(runs.masks + d * kMaskWords);
        const uint32_t word = elast[k] >> 5, bit = 1u << (elast[k] & 31u);
#pragma unroll
        for (int q = 0; q < kMaskWords / 4; ++q)
          dst[q] = make_uint4(word == 4u * q ? bit : 0u, word == 4u * q + 1u ? bit : 0u, word == 4u * q + 2u ? bit : 0u,
                              word == 4u * q + 3u ? bit : 0u);
      }
      WALK_PROF(7);
      if (single && !was_split) break;
      continue;
    }
    if (lane == 63) S.wsum[wid] = inc;
    __syncthreads();   // (also: the records are out, the accumulator area is free)
    uint32_t wbase = 0, nruns = 0;
#pragma unroll
    for (int w = 0; w < kWalkRays / 64; ++w) {
      const uint32_t v = S.wsum[w];
      if (w < wid) wbase += v;
      nruns += v;
    }
    // the runs go to the tile's own slots, behind those of its earlier flushes
    const uint32_t emitted = S.run_total;
    const bool fits_runs = emitted + nruns <= (1u << runs.r1_log2);
    if (!fits_runs && tid == 0) {
      atomicOr(&ctr->err, kErrScratch);
      atomicMax(&ctr->run_need, emitted + nruns);
    }
    {
      uint32_t m = wbase + inc - nneed;
#pragma unroll
      for (int k = 0; k < kPer; ++k) e_midx[tid + k * kWalkRays] = (need & (1u << k)) ? (uint16_t)m++ : (uint16_t)0xFFFFu;
    }
    for (uint32_t r0 = 0; fits_runs && r0 < nruns; r0 += kMaskCap) {
      __syncthreads();   // e_midx complete / the previous round's masks are out
#pragma unroll
      for (int k = 0; k < kMaskCap * kMaskWords / kWalkRays; ++k) raw[tid + k * kWalkRays] = 0u;
      __syncthreads();
      if (walks && nmine) {
        const uint32_t logged = min(nmine, (uint32_t)kLogLen);
        for (uint32_t k = 0; k < logged; ++k) {
          const uint32_t m = (uint32_t)e_midx[vlog[k * kWalkRays + tid]] - r0;   // 0xFFFF - r0 stays out of range
          if (m < (uint32_t)kMaskCap) atomicOr(&raw[m * kMaskWords + (tid >> 5)], 1u << (tid & 31));
        }
        if (nmine > (uint32_t)kLogLen) {   // the log is full: the rest of the ray is walked again
          walk_one(P, pose, ray, st.vlo, st.vhi, [&](uint32_t k, int vx, int vy, int vz, float) {
            if (k >= (uint32_t)kLogLen) {
              uint32_t key;
              const int e = rel_key(vx, vy, vz, fox, foy, foz, &key) ? table_find(S, table_key(key)) : -1;
              const uint32_t m = e >= 0 ? (uint32_t)e_midx[e] - r0 : 0xFFFFFFFFu;
              if (m < (uint32_t)kMaskCap) atomicOr(&raw[m * kMaskWords + (tid >> 5)], 1u << (tid & 31));
            }
            return true;
          });
        }
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        const uint32_t m = (uint32_t)e_midx[tid + k * kWalkRays];
        if (m == 0xFFFFu || m < r0 || m >= r0 + (uint32_t)kMaskCap) continue;
        const size_t d = ((size_t)tile << runs.r1_log2) + emitted + m;
        runs.dkey[d] = vkey[k];
        const uint32_t* mk = raw + (m - r0) * kMaskWords;
        uint4* dst = reinterpret_cast<uint4*>(runs.masks + d * kMaskWords);
#pragma unroll
        for (int q = 0; q < kMaskWords / 4; ++q) dst[q] = make_uint4(mk[4 * q], mk[4 * q + 1], mk[4 * q + 2], mk[4 * q + 3]);
      }
    }
    __syncthreads();   // (everybody has read run_total)
    if (tid == 0) S.run_total = emitted + nruns;
    WALK_PROF(7);   // runs
    if (single && !was_split) break;
  }

  // ---- tile epilogue: run and visit counts
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) my_visits += (uint32_t)__shfl_xor((int)my_visits, off);
  if (lane == 0 && my_visits) atomicAdd(&S.vis_total, my_visits);
  __syncthreads();
  if (tid == 0) {
    if (kRuns) runs.run_cnt[tile] = min(S.run_total, 1u << runs.r1_log2);
    if (kAcc) {
      out.tile_visits[tile] = S.vis_total;
      if (flushes == 0) out.seg_cnt[tile] = 0;
    }
    if (was_split) atomicAdd(&ctr->split_tiles, 1u);
  }
  WALK_PROF(8);     // epilogue
  WALK_PROF_END();
  }   // (tiles of the list)
}

// ------------------------------------------------------------------ walk_fast: the common case of walk_tiles, alone
// A tile that lies inside one cloud, whose rays fit the key box and whose voxels fit the table — every tile of a depth
// camera's clouds but a handful — has ONE sub-tile and ONE flush.  This kernel is that case and nothing else: no
// sub-tile stack, no windows of a single ray, no spill regions, and so few enough live values that the voxel loop runs
// out of registers (walk_tiles with every path in one body spilled into its loop), and FOUR barriers per tile instead
// of eighteen (a barrier costs a tile ~1 us: the slowest of eight waves, their memory operations drained): key origin;
// end of the walk; ranks / entry count / "does a needy voxel have several rays"; counts out.  A tile that does not
// qualify is DEFERRED: all it has done to global memory when that is known is to enter chunks its voxels need anyway
// (every entry resolves its own chunk with dir_find_or_insert — no directory phase, no prefetch), no records, segments
// or runs; its index goes to a list and walk_tiles (the general kernel) walks the list behind this kernel.  Same
// outputs, same regions, bit for bit the same records either way.
// E = table entries (see FastShared).  tile_list = nullptr: workgroup b walks tile b; otherwise the tiles of a list an
// earlier launch left (entries with bit 31: tiles that overflowed ITS table — the others are not this kernel's case
// either and go straight on to `deferred`), *ntile_list of them.  rec_stride = records a tile owns in out.rec (the
// host sizes the regions for the largest table it launches).
constexpr int walk_fast_waves(int E) { return E > 2048 ? 2 : (E > 1024 ? 4 : 6); }   // waves per SIMD the tile's LDS allows
template <int E, bool kGrid>
__device__ __forceinline__ void walk_fast_tile(
    const Params& P, float scale_u, float scale_w, const float* __restrict__ xyz, int npoints,
    const int32_t* __restrict__ offsets, int nclouds, const Pose* __restrict__ poses, const Directory& dir,
    int32_t* __restrict__ num_chunks, WalkCounters* __restrict__ ctr, const uint32_t* __restrict__ rgbw,
    const uint32_t* __restrict__ sat, const AccOut& out, const RunOut& runs, const TileMap& tmap, uint32_t rec_stride,
    uint32_t* __restrict__ deferred, uint32_t* __restrict__ ndeferred, const uint32_t tile, const bool listed_other,
    const GridSrc* __restrict__ grid_arg) {
  // (kGrid a template parameter: the point-stream instances keep the registers they had before round 5)
  const GridSrc* __restrict__ const grid = kGrid ? grid_arg : nullptr;
  constexpr int kPer = E / kWalkRays;
  constexpr int kLimit = E * 7 / 8;          // entries a tile may use (kWalkLimit of the general kernel's table)
  constexpr int kMaskCapE = E / 4;           // ray masks built per round: the area of the accumulators
  __shared__ FastShared<E> S;
  __shared__ uint32_t raw[4 * E];                              // accumulators during the walk, ray masks afterwards
  __shared__ uint16_t vlog[kLogLen * kWalkRays];        // entry of visit k of ray r at [k * kWalkRays + r]
  // mask index of the entry (0xFFFF: none) — in the words of the (chunk, slab) counters, dead once the records are out,
  // where they hold it
  constexpr bool kMidxOverlay = sizeof(S.ccnt) >= E * sizeof(uint16_t);
  __shared__ uint16_t e_midx_own[kMidxOverlay ? 1 : E];
  uint16_t* const e_midx = kMidxOverlay ? reinterpret_cast<uint16_t*>(S.ccnt) : e_midx_own;
  int32_t* const e_wuu = reinterpret_cast<int32_t*>(raw);                                 // sum of w_u * u, fixed point
  unsigned long long* const e_wc = reinterpret_cast<unsigned long long*>(raw + E);   // visits << 32 | sum of w_u
  uint32_t* const e_last = raw + 3 * E;                                             // last visiting ray
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // (tile = the local tile: output regions; listed_other = a listed tile that is not an overflow: passed on)
  const uint32_t gtile = tmap.tile_of(tile);             // its place in the point stream
  const GridTile gt = grid ? grid_tile(*grid, gtile) : GridTile{0u, 0u, 0u};
  const TileSpan span = grid ? TileSpan{(int)gt.cloud, 0u, (uint32_t)kWalkRays}   // (a block of grid pixels of one image)
                      : tmap.tables ? tile_span_tables(offsets, nclouds, gtile, kWalkRays)   // (512 points of one cloud:
                                    : tile_span(offsets, nclouds, gtile, kWalkRays);          //  tsdf_directory.hpp)
  const uint32_t first = span.first, nrays = span.nrays;
  const int cloud = span.cloud;
  bool defer = P.shard_count > 1 || listed_other;   // (uniform)
  WALK_PROF_BEGIN();
  uint32_t nv = 0;
  uint32_t rid = (uint32_t)threadIdx.x;   // the ray of the tile this thread walks
  bool walks = false;
  int ox = 0, oy = 0, oz = 0;
  if (!defer) {
    subtile_reset(S, tid);
#pragma unroll
    for (int k = 0; k < kWalkChunks * kSlabs / kWalkRays; ++k) S.rcnt[tid + k * kWalkRays] = 0u;
#pragma unroll
    for (int k = 0; k < 4 * kPer; ++k) raw[tid + k * kWalkRays] = 0u;
    if (tid == 0) {
      S.run_total = 0;
      S.vis_total = 0;
      S.any_cold = 0;
    }
#if PLVS_WALK_SORT
    {
      // Rays of like length into the same wave: a wave runs its voxel loop as long as its longest ray, and the rays of a
      // tile (a strip of the image) differ by the depths they end at — the truncation band grows with the square of it.
      // Counting sort of the tile's rays by depth bucket (12.5 cm), in the visit log's memory (free until the walk); a
      // thread then walks ray `rid`, and everything that names a ray — the last visitor of an entry, the bits of the
      // ray masks, the point it reads — names it by rid, so the tile's outputs do not depend on the assignment.
      uint32_t* const bins = reinterpret_cast<uint32_t*>(vlog) + 256;   // 64 words behind the 512 u16 of the permutation
      if (tid < 64) bins[tid] = 0u;
      __syncthreads();
      int b = 63;
      uint32_t r = 0;
      if ((uint32_t)tid < nrays) {
        const float z = grid ? 0.0f : xyz[3 * (size_t)(first + (uint32_t)tid) + 2];   // (the switch is a cloud-mode experiment)
        b = z < 0.01f ? 63 : min(62, (int)(z * 8.0f));
        r = atomicAdd(&bins[b], 1u);
      }
      __syncthreads();
      const uint32_t c = bins[lane];
      const uint32_t excl = wave_scan_incl(c) - c;   // (every wave scans the 64 bins for itself)
      const uint32_t start = (uint32_t)__shfl((int)excl, b);
      if ((uint32_t)tid < nrays) vlog[start + r] = (uint16_t)tid;   // (the permutation lies in front of the bins)
      __syncthreads();
      if ((uint32_t)tid < nrays) rid = (uint32_t)vlog[tid];          // (a thread's first log entry is this very slot)
    }
#endif
    const Pose& pose = poses[cloud];   // (uniform address: scalar loads where it is used)
    Ray ray;
    walks = (uint32_t)tid < nrays && tile_ray_src(P, xyz, grid, gt, pose, first, rid, &ray, &ctr->err);
    const float wu = walks ? P.weight / (2.0f * ray.truncation) : 0.0f;
    {   // origin of the voxel keys: below the start voxel of the first walking ray, on a chunk boundary
      const unsigned long long wm = __ballot(walks);
      if (lane == 0) S.cand[wid] = wm ? 1u : 0u;
      if (wm && lane == __ffsll((long long)wm) - 1)
        for (int k = 0; k < 3; ++k) S.worg[wid][k] = origin_of((int)floorf(ray.start[k]));
    }
    WALK_PROF(0);   // set-up: the ray, the origin vote
    __syncthreads();                                                                        // ---- barrier 1
    WALK_PROF(1);   // (wait)
    {
      int w0 = 0;
#pragma unroll
      for (int w = kWalkRays / 64 - 1; w >= 0; --w) w0 = S.cand[w] ? w : w0;
      ox = S.worg[w0][0]; oy = S.worg[w0][1]; oz = S.worg[w0][2];
      // no ray at all (a block of grid pixels without a valid depth: sky, a wall beyond max_depth, a hole — a sixth of the
      // tiles of an office stream): the tile is over, three barriers and a flush of nothing earlier
      uint32_t anyc = 0;
#pragma unroll
      for (int w = 0; w < kWalkRays / 64; ++w) anyc |= S.cand[w];
      if (anyc == 0u) {   // (uniform: every thread reads the same words)
        if (tid == 0) {
          out.seg_cnt[tile] = 0;
          runs.run_cnt[tile] = 0;
          out.tile_visits[tile] = 0;
        }
        return;
      }
    }
    // ---- does the ray's whole box (start voxel +- its reach) lie inside the key box?
    bool fits = false;
    if (walks) {
      const int reach = lean_reach(ray);
      const int rx = (int)floorf(ray.start[0]) - ox, ry = (int)floorf(ray.start[1]) - oy, rz = (int)floorf(ray.start[2]) - oz;
      fits = min(min(rx, ry), rz) - reach >= 0 && max(max(rx, ry), rz) + reach <= 1023;
      if (!fits) S.overflow = 1u;   // the rays of the tile are too far apart
    }
    if (fits) {
#if !(PLVS_WALK_EXP & 32)   // (bit 32, timing experiment: no voxel loop)
      nv = walk_lean<true, true>(P, pose, ray, S, ox, oy, oz, tid, wu * scale_u, (uint32_t)__float2int_rn(wu * scale_w), e_wuu,
                                 e_wc, e_last, vlog, rid);
#endif
    }
  }
#if (PLVS_WALK_EXP & 16)   // (bit 16, timing experiment: no flush)
  if (tid == 0) { out.seg_cnt[tile] = 0; runs.run_cnt[tile] = 0; out.tile_visits[tile] = nv; }
  return;
#endif
  uint32_t ekey[kPer];
  WALK_PROF(2);     // the voxel loop of this wave
  if (!defer) {   // entries in use (nobody counts them during the walk)
    __syncthreads();                                                                        // ---- barrier 2
    WALK_PROF(3);   // (wait for the slowest wave)
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      ekey[k] = S.ekey[tid + k * kWalkRays];
      if (ekey[k] != kKeyEmpty) ekey[k] = voxel_key(ekey[k]);   // (the table holds table keys)
    }
    uint32_t wave_n = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) wave_n += (uint32_t)__popcll(__ballot(ekey[k] != kKeyEmpty));
    if (lane == 0 && wave_n) atomicAdd(&S.nent, wave_n);
  } else {
#pragma unroll
    for (int k = 0; k < kPer; ++k) ekey[k] = kKeyEmpty;
  }
  // ---- entries -> chunks -> pool slots, ranks inside the (chunk, slab) groups, colour weights: no barrier in between.
  // The chunk cache hands every thread the same index for a chunk whoever inserts it first; nearly every chunk's
  // slot is already there (the walk's set-up asked the directory for the chunks its rays start and end in); a thread
  // whose entry lies in a chunk still without one — a chunk a ray only passes through, one not at its home entry of
  // the directory, a first touch — asks the directory itself (find or insert: a chunk joins the map when one of its
  // voxels takes an update).  Whether the tile stands at all (table, key box and cache held everything) is only known
  // at the next barrier: a tile that does not is deferred there — all it has done to global memory by then is to
  // enter chunks its voxels need anyway.
  int ci[kPer];
  uint32_t rank[kPer], rrank[kPer], vkey[kPer], elast[kPer];
  unsigned long long ewc[kPer];
  uint32_t need = 0;              // entries of this thread that leave the tile as runs
  bool multi = false;             // ... one of them visited by more than one ray
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    ci[k] = -2;
    rank[k] = 0;
    rrank[k] = 0;
    vkey[k] = 0xFFFFFFFFu;        // (no entry / no pool slot: the directory is full)
    elast[k] = 0;
    ewc[k] = 0;
  }
  if (!defer) {
    // (three passes over the thread's entries instead of one: the colour-weight words of ALL of them are on their way
    // from global memory while the ranks are taken — one pass paid a global round trip per entry, a fifth of the tile)
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      if (ekey[k] == kKeyEmpty) continue;
      bool won;
      ci[k] = chunk_cache_insert(S, chunk_code(ekey[k]), &won);
      if (ci[k] < 0) {
        S.overflow = 1u;   // more chunks than the cache holds
        continue;
      }
      int slot = S.cslot[ci[k]];
      if (slot < 0) {
        int cx, cy, cz;
        chunk_of_code(chunk_code(ekey[k]), ox, oy, oz, &cx, &cy, &cz);
        slot = dir_find_or_insert(dir, cx, cy, cz, num_chunks, &ctr->err);
        if (slot >= 0) S.cslot[ci[k]] = slot;
      }
      if (slot < 0) continue;
      vkey[k] = (uint32_t)slot * (uint32_t)kChunkVox + voxel_in_chunk(ekey[k]);
    }
    uint32_t cw[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      cw[k] = 0u;
      if (vkey[k] != 0xFFFFFFFFu) cw[k] = sat ? sat[vkey[k] >> 5] : rgbw[vkey[k]];
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      if (vkey[k] == 0xFFFFFFFFu) continue;
      const uint32_t vid = vkey[k] % (uint32_t)kChunkVox;
      ewc[k] = e_wc[tid + k * kWalkRays];
      elast[k] = e_last[tid + k * kWalkRays];
      rank[k] = atomicAdd(&S.ccnt[ci[k] * kSlabs + (int)(vid / kSlabVox)], 1u);
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      if (vkey[k] == 0xFFFFFFFFu) continue;
      const bool cold = sat ? ((cw[k] >> (vkey[k] & 31u)) & 1u) == 0u : (cw[k] >> 24) < 254u;
      if (cold) {   // its colour still depends on the order of the visits
        need |= 1u << k;
        multi = multi || (uint32_t)(ewc[k] >> 32) > 1u;
        // (its place among the runs of its (chunk, slab) group — the runs of a tile leave it grouped like its records)
        rrank[k] = atomicAdd(&S.rcnt[ci[k] * kSlabs + (int)((vkey[k] % (uint32_t)kChunkVox) / kSlabVox)], 1u);
      }
    }
  }
  // (one barrier: the ranks are complete, the tile's entries are counted — and does any voxel that needs a run have
  // more than one visiting ray?)
  WALK_PROF(4);     // entries -> chunks, ranks, colour weights
  // (... and does any voxel need a run at all?  A tile of a saturated map has none: no scan of the run groups.  A flag in
  // LDS: __syncthreads_or returns a truth value, not the OR of its arguments)
  if (need) S.any_cold = 1u;
  const bool any_multi = __syncthreads_or(multi ? 1 : 0) != 0;                              // ---- barrier 3
  const bool any_cold = S.any_cold != 0u;
  WALK_PROF(5);     // (wait)
  const bool too_many = !defer && S.nent > (uint32_t)kLimit;   // (the voxels do not fit: the next kernel's table is larger / walk_tiles cuts the tile at once)
  if (E == 2048 && tid == 0 && !defer && S.nent > 1024u * 7u / 8u) atomicAdd(&ctr->over_small, 1u);   // (the host's choice of the next call's table)
  if (!defer) defer = S.overflow != 0 || too_many;
  if (defer) {
    if (tid == 0) {
      out.seg_cnt[tile] = 0;
      runs.run_cnt[tile] = 0;
      out.tile_visits[tile] = 0;
      deferred[atomicAdd(ndeferred, 1u)] = tile | (too_many ? 0x80000000u : 0u);
    }
    return;
  }
  {   // visits of the tile
    const uint32_t v = wave_sum(nv);
    if (lane == 0 && v) atomicAdd(&S.vis_total, v);
  }
  // ---- records: the (chunk, slab) groups are placed by a scan over the cache's 64 chunks, one lane per chunk.  EVERY
  // wave runs the scan and writes the same table of group bases — a wave reads back its own LDS writes in order, so
  // nobody waits for wave 0 behind a barrier; wave 0 alone writes the segment descriptors.
  const uint32_t rbase = tile * rec_stride;
  uint32_t nruns;   // runs of the tile (every wave computes it)
  {
    uint32_t sub[kSlabs], c = 0, rsub[kSlabs], rc = 0;
#pragma unroll
    for (int s = 0; s < kSlabs; ++s) {
      sub[s] = c;
      c += S.ccnt[lane * kSlabs + s];
      rsub[s] = 0;
    }
    uint32_t rinc = 0;
    nruns = 0;
    if (any_cold) {   // (uniform)
#pragma unroll
      for (int s = 0; s < kSlabs; ++s) {
        rsub[s] = rc;
        rc += S.rcnt[lane * kSlabs + s];
      }
      rinc = wave_scan_incl(rc);
      nruns = (uint32_t)__builtin_amdgcn_readlane((int)rinc, 63);
#pragma unroll
      for (int s = 0; s < kSlabs; ++s) S.rcbase[lane * kSlabs + s] = (uint16_t)(rinc - rc + rsub[s]);
    }
    // (records and segments in one scan: a chunk holds at most 1024 entries of the tile)
    const uint32_t both = wave_scan_incl(c | (c ? 1u << 16 : 0u));
    const uint32_t inc = both & 0xFFFFu, sinc = both >> 16;
#pragma unroll
    for (int s = 0; s < kSlabs; ++s) S.cbase[lane * kSlabs + s] = (uint16_t)(inc - c + sub[s]);
    if (wid == 0) {
      const uint32_t stot = (uint32_t)__builtin_amdgcn_readlane((int)sinc, 63);
      if (lane == 0) out.seg_cnt[tile] = stot;
      if (c) {
        const uint32_t sg = tile * (uint32_t)kWalkChunks + sinc - 1u;
        out.seg[2 * (size_t)sg] = make_uint4((uint32_t)S.cslot[lane], rbase + inc - c, c, gtile);
        out.seg[2 * (size_t)sg + 1] = pack_suboffsets(sub);
        if (runs.rseg) {
          runs.rseg[2 * (size_t)sg] = make_uint4((tile << runs.r1_log2) + rinc - rc, rc, 0u, 0u);
          runs.rseg[2 * (size_t)sg + 1] = pack_suboffsets(rsub);
        }
        if (out.chunk_nseg) atomicAdd(&out.chunk_nseg[S.cslot[lane]], 1u);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    if (vkey[k] == 0xFFFFFFFFu) continue;
    const uint32_t vid = vkey[k] % (uint32_t)kChunkVox;
    const uint4 r = make_uint4(vid | ((uint32_t)(ewc[k] >> 32) << 12),
                               grid ? grid_order_key(*grid, gt, elast[k]) : first + elast[k],
                               (uint32_t)e_wuu[tid + k * kWalkRays], (uint32_t)ewc[k]);
    out.rec[rbase + S.cbase[ci[k] * kSlabs + (int)(vid / kSlabVox)] + rank[k]] = r;
  }
  WALK_PROF(6);     // records
  // ---- runs: an entry that needs one has its place: the first run of its (chunk, slab) group + its rank in it
  const bool fits_runs = nruns <= (1u << runs.r1_log2);
  if (!fits_runs && tid == 0) {
    atomicOr(&ctr->err, kErrScratch);
    atomicMax(&ctr->run_need, nruns);
  }
  uint32_t mrun[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k)
    mrun[k] = (need & (1u << k)) ? (uint32_t)S.rcbase[ci[k] * kSlabs + (int)((vkey[k] % (uint32_t)kChunkVox) / kSlabVox)] + rrank[k] : 0xFFFFu;
  if (!any_multi) {
    // every voxel that needs a run has ONE visiting ray — the steady state of a map whose colours have mostly
    // saturated: what is still below 254 is what few rays reach — and the accumulator named it (e_last): the mask is
    // that one bit, no pass over the visit logs, no barrier.
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      if (!(need & (1u << k)) || !fits_runs) continue;
      const size_t d = ((size_t)tile << runs.r1_log2) + mrun[k];
      runs.dkey[d] = vkey[k];
      uint4* dst = reinterpret_cast<uint4*>(runs.masks + d * kMaskWords);
      const uint32_t word = elast[k] >> 5, bit = 1u << (elast[k] & 31u);
#pragma unroll
      for (int q = 0; q < kMaskWords / 4; ++q)
        dst[q] = make_uint4(word == 4u * q ? bit : 0u, word == 4u * q + 1u ? bit : 0u, word == 4u * q + 2u ? bit : 0u,
                            word == 4u * q + 3u ? bit : 0u);
    }
  } else {
    // the masks come from the visit logs: bit r of a voxel's mask = ray r of the tile visits it
    __syncthreads();   // (the records are out: the accumulator area and the words of the (chunk, slab) counters are free)
#pragma unroll
    for (int k = 0; k < kPer; ++k) e_midx[tid + k * kWalkRays] = (uint16_t)mrun[k];
    for (uint32_t r0 = 0; fits_runs && r0 < nruns; r0 += kMaskCapE) {
      __syncthreads();   // e_midx complete / the previous round's masks are out
#pragma unroll
      for (int k = 0; k < kMaskCapE * kMaskWords / kWalkRays; ++k) raw[tid + k * kWalkRays] = 0u;
      __syncthreads();
      if (nv) {
        const uint32_t logged = min(nv, (uint32_t)kLogLen);
        for (uint32_t k = 0; k < logged; ++k) {
          const uint32_t m = (uint32_t)e_midx[vlog[k * kWalkRays + tid]] - r0;   // 0xFFFF - r0 stays out of range
          if (m < (uint32_t)kMaskCapE) atomicOr(&raw[m * kMaskWords + (rid >> 5)], 1u << (rid & 31));
        }
        if (nv > (uint32_t)kLogLen) {   // the log is full: the rest of the ray is walked again
          const Pose& pose = poses[cloud];
          Ray ray;
          if (tile_ray_src(P, xyz, grid, gt, pose, first, rid, &ray, &ctr->err))
            walk_one(P, pose, ray, 0u, 0xFFFFFFFFu, [&](uint32_t k, int vx, int vy, int vz, float) {
              if (k >= (uint32_t)kLogLen) {
                uint32_t key;
                const int e = rel_key(vx, vy, vz, ox, oy, oz, &key) ? table_find(S, table_key(key)) : -1;
                const uint32_t m = e >= 0 ? (uint32_t)e_midx[e] - r0 : 0xFFFFFFFFu;
                if (m < (uint32_t)kMaskCapE) atomicOr(&raw[m * kMaskWords + (rid >> 5)], 1u << (rid & 31));
              }
              return true;
            });
        }
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        const uint32_t m = (uint32_t)e_midx[tid + k * kWalkRays];
        if (m == 0xFFFFu || m < r0 || m >= r0 + (uint32_t)kMaskCapE) continue;
        const size_t d = ((size_t)tile << runs.r1_log2) + m;
        runs.dkey[d] = vkey[k];
        const uint32_t* mk = raw + (m - r0) * kMaskWords;
        uint4* dst = reinterpret_cast<uint4*>(runs.masks + d * kMaskWords);
#pragma unroll
        for (int q = 0; q < kMaskWords / 4; ++q) dst[q] = make_uint4(mk[4 * q], mk[4 * q + 1], mk[4 * q + 2], mk[4 * q + 3]);
      }
    }
  }
  if (tid == 0) S.run_total = nruns;
  WALK_PROF(7);     // runs
  // ---- tile epilogue: run and visit counts
  __syncthreads();                                                                          // ---- barrier 4
  WALK_PROF(8);
  WALK_PROF_END();
  if (tid == 0) {
    runs.run_cnt[tile] = min(S.run_total, 1u << runs.r1_log2);
    out.tile_visits[tile] = S.vis_total;
  }
}

template <int E, bool kGrid = false>
__global__ __launch_bounds__(kWalkRays, walk_fast_waves(E)) void walk_fast(
    Params P, float scale_u, float scale_w, const float* __restrict__ xyz, int npoints,
    const int32_t* __restrict__ offsets, int nclouds, const Pose* __restrict__ poses, Directory dir,
    int32_t* __restrict__ num_chunks, WalkCounters* __restrict__ ctr, const uint32_t* __restrict__ rgbw,
    const uint32_t* __restrict__ sat, AccOut out, RunOut runs, TileMap tmap, uint32_t rec_stride,
    const uint32_t* __restrict__ tile_list, const uint32_t* __restrict__ ntile_list, uint32_t* __restrict__ deferred,
    uint32_t* __restrict__ ndeferred, const GridSrc* __restrict__ grid) {
  if (!tile_list) {
    walk_fast_tile<E, kGrid>(P, scale_u, scale_w, xyz, npoints, offsets, nclouds, poses, dir, num_chunks, ctr, rgbw, sat, out, runs,
                      tmap, rec_stride, deferred, ndeferred, blockIdx.x, false, grid);
    return;
  }
  // a list: the workgroups of a small grid take its entries in turn (a grid of one workgroup per POSSIBLE entry would
  // queue thousands of empty workgroups behind the one-per-CU limit of the large table's LDS)
  const uint32_t nlist = *ntile_list;
  for (uint32_t lb = blockIdx.x; lb < nlist; lb += gridDim.x) {
    const uint32_t e = tile_list[lb];
    walk_fast_tile<E, kGrid>(P, scale_u, scale_w, xyz, npoints, offsets, nclouds, poses, dir, num_chunks, ctr, rgbw, sat, out, runs,
                      tmap, rec_stride, deferred, ndeferred, e & 0x7FFFFFFFu, (e >> 31) == 0u, grid);
    __syncthreads();   // (the next tile reuses the shared state)
  }
}

// ------------------------------------------------------------------ segments -> chunk order
// Counting sort of the segment descriptors by chunk.  A workgroup takes kSegSpan descriptor slots
// (segments of neighbouring tiles: a handful of chunks), counts them per chunk in an LDS table and
// goes to the global per-chunk counters once per (workgroup, chunk): seg_pass<false> counts,
// seg_scan scans (and lists the updated chunks, sums the per-tile visit counts), seg_pass<true> places.
constexpr int kSegSpan = 1024;     // descriptor slots per workgroup (16 tiles)
constexpr int kSegSpanLong = 4096; // ... of seg_pass<true> in a long call (64 tiles: a quarter of the atomics on the chunks' counters)
constexpr uint32_t kPartSegs = 256;   // segments of a busy chunk one work item of the apply stage takes (default)
constexpr uint32_t kPartMin = 512;    // a chunk with more segments than this is applied in parts (default; round 6: 2048 -> 512, apply 0.18 -> 0.14 ms on the stream)
constexpr int kSegTable = 512;
template <bool kScatter, int kSpan = kSegSpan>
__global__ __launch_bounds__(256) void seg_pass(const uint4* __restrict__ seg, uint32_t seg_cap, uint32_t ntiles,
                                                const uint32_t* __restrict__ seg_cnt, uint32_t* __restrict__ chunk_nseg,
                                                const uint32_t* __restrict__ chunk_off, uint32_t* __restrict__ chunk_fill,
                                                uint4* __restrict__ sorted, const WalkCounters* __restrict__ ctr) {
  __shared__ uint32_t hkey[kSegTable], hcnt[kSegTable], hbase[kSegTable];
  const int tid = threadIdx.x;
  const uint32_t own = ntiles * (uint32_t)kWalkChunks;
  const uint32_t n = min(own + ctr->seg_top, seg_cap);
  for (int k = tid; k < kSegTable; k += 256) {
    hkey[k] = 0xFFFFFFFFu;
    hcnt[k] = 0;
  }
  __syncthreads();
  constexpr int kPer = kSpan / 256;
  int ent[kPer];
  uint32_t rnk[kPer], slot[kPer];
#pragma unroll
  for (int q = 0; q < kPer; ++q) {
    const uint32_t j = blockIdx.x * (uint32_t)kSpan + (uint32_t)(q * 256 + tid);
    ent[q] = -2;   // no segment here
    if (j >= n || (j < own && (j % kWalkChunks) >= seg_cnt[j / kWalkChunks])) continue;
    slot[q] = seg[2 * (size_t)j].x;
    uint32_t h = (slot[q] * 2654435761u) >> (32 - 9);
    ent[q] = -1;   // table full: straight to the global counters
    for (int probe = 0; probe < kSegTable; ++probe) {
      uint32_t cur = hkey[h];
      if (cur == 0xFFFFFFFFu) cur = atomicCAS(&hkey[h], 0xFFFFFFFFu, slot[q]);
      if (cur == 0xFFFFFFFFu || cur == slot[q]) {
        ent[q] = (int)h;
        break;
      }
      h = (h + 1) & (kSegTable - 1);
    }
    if (ent[q] >= 0) rnk[q] = atomicAdd(&hcnt[ent[q]], 1u);
  }
  __syncthreads();
  for (int k = tid; k < kSegTable; k += 256) {
    if (hkey[k] == 0xFFFFFFFFu) continue;
    if (kScatter) hbase[k] = chunk_off[hkey[k]] + atomicAdd(&chunk_fill[hkey[k]], hcnt[k]);
    else atomicAdd(&chunk_nseg[hkey[k]], hcnt[k]);
  }
  if (kScatter) __syncthreads();
#pragma unroll
  for (int q = 0; q < kPer; ++q) {
    if (ent[q] == -2) continue;
    const uint32_t j = blockIdx.x * (uint32_t)kSpan + (uint32_t)(q * 256 + tid);
    if (kScatter) {
      const uint32_t at = ent[q] >= 0 ? hbase[ent[q]] + rnk[q] : chunk_off[slot[q]] + atomicAdd(&chunk_fill[slot[q]], 1u);
      sorted[2 * (size_t)at] = seg[2 * (size_t)j];
      sorted[2 * (size_t)at + 1] = seg[2 * (size_t)j + 1];
    } else if (ent[q] == -1) {
      atomicAdd(&chunk_nseg[slot[q]], 1u);
    }
  }
}

__global__ __launch_bounds__(1024) void seg_scan(const uint32_t* __restrict__ chunk_nseg, uint32_t* __restrict__ chunk_off,
                                                 uint32_t* __restrict__ chunk_fill, uint32_t* __restrict__ active,
                                                 uint32_t* __restrict__ active_off, WalkCounters* __restrict__ ctr,
                                                 const int32_t* __restrict__ num_chunks, int max_chunks,
                                                 const uint32_t* __restrict__ tile_visits,
                                                 const uint32_t* __restrict__ run_cnt, uint32_t ntiles,
                                                 uint32_t* __restrict__ part_off, uint32_t* __restrict__ multi_idx,
                                                 uint32_t multi_cap, uint32_t part_segs, uint32_t part_min,
                                                 uint32_t* __restrict__ active_idx = nullptr) {
  // (active_idx, may be null: slot of an updated chunk -> its place in `active`)
  // part_off / multi_idx (per updated chunk, in `active` order): the chunk's first part in the apply stage's
  // item list (a part = part_segs segments of a chunk with more than part_min of them) and its index among the
  // chunks applied in parts
  __shared__ uint32_t wsum[16], wact[16], wprt[16], wmul[16];
  __shared__ uint32_t carry, acarry, pcarry, mcarry;
  __shared__ unsigned long long vsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n = min(*num_chunks, max_chunks);
  if (tid == 0) { carry = 0; acarry = 0; pcarry = 0; mcarry = 0; }
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int s = base + tid;
    const uint32_t c = s < n ? chunk_nseg[s] : 0u;
    const uint32_t a = c ? 1u : 0u;
    const uint32_t m = c > part_min ? 1u : 0u;
    uint32_t inc = c, ainc = a, minc = m;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)inc, off), aup = (uint32_t)__shfl_up((int)ainc, off);
      const uint32_t mup = (uint32_t)__shfl_up((int)minc, off);
      if (lane >= off) { inc += up; ainc += aup; minc += mup; }
    }
    if (lane == 63) { wsum[wid] = inc; wact[wid] = ainc; wmul[wid] = minc; }
    __syncthreads();
    uint32_t wb = carry, ab = acarry, mb = mcarry, tot = 0, atot = 0, mtot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      if (w < wid) { wb += wsum[w]; ab += wact[w]; mb += wmul[w]; }
      tot += wsum[w];
      atot += wact[w];
      mtot += wmul[w];
    }
    // a busy chunk beyond the accumulators the host has provided is applied in one part (slower, never wrong);
    // the host sees num_multi and provides more for the next call
    const uint32_t midx = mb + minc - 1u;
    const bool multi = m && midx < multi_cap;
    const uint32_t np = multi ? (c + part_segs - 1u) / part_segs : a;
    uint32_t pinc = np;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t pup = (uint32_t)__shfl_up((int)pinc, off);
      if (lane >= off) pinc += pup;
    }
    if (lane == 63) wprt[wid] = pinc;
    __syncthreads();
    uint32_t pb = pcarry, ptot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      if (w < wid) pb += wprt[w];
      ptot += wprt[w];
    }
    if (s < n) {
      chunk_off[s] = wb + inc - c;
      chunk_fill[s] = 0;
      if (c) {
        const uint32_t at = ab + ainc - 1u;
        active[at] = (uint32_t)s;
        if (active_idx) active_idx[s] = at;
        active_off[at] = wb + inc - c;
        part_off[at] = pb + pinc - np;
        multi_idx[at] = multi ? midx : 0xFFFFFFFFu;
      }
    }
    __syncthreads();
    if (tid == 0) { carry += tot; acarry += atot; pcarry += ptot; mcarry += mtot; }
    __syncthreads();
  }
  unsigned long long v = 0, nr = 0;
  for (uint32_t t = tid; t < ntiles; t += 1024) {
    v += tile_visits[t];
    nr += run_cnt ? run_cnt[t] : 0u;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    v += (unsigned long long)__shfl_xor((long long)v, off);
    nr += (unsigned long long)__shfl_xor((long long)nr, off);
  }
  __shared__ unsigned long long rsum[16];
  if (lane == 0) { vsum[wid] = v; rsum[wid] = nr; }
  __syncthreads();
  if (tid == 0) {
    unsigned long long tot = 0, rtot = 0;
    for (int w = 0; w < 16; ++w) { tot += vsum[w]; rtot += rsum[w]; }
    ctr->total_visits = tot;
    ctr->num_desc = (uint32_t)rtot;
    if (run_cnt) ctr[1].num_desc = (uint32_t)rtot;   // (the colour side's counters: what a scan of the run counts leaves there too)
    ctr->num_updated = acarry;
    active_off[acarry] = carry;
    part_off[acarry] = pcarry;
    ctr->num_parts = pcarry;
    ctr->num_multi = mcarry;
  }
}

// The apply stage ("LDS-staged blocks").  Work item = (updated chunk, part, slab of kSlabVox voxels): the records of
// the slab — contiguous inside every segment — are added into LDS accumulators (64-bit fixed point: the result does
// not depend on the order of the records).  A chunk normally is one part and takes its updates at once, ONE per
// visited voxel.  A chunk with more than part_min segments (an owner of the ray-sharded integrate collects the
// segments of every rank; a long batch over a small scene) is cut into parts of part_segs segments: the parts add
// their sums to the chunk's accumulators in global memory, the part that finishes last applies the update and leaves
// the accumulators zero for the next call.  Sixteen lanes share a segment, four segments per group are in flight.
struct PartAcc {           // per (chunk applied in parts, voxel); `done` per (such chunk, slab)
  long long* wuu;
  unsigned long long* w;
  uint32_t *last, *cnt, *done;
};
// kWide: the records are the 32-byte per-voxel sums a rank of the ray-sharded integrate sends to a chunk's owner,
//   {voxel | count << 12, last point, sum w_u*u (64 bit)}, {sum w_u (64 bit), 0, 0}.
// kEmit (a rank's own aggregation before the exchange): instead of updating the voxels, the sums of the slab's touched
//   voxels leave as wide records + one descriptor (chunk id, slab) in the send region of the chunk's owner.
struct EmitOut {
  const int32_t* slot_ids;      // walk directory: slot -> chunk id
  const uint32_t* owner;        // [active index] destination rank of the chunk
  const uint32_t* seg_region;   // [rank] first descriptor / first record of the rank's send region,
  const uint32_t* rec_region;
  uint32_t* seg_fill;           // [rank] descriptors / records written so far
  uint32_t* rec_fill;
  uint4* seg_out;
  uint4* rec_out;
};
constexpr int kApplyThreads = 512;
template <bool kWide, bool kEmit>
__global__ __launch_bounds__(kApplyThreads) void apply_chunks(
    const uint4* __restrict__ sorted_seg, const uint32_t* __restrict__ active, const uint32_t* __restrict__ active_off,
    const uint32_t* __restrict__ part_off, const uint32_t* __restrict__ multi_idx, uint32_t part_segs, PartAcc acc,
    const uint4* __restrict__ rec, double inv_scale_u, double inv_scale_w, const uint32_t* __restrict__ kfid_of_point,
    float* __restrict__ sdf, float* __restrict__ weight, uint32_t* __restrict__ vkfid, WalkCounters* __restrict__ ctr,
    EmitOut emit, uint32_t last_shift) {
  // last_shift: 0 = a record's last visitor is a point of the stream (kfid_of_point per point); otherwise it is the order
  // key of a depth-image call (GridSrc::key_bits) and kfid_of_point holds one id per image
  __shared__ long long a_wuu[kSlabVox];
  __shared__ unsigned long long a_w[kSlabVox];
  __shared__ uint32_t a_last[kSlabVox], a_cnt[kSlabVox];
  __shared__ uint32_t is_last;
  constexpr int kPieceBatch = 2 * kApplyThreads;      // segments listed at a time
  __shared__ uint32_t p_lo[kPieceBatch], p_pre[kPieceBatch], p_wave[kApplyThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63;
  if (ctr->err) return;   // the walk ran out of scratch: the host grows it and repeats the call, the map stays as it was
  const uint32_t nchunks = ctr->num_updated, nitems = ctr->num_parts * kSlabs;
  uint32_t voxels = 0, longest = 0;
  struct Rec { uint4 a, b; };
  auto load = [&](uint32_t r) {   // record r of the call
    Rec q;
    if (kWide) {
      q.a = rec[2 * (size_t)r];
      q.b = rec[2 * (size_t)r + 1];
    } else {
      q.a = rec[r];
      q.b = make_uint4(0u, 0u, 0u, 0u);
    }
    return q;
  };
  auto add = [&](const Rec& q) {
    const unsigned long long wuu = kWide ? ((unsigned long long)q.a.z | ((unsigned long long)q.a.w << 32))
                                         : (unsigned long long)(long long)(int32_t)q.a.z;
    const unsigned long long w = kWide ? ((unsigned long long)q.b.x | ((unsigned long long)q.b.y << 32))
                                       : (unsigned long long)q.a.w;
    const uint32_t v = (q.a.x & 0xFFFu) % kSlabVox;
    atomicAdd((unsigned long long*)&a_wuu[v], wuu);
    atomicAdd(&a_w[v], w);
    atomicMax(&a_last[v], q.a.y);
    atomicAdd(&a_cnt[v], q.a.x >> 12);
  };
#ifdef PLVS_WALK_PROF
  long long ap_t = (long long)clock64();
  unsigned long long ap_acc[5] = {0, 0, 0, 0, 0}, ap_items = 0, ap_max = 0;
#define APPLY_PROF(k) do { const long long now = (long long)clock64(); ap_acc[k] += (unsigned long long)(now - ap_t); ap_t = now; } while (0)
#else
#define APPLY_PROF(k) do {} while (0)
#endif
  for (uint32_t item = blockIdx.x; item < nitems; item += gridDim.x) {
#ifdef PLVS_WALK_PROF
    const long long ap_item0 = (long long)clock64();
    ap_t = ap_item0;
#endif
    const uint32_t pi = item / kSlabs, slab = item % kSlabs;
    // the chunk of part pi: the last a with part_off[a] <= pi — a 64-way search by every wave (two dependent loads for
    // up to 4096 updated chunks; a binary search is a dozen, and an item is short)
    uint32_t lo = 0, span = nchunks;
    while (span > 1) {
      const uint32_t step = (span + 63u) >> 6, at = lo + (uint32_t)lane * step;
      const unsigned long long le = __ballot(at < lo + span && part_off[at] <= pi);   // (lane 0 always: part_off[lo] <= pi)
      const uint32_t k = 63u - (uint32_t)__clzll((long long)le), end = lo + span;
      lo += k * step;
      span = min(step, end - lo);
    }
    const uint32_t a = lo, nparts = part_off[a + 1] - part_off[a];
    for (int v = tid; v < kSlabVox; v += kApplyThreads) {
      a_wuu[v] = 0;
      a_w[v] = 0;
      a_last[v] = 0;
      a_cnt[v] = 0;
    }
    __syncthreads();
    APPLY_PROF(0);   // item set-up
    const uint32_t s0 = active_off[a] + (pi - part_off[a]) * part_segs;
    const uint32_t s1 = nparts > 1 ? min(active_off[a + 1], s0 + part_segs) : active_off[a + 1];
    // The slab's records of every segment form one PIECE (contiguous; 0 to 512 records — a tile on a wall puts hundreds of
    // voxels into one slab of one chunk, a tile that grazes the chunk a handful).  Round 5, late: the pieces are listed in LDS
    // (first record, running total) and the workgroup then runs over ALL their records as one flat range, four loads per
    // thread in flight, every lane busy whatever the sizes of the pieces.  (Before: sixteen lanes per segment, four segments
    // per group in flight; a wave iterated as long as the LONGEST of its sixteen pieces, two dependent loads per trip — the
    // item of a hot chunk, 1 000-2 000 segments, took 0.3 ms, 95 % of it in that loop: the whole length of the kernel.)
    for (uint32_t sb0 = s0; sb0 < s1; sb0 += (uint32_t)kPieceBatch) {
      const uint32_t nb = min((uint32_t)kPieceBatch, s1 - sb0);
      uint32_t len[kPieceBatch / kApplyThreads], lo_p[kPieceBatch / kApplyThreads], mine = 0;
#pragma unroll
      for (int k = 0; k < kPieceBatch / kApplyThreads; ++k) {
        const uint32_t i = (uint32_t)tid * (kPieceBatch / kApplyThreads) + k;   // (consecutive pieces per thread: one scan)
        len[k] = lo_p[k] = 0;
        if (i < nb) {
          const uint4 d0 = sorted_seg[2 * (size_t)(sb0 + i)], d1 = sorted_seg[2 * (size_t)(sb0 + i) + 1];
          const uint32_t w0 = slab < 2 ? d1.x : slab < 4 ? d1.y : slab < 6 ? d1.z : d1.w;
          const uint32_t w1 = slab + 1 < 2 ? d1.x : slab + 1 < 4 ? d1.y : slab + 1 < 6 ? d1.z : d1.w;
          const uint32_t o = (w0 >> ((slab & 1) * 16)) & 0xFFFFu;
          const uint32_t e = slab + 1 < kSlabs ? (w1 >> (((slab + 1) & 1) * 16)) & 0xFFFFu : d0.z;
          lo_p[k] = d0.y + o;
          len[k] = e - o;
        }
        mine += len[k];
      }
      // exclusive prefix of the pieces' lengths over the workgroup
      uint32_t inc = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)inc, off);
        if (lane >= off) inc += up;
      }
      if (lane == 63) p_wave[tid >> 6] = inc;
      __syncthreads();
      uint32_t wbase = 0, total = 0;
#pragma unroll
      for (int w = 0; w < kApplyThreads / 64; ++w) {
        if (w < (tid >> 6)) wbase += p_wave[w];
        total += p_wave[w];
      }
      uint32_t run = wbase + inc - mine;
#pragma unroll
      for (int k = 0; k < kPieceBatch / kApplyThreads; ++k) {
        const uint32_t i = (uint32_t)tid * (kPieceBatch / kApplyThreads) + k;
        p_lo[i] = lo_p[k];
        p_pre[i] = run;       // (pieces behind nb: length 0, prefix = total)
        run += len[k];
      }
      __syncthreads();
      auto record_of = [&](uint32_t f) -> uint32_t {   // flat index -> record: the last piece with p_pre <= f
        uint32_t lo = 0, hi = nb;                       // (empty pieces share a prefix with their successor: the LAST one wins,
        while (hi - lo > 1) {                           //  and it is the non-empty one or followed only by empty ones... see below)
          const uint32_t mid = (lo + hi) >> 1;
          if (p_pre[mid] <= f) lo = mid; else hi = mid;
        }
        return p_lo[lo] + (f - p_pre[lo]);
      };
      // (the search returns the LAST piece whose prefix is <= f: of a run of pieces with the same prefix — all but the last
      // empty — that is the non-empty one, the only one f can lie in)
      constexpr int kFlat = 4;
      for (uint32_t f0 = (uint32_t)tid; f0 < total; f0 += (uint32_t)(kFlat * kApplyThreads)) {
        Rec q[kFlat];
#pragma unroll
        for (int j = 0; j < kFlat; ++j) {
          const uint32_t f = f0 + (uint32_t)(j * kApplyThreads);
          if (f < total) q[j] = load(record_of(f));
        }
#pragma unroll
        for (int j = 0; j < kFlat; ++j)
          if (f0 + (uint32_t)(j * kApplyThreads) < total) add(q[j]);
      }
      __syncthreads();   // (p_lo / p_pre are rewritten by the next batch)
    }
    APPLY_PROF(1);   // thread 0's share of the segments
    __syncthreads();
    APPLY_PROF(2);   // ... and the wait for the slowest wave
    const size_t pool0 = (size_t)active[a] * kChunkVox + (size_t)slab * kSlabVox;
    bool apply = true;
    if (nparts > 1) {
      // ---- this part's sums join the chunk's; the last part to arrive applies them.  Returning atomics: once their
      // results are back they have been performed, so the ticket below is taken after them without a release fence
      // (which would write the whole L2 back).
      const size_t g0 = ((size_t)multi_idx[a] * kSlabs + slab) * kSlabVox;
      uint32_t seen = 0;
      for (int v = tid; v < kSlabVox; v += kApplyThreads) {
        if (a_cnt[v] == 0) continue;
        seen += (uint32_t)atomicAdd((unsigned long long*)&acc.wuu[g0 + v], (unsigned long long)a_wuu[v]);
        seen += (uint32_t)atomicAdd(&acc.w[g0 + v], a_w[v]);
        seen += atomicMax(&acc.last[g0 + v], a_last[v]);
        seen += atomicAdd(&acc.cnt[g0 + v], a_cnt[v]);
      }
      if (seen == 0x9E3779B9u) a_last[tid % kSlabVox] = seen;   // (keeps the results alive; never the point)
      __syncthreads();
      if (tid == 0) is_last = atomicAdd(&acc.done[(size_t)multi_idx[a] * kSlabs + slab], 1u) == nparts - 1u ? 1u : 0u;
      __syncthreads();
      apply = is_last != 0u;
      if (apply) {   // (read-modify-write reads: performed where the other parts' atomics were)
        for (int v = tid; v < kSlabVox; v += kApplyThreads) {
          a_cnt[v] = atomicExch(&acc.cnt[g0 + v], 0u);
          if (a_cnt[v]) {
            a_wuu[v] = (long long)atomicExch((unsigned long long*)&acc.wuu[g0 + v], 0ull);
            a_w[v] = atomicExch(&acc.w[g0 + v], 0ull);
            a_last[v] = atomicExch(&acc.last[g0 + v], 0u);
          }
        }
        if (tid == 0) atomicExch(&acc.done[(size_t)multi_idx[a] * kSlabs + slab], 0u);
      }
    }
    APPLY_PROF(3);   // parts: the sums join the chunk's accumulators
    if (apply && kEmit) {
      // ---- the slab's sums leave for the chunk's owner: one wide record per touched voxel, one descriptor
      static_assert(kSlabVox == kApplyThreads, "a thread per voxel of the slab");
      const uint32_t c = a_cnt[tid];
      const unsigned long long touched = __ballot(c != 0u);
      __shared__ uint32_t wtot[kApplyThreads / 64];
      __shared__ uint32_t ebase;
      if (lane == 0) wtot[tid >> 6] = (uint32_t)__popcll(touched);
      __syncthreads();
      uint32_t before = 0, total = 0;
#pragma unroll
      for (int w = 0; w < kApplyThreads / 64; ++w) {
        if (w < (tid >> 6)) before += wtot[w];
        total += wtot[w];
      }
      const uint32_t p = emit.owner[a];
      if (tid == 0 && total) {
        ebase = atomicAdd(&emit.rec_fill[p], total);
        const uint32_t sg = emit.seg_region[p] + atomicAdd(&emit.seg_fill[p], 1u);
        const int32_t* id = emit.slot_ids + 3 * (size_t)active[a];
        unsigned long long key = 0;
        pack_block(id[0], id[1], id[2], &key);
        uint32_t sub[kSlabs];
#pragma unroll
        for (int k = 0; k < kSlabs; ++k) sub[k] = (uint32_t)k > slab ? total : 0u;
        const uint4 so = pack_suboffsets(sub);
        emit.seg_out[2 * (size_t)sg] = make_uint4((uint32_t)key, ebase, (uint32_t)(key >> 32), slab);
        emit.seg_out[2 * (size_t)sg + 1] = make_uint4(so.x | total, so.y, so.z, so.w);   // (offset 0 is 0: its bits carry the count)
      }
      __syncthreads();
      if (c) {
        const uint32_t at = emit.rec_region[p] + ebase + before + (uint32_t)__popcll(touched & ((1ull << lane) - 1ull));
        const unsigned long long wuu = (unsigned long long)a_wuu[tid], w = a_w[tid];
        emit.rec_out[2 * (size_t)at] = make_uint4((slab * (uint32_t)kSlabVox + (uint32_t)tid) | (c << 12), a_last[tid],
                                                   (uint32_t)wuu, (uint32_t)(wuu >> 32));
        emit.rec_out[2 * (size_t)at + 1] = make_uint4((uint32_t)w, (uint32_t)(w >> 32), 0u, 0u);
        ++voxels;
      }
    } else if (apply) {
      for (int v = tid; v < kSlabVox; v += kApplyThreads) {
        const uint32_t c = a_cnt[v];
        if (c) {
          const float m = (float)((double)a_wuu[v] * inv_scale_u), ws = (float)((double)a_w[v] * inv_scale_w);
          const float W = weight[pool0 + v], Sd = sdf[pool0 + v];
          const float wn = W + ws;
          sdf[pool0 + v] = (W * Sd + m) / wn;
          weight[pool0 + v] = wn;
          vkfid[pool0 + v] = kfid_of_point ? kfid_of_point[a_last[v] >> last_shift] : 0u;
          ++voxels;
          longest = max(longest, c);
        }
      }
    }
    __syncthreads();
    APPLY_PROF(4);   // the voxel updates
#ifdef PLVS_WALK_PROF
    if (threadIdx.x == 0 && item < 8192u) {
      g_apply_items[item][0] = (unsigned long long)((long long)clock64() - ap_item0);
      g_apply_items[item][1] = (unsigned long long)(s1 - s0) << 32;
      g_apply_items[item][2] = 0;
      g_apply_items[item][3] = 0;
    }
    ++ap_items;
    ap_max = max(ap_max, ((unsigned long long)((long long)clock64() - ap_item0) << 24) | ((unsigned long long)min(s1 - s0, 0xFFFFFu) << 4) |
                             (unsigned long long)min(nparts, 15u));   // cycles | segments of the item | parts of its chunk
#endif
  }
#ifdef PLVS_WALK_PROF
  if (threadIdx.x == 0) {
    for (int k = 0; k < 5; ++k) atomicAdd(&g_walk_prof[9 + k], ap_acc[k]);
    atomicAdd(&g_walk_prof[14], ap_items);
    atomicMax(&g_walk_prof[15], ap_max);
  }
#endif
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    voxels += (uint32_t)__shfl_xor((int)voxels, off);
    longest = max(longest, (uint32_t)__shfl_xor((int)longest, off));
  }
  // one update of the call's counters per workgroup, and the maximum only when it would move (thousands of waves
  // bumping the same two words are serviced one after the other at the memory side)
  __shared__ uint32_t w_vox[kApplyThreads / 64], w_long[kApplyThreads / 64];
  if (lane == 0) {
    w_vox[tid >> 6] = voxels;
    w_long[tid >> 6] = longest;
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t v = 0, l = 0;
#pragma unroll
    for (int w = 0; w < kApplyThreads / 64; ++w) {
      v += w_vox[w];
      l = max(l, w_long[w]);
    }
    if (v) atomicAdd(&ctr->num_heads, v);
    if (l > ctr->max_run) atomicMax(&ctr->max_run, l);
  }
}

// Self-test of sqrt_rn_normal / div_rn_normal against the compiler's sqrtf and `/` (GPU test): pseudo-random
// operands over the ranges the walk meets, plus every significand at a few exponents for the square root.
__global__ void selftest_walk_math_kernel(uint32_t seed, uint32_t* __restrict__ mismatches) {
  uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + seed;
  for (int it = 0; it < 64; ++it) {
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    const uint32_t ma = x & 0x7FFFFFu;
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    const uint32_t mb = x & 0x7FFFFFu;
    const int ea = (int)((x >> 23) & 31u) - 16, eb = (int)((x >> 28) & 15u) - 8;
    const float a = __uint_as_float(((uint32_t)(ea + 127) << 23) | ma);
    const float b = __uint_as_float(((uint32_t)(eb + 127) << 23) | mb | ((x >> 27) & 1u ? 0x80000000u : 0u));
    if (__float_as_uint(sqrt_rn_normal(a)) != __float_as_uint(sqrtf(a))) atomicAdd(&mismatches[0], 1u);
    if (__float_as_uint(div_rn_normal(a, b)) != __float_as_uint(a / b)) atomicAdd(&mismatches[1], 1u);
  }
}

// Runs of the per-tile regions -> dense (key, slot) pairs in tile order (the input of the stable sort
// by voxel key).  One wave per tile.
// A call that launches its colour chain BEFORE the host knows the number of runs (small calls: the host read in the middle
// of the chain was a fifth of their time; since the end of round 5 every call whose predecessor left a count) sorts at most
// `limit` pairs, a bound taken from the call before: the sort and the kernels behind it read the real number from the device
// (radix_sort_pairs_bound; `pad`: the one-launch sort of a small bound takes exactly `limit` pairs instead, the surplus filled
// with keys that sort last), and one extra workgroup says in *skip whether the bounds held — the runs fit `limit`, the map's
// chunks `chunk_limit` (the key bits of the sort).  If not, the fold does nothing and the host repeats the chain with the real
// numbers.  Other calls: limit = 0xFFFFFFFF, no extra workgroups.
struct RunGuard {
  uint32_t limit;
  const uint32_t* total;        // runs of the call (the scan of run_cnt left it)
  const int32_t* num_chunks;    // chunks of the map after the walk
  int chunk_limit;
  const uint32_t* err;          // the walk's error word: a walk that has to be repeated leaves nothing to fold
  uint32_t* skip;
  uint32_t pad;                 // fill dkey[total .. limit) (a sort that takes `limit` pairs whatever their number)
  uint32_t* zero;               // words the sort behind the compaction wants zeroed (radix_sort_zero_words), or null
  uint32_t zero_words;
};
__global__ __launch_bounds__(256) void compact_runs(const uint32_t* __restrict__ runkey, const uint32_t* __restrict__ run_cnt,
                                                    const uint32_t* __restrict__ run_off, uint32_t ntiles,
                                                    uint32_t r1_log2, uint32_t* __restrict__ dkey,
                                                    uint32_t* __restrict__ dval, RunGuard guard) {
  const uint32_t tile_blocks = (ntiles + 3u) / 4u;
  // (on the side: the status words of the sort that follows — a launch of its own otherwise, 25 us in front of the chain)
  if (guard.zero != nullptr)
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < guard.zero_words; i += gridDim.x * 256u) guard.zero[i] = 0u;
  if (blockIdx.x >= tile_blocks) {
    const uint32_t total = *guard.total, b = blockIdx.x - tile_blocks, nb = gridDim.x - tile_blocks;
    if (b == 0 && threadIdx.x == 0)
      *guard.skip = (total > guard.limit || *guard.num_chunks > guard.chunk_limit || *guard.err != 0u) ? 1u : 0u;
    if (guard.pad)
      for (uint32_t i = total + b * 256u + threadIdx.x; i < guard.limit; i += nb * 256u) {
        dkey[i] = 0xFFFFFFFFu;
        dval[i] = 0u;
      }
    return;
  }
  const uint32_t t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (t >= ntiles || (guard.err != nullptr && *guard.err != 0u)) return;
  const uint32_t n = run_cnt[t], o = run_off[t], base = t << r1_log2;
  for (uint32_t k = threadIdx.x & 63; k < n && o + k < guard.limit; k += 64) {
    dkey[o + k] = runkey[base + k];
    dval[o + k] = base + k;
  }
}

// The same, down to the voxel heads, for a call with few runs (a map whose colours have saturated keeps a
// trickle of them at its rim): one workgroup compacts the runs of the tiles into LDS as (key << 32 | slot),
// sorts them there (bitonic; slot order = tile order, so equal keys stay in point order) and lists the first
// run of every voxel — one launch instead of a compaction, the radix passes and the head scan, none of
// which is worth its launch for a few hundred runs.  ctr = the colour side's counters (num_desc: the runs).
constexpr uint32_t kSmallRuns = 4096;
__global__ __launch_bounds__(1024) void sort_runs_small(const uint32_t* __restrict__ runkey, const uint32_t* __restrict__ run_cnt,
                                                        uint32_t ntiles, uint32_t r1_log2, WalkCounters* __restrict__ ctr,
                                                        uint32_t* __restrict__ skeys, uint32_t* __restrict__ sval,
                                                        uint32_t* __restrict__ heads, uint32_t* __restrict__ skip) {
  __shared__ unsigned long long item[kSmallRuns];
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t D = ctr->num_desc;   // <= kSmallRuns: the host chose this path knowing D, or (skip != nullptr) expecting it
  if (skip != nullptr) {
    const bool bad = D > kSmallRuns || (ctr - 1)->err != 0u;   // (ctr = the colour side's counters, behind the walk's)
    if (tid == 0) *skip = bad ? 1u : 0u;
    if (bad) return;
  }
  uint32_t P = 64;
  while (P < D) P <<= 1;
  for (uint32_t k = tid; k < P; k += 1024) item[k] = ~0ull;
  if (tid == 0) carry = 0;
  __syncthreads();
  auto block_scan = [&](uint32_t c, uint32_t* total) {   // exclusive prefix of c over the workgroup, after `carry`
    uint32_t inc = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)inc, off);
      if (lane >= off) inc += up;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    uint32_t wb = carry, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      if (w < wid) wb += wsum[w];
      tot += wsum[w];
    }
    *total = tot;
    return wb + inc - c;
  };
  for (uint32_t base = 0; base < ntiles; base += 1024) {
    const uint32_t t = base + (uint32_t)tid;
    const uint32_t n = t < ntiles ? run_cnt[t] : 0u;
    uint32_t tot;
    const uint32_t o = block_scan(n, &tot);
    for (uint32_t k = 0; k < n; ++k) {
      const uint32_t slot = (t << r1_log2) + k;
      item[o + k] = ((unsigned long long)runkey[slot] << 32) | slot;
    }
    __syncthreads();
    if (tid == 0) carry += tot;
    __syncthreads();
  }
  for (uint32_t k = 2; k <= P; k <<= 1)
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = tid; i < P; i += 1024) {
        const uint32_t x = i ^ j;
        if (x > i) {
          const unsigned long long a = item[i], b = item[x];
          if ((a > b) == ((i & k) == 0)) {
            item[i] = b;
            item[x] = a;
          }
        }
      }
      __syncthreads();
    }
  if (tid == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < D; base += 1024) {
    const uint32_t j = base + (uint32_t)tid;
    uint32_t head = 0;
    if (j < D) {
      const unsigned long long it = item[j];
      skeys[j] = (uint32_t)(it >> 32);
      sval[j] = (uint32_t)it;
      head = (j == 0 || (uint32_t)(item[j - 1] >> 32) != (uint32_t)(it >> 32)) ? 1u : 0u;
    }
    uint32_t tot;
    const uint32_t o = block_scan(head, &tot);
    if (head) heads[o] = j;
    __syncthreads();
    if (tid == 0) carry += tot;
    __syncthreads();
  }
  if (tid == 0) ctr->num_heads = carry;
}

// ... and for a call with a MODERATE number of runs (a single key frame over new ground: 150 tiles of a few hundred cold
// voxels; the rim of a saturated map in a long call): the same three steps — the runs of the tiles listed densely in tile
// order, a stable sort by voxel key, the first run of every voxel — by ONE workgroup in one launch, where the general form
// takes eight (scan, compaction, two or three passes of histogram / scan / scatter, heads: 60 us of launches for 25 000
// runs, a third of a one-key-frame call).  LSD radix over 8-bit digits through two global buffers: a wave owns a contiguous
// slice, counts its digits into its own LDS histogram, and places its pairs batch by batch (64 at a time; a pair's rank
// among the batch's equal digits from eight ballots) behind everything smaller and behind the earlier waves' equals —
// stable, no atomics on shared counters.  passes = ceil(key bits / 8): the result is in (k0, v0) after an even number.
// limit / chunk_limit / skip: as compact_runs' guard (a chain launched on predicted sizes).
constexpr uint32_t kMediumRuns = 65536;
template <int kDigitBits>   // 8, or 10 where two passes of 10 bits cover the key (a map of up to 256 chunks: a third less work)
__global__ __launch_bounds__(1024) void sort_runs_medium(const uint32_t* __restrict__ runkey, const uint32_t* __restrict__ run_cnt,
                                                         uint32_t ntiles, uint32_t r1_log2, WalkCounters* __restrict__ ctr,
                                                         uint32_t* __restrict__ k0, uint32_t* __restrict__ v0,
                                                         uint32_t* __restrict__ k1, uint32_t* __restrict__ v1,
                                                         uint32_t* __restrict__ tile_off, uint32_t* __restrict__ heads, int passes,
                                                         uint32_t limit, const int32_t* __restrict__ num_chunks, int chunk_limit,
                                                         uint32_t* __restrict__ skip) {
  constexpr int kWaves = 16, kBins = 1 << kDigitBits;
  __shared__ uint32_t hist[kWaves][kBins];
  __shared__ uint32_t wsum[kWaves];
  __shared__ uint32_t carry;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  auto block_scan = [&](uint32_t c, uint32_t* total) {   // exclusive prefix of c over the workgroup, after `carry`
    const uint32_t inc = wave_scan_incl(c);
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    uint32_t wb = carry, tot = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      if (w < wid) wb += wsum[w];
      tot += wsum[w];
    }
    *total = tot;
    return wb + inc - c;
  };
  // ---- where every tile's runs go
  for (uint32_t base = 0; base < ntiles; base += 1024) {
    const uint32_t t = base + (uint32_t)tid;
    const uint32_t n = t < ntiles ? run_cnt[t] : 0u;
    uint32_t tot;
    const uint32_t o = block_scan(n, &tot);
    if (t < ntiles) tile_off[t] = o;
    __syncthreads();
    if (tid == 0) carry += tot;
    __syncthreads();
  }
  const uint32_t D = carry;
  if (tid == 0) ctr->num_desc = D;   // (the colour side's counters: what the fold reads as the number of runs)
  const bool bad = D > limit || D > kMediumRuns || (skip != nullptr && (*num_chunks > chunk_limit || (ctr - 1)->err != 0u));
  if (skip != nullptr && tid == 0) *skip = bad ? 1u : 0u;
  if (bad || D == 0u) {
    if (tid == 0 && !bad) ctr->num_heads = 0;
    return;
  }
  __syncthreads();   // (tile_off: written by other threads of this workgroup)
  if (D >= 8u * ntiles) {
    for (uint32_t t = (uint32_t)wid; t < ntiles; t += kWaves) {   // a wave per tile: (key, slot) in tile order
      const uint32_t n = run_cnt[t], o = tile_off[t], slot0 = t << r1_log2;
      for (uint32_t j = (uint32_t)lane; j < n; j += 64) {
        k0[o + j] = runkey[slot0 + j];
        v0[o + j] = slot0 + j;
      }
    }
  } else {   // many tiles with a run or two each (the rim of a saturated map in a long call): a thread per tile
    for (uint32_t t = (uint32_t)tid; t < ntiles; t += 1024) {
      const uint32_t n = run_cnt[t], o = tile_off[t], slot0 = t << r1_log2;
      for (uint32_t j = 0; j < n; ++j) {
        k0[o + j] = runkey[slot0 + j];
        v0[o + j] = slot0 + j;
      }
    }
  }
  __syncthreads();
  // ---- the passes
  const uint32_t per_wave = ((D + kWaves - 1) / kWaves + 63u) & ~63u;   // a slice of whole batches
  const uint32_t lo = min(D, (uint32_t)wid * per_wave), hi = min(D, lo + per_wave);
  const unsigned long long below = (1ull << lane) - 1ull;
  uint32_t *ki = k0, *vi = v0, *ko = k1, *vo = v1;
  for (int p = 0; p < passes; ++p) {
    const int shift = kDigitBits * p;
    for (int d = lane; d < kBins; d += 64) hist[wid][d] = 0u;
    for (uint32_t j0 = lo; j0 < hi; j0 += 4 * 64) {   // (four independent loads in flight per lane: one CU's global latency is the pass)
      uint32_t kk[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t j = j0 + (uint32_t)(u * 64 + lane);
        kk[u] = j < hi ? ki[j] : 0u;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (j0 + (uint32_t)(u * 64 + lane) < hi) atomicAdd(&hist[wid][(kk[u] >> shift) & (uint32_t)(kBins - 1)], 1u);
    }
    __syncthreads();
    {   // hist[w][d] -> where wave w's first pair of digit d goes: digits ascending, waves ascending inside a digit
      constexpr int kOwn = kBins * kWaves / 1024;   // (digit, wave) pairs a thread owns, consecutive in digit-major order
      uint32_t c[kOwn], sum = 0;
#pragma unroll
      for (int q = 0; q < kOwn; ++q) {
        const int i = kOwn * tid + q;
        c[q] = hist[i & (kWaves - 1)][i >> 4];
        sum += c[q];
      }
      if (tid == 0) carry = 0;
      uint32_t tot;
      const uint32_t o = block_scan(sum, &tot);   // (its barrier orders the reads above before the writes below)
      __syncthreads();
      uint32_t run = o;
#pragma unroll
      for (int q = 0; q < kOwn; ++q) {
        const int i = kOwn * tid + q;
        hist[i & (kWaves - 1)][i >> 4] = run;
        run += c[q];
      }
    }
    __syncthreads();
    for (uint32_t jq = lo; jq < hi; jq += 4 * 64) {
      uint32_t kq[4], vq[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {   // (the loads of four batches in flight, then the batches in order)
        const uint32_t j = jq + (uint32_t)(u * 64 + lane);
        kq[u] = j < hi ? ki[j] : 0u;
        vq[u] = j < hi ? vi[j] : 0u;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
      const uint32_t j0 = jq + (uint32_t)(u * 64);
      if (j0 >= hi) break;   // (uniform over the wave)
      const uint32_t j = j0 + (uint32_t)lane;
      const bool live = j < hi;
      const uint32_t key = kq[u], val = vq[u];
      const uint32_t dg = (key >> shift) & (uint32_t)(kBins - 1);
      unsigned long long same = __ballot(live);
#pragma unroll
      for (int b = 0; b < kDigitBits; ++b) {
        const unsigned long long m = __ballot((dg >> b) & 1u);
        same &= ((dg >> b) & 1u) ? m : ~m;
      }
      if (live) {
        const uint32_t at = hist[wid][dg] + (uint32_t)__popcll(same & below);
        ko[at] = key;
        vo[at] = val;
      }
      __builtin_amdgcn_wave_barrier();
      if (live && (same >> lane) <= 1ull) hist[wid][dg] += (uint32_t)__popcll(same);   // (the digit's last lane of the batch)
      __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
    uint32_t* t = ki; ki = ko; ko = t;
    t = vi; vi = vo; vo = t;
  }
  // ---- the first run of every voxel
  if (tid == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < D; base += 1024) {
    const uint32_t j = base + (uint32_t)tid;
    const uint32_t head = (j < D && (j == 0 || ki[j - 1] != ki[j])) ? 1u : 0u;
    uint32_t tot;
    const uint32_t o = block_scan(head, &tot);
    if (head) heads[o] = j;
    __syncthreads();
    if (tid == 0) carry += tot;
    __syncthreads();
  }
  if (tid == 0) ctr->num_heads = carry;
}

// ColorVoxel::IntegrateSimple visit by visit, for the voxels whose colour weight is below 254.  One
// wave per voxel: the lanes take its runs (sorted: tile order = point order) 64 at a time, place the
// colours of their visits — the rays of a mask in ascending order — in LDS at the visit's rank, and three
// lanes fold the red, green and blue sequences (the weight is common) until the weight reaches 254: at
// most 254 steps in the life of a voxel, none once it is there.
constexpr int kFoldWaves = 4;
// Where the fold finds run `val`: the per-tile regions of the walk (words = kMaskWords: the mask at
// base + val * words, the tile from the slot number) or the runs an owner received in the ray-sharded integrate
// (words = kWireRun: {chunk key (2 words), voxel | tile << 12, six ray spans}).
//
// Wire form of a run (round 4; it was the 512-bit mask behind a 16-byte header, 80 bytes): the rays of a tile that see
// one voxel are a few stretches of consecutive points (neighbouring pixels of one or two image rows), so the mask
// travels as SPANS, 16 bits each: first ray (9 bits) | length - 1 (7 bits) << 9, 0xFFFF = none.  Six spans per
// 24-byte record; a run with more spans (or one longer than 128 rays) continues in further records of the same
// (voxel, tile), which the stable sorts keep in order.
constexpr uint32_t kWireRun = 6;
constexpr uint32_t kWireSpans = 6;
constexpr uint32_t kSpanNone = 0xFFFFu;
constexpr uint32_t kWireTileBits = 20;   // tile index of the call in the upper bits of word 2
// Spans of a ray mask, in ascending order: calls emit(start, length) for each (length <= 128); returns their number.
// ------------------------------------------------------------------ a long call's runs, chunk by chunk
// The stable sort of ALL runs by voxel (compaction, digit totals, three radix passes, heads: 0.2 ms for the 2.5 million runs of
// 100 key frames over new ground, the tail of the step) without sorting them all: walk_fast leaves a tile's runs grouped by
// (chunk, slab) and says beside every segment descriptor where they are (RunOut::rseg).  A ROW = (updated chunk, slab); the
// descriptor slots are cut into blocks of kSegSpan (16 tiles):
//   runs_count     a workgroup per block: the runs of every row in the block -> matrix [row][block]; per descriptor, the runs
//                  of its rows in the block's earlier tiles (rpre)
//   runs_rowscan   a wave per row: exclusive scan along the blocks; the row's total gives it a region of the output and its
//                  parts of kCollectPart runs
//   runs_scatter   a workgroup per block: the slots of its runs to region + matrix + rpre — the row's runs in TILE order,
//                  no atomics, nothing sorted
//   parts_count    a workgroup per part: the keys of its runs fetched, a histogram over the slab's 512 voxels
//   parts_place    a workgroup per part: stable counting sort by voxel — the row's parts before this one, per-wave
//                  histograms, ranks by ballots; the first part of a row lists the voxels' first runs (heads)
// and the fold reads keys, slots and heads as the general chain leaves them.  Only calls whose tiles all went through
// walk_fast (no tile left to walk_tiles, whose runs are not grouped).  More updated chunks than the matrix has rows for, or
// more chunks in a block than runs_count's table holds, set `skip`: the fold leaves at once and the host runs the general chain.
constexpr uint32_t kCollectPart = 4096;
constexpr int kRunTable = 256;            // chunks of one block (16 tiles) runs_count has room for
constexpr int kSpanTiles = kSegSpan / kWalkChunks;
__device__ __forceinline__ uint32_t suboffset_of(const uint4& p, int s) {   // (pack_suboffsets)
  const uint32_t w = s < 2 ? p.x : (s < 4 ? p.y : (s < 6 ? p.z : p.w));
  return (s & 1) ? w >> 16 : w & 0xFFFFu;
}
// runs of slab s of the segment whose run descriptor is (r0, r1): first slot, count
__device__ __forceinline__ void slab_runs(const uint4& r0, const uint4& r1, int s, uint32_t* first, uint32_t* cnt) {
  const uint32_t lo = suboffset_of(r1, s), hi = s + 1 < kSlabs ? suboffset_of(r1, s + 1) : r0.y;
  *first = r0.x + lo;
  *cnt = hi - lo;
}

__global__ __launch_bounds__(kSegSpan) void runs_count(const uint4* __restrict__ seg, const uint4* __restrict__ rseg, uint32_t ntiles,
                                                       const uint32_t* __restrict__ seg_cnt, const uint32_t* __restrict__ active_idx,
                                                       uint32_t rows_cap, uint32_t nblocks, uint32_t* __restrict__ M,
                                                       uint4* __restrict__ rpre, WalkCounters* __restrict__ ctr,
                                                       const uint32_t* __restrict__ left_to_walk_tiles) {
  // (a thread per descriptor slot: one chain of dependent loads per thread, all of the block's in flight together)
  __shared__ uint32_t hkey[kRunTable];
  __shared__ alignas(16) uint16_t tbl[kRunTable][kSpanTiles][kSlabs];
  static_assert(kSpanTiles == 16 && kSegSpan == 1024, "16 tiles per block");
  const int tid = threadIdx.x;
  // (a tile left to walk_tiles: its runs are not grouped by chunk, its segments may lie in the spill area — not this chain's call)
  // (... nor a call that has to be repeated with more room: err); the run descriptors of such tiles are not valid
  if (*left_to_walk_tiles != 0u || ctr->seg_top != 0u || ctr->err != 0u) {   // (uniform)
    if (blockIdx.x == 0 && tid == 0) ctr[1].skip = 1u;
    return;
  }
  static_assert(kRunTable <= kSegSpan && (kRunTable & (kRunTable - 1)) == 0, "a thread per table entry");
  if (tid < kRunTable) hkey[tid] = 0xFFFFFFFFu;
  for (int k = tid; k < kRunTable * kSpanTiles * kSlabs / 2; k += kSegSpan) reinterpret_cast<uint32_t*>(&tbl[0][0][0])[k] = 0u;
  __syncthreads();
  const uint32_t j = blockIdx.x * (uint32_t)kSegSpan + (uint32_t)tid;
  const uint32_t tile = j / kWalkChunks;
  const int tl = tid / kWalkChunks;
  int ent = -1;
  if (tile < ntiles && (j % kWalkChunks) < seg_cnt[tile]) {
    const uint4 r0 = rseg[2 * (size_t)j], r1 = rseg[2 * (size_t)j + 1];
    const uint32_t slot = seg[2 * (size_t)j].x;
    if (r0.y != 0u) {   // (runs in this chunk)
      uint32_t h = (slot * 2654435761u) >> (32 - log2_of(kRunTable));
      for (int probe = 0; probe < kRunTable; ++probe) {
        uint32_t cur = hkey[h];
        if (cur == 0xFFFFFFFFu) cur = atomicCAS(&hkey[h], 0xFFFFFFFFu, slot);
        if (cur == 0xFFFFFFFFu || cur == slot) {
          ent = (int)h;
          break;
        }
        h = (h + 1) & (kRunTable - 1);
      }
      if (ent < 0) {
        ctr[1].skip = 1u;   // more chunks in the block than the table holds
      } else {
        uint32_t c[kSlabs];
#pragma unroll
        for (int s = 0; s < kSlabs; ++s) {
          uint32_t f;
          slab_runs(r0, r1, s, &f, &c[s]);
        }
        *reinterpret_cast<uint4*>(&tbl[ent][tl][0]) = pack_suboffsets(c);
      }
    }
  }
  __syncthreads();
  if (ent >= 0) {
    uint32_t pre[kSlabs];
#pragma unroll
    for (int s = 0; s < kSlabs; ++s) pre[s] = 0;
    for (int t = 0; t < tl; ++t) {
      const uint4 row = *reinterpret_cast<const uint4*>(&tbl[ent][t][0]);
      pre[0] += row.x & 0xFFFFu; pre[1] += row.x >> 16;
      pre[2] += row.y & 0xFFFFu; pre[3] += row.y >> 16;
      pre[4] += row.z & 0xFFFFu; pre[5] += row.z >> 16;
      pre[6] += row.w & 0xFFFFu; pre[7] += row.w >> 16;
    }
    rpre[j] = pack_suboffsets(pre);   // (at most 15 tiles x 512 runs each)
  }
  for (int cell = tid; cell < kRunTable * kSlabs; cell += kSegSpan) {   // the (chunk, slab) cells of the table
    const int e = cell / kSlabs, s = cell % kSlabs;
    if (hkey[e] != 0xFFFFFFFFu) {
      uint32_t tot = 0;
#pragma unroll
      for (int t = 0; t < kSpanTiles; ++t) tot += tbl[e][t][s];
      if (tot != 0u) {
        const uint32_t row = active_idx[hkey[e]] * (uint32_t)kSlabs + (uint32_t)s;
        if (row < rows_cap) M[(size_t)row * nblocks + blockIdx.x] = tot;
        else ctr[1].skip = 1u;   // more updated chunks than the matrix has rows for
      }
    }
  }
}

__global__ __launch_bounds__(256) void runs_rowscan(uint32_t* __restrict__ M, uint32_t rows_cap, uint32_t nblocks,
                                                    const WalkCounters* __restrict__ ctr, uint32_t* __restrict__ item_cnt) {
  if (ctr[1].skip != 0u) return;
  const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const uint32_t nrows = min(rows_cap, ctr->num_updated * (uint32_t)kSlabs);
  for (uint32_t row = blockIdx.x * 4u + wid; row < nrows; row += gridDim.x * 4u) {
    uint32_t* const m = M + (size_t)row * nblocks;
    uint32_t running = 0;
    // (a lane takes 16 consecutive blocks: the loads of a round of 1024 blocks are in flight together, one scan per round)
    for (uint32_t b0 = 0; b0 < nblocks; b0 += 1024) {
      uint32_t v[16], sum = 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const uint32_t b = b0 + lane * 16u + (uint32_t)k;
        v[k] = b < nblocks ? m[b] : 0u;
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) sum += v[k];
      const uint32_t inc = wave_scan_incl(sum);
      uint32_t at = running + inc - sum;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const uint32_t b = b0 + lane * 16u + (uint32_t)k;
        if (b < nblocks && v[k] != 0u) m[b] = at;   // (only the cells that hold runs are read again)
        at += v[k];
      }
      running += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
    if (lane == 0) item_cnt[row] = running;
  }
}

// The rows' regions and parts: one workgroup scans the rows' totals (a counter bumped by every row instead serialises a
// thousand same-address atomics).
__global__ __launch_bounds__(1024) void rows_place(const uint32_t* __restrict__ item_cnt, uint32_t rows_cap, uint32_t run_bound,
                                                   uint32_t parts_cap, WalkCounters* __restrict__ ctr,
                                                   uint32_t* __restrict__ item_base, uint32_t* __restrict__ item_part0,
                                                   uint32_t* __restrict__ part_item, const uint32_t* __restrict__ pub_b,
                                                   uint32_t* __restrict__ host_a, uint32_t* __restrict__ host_b,
                                                   uint32_t nb, uint32_t* __restrict__ host_seq, uint32_t seq) {
  // (host_seq, may be null: the kernel ends by publishing the counters — both WalkCounters to host_a, nb words of pub_b to
  // host_b, then the sequence number — as tsdf_chisel.hip's publish_counters does: the last single-workgroup kernel behind
  // the walk on the caller's stream saves the call a launch of that one)
  __shared__ uint32_t wt[16], wp[16];
  __shared__ uint32_t ct, cp;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const bool go = ctr[1].skip == 0u;   // (uniform)
  const uint32_t nrows = go ? min(rows_cap, ctr->num_updated * (uint32_t)kSlabs) : 0u;
  if (tid == 0) { ct = 0; cp = 0; }
  __syncthreads();
  for (uint32_t r0 = 0; r0 < nrows; r0 += 1024) {
    const uint32_t r = r0 + tid;
    const uint32_t T = r < nrows ? item_cnt[r] : 0u, np = (T + kCollectPart - 1u) / kCollectPart;
    const uint32_t ti = wave_scan_incl(T), pi = wave_scan_incl(np);
    if (lane == 63) { wt[wid] = ti; wp[wid] = pi; }
    __syncthreads();
    uint32_t tb = ct, pb = cp, tt = 0, pt = 0;
#pragma unroll
    for (uint32_t w = 0; w < 16; ++w) {
      if (w < wid) { tb += wt[w]; pb += wp[w]; }
      tt += wt[w];
      pt += wp[w];
    }
    if (r < nrows) {
      const uint32_t p0 = pb + pi - np;
      item_base[r] = tb + ti - T;
      item_part0[r] = p0;
      for (uint32_t i = 0; i < np && p0 + i < parts_cap; ++i) part_item[p0 + i] = r;
    }
    __syncthreads();
    if (tid == 0) { ct += tt; cp += pt; }
    __syncthreads();
  }
  if (tid == 0 && go) {
    ctr[1].collect_top = ct;
    ctr[1].collect_parts = cp;
    if (ct > run_bound || cp > parts_cap) ctr[1].skip = 1u;   // (buffers sized on the call before: the host runs the general chain)
  }
  if (host_seq != nullptr) {
    __syncthreads();
    const uint32_t* a = reinterpret_cast<const uint32_t*>(ctr);
    for (uint32_t k = tid; k < (uint32_t)(2 * sizeof(WalkCounters) / sizeof(uint32_t)); k += 1024) host_a[k] = a[k];
    for (uint32_t k = tid; k < nb; k += 1024) host_b[k] = pub_b[k];
    __threadfence_system();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ __launch_bounds__(256) void runs_scatter(const uint4* __restrict__ seg, const uint4* __restrict__ rseg, uint32_t ntiles,
                                                    const uint32_t* __restrict__ seg_cnt, const uint32_t* __restrict__ active_idx,
                                                    uint32_t rows_cap, uint32_t nblocks, const uint32_t* __restrict__ M,
                                                    const uint4* __restrict__ rpre, const uint32_t* __restrict__ item_base,
                                                    const WalkCounters* __restrict__ ctr, uint32_t* __restrict__ tv) {
  // (a thread per descriptor slot, 256 slots = four tiles per workgroup; the loads of the eight slabs' places go out together;
  // a group of 32 runs or more is written by the whole wave)
  if (ctr[1].skip != 0u) return;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  const uint32_t tile = j / kWalkChunks, block = j / (uint32_t)kSegSpan;
  uint32_t f[kSlabs], c[kSlabs], at[kSlabs];
#pragma unroll
  for (int s = 0; s < kSlabs; ++s) f[s] = c[s] = at[s] = 0u;
  uint4 r0 = make_uint4(0u, 0u, 0u, 0u);
  if (tile < ntiles && (j % kWalkChunks) < seg_cnt[tile]) r0 = rseg[2 * (size_t)j];
  if (r0.y != 0u) {
    const uint4 r1 = rseg[2 * (size_t)j + 1], pre = rpre[j];
    const uint32_t row0 = active_idx[seg[2 * (size_t)j].x] * (uint32_t)kSlabs;
#pragma unroll
    for (int s = 0; s < kSlabs; ++s) slab_runs(r0, r1, s, &f[s], &c[s]);
    const uint4 ib0 = *reinterpret_cast<const uint4*>(item_base + row0), ib1 = *reinterpret_cast<const uint4*>(item_base + row0 + 4);
    const uint32_t ib[kSlabs] = {ib0.x, ib0.y, ib0.z, ib0.w, ib1.x, ib1.y, ib1.z, ib1.w};
    uint32_t m[kSlabs];
#pragma unroll
    for (int s = 0; s < kSlabs; ++s) m[s] = c[s] ? M[(size_t)(row0 + (uint32_t)s) * nblocks + block] : 0u;
#pragma unroll
    for (int s = 0; s < kSlabs; ++s) at[s] = ib[s] + m[s] + suboffset_of(pre, s);
  }
#pragma unroll
  for (int s = 0; s < kSlabs; ++s) {
    const bool big = c[s] >= 32u;
    if (!big)
      for (uint32_t k = 0; k < c[s]; ++k) tv[at[s] + k] = f[s] + k;
    unsigned long long todo = __ballot(big);
    while (todo) {
      const int src = __ffsll((long long)todo) - 1;
      todo &= todo - 1ull;
      const uint32_t bp = (uint32_t)__shfl((int)at[s], src), bf = (uint32_t)__shfl((int)f[s], src);
      const uint32_t bc = (uint32_t)__shfl((int)c[s], src);
      for (uint32_t k = lane; k < bc; k += 64) tv[bp + k] = bf + k;
    }
  }
}

constexpr int kCollectFlight = 4;   // loads of 64 a wave has in flight
__global__ __launch_bounds__(256) void parts_count(const uint32_t* __restrict__ part_item, const uint32_t* __restrict__ item_part0,
                                                   const uint32_t* __restrict__ item_base, const uint32_t* __restrict__ item_cnt,
                                                   const uint32_t* __restrict__ runkey, const WalkCounters* __restrict__ ctr,
                                                   const uint32_t* __restrict__ tv, uint32_t* __restrict__ tk,
                                                   uint32_t* __restrict__ phist) {
  __shared__ uint32_t hist[kSlabVox];
  if (ctr[1].skip != 0u) return;
  const uint32_t tid = threadIdx.x, nparts = ctr[1].collect_parts;
  for (uint32_t p = blockIdx.x; p < nparts; p += gridDim.x) {
    const uint32_t row = part_item[p], lo = (p - item_part0[row]) * kCollectPart, hi = min(item_cnt[row], lo + kCollectPart);
    const uint32_t base = item_base[row];
    for (uint32_t k = tid; k < (uint32_t)kSlabVox; k += 256) hist[k] = 0u;
    __syncthreads();
    for (uint32_t e0 = lo; e0 < hi; e0 += 256 * kCollectFlight) {
      uint32_t slot[kCollectFlight], k[kCollectFlight];
#pragma unroll
      for (int b = 0; b < kCollectFlight; ++b) {
        const uint32_t e = e0 + 256u * (uint32_t)b + tid;
        slot[b] = e < hi ? tv[base + e] : 0xFFFFFFFFu;
      }
#pragma unroll
      for (int b = 0; b < kCollectFlight; ++b) k[b] = slot[b] != 0xFFFFFFFFu ? runkey[slot[b]] : 0u;
#pragma unroll
      for (int b = 0; b < kCollectFlight; ++b) {
        if (slot[b] == 0xFFFFFFFFu) continue;
        tk[base + e0 + 256u * (uint32_t)b + tid] = k[b];
        atomicAdd(&hist[k[b] % (uint32_t)kSlabVox], 1u);
      }
    }
    __syncthreads();
    for (uint32_t k = tid; k < (uint32_t)kSlabVox; k += 256) phist[(size_t)p * kSlabVox + k] = hist[k];
    __syncthreads();
  }
}

// Voxels of a row that have runs = the entries it adds to the head list (a wave per row sums its parts' histograms), so that
// parts_place finds a row's place in the list by a sum over the rows before it instead of a same-address atomic per row.
// (A kernel of its own: the last part of a row to finish could do it behind a device-scope fence — which on this device
// writes the L2 back, 150 us for the six hundred workgroups of parts_count.)
__global__ __launch_bounds__(256) void rows_heads(const uint32_t* __restrict__ item_part0, const uint32_t* __restrict__ item_cnt,
                                                  uint32_t* __restrict__ phist, const WalkCounters* __restrict__ ctr,
                                                  uint32_t rows_cap, uint32_t* __restrict__ row_heads,
                                                  uint32_t* __restrict__ row_tot) {
  // (... and turns the histograms of a row's parts into their exclusive prefix along the parts, the row's totals beside them
  // (row_tot): a part of parts_place then reads two rows of 512 counts, not one per part of its row — a voxel slab seen by a
  // thousand key frames has hundreds of parts)
  if (ctr[1].skip != 0u) return;
  const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const uint32_t nrows = min(rows_cap, ctr->num_updated * (uint32_t)kSlabs);
  for (uint32_t row = blockIdx.x * 4u + wid; row < nrows; row += gridDim.x * 4u) {
    const uint32_t T = item_cnt[row], P = (T + kCollectPart - 1u) / kCollectPart, p0 = item_part0[row];
    uint32_t nz = 0;
    if (P) {
      uint32_t h[kSlabVox / 64];
#pragma unroll
      for (int k = 0; k < kSlabVox / 64; ++k) h[k] = 0;
      for (uint32_t q = 0; q < P; ++q)
#pragma unroll
        for (int k = 0; k < kSlabVox / 64; ++k) {
          uint32_t* const cell = &phist[(size_t)(p0 + q) * kSlabVox + (uint32_t)k * 64u + lane];
          const uint32_t c = *cell;
          *cell = h[k];
          h[k] += c;
        }
#pragma unroll
      for (int k = 0; k < kSlabVox / 64; ++k) {
        row_tot[(size_t)row * kSlabVox + (uint32_t)k * 64u + lane] = h[k];
        nz += (uint32_t)__popcll(__ballot(h[k] != 0u));
      }
    }
    if (lane == 0) row_heads[row] = nz;
  }
}

__global__ __launch_bounds__(512) void parts_place(const uint32_t* __restrict__ part_item, const uint32_t* __restrict__ item_part0,
                                                   const uint32_t* __restrict__ item_base, const uint32_t* __restrict__ item_cnt,
                                                   const uint32_t* __restrict__ phist, WalkCounters* __restrict__ ctr,
                                                   const uint32_t* __restrict__ tk, const uint32_t* __restrict__ tv,
                                                   uint32_t* __restrict__ ok, uint32_t* __restrict__ ov,
                                                   uint32_t* __restrict__ heads, const uint32_t* __restrict__ row_heads,
                                                   uint32_t rows_cap, const uint32_t* __restrict__ row_tot) {
  constexpr int kWaves = 8, kBatches = kCollectPart / (64 * kWaves);   // a wave: 512 consecutive runs of the part
  __shared__ uint32_t hist[kWaves][kSlabVox];
  __shared__ uint32_t wtot[kWaves], wflag[kWaves], wsum[kWaves];
  static_assert(kSlabVox == 512, "a thread per voxel of the slab");
  if (ctr[1].skip != 0u) return;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t nparts = ctr[1].collect_parts;
  const uint32_t nrows = min(rows_cap, ctr->num_updated * (uint32_t)kSlabs);
  for (uint32_t p = blockIdx.x; p < nparts; p += gridDim.x) {
    const uint32_t row = part_item[p], p0 = item_part0[row], pi = p - p0, T = item_cnt[row];
    const uint32_t base = item_base[row], lo = pi * kCollectPart, hi = min(T, lo + kCollectPart);
#pragma unroll
    for (int w = 0; w < kWaves; ++w) hist[w][tid] = 0u;
    __syncthreads();
    const uint32_t wlo = lo + wid * (kCollectPart / kWaves);
    uint32_t k[kBatches], v[kBatches];
#pragma unroll
    for (int b = 0; b < kBatches; ++b) {
      const uint32_t e = wlo + 64u * (uint32_t)b + lane;
      k[b] = e < hi ? tk[base + e] : 0u;
      v[b] = e < hi ? tv[base + e] : 0u;
    }
#pragma unroll
    for (int b = 0; b < kBatches; ++b)
      if (wlo + 64u * (uint32_t)b + lane < hi) atomicAdd(&hist[wid][k[b] % (uint32_t)kSlabVox], 1u);
    __syncthreads();
    {   // thread x: voxel x of the slab
      uint32_t tot = 0;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) {
        const uint32_t c = hist[w][tid];
        hist[w][tid] = tot;
        tot += c;
      }
      // the voxel's runs in the whole row / in the row's parts before this one (rows_heads)
      const uint32_t all = row_tot[(size_t)row * kSlabVox + tid], before = phist[(size_t)p * kSlabVox + tid];
      const uint32_t flag = (pi == 0u && all) ? 1u : 0u;
      const uint32_t inc = wave_scan_incl(all), finc = wave_scan_incl(flag);
      // (the row's place in the head list: the head counts of the rows before it — parts_count left them; the first part of
      // all also leaves the list's length)
      uint32_t hsum = 0;
      if (pi == 0u) {   // (uniform)
        for (uint32_t r = tid; r < row; r += 512) hsum += row_heads[r];
        uint32_t all_rows = 0;
        if (p == 0u) {
          for (uint32_t r = tid; r < nrows; r += 512) all_rows += row_heads[r];
          all_rows = wave_sum(all_rows);
        }
        hsum = wave_sum(hsum);
        if (lane == 0) wsum[wid] = hsum;
        if (p == 0u && lane == 0 && all_rows) atomicAdd(&ctr[1].num_heads, all_rows);
      }
      if (lane == 63) {
        wtot[wid] = inc;
        wflag[wid] = finc;
      }
      __syncthreads();
      uint32_t wb = 0, fb = 0, hbase = 0;
#pragma unroll
      for (uint32_t w = 0; w < (uint32_t)kWaves; ++w) {
        if (w < wid) {
          wb += wtot[w];
          fb += wflag[w];
        }
        hbase += pi == 0u ? wsum[w] : 0u;
      }
      const uint32_t start = wb + inc - all;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) hist[w][tid] += start + before;
      __syncthreads();
      if (flag) heads[hbase + fb + finc - 1u] = base + start;
    }
    // ---- placement: the lanes of a batch that name the same voxel keep their order (ballots)
#pragma unroll
    for (int b = 0; b < kBatches; ++b) {
      const bool valid = wlo + 64u * (uint32_t)b + lane < hi;
      const uint32_t d = k[b] % (uint32_t)kSlabVox;
      unsigned long long same = __ballot(valid);
#pragma unroll
      for (int bit = 0; bit < 9; ++bit) {
        const unsigned long long bb = __ballot((d >> bit) & 1u);
        same &= ((d >> bit) & 1u) ? bb : ~bb;
      }
      if (valid) {
        const uint32_t rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        const uint32_t pos = hist[wid][d] + rank;
        ok[base + pos] = k[b];
        ov[base + pos] = v[b];
        if ((same >> lane) == 1ull) hist[wid][d] = pos + 1u;   // (the group's last lane: the next batch goes on behind it)
      }
    }
    __syncthreads();
  }
}

template <typename Emit>
__device__ __forceinline__ uint32_t mask_spans(const uint32_t* m, Emit emit) {
  uint32_t n = 0;
  int start = -1;   // first ray of the open stretch
  auto close = [&](uint32_t end) {
    uint32_t first = (uint32_t)start, len = end - first;
    for (; len > 128u; first += 128u, len -= 128u, ++n) emit(first, 128u);
    emit(first, len);
    ++n;
    start = -1;
  };
#pragma unroll
  for (int w = 0; w < kMaskWords; ++w) {
    const uint32_t bits = m[w];
    uint32_t pos = 0;
    while (pos < 32u) {
      if (start < 0) {   // the next one at or after pos
        const uint32_t rest = bits >> pos;
        if (rest == 0u) break;
        pos += (uint32_t)__ffs((int)rest) - 1u;
        start = w * 32 + (int)pos;
      } else {           // the next zero
        const uint32_t rest = ~bits >> pos;
        if (rest == 0u) break;   // ones to the end of the word: the stretch goes on
        pos += (uint32_t)__ffs((int)rest) - 1u;
        close((uint32_t)w * 32u + pos);
      }
    }
  }
  if (start >= 0) close((uint32_t)kMaskWords * 32u);
  return n;
}
struct RunSrc {
  const uint32_t* base;
  uint32_t words, r1_log2;
  TileMap tmap;
  const int32_t* offsets;   // the call's offsets + tile table (tile -> first point: tile_span)
  int nclouds;
  const uint32_t* tile_first;   // first point of every tile of the call (nullptr: searched in the table above)
};
constexpr int kFoldGroup = 8;     // voxels a wave folds together (staging: 8 lanes each; fold: 4 lanes each: r, g, b, idle)
constexpr int kFoldSteps = 256;   // >= 254: the visits that can still count for a voxel
// kGrid: the runs of a depth-image call (GridSrc): tiles are 32 x 16 blocks of grid pixels, `rgb` = the colour images.  The
// runs of a voxel arrive in tile order = (image, band of 16 grid rows, column block); the reference's point order inside a
// band goes row by row ACROSS its tiles, so the runs of one band are staged together, mask word by mask word (a word = one
// grid row of a tile).  A staged visit is its order key (image << key_bits | raster index), its colour is read at that pixel.
template <bool kGrid>
__global__ __launch_bounds__(64 * kFoldWaves) void fold_colours_masks(
    const uint32_t* __restrict__ skeys, const uint32_t* __restrict__ sorted_val, const uint32_t* __restrict__ nd_dev,
    RunSrc src, const uint32_t* __restrict__ vj0, const uint8_t* __restrict__ rgb, uint32_t* __restrict__ rgbw,
    const uint32_t* __restrict__ num_heads, uint32_t* __restrict__ sat_list, uint32_t* __restrict__ sat_count,
    const uint32_t* __restrict__ skip, const GridSrc grid) {
  // sat_list (ray-sharded integrate): the voxels whose colour weight reaches 254 in this call
  // skip (a chain launched on predicted sizes, compact_runs): non-zero = the prediction failed, nothing here is valid
  if (skip != nullptr && *skip != 0u) return;
  // A wave takes kFoldGroup voxels of the head list at a time.
  //  staging  eight lanes per voxel, one lane per RUN, eight runs of every voxel at a time: the lane reads its run's ray
  //           mask and writes the colours of its rays — bits ascending = point order — into the stage of its voxel, at
  //           the place a prefix sum over the voxel's eight lanes gives it (runs of a voxel in tile order: the stable
  //           sort kept it); only the first 254 - weight visits of a voxel count, a voxel that has them stops reading.
  //           64 independent mask and colour loads in flight per step.
  //  fold     four lanes per voxel (r, g, b, -): ColorVoxel::IntegrateSimple visit by visit, 1 / (1 + weight) from a
  //           table of the 254 quotients the reference's division can produce.
  // (The first version gave a whole wave to ONE voxel and folded on three of its lanes: the 60 000 voxels of a
  // 5-key-frame call on a young map took 81 us, nearly all of it the serial chains at 3 / 64 lanes.)
  __shared__ uint32_t stage[kFoldWaves][kFoldGroup][kFoldSteps];
  __shared__ float rcp[256];
  __shared__ uint32_t have[kFoldWaves][kFoldGroup];
  for (int k = threadIdx.x; k < 256; k += blockDim.x) rcp[k] = 1.f / (float)(1u + (uint32_t)k);
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const uint32_t nvox = *num_heads, nd = *nd_dev;
  const uint32_t ngroups = (nvox + kFoldGroup - 1) / kFoldGroup, nwaves = gridDim.x * kFoldWaves;
  for (uint32_t grp = blockIdx.x * kFoldWaves + wid; grp < ngroups; grp += nwaves) {
    const uint32_t v0 = grp * kFoldGroup, nv = min((uint32_t)kFoldGroup, nvox - v0);
    // ---- the group's voxels: lane i < nv holds voxel v0 + i
    uint32_t my_j0 = nd, my_key = 0, my_col = 0xFF000000u;
    if ((uint32_t)lane < nv) {
      my_j0 = vj0[v0 + lane];
      my_key = skeys[my_j0];
      my_col = rgbw[my_key];
    }
    // ---- staging: lanes 8 v .. 8 v + 7 work on voxel v
    if constexpr (kGrid) {
      // A lane per BAND of the voxel (a band's runs = its column blocks, consecutive in the sorted order; nearly always
      // one, two where the voxel's pixels straddle a block boundary): the lane merges its band's masks word by word — row
      // w of block A, then row w of block B ... — which is the reference's raster order inside the band; bands follow
      // each other in point order, so the lanes' counts chain by a prefix sum as the runs of the point-stream form do.
      // The eight lanes look at eight runs at a time and take the complete bands among them; a band of more than three
      // runs (a voxel a hand's breadth from the camera) takes the general form below, alone.
      const int vl = lane >> 3, sub = lane & 7;
      const uint32_t j0v = (uint32_t)__shfl((int)my_j0, vl), keyv = (uint32_t)__shfl((int)my_key, vl);
      const uint32_t cwv = (uint32_t)__shfl((int)my_col, vl) >> 24;
      const uint32_t needv = cwv >= 254u ? 0u : 254u - cwv;
      uint32_t havev = 0, jn = j0v;
      bool open = (uint32_t)vl < nv && needv > 0u;
      const uint32_t gshift = 8u * (uint32_t)vl;
      auto run_tile = [&](uint32_t j) { return src.tmap.tile_of(sorted_val[j] >> src.r1_log2); };
      // (general form) one mask word of every run of a chunk of up to eight runs of one band: grid row w, tile after tile
      auto stage_row = [&](uint32_t bits, const GridTile& t, uint32_t w) {
        const uint32_t cnt = (uint32_t)__popc(bits);
        uint32_t inc = cnt;
#pragma unroll
        for (int d = 1; d < 8; d <<= 1) {
          const uint32_t up = (uint32_t)__shfl_up((int)inc, d, 8);
          if (sub >= d) inc += up;
        }
        uint32_t at = havev + inc - cnt;
        const uint32_t key0 = (t.cloud << grid.key_bits) | ((t.row0 + w) * grid.gw + t.col0);
        while (bits && at < needv) {
          const int bpos = __ffs((int)bits) - 1;
          bits &= bits - 1u;
          stage[wid][vl][at++] = key0 + (uint32_t)bpos;
        }
        havev += (uint32_t)__shfl((int)inc, 7, 8);
      };
      while (__any(open)) {
        const uint32_t j = jn + (uint32_t)sub;
        // (key and slot of a run are fetched TOGETHER, the slot whether or not the key turns out to be the voxel's: a round
        // is then two dependent memory latencies — keys and slots, masks — instead of three; a voxel 4 m away needs ~150
        // runs = 19 rounds for its 254 visits, and the slowest voxel of its eight is the wave's time)
        const uint32_t jc = min(j, nd - 1u), j8 = min(jn + 8u, nd - 1u);
        const uint32_t key_j = skeys[jc], val_j = sorted_val[jc], key_8 = skeys[j8], val_8 = sorted_val[j8];
        const bool mine = open && j < nd && key_j == keyv;
        const uint32_t val = mine ? val_j : 0u;
        const uint32_t gtile = mine ? src.tmap.tile_of(val >> src.r1_log2) : 0u;
        const uint32_t band = mine ? grid_band(grid, gtile) : 0xFFFFFFFFu;   // (image and band in one number: tiles are numbered band by band)
        uint32_t band8 = 0xFFFFFFFEu;   // the band of the run behind these eight: is the last band here complete?
        if (open && jn + 8u < nd && key_8 == keyv) band8 = grid_band(grid, src.tmap.tile_of(val_8 >> src.r1_log2));
        const uint32_t bprev = (uint32_t)__shfl_up((int)band, 1, 8);
        const bool start = mine && (sub == 0 || band != bprev);
        const uint32_t minem = (uint32_t)(__ballot(mine) >> gshift) & 0xFFu, startm = (uint32_t)(__ballot(start) >> gshift) & 0xFFu;
        const uint32_t nvalid = (uint32_t)__popc(minem);
        const bool tail_open = nvalid == 8u && band8 == (uint32_t)__shfl((int)band, 7, 8);
        uint32_t take = tail_open ? 31u - (uint32_t)__clz((int)startm) : nvalid;   // runs of complete bands
        const uint32_t above = startm & ~((2u << sub) - 1u);
        const uint32_t nb = (above ? (uint32_t)__ffs((int)above) - 1u : nvalid) - (uint32_t)sub;   // (a band leader's runs)
        const uint32_t bigm = (uint32_t)(__ballot(start && (uint32_t)sub < take && nb > 3u) >> gshift) & 0xFFu;
        if (bigm) take = min(take, (uint32_t)__ffs((int)bigm) - 1u);
        if (!open) {
          // (the voxel is done: its lanes idle through the other voxels' rounds)
        } else if (minem == 0u) {
          open = false;   // the voxel has no further run
        } else if (take > 0u) {
          const bool lead = start && (uint32_t)sub < take;
          const uint32_t val1 = (uint32_t)__shfl_down((int)val, 1, 8), val2 = (uint32_t)__shfl_down((int)val, 2, 8);
          const uint32_t gt1 = (uint32_t)__shfl_down((int)gtile, 1, 8), gt2 = (uint32_t)__shfl_down((int)gtile, 2, 8);
          uint32_t mA[kMaskWords], mB[kMaskWords], mC[kMaskWords];
#pragma unroll
          for (int w = 0; w < kMaskWords; ++w) mA[w] = mB[w] = mC[w] = 0u;
          auto load_mask = [&](uint32_t v, uint32_t* m) {
            const uint4* m4 = reinterpret_cast<const uint4*>(src.base + (size_t)v * src.words);
#pragma unroll
            for (int q = 0; q < kMaskWords / 4; ++q) {
              const uint4 a = m4[q];
              m[4 * q] = a.x; m[4 * q + 1] = a.y; m[4 * q + 2] = a.z; m[4 * q + 3] = a.w;
            }
          };
          if (lead) load_mask(val, mA);
          // (three bands in four have ONE run: a wave whose bands all do skips the second and third run's registers, loads
          // and empty bit loops altogether)
          const bool wide = __any(lead && nb > 1u);
          if (wide) {
            if (lead && nb > 1u) load_mask(val1, mB);
            if (lead && nb > 2u) load_mask(val2, mC);
          }
          uint32_t cnt = 0;
#pragma unroll
          for (int w = 0; w < kMaskWords; ++w) cnt += (uint32_t)__popc(mA[w]);
          if (wide) {
#pragma unroll
            for (int w = 0; w < kMaskWords; ++w) cnt += (uint32_t)(__popc(mB[w]) + __popc(mC[w]));
          }
          uint32_t inc = cnt;
#pragma unroll
          for (int d = 1; d < 8; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)inc, d, 8);
            if (sub >= d) inc += up;
          }
          uint32_t at = havev + inc - cnt;
          const GridTile tA = grid_tile_of_band(grid, gtile, band);
          // (the other runs of the band: its column blocks)
          const uint32_t colB = (gt1 - band * grid.ntx) * (uint32_t)kGridTileW, colC = (gt2 - band * grid.ntx) * (uint32_t)kGridTileW;
          uint32_t rowkey = (tA.cloud << grid.key_bits) | (tA.row0 * grid.gw);
          auto put = [&](uint32_t bits, uint32_t key0) {
            while (bits && at < needv) {
              const int bpos = __ffs((int)bits) - 1;
              bits &= bits - 1u;
              stage[wid][vl][at++] = key0 + (uint32_t)bpos;
            }
          };
          if (wide) {
#pragma unroll
            for (int w = 0; w < kMaskWords; ++w) {
              put(mA[w], rowkey + tA.col0);
              put(mB[w], rowkey + colB);
              put(mC[w], rowkey + colC);
              rowkey += grid.gw;
            }
          } else {
#pragma unroll
            for (int w = 0; w < kMaskWords; ++w) {
              put(mA[w], rowkey + tA.col0);
              rowkey += grid.gw;
            }
          }
          havev += (uint32_t)__shfl((int)inc, 7, 8);
          jn += take;
          open = havev < needv && !(take == nvalid && nvalid < 8u);
        } else {
          // ---- the general form, for the band that starts at run jn: its runs [jn, jn + nbt), eight at a time per grid row
          const uint32_t band0 = (uint32_t)__shfl((int)band, 0, 8);
          uint32_t nbt = 0u;
          for (;; nbt += 8u) {
            const uint32_t jc = jn + nbt + (uint32_t)sub;
            const bool in = jc < nd && skeys[jc] == keyv && grid_band(grid, run_tile(jc)) == band0;
            const uint32_t cm = (uint32_t)(__ballot(in) >> gshift) & 0xFFu;
            if (cm != 0xFFu) {
              nbt += (uint32_t)__popc(cm);
              break;
            }
          }
          for (uint32_t w = 0; w < (uint32_t)kMaskWords && havev < needv; ++w)
            for (uint32_t c0 = 0; c0 < nbt; c0 += 8u) {
              const uint32_t jc = jn + c0 + (uint32_t)sub;
              GridTile t{0u, 0u, 0u};
              uint32_t bits = 0u;
              if (c0 + (uint32_t)sub < nbt) {
                t = grid_tile(grid, run_tile(jc));
                bits = src.base[(size_t)sorted_val[jc] * src.words + w];
              }
              stage_row(bits, t, w);
            }
          jn += nbt;
          open = havev < needv;
        }
      }
      if (sub == 0 && vl < kFoldGroup) have[wid][vl] = min(havev, needv);
    } else {
      const int vl = lane >> 3, sub = lane & 7;
      const uint32_t j0v = (uint32_t)__shfl((int)my_j0, vl), keyv = (uint32_t)__shfl((int)my_key, vl);
      const uint32_t cwv = (uint32_t)__shfl((int)my_col, vl) >> 24;
      const uint32_t needv = cwv >= 254u ? 0u : 254u - cwv;
      uint32_t havev = 0;
      bool open = (uint32_t)vl < nv && needv > 0u;
      for (uint32_t jb = 0; __any(open); jb += 8) {
        const uint32_t j = j0v + jb + (uint32_t)sub;
        const uint32_t jc = min(j, nd - 1u);
        const uint32_t key_j = skeys[jc], val_j = sorted_val[jc];   // (together: see the depth-image form above)
        const bool mine = open && j < nd && key_j == keyv;
        uint32_t m[kMaskWords];
#pragma unroll
        for (int w = 0; w < kMaskWords; ++w) m[w] = 0;
        size_t p0 = 0;
        uint32_t cnt = 0;
        uint32_t sp[kWireSpans];   // (received runs: the spans of the record)
#pragma unroll
        for (int q = 0; q < (int)kWireSpans; ++q) sp[q] = kSpanNone;
        const bool wire = src.words == kWireRun;
        if (mine) {
          const uint32_t val = val_j;
          const uint32_t* run = src.base + (size_t)val * src.words;
          uint32_t gt;
          if (wire) {
            const uint2 a = reinterpret_cast<const uint2*>(run)[1], c = reinterpret_cast<const uint2*>(run)[2];
            gt = a.x >> 12;
            sp[0] = a.y & 0xFFFFu; sp[1] = a.y >> 16; sp[2] = c.x & 0xFFFFu; sp[3] = c.x >> 16; sp[4] = c.y & 0xFFFFu; sp[5] = c.y >> 16;
#pragma unroll
            for (int q = 0; q < (int)kWireSpans; ++q) cnt += sp[q] == kSpanNone ? 0u : (sp[q] >> 9) + 1u;
          } else {
            const uint4* m4 = reinterpret_cast<const uint4*>(run);
#pragma unroll
            for (int q = 0; q < kMaskWords / 4; ++q) {
              const uint4 a = m4[q];
              m[4 * q] = a.x; m[4 * q + 1] = a.y; m[4 * q + 2] = a.z; m[4 * q + 3] = a.w;
            }
            gt = src.tmap.tile_of(val >> src.r1_log2);
#pragma unroll
            for (int w = 0; w < kMaskWords; ++w) cnt += (uint32_t)__popc(m[w]);
          }
          p0 = src.tile_first ? (size_t)src.tile_first[gt] : (size_t)tile_span(src.offsets, src.nclouds, gt, kWalkRays).first;
        }
        uint32_t inc = cnt;   // prefix over the voxel's eight lanes
#pragma unroll
        for (int d = 1; d < 8; d <<= 1) {
          const uint32_t up = (uint32_t)__shfl_up((int)inc, d, 8);
          if (sub >= d) inc += up;
        }
        uint32_t at = havev + inc - cnt;
        if (wire) {
#pragma unroll
          for (int q = 0; q < (int)kWireSpans; ++q) {
            if (sp[q] == kSpanNone) continue;
            const uint32_t first = sp[q] & 0x1FFu, len = (sp[q] >> 9) + 1u;
            for (uint32_t k = 0; k < len && at < needv; ++k) stage[wid][vl][at++] = (uint32_t)(p0 + (size_t)(first + k));
          }
        } else {
#pragma unroll
          for (int w = 0; w < kMaskWords; ++w) {
            uint32_t bits = m[w];
            while (bits && at < needv) {
              const int bpos = __ffs((int)bits) - 1;
              bits &= bits - 1u;
              stage[wid][vl][at++] = (uint32_t)(p0 + (size_t)(w * 32 + bpos));   // the visit's point; its colour below
            }
          }
        }
        havev += (uint32_t)__shfl((int)inc, 7, 8);
        const uint32_t group_mine = (uint32_t)(__ballot(mine) >> (8 * vl)) & 0xFFu;
        open = open && group_mine == 0xFFu && havev < needv;   // more runs of this voxel may follow, and they still count
      }
      if (sub == 0 && vl < kFoldGroup) have[wid][vl] = min(havev, needv);
    }
    __builtin_amdgcn_wave_barrier();
    // ---- the colours of the staged visits, all 64 lanes, eight independent slots per lane and round (a lane that
    // fetched the colour of each of its visits inside the bit loop above paid one memory latency per visit: 0.67 ms
    // for the first call over a fresh map)
#pragma unroll
    for (int c = 0; c < kFoldGroup * kFoldSteps / 64 / 8; ++c) {
      uint32_t idx[8], colr[8];
      bool ok[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = c * 8 + u, v = (i * 64) / kFoldSteps, k = (i * 64) % kFoldSteps + lane;
        ok[u] = (uint32_t)k < have[wid][v];
        idx[u] = ok[u] ? stage[wid][v][k] : 0u;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        colr[u] = 0;
        if (ok[u]) {
          const uint8_t* px;
          if constexpr (kGrid) {   // (a staged visit is its order key: image, grid pixel)
            const uint32_t r = idx[u] & ((1u << grid.key_bits) - 1u), m = r / grid.gw, n = r - m * grid.gw;
            px = rgb + (size_t)(idx[u] >> grid.key_bits) * grid.bgr_image_stride + (size_t)(m * grid.step) * grid.bgr_pitch +
                 (size_t)(n * grid.step) * 3u;
          } else {
            px = rgb + 3 * (size_t)idx[u];
          }
          colr[u] = colour_roundtrip(px[0]) | (colour_roundtrip(px[1]) << 8) | (colour_roundtrip(px[2]) << 16);
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = c * 8 + u, v = (i * 64) / kFoldSteps, k = (i * 64) % kFoldSteps + lane;
        if (ok[u]) stage[wid][v][k] = colr[u];
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- fold: lane = voxel * 4 + channel
    {
      const int fv = lane >> 2, fc = lane & 3;
      const uint32_t col0 = (uint32_t)__shfl((int)my_col, fv);
      const uint32_t key = (uint32_t)__shfl((int)my_key, fv);
      const bool active = (uint32_t)fv < nv && fv < kFoldGroup;
      const uint32_t cw0 = col0 >> 24;
      const uint32_t steps = (active && cw0 < 254u) ? min(have[wid][fv & (kFoldGroup - 1)], 254u - cw0) : 0u;
      uint32_t ch = (col0 >> (8 * fc)) & 255u;
      if (fc < 3) {
        for (uint32_t k = 0; k < steps; ++k) {
          const uint32_t cw = cw0 + k;
          const uint32_t x = (stage[wid][fv & (kFoldGroup - 1)][k] >> (8 * fc)) & 255u;
          ch = (uint32_t)(uint8_t)((float)(cw * ch + x) * rcp[cw]);
        }
      }
      const uint32_t g = (uint32_t)__shfl_down((int)ch, 1), bl = (uint32_t)__shfl_down((int)ch, 2);
      if (active && fc == 0 && cw0 < 254u) {
        rgbw[key] = ch | (g << 8) | (bl << 16) | ((cw0 + steps) << 24);
        if (sat_list && cw0 + steps >= 254u) sat_list[atomicAdd(sat_count, 1u)] = key;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace
