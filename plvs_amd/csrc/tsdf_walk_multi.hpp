// walk_multi (round 6): the order-free walk of a depth-image call with the SAME block of grid pixels of several
// CONSECUTIVE key frames in one voxel table (included by tsdf_chisel.hip behind tsdf_walk.hpp).
//
// Why.  walk_fast (tsdf_walk.hpp) gives a workgroup one tile — a 32 x 16 block of grid pixels of ONE image — and the
// tile leaves one record per voxel it touched: 15 000 tiles of ~400 voxels per 100-key-frame call, 6 M records for the
// 60-100 k voxels the call updates.  Everything behind the voxel loop is a function of that number: the entries ->
// chunks -> pool slots pass of the flush (a fifth of the walk), the record writes, the segment sort, apply_chunks.
// Key frames of a camera stream overlap: the same block of pixels of the next key frame sees nine tenths of the same
// voxels (PLVS inserts a key frame every few centimetres / degrees; src/PointCloudMapping.cc:537-556 hands them over in
// order).  So a workgroup here takes a TASK — block (band, column) of images img0 .. img0 + n - 1 — walks the images
// one after the other into ONE table and flushes the records when the task ends (or earlier, when the table would not
// hold the next image: a camera that turns quickly simply gets shorter windows).  Sums are fixed point and the last
// visitor is a maximum of order keys, so records of any grouping give apply_chunks the same integers as walk_fast's:
// the maps are bit-identical.
//
// Colour runs (ColorVoxel::IntegrateSimple is order dependent until weight 254, ColorVoxel.h:91-110) stay per image: after
// the walk of image k the tile's (voxel, ray mask) runs go to the run slots of THAT image's tile exactly as walk_fast
// writes them, so compact_runs / the sort / fold_colours_masks see nothing new.  What is new is which runs are written
// at all.  A run of image i can only matter if fewer than 254 - w0 visits of the voxel precede image i; visits that
// provably precede are
//   (a) the task's own visits from its earlier images of the open window (the accumulator's count), and
//   (b) for tasks of the call's SECOND part (images >= n1): what tasks of the first part (images < n1) had counted into
//       the per-voxel plane `cnt` when they flushed.  Any value read there is a lower bound on the visits before image
//       n1 (only first-part tasks add, adds are atomic, a stale cache line is an older = smaller sum), so the rule needs
//       no ordering between workgroups: whatever is dropped is a run whose every visit the fold would have skipped.
// The fold leaves cnt zero for the next call (it visits every voxel that has a run, and a voxel that was counted has one
// from the task that counted it).
#pragma once
#include "tsdf_walk.hpp"

namespace {

#ifndef PLVS_MULTI_WAVES
#define PLVS_MULTI_WAVES 6
#endif
constexpr uint32_t kVkUnset = 0xFFFFFFFFu;   // e_vkey: the entry has no pool slot yet
constexpr uint32_t kVkDone = 0x80000000u;    //   the voxel has (at least) the visits its colour can still take
constexpr int kMultiMasks = 256;             // ray masks built per round of the run emission (LDS: the visit log's words)
constexpr uint32_t kMultiCountCap = 511u;    // an entry with more visits than this closes the window: a further image adds
                                             // up to 512, and the 32-bit sums of a record hold 1024 terms (scale_u / scale_w)

struct MultiPlan {
  uint32_t nimg;       // images of the call
  uint32_t n1;         // images [0, n1): first part (count into cnt), [n1, nimg): second part (read cnt)
  uint32_t k1, k2;     // images per task in the two parts
  uint32_t g1;         // tasks per block position in the first part = ceil(n1 / k1)
  uint32_t tpi;        // tiles per image = ntx * nty
  uint32_t fill;       // entries beyond which a window is not expected to hold another image
};

template <int E>
struct MultiShared {
  static constexpr int kEntries = E, kBucketCount = E / 4, kBucketShift = 32 - log2_of(E / 4);
  static_assert((E & (E - 1)) == 0 && E >= 1024, "power-of-two table");
  alignas(16) uint32_t ekey[E];
  uint32_t cand[kWalkRays / 64];
  int32_t worg[kWalkRays / 64][3];
  uint32_t run_total, vis_total;
  uint32_t ccode[kWalkChunks];
  int32_t cslot[kWalkChunks];
  uint32_t ccnt[kWalkChunks * kSlabs];       // (chunk, slab) counters of a flush; between flushes: the mask index of an entry
  uint16_t cbase[kWalkChunks * kSlabs];
  uint32_t nent, overflow, big;
  uint32_t any, bad2;                        // an image has a ray / the entries -> chunks pass failed (chunk cache)
  uint32_t wsum[kWalkRays / 64];
};

// The LDS of a task, at namespace scope: the image walk below is a FUNCTION of its own (not inlined), and a function
// reaches LDS by name only (through a pointer argument its accesses would be flat_* instructions).
template <int E>
struct MultiLds {
  MultiShared<E> S;
  uint32_t raw[4 * E];                        // the accumulators
  // entry of visit j of ray r of the current image at [j * kWalkRays + r]; once every ray has turned its log into mask
  // indices (registers) the same words hold the ray masks of a round
  uint16_t vlog[kLogLen * kWalkRays];
  uint32_t e_vkey[E];                         // pool slot * 4096 + voxel | kVkDone; kVkUnset
  uint8_t e_need[E];                          // visits the voxel's colour could still take when the entry was resolved
  uint8_t e_ci[E];                            // the entry's chunk in the cache
  int32_t org[3];                             // origin of the open window's keys
  Params P;                                   // the call's parameters and scales, for the image walk
  float scale_u, scale_w;
};
template <int E>
__shared__ MultiLds<E> g_multi;

__device__ __forceinline__ const void* uniform_ptr(const void* p) {   // a wave-uniform pointer -> scalar registers
  const unsigned long long v = (unsigned long long)p;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return (const void*)(((unsigned long long)hi << 32) | lo);
}

// One image of a task: the rays of the block, the origin vote of a fresh window, barrier 0, the voxel loop.  Returns the
// visits of this thread's ray.  NOT inlined: inside the task's image loop the voxel loop shared its register allocation
// with everything the loop's other phases keep alive (addresses of a dozen LDS arrays, constants, the task's state) and
// spilled its own state — a scratch reload per voxel step, or 124 registers and two workgroups per CU instead of three.
// As a function it has the registers to itself; the caller saves what it needs around the call, once per image.
// flags: k (the image's index inside the task: the last visitor of an entry is (k << 9 | ray)) | fresh window << 8.
template <int E>
__device__ __attribute__((noinline)) uint32_t multi_walk_image(const Pose* pose_arg, const GridSrc* grid_arg, uint32_t* err_arg,
                                                               uint32_t img_arg, uint32_t rowcol_arg, uint32_t flags_arg) {
  MultiLds<E>& L = g_multi<E>;
  MultiShared<E>& S = L.S;
  const Pose* __restrict__ const posep = static_cast<const Pose*>(uniform_ptr(pose_arg));
  const GridSrc* __restrict__ const grid = static_cast<const GridSrc*>(uniform_ptr(grid_arg));
  uint32_t* const err = const_cast<uint32_t*>(static_cast<const uint32_t*>(uniform_ptr(err_arg)));
  const uint32_t img = (uint32_t)__builtin_amdgcn_readfirstlane((int)img_arg);
  const uint32_t rowcol = (uint32_t)__builtin_amdgcn_readfirstlane((int)rowcol_arg);
  const uint32_t flags = (uint32_t)__builtin_amdgcn_readfirstlane((int)flags_arg);
  const uint32_t k = flags & 0xFFu;
  const bool fresh = (flags >> 8) != 0u;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  Params P;
  P.resolution = uniform_f(L.P.resolution); P.round_to_voxel = uniform_f(L.P.round_to_voxel); P.half_voxel = uniform_f(L.P.half_voxel);
  P.rounding = uniform_f(L.P.rounding); P.diag = uniform_f(L.P.diag); P.tq = uniform_f(L.P.tq); P.tl = uniform_f(L.P.tl);
  P.tc = uniform_f(L.P.tc); P.ts = uniform_f(L.P.ts); P.weight = uniform_f(L.P.weight);
  P.shard_rank = 0; P.shard_count = 1;
  const float scale_u = uniform_f(L.scale_u), scale_w = uniform_f(L.scale_w);
  const GridTile gt{img, rowcol & 0xFFFFu, rowcol >> 16};
  const Pose& pose = *posep;
  Ray ray;
  const bool walks = tile_ray_src(P, nullptr, grid, gt, pose, 0u, (uint32_t)tid, &ray, err);
  const float wu = walks ? P.weight / (2.0f * ray.truncation) : 0.0f;
  if (fresh) {   // (uniform) origin of the window's keys: below the start voxel of its first walking ray, on a chunk boundary
    const unsigned long long wm = __ballot(walks);
    if (lane == 0) S.cand[wid] = wm ? 1u : 0u;
    if (wm && lane == __ffsll((long long)wm) - 1)
      for (int a = 0; a < 3; ++a) S.worg[wid][a] = origin_of((int)floorf(ray.start[a]));
  }
  if (walks) S.any = 1u;
  __syncthreads();                                                                           // ---- barrier 0
  int ox, oy, oz;
  if (fresh) {
    int w0 = 0;
#pragma unroll
    for (int w = kWalkRays / 64 - 1; w >= 0; --w) w0 = S.cand[w] ? w : w0;
    ox = __builtin_amdgcn_readfirstlane(S.worg[w0][0]);
    oy = __builtin_amdgcn_readfirstlane(S.worg[w0][1]);
    oz = __builtin_amdgcn_readfirstlane(S.worg[w0][2]);
    if (tid == 0) {   // (the later images of the window and the caller read it behind barriers 1 / 0; garbage if no ray walks)
      L.org[0] = ox; L.org[1] = oy; L.org[2] = oz;
    }
  } else {
    ox = __builtin_amdgcn_readfirstlane(L.org[0]);
    oy = __builtin_amdgcn_readfirstlane(L.org[1]);
    oz = __builtin_amdgcn_readfirstlane(L.org[2]);
  }
  uint32_t nv = 0;
  if (walks) {
    const int reach = lean_reach(ray);
    const int rx = (int)floorf(ray.start[0]) - ox, ry = (int)floorf(ray.start[1]) - oy, rz = (int)floorf(ray.start[2]) - oz;
    const bool fits = min(min(rx, ry), rz) - reach >= 0 && max(max(rx, ry), rz) + reach <= 1023;
    if (!fits) S.overflow = 1u;   // the rays of the window are too far apart
    else
      nv = walk_lean<true, true>(P, pose, ray, S, ox, oy, oz, tid, wu * scale_u, (uint32_t)__float2int_rn(wu * scale_w),
                                 reinterpret_cast<int32_t*>(L.raw), reinterpret_cast<unsigned long long*>(L.raw + E), L.raw + 3 * E,
                                 L.vlog, (k << 9) | (uint32_t)tid);
  }
  return nv;
}

template <int E>
__global__ __launch_bounds__(kWalkRays, PLVS_MULTI_WAVES) void walk_multi(
    Params P, float scale_u, float scale_w, const Pose* __restrict__ poses, Directory dir, int32_t* __restrict__ num_chunks,
    WalkCounters* __restrict__ ctr, const uint32_t* __restrict__ rgbw, uint32_t* __restrict__ cnt, AccOut out, RunOut runs,
    uint32_t rec_stride, uint32_t* __restrict__ deferred, uint32_t* __restrict__ ndeferred, const GridSrc* __restrict__ grid,
    MultiPlan plan) {
  constexpr int kPer = E / kWalkRays;
  constexpr int kLimit = E * 7 / 8;
  static_assert(sizeof(((MultiShared<E>*)nullptr)->ccnt) >= E * sizeof(uint16_t), "the mask indices live in the (chunk, slab) counters");
  MultiLds<E>& L = g_multi<E>;
  MultiShared<E>& S = L.S;
  uint32_t* const raw = L.raw;
  uint16_t* const vlog = L.vlog;
  uint32_t* const masks = reinterpret_cast<uint32_t*>(L.vlog);
  static_assert(kMultiMasks * kMaskWords * sizeof(uint32_t) == sizeof(L.vlog), "a round's masks fill the visit log");
  uint32_t* const e_vkey = L.e_vkey;
  uint8_t* const e_need = L.e_need;
  uint8_t* const e_ci = L.e_ci;
  uint16_t* const e_midx = reinterpret_cast<uint16_t*>(S.ccnt);
  int32_t* const e_wuu = reinterpret_cast<int32_t*>(raw);
  unsigned long long* const e_wc = reinterpret_cast<unsigned long long*>(raw + E);
  uint32_t* const e_last = raw + 3 * E;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

  // ---- the task
  const uint32_t gi = blockIdx.x / plan.tpi, bpos = blockIdx.x - gi * plan.tpi;
  uint32_t img0, n;
  bool adds;
  if (gi < plan.g1) {
    img0 = gi * plan.k1;
    n = min(plan.k1, plan.n1 - img0);
    adds = cnt != nullptr;
  } else {
    img0 = plan.n1 + (gi - plan.g1) * plan.k2;
    n = min(plan.k2, plan.nimg - img0);
    adds = false;
  }
  const bool reads = cnt != nullptr && !adds;
  const uint32_t band = bpos / grid->ntx, row0 = band * (uint32_t)kGridTileH, col0 = (bpos - band * grid->ntx) * (uint32_t)kGridTileW;

  auto reset_window = [&](int tid) {   // (inside the image loop: the opaque copy of the thread id, see below)
    subtile_reset(S, tid);
#pragma unroll
    for (int q = 0; q < 4 * kPer; ++q) raw[tid + q * kWalkRays] = 0u;
#pragma unroll
    for (int q = 0; q < kPer; ++q) e_vkey[tid + q * kWalkRays] = kVkUnset;
    if (tid == 0) {
      S.run_total = 0;
      S.vis_total = 0;
      S.big = 0;
      S.bad2 = 0;
    }
  };
  if (tid == 0) {
    S.any = 0;
    L.P = P;
    L.scale_u = scale_u;
    L.scale_w = scale_w;
  }
  WALK_PROF_BEGIN();
  reset_window(tid);
  __syncthreads();   // (the parameters are in LDS for the image walk)
  uint32_t wfirst = 0;        // first image of the open window (index inside the task)
  bool fresh = true;          // the window has no key origin yet
  uint32_t prev_nent = 0;     // entries when the previous image of the window was done

  for (uint32_t k = 0; k < n; ++k) {
    const uint32_t img = img0 + k, gtile = img * plan.tpi + bpos;
    const GridTile gt{img, row0, col0};
    const Pose& pose = poses[img];
    const uint32_t nv = multi_walk_image<E>(&poses[img], grid, &ctr->err, img, row0 | (col0 << 16), k | (fresh ? 0x100u : 0u));
    WALK_PROF(2);   // the voxel loop of this wave
    // (the indices below are taken from an opaque copy of the thread id: computed from `tid` itself, the addresses of this
    // thread's slots in a dozen LDS arrays are loop invariants, the compiler hoists them out of the IMAGE loop and they
    // sit in registers through the voxel loop — which then spills its own state: a scratch reload per voxel step)
    int tq = tid;
    asm volatile("" : "+v"(tq));
    __syncthreads();                                                                         // ---- barrier 1
    WALK_PROF(3);
    bool bad = __builtin_amdgcn_readfirstlane((int)S.overflow) != 0;   // (set by the walk only: table full, key box)
    if (fresh && __builtin_amdgcn_readfirstlane((int)S.any) != 0) fresh = false;   // (the window has its origin: L.org)
    const int ox = __builtin_amdgcn_readfirstlane(L.org[0]), oy = __builtin_amdgcn_readfirstlane(L.org[1]),
              oz = __builtin_amdgcn_readfirstlane(L.org[2]);
    if (tq == 0) S.any = 0u;                                          // (read between barriers 0 and 1, set again before the next 0)
    // ---- this thread's table slots: new entries get their pool slot and colour state; which entries leave a run
    uint32_t vk[kPer];
    uint32_t need_run = 0, nneed = 0, nnew = 0;
#pragma unroll
    for (int q = 0; q < kPer; ++q) vk[q] = kVkUnset;
    if (!bad) {
      uint32_t tk[kPer];
      bool fresh_e[kPer];
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        const int e = tq + q * kWalkRays;
        tk[q] = S.ekey[e];
        fresh_e[q] = false;
        if (tk[q] == kKeyEmpty) continue;
        vk[q] = e_vkey[e];
        if (vk[q] != kVkUnset) continue;
        fresh_e[q] = true;
        ++nnew;
        const uint32_t key = voxel_key(tk[q]);
        bool won;
        const int ci = chunk_cache_insert(S, chunk_code(key), &won);
        if (ci < 0) {
          S.bad2 = 1u;     // more chunks than the cache holds
          continue;
        }
        int slot = S.cslot[ci];
        if (slot < 0) {
          int cx, cy, cz;
          chunk_of_code(chunk_code(key), ox, oy, oz, &cx, &cy, &cz);
          slot = dir_find_or_insert(dir, cx, cy, cz, num_chunks, &ctr->err);
          if (slot >= 0) S.cslot[ci] = slot;
        }
        if (slot < 0) continue;   // (the pool is full: the call fails with the error dir_find_or_insert has set)
        e_ci[e] = (uint8_t)ci;
        vk[q] = (uint32_t)slot * (uint32_t)kChunkVox + voxel_in_chunk(key);
      }
      uint32_t cw[kPer], seen[kPer];
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        cw[q] = 0xFF000000u;
        seen[q] = 0u;
        if (fresh_e[q] && vk[q] != kVkUnset) {
          cw[q] = rgbw[vk[q]];
          if (reads) seen[q] = cnt[vk[q]];
        }
      }
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        const int e = tq + q * kWalkRays;
        if (tk[q] == kKeyEmpty || vk[q] == kVkUnset) continue;
        uint32_t need;
        if (fresh_e[q]) {
          const uint32_t w0 = cw[q] >> 24;
          need = w0 >= 254u ? 0u : 254u - w0;
          need -= min(need, seen[q]);
          e_need[e] = (uint8_t)need;
          if (need == 0u) vk[q] |= kVkDone;
        } else {
          need = e_need[e];
        }
        const uint32_t count = (uint32_t)(e_wc[e] >> 32);
        const bool visited = (e_last[e] >> 9) == k;   // (an entry is in the table because some image of the window visited it)
        if (visited && !(vk[q] & kVkDone)) {
          need_run |= 1u << q;
          ++nneed;
        }
        if (count >= need) vk[q] |= kVkDone;          // ... from the next image on
        if (count > kMultiCountCap) S.big = 1u;
        e_vkey[e] = vk[q];
      }
      const uint32_t wn = wave_sum(nnew);
      if (lane == 0 && wn) atomicAdd(&S.nent, wn);
      const uint32_t v = wave_sum(nv);
      if (lane == 0 && v) atomicAdd(&S.vis_total, v);
    }
    const uint32_t inc = wave_scan_incl(nneed);
    if (lane == 63) S.wsum[wid] = inc;
    WALK_PROF(4);   // entries -> chunks, colour state, which entries leave a run
    __syncthreads();                                                                         // ---- barrier 2
    WALK_PROF(5);
    bad = bad || __builtin_amdgcn_readfirstlane((int)S.bad2) != 0;
    const uint32_t nent = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.nent);
    if (bad || nent > (uint32_t)kLimit) {
      // the window does not fit (table, key box, chunk cache): its images go to the one-tile kernels behind this launch,
      // whole — nothing of the window has left the workgroup but runs, and their counts are taken back here
      __syncthreads();
      if (tq == 0)
        for (uint32_t j = wfirst; j <= k; ++j) {
          const uint32_t t = (img0 + j) * plan.tpi + bpos;
          out.seg_cnt[t] = 0;
          runs.run_cnt[t] = 0;
          out.tile_visits[t] = 0;
          deferred[atomicAdd(ndeferred, 1u)] = t | 0x80000000u;
        }
      reset_window(tq);
      wfirst = k + 1;
      fresh = true;
      prev_nent = 0;
      continue;
    }
    // ---- runs of this image: bit r of a voxel's mask = ray r of the tile visits it
    uint32_t wbase = 0, nruns = 0;
#pragma unroll
    for (int w = 0; w < kWalkRays / 64; ++w) {
      const uint32_t v = S.wsum[w];
      if (w < wid) wbase += v;
      nruns += v;
    }
    nruns = (uint32_t)__builtin_amdgcn_readfirstlane((int)nruns);
    if (nruns) {   // (uniform)
      const bool fits_runs = nruns <= (1u << runs.r1_log2);
      if (!fits_runs && tq == 0) {
        atomicOr(&ctr->err, kErrScratch);
        atomicMax(&ctr->run_need, nruns);
      }
      {
        uint32_t m = wbase + inc - nneed;
#pragma unroll
        for (int q = 0; q < kPer; ++q) e_midx[tq + q * kWalkRays] = (need_run & (1u << q)) ? (uint16_t)m++ : (uint16_t)0xFFFFu;
      }
      // every ray turns its log into mask indices (two per register) — then the log's words are free for the masks
      __syncthreads();   // e_midx complete
      uint32_t mreg[kLogLen / 2];
      const uint32_t logged = min(nv, (uint32_t)kLogLen);
#pragma unroll
      for (int j = 0; j < kLogLen; j += 2) {
        const uint32_t a = (uint32_t)j < logged ? (uint32_t)e_midx[vlog[j * kWalkRays + tq]] : 0xFFFFu;
        const uint32_t b = (uint32_t)(j + 1) < logged ? (uint32_t)e_midx[vlog[(j + 1) * kWalkRays + tq]] : 0xFFFFu;
        mreg[j / 2] = a | (b << 16);
      }
      for (uint32_t r0 = 0; fits_runs && r0 < nruns; r0 += kMultiMasks) {
        __syncthreads();   // the logs have been read / the previous round's masks are out
#pragma unroll
        for (int q = 0; q < kMultiMasks * kMaskWords / kWalkRays; ++q) masks[tq + q * kWalkRays] = 0u;
        __syncthreads();
        if (nv) {
          const uint32_t bit = 1u << (tq & 31);
          uint32_t* const mine = masks + ((uint32_t)tq >> 5);
#pragma unroll
          for (int j = 0; j < kLogLen; ++j) {
            const uint32_t m = ((mreg[j / 2] >> (16 * (j & 1))) & 0xFFFFu) - r0;   // 0xFFFF - r0 stays out of range
            if (m < (uint32_t)kMultiMasks) atomicOr(&mine[m * kMaskWords], bit);
          }
          if (nv > (uint32_t)kLogLen) {   // the log is full: the rest of the ray is walked again
            Ray ray2;
            if (tile_ray_src(P, nullptr, grid, gt, pose, 0u, (uint32_t)tq, &ray2, &ctr->err))
              walk_one(P, pose, ray2, 0u, 0xFFFFFFFFu, [&](uint32_t j, int vx, int vy, int vz, float) {
                if (j >= (uint32_t)kLogLen) {
                  uint32_t key;
                  const int e = rel_key(vx, vy, vz, ox, oy, oz, &key) ? table_find(S, table_key(key)) : -1;
                  const uint32_t m = e >= 0 ? (uint32_t)e_midx[e] - r0 : 0xFFFFFFFFu;
                  if (m < (uint32_t)kMultiMasks) atomicOr(&mine[m * kMaskWords], bit);
                }
                return true;
              });
          }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
          const uint32_t m = (uint32_t)e_midx[tq + q * kWalkRays];
          if (m == 0xFFFFu || m < r0 || m >= r0 + (uint32_t)kMultiMasks) continue;
          const size_t d = ((size_t)gtile << runs.r1_log2) + m;
          runs.dkey[d] = vk[q] & ~kVkDone;
          const uint32_t* mk = masks + (m - r0) * kMaskWords;
          uint4* dst = reinterpret_cast<uint4*>(runs.masks + d * kMaskWords);
#pragma unroll
          for (int w = 0; w < kMaskWords / 4; ++w) dst[w] = make_uint4(mk[4 * w], mk[4 * w + 1], mk[4 * w + 2], mk[4 * w + 3]);
        }
      }
      if (tq == 0) runs.run_cnt[gtile] = min(nruns, 1u << runs.r1_log2);
    } else if (tq == 0) {
      runs.run_cnt[gtile] = 0;
    }
    WALK_PROF(7);   // runs
    // ---- does the window stay open?  It closes with the task, when the table would not hold another image like this
    // one, and before a record's 32-bit sums could overflow.
    const bool last = k + 1 == n;
    const uint32_t grown = nent - prev_nent;
    const uint32_t expect = (k == wfirst ? nent / 4u : 2u * grown) + 32u;
    const bool big = __builtin_amdgcn_readfirstlane((int)S.big) != 0;   // (written before barrier 2, cleared only by reset_window)
    const bool flush = last || big || nent + expect > plan.fill;
    if (!flush || nent == 0u) {
      if (tq == 0) {
        out.seg_cnt[gtile] = 0;
        out.tile_visits[gtile] = 0;
      }
      prev_nent = nent;
      if (flush && !last) {   // (an empty window: nothing to reset but the origin)
        wfirst = k + 1;
        fresh = true;
      }
      continue;
    }
    if (tq == 0) {   // (developer trace: windows flushed, and how many of them early for the count cap / the table)
      atomicAdd(&ctr->split_tiles, 1u);
      if (!last && big) atomicAdd(&ctr->ncold, 1u);
      else if (!last) atomicAdd(&ctr->over_small, 1u);
    }
    // ---- records: the (chunk, slab) groups are placed by a scan over the cache's 64 chunks, as walk_fast does
    __syncthreads();   // (the mask indices in S.ccnt are dead)
    S.ccnt[tq] = 0u;
    static_assert(kWalkChunks * kSlabs == kWalkRays, "one counter per thread");
    __syncthreads();
    uint32_t rank[kPer];
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      rank[q] = 0;
      if (vk[q] == kVkUnset) continue;
      const int e = tq + q * kWalkRays;
      const uint32_t vid = (vk[q] & ~kVkDone) % (uint32_t)kChunkVox;
      rank[q] = atomicAdd(&S.ccnt[(int)e_ci[e] * kSlabs + (int)(vid / kSlabVox)], 1u);
      if (adds && e_need[e] != 0u) atomicAdd(&cnt[vk[q] & ~kVkDone], min((uint32_t)(e_wc[e] >> 32), 254u));
    }
    __syncthreads();
    const uint32_t rbase = gtile * rec_stride;
    {
      uint32_t sub[kSlabs], c = 0;
#pragma unroll
      for (int s = 0; s < kSlabs; ++s) {
        sub[s] = c;
        c += S.ccnt[lane * kSlabs + s];
      }
      const uint32_t both = wave_scan_incl(c | (c ? 1u << 16 : 0u));
      const uint32_t rinc = both & 0xFFFFu, sinc = both >> 16;
#pragma unroll
      for (int s = 0; s < kSlabs; ++s) S.cbase[lane * kSlabs + s] = (uint16_t)(rinc - c + sub[s]);
      if (wid == 0) {
        const uint32_t stot = (uint32_t)__builtin_amdgcn_readlane((int)sinc, 63);
        if (lane == 0) {
          out.seg_cnt[gtile] = stot;
          out.tile_visits[gtile] = S.vis_total;
        }
        if (c) {
          const uint32_t sg = gtile * (uint32_t)kWalkChunks + sinc - 1u;
          out.seg[2 * (size_t)sg] = make_uint4((uint32_t)S.cslot[lane], rbase + rinc - c, c, gtile);
          out.seg[2 * (size_t)sg + 1] = pack_suboffsets(sub);
          if (out.chunk_nseg) atomicAdd(&out.chunk_nseg[S.cslot[lane]], 1u);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      if (vk[q] == kVkUnset) continue;
      const int e = tq + q * kWalkRays;
      const uint32_t vid = (vk[q] & ~kVkDone) % (uint32_t)kChunkVox;
      const unsigned long long wc = e_wc[e];
      const uint32_t el = e_last[e];
      const uint4 r = make_uint4(vid | ((uint32_t)(wc >> 32) << 12),
                                 grid_order_key(*grid, GridTile{img0 + (el >> 9), row0, col0}, el & 511u),
                                 (uint32_t)e_wuu[e], (uint32_t)wc);
      out.rec[rbase + S.cbase[(int)e_ci[e] * kSlabs + (int)(vid / kSlabVox)] + rank[q]] = r;
    }
    WALK_PROF(6);   // records
    if (!last) {
      __syncthreads();   // (the group bases and counters have been read)
      reset_window(tq);
      wfirst = k + 1;
      fresh = true;
      prev_nent = 0;
    }
  }
  WALK_PROF(8);
  WALK_PROF_END();
}

}  // namespace
