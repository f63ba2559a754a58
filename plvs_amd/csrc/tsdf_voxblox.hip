// TSDF integrate for the voxblox back end (PointCloudMapVoxblox::InsertCloud ->
// TsdfServer::insertPointCloud -> SimpleTsdfIntegrator::integratePointCloud).
//
// Same device pipeline as the chisel path (tsdf_chisel.hip): count -> scan ->
// fill -> stable radix sort by voxel -> expand (order-independent operands) ->
// chain (one thread folds each voxel's records in the reference's visiting
// order).  The visiting order is voxblox's ThreadSafeIndex "mixed" order; a
// record's sequence number is the position of its point in that order, so the
// single-thread schedule of the reference is reproduced exactly.
//
// HBM layout: three planes distance / weight / rgba of max_blocks*4096 dwords
// (12 B per voxel, 48 KiB per block, voxel.h:12-18), the shared block directory
// (tsdf_directory.hpp), and per call one (u32 key, u32 sequence) pair plus
// 12 B of operands per voxel visit.
#include <algorithm>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "common.hpp"
#include "device_utils.hpp"
#include "tsdf_directory.hpp"
#include "tsdf_voxblox_core.hpp"
#include "tsdf_voxblox_view.hpp"

using namespace plvs;
using namespace plvs::tsdf;
using namespace plvs::vbx;

namespace {

constexpr uint32_t kErrNonFinite = 8u;
constexpr int kMaxRaySteps = 1 << 16;

struct VCounters {
  uint32_t total_visits;
  int32_t num_blocks;
  uint32_t err;
  uint32_t num_heads;
  uint32_t num_updated;
  uint32_t max_run;
};

__global__ void vb_publish_counters(const VCounters* __restrict__ ctr, VCounters* __restrict__ host_ctr) {
  const uint32_t* a = reinterpret_cast<const uint32_t*>(ctr);
  uint32_t* b = reinterpret_cast<uint32_t*>(host_ctr);
  for (int k = threadIdx.x; k < (int)(sizeof(VCounters) / sizeof(uint32_t)); k += blockDim.x) b[k] = a[k];
  __threadfence_system();
}

__device__ __forceinline__ PoseRt make_pose(const float* __restrict__ Twc, int c) {
  PoseRt p;
  const float* T = Twc + 12 * c;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) p.R[3 * i + j] = T[4 * i + j];
    p.t[i] = T[4 * i + 3];
  }
  quat_from_matrix(p.R, p.q);
  return p;
}

// The poses of a call with their quaternions, once per cloud (the kernels used to redo the conversion — a square root
// and a division — for every point and every voxel visit).
__global__ void vb_pose_prep(const float* __restrict__ Twc, int nclouds, PoseRt* __restrict__ poses) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < nclouds) poses[c] = make_pose(Twc, c);
}

__device__ __forceinline__ PoseRt load_pose(const PoseRt* __restrict__ poses, int c) { return poses[c]; }

// Which point does sequence position i of the batch denote?
__device__ __forceinline__ int point_of_seq(const int32_t* __restrict__ offsets, int nclouds, int i,
                                            int* cloud) {
  const int c = cloud_of(offsets, nclouds, i);
  const int beg = offsets[c], cnt = offsets[c + 1] - beg;
  *cloud = c;
  return beg + (int)mixed_index((uint32_t)(i - beg), (uint32_t)cnt);
}

// The flavours of the ray passes.  kSimple: camera rays in the mixed visiting order.  kWorld: the
// world-cloud-with-normals flavour (make_ray_world): cloud order, no validity test; `aux` = normals, n x 3.  kMerged:
// MergedTsdfIntegrator's bundles in their integration order (make_ray_merged): xyz = merged points, `aux` = merged
// weights (n), `clr` = the bundles' clearing flags.
// kFast: FastTsdfIntegrator's rays (tsdf_voxblox_fast.hpp has decided which rays live and how many voxels each updates):
// the mixed order of kSimple, cast from the surface end, `aux` = the rays' update counts (uint32, 0 = no ray).
enum VbMode { kSimple = 0, kWorld = 1, kMerged = 2, kFast = 3 };

// The fill pass stages a wave's records in LDS: the 64 rays of a wave own ONE contiguous range of the record arrays
// (their counts were scanned in ray order), so the wave writes it with consecutive lanes on consecutive words instead of
// 64 lanes on 64 short pieces.  A wave whose range exceeds kFillStage records (carving) writes the excess directly.
constexpr int kFillStage = 1536;
template <bool kFill, int kMode>
__global__ __launch_bounds__(256) void vb_ray_pass(
    Params P, const float* __restrict__ xyz, const float* __restrict__ aux, const uint8_t* __restrict__ clr, int npoints,
    const int32_t* __restrict__ offsets, int nclouds, const PoseRt* __restrict__ Twc, Directory dir,
    VCounters* __restrict__ ctr, uint32_t* __restrict__ counts, uint32_t* __restrict__ rec_keys,
    uint32_t* __restrict__ rec_seq) {
  __shared__ uint32_t s_key[kFill ? 4 : 1][kFill ? kFillStage : 1], s_seq[kFill ? 4 : 1][kFill ? kFillStage : 1];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const bool valid = i < npoints;
  if (!kFill && !valid) return;
  uint32_t n = 0;
  const uint32_t out = (kFill && valid) ? counts[i] : 0u;
  // (lane 0 of a wave is valid whenever any lane is: the wave's range starts at its first ray's offset)
  const uint32_t wbase = kFill ? (uint32_t)__builtin_amdgcn_readfirstlane((int)out) : 0u;
  if (valid) {
    int cloud = 0;
    const int p = (kMode == kWorld || kMode == kMerged) ? i : point_of_seq(offsets, nclouds, i, &cloud);
    const float px = xyz[3 * (size_t)p], py = xyz[3 * (size_t)p + 1], pz = xyz[3 * (size_t)p + 2];
    if (!(isfinite(px) && isfinite(py) && isfinite(pz))) {
      // the reference filters such points out BEFORE the mixed order is formed;
      // PLVS's cloud generator never emits them, so refuse instead of diverging
      if (!kFill) atomicOr(&ctr->err, kErrNonFinite);
    } else {
      const PoseRt pose = load_pose(Twc, cloud);
      Ray ray;
      bool walk = true;
      if (kMode == kWorld) {
        float rs[3];
        make_ray_world(P, pose, px, py, pz, aux[3 * (size_t)p], aux[3 * (size_t)p + 1], aux[3 * (size_t)p + 2], &ray, rs);
      } else if (kMode == kMerged) {
        make_ray_merged(P, pose, px, py, pz, clr[p] != 0, &ray);
      } else {
        walk = make_ray(P, pose, px, py, pz, &ray, kMode == kFast);
      }
      uint32_t limit = 0xFFFFFFFFu;   // voxels the ray may update
      if (kMode == kFast) {
        limit = reinterpret_cast<const uint32_t*>(aux)[i];
        walk = walk && limit > 0u;
      }
      if (walk) {
        int lb[3] = {0, 0, 0}, lslot = -1;
        bool have_last = false;
        int steps = ray.steps < kMaxRaySteps ? ray.steps : kMaxRaySteps;
        if (kMode == kFast && (uint32_t)steps >= limit) steps = (int)limit - 1;
        for (int s = 0; s <= steps; ++s) {
          int g[3], b[3], vid;
          ray_step(&ray, g);
          const bool ok = block_of(P, g, b, &vid);   // no early continue: every lane takes one step per trip
          if (ok && (!have_last || b[0] != lb[0] || b[1] != lb[1] || b[2] != lb[2])) {
            lb[0] = b[0]; lb[1] = b[1]; lb[2] = b[2];
            have_last = true;
            if (kFill) {
              lslot = dir_find(dir, b[0], b[1], b[2]);
              if (lslot < 0) atomicOr(&ctr->err, kErrDirectoryMiss);
            } else {
              dir_insert(dir, b[0], b[1], b[2], &ctr->num_blocks, &ctr->err);
              if (((g[0] - b[0] * 16) | (g[1] - b[1] * 16) | (g[2] - b[2] * 16)) & ~15)
                atomicOr(&ctr->err, kErrCoordRange);  // float block lookup left the integer grid
            }
          }
          if (ok && kFill) {
            // (a directory miss is an error the host reports: the record still gets a defined key)
            const uint32_t key = lslot >= 0 ? (uint32_t)lslot * (uint32_t)kBlockVox + (uint32_t)vid : 0u;
            const uint32_t at = out + n - wbase;
            if (at < (uint32_t)kFillStage) {
              s_key[wid][at] = key;
              s_seq[wid][at] = (uint32_t)i;
            } else {
              rec_keys[out + n] = key;
              rec_seq[out + n] = (uint32_t)i;
            }
          }
          n += ok ? 1u : 0u;
        }
      }
    }
  }
  if (!kFill) {
    counts[i] = n;
    return;
  }
  // the wave's staged records leave in one piece (a wave's LDS operations execute in order: no barrier)
  uint32_t wend = out + n;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) wend = max(wend, (uint32_t)__shfl_xor((int)wend, off));
  const uint32_t cnt = min(wend - wbase, (uint32_t)kFillStage);
  __builtin_amdgcn_wave_barrier();
  for (uint32_t j = (uint32_t)lane; j < cnt; j += 64u) {
    rec_keys[wbase + j] = s_key[wid][j];
    rec_seq[wbase + j] = s_seq[wid][j];
  }
}

#include "tsdf_voxblox_fast.hpp"
#include "tsdf_voxblox_shard.hpp"

// MergedTsdfIntegrator::bundleRays, the per-point part (tsdf_integrator.cc:361-386): isPointValid -> kind (0 skipped,
// 1 normal, 2 clearing) and the voxel T_G_C * point_C ends in.  The grouping itself needs the reference's hash map and
// is done on the host (plvs_hip_tsdf_voxblox_integrate_merged).
__global__ __launch_bounds__(256) void vb_merge_keys(Params P, const float* __restrict__ xyz, int n,
                                                     const PoseRt* __restrict__ Twc, VCounters* __restrict__ ctr,
                                                     uint8_t* __restrict__ kind, int32_t* __restrict__ g) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float px = xyz[3 * (size_t)i], py = xyz[3 * (size_t)i + 1], pz = xyz[3 * (size_t)i + 2];
  uint8_t k = 0;
  int v[3] = {0, 0, 0};
  if (!(isfinite(px) && isfinite(py) && isfinite(pz))) {
    atomicOr(&ctr->err, kErrNonFinite);
  } else {
    const float ray_distance = sqrtf(vsum3(px * px, py * py, pz * pz));
    if (ray_distance < P.min_ray) k = 0;
    else if (ray_distance > P.max_ray) k = P.allow_clear ? 2 : 0;
    else k = 1;
    if (k) {
      const PoseRt pose = load_pose(Twc, 0);
      float pG[3];
      quat_transform(pose, px, py, pz, pG);
      for (int c = 0; c < 3; ++c) v[c] = (int)floorf(pG[c] * P.voxel_size_inv + 1e-6f);
    }
  }
  kind[i] = k;
  g[3 * (size_t)i] = v[0];
  g[3 * (size_t)i + 1] = v[1];
  g[3 * (size_t)i + 2] = v[2];
}

// integrateVoxel's fold of a bundle's points into one (tsdf_integrator.cc:404-416), one thread per bundle: the
// recurrence is short (a handful of points per voxel) and sequential in float.
__global__ __launch_bounds__(256) void vb_merge_bundles(const float* __restrict__ xyz, const uint32_t* __restrict__ rgba,
                                                        const uint32_t* __restrict__ first, const uint32_t* __restrict__ pts,
                                                        const uint8_t* __restrict__ clr, int nb, float* __restrict__ mxyz,
                                                        uint32_t* __restrict__ mcol, float* __restrict__ mw) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  uint32_t colour = 0;   // Color()
  float m0 = 0.f, m1 = 0.f, m2 = 0.f, W = 0.f;
  const uint32_t end = clr[b] ? first[b] + 1u : first[b + 1];   // only the first point of a clearing bundle
  for (uint32_t j = first[b]; j < end; ++j) {
    const size_t p = pts[j];
    const float px = xyz[3 * p], py = xyz[3 * p + 1], pz = xyz[3 * p + 2];
    const float w = fabsf(pz) > 1e-6f ? 1.0f / (pz * pz) : 0.0f;   // getVoxelWeight
    const float tot = W + w;
    m0 = (m0 * W + px * w) / tot;
    m1 = (m1 * W + py * w) / tot;
    m2 = (m2 * W + pz * w) / tot;
    colour = blend_colours(colour, W, rgba[p], w);
    W += w;
  }
  mxyz[3 * (size_t)b] = m0;
  mxyz[3 * (size_t)b + 1] = m1;
  mxyz[3 * (size_t)b + 2] = m2;
  mcol[b] = colour;
  mw[b] = W;
}

constexpr int kExpandThreads = 1024;
template <int kMode>
__global__ __launch_bounds__(kExpandThreads) void vb_expand(
    Params P, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ seqs, uint32_t n,
    const float* __restrict__ xyz, const float* __restrict__ aux, const uint32_t* __restrict__ rgba,
    const int32_t* __restrict__ offsets, int nclouds, const PoseRt* __restrict__ Twc,
    const int32_t* __restrict__ slot_ids, float2* __restrict__ rec, uint32_t* __restrict__ rec_c,
    uint32_t* __restrict__ heads, uint32_t* __restrict__ updated_slots,
    VCounters* __restrict__ ctr) {
  __shared__ uint32_t wave_cnt[2][kExpandThreads / 64];
  __shared__ uint32_t block_base[2];
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  bool head = false, chead = false;
  uint32_t key = 0;
  if (r < n) {
    key = keys[r];
    const uint32_t prev = r ? keys[r - 1] : ~key;
    const uint32_t next = (r + 1 < n) ? keys[r + 1] : ~key;
    head = (r == 0) || (key != prev);
    chead = (r == 0) || ((key >> 12) != (prev >> 12));
    int cloud = 0;
    const int p = kMode != kSimple ? (int)seqs[r] : point_of_seq(offsets, nclouds, (int)seqs[r], &cloud);
    const PoseRt pose = load_pose(Twc, cloud);
    const float px = xyz[3 * (size_t)p], py = xyz[3 * (size_t)p + 1], pz = xyz[3 * (size_t)p + 2];
    const uint32_t slot = key >> 12, vid = key & 4095u;
    const int g[3] = {slot_ids[3 * slot + 0] * 16 + (int)(vid & 15u),
                      slot_ids[3 * slot + 1] * 16 + (int)((vid >> 4) & 15u),
                      slot_ids[3 * slot + 2] * 16 + (int)(vid >> 8)};
    float sdf, uw;
    if (kMode == kWorld) {   // updateTsdfVoxel(ray_start, point_G, ..., weight 1), tsdf_integrator.cc:78
      Ray ray;
      float rs[3];
      make_ray_world(P, pose, px, py, pz, aux[3 * (size_t)p], aux[3 * (size_t)p + 1], aux[3 * (size_t)p + 2], &ray, rs);
      visit_operands(P, rs, ray.pG, g, 1.0f, &sdf, &uw);
    } else if (kMode == kMerged) {   // updateTsdfVoxel(origin, merged_point_G, ..., merged_color, merged_weight), :443
      float pG[3];
      quat_transform(pose, px, py, pz, pG);
      visit_operands(P, pose.t, pG, g, aux[p], &sdf, &uw);
    } else {
      float pG[3];
      quat_transform(pose, px, py, pz, pG);
      const float weight = fabsf(pz) > 1e-6f ? 1.0f / (pz * pz) : 0.0f;
      visit_operands(P, pose.t, pG, g, weight, &sdf, &uw);
    }
    // uw >= 0: its sign bit marks the LAST record of the voxel run
    rec[r] = make_float2(sdf, (key != next) ? -uw : uw);
    rec_c[r] = rgba[p];
  }
  const unsigned long long mh = __ballot(head), mc = __ballot(chead);
  if (lane == 0) {
    wave_cnt[0][wid] = (uint32_t)__popcll(mh);
    wave_cnt[1][wid] = (uint32_t)__popcll(mc);
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    uint32_t tot = 0;
    for (int w = 0; w < kExpandThreads / 64; ++w) {
      const uint32_t c = wave_cnt[threadIdx.x][w];
      wave_cnt[threadIdx.x][w] = tot;
      tot += c;
    }
    block_base[threadIdx.x] =
        tot ? atomicAdd(threadIdx.x == 0 ? &ctr->num_heads : &ctr->num_updated, tot) : 0u;
  }
  __syncthreads();
  if (head) heads[block_base[0] + wave_cnt[0][wid] + (uint32_t)__popcll(mh & lt)] = r;
  if (chead) updated_slots[block_base[1] + wave_cnt[1][wid] + (uint32_t)__popcll(mc & lt)] = key >> 12;
}

// The order-dependent fold of updateTsdfVoxel over the sorted records, record-centric: a workgroup takes kChainChunk
// consecutive records into LDS with coalesced loads and folds the voxel runs that START inside its chunk (a run that
// runs past the chunk's end reads on from global memory; one that started before belongs to the workgroup before).
// (The first version gave a thread one voxel run and read it at a stride of the run lengths: 12 B per visit in
// scattered pieces, 0.34 ms for the 11 M visits of a step.)
//
// Two ways to fold a run, by its length (round 3):
//  * SHORT runs (< kLongRun visits, wholly inside the chunk): a lane per run, voxel_fold visit by visit; a lane that
//    finishes takes the chunk's next short run at once.  In such a wave some lane starts a run in nearly every trip
//    of the loop, so a visit costs the wave the run-start path too (two global round trips): ≈ 0.6 us.  Fine for a few
//    visits — but a voxel that every key frame of the call sees has hundreds, and the kernel used to end with such
//    lanes (0.36 ms for 11 M visits).
//  * LONG runs (and the run that leaves the chunk): EIGHT lanes per run, eight visits per trip.  What makes a visit
//    expensive — three IEEE divisions, the byte <-> float conversions of four colour channels — does not depend on
//    the running distance or colour: the weight sequence W_k = min(W_{k-1} + w_k, max) only needs the records, and
//    with it the divisor (W_{k-1} + w_k), both blend factors and the product sdf*w of EVERY visit are known up front.
//    So per trip: the eight lanes take eight records; the weight chain runs through the eight visits (plain running
//    sums — the 1e-6 floor and the max_weight ceiling are checked afterwards and a trip they act on is redone);
//    lane g computes visit g's operands — one correctly rounded reciprocal and three exact quotients from it — into
//    LDS; then the order-dependent part runs with lane 0 carrying the distance (mul, add, the exact-quotient step
//    mul, fma, fma — dist_update_rcp's form — and a median for the clamp) and lanes 1-4 a colour channel each
//    (mul, add, round).  A trip of eight visits takes ≈ 1.3 us; nothing in it waits for global memory (the next
//    trip's records are fetched before the chains start, the voxels of all long runs are staged in LDS up front).
//    The groups take the chunk's long runs longest first, each the next one as soon as its own ends.
// Measured (MI355X, 25 key frames per call, 11.2 M visits): 0.36 -> 0.20 ms, the call 1.07 -> 0.92 ms.  The
// kernel is now bound by each chunk's longest run (a workgroup lives as long as it: ≈ 25 us on average, two to three
// times the 8 trips a group averages) at the four workgroups per CU its 40.8 KB of LDS allow (three at 45 KB: + 5 %).  Did not help: four
// lanes per run (slower: twice the trips on the critical run), 128- and 64-thread workgroups, a quarter fewer
// instructions per trip, a lane-path threshold anywhere from 8 to 128.
#ifndef PLVS_VB_LONG_RUN
#define PLVS_VB_LONG_RUN 16
#endif
constexpr int kChainChunk = 2048;
constexpr int kChainThreads = 256;
constexpr int kLongRun = PLVS_VB_LONG_RUN;
constexpr int kG = 8;   // lanes per long run = visits per trip
constexpr int kChainGroups = kChainThreads / kG;
constexpr int kMaxLong = kChainChunk / kLongRun + 2;

#ifndef PLVS_VB_PROF
#define PLVS_VB_PROF 0
#endif
#if PLVS_VB_PROF   // developer build: the times (100 MHz ticks) at which every wave passes its stages, read by plvs_hip_debug_chain_prof
constexpr int kProfWaves = 1 << 16;
__device__ unsigned long long g_chain_prof[kProfWaves][4];
#define CHAIN_PROBE(i)                                                                                         \
  if (lane == 0 && blockIdx.x * (kChainThreads / 64) + wid < kProfWaves)                                       \
    g_chain_prof[blockIdx.x * (kChainThreads / 64) + wid][i] = wall_clock64();
#else
#define CHAIN_PROBE(i)
#endif

// RN(1/b) for b in [2^-20, 2^40] (tsdf_chisel.hip's rcp_rn: checked for every significand by plvs_hip_selftest_rcp) and
// RN(a/b) from it (dist_update_rcp's correction step), exact for a = 0 or |a| in [2^-60, 2^60]
__device__ __forceinline__ float vb_rcp_rn(float b) {
  const float y0 = __builtin_amdgcn_rcpf(b);
  const float e = fmaf(-b, y0, 1.0f);
  return fmaf(e, y0, y0);
}
__device__ __forceinline__ float vb_quot(float a, float b, float y) {
  const float q = a * y;
  const float r = fmaf(-q, b, a);
  return fmaf(r, y, q);
}
__device__ __forceinline__ bool vb_quot_ok(float a) { return a == 0.0f || (fabsf(a) >= 0x1p-60f && fabsf(a) <= 0x1p60f); }

// record rr of the chunk (LDS), of the records behind it (global) or a terminator beyond the call's last record
__device__ __forceinline__ void chain_load(const float2* s_rec, const uint32_t* s_col, const float2* __restrict__ rec,
                                           const uint32_t* __restrict__ rec_c, uint32_t c0, uint32_t n, uint32_t nrec,
                                           uint32_t rr, float2* v, uint32_t* col) {
  const uint32_t rl = min(rr, n - 1u);
  *v = s_rec[rl];
  *col = s_col[rl];
  if (rr >= n) {   // (only the run that leaves the chunk gets here)
    if (c0 + rr < nrec) {
      *v = rec[c0 + rr];
      *col = rec_c[c0 + rr];
    } else {
      *v = make_float2(0.f, -0.0f);
      *col = 0u;
    }
  }
}

__global__ __launch_bounds__(kChainThreads) void vb_chain_chunks(
    Params P, const uint32_t* __restrict__ keys, uint32_t nrec, const float2* __restrict__ rec,
    const uint32_t* __restrict__ rec_c, VCounters* __restrict__ ctr, float* __restrict__ dist,
    float* __restrict__ weight, uint32_t* __restrict__ rgba) {
  __shared__ float2 s_rec[kChainChunk];
  __shared__ uint32_t s_col[kChainChunk];
  __shared__ uint16_t s_head[kChainChunk + 2];   // positions of the run heads of the chunk, ascending; then the chunk's end
  __shared__ uint16_t s_long[kMaxLong];          // the long ones
  __shared__ unsigned long long s_mask[kChainChunk / 64];
  __shared__ uint32_t s_pre[kChainChunk / 64];
  // the per-visit operands of a trip, the distance lane's and the colour lanes' apart
  // (40.8 KB in all: four workgroups per CU.  The short runs' list lies over the distance operands — it is dead before the
  // long loop's first trip, a barrier between — and the operand rows are unpadded: the bank conflicts of the groups'
  // broadcast reads cost 1 %, the fourth workgroup gains 5 % of the call.  Chunks of 1536 / 1024 records: slower.)
  __shared__ float4 s_opd[kChainGroups][kG], s_opc[kChainGroups][kG];
  static_assert(sizeof(s_opd) >= kChainChunk * sizeof(uint16_t), "the short runs' list lies over the distance operands");
  uint16_t* const s_short = reinterpret_cast<uint16_t*>(&s_opd[0][0]);   // the short runs (indices into s_head), any order
  __shared__ __attribute__((aligned(16))) float s_uw[kChainGroups][kG];
  __shared__ float4 s_state[kMaxLong];           // a long run's voxel: distance, weight, colour, its index in the pool
  __shared__ uint32_t s_cls[32];
  __shared__ uint32_t s_nheads, s_nshort, s_nlong, s_next, s_next_long;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const uint32_t c0 = blockIdx.x * (uint32_t)kChainChunk;
  if (c0 >= nrec) return;
  const uint32_t n = min((uint32_t)kChainChunk, nrec - c0);
  CHAIN_PROBE(0)
  if (tid < 32) s_cls[tid] = 0;
  if (tid == 0) {
    s_nshort = 0;
    s_nlong = 0;
    s_next = (uint32_t)kChainThreads;   // the next short run to hand out; the first ones go by thread index
    s_next_long = (uint32_t)kChainGroups;   // the same for long runs and groups
  }
  // ---- the chunk into LDS; its run heads in record order.  All loads of the thread are issued before the first is
  // used (clamped addresses instead of branches): one memory latency per chunk, not one per round.
  constexpr int kRounds = kChainChunk / kChainThreads;
  float2 l_rec[kRounds];
  uint32_t l_col[kRounds], l_key[kRounds], l_prev[kRounds];
#pragma unroll
  for (int k = 0; k < kRounds; ++k) {
    const uint32_t at = c0 + min((uint32_t)(k * kChainThreads + tid), n - 1u);
    l_rec[k] = rec[at];
    l_col[k] = rec_c[at];
    l_key[k] = keys[at];
    l_prev[k] = keys[max(at, 1u) - 1u];
  }
  uint32_t myheads = 0;
#pragma unroll
  for (int k = 0; k < kRounds; ++k) {
    const uint32_t r = (uint32_t)(k * kChainThreads + tid);
    bool head = false;
    if (r < n) {
      s_rec[r] = l_rec[k];
      s_col[r] = l_col[k];
      head = (c0 + r == 0u) || l_prev[k] != l_key[k];
    }
    const unsigned long long m = __ballot(head);
    if (lane == 0) s_mask[k * (kChainThreads / 64) + wid] = m;
    myheads |= head ? (1u << k) : 0u;
  }
  __syncthreads();
  if (tid < kChainChunk / 64) {   // (32 words: half of wave 0)
    const uint32_t c = (uint32_t)__popcll(s_mask[tid]);
    uint32_t incl = c;
#pragma unroll
    for (int off = 1; off < kChainChunk / 64; off <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
      if (tid >= off) incl += up;
    }
    s_pre[tid] = incl - c;
    if (tid == kChainChunk / 64 - 1) s_nheads = incl;
  }
  __syncthreads();
  const uint32_t nheads = s_nheads;
#pragma unroll
  for (int k = 0; k < kChainChunk / kChainThreads; ++k) {
    if ((myheads >> k) & 1u) {
      const int w = k * (kChainThreads / 64) + wid;
      s_head[s_pre[w] + (uint32_t)__popcll(s_mask[w] & lt)] = (uint16_t)(k * kChainThreads + tid);
    }
  }
  if (tid == 0) s_head[nheads] = (uint16_t)n;
  __syncthreads();
  // ---- short and long runs.  The chunk's last run is long when it goes on behind the chunk.  The long ones are put
  // in classes of descending length (a counting sort over trips of eight visits; the run that leaves the chunk first):
  // the groups take them in that order, so a workgroup does not end with one group on a long run it started last.
  const bool crossing = !(__float_as_uint(s_rec[n - 1u].y) >> 31);
  auto run_class = [&](uint32_t h, bool* is_long) -> uint32_t {
    const uint32_t len = (uint32_t)s_head[h + 1u] - (uint32_t)s_head[h];
    const bool leaves = crossing && h + 1u == nheads;
    *is_long = len >= (uint32_t)kLongRun || leaves;
    return leaves ? 0u : 31u - min((len + 7u) >> 3, 31u);
  };
  for (uint32_t base = 0; base < nheads; base += (uint32_t)kChainThreads) {
    const uint32_t h = base + (uint32_t)tid;
    bool is_long = false, is_short = false;
    if (h < nheads) {
      const uint32_t cls = run_class(h, &is_long);
      is_short = !is_long;
      if (is_long) atomicAdd(&s_cls[cls], 1u);
    }
    const unsigned long long ms = __ballot(is_short);
    uint32_t bs = 0;
    if (lane == 0 && ms) bs = atomicAdd(&s_nshort, (uint32_t)__popcll(ms));
    bs = (uint32_t)__shfl((int)bs, 0);
    if (is_short) s_short[bs + (uint32_t)__popcll(ms & lt)] = (uint16_t)h;
  }
  __syncthreads();
  if (tid < 32) {
    const uint32_t c = s_cls[tid];
    uint32_t incl = c;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
      if (tid >= off) incl += up;
    }
    s_cls[tid] = incl - c;   // (from here on: where the class's next run goes)
    if (tid == 31) s_nlong = incl;
  }
  __syncthreads();
  for (uint32_t base = 0; base < nheads; base += (uint32_t)kChainThreads) {
    const uint32_t h = base + (uint32_t)tid;
    if (h < nheads) {
      bool is_long;
      const uint32_t cls = run_class(h, &is_long);
      if (is_long) s_long[atomicAdd(&s_cls[cls], 1u)] = (uint16_t)h;
    }
  }
  __syncthreads();
  const uint32_t nshort = s_nshort, nlong = s_nlong;
  uint32_t longest = 0;
  // the voxels of the long runs into LDS, all loads side by side (a group that starts a run inside the trip loop below
  // must not make its wave wait for global memory)
  for (uint32_t j = (uint32_t)tid; j < nlong; j += (uint32_t)kChainThreads) {
    const uint32_t a = keys[c0 + s_head[s_long[j]]];
    s_state[j] = make_float4(dist[a], weight[a], __uint_as_float(rgba[a]), __uint_as_float(a));
  }
  CHAIN_PROBE(1)
  // ---- short runs: a lane per run
  {
    uint32_t i = (uint32_t)tid, r = 0;
    size_t a = 0;
    float D = 0.f, W = 0.f;
    uint32_t C = 0;
    bool have = false;
    for (;;) {
      if (!have) {
        if (i >= nshort) break;
        const uint32_t h = s_short[i];
        r = s_head[h];
        longest = max(longest, (uint32_t)s_head[h + 1u] - r);
        a = (size_t)keys[c0 + r];
        D = dist[a];
        W = weight[a];
        C = rgba[a];
        have = true;
      }
      const float2 v = s_rec[r];
      voxel_fold(P, D, W, C, v.x, fabsf(v.y), s_col[r]);
      if (__float_as_uint(v.y) >> 31) {   // (its sign bit marks the last record of the run)
        dist[a] = D;
        weight[a] = W;
        rgba[a] = C;
        have = false;
        i = atomicAdd(&s_next, 1u);
      } else {
        ++r;
      }
    }
  }
  CHAIN_PROBE(2)
  // ---- long runs: kG lanes per run, kG visits per trip.  Lane 0 of the group carries the distance, lanes 1-4 a colour
  // channel each.  ONE loop: a group whose run ends takes the
  // chunk's next long run in the same trip, so the groups of a wave never wait for each other's runs.
  __syncthreads();   // (s_state complete)
  const int grp = tid / kG, g = tid % kG;
  const int sh = 8 * ((g - 1) & 3);   // lanes 1-4 (and, idle, 0 and 5-7): the colour channel
  {
    // (the first runs — the longest — dealt round the waves, not eight in a row to each: a workgroup's waves sit on
    // different SIMDs)
    constexpr int kGroupsPerWave = 64 / kG, kWaves = kChainThreads / 64;
    uint32_t j = (uint32_t)((grp % kGroupsPerWave) * kWaves + grp / kGroupsPerWave), r = 0, visits = 0, col = 0;
    float W = 0.f, X = 0.f, Dx = 0.f;
    float2 v = make_float2(0.f, 0.f);
    bool have = false;
    for (;;) {
      if (!have) {
        if (j >= nlong) break;
        const float4 st = s_state[j];
        Dx = st.x;                                              // (lane 0's)
        W = st.y;
        X = (float)((__float_as_uint(st.z) >> sh) & 255u);      // the lane's colour channel
        r = s_head[s_long[j]];
        visits = 0;
        chain_load(s_rec, s_col, rec, rec_c, c0, n, nrec, r + (uint32_t)g, &v, &col);
        have = true;
      }
      // the visits of this trip: up to the run's last record
      const unsigned long long bal = __ballot(__float_as_uint(v.y) >> 31);
      const uint32_t gm = (uint32_t)(bal >> (lane & ~(kG - 1))) & ((1u << kG) - 1u);
      const int nvalid = gm ? __ffs((int)gm) : kG;
      const bool done = gm != 0u;
      const float sdf = v.x, uw = fabsf(v.y);
      s_uw[grp][g] = uw;
      float2 vn = make_float2(0.f, 0.f);
      uint32_t coln = 0;
      if (!done) chain_load(s_rec, s_col, rec, rec_c, c0, n, nrec, r + (uint32_t)(kG + g), &vn, &coln);
      __builtin_amdgcn_wave_barrier();
      // the weight chain: lane g takes the steps of the visits before its own
      float u[kG];
#pragma unroll
      for (int k = 0; k < kG; k += 4)
        *reinterpret_cast<float4*>(&u[k]) = *reinterpret_cast<const float4*>(&s_uw[grp][k]);
      // (plain running sums first: the 1e-6 floor and the max_weight ceiling of updateTsdfVoxel almost never act, and a
      // lane whose own sum is clean knows that the sums before it were)
      float w_prev = W;
#pragma unroll
      for (int k = 0; k < kG - 1; ++k) {
        const float nwk = w_prev + u[k];
        w_prev = (k < g) ? nwk : w_prev;
      }
      float nw = w_prev + uw;
      {
        const unsigned long long balw = __ballot((g < nvalid) && !((nw >= 1e-6f) && (nw < P.max_weight)));
        if ((((uint32_t)(balw >> (lane & ~(kG - 1)))) & ((1u << kG) - 1u)) != 0u) {
          w_prev = W;
          for (int k = 0; k < kG - 1; ++k) {
            const float nwk = w_prev + u[k];
            const float stepped = (nwk < 1e-6f) ? w_prev : ((nwk < P.max_weight) ? nwk : P.max_weight);
            w_prev = (k < g) ? stepped : w_prev;
          }
          nw = w_prev + uw;
        }
      }
      const bool skip = (g >= nvalid) || (nw < 1e-6f);
      const float w_after = skip ? w_prev : ((nw < P.max_weight) ? nw : P.max_weight);
      W = __shfl(w_after, nvalid - 1, kG);   // (the weight after the trip's last visit)
      // visit g's operands (blend_colours' total = w1 + w2 is nw): the distance half, the colour half
      const bool rcp_ok = (nw >= 0x1p-20f) && (nw <= 0x1p40f);
      const unsigned long long balr = __ballot(!skip && !rcp_ok);
      bool inexact = (((uint32_t)(balr >> (lane & ~(kG - 1)))) & ((1u << kG) - 1u)) != 0u;
      {
        const bool blend = !skip && (fabsf(sdf) < P.truncation);
        // 1 / nw correctly rounded (v_rcp_f32 and one Newton step: plvs_hip_selftest_rcp), and both blend factors as exact
        // quotients from it (the same correction step as the distance's); operands outside the exact ranges divide
        float y = vb_rcp_rn(nw), w1n = vb_quot(w_prev, nw, y), w2n = vb_quot(uw, nw, y);
        if (!skip && !(rcp_ok && vb_quot_ok(w_prev) && vb_quot_ok(uw))) {   // (a skipped visit's operands are not used)
          y = 1.0f / nw;
          w1n = w_prev / nw;
          w2n = uw / nw;
        }
        s_opd[grp][g] = make_float4(skip ? -w_prev : w_prev, sdf * uw, y, nw);
        s_opc[grp][g] = make_float4(w1n, w2n, __uint_as_float(col), blend ? 1.0f : 0.0f);
      }
      __builtin_amdgcn_wave_barrier();
      float4 opc[kG];
#pragma unroll
      for (int k = 0; k < kG; ++k) opc[k] = s_opc[grp][k];
      // the order-dependent part.  A colour step: round(a*w1 + b*w2) of non-negative operands with w1 + w2 = 1 up to
      // roundings is an integer in [0, 255] — blend_colours' cast to a byte and back changes nothing.
      auto colour_steps = [&](float x) {
#pragma unroll
        for (int k = 0; k < kG; ++k) {
          const float b = (float)((__float_as_uint(opc[k].z) >> sh) & 255u);
          const float t = x * opc[k].x + b * opc[k].y;
          const float tr = truncf(t);
          const float nc = tr + (((t - tr) >= 0.5f) ? 1.0f : 0.0f);   // roundf of t >= 0
          x = (opc[k].w != 0.0f) ? nc : x;
        }
        return x;
      };
      if (g == 0) {
        float4 opd[kG];
#pragma unroll
        for (int k = 0; k < kG; ++k) opd[k] = s_opd[grp][k];
        // (the quotient from the reciprocal is exact inside dist_update_rcp_exact's operand ranges; a trip that leaves
        // them — none does on real data — is redone with the division itself)
        const float x0 = Dx;
#pragma unroll
        for (int k = 0; k < kG; ++k) {
          const bool skipk = __float_as_uint(opd[k].x) >> 31;
          const float t = opd[k].y + Dx * fabsf(opd[k].x);   // sdf * w + D * W
          const float y = opd[k].z, nwk = opd[k].w;
          const float q = t * y;
          const float rem = fmaf(-q, nwk, t);
          const float nd = fmaf(rem, y, q);
          const float at = fabsf(t);
          inexact |= !skipk && !(at >= 0x1p-60f && at <= 0x1p60f);
          // (an exact quotient of in-range operands is finite: the median IS voxel_fold's pair of std::min / std::max)
          Dx = skipk ? Dx : __builtin_amdgcn_fmed3f(nd, -P.truncation, P.truncation);
        }
        if (inexact) {
          Dx = x0;
          for (int k = 0; k < kG; ++k) {
            const float t = opd[k].y + Dx * fabsf(opd[k].x);
            float nd = t / opd[k].w;
            nd = (nd > 0.0f) ? ((nd < P.truncation) ? nd : P.truncation) : ((-P.truncation < nd) ? nd : -P.truncation);
            Dx = (__float_as_uint(opd[k].x) >> 31) ? Dx : nd;
          }
        }
      } else {
        X = colour_steps(X);
      }
      __builtin_amdgcn_wave_barrier();
      visits += (uint32_t)nvalid;
      if (done) {
        const uint32_t cx = (uint32_t)X;
        const uint32_t C = (uint32_t)__shfl((int)cx, 1, kG) | ((uint32_t)__shfl((int)cx, 2, kG) << 8) |
                           ((uint32_t)__shfl((int)cx, 3, kG) << 16) | ((uint32_t)__shfl((int)cx, 4, kG) << 24);
        uint32_t jn = 0;
        if (g == 0) {
          s_state[j] = make_float4(Dx, W, __uint_as_float(C), s_state[j].w);
          jn = atomicAdd(&s_next_long, 1u);
        }
        j = (uint32_t)__shfl((int)jn, 0, kG);
        longest = max(longest, visits);
        have = false;
      } else {
        v = vn;
        col = coln;
        r += (uint32_t)kG;
      }
    }
  }
  __syncthreads();
  for (uint32_t j = (uint32_t)tid; j < nlong; j += (uint32_t)kChainThreads) {
    const float4 st = s_state[j];
    const size_t a = (size_t)__float_as_uint(st.w);
    dist[a] = st.x;
    weight[a] = st.y;
    rgba[a] = __float_as_uint(st.z);
  }
  CHAIN_PROBE(3)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) longest = max(longest, (uint32_t)__shfl_xor((int)longest, off));
  if (lane == 0 && longest > ctr->max_run) atomicMax(&ctr->max_run, longest);
}

// The updated-block list without the blocks that have not joined the layer yet (slots >= visible), order kept; one
// workgroup (the list has a few thousand entries).  *n_out = the new length.
__global__ __launch_bounds__(1024) void vb_filter_slots(uint32_t* __restrict__ slots, uint32_t n, uint32_t visible,
                                                        uint32_t* __restrict__ n_out) {
  __shared__ uint32_t s_cnt[16];
  __shared__ uint32_t s_base;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) s_base = 0u;
  __syncthreads();
  for (uint32_t i0 = 0; i0 < n; i0 += 1024u) {
    const uint32_t i = i0 + (uint32_t)tid;
    const uint32_t v = i < n ? slots[i] : 0u;
    const bool keep = i < n && v < visible;
    const unsigned long long m = __ballot(keep);
    if (lane == 0) s_cnt[wid] = (uint32_t)__popcll(m);
    __syncthreads();   // (every slot of this round has been read: compaction only moves entries towards the front)
    uint32_t at = s_base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wid; ++w) at += s_cnt[w];
    if (keep) slots[at] = v;
    __syncthreads();
    if (tid == 0) {
      uint32_t t = 0;
      for (int w = 0; w < 16; ++w) t += s_cnt[w];
      s_base += t;
    }
    __syncthreads();
  }
  if (tid == 0) *n_out = s_base;
}

__global__ void vb_gather_slot_ids(const uint32_t* __restrict__ slots, int n,
                                   const int32_t* __restrict__ slot_ids, int32_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint32_t s = slots[i];
    out[3 * i] = slot_ids[3 * s];
    out[3 * i + 1] = slot_ids[3 * s + 1];
    out[3 * i + 2] = slot_ids[3 * s + 2];
  }
}

}  // namespace

struct plvs_tsdf_voxblox {
  plvs_tsdf_voxblox_params prm;
  Params P;
  Directory dir;
  float* dist = nullptr;
  float* weight = nullptr;
  uint32_t* rgba = nullptr;
  VCounters* d_ctr = nullptr;
  VCounters* h_ctr = nullptr;
  int num_blocks = 0;
  int visible_blocks = 0;              // blocks the layer shows: all, unless world-cloud blocks wait for the next camera-ray call
  bool defer_world_blocks = false;
  bool poisoned = false;
  DevBuf<uint32_t> counts, keys0, keys1, seq0, seq1, heads, updated, scratch, rec_c;
  DevBuf<float2> rec;
  DevBuf<uint32_t> upd_merge;          // (the updated list when waiting world-cloud blocks join it)
  DevBuf<int32_t> offsets;
  DevBuf<float> st_xyz, st_Twc, st_nrm;
  DevBuf<uint32_t> st_rgba;
  DevBuf<PoseRt> poses;
  // merged integrator: per-point kinds / end voxels, the bundles (CSR) and their merged points
  DevBuf<uint8_t> mg_kind, mg_clr;
  DevBuf<int32_t> mg_g;
  DevBuf<uint32_t> mg_first, mg_pts, mg_col;
  DevBuf<float> mg_xyz, mg_w;
  // fast integrator (tsdf_voxblox_fast.hpp): the two approximate sets as the reference keeps them, the offset of the next
  // scan, and the scratch of the rounds
  DevBuf<unsigned long long> ap_start, ap_seen, ff_shash, ff_qhash;
  DevBuf<uint32_t> ff_skey0, ff_skey1, ff_sval0, ff_sval1, ff_full, ff_Q, ff_L, ff_qoff, ff_qkey0, ff_qkey1, ff_qval0, ff_qval1, ff_flags;
  DevBuf<uint8_t> ff_seen;
  uint32_t* h_ff = nullptr;            // pinned: {queries of the next round, did a ray change}
  uint32_t ap_next = 1;                // (the reference's sets start at offset 0 and every scan begins with offset + 1)
  bool ap_ready = false;
  int fast_rounds = 0;                 // rounds of the last fast call (diagnostic)
  bool fast_sequential = false;        //   ... and whether it was finished on one thread (vbf_sequential)
  // queued key-frame clouds (plvs_hip_tsdf_voxblox_queue / _flush): uploaded, not yet integrated
  DevBuf<float> q_xyz, q_Twc_dev;   // (the poses get a buffer of their own: a host-flavour integrate that finds a queue has
                                    //  staged ITS pose in st_Twc already)
  DevBuf<uint32_t> q_rgba;
  std::vector<int32_t> q_offsets;   // [clouds + 1] once anything is queued
  std::vector<float> q_Twc;         // 12 per cloud
  plvs_tsdf_stats stats{};
  uint32_t last_updated = 0;
  void* ext = nullptr;                 // meshing scratch (tsdf_voxblox_mesh.hip), freed with the map
  void (*ext_free)(void*) = nullptr;
  // halo of a sharded map (meshing): ghost copies of other ranks' blocks in the pool slots past num_blocks
  Directory gdir{};
  int ghost_count = 0;
  DevBuf<uint32_t> halo_row;
  // ray-sharded integrate (tsdf_voxblox_shard.hpp): this rank's visit records in sequence order, their destinations and the
  // stable partition by destination; the owner's translated keys
  DevBuf<uint4> sv_rec;
  DevBuf<uint32_t> sv_dest, sv_dest1, sv_idx, sv_idx1, sv_cnt, sv_vkey, sv_seq;
  uint32_t* h_sv_cnt = nullptr;        // pinned: records per destination
  uint32_t sv_V = 0;
  int sv_phase = 0;                    // 0 idle, 1 walked, 2 packed
  bool sv_partitioned = false;
};

template <typename T>
static hipError_t vb_grow_keep(DevBuf<T>& b, size_t used, size_t want) {   // reserve() that keeps the first `used` elements
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
extern "C" int plvs_hip_tsdf_voxblox_flush(plvs_tsdf_voxblox* h);
#define VB_FLUSH_QUEUE(h)                                          \
  do {                                                             \
    if ((h) && !(h)->q_offsets.empty()) {                          \
      const int rc_flush_ = plvs_hip_tsdf_voxblox_flush(h);        \
      if (rc_flush_ != PLVS_OK) return rc_flush_;                  \
    }                                                              \
  } while (0)

namespace plvs {
namespace vbx {

bool voxblox_map_view(plvs_tsdf_voxblox* h, VoxbloxMapView* v) {
  if (h == nullptr || v == nullptr || h->poisoned) return false;
  if (!h->q_offsets.empty() && plvs_hip_tsdf_voxblox_flush(h) != PLVS_OK) return false;   // (the meshers read the map)
  v->voxel_size = h->P.voxel_size;
  v->voxel_size_inv = h->P.voxel_size_inv;
  v->dir = h->dir;
  v->distance = h->dist;
  v->weight = h->weight;
  v->rgba = h->rgba;
  v->num_blocks = h->num_blocks;
  v->visible_blocks = h->visible_blocks;
  v->shard_count = h->P.shard_count;
  v->ghost = h->gdir;
  v->ext = &h->ext;
  v->ext_free = &h->ext_free;
  return true;
}

}  // namespace vbx
}  // namespace plvs

static int vb_read_counters(plvs_tsdf_voxblox* h, hipStream_t s) {
  // (published by a kernel's stores into the pinned copy: a small copy command costs tens of microseconds of queueing)
  hipLaunchKernelGGL(vb_publish_counters, dim3(1), dim3(64), 0, s, h->d_ctr, h->h_ctr);
  PLVS_KERNEL_CHECK();
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  return PLVS_OK;
}

extern "C" {

int plvs_hip_tsdf_voxblox_default_params(float voxel_size, int use_carving,
                                         plvs_tsdf_voxblox_params* p) {
  PLVS_REQUIRE(p, "params is null");
  PLVS_REQUIRE(voxel_size > 0.0f, "voxel_size must be positive");
  p->voxel_size = voxel_size;        // tsdf_voxel_size = PointCloudMapping.resolution
  p->truncation = 0.1f;              // src/PointCloudMapVoxblox.cc:57
  p->max_weight = 10000.0f;          // :58
  p->min_ray_length = 0.1f;          // :60
  p->max_ray_length = 5.0f;          // :61
  p->voxel_carving = use_carving ? 1 : 0;  // :59
  p->max_blocks = 32768;             // 1.5 GiB of voxel pool
  p->shard_rank = 0;
  p->shard_count = 1;
  return PLVS_OK;
}

int plvs_hip_tsdf_voxblox_destroy(plvs_tsdf_voxblox* h) {
  if (!h) return PLVS_OK;
  if (h->ext && h->ext_free) h->ext_free(h->ext);
  (void)hipFree(h->gdir.keys); (void)hipFree(h->gdir.slots);
  h->halo_row.release();
  (void)hipFree(h->dir.keys); (void)hipFree(h->dir.slots); (void)hipFree(h->dir.slot_ids);
  (void)hipFree(h->dist); (void)hipFree(h->weight); (void)hipFree(h->rgba); (void)hipFree(h->d_ctr);
  if (h->h_ctr) (void)hipHostFree(h->h_ctr);
  if (h->h_ff) (void)hipHostFree(h->h_ff);
  if (h->h_sv_cnt) (void)hipHostFree(h->h_sv_cnt);
  h->sv_rec.release(); h->sv_dest.release(); h->sv_dest1.release(); h->sv_idx.release(); h->sv_idx1.release(); h->sv_cnt.release();
  h->sv_vkey.release(); h->sv_seq.release();
  h->q_xyz.release(); h->q_rgba.release(); h->q_Twc_dev.release();
  h->ap_start.release(); h->ap_seen.release(); h->ff_shash.release(); h->ff_qhash.release(); h->ff_skey0.release(); h->ff_skey1.release();
  h->ff_sval0.release(); h->ff_sval1.release(); h->ff_full.release(); h->ff_Q.release(); h->ff_L.release(); h->ff_qoff.release();
  h->ff_qkey0.release(); h->ff_qkey1.release(); h->ff_qval0.release(); h->ff_qval1.release(); h->ff_flags.release(); h->ff_seen.release();
  h->counts.release(); h->keys0.release(); h->keys1.release(); h->seq0.release(); h->seq1.release();
  h->heads.release(); h->updated.release(); h->scratch.release(); h->rec_c.release(); h->rec.release(); h->upd_merge.release();
  h->offsets.release(); h->st_xyz.release(); h->st_Twc.release(); h->st_nrm.release(); h->st_rgba.release();
  h->mg_kind.release(); h->mg_clr.release(); h->mg_g.release(); h->mg_first.release(); h->mg_pts.release(); h->mg_col.release();
  h->mg_xyz.release(); h->mg_w.release(); h->poses.release();
  delete h;
  return PLVS_OK;
}

int plvs_hip_tsdf_voxblox_clear(plvs_tsdf_voxblox* h) {
  PLVS_REQUIRE(h, "null handle");
  if (h->gdir.keys != nullptr) PLVS_HIP_TRY(hipMemset(h->gdir.keys, 0xFF, ((size_t)h->gdir.mask + 1) * sizeof(unsigned long long)));
  h->ghost_count = 0;
  const size_t cap = (size_t)h->dir.mask + 1;
  const size_t nvox = (size_t)h->prm.max_blocks * kBlockVox;
  PLVS_HIP_TRY(hipMemset(h->dir.keys, 0xFF, cap * sizeof(unsigned long long)));
  PLVS_HIP_TRY(hipMemset(h->dir.slots, 0xFF, cap * sizeof(int32_t)));
  PLVS_HIP_TRY(hipMemset(h->dist, 0, nvox * sizeof(float)));     // TsdfVoxel defaults: 0, 0, Color()
  PLVS_HIP_TRY(hipMemset(h->weight, 0, nvox * sizeof(float)));
  PLVS_HIP_TRY(hipMemset(h->rgba, 0, nvox * sizeof(uint32_t)));
  PLVS_HIP_TRY(hipMemset(h->d_ctr, 0, sizeof(VCounters)));
  PLVS_HIP_TRY(hipDeviceSynchronize());
  h->num_blocks = 0;
  h->visible_blocks = 0;
  h->poisoned = false;
  h->stats = plvs_tsdf_stats{};
  h->last_updated = 0;
  h->ap_ready = false;   // (a new map = a new integrator: fresh sets, offset 0)
  h->ap_next = 1;
  h->q_offsets.clear();  // (queued clouds belong to the map that is gone)
  h->q_Twc.clear();
  return PLVS_OK;
}

int plvs_hip_tsdf_voxblox_create(const plvs_tsdf_voxblox_params* p, plvs_tsdf_voxblox** out) {
  PLVS_REQUIRE(p && out, "null argument");
  PLVS_REQUIRE(p->voxel_size > 0.0f, "voxel_size must be positive");
  PLVS_REQUIRE(p->max_blocks > 0 && p->max_blocks <= (1 << 20), "max_blocks must be in (0, 2^20]");
  PLVS_REQUIRE(p->shard_count <= 1 || (p->shard_rank >= 0 && p->shard_rank < p->shard_count),
               "shard_rank out of range");
  plvs_tsdf_voxblox* h = new plvs_tsdf_voxblox();
  h->prm = *p;
  Params& P = h->P;
  P.voxel_size = p->voxel_size;
  P.voxel_size_inv = (float)(1.0 / p->voxel_size);  // tsdf_integrator.cc:17
  P.vps_inv = (float)(1.0 / 16);                    // :19
  P.truncation = p->truncation;
  P.max_weight = p->max_weight;
  P.min_ray = p->min_ray_length;
  P.max_ray = p->max_ray_length;
  P.carving = p->voxel_carving ? 1 : 0;
  P.allow_clear = P.carving;  // allow_clear = true in PLVS, but it needs carving (:26-28)
  P.shard_rank = p->shard_rank;
  P.shard_count = p->shard_count < 1 ? 1 : p->shard_count;
  uint32_t cap = 1024;
  while (cap < 2u * (uint32_t)p->max_blocks) cap <<= 1;
  h->dir.mask = cap - 1;
  h->dir.max_blocks = p->max_blocks;
  const size_t nvox = (size_t)p->max_blocks * kBlockVox;
#define VB_TRY(call)                                                            \
  do {                                                                          \
    hipError_t _e = (call);                                                     \
    if (_e != hipSuccess) {                                                     \
      plvs::set_error("%s failed: %s", #call, hipGetErrorString(_e));          \
      plvs_hip_tsdf_voxblox_destroy(h);                                         \
      return PLVS_ERR_HIP;                                                      \
    }                                                                           \
  } while (0)
  VB_TRY(hipMalloc((void**)&h->dir.keys, (size_t)cap * sizeof(unsigned long long)));
  VB_TRY(hipMalloc((void**)&h->dir.slots, (size_t)cap * sizeof(int32_t)));
  VB_TRY(hipMalloc((void**)&h->dir.slot_ids, (size_t)p->max_blocks * 3 * sizeof(int32_t)));
  VB_TRY(hipMalloc((void**)&h->dist, nvox * sizeof(float)));
  VB_TRY(hipMalloc((void**)&h->weight, nvox * sizeof(float)));
  VB_TRY(hipMalloc((void**)&h->rgba, nvox * sizeof(uint32_t)));
  VB_TRY(hipMalloc((void**)&h->d_ctr, sizeof(VCounters)));
  VB_TRY(hipHostMalloc((void**)&h->h_ctr, sizeof(VCounters)));
#undef VB_TRY
  *out = h;
  int rc = plvs_hip_tsdf_voxblox_clear(h);
  if (rc != PLVS_OK) {
    plvs_hip_tsdf_voxblox_destroy(h);
    *out = nullptr;
  }
  return rc;
}

}  // extern "C"

namespace {
// One block id -> its pool slot, created if absent (plvs_hip_tsdf_voxblox_upload_block).
__global__ void vb_block_slot_of(Directory dir, int x, int y, int z, VCounters* __restrict__ ctr, int32_t* __restrict__ slot_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  dir_insert(dir, x, y, z, &ctr->num_blocks, &ctr->err);
  *slot_out = dir_find(dir, x, y, z);
}
}  // namespace

// ------------------------------------------------------------------ halo of a sharded map (meshing)
// A block's mesh reads its +x / +y / +z neighbour blocks (mesh_integrator.h:299-337), which block-hash sharding
// puts on other ranks: the caller asks their owners for them (the ids are known on the host: the seven neighbours of
// every block it meshes), the owners answer with found flags and one payload row per block that exists (three planes
// of 4096 words: distance, weight, rgba), the rows become ghost blocks past num_blocks until the next integrate call.
namespace {

constexpr int kVbHaloWords = 3 * kBlockVox;

__global__ void vb_halo_lookup(Directory dir, const int32_t* __restrict__ ids, int n, uint32_t* __restrict__ found) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) found[i] = dir_find(dir, ids[3 * i], ids[3 * i + 1], ids[3 * i + 2]) >= 0 ? 1u : 0u;
}

// row[i] = number of found blocks before request i (one workgroup).
__global__ __launch_bounds__(1024) void vb_halo_rows(const uint32_t* __restrict__ found, int n, uint32_t* __restrict__ row) {
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

__global__ __launch_bounds__(256) void vb_halo_export(Directory dir, const float* __restrict__ dist,
                                                      const float* __restrict__ weight, const uint32_t* __restrict__ rgba,
                                                      const int32_t* __restrict__ ids, const uint32_t* __restrict__ found,
                                                      const uint32_t* __restrict__ row, uint32_t* __restrict__ payload) {
  const int i = blockIdx.x;
  if (!found[i]) return;
  __shared__ int s_slot;
  if (threadIdx.x == 0) s_slot = dir_find(dir, ids[3 * i], ids[3 * i + 1], ids[3 * i + 2]);
  __syncthreads();
  const int slot = s_slot;
  if (slot < 0) return;
  const size_t src = (size_t)slot * kBlockVox;
  uint4* dst = reinterpret_cast<uint4*>(payload + (size_t)row[i] * kVbHaloWords);
  const uint4* p0 = reinterpret_cast<const uint4*>(dist + src);
  const uint4* p1 = reinterpret_cast<const uint4*>(weight + src);
  const uint4* p2 = reinterpret_cast<const uint4*>(rgba + src);
  for (int v = threadIdx.x; v < kBlockVox / 4; v += 256) {
    dst[v] = p0[v];
    dst[kBlockVox / 4 + v] = p1[v];
    dst[2 * (kBlockVox / 4) + v] = p2[v];
  }
}

// id -> ghost slot (base + payload row); a block nobody has gets no entry (the look-up then says "does not exist").
__global__ void vb_halo_insert(Directory g, const int32_t* __restrict__ ids, const uint32_t* __restrict__ found,
                               const uint32_t* __restrict__ row, int n, int base) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !found[i]) return;
  const int x = ids[3 * i], y = ids[3 * i + 1], z = ids[3 * i + 2];
  unsigned long long key;
  if (!pack_block(x, y, z, &key)) return;
  uint32_t hsh = dir_hash(x, y, z, g.mask);
  for (uint32_t probe = 0; probe <= g.mask; ++probe) {
    unsigned long long cur = g.keys[hsh];
    if (cur == key) return;
    if (cur == kEmptyKey) {
      cur = atomicCAS(&g.keys[hsh], kEmptyKey, key);
      if (cur == kEmptyKey) {
        g.slots[hsh] = base + (int)row[i];
        return;
      }
      if (cur == key) return;
    }
    hsh = (hsh + 1) & g.mask;
  }
}

__global__ __launch_bounds__(256) void vb_halo_import(float* __restrict__ dist, float* __restrict__ weight,
                                                      uint32_t* __restrict__ rgba, const uint32_t* __restrict__ found,
                                                      const uint32_t* __restrict__ row, const uint32_t* __restrict__ payload,
                                                      int base) {
  const int i = blockIdx.x;
  if (!found[i]) return;
  const size_t dst = (size_t)(base + (int)row[i]) * kBlockVox;
  const uint4* src = reinterpret_cast<const uint4*>(payload + (size_t)row[i] * kVbHaloWords);
  uint4* p0 = reinterpret_cast<uint4*>(dist + dst);
  uint4* p1 = reinterpret_cast<uint4*>(weight + dst);
  uint4* p2 = reinterpret_cast<uint4*>(rgba + dst);
  for (int v = threadIdx.x; v < kBlockVox / 4; v += 256) {
    p0[v] = src[v];
    p1[v] = src[kBlockVox / 4 + v];
    p2[v] = src[2 * (kBlockVox / 4) + v];
  }
}

}  // namespace

// Drops the ghosts: the integrate calls allocate new blocks in the slots they occupy (never-used slots are all zero).
static int vb_halo_drop(plvs_tsdf_voxblox* h, hipStream_t s) {
  if (h->ghost_count == 0) return PLVS_OK;
  const size_t at = (size_t)h->num_blocks * kBlockVox, len = (size_t)h->ghost_count * kBlockVox;
  PLVS_HIP_TRY(hipMemsetAsync(h->dist + at, 0, len * sizeof(float), s));
  PLVS_HIP_TRY(hipMemsetAsync(h->weight + at, 0, len * sizeof(float), s));
  PLVS_HIP_TRY(hipMemsetAsync(h->rgba + at, 0, len * sizeof(uint32_t), s));
  PLVS_HIP_TRY(hipMemsetAsync(h->gdir.keys, 0xFF, ((size_t)h->gdir.mask + 1) * sizeof(unsigned long long), s));
  h->ghost_count = 0;
  return PLVS_OK;
}

extern "C" {

// Creates or REPLACES one block with the given voxel planes (host, 4096 each, index x + 16 * (y + 16 * z)): what
// Layer::addBlockFromProto does with BlockMergingStrategy::kReplace for every block of a saved layer
// (TsdfServer::loadMap, tsdf_server.cc:865-872 -> io::LoadBlocksFromFile; core/layer_inl.h:195-197, :215) once the
// protobuf has been read on the host.  The caller marks the block updated, as layer_inl.h:215 does.
int plvs_hip_tsdf_voxblox_upload_block(plvs_tsdf_voxblox* h, int bx, int by, int bz, const float* distance, const float* weight,
                                       const uint32_t* rgba) {
  PLVS_REQUIRE(h && distance && weight && rgba, "null argument");
  VB_FLUSH_QUEUE(h);
  PLVS_REQUIRE(!h->poisoned, "handle is in a failed state (clear it)");
  if (h->P.shard_count > 1)
    PLVS_REQUIRE(shard_of(owner_hash(bx, by, bz), h->P.shard_count) == h->P.shard_rank, "the block belongs to another rank");
  int rc = vb_halo_drop(h, nullptr);
  if (rc != PLVS_OK) return rc;
  PLVS_HIP_TRY(hipMemset(&h->d_ctr->err, 0, sizeof(uint32_t)));
  int32_t* d_slot = reinterpret_cast<int32_t*>(&h->d_ctr->total_visits);   // (a counter no call is using now)
  hipLaunchKernelGGL(vb_block_slot_of, dim3(1), dim3(64), 0, nullptr, h->dir, bx, by, bz, h->d_ctr, d_slot);
  PLVS_KERNEL_CHECK();
  rc = vb_read_counters(h, nullptr);
  if (rc != PLVS_OK) return rc;
  if (h->h_ctr->err) {
    h->poisoned = true;
    plvs::set_error("upload_block: %s", (h->h_ctr->err & kErrPoolFull) ? "block pool full (raise max_blocks)" : "block id out of range");
    return PLVS_ERR_CAPACITY;
  }
  const int slot = (int)h->h_ctr->total_visits;
  PLVS_REQUIRE(slot >= 0, "internal: the block was not inserted");
  h->num_blocks = h->h_ctr->num_blocks;
  h->visible_blocks = h->num_blocks;   // (a block handed to the layer directly: whatever waited joins with it)
  const size_t off = (size_t)slot * kBlockVox;
  PLVS_HIP_TRY(hipMemcpy(h->dist + off, distance, kBlockVox * sizeof(float), hipMemcpyHostToDevice));
  PLVS_HIP_TRY(hipMemcpy(h->weight + off, weight, kBlockVox * sizeof(float), hipMemcpyHostToDevice));
  PLVS_HIP_TRY(hipMemcpy(h->rgba + off, rgba, kBlockVox * sizeof(uint32_t), hipMemcpyHostToDevice));
  return PLVS_OK;
}

int plvs_hip_tsdf_voxblox_halo_lookup(plvs_tsdf_voxblox* h, const int32_t* d_ids_xyz, int n, uint32_t* d_found, void* stream) {
  PLVS_REQUIRE(h && !h->poisoned, "unusable handle");
  VB_FLUSH_QUEUE(h);
  PLVS_REQUIRE(n >= 0, "negative size");
  if (n == 0) return PLVS_OK;
  PLVS_REQUIRE(d_ids_xyz && d_found, "null argument");
  hipLaunchKernelGGL(vb_halo_lookup, dim3(ceil_div((size_t)n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), h->dir,
                     d_ids_xyz, n, d_found);
  PLVS_KERNEL_CHECK();
  return PLVS_OK;
}

int plvs_hip_tsdf_voxblox_halo_export(plvs_tsdf_voxblox* h, const int32_t* d_ids_xyz, const uint32_t* d_found, int n,
                                      uint32_t* d_payload, void* stream) {
  PLVS_REQUIRE(h && !h->poisoned, "unusable handle");
  VB_FLUSH_QUEUE(h);
  PLVS_REQUIRE(n >= 0, "negative size");
  if (n == 0 || d_payload == nullptr) return PLVS_OK;   // (no payload buffer: the caller saw no flag set)
  PLVS_REQUIRE(d_ids_xyz && d_found, "null argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  PLVS_HIP_TRY(h->halo_row.reserve((size_t)n));
  hipLaunchKernelGGL(vb_halo_rows, dim3(1), dim3(1024), 0, s, d_found, n, h->halo_row.p);
  hipLaunchKernelGGL(vb_halo_export, dim3((unsigned)n), dim3(256), 0, s, h->dir, h->dist, h->weight, h->rgba, d_ids_xyz,
                     d_found, h->halo_row.p, d_payload);
  PLVS_KERNEL_CHECK();
  return PLVS_OK;
}

int plvs_hip_tsdf_voxblox_halo_import(plvs_tsdf_voxblox* h, const int32_t* d_ids_xyz, const uint32_t* d_found,
                                      const uint32_t* d_payload, int n, int nfound, void* stream) {
  PLVS_REQUIRE(h && !h->poisoned, "unusable handle");
  VB_FLUSH_QUEUE(h);
  PLVS_REQUIRE(n >= 0 && nfound >= 0 && nfound <= n, "bad sizes");
  if (n == 0 || nfound == 0) return PLVS_OK;
  PLVS_REQUIRE(d_ids_xyz && d_found && d_payload, "null argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if ((long long)h->num_blocks + h->ghost_count + nfound > (long long)h->prm.max_blocks) {
    plvs::set_error("halo_import: %d own + %d ghost + %d new blocks exceed the pool (%d)", h->num_blocks, h->ghost_count, nfound,
                    h->prm.max_blocks);
    return PLVS_ERR_CAPACITY;
  }
  if (h->gdir.keys == nullptr) {
    size_t cap = 1024;
    while (cap < 2 * (size_t)h->prm.max_blocks) cap <<= 1;
    PLVS_HIP_TRY(hipMalloc(&h->gdir.keys, cap * sizeof(unsigned long long)));
    PLVS_HIP_TRY(hipMalloc(&h->gdir.slots, cap * sizeof(int32_t)));
    h->gdir.slot_ids = nullptr;
    h->gdir.mask = (uint32_t)(cap - 1);
    h->gdir.max_blocks = h->prm.max_blocks;
    PLVS_HIP_TRY(hipMemsetAsync(h->gdir.keys, 0xFF, cap * sizeof(unsigned long long), s));
  }
  const int base = h->num_blocks + h->ghost_count;
  PLVS_HIP_TRY(h->halo_row.reserve((size_t)n));
  hipLaunchKernelGGL(vb_halo_rows, dim3(1), dim3(1024), 0, s, d_found, n, h->halo_row.p);
  hipLaunchKernelGGL(vb_halo_insert, dim3(ceil_div((size_t)n, 256)), dim3(256), 0, s, h->gdir, d_ids_xyz, d_found, h->halo_row.p, n,
                     base);
  hipLaunchKernelGGL(vb_halo_import, dim3((unsigned)n), dim3(256), 0, s, h->dist, h->weight, h->rgba, d_found, h->halo_row.p,
                     d_payload, base);
  PLVS_KERNEL_CHECK();
  h->ghost_count += nfound;
  return PLVS_OK;
}

int plvs_hip_tsdf_voxblox_halo_clear(plvs_tsdf_voxblox* h) {
  PLVS_REQUIRE(h, "null handle");
  VB_FLUSH_QUEUE(h);
  int rc = vb_halo_drop(h, nullptr);
  if (rc != PLVS_OK) return rc;
  PLVS_HIP_TRY(hipStreamSynchronize(nullptr));
  return PLVS_OK;
}

}  // extern "C"

// Blocks a world cloud left waiting (slots [lo, hi)) have just joined the layer with their Block::updated() flags set
// (integrateWorlPointCloud marks them, tsdf_integrator.cc:76-80): they belong to this call's updated list even if this
// call's rays did not touch them.  Rare (once after a LoadMap): done on the host.
static int vb_publish_waiting(plvs_tsdf_voxblox* h, hipStream_t s, int published_lo, int published_hi) {
  if (published_hi <= published_lo) return PLVS_OK;
  std::vector<uint32_t> upd(h->last_updated);
  if (!upd.empty()) PLVS_HIP_TRY(hipMemcpyAsync(upd.data(), h->updated.p, upd.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  std::vector<uint8_t> in((size_t)(published_hi - published_lo), 0);
  for (uint32_t v : upd)
    if ((int)v >= published_lo && (int)v < published_hi) in[(size_t)((int)v - published_lo)] = 1;
  for (int v = published_lo; v < published_hi; ++v)
    if (!in[(size_t)(v - published_lo)]) upd.push_back((uint32_t)v);
  std::sort(upd.begin(), upd.end());
  PLVS_HIP_TRY(h->upd_merge.reserve(upd.size() + 1));
  PLVS_HIP_TRY(hipMemcpy(h->upd_merge.p, upd.data(), upd.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  std::swap(h->updated.p, h->upd_merge.p);
  std::swap(h->updated.cap, h->upd_merge.cap);
  h->last_updated = (uint32_t)upd.size();
  h->stats.updated_chunks = (int32_t)upd.size();
  return PLVS_OK;
}

// mode kWorld: the world-cloud-with-normals flavour (integrateWorlPointCloud), d_aux = normals; mode kMerged:
// MergedTsdfIntegrator's bundles, d_aux = merged weights, d_clr = clearing flags.  Both: one cloud.
static int vb_fast_tables(plvs_tsdf_voxblox* h, hipStream_t s, bool reset) {
  if (h->ap_ready && !reset) return PLVS_OK;
  PLVS_HIP_TRY(h->ap_start.reserve(kApproxWords));
  PLVS_HIP_TRY(h->ap_seen.reserve(kApproxWords));
  if (!h->h_ff) PLVS_HIP_TRY(hipHostMalloc((void**)&h->h_ff, 2 * sizeof(uint32_t)));
  hipLaunchKernelGGL(vbf_init_table, dim3(ceil_div(kApproxWords, 256)), dim3(256), 0, s, h->ap_start.p, kApproxWords);
  hipLaunchKernelGGL(vbf_init_table, dim3(ceil_div(kApproxWords, 256)), dim3(256), 0, s, h->ap_seen.p, kApproxWords);
  PLVS_KERNEL_CHECK();
  h->ap_ready = true;
  return PLVS_OK;
}

// Which rays of the batch's scans are cast and how many voxels each updates (h->ff_L, per sequence position): the
// rounds described in tsdf_voxblox_fast.hpp.  The offsets and poses of the call are on the device already.
static int vb_fast_plan(plvs_tsdf_voxblox* h, const float* d_xyz, int n, int nclouds, const PoseRt* d_poses, uint32_t first_offset,
                        hipStream_t s) {
  const unsigned nb = ceil_div((size_t)n, 256);
  PLVS_HIP_TRY(h->ff_skey0.reserve((size_t)n)); PLVS_HIP_TRY(h->ff_skey1.reserve((size_t)n));
  PLVS_HIP_TRY(h->ff_sval0.reserve((size_t)n)); PLVS_HIP_TRY(h->ff_sval1.reserve((size_t)n));
  PLVS_HIP_TRY(h->ff_shash.reserve((size_t)n)); PLVS_HIP_TRY(h->ff_full.reserve((size_t)n));
  PLVS_HIP_TRY(h->ff_Q.reserve((size_t)n)); PLVS_HIP_TRY(h->ff_L.reserve((size_t)n));
  PLVS_HIP_TRY(h->ff_qoff.reserve((size_t)n)); PLVS_HIP_TRY(h->ff_flags.reserve(2));
  PLVS_HIP_TRY(h->scratch.reserve(std::max(radix_scratch_words((size_t)n), scan_scratch_words((size_t)n))));
  hipLaunchKernelGGL(vbf_start, dim3(nb), dim3(256), 0, s, h->P, d_xyz, n, h->offsets.p, nclouds, d_poses, first_offset,
                     h->ff_skey0.p, h->ff_sval0.p, h->ff_shash.p, h->ff_full.p, h->d_ctr);
  PLVS_KERNEL_CHECK();
  bool second = false;
  PLVS_HIP_TRY(radix_sort_pairs(h->ff_skey0.p, h->ff_sval0.p, h->ff_skey1.p, h->ff_sval1.p, (size_t)n, 0, kApproxKeyBits,
                                h->scratch.p, s, &second));
  const uint32_t* sk = second ? h->ff_skey1.p : h->ff_skey0.p;
  const uint32_t* sv = second ? h->ff_sval1.p : h->ff_sval0.p;
  hipLaunchKernelGGL(vbf_alive, dim3(nb), dim3(256), 0, s, sk, sv, (uint32_t)n, h->ff_shash.p, h->ap_start.p, h->ff_full.p, h->ff_Q.p);
  hipLaunchKernelGGL(vbf_write_back, dim3(nb), dim3(256), 0, s, sk, sv, (uint32_t)n, h->ff_shash.p, h->ap_start.p);
  PLVS_KERNEL_CHECK();
  h->fast_rounds = 0;
  h->fast_sequential = false;
  PLVS_HIP_TRY(hipMemsetAsync(h->ff_flags.p, 0, 2 * sizeof(uint32_t), s));
  const uint32_t *qk = nullptr, *qv = nullptr;
  uint32_t M = 0;
  for (;;) {
    // queries of the round; did the round before change anything?
    PLVS_HIP_TRY(exclusive_scan_u32(h->ff_Q.p, h->ff_qoff.p, (size_t)n, h->ff_flags.p, h->scratch.p, s));
    hipLaunchKernelGGL(vbf_publish, dim3(1), dim3(1), 0, s, h->ff_flags.p, h->ff_flags.p + 1, h->h_ff);
    PLVS_KERNEL_CHECK();
    PLVS_HIP_TRY(hipStreamSynchronize(s));
    if (h->fast_rounds > 0 && h->h_ff[1] == 0) break;   // (the sorted queries of the last round are the scans' queries)
    {   // a cloud whose rounds do not settle (one round per ray at worst): finish on one thread, in the reference's own order
      static const int max_rounds = plvs::env_int("PLVS_VB_FAST_MAX_ROUNDS", 512, 0, 100000);
      if (h->fast_rounds >= max_rounds) {
        hipLaunchKernelGGL(vbf_sequential, dim3(1), dim3(1), 0, s, h->P, d_xyz, n, h->offsets.p, nclouds, d_poses, first_offset,
                           h->ff_Q.p, h->ff_full.p, h->ap_seen.p, h->ff_L.p);
        PLVS_KERNEL_CHECK();
        h->fast_sequential = true;
        return PLVS_OK;   // (the observed set already holds what the queries leave)
      }
    }
    M = h->h_ff[0];
    ++h->fast_rounds;
    PLVS_HIP_TRY(hipMemsetAsync(h->ff_flags.p + 1, 0, sizeof(uint32_t), s));
    if (M == 0) {   // no ray at all: nothing to ask
      hipLaunchKernelGGL(vbf_trim, dim3(nb), dim3(256), 0, s, n, h->ff_qoff.p, h->ff_full.p, (const uint8_t*)nullptr, h->ff_Q.p,
                         h->ff_L.p, h->ff_flags.p + 1);
      PLVS_KERNEL_CHECK();
      continue;
    }
    PLVS_HIP_TRY(h->ff_qkey0.reserve(M)); PLVS_HIP_TRY(h->ff_qkey1.reserve(M));
    PLVS_HIP_TRY(h->ff_qval0.reserve(M)); PLVS_HIP_TRY(h->ff_qval1.reserve(M));
    PLVS_HIP_TRY(h->ff_qhash.reserve(M)); PLVS_HIP_TRY(h->ff_seen.reserve(M));
    PLVS_HIP_TRY(h->scratch.reserve(std::max(radix_scratch_words((size_t)M), scan_scratch_words((size_t)n))));
    hipLaunchKernelGGL(vbf_emit, dim3(nb), dim3(256), 0, s, h->P, d_xyz, n, h->offsets.p, nclouds, d_poses, first_offset, h->ff_Q.p,
                       h->ff_qoff.p, h->ff_qkey0.p, h->ff_qval0.p, h->ff_qhash.p);
    PLVS_KERNEL_CHECK();
    bool sec = false;
    PLVS_HIP_TRY(radix_sort_pairs(h->ff_qkey0.p, h->ff_qval0.p, h->ff_qkey1.p, h->ff_qval1.p, (size_t)M, 0, kApproxKeyBits,
                                  h->scratch.p, s, &sec));
    qk = sec ? h->ff_qkey1.p : h->ff_qkey0.p;
    qv = sec ? h->ff_qval1.p : h->ff_qval0.p;
    hipLaunchKernelGGL(vbf_seen, dim3(ceil_div((size_t)M, 256)), dim3(256), 0, s, qk, qv, M, h->ff_qhash.p, h->ap_seen.p, h->ff_seen.p);
    hipLaunchKernelGGL(vbf_trim, dim3(nb), dim3(256), 0, s, n, h->ff_qoff.p, h->ff_full.p, h->ff_seen.p, h->ff_Q.p, h->ff_L.p,
                       h->ff_flags.p + 1);
    PLVS_KERNEL_CHECK();
  }
  if (M > 0 && qk) {
    hipLaunchKernelGGL(vbf_write_back, dim3(ceil_div((size_t)M, 256)), dim3(256), 0, s, qk, qv, M, h->ff_qhash.p, h->ap_seen.p);
    PLVS_KERNEL_CHECK();
  }
  return PLVS_OK;
}

static int vb_integrate_impl(plvs_tsdf_voxblox* h, const float* d_xyz, const uint8_t* d_rgba, const int32_t* offsets,
                             int nclouds, const float* d_Twc, void* stream, int mode, const float* d_aux,
                             const uint8_t* d_clr, uint32_t fast_offset = 0) {
  PLVS_REQUIRE(h, "null handle");
  PLVS_REQUIRE(!h->poisoned, "handle is in a failed state (clear it)");
  PLVS_REQUIRE(offsets && nclouds >= 0, "bad offsets");
  hipStream_t s = static_cast<hipStream_t>(stream);
  h->stats = plvs_tsdf_stats{};
  h->last_updated = 0;
  // a camera cloud, even an empty one, starts with updateLayerWithStoredBlocks (tsdf_integrator.cc:306 / :343)
  auto publish_all = [&]() -> int {
    if (mode == kWorld) return PLVS_OK;
    const int lo = h->visible_blocks, hi = h->num_blocks;
    h->visible_blocks = h->num_blocks;
    return vb_publish_waiting(h, s, lo, hi);
  };
  if (nclouds == 0) return publish_all();
  const int n = offsets[nclouds] - offsets[0];
  PLVS_REQUIRE(offsets[0] == 0 && n >= 0, "offsets must start at 0 and be non-decreasing");
  for (int c = 0; c < nclouds; ++c) PLVS_REQUIRE(offsets[c + 1] >= offsets[c], "offsets must be non-decreasing");
  h->stats.points = n;
  if (n == 0) return publish_all();
  PLVS_REQUIRE(d_xyz && d_rgba && d_Twc, "null device pointer");
  PLVS_REQUIRE((reinterpret_cast<uintptr_t>(d_rgba) & 3) == 0, "rgba must be 4-byte aligned");
  {
    int rc = vb_halo_drop(h, s);   // new blocks go into the pool slots a meshing halo may still occupy
    if (rc != PLVS_OK) return rc;
  }
  const uint32_t* d_col = reinterpret_cast<const uint32_t*>(d_rgba);

  PLVS_HIP_TRY(h->offsets.reserve((size_t)nclouds + 1));
  PLVS_HIP_TRY(h->counts.reserve((size_t)n));
  PLVS_HIP_TRY(h->scratch.reserve(scan_scratch_words((size_t)n)));
  PLVS_HIP_TRY(hipMemcpyAsync(h->offsets.p, offsets, ((size_t)nclouds + 1) * sizeof(int32_t),
                              hipMemcpyHostToDevice, s));
  PLVS_HIP_TRY(hipMemsetAsync(&h->d_ctr->total_visits, 0, sizeof(uint32_t), s));
  PLVS_HIP_TRY(hipMemsetAsync(&h->d_ctr->err, 0, 4 * sizeof(uint32_t), s));
  PLVS_HIP_TRY(h->poses.reserve((size_t)nclouds));
  hipLaunchKernelGGL(vb_pose_prep, dim3(ceil_div((size_t)nclouds, 64)), dim3(64), 0, s, d_Twc, nclouds, h->poses.p);
  const PoseRt* const d_poses = h->poses.p;
  const dim3 rgrid(ceil_div((size_t)n, 256)), rblock(256);
  if (mode == kFast) {   // which rays are cast, and how far: the update counts take aux's place
    int rcf = vb_fast_plan(h, d_xyz, n, nclouds, d_poses, fast_offset, s);
    if (rcf != PLVS_OK) return rcf;
    d_aux = reinterpret_cast<const float*>(h->ff_L.p);
  }
#define VB_RAY_PASS(FILL, MODE, K, Q)                                                                                    \
  hipLaunchKernelGGL((vb_ray_pass<FILL, MODE>), rgrid, rblock, 0, s, h->P, d_xyz, d_aux, d_clr, n, h->offsets.p, nclouds, \
                     d_poses, h->dir, h->d_ctr, h->counts.p, K, Q)
  if (mode == kWorld) VB_RAY_PASS(false, kWorld, (uint32_t*)nullptr, (uint32_t*)nullptr);
  else if (mode == kMerged) VB_RAY_PASS(false, kMerged, (uint32_t*)nullptr, (uint32_t*)nullptr);
  else if (mode == kFast) VB_RAY_PASS(false, kFast, (uint32_t*)nullptr, (uint32_t*)nullptr);
  else VB_RAY_PASS(false, kSimple, (uint32_t*)nullptr, (uint32_t*)nullptr);
  PLVS_KERNEL_CHECK();
  PLVS_HIP_TRY(exclusive_scan_u32(h->counts.p, h->counts.p, (size_t)n, &h->d_ctr->total_visits,
                                  h->scratch.p, s));
  int rc = vb_read_counters(h, s);
  if (rc != PLVS_OK) return rc;
  if (h->h_ctr->err) {
    h->poisoned = true;
    plvs::set_error("tsdf_voxblox integrate: %s%s%s",
                    (h->h_ctr->err & kErrPoolFull) ? "block pool full (raise max_blocks) " : "",
                    (h->h_ctr->err & kErrCoordRange) ? "block id outside +-2^20 " : "",
                    (h->h_ctr->err & kErrNonFinite) ? "non-finite point in the cloud " : "");
    return (h->h_ctr->err & kErrNonFinite) ? PLVS_ERR_INVALID_ARG : PLVS_ERR_CAPACITY;
  }
  const uint32_t V = h->h_ctr->total_visits;
  const int before = h->num_blocks;
  h->num_blocks = h->h_ctr->num_blocks;
  // integratePointCloud starts with updateLayerWithStoredBlocks (tsdf_integrator.cc:306, :343): whatever a world cloud
  // left waiting joins the layer now; integrateWorlPointCloud itself never calls it (:35-82)
  const int published_lo = h->visible_blocks, published_hi = (mode == kWorld && h->defer_world_blocks) ? h->visible_blocks : before;
  if (!(mode == kWorld && h->defer_world_blocks)) h->visible_blocks = h->num_blocks;
  h->stats.visits = V;
  h->stats.new_chunks = h->num_blocks - before;
  if (V == 0) return vb_publish_waiting(h, s, published_lo, published_hi);
  PLVS_HIP_TRY(h->keys0.reserve(V));
  PLVS_HIP_TRY(h->keys1.reserve(V));
  PLVS_HIP_TRY(h->seq0.reserve(V));
  PLVS_HIP_TRY(h->seq1.reserve(V));
  PLVS_HIP_TRY(h->heads.reserve(V));
  PLVS_HIP_TRY(h->rec.reserve(V));
  PLVS_HIP_TRY(h->rec_c.reserve(V));
  PLVS_HIP_TRY(h->updated.reserve((size_t)h->num_blocks + 1));
  PLVS_HIP_TRY(h->scratch.reserve(radix_scratch_words(V)));
  if (mode == kWorld) VB_RAY_PASS(true, kWorld, h->keys0.p, h->seq0.p);
  else if (mode == kMerged) VB_RAY_PASS(true, kMerged, h->keys0.p, h->seq0.p);
  else if (mode == kFast) VB_RAY_PASS(true, kFast, h->keys0.p, h->seq0.p);
  else VB_RAY_PASS(true, kSimple, h->keys0.p, h->seq0.p);
#undef VB_RAY_PASS
  PLVS_KERNEL_CHECK();
  int key_bits = 12;
  while ((1ll << (key_bits - 12)) < (long long)h->num_blocks) ++key_bits;
  bool second = false;
  PLVS_HIP_TRY(radix_sort_pairs(h->keys0.p, h->seq0.p, h->keys1.p, h->seq1.p, V, 0, key_bits,
                                h->scratch.p, s, &second));
  const uint32_t* keys = second ? h->keys1.p : h->keys0.p;
  const uint32_t* seqs = second ? h->seq1.p : h->seq0.p;
  // (Folding the records in slices on a second stream while the next slice is expanded was measured in round 3: the two
  // kernels slow each other down by more than the overlap gains — 1.03-1.13 ms with two slices, 1.08-1.20 with four,
  // against 0.93-1.03 for one after the other.)
#define VB_EXPAND(MODE)                                                                                             \
  hipLaunchKernelGGL(vb_expand<MODE>, dim3(ceil_div(V, kExpandThreads)), dim3(kExpandThreads), 0, s, h->P, keys, seqs, V, \
                     d_xyz, d_aux, d_col, h->offsets.p, nclouds, d_poses, h->dir.slot_ids, h->rec.p, h->rec_c.p,         \
                     h->heads.p, h->updated.p, h->d_ctr)
  if (mode == kWorld) VB_EXPAND(kWorld);
  else if (mode == kMerged) VB_EXPAND(kMerged);
  else VB_EXPAND(kSimple);
#undef VB_EXPAND
  PLVS_KERNEL_CHECK();
  hipLaunchKernelGGL(vb_chain_chunks, dim3(ceil_div(V, kChainChunk)), dim3(kChainThreads), 0, s, h->P, keys, V, h->rec.p,
                     h->rec_c.p, h->d_ctr, h->dist, h->weight, h->rgba);
  PLVS_KERNEL_CHECK();
  rc = vb_read_counters(h, s);
  if (rc != PLVS_OK) return rc;
  if (h->h_ctr->err) {
    h->poisoned = true;
    plvs::set_error("tsdf_voxblox integrate: internal directory miss (err=%u)", h->h_ctr->err);
    return PLVS_ERR_CAPACITY;
  }
  h->stats.updated_chunks = (int32_t)h->h_ctr->num_updated;
  h->stats.voxels = (int32_t)h->h_ctr->num_heads;
  h->stats.max_run = (int32_t)h->h_ctr->max_run;
  h->last_updated = h->h_ctr->num_updated;
  {
    int rc2 = vb_publish_waiting(h, s, published_lo, published_hi);
    if (rc2 != PLVS_OK) return rc2;
  }
  if (h->visible_blocks < h->num_blocks && h->last_updated > 0) {   // Block::updated() of a block outside the layer is not seen
    hipLaunchKernelGGL(vb_filter_slots, dim3(1), dim3(1024), 0, s, h->updated.p, h->last_updated, (uint32_t)h->visible_blocks,
                       &h->d_ctr->num_updated);
    PLVS_KERNEL_CHECK();
    rc = vb_read_counters(h, s);
    if (rc != PLVS_OK) return rc;
    h->last_updated = h->h_ctr->num_updated;
  }
  return PLVS_OK;
}

extern "C" {

int plvs_hip_tsdf_voxblox_integrate_batch_dev(plvs_tsdf_voxblox* h, const float* d_xyz,
                                              const uint8_t* d_rgba, const int32_t* offsets,
                                              int nclouds, const float* d_Twc, void* stream) {
  VB_FLUSH_QUEUE(h);
  return vb_integrate_impl(h, d_xyz, d_rgba, offsets, nclouds, d_Twc, stream, kSimple, nullptr, nullptr);
}

// ---- the ray-sharded integrate (tsdf_voxblox_shard.hpp)
int plvs_hip_tsdf_voxblox_shard_walk(plvs_tsdf_voxblox* h, const float* d_xyz, const int32_t* offsets, int nclouds,
                                     const float* d_Twc, int64_t* send_counts, void* stream) {
  PLVS_REQUIRE(h && offsets && send_counts && nclouds >= 0, "bad argument");
  PLVS_REQUIRE(!h->poisoned, "handle is in a failed state (clear it)");
  VB_FLUSH_QUEUE(h);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int N = std::max(1, h->P.shard_count), rank = N > 1 ? h->P.shard_rank : 0;
  PLVS_REQUIRE(N <= 64, "at most 64 ranks");
  for (int p = 0; p < N; ++p) send_counts[p] = 0;
  h->sv_V = 0;
  h->sv_phase = 1;
  h->sv_partitioned = false;
  if (nclouds == 0) return PLVS_OK;
  const int n = offsets[nclouds] - offsets[0];
  PLVS_REQUIRE(offsets[0] == 0 && n >= 0, "offsets must start at 0 and be non-decreasing");
  for (int c = 0; c < nclouds; ++c) PLVS_REQUIRE(offsets[c + 1] >= offsets[c], "offsets must be non-decreasing");
  PLVS_REQUIRE((size_t)nclouds < ((size_t)1 << 20), "at most 2^20 clouds per call");
  if (n == 0) return PLVS_OK;
  PLVS_REQUIRE(d_xyz && d_Twc, "null device pointer");
  if (!h->h_sv_cnt) PLVS_HIP_TRY(hipHostMalloc((void**)&h->h_sv_cnt, 64 * sizeof(uint32_t)));
  PLVS_HIP_TRY(h->offsets.reserve((size_t)nclouds + 1));
  PLVS_HIP_TRY(h->counts.reserve((size_t)n));
  PLVS_HIP_TRY(h->scratch.reserve(scan_scratch_words((size_t)n)));
  PLVS_HIP_TRY(h->sv_cnt.reserve(64));
  PLVS_HIP_TRY(hipMemcpyAsync(h->offsets.p, offsets, ((size_t)nclouds + 1) * sizeof(int32_t), hipMemcpyHostToDevice, s));
  PLVS_HIP_TRY(hipMemsetAsync(&h->d_ctr->total_visits, 0, sizeof(uint32_t), s));
  PLVS_HIP_TRY(hipMemsetAsync(&h->d_ctr->err, 0, 4 * sizeof(uint32_t), s));
  PLVS_HIP_TRY(h->poses.reserve((size_t)nclouds));
  hipLaunchKernelGGL(vb_pose_prep, dim3(ceil_div((size_t)nclouds, 64)), dim3(64), 0, s, d_Twc, nclouds, h->poses.p);
  Params Pw = h->P;   // this rank's rays go through every block they cross
  Pw.shard_count = 1;
  Pw.shard_rank = 0;
  const dim3 rgrid(ceil_div((size_t)n, 256)), rblock(256);
  hipLaunchKernelGGL(vb_shard_ray_pass<false>, rgrid, rblock, 0, s, Pw, d_xyz, n, h->offsets.p, nclouds, h->poses.p, rank, N,
                     h->d_ctr, h->counts.p, (uint4*)nullptr, (uint32_t*)nullptr);
  PLVS_KERNEL_CHECK();
  PLVS_HIP_TRY(exclusive_scan_u32(h->counts.p, h->counts.p, (size_t)n, &h->d_ctr->total_visits, h->scratch.p, s));
  int rc = vb_read_counters(h, s);
  if (rc != PLVS_OK) return rc;
  if (h->h_ctr->err) {
    plvs::set_error("tsdf_voxblox shard_walk: %s%s", (h->h_ctr->err & kErrCoordRange) ? "block id outside +-2^20 " : "",
                    (h->h_ctr->err & kErrNonFinite) ? "non-finite point in the cloud " : "");
    return (h->h_ctr->err & kErrNonFinite) ? PLVS_ERR_INVALID_ARG : PLVS_ERR_CAPACITY;
  }
  const uint32_t V = h->h_ctr->total_visits;
  h->sv_V = V;
  if (V == 0) return PLVS_OK;
  PLVS_HIP_TRY(h->sv_rec.reserve(V));
  PLVS_HIP_TRY(h->sv_dest.reserve(V));
  hipLaunchKernelGGL(vb_shard_ray_pass<true>, rgrid, rblock, 0, s, Pw, d_xyz, n, h->offsets.p, nclouds, h->poses.p, rank, N,
                     h->d_ctr, h->counts.p, h->sv_rec.p, h->sv_dest.p);
  PLVS_KERNEL_CHECK();
  if (N == 1) {
    send_counts[0] = (int64_t)V;
    return PLVS_OK;
  }
  // stable partition by destination: one radix pass over (destination, record index)
  PLVS_HIP_TRY(h->sv_dest1.reserve(V));
  PLVS_HIP_TRY(h->sv_idx.reserve(V));
  PLVS_HIP_TRY(h->sv_idx1.reserve(V));
  PLVS_HIP_TRY(h->scratch.reserve(radix_scratch_words(V)));
  hipLaunchKernelGGL(vb_iota, dim3(ceil_div((size_t)V, 256)), dim3(256), 0, s, h->sv_idx.p, V);
  int bits = 1;
  while ((1 << bits) < N) ++bits;
  bool second = false;
  PLVS_HIP_TRY(radix_sort_pairs(h->sv_dest.p, h->sv_idx.p, h->sv_dest1.p, h->sv_idx1.p, V, 0, bits, h->scratch.p, s, &second));
  if (second) {   // (the gather of shard_pack reads sv_idx, the counts below sv_dest)
    std::swap(h->sv_dest, h->sv_dest1);
    std::swap(h->sv_idx, h->sv_idx1);
  }
  h->sv_partitioned = true;
  hipLaunchKernelGGL(vb_shard_dest_counts, dim3(1), dim3(64), 0, s, h->sv_dest.p, V, N, h->sv_cnt.p);
  PLVS_KERNEL_CHECK();
  PLVS_HIP_TRY(hipMemcpyAsync(h->h_sv_cnt, h->sv_cnt.p, (size_t)N * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  for (int p = 0; p < N; ++p) send_counts[p] = (int64_t)h->h_sv_cnt[p];
  return PLVS_OK;
}

int plvs_hip_tsdf_voxblox_shard_pack(plvs_tsdf_voxblox* h, void* d_send, void* stream) {
  PLVS_REQUIRE(h, "null handle");
  PLVS_REQUIRE(h->sv_phase == 1, "shard_pack follows shard_walk");
  h->sv_phase = 2;
  if (h->sv_V == 0) return PLVS_OK;
  PLVS_REQUIRE(d_send, "null send buffer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(vb_shard_gather, dim3(ceil_div((size_t)h->sv_V, 256)), dim3(256), 0, s, h->sv_rec.p,
                     h->sv_partitioned ? (const uint32_t*)h->sv_idx.p : (const uint32_t*)nullptr, h->sv_V, static_cast<uint4*>(d_send));
  PLVS_KERNEL_CHECK();
  return PLVS_OK;
}

int plvs_hip_tsdf_voxblox_shard_apply(plvs_tsdf_voxblox* h, const void* d_recv, const int64_t* recv_counts, const float* d_xyz,
                                      const uint8_t* d_rgba, const int32_t* offsets, int nclouds, const float* d_Twc, void* stream) {
  PLVS_REQUIRE(h && recv_counts && offsets && nclouds >= 0, "bad argument");
  PLVS_REQUIRE(!h->poisoned, "handle is in a failed state (clear it)");
  PLVS_REQUIRE(h->sv_phase == 2, "shard_apply follows shard_pack");
  h->sv_phase = 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int N = std::max(1, h->P.shard_count), rank = N > 1 ? h->P.shard_rank : 0;
  h->stats = plvs_tsdf_stats{};
  h->last_updated = 0;
  size_t total = 0;
  for (int p = 0; p < N; ++p) {
    PLVS_REQUIRE(recv_counts[p] >= 0, "negative receive count");
    total += (size_t)recv_counts[p];
  }
  PLVS_REQUIRE(total < 0xFFFFFFFFull, "receive buffer beyond the index range (split the batch)");
  // a camera cloud, even an empty one, starts with updateLayerWithStoredBlocks (tsdf_integrator.cc:306 / :343)
  const int n = nclouds > 0 ? offsets[nclouds] - offsets[0] : 0;
  h->stats.points = n;
  if (total == 0) {
    const int lo = h->visible_blocks, hi = h->num_blocks;
    h->visible_blocks = h->num_blocks;
    return vb_publish_waiting(h, s, lo, hi);
  }
  PLVS_REQUIRE(d_recv && d_xyz && d_rgba && d_Twc, "null device pointer");
  PLVS_REQUIRE((reinterpret_cast<uintptr_t>(d_rgba) & 3) == 0, "rgba must be 4-byte aligned");
  {
    int rc = vb_halo_drop(h, s);   // new blocks go into the pool slots a meshing halo may still occupy
    if (rc != PLVS_OK) return rc;
  }
  const uint32_t V = (uint32_t)total;
  const uint32_t* d_col = reinterpret_cast<const uint32_t*>(d_rgba);
  PLVS_HIP_TRY(h->offsets.reserve((size_t)nclouds + 1));
  PLVS_HIP_TRY(hipMemcpyAsync(h->offsets.p, offsets, ((size_t)nclouds + 1) * sizeof(int32_t), hipMemcpyHostToDevice, s));
  PLVS_HIP_TRY(h->poses.reserve((size_t)nclouds));
  hipLaunchKernelGGL(vb_pose_prep, dim3(ceil_div((size_t)nclouds, 64)), dim3(64), 0, s, d_Twc, nclouds, h->poses.p);
  PLVS_HIP_TRY(hipMemsetAsync(&h->d_ctr->total_visits, 0, sizeof(uint32_t), s));
  PLVS_HIP_TRY(hipMemsetAsync(&h->d_ctr->err, 0, 4 * sizeof(uint32_t), s));
  PLVS_HIP_TRY(h->keys0.reserve(V));
  PLVS_HIP_TRY(h->keys1.reserve(V));
  PLVS_HIP_TRY(h->seq0.reserve(V));
  PLVS_HIP_TRY(h->seq1.reserve(V));
  PLVS_HIP_TRY(h->heads.reserve(V));
  PLVS_HIP_TRY(h->rec.reserve(V));
  PLVS_HIP_TRY(h->rec_c.reserve(V));
  PLVS_HIP_TRY(h->scratch.reserve(radix_scratch_words(V)));
  const bool by_cloud = N > 1;   // (one source: the records are in sequence order already)
  if (by_cloud) {
    PLVS_HIP_TRY(h->sv_vkey.reserve(V));
    PLVS_HIP_TRY(h->sv_seq.reserve(V));
  }
  hipLaunchKernelGGL(vb_shard_insert, dim3(ceil_div((size_t)V, 256)), dim3(256), 0, s, static_cast<const uint4*>(d_recv), V, h->dir,
                     rank, N, h->d_ctr);
  hipLaunchKernelGGL(vb_shard_translate, dim3(ceil_div((size_t)V, 256)), dim3(256), 0, s, static_cast<const uint4*>(d_recv), V,
                     h->dir, h->d_ctr, by_cloud ? h->sv_vkey.p : h->keys0.p, by_cloud ? h->sv_seq.p : h->seq0.p,
                     by_cloud ? h->keys0.p : (uint32_t*)nullptr, by_cloud ? h->seq0.p : (uint32_t*)nullptr);
  PLVS_KERNEL_CHECK();
  int rc = vb_read_counters(h, s);
  if (rc != PLVS_OK) return rc;
  if (h->h_ctr->err) {
    h->poisoned = true;
    plvs::set_error("tsdf_voxblox shard_apply: %s%s%s", (h->h_ctr->err & kErrPoolFull) ? "block pool full (raise max_blocks) " : "",
                    (h->h_ctr->err & kErrCoordRange) ? "block id outside +-2^20 " : "",
                    (h->h_ctr->err & kErrDirectoryMiss) ? "a record for a block of another rank " : "");
    return PLVS_ERR_CAPACITY;
  }
  const int before = h->num_blocks;
  h->num_blocks = h->h_ctr->num_blocks;
  const int published_lo = h->visible_blocks, published_hi = before;
  h->visible_blocks = h->num_blocks;
  h->stats.visits = V;
  h->stats.new_chunks = h->num_blocks - before;
  PLVS_HIP_TRY(h->updated.reserve((size_t)h->num_blocks + 1));
  bool second = false;
  if (by_cloud) {   // (cloud, record index) -> the records in cloud order, each cloud's in its sender's (sequence) order
    int cbits = 1;
    while ((1ll << cbits) < (long long)nclouds) ++cbits;
    PLVS_HIP_TRY(radix_sort_pairs(h->keys0.p, h->seq0.p, h->keys1.p, h->seq1.p, V, 0, cbits, h->scratch.p, s, &second));
    const uint32_t* order = second ? h->seq1.p : h->seq0.p;
    uint32_t* k_out = second ? h->keys0.p : h->keys1.p;   // (the pair of buffers the sort has left free)
    uint32_t* q_out = second ? h->seq0.p : h->seq1.p;
    hipLaunchKernelGGL(vb_shard_permute, dim3(ceil_div((size_t)V, 256)), dim3(256), 0, s, h->sv_vkey.p, h->sv_seq.p, order, V, k_out, q_out);
    PLVS_KERNEL_CHECK();
    if (!second) {   // the permuted records sit in keys1 / seq1: make them the sort's input pair
      std::swap(h->keys0, h->keys1);
      std::swap(h->seq0, h->seq1);
    }
  }
  int key_bits = 12;
  while ((1ll << (key_bits - 12)) < (long long)h->num_blocks) ++key_bits;
  second = false;
  PLVS_HIP_TRY(radix_sort_pairs(h->keys0.p, h->seq0.p, h->keys1.p, h->seq1.p, V, 0, key_bits, h->scratch.p, s, &second));
  const uint32_t* keys = second ? h->keys1.p : h->keys0.p;
  const uint32_t* seqs = second ? h->seq1.p : h->seq0.p;
  hipLaunchKernelGGL(vb_expand<kSimple>, dim3(ceil_div(V, kExpandThreads)), dim3(kExpandThreads), 0, s, h->P, keys, seqs, V, d_xyz,
                     (const float*)nullptr, d_col, h->offsets.p, nclouds, h->poses.p, h->dir.slot_ids, h->rec.p, h->rec_c.p,
                     h->heads.p, h->updated.p, h->d_ctr);
  PLVS_KERNEL_CHECK();
  hipLaunchKernelGGL(vb_chain_chunks, dim3(ceil_div(V, kChainChunk)), dim3(kChainThreads), 0, s, h->P, keys, V, h->rec.p,
                     h->rec_c.p, h->d_ctr, h->dist, h->weight, h->rgba);
  PLVS_KERNEL_CHECK();
  rc = vb_read_counters(h, s);
  if (rc != PLVS_OK) return rc;
  if (h->h_ctr->err) {
    h->poisoned = true;
    plvs::set_error("tsdf_voxblox shard_apply: internal directory miss (err=%u)", h->h_ctr->err);
    return PLVS_ERR_CAPACITY;
  }
  h->stats.updated_chunks = (int32_t)h->h_ctr->num_updated;
  h->stats.voxels = (int32_t)h->h_ctr->num_heads;
  h->stats.max_run = (int32_t)h->h_ctr->max_run;
  h->last_updated = h->h_ctr->num_updated;
  return vb_publish_waiting(h, s, published_lo, published_hi);
}

int plvs_hip_tsdf_voxblox_integrate_fast_batch_dev(plvs_tsdf_voxblox* h, const float* d_xyz, const uint8_t* d_rgba,
                                                   const int32_t* offsets, int nclouds, const float* d_Twc, void* stream) {
  PLVS_REQUIRE(h, "null handle");
  VB_FLUSH_QUEUE(h);
  PLVS_REQUIRE(offsets && nclouds >= 0, "bad offsets");
  hipStream_t s = static_cast<hipStream_t>(stream);
  // Every cloud is a scan, every scan begins with the sets' "reset": offset + 1, and the 10 000th zeroes them
  // (approx_hash_array.h:141-150).  A batch is cut where that happens.
  std::vector<int32_t> sub;
  int c0 = 0;
  int rounds = 0;
  plvs_tsdf_stats total{};
  while (c0 < nclouds) {
    const bool zero = h->ap_next >= kApproxReset;
    int rc = vb_fast_tables(h, s, zero);
    if (rc != PLVS_OK) return rc;
    if (zero) h->ap_next = 0;
    const int count = std::min<int>(nclouds - c0, (int)(kApproxReset - h->ap_next));
    sub.assign((size_t)count + 1, 0);
    for (int c = 0; c <= count; ++c) sub[(size_t)c] = offsets[c0 + c] - offsets[c0];
    const size_t p0 = (size_t)offsets[c0];
    rc = vb_integrate_impl(h, d_xyz ? d_xyz + 3 * p0 : nullptr, d_rgba ? d_rgba + 4 * p0 : nullptr, sub.data(), count,
                           d_Twc ? d_Twc + 12 * (size_t)c0 : nullptr, stream, kFast, nullptr, nullptr, h->ap_next);
    if (rc != PLVS_OK) return rc;
    h->ap_next += (uint32_t)count;
    c0 += count;
    rounds = std::max(rounds, h->fast_rounds);
    total.points += h->stats.points; total.visits += h->stats.visits; total.new_chunks += h->stats.new_chunks;
    total.updated_chunks = h->stats.updated_chunks; total.voxels = h->stats.voxels; total.max_run = h->stats.max_run;
  }
  if (nclouds > 0) h->stats = total;
  h->fast_rounds = rounds;
  return PLVS_OK;
}

int plvs_hip_tsdf_voxblox_integrate_fast(plvs_tsdf_voxblox* h, const float* xyz, const uint8_t* rgba, int n, const float* Twc) {
  PLVS_REQUIRE(h, "null handle");
  VB_FLUSH_QUEUE(h);
  PLVS_REQUIRE(n >= 0 && Twc, "bad arguments");
  if (n == 0) {   // (an empty scan still moves the sets on)
    const int32_t none[2] = {0, 0};
    return plvs_hip_tsdf_voxblox_integrate_fast_batch_dev(h, nullptr, nullptr, none, 1, nullptr, nullptr);
  }
  PLVS_REQUIRE(xyz && rgba, "null cloud pointer");
  PLVS_HIP_TRY(h->st_xyz.reserve((size_t)n * 3));
  PLVS_HIP_TRY(h->st_rgba.reserve((size_t)n));
  PLVS_HIP_TRY(h->st_Twc.reserve(12));
  PLVS_HIP_TRY(hipMemcpy(h->st_xyz.p, xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
  PLVS_HIP_TRY(hipMemcpy(h->st_rgba.p, rgba, (size_t)n * 4, hipMemcpyHostToDevice));
  PLVS_HIP_TRY(hipMemcpy(h->st_Twc.p, Twc, 12 * sizeof(float), hipMemcpyHostToDevice));
  const int32_t offsets[2] = {0, n};
  int rc = plvs_hip_tsdf_voxblox_integrate_fast_batch_dev(h, h->st_xyz.p, reinterpret_cast<const uint8_t*>(h->st_rgba.p), offsets, 1,
                                                        h->st_Twc.p, nullptr);
  if (rc != PLVS_OK) return rc;
  PLVS_HIP_TRY(hipDeviceSynchronize());
  return PLVS_OK;
}

int plvs_hip_tsdf_voxblox_fast_rounds(plvs_tsdf_voxblox* h, int* rounds) {
  PLVS_REQUIRE(h && rounds, "null argument");
  *rounds = h->fast_rounds;
  return PLVS_OK;
}

int plvs_hip_tsdf_voxblox_integrate_world_normals(plvs_tsdf_voxblox* h, const float* xyz, const uint8_t* rgba,
                                                  const float* normals, int n, const float* Twc) {
  PLVS_REQUIRE(h, "null handle");
  VB_FLUSH_QUEUE(h);
  PLVS_REQUIRE(n >= 0 && Twc, "bad arguments");
  if (n == 0) {
    h->stats = plvs_tsdf_stats{};
    h->last_updated = 0;
    return PLVS_OK;
  }
  PLVS_REQUIRE(xyz && rgba && normals, "null cloud pointer");
  PLVS_HIP_TRY(h->st_xyz.reserve((size_t)n * 3));
  PLVS_HIP_TRY(h->st_rgba.reserve((size_t)n));
  PLVS_HIP_TRY(h->st_Twc.reserve(12));
  PLVS_HIP_TRY(h->st_nrm.reserve((size_t)n * 3));
  PLVS_HIP_TRY(hipMemcpy(h->st_xyz.p, xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
  PLVS_HIP_TRY(hipMemcpy(h->st_rgba.p, rgba, (size_t)n * 4, hipMemcpyHostToDevice));
  PLVS_HIP_TRY(hipMemcpy(h->st_nrm.p, normals, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
  PLVS_HIP_TRY(hipMemcpy(h->st_Twc.p, Twc, 12 * sizeof(float), hipMemcpyHostToDevice));
  const int32_t offsets[2] = {0, n};
  const int rc = vb_integrate_impl(h, h->st_xyz.p, reinterpret_cast<const uint8_t*>(h->st_rgba.p), offsets, 1, h->st_Twc.p,
                                   nullptr, kWorld, h->st_nrm.p, nullptr);
  (void)hipDeviceSynchronize();
  return rc;
}

int plvs_hip_tsdf_voxblox_integrate_merged(plvs_tsdf_voxblox* h, const float* xyz, const uint8_t* rgba, int n,
                                           const float* Twc) {
  PLVS_REQUIRE(h, "null handle");
  VB_FLUSH_QUEUE(h);
  PLVS_REQUIRE(n >= 0 && Twc, "bad arguments");
  PLVS_REQUIRE(!h->poisoned, "handle is in a failed state (clear it)");
  h->stats = plvs_tsdf_stats{};
  h->last_updated = 0;
  if (n == 0) {   // (an empty cloud still publishes what a world cloud left waiting, as in the simple flavour)
    const int32_t none[2] = {0, 0};
    return plvs_hip_tsdf_voxblox_integrate_batch_dev(h, nullptr, nullptr, none, 1, nullptr, nullptr);
  }
  PLVS_REQUIRE(xyz && rgba, "null cloud pointer");
  hipStream_t s = nullptr;
  PLVS_HIP_TRY(h->st_xyz.reserve((size_t)n * 3));
  PLVS_HIP_TRY(h->st_rgba.reserve((size_t)n));
  PLVS_HIP_TRY(h->st_Twc.reserve(12));
  PLVS_HIP_TRY(h->mg_kind.reserve((size_t)n));
  PLVS_HIP_TRY(h->mg_g.reserve((size_t)n * 3));
  PLVS_HIP_TRY(hipMemcpy(h->st_xyz.p, xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
  PLVS_HIP_TRY(hipMemcpy(h->st_rgba.p, rgba, (size_t)n * 4, hipMemcpyHostToDevice));
  PLVS_HIP_TRY(hipMemcpy(h->st_Twc.p, Twc, 12 * sizeof(float), hipMemcpyHostToDevice));
  PLVS_HIP_TRY(hipMemsetAsync(&h->d_ctr->err, 0, sizeof(uint32_t), s));
  PLVS_HIP_TRY(h->poses.reserve(1));
  hipLaunchKernelGGL(vb_pose_prep, dim3(1), dim3(64), 0, s, h->st_Twc.p, 1, h->poses.p);
  hipLaunchKernelGGL(vb_merge_keys, dim3(ceil_div((size_t)n, 256)), dim3(256), 0, s, h->P, h->st_xyz.p, n, h->poses.p, h->d_ctr,
                     h->mg_kind.p, h->mg_g.p);
  PLVS_KERNEL_CHECK();
  std::vector<uint8_t> kind((size_t)n);
  std::vector<int32_t> g((size_t)n * 3);
  PLVS_HIP_TRY(hipMemcpy(kind.data(), h->mg_kind.p, (size_t)n, hipMemcpyDeviceToHost));
  PLVS_HIP_TRY(hipMemcpy(g.data(), h->mg_g.p, (size_t)n * 3 * sizeof(int32_t), hipMemcpyDeviceToHost));
  int rc = vb_read_counters(h, s);
  if (rc != PLVS_OK) return rc;
  if (h->h_ctr->err & kErrNonFinite) {
    plvs::set_error("tsdf_voxblox integrate_merged: non-finite point in the cloud");
    return PLVS_ERR_INVALID_ARG;
  }
  // bundleRays (tsdf_integrator.cc:361-386): points in the mixed visiting order into voxel_map / clear_map.  The
  // integration order of the bundles is the iteration order of those maps (integrateVoxels, :448-470: begin(), ++it) —
  // AnyIndexHashMapType = std::unordered_map with AnyIndexHash (core/block_hash.h:15-34) — so the host fills the
  // same container with the same hash in the same sequence and walks it; nothing is computed here but that order.
  struct Key {
    int32_t v[3];
    bool operator==(const Key& o) const { return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2]; }
  };
  struct KeyHash {
    std::size_t operator()(const Key& k) const {
      return (static_cast<unsigned int>(k.v[0]) * std::size_t(73856093) ^ k.v[1] * std::size_t(19349663) ^
              k.v[2] * std::size_t(83492791));
    }
  };
  using BundleMap = std::unordered_map<Key, std::vector<uint32_t>, KeyHash>;
  BundleMap voxel_map, clear_map;
  for (uint32_t sq = 0; sq < (uint32_t)n; ++sq) {
    const uint32_t p = mixed_index(sq, (uint32_t)n);
    if (kind[p] == 0) continue;
    const Key k{{g[3 * (size_t)p], g[3 * (size_t)p + 1], g[3 * (size_t)p + 2]}};
    (kind[p] == 2 ? clear_map : voxel_map)[k].push_back(p);
  }
  const size_t nb = voxel_map.size() + clear_map.size();
  h->stats.points = n;
  if (nb == 0) {   // (every point skipped: nothing to cast, but the call still publishes waiting world-cloud blocks)
    const int32_t none[2] = {0, 0};
    const int rc0 = plvs_hip_tsdf_voxblox_integrate_batch_dev(h, nullptr, nullptr, none, 1, nullptr, nullptr);
    h->stats.points = n;
    return rc0;
  }
  std::vector<uint32_t> first(nb + 1), pts;
  std::vector<uint8_t> clr(nb);
  pts.reserve((size_t)n);
  size_t b = 0;
  for (int pass = 0; pass < 2; ++pass)
    for (const auto& kv : (pass ? clear_map : voxel_map)) {
      first[b] = (uint32_t)pts.size();
      clr[b] = (uint8_t)pass;
      pts.insert(pts.end(), kv.second.begin(), kv.second.end());
      ++b;
    }
  first[nb] = (uint32_t)pts.size();
  PLVS_HIP_TRY(h->mg_first.reserve(nb + 1));
  PLVS_HIP_TRY(h->mg_pts.reserve(pts.size()));
  PLVS_HIP_TRY(h->mg_clr.reserve(nb));
  PLVS_HIP_TRY(h->mg_xyz.reserve(3 * nb));
  PLVS_HIP_TRY(h->mg_col.reserve(nb));
  PLVS_HIP_TRY(h->mg_w.reserve(nb));
  PLVS_HIP_TRY(hipMemcpy(h->mg_first.p, first.data(), (nb + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
  PLVS_HIP_TRY(hipMemcpy(h->mg_pts.p, pts.data(), pts.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  PLVS_HIP_TRY(hipMemcpy(h->mg_clr.p, clr.data(), nb, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(vb_merge_bundles, dim3(ceil_div(nb, 256)), dim3(256), 0, s, h->st_xyz.p, h->st_rgba.p, h->mg_first.p,
                     h->mg_pts.p, h->mg_clr.p, (int)nb, h->mg_xyz.p, h->mg_col.p, h->mg_w.p);
  PLVS_KERNEL_CHECK();
  const int32_t offsets[2] = {0, (int32_t)nb};
  rc = vb_integrate_impl(h, h->mg_xyz.p, reinterpret_cast<const uint8_t*>(h->mg_col.p), offsets, 1, h->st_Twc.p, nullptr, kMerged,
                         h->mg_w.p, h->mg_clr.p);
  (void)hipDeviceSynchronize();
  h->stats.points = n;
  return rc;
}

// Queued insertion, as for the chisel map (plvs_hip_tsdf_chisel_queue): PLVS inserts one key frame per InsertCloud and reads
// the layer only in UpdateMap (src/PointCloudMapping.cc:537-556, 594-598; PointCloudMapVoxblox::UpdateMap :160-179).  _queue
// uploads and returns, _flush integrates what waits as ONE batch of the simple integrator — every cloud a scan of its own,
// in order: the layer of the call-by-call sequence, bit for bit.  Every entry point that reads or changes the map flushes
// first; clear drops the queue.
int plvs_hip_tsdf_voxblox_queue(plvs_tsdf_voxblox* h, const float* xyz, const uint8_t* rgba, int n, const float* Twc) {
  PLVS_REQUIRE(h, "null handle");
  PLVS_REQUIRE(n >= 0 && Twc, "bad arguments");
  PLVS_REQUIRE(n == 0 || (xyz && rgba), "null cloud pointer");
  const bool first = h->q_offsets.empty();
  const size_t at = first ? 0 : (size_t)h->q_offsets.back();
  PLVS_REQUIRE(at + (size_t)n < 0x7FFFFFFFull, "too many queued points");
  if (n) {
    PLVS_HIP_TRY(vb_grow_keep(h->q_xyz, 3 * at, 3 * (at + (size_t)n)));
    PLVS_HIP_TRY(vb_grow_keep(h->q_rgba, at, at + (size_t)n));
    PLVS_HIP_TRY(hipMemcpy(h->q_xyz.p + 3 * at, xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
    PLVS_HIP_TRY(hipMemcpy(h->q_rgba.p + at, rgba, (size_t)n * 4, hipMemcpyHostToDevice));
  }
  if (first) h->q_offsets.push_back(0);
  h->q_offsets.push_back((int32_t)(at + (size_t)n));   // (an empty cloud is a scan too: it publishes waiting blocks)
  h->q_Twc.insert(h->q_Twc.end(), Twc, Twc + 12);
  return PLVS_OK;
}

int plvs_hip_tsdf_voxblox_queued(plvs_tsdf_voxblox* h, int* nclouds) {
  PLVS_REQUIRE(h && nclouds, "null argument");
  *nclouds = h->q_offsets.empty() ? 0 : (int)h->q_offsets.size() - 1;
  return PLVS_OK;
}

int plvs_hip_tsdf_voxblox_flush(plvs_tsdf_voxblox* h) {
  PLVS_REQUIRE(h, "null handle");
  if (h->q_offsets.empty()) return PLVS_OK;
  const int nclouds = (int)h->q_offsets.size() - 1;
  std::vector<int32_t> offsets;
  std::vector<float> Twc;
  offsets.swap(h->q_offsets);   // (the queue is empty from here on: the integrate below flushes nothing)
  Twc.swap(h->q_Twc);
  PLVS_HIP_TRY(h->q_Twc_dev.reserve((size_t)12 * nclouds));
  PLVS_HIP_TRY(hipMemcpy(h->q_Twc_dev.p, Twc.data(), (size_t)12 * nclouds * sizeof(float), hipMemcpyHostToDevice));
  int rc = plvs_hip_tsdf_voxblox_integrate_batch_dev(h, h->q_xyz.p, reinterpret_cast<const uint8_t*>(h->q_rgba.p), offsets.data(), nclouds,
                                                     h->q_Twc_dev.p, nullptr);
  if (rc != PLVS_OK) {
    // The queue was taken before the batch ran (a reader that flushes must not flush again from inside it): its clouds are
    // gone.  Say so, and how many — the error surfaces from whichever call triggered the flush, possibly a reader.
    char own[400];
    snprintf(own, sizeof own, "%s", plvs::last_error_buf());
    plvs::set_error("%s — raised by the flush of %d queued key-frame cloud%s (plvs_hip_tsdf_voxblox_queue): NONE of them was "
                    "integrated and they are dropped; queue them again after clearing / enlarging the map", own, nclouds,
                    nclouds == 1 ? "" : "s");
    return rc;
  }
  PLVS_HIP_TRY(hipDeviceSynchronize());
  return PLVS_OK;
}

int plvs_hip_tsdf_voxblox_integrate(plvs_tsdf_voxblox* h, const float* xyz, const uint8_t* rgba,
                                    int n, const float* Twc) {
  PLVS_REQUIRE(h, "null handle");
  VB_FLUSH_QUEUE(h);
  PLVS_REQUIRE(n >= 0 && Twc, "bad arguments");
  if (n == 0) {   // (an empty cloud still publishes what a world cloud left waiting)
    const int32_t none[2] = {0, 0};
    return plvs_hip_tsdf_voxblox_integrate_batch_dev(h, nullptr, nullptr, none, 1, nullptr, nullptr);
  }
  PLVS_REQUIRE(xyz && rgba, "null cloud pointer");
  PLVS_HIP_TRY(h->st_xyz.reserve((size_t)n * 3));
  PLVS_HIP_TRY(h->st_rgba.reserve((size_t)n));
  PLVS_HIP_TRY(h->st_Twc.reserve(12));
  PLVS_HIP_TRY(hipMemcpy(h->st_xyz.p, xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
  PLVS_HIP_TRY(hipMemcpy(h->st_rgba.p, rgba, (size_t)n * 4, hipMemcpyHostToDevice));
  PLVS_HIP_TRY(hipMemcpy(h->st_Twc.p, Twc, 12 * sizeof(float), hipMemcpyHostToDevice));
  const int32_t offsets[2] = {0, n};
  int rc = plvs_hip_tsdf_voxblox_integrate_batch_dev(
      h, h->st_xyz.p, reinterpret_cast<const uint8_t*>(h->st_rgba.p), offsets, 1, h->st_Twc.p, nullptr);
  if (rc != PLVS_OK) return rc;
  PLVS_HIP_TRY(hipDeviceSynchronize());
  return PLVS_OK;
}

int plvs_hip_tsdf_voxblox_set_deferred_world_blocks(plvs_tsdf_voxblox* h, int enable) {
  PLVS_REQUIRE(h, "null handle");
  VB_FLUSH_QUEUE(h);
  h->defer_world_blocks = enable != 0;
  if (!h->defer_world_blocks) h->visible_blocks = h->num_blocks;
  return PLVS_OK;
}

int plvs_hip_tsdf_voxblox_last_stats(plvs_tsdf_voxblox* h, plvs_tsdf_stats* s) {
  PLVS_REQUIRE(h && s, "null argument");
  VB_FLUSH_QUEUE(h);
  *s = h->stats;
  return PLVS_OK;
}

int plvs_hip_tsdf_voxblox_num_blocks(plvs_tsdf_voxblox* h, int* n) {
  PLVS_REQUIRE(h && n, "null argument");
  VB_FLUSH_QUEUE(h);
  *n = h->visible_blocks;
  return PLVS_OK;
}

int plvs_hip_tsdf_voxblox_block_ids(plvs_tsdf_voxblox* h, int32_t* ids_xyz, int cap, int* n) {
  PLVS_REQUIRE(h && n, "null argument");
  VB_FLUSH_QUEUE(h);
  *n = h->visible_blocks;
  const int m = h->visible_blocks < cap ? h->visible_blocks : cap;
  if (m > 0) {
    PLVS_REQUIRE(ids_xyz, "null output");
    PLVS_HIP_TRY(hipMemcpy(ids_xyz, h->dir.slot_ids, (size_t)m * 3 * sizeof(int32_t), hipMemcpyDeviceToHost));
  }
  return PLVS_OK;
}

int plvs_hip_tsdf_voxblox_updated_block_ids_dev(plvs_tsdf_voxblox* h, int32_t* d_ids_xyz, int cap,
                                                int* n, void* stream) {
  PLVS_REQUIRE(h && n, "null argument");
  VB_FLUSH_QUEUE(h);
  *n = (int)h->last_updated;
  const int m = (int)h->last_updated < cap ? (int)h->last_updated : cap;
  if (m <= 0) return PLVS_OK;
  PLVS_REQUIRE(d_ids_xyz, "null output");
  hipLaunchKernelGGL(vb_gather_slot_ids, dim3(ceil_div((size_t)m, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), h->updated.p, m, h->dir.slot_ids, d_ids_xyz);
  PLVS_KERNEL_CHECK();
  return PLVS_OK;
}

int plvs_hip_tsdf_voxblox_updated_block_ids(plvs_tsdf_voxblox* h, int32_t* ids_xyz, int cap, int* n) {
  PLVS_REQUIRE(h && n, "null argument");
  VB_FLUSH_QUEUE(h);
  *n = (int)h->last_updated;
  const int m = (int)h->last_updated < cap ? (int)h->last_updated : cap;
  if (m <= 0) return PLVS_OK;
  PLVS_REQUIRE(ids_xyz, "null output");
  // small lists: resolve slot -> id on the host
  std::vector<uint32_t> slots(h->last_updated);
  std::vector<int32_t> all((size_t)h->num_blocks * 3);
  PLVS_HIP_TRY(hipMemcpy(slots.data(), h->updated.p, slots.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  PLVS_HIP_TRY(hipMemcpy(all.data(), h->dir.slot_ids, all.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  for (int i = 0; i < m; ++i) memcpy(ids_xyz + 3 * i, all.data() + 3 * (size_t)slots[(size_t)i], 3 * sizeof(int32_t));
  return PLVS_OK;
}

int plvs_hip_tsdf_voxblox_download_block(plvs_tsdf_voxblox* h, int bx, int by, int bz,
                                         float* distance, float* weight, uint32_t* rgba) {
  PLVS_REQUIRE(h && distance && weight && rgba, "null argument");
  VB_FLUSH_QUEUE(h);
  int32_t* all = new int32_t[(size_t)(h->num_blocks > 0 ? h->num_blocks : 1) * 3];
  hipError_t e = hipSuccess;
  if (h->num_blocks > 0)
    e = hipMemcpy(all, h->dir.slot_ids, (size_t)h->num_blocks * 3 * sizeof(int32_t), hipMemcpyDeviceToHost);
  int slot = -1;
  if (e == hipSuccess)
    for (int i = 0; i < h->visible_blocks; ++i)
      if (all[3 * i] == bx && all[3 * i + 1] == by && all[3 * i + 2] == bz) { slot = i; break; }
  delete[] all;
  PLVS_HIP_TRY(e);
  if (slot < 0) {
    plvs::set_error("block (%d,%d,%d) does not exist", bx, by, bz);
    return PLVS_ERR_INVALID_ARG;
  }
  const size_t off = (size_t)slot * kBlockVox;
  PLVS_HIP_TRY(hipMemcpy(distance, h->dist + off, kBlockVox * sizeof(float), hipMemcpyDeviceToHost));
  PLVS_HIP_TRY(hipMemcpy(weight, h->weight + off, kBlockVox * sizeof(float), hipMemcpyDeviceToHost));
  PLVS_HIP_TRY(hipMemcpy(rgba, h->rgba + off, kBlockVox * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return PLVS_OK;
}

}  // extern "C"

#if PLVS_VB_PROF
extern "C" int plvs_hip_debug_chain_prof(unsigned long long* out, int nwaves) {
  PLVS_HIP_TRY(hipDeviceSynchronize());
  PLVS_HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chain_prof), (size_t)nwaves * 4 * sizeof(unsigned long long)));
  return PLVS_OK;
}
#endif
