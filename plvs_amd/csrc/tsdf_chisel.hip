// TSDF integrate for the open_chisel back end (PointCloudMapChisel::InsertCloud
// -> Chisel::IntegratePointCloudWidthDepth, point-cloud part).
//
// Data layout in HBM (per handle):
//   voxel pool      four planes sdf / weight / kfid / rgbw, each
//                   max_chunks * 4096 dwords; chunk slot s owns words
//                   [s*4096, (s+1)*4096) of every plane (64 KiB per chunk).
//   chunk directory open-addressing hash (2*max_chunks, power of two) from the
//                   packed 3x21-bit chunk id to the pool slot, plus slot -> id.
//   per call        one u32 (slot*4096 + voxel) key and one u32 point index per
//                   voxel visit.
//
// The reference integrates points strictly in order, and both the running
// weighted mean (f32) and the truncating u8 colour mean are order dependent.
// The device path keeps that order exactly:
//   1. ray_count   one thread per point walks its Amanatides-Woo ray and counts
//                  the voxels that take an update; first-touch chunks are
//                  inserted into the directory.
//   2. scan        exclusive scan of the counts = visit offsets (point order).
//   3. ray_fill    the same walk writes (voxel key, point index) records at
//                  those offsets, i.e. globally sorted by point.
//   4. radix sort  stable by voxel key -> per voxel, records in point order.
//   5. expand      one thread per sorted record derives the update operands
//                  (w_u*u, w_u) from the point — everything that is not order
//                  dependent — and compacts the run heads.
//   6. chain       one thread per voxel run folds its records sequentially in
//                  registers (the f32 weighted mean and the truncating u8
//                  colour mean): each voxel is read and written once per call.
// Results are bit-identical to the sequential CPU loop.
#include "common.hpp"
#include "device_utils.hpp"
#include "tsdf_chisel_core.hpp"
#include "tsdf_directory.hpp"

using namespace plvs;
using namespace plvs::chisel;
using namespace plvs::tsdf;

namespace {

constexpr int kNumStages = 6;
const char* const kStageNames[kNumStages] = {"ray_count", "scan", "ray_fill", "radix_sort",
                                             "expand_records", "chain_runs"};

#ifndef PLVS_CHAIN_PROBE
#define PLVS_CHAIN_PROBE 0
#endif

struct Counters {           // device-side, read back once per call
  uint32_t total_visits;
  int32_t num_chunks;
  uint32_t err;
  uint32_t num_heads;
  uint32_t num_updated;
  uint32_t max_run;
#if PLVS_CHAIN_PROBE
  uint32_t pad_;
  unsigned long long probe[6];   // instrumentation build only
#endif
};

__global__ void pose_prep(const float* __restrict__ Twc, int nclouds, Pose* __restrict__ poses) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < nclouds) make_pose(Twc + 12 * c, &poses[c]);
}

__global__ void pool_init(float* __restrict__ sdf, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t j = i; j < n; j += stride) sdf[j] = 99999.0f;
}

// Pass 1 (kFill == false): count the updating visits of each point and insert
// first-touch chunks.  Pass 2 (kFill == true): write the visit records.
template <bool kFill>
__global__ __launch_bounds__(256) void ray_pass(
    Params P, const float* __restrict__ xyz, int npoints, const int32_t* __restrict__ offsets,
    int nclouds, const Pose* __restrict__ poses, Directory dir, Counters* __restrict__ ctr,
    uint32_t* __restrict__ counts /* pass 1 out, pass 2: scanned offsets */,
    uint32_t* __restrict__ rec_keys, uint32_t* __restrict__ rec_pts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npoints) return;
  const Pose pose = poses[cloud_of(offsets, nclouds, i)];
  Ray ray;
  uint32_t n = 0;
  if (make_ray(P, pose, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], &ray)) {
    const uint32_t out = kFill ? counts[i] : 0u;
    RayCursor cur;
    ray_begin(ray, &cur);
    int vx, vy, vz;
    int lcx = 0, lcy = 0, lcz = 0, lslot = -1;  // last chunk seen by this ray
    bool have_last = false;
    while (ray_next(&cur, &vx, &vy, &vz)) {
      Visit v;
      if (!resolve_visit(P, pose, ray, vx, vy, vz, &v)) continue;
      if (!have_last || v.cx != lcx || v.cy != lcy || v.cz != lcz) {
        lcx = v.cx; lcy = v.cy; lcz = v.cz;
        have_last = true;
        if (kFill) {
          lslot = dir_find(dir, lcx, lcy, lcz);
          if (lslot < 0) atomicOr(&ctr->err, kErrDirectoryMiss);
        } else {
          dir_insert(dir, lcx, lcy, lcz, &ctr->num_chunks, &ctr->err);
          // the float chunk lookup (GetIDAt) and the integer voxel grid only
          // disagree ~100 km from the origin; fail loudly instead of diverging
          if (((vx - lcx * 16) | (vy - lcy * 16) | (vz - lcz * 16)) & ~15)
            atomicOr(&ctr->err, kErrCoordRange);
        }
      }
      if (kFill && lslot >= 0) {
        rec_keys[out + n] = (uint32_t)lslot * (uint32_t)kChunkVox + (uint32_t)v.vid;
        rec_pts[out + n] = (uint32_t)i;
      }
      ++n;
    }
  }
  if (!kFill) counts[i] = n;
}

// Per sorted record: the update operands the sequential chain needs, computed
// fully in parallel, plus the compaction of run heads / updated chunks.
//   rec[r].x = w_u * u        the rounded product DistVoxel::Integrate adds
//   rec[r].y = +-w_u          ConstantWeighter weight of the record's point;
//                             negative on the LAST record of a voxel run
//   rec_c[r] = r | g<<8 | b<<16 of the point (after the reference's u8->f32->u8 trip)
constexpr int kExpandThreads = 1024;
__global__ __launch_bounds__(kExpandThreads) void expand_records(
    Params P, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ pts, uint32_t n,
    const float* __restrict__ xyz, const uint8_t* __restrict__ rgb,
    const int32_t* __restrict__ offsets, int nclouds, const Pose* __restrict__ poses,
    const int32_t* __restrict__ slot_ids, float2* __restrict__ rec, uint32_t* __restrict__ rec_c,
    uint32_t* __restrict__ heads, uint32_t* __restrict__ updated_slots,
    Counters* __restrict__ ctr, const uint32_t* __restrict__ kfid, uint32_t* __restrict__ vkfid) {
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
    const int p = (int)pts[r];
    const uint32_t slot = key >> 12, vid = key & 4095u;
    const int vx = slot_ids[3 * slot + 0] * 16 + (int)(vid & 15u);
    const int vy = slot_ids[3 * slot + 1] * 16 + (int)((vid >> 4) & 15u);
    const int vz = slot_ids[3 * slot + 2] * 16 + (int)(vid >> 8);
    const float c0 = (float)vx * P.resolution + P.half_voxel;
    const float c1 = (float)vy * P.resolution + P.half_voxel;
    const float c2 = (float)vz * P.resolution + P.half_voxel;
    const Pose& pose = poses[cloud_of(offsets, nclouds, p)];
    const float depth = xyz[3 * (size_t)p + 2];
    const float tr = truncation_of(P, depth);
    const float u = signed_dist(pose, depth, c0, c1, c2);
    const float wu = P.weight / (2.0f * tr);
    rec[r] = make_float2(wu * u, (key != next) ? -wu : wu);
    if (key != next) vkfid[key] = kfid ? kfid[p] : 0u;   // SetKfid: the last update of the run wins
    rec_c[r] = colour_roundtrip(rgb[3 * (size_t)p + 0]) | (colour_roundtrip(rgb[3 * (size_t)p + 1]) << 8) |
               (colour_roundtrip(rgb[3 * (size_t)p + 2]) << 16);
  }
  // block-aggregated, order-free compaction of the two head lists
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
//    Nothing is loaded there: the keyframe id of the last record is written by
//    expand_records, the longest run is reduced once per wave.
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
    const uint32_t* __restrict__ keys, uint32_t nrec, const float2* __restrict__ rec,
    const uint32_t* __restrict__ heads, Counters* __restrict__ ctr, float* __restrict__ sdf,
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
    const uint32_t r0 = live ? heads[h] : 0u;
    const size_t a = live ? (size_t)keys[r0] : 0;   // slot*4096 + vid
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

// The truncating u8 colour mean of ColorVoxel::IntegrateSimple: order
// dependent, but frozen for good once the colour weight reaches 254, so a run
// contributes at most (254 - weight) steps.  One thread per voxel run.
__global__ __launch_bounds__(256) void chain_colours(
    const uint32_t* __restrict__ keys, uint32_t nrec, const float2* __restrict__ rec,
    const uint32_t* __restrict__ rec_c, const uint32_t* __restrict__ heads,
    const Counters* __restrict__ ctr, uint32_t* __restrict__ rgbw) {
  const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= ctr->num_heads) return;
  uint32_t r = heads[h];
  const size_t a = (size_t)keys[r];
  uint32_t col = rgbw[a];
  if ((col >> 24) >= 254u) return;
  for (;;) {
    uint32_t c[4];
    float y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t rr = (r + j < nrec) ? r + j : nrec - 1;
      c[j] = rec_c[rr];
      y[j] = rec[rr].y;
    }
    bool done = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!done) {
        colour_update(col, c[j] & 255u, (c[j] >> 8) & 255u, (c[j] >> 16) & 255u);
        done = (y[j] < 0.0f) || ((col >> 24) >= 254u);
      }
    }
    if (done) break;
    r += 4;
  }
  rgbw[a] = col;
}

// Hardware assumption of chain_runs, checked exhaustively: rcp_rn(b) == RN(1/b) for every
// significand at the given exponent.
__global__ void selftest_rcp_kernel(int exponent, uint32_t* __restrict__ mismatches) {
  const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
  const float b = __uint_as_float(((uint32_t)(exponent + 127) << 23) | m);
  if (__float_as_uint(rcp_rn(b)) != __float_as_uint(1.0f / b)) atomicAdd(mismatches, 1u);
}

}  // namespace

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
  DevBuf<uint32_t> counts, keys0, keys1, pts0, pts1, heads, updated, scratch;
  DevBuf<float2> rec;
  DevBuf<uint32_t> rec_c;
  DevBuf<Pose> poses;
  DevBuf<int32_t> offsets;
  // host-flavour staging
  DevBuf<float> st_xyz, st_Twc;
  DevBuf<uint8_t> st_rgb;
  DevBuf<uint32_t> st_kfid;
  plvs_tsdf_stats stats{};
  uint32_t last_updated = 0;
  hipStream_t side = nullptr;   // second stream for the colour chain
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // optional per-stage timing (HIP events on the caller's stream)
  bool profiling = false;
  hipEvent_t ev[kNumStages + 1] = {};
  double stage_ms[kNumStages] = {};
  int64_t prof_calls = 0;
};

static int read_counters(plvs_tsdf_chisel* h, hipStream_t s) {
  PLVS_HIP_TRY(hipMemcpyAsync(h->h_ctr, h->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipStreamSynchronize(s));
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
  CREATE_TRY(hipHostMalloc((void**)&h->h_ctr, sizeof(Counters)));
  CREATE_TRY(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
  CREATE_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
  CREATE_TRY(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
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
  (void)hipFree(h->dir.keys);
  (void)hipFree(h->dir.slots);
  (void)hipFree(h->dir.slot_ids);
  (void)hipFree(h->sdf);
  (void)hipFree(h->weight);
  (void)hipFree(h->kfid);
  (void)hipFree(h->rgbw);
  (void)hipFree(h->d_ctr);
  if (h->h_ctr) (void)hipHostFree(h->h_ctr);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->side) (void)hipStreamDestroy(h->side);
  for (int i = 0; i <= kNumStages; ++i)
    if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
  h->counts.release(); h->keys0.release(); h->keys1.release(); h->pts0.release(); h->pts1.release();
  h->rec.release(); h->rec_c.release();
  h->heads.release(); h->updated.release(); h->scratch.release(); h->poses.release();
  h->offsets.release(); h->st_xyz.release(); h->st_Twc.release(); h->st_rgb.release();
  h->st_kfid.release();
  delete h;
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_clear(plvs_tsdf_chisel* h) {
  PLVS_REQUIRE(h, "null handle");
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
  h->poisoned = false;
  h->stats = plvs_tsdf_stats{};
  h->last_updated = 0;
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_integrate_batch_dev(plvs_tsdf_chisel* h, const float* d_xyz,
                                             const uint8_t* d_rgb, const uint32_t* d_kfid,
                                             const int32_t* offsets, int nclouds,
                                             const float* d_Twc, void* stream) {
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

  PLVS_HIP_TRY(h->offsets.reserve((size_t)nclouds + 1));
  PLVS_HIP_TRY(h->poses.reserve((size_t)nclouds));
  PLVS_HIP_TRY(h->counts.reserve((size_t)n));
  PLVS_HIP_TRY(h->scratch.reserve(scan_scratch_words((size_t)n)));
  PLVS_HIP_TRY(hipMemcpyAsync(h->offsets.p, offsets, ((size_t)nclouds + 1) * sizeof(int32_t),
                              hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(pose_prep, dim3(ceil_div((size_t)nclouds, 64)), dim3(64), 0, s, d_Twc, nclouds,
                     h->poses.p);
  // reset the per-call counters, keep num_chunks
  PLVS_HIP_TRY(hipMemsetAsync(&h->d_ctr->total_visits, 0, sizeof(uint32_t), s));
  PLVS_HIP_TRY(hipMemsetAsync(&h->d_ctr->err, 0, 4 * sizeof(uint32_t), s));

#define STAGE_MARK(i) \
  do { if (h->profiling) PLVS_HIP_TRY(hipEventRecord(h->ev[i], s)); } while (0)
  const dim3 rgrid(ceil_div((size_t)n, 256)), rblock(256);
  STAGE_MARK(0);
  hipLaunchKernelGGL(ray_pass<false>, rgrid, rblock, 0, s, h->P, d_xyz, n, h->offsets.p, nclouds,
                     h->poses.p, h->dir, h->d_ctr, h->counts.p, (uint32_t*)nullptr,
                     (uint32_t*)nullptr);
  PLVS_KERNEL_CHECK();
  STAGE_MARK(1);
  PLVS_HIP_TRY(exclusive_scan_u32(h->counts.p, h->counts.p, (size_t)n, &h->d_ctr->total_visits,
                                  h->scratch.p, s));
  STAGE_MARK(2);
  int rc = read_counters(h, s);
  if (rc != PLVS_OK) return rc;
  if (h->h_ctr->err) {
    h->poisoned = true;
    plvs::set_error("tsdf_chisel integrate: %s%s",
                    (h->h_ctr->err & kErrPoolFull) ? "chunk pool full (raise max_chunks) " : "",
                    (h->h_ctr->err & kErrCoordRange) ? "chunk id outside +-2^20 " : "");
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

  PLVS_HIP_TRY(h->keys0.reserve(V));
  PLVS_HIP_TRY(h->keys1.reserve(V));
  PLVS_HIP_TRY(h->pts0.reserve(V));
  PLVS_HIP_TRY(h->pts1.reserve(V));
  PLVS_HIP_TRY(h->heads.reserve(V));
  PLVS_HIP_TRY(h->rec.reserve((size_t)V + 2));   // chain_runs reads record pairs
  PLVS_HIP_TRY(h->rec_c.reserve(V));
  PLVS_HIP_TRY(h->updated.reserve((size_t)h->num_chunks + 1));
  PLVS_HIP_TRY(h->scratch.reserve(radix_scratch_words(V)));
  float ms_a[2] = {0.f, 0.f};
  if (h->profiling) {  // stages 0,1 are complete (the counter read synchronised)
    PLVS_HIP_TRY(hipEventElapsedTime(&ms_a[0], h->ev[0], h->ev[1]));
    PLVS_HIP_TRY(hipEventElapsedTime(&ms_a[1], h->ev[1], h->ev[2]));
  }
  STAGE_MARK(2);
  hipLaunchKernelGGL(ray_pass<true>, rgrid, rblock, 0, s, h->P, d_xyz, n, h->offsets.p, nclouds,
                     h->poses.p, h->dir, h->d_ctr, h->counts.p, h->keys0.p, h->pts0.p);
  PLVS_KERNEL_CHECK();
  STAGE_MARK(3);
  int key_bits = 12;
  while ((1ll << (key_bits - 12)) < (long long)h->num_chunks) ++key_bits;
  bool second = false;
  PLVS_HIP_TRY(radix_sort_pairs(h->keys0.p, h->pts0.p, h->keys1.p, h->pts1.p, V, 0, key_bits,
                                h->scratch.p, s, &second));
  const uint32_t* keys = second ? h->keys1.p : h->keys0.p;
  const uint32_t* pts = second ? h->pts1.p : h->pts0.p;
  STAGE_MARK(4);
  hipLaunchKernelGGL(expand_records, dim3(ceil_div(V, kExpandThreads)), dim3(kExpandThreads), 0, s,
                     h->P, keys, pts, V, d_xyz, d_rgb, h->offsets.p, nclouds, h->poses.p,
                     h->dir.slot_ids, h->rec.p, h->rec_c.p, h->heads.p, h->updated.p, h->d_ctr, d_kfid,
                     h->kfid);
  PLVS_KERNEL_CHECK();
  STAGE_MARK(5);
  // one thread per voxel run; launched over V (upper bound of the run count),
  // surplus threads exit on the device-side head count.
  // The colour chain touches only rgbw and the distance chain only sdf/weight/kfid; both are
  // latency-bound with few waves, so they run side by side on two streams.
  PLVS_HIP_TRY(hipEventRecord(h->ev_fork, s));
  PLVS_HIP_TRY(hipStreamWaitEvent(h->side, h->ev_fork, 0));
  hipLaunchKernelGGL(chain_colours, dim3(ceil_div(V, 256)), dim3(256), 0, h->side, keys, V, h->rec.p,
                     h->rec_c.p, h->heads.p, h->d_ctr, h->rgbw);
  PLVS_HIP_TRY(hipEventRecord(h->ev_join, h->side));
  // one thread per voxel run; launched over V (upper bound of the run count), surplus
  // waves exit on the device-side head count
  hipLaunchKernelGGL(chain_runs, dim3(std::min<size_t>(ceil_div(V, 64), 16384)), dim3(64), 0, s, keys, V,
                     h->rec.p, h->heads.p, h->d_ctr, h->sdf, h->weight);
  PLVS_HIP_TRY(hipStreamWaitEvent(s, h->ev_join, 0));
  PLVS_KERNEL_CHECK();
  STAGE_MARK(6);
#undef STAGE_MARK
  rc = read_counters(h, s);
  if (rc != PLVS_OK) return rc;
  if (h->profiling) {
    h->stage_ms[0] += ms_a[0];
    h->stage_ms[1] += ms_a[1];
    for (int i = 2; i < kNumStages; ++i) {
      float ms = 0.f;
      PLVS_HIP_TRY(hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]));
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

int plvs_hip_tsdf_chisel_integrate(plvs_tsdf_chisel* h, const float* xyz, const uint8_t* rgb,
                                   const uint32_t* kfid, int n, const float* Twc) {
  PLVS_REQUIRE(h, "null handle");
  PLVS_REQUIRE(n >= 0 && Twc, "bad arguments");
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
  *nstages = kNumStages;
  if (calls) *calls = h->prof_calls;
  for (int i = 0; i < kNumStages && i < cap; ++i) ms[i] = h->stage_ms[i];
  return PLVS_OK;
}

const char* plvs_hip_tsdf_chisel_stage_name(int i) {
  return (i >= 0 && i < kNumStages) ? kStageNames[i] : "";
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
  PLVS_REQUIRE(h && s, "null argument");
  *s = h->stats;
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_num_chunks(plvs_tsdf_chisel* h, int* n) {
  PLVS_REQUIRE(h && n, "null argument");
  *n = h->num_chunks;
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_chunk_ids(plvs_tsdf_chisel* h, int32_t* ids_xyz, int cap, int* n) {
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
