// Ray-sharded multi-GPU integrate of the open_chisel back end (order-free mode; included by tsdf_chisel.hip
// after tsdf_walk.hpp).  New design: the reference is a single process.
//
// The map is sharded by chunk (owner = three-prime ChunkHasher mod N, ChunkManager.h:42-54), the WORK by
// tile of the point stream: rank r walks the tiles t = r (mod N) of every call — its share of the rays,
// whatever chunks they cross — and what it collects for a chunk travels to the chunk's owner:
//
//   shard_walk    walk_tiles over the rank's tiles into the rank's WALK DIRECTORY (every chunk the rank has ever
//                 walked through: ids + one bit per voxel "colour saturated", no voxel data), segment sort;
//                 per chunk its owner and record total (shard_chunk_totals), per destination the runs
//                 (shard_run_count) and the place of every chunk in the send buffers (shard_plan) -> send counts
//   shard_pack    segment descriptors (chunk id instead of slot, record offsets relative to the destination's
//                 block), records and colour runs, grouped by destination, into the caller's send buffers
//   (exchange)    ONE all-to-all of the three buffers — RCCL send/recv behind plvs_hip_tsdf_chisel_integrate_sharded,
//                 torch.distributed in the Python mirror, device copies between virtual ranks in the tests
//   shard_apply   received descriptors -> slots of the owner's directory (shard_translate, first-touch chunks are
//                 inserted here), the segment sort and apply_chunks of the single-device path on the received
//                 records.  The sums are integers: the result is bit-identical to the single-device order-free
//                 integrate whatever N is.
//                 Colours (the truncating u8 mean is order dependent below weight 254): the received runs
//                 (voxel, tile, ray mask) are sorted by (voxel, tile) and folded as on a single device.  A walker
//                 sends a run for every voxel it does not KNOW to be saturated; owners list the voxels that
//                 reach 254 in a call (shard_saturated), the lists are all-gathered and every rank notes them in
//                 its walk directory (shard_note_saturated).  Late knowledge only costs surplus runs (the fold
//                 stops at 254 by itself); colour weights never decrease short of Clear().
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "tsdf_walk.hpp"

namespace {

// Per updated chunk of the call (index a in `active`): its owner, the records of its segments and, per segment,
// the records of the chunk's earlier segments (seg_pre).
__global__ __launch_bounds__(256) void shard_chunk_totals(const uint4* __restrict__ sorted_seg,
                                                          const uint32_t* __restrict__ active,
                                                          const uint32_t* __restrict__ active_off,
                                                          const int32_t* __restrict__ slot_ids, int nranks,
                                                          const WalkCounters* __restrict__ ctr,
                                                          uint32_t* __restrict__ nrec, uint32_t* __restrict__ owner,
                                                          uint32_t* __restrict__ slot_owner, uint32_t* __restrict__ seg_pre) {
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t run;
  if (ctr->err) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t n = ctr->num_updated;
  for (uint32_t a = blockIdx.x; a < n; a += gridDim.x) {
    const uint32_t s0 = active_off[a], s1 = active_off[a + 1];
    if (tid == 0) run = 0;
    __syncthreads();
    for (uint32_t b = s0; b < s1; b += 256) {
      const uint32_t j = b + (uint32_t)tid;
      const uint32_t c = j < s1 ? sorted_seg[2 * (size_t)j].z : 0u;
      uint32_t inc = c;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)inc, off);
        if (lane >= off) inc += up;
      }
      if (lane == 63) wsum[wid] = inc;
      __syncthreads();
      uint32_t wb = run, tot = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        if (w < wid) wb += wsum[w];
        tot += wsum[w];
      }
      if (j < s1) seg_pre[j] = wb + inc - c;
      __syncthreads();
      if (tid == 0) run += tot;
      __syncthreads();
    }
    if (tid == 0) {
      const int32_t* id = slot_ids + 3 * (size_t)active[a];
      nrec[a] = run;
      owner[a] = (uint32_t)shard_of(chunk_hash(id[0], id[1], id[2]), nranks);
      slot_owner[active[a]] = owner[a];
    }
    __syncthreads();
  }
}

// Places of the chunks in the send buffers: destination by destination, chunks in `active` order.
// counts[3p .. 3p+2] = segments, records and runs for rank p; obase[2p], [2p+1] = the first two as running
// offsets, run_base[p] = the third.
__global__ __launch_bounds__(1024) void shard_plan(const uint32_t* __restrict__ active_off, const uint32_t* __restrict__ nrec,
                                                   const uint32_t* __restrict__ owner, int nranks,
                                                   const WalkCounters* __restrict__ ctr, uint32_t* __restrict__ seg_dst,
                                                   uint32_t* __restrict__ rec_dst, long long* __restrict__ counts,
                                                   uint32_t* __restrict__ obase, const uint32_t* __restrict__ run_counts,
                                                   uint32_t* __restrict__ run_base, uint32_t* __restrict__ run_fill) {
  __shared__ uint32_t wsum_s[16], wsum_r[16];
  __shared__ uint32_t carry_s, carry_r;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t n = ctr->err ? 0u : ctr->num_updated;
  if (tid == 0) { carry_s = 0; carry_r = 0; }
  __syncthreads();
  for (int p = 0; p < nranks; ++p) {
    const uint32_t base_s = carry_s, base_r = carry_r;
    __syncthreads();
    if (tid == 0) { obase[2 * p] = base_s; obase[2 * p + 1] = base_r; }
    for (uint32_t b = 0; b < n; b += 1024) {
      const uint32_t a = b + (uint32_t)tid;
      const bool mine = a < n && owner[a] == (uint32_t)p;
      const uint32_t cs = mine ? active_off[a + 1] - active_off[a] : 0u, cr = mine ? nrec[a] : 0u;
      uint32_t is = cs, ir = cr;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t us = (uint32_t)__shfl_up((int)is, off), ur = (uint32_t)__shfl_up((int)ir, off);
        if (lane >= off) { is += us; ir += ur; }
      }
      if (lane == 63) { wsum_s[wid] = is; wsum_r[wid] = ir; }
      __syncthreads();
      uint32_t ws = carry_s, wr = carry_r, ts = 0, tr = 0;
#pragma unroll
      for (int w = 0; w < 16; ++w) {
        if (w < wid) { ws += wsum_s[w]; wr += wsum_r[w]; }
        ts += wsum_s[w];
        tr += wsum_r[w];
      }
      if (mine) {
        seg_dst[a] = ws + is - cs;
        rec_dst[a] = wr + ir - cr;
      }
      __syncthreads();
      if (tid == 0) { carry_s += ts; carry_r += tr; }
      __syncthreads();
    }
    if (tid == 0) {
      counts[3 * p] = (long long)(carry_s - base_s);
      counts[3 * p + 1] = (long long)(carry_r - base_r);
    }
    __syncthreads();
  }
  if (tid == 0) {
    uint32_t b = 0;
    for (int p = 0; p < nranks; ++p) {
      const uint32_t c = ctr->err ? 0u : run_counts[p];
      run_base[p] = b;
      run_fill[p] = 0;
      counts[3 * p + 2] = (long long)c;
      b += c;
    }
  }
}

// The wire form of a segment descriptor (two uint4):
//   {chunk key low, first record relative to the destination's record block, chunk key high, tile},
//   {slab offsets 1..7 as in the local form, the record count in the 16 bits of offset 0 (always 0)}.
// A wave per segment: its lanes copy the records (contiguous 16-byte items), lane 0 writes the descriptor.
__global__ __launch_bounds__(256) void shard_pack_segments(
    const uint4* __restrict__ sorted_seg, const uint4* __restrict__ rec, const uint32_t* __restrict__ active,
    const uint32_t* __restrict__ active_off, const int32_t* __restrict__ slot_ids, const uint32_t* __restrict__ owner,
    const uint32_t* __restrict__ seg_dst, const uint32_t* __restrict__ rec_dst, const uint32_t* __restrict__ seg_pre,
    const uint32_t* __restrict__ obase, const WalkCounters* __restrict__ ctr, uint4* __restrict__ seg_out,
    uint4* __restrict__ rec_out) {
  if (ctr->err) return;
  const int lane = threadIdx.x & 63;
  const uint32_t n = ctr->num_updated;
  const uint32_t total = active_off[n];
  const uint32_t nwaves = gridDim.x * 4u;
  for (uint32_t j = blockIdx.x * 4u + (threadIdx.x >> 6); j < total; j += nwaves) {
    uint32_t lo = 0, hi = n - 1;   // the chunk of segment j: the last a with active_off[a] <= j
    while (lo < hi) {
      const uint32_t mid = (lo + hi + 1) >> 1;
      if (active_off[mid] <= j) lo = mid; else hi = mid - 1;
    }
    const uint32_t a = lo;
    const uint4 d0 = sorted_seg[2 * (size_t)j];
    const uint32_t ds = rec_dst[a] + seg_pre[j];
    if (lane == 0) {
      const uint4 d1 = sorted_seg[2 * (size_t)j + 1];
      const int32_t* id = slot_ids + 3 * (size_t)active[a];
      unsigned long long key = 0;
      pack_block(id[0], id[1], id[2], &key);   // (in range: the walk packed it before)
      const size_t o = (size_t)seg_dst[a] + (j - active_off[a]);
      seg_out[2 * o] = make_uint4((uint32_t)key, ds - obase[2 * owner[a] + 1], (uint32_t)(key >> 32), d0.w);
      seg_out[2 * o + 1] = make_uint4(d1.x | d0.z, d1.y, d1.z, d1.w);
    }
    for (uint32_t r = (uint32_t)lane; r < d0.z; r += 64) rec_out[(size_t)ds + r] = rec[(size_t)d0.y + r];
  }
}

// Received descriptors (grouped by source rank; src_off[q], src_off[nranks + 1 + q] = first segment / first
// record of source q's block) -> the local form, with the chunk's slot in the owner's directory (first-touch
// chunks are inserted) and record offsets into the whole receive buffer.
__global__ __launch_bounds__(256) void shard_translate(const uint4* __restrict__ seg_in, uint32_t total,
                                                       const uint32_t* __restrict__ src_off, int nranks, Directory dir,
                                                       int32_t* __restrict__ num_chunks, uint32_t* __restrict__ err,
                                                       uint4* __restrict__ seg_out) {
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  if (j >= total) return;
  int q = 0;
  while (q + 1 < nranks && src_off[q + 1] <= j) ++q;
  const uint4 d0 = seg_in[2 * (size_t)j], d1 = seg_in[2 * (size_t)j + 1];
  const unsigned long long key = (unsigned long long)d0.x | ((unsigned long long)d0.z << 32);
  const int x = (int)((key >> 42) & 0x1FFFFFu) - kCoordBias, y = (int)((key >> 21) & 0x1FFFFFu) - kCoordBias,
            z = (int)(key & 0x1FFFFFu) - kCoordBias;
  const int slot = dir_find_or_insert(dir, x, y, z, num_chunks, err);
  uint32_t c = d1.x & 0xFFFFu;
  if (slot < 0) c = 0;   // (err is set: apply_chunks leaves at once, the host reports it)
  seg_out[2 * (size_t)j] = make_uint4(slot < 0 ? 0u : (uint32_t)slot, d0.y + src_off[nranks + 1 + q], c, d0.w);
  seg_out[2 * (size_t)j + 1] = make_uint4(d1.x & 0xFFFF0000u, d1.y, d1.z, d1.w);
}

// ---- colour runs
// Runs per destination (dkey = dense run keys in local tile order, slot * 4096 + voxel of the walk directory).
__global__ __launch_bounds__(256) void shard_run_count(const uint32_t* __restrict__ dkey, const uint32_t* __restrict__ nd_dev,
                                                       const uint32_t* __restrict__ slot_owner, int nranks,
                                                       uint32_t* __restrict__ run_counts, const WalkCounters* __restrict__ ctr) {
  __shared__ uint32_t hist[64];
  if (ctr->err) return;   // (a walk that ran out of room leaves unwritten run slots behind: the host repeats the call)
  if (threadIdx.x < 64) hist[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t nd = *nd_dev;
  for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < nd; j += gridDim.x * 256u)
    atomicAdd(&hist[slot_owner[dkey[j] >> 12]], 1u);
  __syncthreads();
  if ((int)threadIdx.x < nranks && hist[threadIdx.x]) atomicAdd(&run_counts[threadIdx.x], hist[threadIdx.x]);
}

// Wire form of a run, kWireRun words: {chunk key low, chunk key high, voxel, tile, ray mask}.  The runs of a
// destination keep no particular order (the owner sorts by voxel and tile); a workgroup reserves its places
// with one atomic per destination.
constexpr int kRunSpan = 1024;
__global__ __launch_bounds__(256) void shard_run_pack(const uint32_t* __restrict__ dkey, const uint32_t* __restrict__ dval,
                                                      const uint32_t* __restrict__ nd_dev, const uint32_t* __restrict__ masks,
                                                      uint32_t r1_log2, TileMap tmap, const int32_t* __restrict__ slot_ids,
                                                      const uint32_t* __restrict__ slot_owner, const uint32_t* __restrict__ run_base,
                                                      uint32_t* __restrict__ run_fill, uint32_t* __restrict__ out) {
  __shared__ uint32_t hist[64], base[64];
  const uint32_t nd = *nd_dev;
  for (uint32_t b0 = blockIdx.x * (uint32_t)kRunSpan; b0 < nd; b0 += gridDim.x * (uint32_t)kRunSpan) {
    if (threadIdx.x < 64) hist[threadIdx.x] = 0;
    __syncthreads();
    uint32_t own[kRunSpan / 256], rnk[kRunSpan / 256];
#pragma unroll
    for (int q = 0; q < kRunSpan / 256; ++q) {
      const uint32_t j = b0 + (uint32_t)(q * 256) + threadIdx.x;
      own[q] = 0xFFFFFFFFu;
      if (j < nd) {
        own[q] = slot_owner[dkey[j] >> 12];
        rnk[q] = atomicAdd(&hist[own[q]], 1u);
      }
    }
    __syncthreads();
    if (threadIdx.x < 64 && hist[threadIdx.x]) base[threadIdx.x] = run_base[threadIdx.x] + atomicAdd(&run_fill[threadIdx.x], hist[threadIdx.x]);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kRunSpan / 256; ++q) {
      if (own[q] == 0xFFFFFFFFu) continue;
      const uint32_t j = b0 + (uint32_t)(q * 256) + threadIdx.x;
      const uint32_t key = dkey[j], val = dval[j];
      const int32_t* id = slot_ids + 3 * (size_t)(key >> 12);
      unsigned long long ck = 0;
      pack_block(id[0], id[1], id[2], &ck);
      uint4* dst = reinterpret_cast<uint4*>(out + (size_t)(base[own[q]] + rnk[q]) * kWireRun);
      const uint4* m4 = reinterpret_cast<const uint4*>(masks + (size_t)val * kMaskWords);
      dst[0] = make_uint4((uint32_t)ck, (uint32_t)(ck >> 32), key & 0xFFFu, tmap.tile_of(val >> r1_log2));
#pragma unroll
      for (int w = 0; w < kMaskWords / 4; ++w) dst[1 + w] = m4[w];
    }
    __syncthreads();
  }
}

// Received runs -> voxel key in the owner's pool (its chunk is there: the records of the same call created it),
// tile, and the run's index as the value the sorts carry.
__global__ __launch_bounds__(256) void shard_run_translate(const uint32_t* __restrict__ runs, uint32_t total, Directory dir,
                                                           uint32_t* __restrict__ err, uint32_t* __restrict__ vkey,
                                                           uint32_t* __restrict__ tile, uint32_t* __restrict__ val) {
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  if (j >= total) return;
  const uint4 hd = *reinterpret_cast<const uint4*>(runs + (size_t)j * kWireRun);
  const unsigned long long key = (unsigned long long)hd.x | ((unsigned long long)hd.y << 32);
  const int slot = dir_find(dir, (int)((key >> 42) & 0x1FFFFFu) - kCoordBias, (int)((key >> 21) & 0x1FFFFFu) - kCoordBias,
                            (int)(key & 0x1FFFFFu) - kCoordBias);
  if (slot < 0) atomicOr(err, kErrDirectoryMiss);
  vkey[j] = (slot < 0 ? 0u : (uint32_t)slot) * (uint32_t)kChunkVox + (hd.z & 0xFFFu);
  tile[j] = hd.w;
  val[j] = j;
}

__global__ void shard_gather_keys(const uint32_t* __restrict__ vkey, const uint32_t* __restrict__ order, uint32_t n,
                                  uint32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = vkey[order[i]];
}

// Voxels that reached colour weight 254 in the call, as {chunk x, y, z, voxel} for the all-gather.
__global__ void shard_saturated_ids(const uint32_t* __restrict__ sat_list, uint32_t n, const int32_t* __restrict__ slot_ids,
                                    int32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t key = sat_list[i];
  const int32_t* id = slot_ids + 3 * (size_t)(key >> 12);
  out[4 * (size_t)i + 0] = id[0];
  out[4 * (size_t)i + 1] = id[1];
  out[4 * (size_t)i + 2] = id[2];
  out[4 * (size_t)i + 3] = (int32_t)(key & 0xFFFu);
}

// ... noted in the walk directory of a rank (chunks it has not walked through yet are entered).
__global__ void shard_note_saturated(const int32_t* __restrict__ list, uint32_t n, Directory xdir, int32_t* __restrict__ xcount,
                                     uint32_t* __restrict__ err, uint32_t* __restrict__ sat) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t* e = list + 4 * (size_t)i;
  const int slot = dir_find_or_insert(xdir, e[0], e[1], e[2], xcount, err);
  if (slot < 0) return;
  const uint32_t v = (uint32_t)slot * (uint32_t)kChunkVox + ((uint32_t)e[3] & 0xFFFu);
  atomicOr(&sat[v >> 5], 1u << (v & 31u));
}

}  // namespace
