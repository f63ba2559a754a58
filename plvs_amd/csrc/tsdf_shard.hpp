// Ray-sharded multi-GPU integrate of the open_chisel back end (order-free mode; included by tsdf_chisel.hip
// after tsdf_walk.hpp).  New design: the reference is a single process.
//
// The map is sharded by chunk (owner = three-prime ChunkHasher mod N, ChunkManager.h:42-54), the WORK by
// tile of the point stream: rank r walks the tiles t = r (mod N) of every call — its share of the rays,
// whatever chunks they cross — and what it collects for a chunk travels to the chunk's owner:
//
//   shard_walk    walk_tiles over the rank's tiles into the rank's WALK DIRECTORY (every chunk the rank has ever
//                 walked through: ids + one bit per voxel "colour saturated", no voxel data), segment sort, then the
//                 rank's OWN aggregation: apply_chunks<emit> adds the records of every (chunk, slab) in LDS and
//                 leaves one 32-byte sum per touched voxel + one descriptor per (chunk, slab) in the send region of
//                 the chunk's owner (shard_chunk_totals / shard_plan size the regions) -> send counts.  A voxel seen
//                 from a thousand tiles of this rank travels once.
//   shard_pack    the used parts of the regions and the colour runs, grouped by destination, into the caller's
//                 send buffers
//   (exchange)    ONE all-to-all of the three buffers — RCCL send/recv behind plvs_hip_tsdf_chisel_integrate_sharded,
//                 torch.distributed in the Python mirror, device copies between virtual ranks in the tests
//   shard_apply   received descriptors -> slots of the owner's directory (shard_translate, first-touch chunks are
//                 inserted here), the segment sort and apply_chunks of the single-device path on the received
//                 sums (N per voxel at most).  The sums are integers: the result is bit-identical to the single-device order-free
//                 integrate whatever N is.
//                 Colours (the truncating u8 mean is order dependent below weight 254): the received runs
//                 (voxel, tile, the tile's rays as spans: kWireRun, tsdf_walk.hpp) are sorted by (voxel, tile) and
//                 folded as on a single device.  A walker
//                 sends a run for every voxel it does not KNOW to be saturated; owners list the voxels that
//                 reach 254 in a call (shard_saturated), the lists are all-gathered and every rank notes them in
//                 its walk directory (shard_note_saturated).  Late knowledge only costs surplus runs (the fold
//                 stops at 254 by itself); colour weights never decrease short of Clear().
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "tsdf_walk.hpp"

namespace {

// Per updated chunk of the call (index a in `active`): its owner and the records of its segments.
__global__ __launch_bounds__(256) void shard_chunk_totals(const uint4* __restrict__ sorted_seg,
                                                          const uint32_t* __restrict__ active,
                                                          const uint32_t* __restrict__ active_off,
                                                          const int32_t* __restrict__ slot_ids, int nranks,
                                                          const WalkCounters* __restrict__ ctr,
                                                          uint32_t* __restrict__ nrec, uint32_t* __restrict__ owner,
                                                          uint32_t* __restrict__ slot_owner) {
  __shared__ uint32_t wsum[4];
  if (ctr->err) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t n = ctr->num_updated;
  for (uint32_t a = blockIdx.x; a < n; a += gridDim.x) {
    const uint32_t s0 = active_off[a], s1 = active_off[a + 1];
    uint32_t c = 0;
    for (uint32_t j = s0 + (uint32_t)tid; j < s1; j += 256) c += sorted_seg[2 * (size_t)j].z;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += (uint32_t)__shfl_xor((int)c, off);
    if (lane == 0) wsum[wid] = c;
    __syncthreads();
    if (tid == 0) {
      const int32_t* id = slot_ids + 3 * (size_t)active[a];
      nrec[a] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
      owner[a] = (uint32_t)shard_of(chunk_hash(id[0], id[1], id[2]), nranks);
      slot_owner[active[a]] = owner[a];
    }
    __syncthreads();
  }
}

// Send regions: destination p gets room for 8 descriptors and min(4096, records) sums per chunk it owns among the
// call's chunks (a chunk has 4096 voxels; it cannot have more touched voxels than records).  plan[0], plan[1] = the
// two totals, seg_region / rec_region[p] = the first descriptor / sum of p's region; the fill cursors are zeroed.
// ctl = {seg_region[64], rec_region[64], seg_fill[64], rec_fill[64]}.
__global__ __launch_bounds__(1024) void shard_plan(const uint32_t* __restrict__ nrec, const uint32_t* __restrict__ owner,
                                                   int nranks, const WalkCounters* __restrict__ ctr,
                                                   uint32_t* __restrict__ ctl, uint32_t* __restrict__ plan) {
  __shared__ uint32_t segs[64], recs[64];
  const int tid = threadIdx.x;
  if (tid < 64) { segs[tid] = 0; recs[tid] = 0; }
  __syncthreads();
  const uint32_t n = ctr->err ? 0u : ctr->num_updated;
  for (uint32_t a = (uint32_t)tid; a < n; a += 1024) {
    atomicAdd(&segs[owner[a]], (uint32_t)kSlabs);
    atomicAdd(&recs[owner[a]], min(nrec[a], (uint32_t)kChunkVox));
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t bs = 0, br = 0;
    for (int p = 0; p < 64; ++p) {
      ctl[p] = bs;
      ctl[64 + p] = br;
      ctl[128 + p] = 0;
      ctl[192 + p] = 0;
      if (p < nranks) { bs += segs[p]; br += recs[p]; }
    }
    plan[0] = bs;
    plan[1] = br;
  }
}

// The used part of every send region -> the caller's buffers, destination after destination
// (ctl as above; dst_off[p], dst_off[64 + p] = where p's descriptors / sums start there).
__global__ __launch_bounds__(256) void shard_copy_regions(const uint4* __restrict__ seg_reg, const uint4* __restrict__ rec_reg,
                                                          const uint32_t* __restrict__ ctl, const uint32_t* __restrict__ dst_off,
                                                          int nranks, uint4* __restrict__ seg_out, uint4* __restrict__ rec_out) {
  for (int p = 0; p < nranks; ++p) {
    const uint32_t ns = 2u * ctl[128 + p], nr = 2u * ctl[192 + p];   // in uint4 (two per descriptor / per sum)
    const uint4* ss = seg_reg + 2 * (size_t)ctl[p];
    const uint4* rs = rec_reg + 2 * (size_t)ctl[64 + p];
    uint4* sd = seg_out + 2 * (size_t)dst_off[p];
    uint4* rd = rec_out + 2 * (size_t)dst_off[64 + p];
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < ns; i += gridDim.x * 256u) sd[i] = ss[i];
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < nr; i += gridDim.x * 256u) rd[i] = rs[i];
  }
}

// The wire form of a segment descriptor (two uint4), written by apply_chunks<emit>:
//   {chunk key low, first sum relative to the destination's block, chunk key high, slab},
//   {slab offsets as in the local form — all the sums lie in ONE slab —, the count in the 16 bits of offset 0}.
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
// Wire records per destination (dkey = dense run keys in local tile order, slot * 4096 + voxel of the walk directory; dval
// = the run's place in the per-tile regions): a run takes one record per six spans of its mask (kWireRun, tsdf_walk.hpp).
__global__ __launch_bounds__(256) void shard_run_count(const uint32_t* __restrict__ dkey, const uint32_t* __restrict__ dval,
                                                       const uint32_t* __restrict__ nd_dev, const uint32_t* __restrict__ masks,
                                                       const uint32_t* __restrict__ slot_owner, int nranks,
                                                       uint32_t* __restrict__ run_counts, uint4* __restrict__ run_first,
                                                       const WalkCounters* __restrict__ ctr) {
  // run_first[j] = {the spans of the run's first record (three words), its number of records}: shard_run_pack reads the
  // 64-byte mask again only for the few runs with more than six spans.
  __shared__ uint32_t hist[64];
  if (ctr->err) return;   // (a walk that ran out of room leaves unwritten run slots behind: the host repeats the call)
  if (threadIdx.x < 64) hist[threadIdx.x] = 0;
  __syncthreads();
  // (the masks read through LDS — four neighbouring lanes per mask, sixteen whole masks per load instruction, then a lane
  // its own 64 bytes from LDS — made it slower, 211 -> 265 us: the kernel is bound by the 350 MB of masks and the
  // divergent span loop, not by how the lanes address them)
  const uint32_t nd = *nd_dev;
  for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < nd; j += gridDim.x * 256u) {
    uint32_t m[kMaskWords];
    const uint4* m4 = reinterpret_cast<const uint4*>(masks + (size_t)dval[j] * kMaskWords);
#pragma unroll
    for (int q = 0; q < kMaskWords / 4; ++q) {
      const uint4 a = m4[q];
      m[4 * q] = a.x; m[4 * q + 1] = a.y; m[4 * q + 2] = a.z; m[4 * q + 3] = a.w;
    }
    uint32_t acc[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, have = 0;
    const uint32_t spans = mask_spans(m, [&](uint32_t first, uint32_t len) {
      if (have < kWireSpans) {
        const uint32_t code = first | ((len - 1u) << 9), sh = (have & 1u) * 16u;
        uint32_t& word = acc[have >> 1];
        word = (word & ~(0xFFFFu << sh)) | (code << sh);
      }
      ++have;
    });
    const uint32_t nrec = max(1u, (spans + kWireSpans - 1u) / kWireSpans);   // (<= 43)
    run_first[j] = make_uint4(acc[0], acc[1], acc[2], nrec);
    atomicAdd(&hist[slot_owner[dkey[j] >> 12]], nrec);
  }
  __syncthreads();
  if ((int)threadIdx.x < nranks && hist[threadIdx.x]) atomicAdd(&run_counts[threadIdx.x], hist[threadIdx.x]);
}

// Bases of the destinations' runs in the send buffer (rc = {counts[64], bases[64], fill cursors[64]}).
__global__ void shard_run_plan(uint32_t* __restrict__ rc, int nranks, const WalkCounters* __restrict__ ctr) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint32_t b = 0;
  for (int p = 0; p < 64; ++p) {
    if (ctr->err || p >= nranks) rc[p] = 0;
    rc[64 + p] = b;
    rc[128 + p] = 0;
    b += rc[p];
  }
}

// The records of a destination keep no particular order between runs (the owner sorts by voxel and tile; the records of
// ONE run are consecutive and in span order); a workgroup reserves its places with one atomic per destination.
constexpr int kRunSpan = 1024;
__global__ __launch_bounds__(256) void shard_run_pack(const uint32_t* __restrict__ dkey, const uint32_t* __restrict__ dval,
                                                      const uint32_t* __restrict__ nd_dev, const uint32_t* __restrict__ masks,
                                                      const uint4* __restrict__ run_first,
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
        rnk[q] = atomicAdd(&hist[own[q]], run_first[j].w);
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
      const uint2 head = make_uint2((uint32_t)ck, (uint32_t)(ck >> 32));
      const uint32_t vt = (key & 0xFFFu) | (tmap.tile_of(val >> r1_log2) << 12);
      uint2* dst = reinterpret_cast<uint2*>(out + (size_t)(base[own[q]] + rnk[q]) * kWireRun);
      const uint4 first = run_first[j];
      if (first.w == 1u) {   // (nearly every run)
        dst[0] = head;
        dst[1] = make_uint2(vt, first.x);
        dst[2] = make_uint2(first.y, first.z);
        continue;
      }
      uint32_t m[kMaskWords];
      const uint4* m4 = reinterpret_cast<const uint4*>(masks + (size_t)val * kMaskWords);
#pragma unroll
      for (int w = 0; w < kMaskWords / 4; ++w) {
        const uint4 a = m4[w];
        m[4 * w] = a.x; m[4 * w + 1] = a.y; m[4 * w + 2] = a.z; m[4 * w + 3] = a.w;
      }
      uint32_t acc[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, have = 0;
      auto flush = [&]() {
        dst[0] = head;
        dst[1] = make_uint2(vt, acc[0]);
        dst[2] = make_uint2(acc[1], acc[2]);
        dst += 3;
        acc[0] = acc[1] = acc[2] = 0xFFFFFFFFu;
        have = 0;
      };
      const uint32_t spans = mask_spans(m, [&](uint32_t first, uint32_t len) {
        if (have == kWireSpans) flush();
        const uint32_t code = first | ((len - 1u) << 9), sh = (have & 1u) * 16u;
        uint32_t& word = acc[have >> 1];
        word = (word & ~(0xFFFFu << sh)) | (code << sh);
        ++have;
      });
      if (have || spans == 0u) flush();
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
  const uint2 ck = reinterpret_cast<const uint2*>(runs + (size_t)j * kWireRun)[0];
  const uint32_t vt = runs[(size_t)j * kWireRun + 2];
  const uint4 hd = make_uint4(ck.x, ck.y, vt & 0xFFFu, vt >> 12);
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


// The rank's message for the saturation all-gather: rows [0, k) = waiting voxels, row `rows` = {k, 0, 0, 0}.
__global__ void shard_sat_message(const int32_t* __restrict__ wait, uint32_t k, uint32_t rows, int32_t* __restrict__ msg) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < k) reinterpret_cast<int4*>(msg)[i] = reinterpret_cast<const int4*>(wait)[i];
  if (i == 0) reinterpret_cast<int4*>(msg)[rows] = make_int4((int)k, 0, 0, 0);
}

// The gathered messages of all ranks (nranks x (rows + 1) entries of {chunk x, y, z, voxel}) noted in one launch: every
// message's length is read from its last row on the device — no host read between the all-gather and this.
__global__ void shard_note_gathered(const int32_t* __restrict__ gathered, uint32_t rows, Directory xdir, int32_t* __restrict__ xcount,
                                    uint32_t* __restrict__ err, uint32_t* __restrict__ sat) {
  const int32_t* list = gathered + 4 * (size_t)blockIdx.y * (rows + 1);
  const uint32_t k = min((uint32_t)list[4 * (size_t)rows], rows);
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  const int32_t* e = list + 4 * (size_t)i;
  const int slot = dir_find_or_insert(xdir, e[0], e[1], e[2], xcount, err);
  if (slot < 0) return;
  const uint32_t v = (uint32_t)slot * (uint32_t)kChunkVox + ((uint32_t)e[3] & 0xFFFu);
  atomicOr(&sat[v >> 5], 1u << (v & 31u));
}

}  // namespace
