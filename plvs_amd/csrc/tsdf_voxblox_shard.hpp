// Ray-sharded multi-GPU integrate of the voxblox back end (SimpleTsdfIntegrator; included by tsdf_voxblox.hip).  New
// design: the reference is a single process.
//
// The map is sharded by block (owner = three-prime AnyIndexHash mod N, block_hash.h:21-24), the WORK by key frame: rank r
// casts the rays of the clouds c = r (mod N) of every call — through whatever blocks they cross — and every voxel
// visit travels to the block's owner as a 16-byte record {block id (packed, 8 B), voxel | cloud << 12, sequence number
// of the ray}.  updateTsdfVoxel is order dependent (clamp, float blend), so the owner applies a voxel's visits in the
// reference's order: the sequence numbers are GLOBAL (cloud after cloud, the mixed order inside a cloud), a sender's
// records leave in sequence order (a stable partition by destination), the owner sorts what it received by cloud and then
// by voxel (stable radix passes) and runs the single-device pipeline's expansion and fold on it — the operands of a
// visit are recomputed from (point, voxel) at the owner, which holds the same clouds.  The union of the shards is the
// single-device layer bit for bit, whatever N is (tests/test_tsdf_voxblox_shard.py).  Before (rounds 1-3) every rank cast
// every ray and kept the visits of its own blocks: the walk — a fifth of a call — did not scale.
//
//   shard_walk    count pass + fill pass over this rank's clouds (no directory: block ids come from the voxel grid),
//                 records in sequence order; destination of every record, stable partition -> send counts
//   shard_pack    the records grouped by destination, in the caller's send buffer
//   (exchange)    counts, then ONE all-to-all of the records
//   shard_apply   block id -> slot of the owner's directory (first-touch blocks inserted), sort by cloud, sort by voxel,
//                 vb_expand, vb_chain_chunks
#pragma once

namespace {

constexpr uint32_t kVbWire = 4;   // 32-bit words of a visit record on the wire

template <bool kFill>
__global__ __launch_bounds__(256) void vb_shard_ray_pass(Params P, const float* __restrict__ xyz, int npoints,
                                                         const int32_t* __restrict__ offsets, int nclouds,
                                                         const PoseRt* __restrict__ Twc, int rank, int nranks,
                                                         VCounters* __restrict__ ctr, uint32_t* __restrict__ counts,
                                                         uint4* __restrict__ rec, uint32_t* __restrict__ dest) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // sequence position
  if (i >= npoints) return;
  uint32_t n = 0;
  const uint32_t out = kFill ? counts[i] : 0u;
  int cloud = 0;
  const int p = point_of_seq(offsets, nclouds, i, &cloud);
  if (cloud % nranks == rank) {
    const float px = xyz[3 * (size_t)p], py = xyz[3 * (size_t)p + 1], pz = xyz[3 * (size_t)p + 2];
    if (!(isfinite(px) && isfinite(py) && isfinite(pz))) {
      if (!kFill) atomicOr(&ctr->err, kErrNonFinite);
    } else {
      const PoseRt pose = load_pose(Twc, cloud);
      Ray ray;
      if (make_ray(P, pose, px, py, pz, &ray, false)) {
        const int steps = ray.steps < kMaxRaySteps ? ray.steps : kMaxRaySteps;
        for (int s = 0; s <= steps; ++s) {
          int g[3], b[3], vid;
          ray_step(&ray, g);
          if (!block_of(P, g, b, &vid)) continue;   // (P.shard_count = 1 here: never)
          unsigned long long key = 0;
          const bool in_range = pack_block(b[0], b[1], b[2], &key) &&
                                ((((g[0] - b[0] * 16) | (g[1] - b[1] * 16) | (g[2] - b[2] * 16)) & ~15) == 0);
          if (!kFill && !in_range) atomicOr(&ctr->err, kErrCoordRange);
          if (kFill) {
            rec[out + n] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), (uint32_t)vid | ((uint32_t)cloud << 12), (uint32_t)i);
            dest[out + n] = (uint32_t)shard_of(owner_hash(b[0], b[1], b[2]), nranks);
          }
          ++n;
        }
      }
    }
  }
  if (!kFill) counts[i] = n;
}

__global__ void vb_iota(uint32_t* __restrict__ a, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = i;
}

// counts[d] = records bound for rank d (dest sorted ascending): the first position of every destination by bisection
__global__ void vb_shard_dest_counts(const uint32_t* __restrict__ sorted_dest, uint32_t n, int nranks, uint32_t* __restrict__ counts) {
  const int d = threadIdx.x;
  if (d >= nranks) return;
  auto first_ge = [&](uint32_t v) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (sorted_dest[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  counts[d] = first_ge((uint32_t)d + 1u) - first_ge((uint32_t)d);
}

__global__ void vb_shard_gather(const uint4* __restrict__ rec, const uint32_t* __restrict__ order, uint32_t n, uint4* __restrict__ out) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) out[j] = rec[order ? order[j] : j];
}

// Received records: first-touch blocks enter the owner's directory (dir_insert never waits: the slot of a block inserted here
// is looked up by the NEXT kernel, as the count pass and the fill pass of the single-device call do — a find-or-insert that
// waits for another lane's slot met a circular wait between waves here, 64 records of a wave naming the same new block).
__global__ __launch_bounds__(256) void vb_shard_insert(const uint4* __restrict__ rec, uint32_t n, Directory dir, int rank, int nranks,
                                                       VCounters* __restrict__ ctr) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint4 r = rec[j];
  if (j > 0) {   // (consecutive records mostly name the same block: one insertion attempt per change)
    const uint4 q = rec[j - 1];
    if (q.x == r.x && q.y == r.y && (j & 63u) != 0u) return;
  }
  const unsigned long long key = (unsigned long long)r.x | ((unsigned long long)r.y << 32);
  const int x = (int)((key >> 42) & 0x1FFFFFu) - kCoordBias, y = (int)((key >> 21) & 0x1FFFFFu) - kCoordBias,
            z = (int)(key & 0x1FFFFFu) - kCoordBias;
  if (shard_of(owner_hash(x, y, z), nranks) != rank) atomicOr(&ctr->err, kErrDirectoryMiss);   // (a record sent to the wrong rank)
  else dir_insert(dir, x, y, z, &ctr->num_blocks, &ctr->err);
}

// ... -> voxel key in the owner's pool, sequence number, cloud.
__global__ __launch_bounds__(256) void vb_shard_translate(const uint4* __restrict__ rec, uint32_t n, Directory dir,
                                                          VCounters* __restrict__ ctr, uint32_t* __restrict__ vkey,
                                                          uint32_t* __restrict__ seq, uint32_t* __restrict__ cloud,
                                                          uint32_t* __restrict__ index) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint4 r = rec[j];
  const unsigned long long key = (unsigned long long)r.x | ((unsigned long long)r.y << 32);
  const int x = (int)((key >> 42) & 0x1FFFFFu) - kCoordBias, y = (int)((key >> 21) & 0x1FFFFFu) - kCoordBias,
            z = (int)(key & 0x1FFFFFu) - kCoordBias;
  const int slot = dir_find(dir, x, y, z);
  if (slot < 0) atomicOr(&ctr->err, kErrDirectoryMiss);
  vkey[j] = (slot < 0 ? 0u : (uint32_t)slot) * (uint32_t)kBlockVox + (r.z & 0xFFFu);
  seq[j] = r.w;
  if (cloud) {
    cloud[j] = r.z >> 12;
    index[j] = j;
  }
}

__global__ void vb_shard_permute(const uint32_t* __restrict__ vkey, const uint32_t* __restrict__ seq, const uint32_t* __restrict__ order,
                                 uint32_t n, uint32_t* __restrict__ keys_out, uint32_t* __restrict__ seq_out) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint32_t j = order[k];
  keys_out[k] = vkey[j];
  seq_out[k] = seq[j];
}

}  // namespace
