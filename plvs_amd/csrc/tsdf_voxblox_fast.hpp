// FastTsdfIntegrator (Thirdparty/voxblox/src/integrator/tsdf_integrator.cc:505-605; PLVS's YAML default) — which rays of a
// scan are cast and how far, decided on the device for the reference's ONE-thread schedule.  Included by
// tsdf_voxblox.hip, which then sends the surviving voxel visits through the ordered pipeline of the simple integrator.
//
// The integrator (per scan, points in the mixed order): a point starts a ray only if no earlier point of the scan lies
// in the same voxel of HALF the voxel size ("start set"); the ray is cast from its far end towards the sensor and stops
// at the third voxel in a row that an earlier ray of the scan went through ("observed set"); every voxel before the stop
// takes updateTsdfVoxel.  Both sets are ApproxHashSet<20, 10000> (utils/approx_hash_array.h:66-160): an array of 2^20 +
// 10 000 words; index x lives at word (hash(x) & 0xFFFFF) + offset and a query "replaceHash" answers "new" unless that
// word holds hash(x), and leaves hash(x) there — whatever was there is forgotten.  A "reset" is offset + 1 (stale
// words stay; they can only be mistaken for the index whose hash is 0); every 10 000 scans the array is zeroed.
//
// One thread walks the points in order; a GPU cannot — a ray's stop depends on the queries of all rays before it.
// But the answer to a query depends only on the PREVIOUS query of the same word (or, for the first one of a batch, on
// what the array holds).  So, given the number of queries Q_r every ray makes, all answers follow from one stable sort
// of the queries by word; from the answers every ray reads off where it stops; the rounds repeat until no Q changes.
// Ray r's answers depend on rays < r and its own earlier steps only: the fixed point is unique and it is the sequential
// schedule (tests/test_tsdf_voxblox_fast.py runs the same procedure on the CPU against the plain loop).  Rounds needed on depth-camera clouds: 4-15 (40 with carving at 2 cm).  The arrays live on the device with the
// map, and keep their content from scan to scan as the reference's do.
#pragma once

namespace {

constexpr int kApproxBits = 20;
constexpr uint32_t kApproxMask = (1u << kApproxBits) - 1u;
constexpr uint32_t kApproxReset = 10000;                                   // full_reset_threshold
constexpr size_t kApproxWords = ((size_t)1 << kApproxBits) + kApproxReset;
constexpr int kApproxKeyBits = 21;                                         // words < 2^21
constexpr uint32_t kNoQuery = 0xFFFFFFFFu;
constexpr uint32_t kFastWindow = 6;   // steps of a ray examined in the first round (most rays stop within three or four)

// AnyIndexHash (core/block_hash.h:21-24): unsigned x, sign-extended y and z, 64-bit products
__device__ __forceinline__ unsigned long long any_index_hash(int x, int y, int z) {
  return ((unsigned long long)(unsigned int)x * 73856093ull) ^ ((unsigned long long)(long long)y * 19349663ull) ^
         ((unsigned long long)(long long)z * 83492791ull);
}

__global__ void vbf_init_table(unsigned long long* __restrict__ t, size_t words) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < words) t[i] = i == 0 ? ~0ull : 0ull;   // pseudo_set_[offset_ = 0] = max (approx_hash_array.h:78, 148)
}

// The start-set query of every point (sequence position i: cloud-major, mixed order inside a cloud) and the length of
// its ray.  first_offset = the set offset of the batch's first scan.
__global__ __launch_bounds__(256) void vbf_start(Params P, const float* __restrict__ xyz, int npoints,
                                                 const int32_t* __restrict__ offsets, int nclouds,
                                                 const PoseRt* __restrict__ Twc, uint32_t first_offset,
                                                 uint32_t* __restrict__ skey, uint32_t* __restrict__ sval,
                                                 unsigned long long* __restrict__ shash, uint32_t* __restrict__ full,
                                                 VCounters* __restrict__ ctr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npoints) return;
  int cloud = 0;
  const int p = point_of_seq(offsets, nclouds, i, &cloud);
  const float px = xyz[3 * (size_t)p], py = xyz[3 * (size_t)p + 1], pz = xyz[3 * (size_t)p + 2];
  uint32_t key = kNoQuery, len = 0;
  unsigned long long hash = 0;
  if (!(isfinite(px) && isfinite(py) && isfinite(pz))) {
    atomicOr(&ctr->err, kErrNonFinite);
  } else {
    const PoseRt pose = load_pose(Twc, cloud);
    Ray ray;
    if (make_ray(P, pose, px, py, pz, &ray, true)) {
      // getGridIndexFromPoint(point_G, start_voxel_subsampling_factor * voxel_size_inv_) (:539-540)
      const float inv = 2.0f * P.voxel_size_inv;
      hash = any_index_hash((int)floorf(ray.pG[0] * inv + 1e-6f), (int)floorf(ray.pG[1] * inv + 1e-6f),
                            (int)floorf(ray.pG[2] * inv + 1e-6f));
      key = ((uint32_t)hash & kApproxMask) + first_offset + (uint32_t)cloud;
      len = (uint32_t)(ray.steps < kMaxRaySteps ? ray.steps : kMaxRaySteps) + 1u;
    }
  }
  skey[i] = key;
  sval[i] = (uint32_t)i;
  shash[i] = hash;
  full[i] = len;
}

// Answers of the start set (queries sorted by word, stable): a point whose word was last asked for the same hash
// starts no ray.  Q = the queries the ray makes in the first round.
__global__ __launch_bounds__(256) void vbf_alive(const uint32_t* __restrict__ skey, const uint32_t* __restrict__ sval, uint32_t n,
                                                 const unsigned long long* __restrict__ shash,
                                                 const unsigned long long* __restrict__ table,
                                                 const uint32_t* __restrict__ full, uint32_t* __restrict__ Q) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t key = skey[p], i = sval[p];
  if (key == kNoQuery) {
    Q[i] = 0;
    return;
  }
  const unsigned long long before = (p > 0 && skey[p - 1] == key) ? shash[sval[p - 1]] : table[key];
  Q[i] = before == shash[i] ? 0u : min(full[i], kFastWindow);
}

// What the queries leave in the array: the hash of the LAST query of every word.
__global__ __launch_bounds__(256) void vbf_write_back(const uint32_t* __restrict__ key, const uint32_t* __restrict__ val, uint32_t n,
                                                      const unsigned long long* __restrict__ hash,
                                                      unsigned long long* __restrict__ table) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t k = key[p];
  if (k == kNoQuery) return;
  if (p + 1 == n || key[p + 1] != k) table[k] = hash[val[p]];
}

// The observed-set queries of a round: ray i asks for the first Q[i] voxels on its way, in (ray, step) order.
__global__ __launch_bounds__(256) void vbf_emit(Params P, const float* __restrict__ xyz, int npoints,
                                                const int32_t* __restrict__ offsets, int nclouds,
                                                const PoseRt* __restrict__ Twc, uint32_t first_offset,
                                                const uint32_t* __restrict__ Q, const uint32_t* __restrict__ qoff,
                                                uint32_t* __restrict__ qkey, uint32_t* __restrict__ qval,
                                                unsigned long long* __restrict__ qhash) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npoints) return;
  const uint32_t nq = Q[i];
  if (nq == 0) return;
  int cloud = 0;
  const int p = point_of_seq(offsets, nclouds, i, &cloud);
  const PoseRt pose = load_pose(Twc, cloud);
  Ray ray;
  make_ray(P, pose, xyz[3 * (size_t)p], xyz[3 * (size_t)p + 1], xyz[3 * (size_t)p + 2], &ray, true);
  const uint32_t o = qoff[i];
  for (uint32_t s = 0; s < nq; ++s) {
    int g[3];
    ray_step(&ray, g);
    const unsigned long long h = any_index_hash(g[0], g[1], g[2]);
    qkey[o + s] = ((uint32_t)h & kApproxMask) + first_offset + (uint32_t)cloud;
    qval[o + s] = o + s;
    qhash[o + s] = h;
  }
}

// Answers of the observed set: seen[q] = the word of query q was last asked for the same hash.
__global__ __launch_bounds__(256) void vbf_seen(const uint32_t* __restrict__ qkey, const uint32_t* __restrict__ qval, uint32_t m,
                                                const unsigned long long* __restrict__ qhash,
                                                const unsigned long long* __restrict__ table, uint8_t* __restrict__ seen) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= m) return;
  const uint32_t key = qkey[p], q = qval[p];
  const unsigned long long before = (p > 0 && qkey[p - 1] == key) ? qhash[qval[p - 1]] : table[key];
  seen[q] = before == qhash[q] ? 1 : 0;
}

// Every ray reads its answers: the stop at the third "seen" in a row (max_consecutive_ray_collisions = 2, :557-566).  A
// ray that finds no stop among the steps it asked for asks for all of them in the next round.  L = voxels updated.
__global__ __launch_bounds__(256) void vbf_trim(int npoints, const uint32_t* __restrict__ qoff, const uint32_t* __restrict__ full,
                                                const uint8_t* __restrict__ seen, uint32_t* __restrict__ Q,
                                                uint32_t* __restrict__ L, uint32_t* __restrict__ changed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npoints) return;
  const uint32_t nq = Q[i];
  if (nq == 0) {
    L[i] = 0;
    return;
  }
  const uint8_t* a = seen + qoff[i];
  uint32_t run = 0, newq = nq, upd = nq;
  for (uint32_t s = 0; s < nq; ++s) {
    run = a[s] ? run + 1u : 0u;
    if (run > 2u) {
      newq = s + 1u;
      upd = s;
      break;
    }
  }
  if (newq == nq && upd == nq && nq < full[i]) newq = full[i];
  if (newq != nq) {
    Q[i] = newq;
    *changed = 1u;
  }
  L[i] = upd;
}

// The rounds above need as many iterations as the longest chain of rays that decide each other's stops: 4-40 on depth-camera
// clouds, but up to one per ray on an adversarial one.  Past a bound the plan is finished the way the reference makes it: ONE
// thread walks the rays that start (Q > 0: the start set was settled before the rounds, in one sort) in sequence order and asks
// the observed set voxel by voxel, leaving each answer's hash in the array at once.  Slow (two dependent memory round trips per
// query) but finite, and by construction the reference's schedule.
__global__ void vbf_sequential(Params P, const float* __restrict__ xyz, int npoints, const int32_t* __restrict__ offsets,
                               int nclouds, const PoseRt* __restrict__ Twc, uint32_t first_offset, const uint32_t* __restrict__ Q,
                               const uint32_t* __restrict__ full, unsigned long long* __restrict__ table, uint32_t* __restrict__ L) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  for (int i = 0; i < npoints; ++i) {
    if (Q[i] == 0u) {
      L[i] = 0u;
      continue;
    }
    int cloud = 0;
    const int p = point_of_seq(offsets, nclouds, i, &cloud);
    const PoseRt pose = load_pose(Twc, cloud);
    Ray ray;
    make_ray(P, pose, xyz[3 * (size_t)p], xyz[3 * (size_t)p + 1], xyz[3 * (size_t)p + 2], &ray, true);
    const uint32_t len = full[i];
    uint32_t run = 0, upd = len;
    for (uint32_t s = 0; s < len; ++s) {
      int g[3];
      ray_step(&ray, g);
      const unsigned long long h = any_index_hash(g[0], g[1], g[2]);
      const uint32_t word = ((uint32_t)h & kApproxMask) + first_offset + (uint32_t)cloud;
      const unsigned long long before = table[word];
      table[word] = h;
      run = before == h ? run + 1u : 0u;
      if (run > 2u) {
        upd = s;
        break;
      }
    }
    L[i] = upd;
  }
}

__global__ void vbf_publish(const uint32_t* __restrict__ total, const uint32_t* __restrict__ changed, uint32_t* __restrict__ host2) {
  host2[0] = *total;
  host2[1] = *changed;
}

}  // namespace
