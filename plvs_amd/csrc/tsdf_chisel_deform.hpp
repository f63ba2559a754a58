// Chisel::Deform on the device (included at the end of tsdf_chisel.hip: it works on the map handle's internals).
//
// Reference: ChunkManager::Deform, Thirdparty/open_chisel/src/ChunkManager.cpp:918-1063, behind Chisel::Deform
// (Chisel.cpp:588-591) <- ChiselServer::Deform (ChiselServer.cpp:617-621) <- PointCloudMapChisel::OnMapChange.
// Every known voxel whose kfid has a transformation moves to R * centre + t; the first voxel to land in a new voxel is
// copied, later ones are merged into it (DistVoxel::Integrate, SetKfid, ColorVoxel::Integrate(r, g, b, 1)).  "First"
// and "later" follow the reference's walk: old chunks in the iteration order of `chunks`, a
// std::unordered_map<ChunkID, ChunkPtr, ChunkHasher>, voxels of a chunk by id.  That order is libstdc++'s and a
// function of the container's insert / erase history: the reference inserts a chunk when a raycast voxel first VISITS
// it (GetOrCreateChunkAt for every voxel, Chisel.cpp:505 / :305) and erases the chunks no update reached at the end of
// the call (GarbageCollect, :574-585).  So the order is taken from the real container, on the host, as for the merged
// voxblox integrator: once plvs_hip_tsdf_chisel_enable_deform is on, every integrate call first reports the chunks it
// visits that the map does not have, in first-visit order (track_visits: one more walk of the rays, no voxel work),
// the host replays those inserts and the garbage collection into a std::unordered_map with ChunkHasher, and deform
// walks that container.  The voxel work itself is on the device:
//   deform_count / deform_emit   one workgroup per old chunk, in container order: a record (new voxel, old voxel) per
//                                moving voxel, dense and in the reference's sequence; new chunks get slots in a second
//                                directory, and the sequence number of their first claim (the order `newChunks` is
//                                filled in, hence the container order after the swap);
//   radix sort (stable)          records by new voxel: per new voxel its claimants in sequence;
//   deform_fold                  one thread per new voxel: copy the first, merge the rest, write the new planes;
//   install                      the new directory and planes replace the old ones.
// Bit-exact against the CPU restatement of the reference, which is itself pinned against the compiled reference
// library (tests/test_tsdf_deform.py, tests/test_oracle_pinned_chisel_map.py).
#pragma once
#include <algorithm>
#include <unordered_map>

namespace {

struct ChunkIdKey {
  int32_t x, y, z;
  bool operator==(const ChunkIdKey& o) const { return x == o.x && y == o.y && z == o.z; }
};
struct ChunkIdHasher {   // ChunkHasher, ChunkManager.h:42-53: int * size_t
  std::size_t operator()(const ChunkIdKey& k) const {
    return ((std::size_t)(int64_t)k.x * 73856093u) ^ ((std::size_t)(int64_t)k.y * 19349663u) ^
           ((std::size_t)(int64_t)k.z * 83492791u);
  }
};
using ChunkOrder = std::unordered_map<ChunkIdKey, bool, ChunkIdHasher>;

constexpr unsigned long long kNoSeq = ~0ull;

}  // namespace

struct ChiselDeformState {
  ChunkOrder chunks;                 // the reference's `chunks`, ids only
  std::vector<ChunkIdKey> fresh;     // inserted by the running integrate call
  // visit set of track_visits: chunk key -> sequence number of its first visit
  unsigned long long* vkeys = nullptr;
  unsigned long long* vseq = nullptr;
  uint32_t vmask = 0;
  uint32_t* d_n = nullptr;           // [0] list length, [1] error, [2..3] discarded (u64), [4..5] undefined (u64)
  DevBuf<unsigned long long> list;   // (key, seq) pairs
  DevBuf<int32_t> ids;
  DevBuf<uint32_t> found, order_slot, chunk_cnt, key0, key1, val0, val1, first, scratch, kf;
  DevBuf<float> rt, nsdf, nweight;
  DevBuf<uint32_t> nkfid, nrgbw;
  Directory ndir{};
  int32_t* d_ncount = nullptr;
  ~ChiselDeformState() {
    (void)hipFree(vkeys); (void)hipFree(vseq); (void)hipFree(d_n); (void)hipFree(d_ncount);
    (void)hipFree(ndir.keys); (void)hipFree(ndir.slots); (void)hipFree(ndir.slot_ids);
    list.release(); ids.release(); found.release(); order_slot.release(); chunk_cnt.release(); key0.release();
    key1.release(); val0.release(); val1.release(); first.release(); scratch.release(); kf.release(); rt.release();
    nsdf.release(); nweight.release(); nkfid.release(); nrgbw.release();
  }
};

namespace {

// ---- tracking: the chunks an integrate call visits that the map does not have, with the first visit of each
template <bool kNormals>
__global__ __launch_bounds__(256) void track_visits(Params P, const float* __restrict__ xyz, const float* __restrict__ normals,
                                                    int npoints, const Pose* __restrict__ poses, Directory dir,
                                                    unsigned long long* __restrict__ vkeys, unsigned long long* __restrict__ vseq,
                                                    uint32_t vmask, uint32_t* __restrict__ err) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npoints) return;
  const Pose pose = poses[0];
  Ray ray;
  RayN aux;
  bool walk = true;
  if (kNormals)
    make_ray_normal(P, pose, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], normals[3 * (size_t)i],
                    normals[3 * (size_t)i + 1], normals[3 * (size_t)i + 2], &ray, &aux);
  else
    walk = make_ray(P, pose, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], &ray);
  if (!walk || !ray_in_coord_range(ray)) return;   // (out of range: the integrate itself fails loudly)
  RayCursor cur;
  ray_begin(ray, &cur);
  int vx, vy, vz, lcx = 0, lcy = 0, lcz = 0;
  bool have_last = false;
  unsigned long long step = 0;
  while (ray_next(&cur, &vx, &vy, &vz)) {
    const int cx = vx >> 4, cy = vy >> 4, cz = vz >> 4;   // GetIDAt(GetCentroid(voxel)) (tsdf_chisel_core.hpp, resolve_visit)
    if (!have_last || cx != lcx || cy != lcy || cz != lcz) {
      lcx = cx; lcy = cy; lcz = cz;
      have_last = true;
      unsigned long long key;
      if (dir_find(dir, cx, cy, cz) < 0 && pack_block(cx, cy, cz, &key)) {
        const unsigned long long seq = ((unsigned long long)i << 24) | step;
        uint32_t h = dir_hash(cx, cy, cz, vmask);
        bool placed = false;
        for (uint32_t probe = 0; probe <= vmask; ++probe) {
          unsigned long long k = vkeys[h];
          if (k != key) k = atomicCAS(&vkeys[h], kEmptyKey, key);
          if (k == key || k == kEmptyKey) {
            atomicMin(&vseq[h], seq);
            placed = true;
            break;
          }
          h = (h + 1) & vmask;
        }
        if (!placed) atomicOr(err, 1u);
      }
    }
    ++step;
  }
}

__global__ void track_collect(unsigned long long* __restrict__ vkeys, unsigned long long* __restrict__ vseq, uint32_t cap,
                              unsigned long long* __restrict__ list, uint32_t* __restrict__ n) {
  const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= cap) return;
  const unsigned long long k = vkeys[h];
  if (k == kEmptyKey) return;
  const uint32_t at = atomicAdd(n, 1u);
  list[2 * (size_t)at] = k;
  list[2 * (size_t)at + 1] = vseq[h];
  vkeys[h] = kEmptyKey;
  vseq[h] = kNoSeq;
}

__global__ void lookup_slots(Directory dir, const int32_t* __restrict__ ids, int n, uint32_t* __restrict__ slot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) slot[i] = (uint32_t)dir_find(dir, ids[3 * i], ids[3 * i + 1], ids[3 * i + 2]);   // 0xFFFFFFFF: absent
}

// ---- the deformation
__device__ __forceinline__ int find_transform(const uint32_t* __restrict__ kfids, int n, uint32_t kf) {
  int lo = 0, hi = n - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    const uint32_t k = kfids[mid];
    if (k == kf) return mid;
    if (k < kf) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}

// does voxel v of the chunk in pool slot `slot` move?  (ChunkManager.cpp:953-966)
__device__ __forceinline__ int moving_voxel(const float* __restrict__ weight, const uint32_t* __restrict__ kfid, uint32_t src,
                                            const uint32_t* __restrict__ kfids, int n_map, bool* known) {
  *known = !((double)weight[src] <= 1e-15);
  if (!*known) return -1;
  return find_transform(kfids, n_map, kfid[src]);
}

// One workgroup per old chunk (container order), 256 threads x 16 consecutive voxels.
__global__ __launch_bounds__(256) void deform_count(const uint32_t* __restrict__ order_slot, const float* __restrict__ weight,
                                                    const uint32_t* __restrict__ kfid, const uint32_t* __restrict__ kfids,
                                                    int n_map, uint32_t* __restrict__ chunk_cnt, uint32_t* __restrict__ ctl) {
  __shared__ uint32_t s_cnt, s_disc;
  if (threadIdx.x == 0) { s_cnt = 0; s_disc = 0; }
  __syncthreads();
  const uint32_t base = order_slot[blockIdx.x] * (uint32_t)kChunkVox + threadIdx.x * 16u;
  uint32_t cnt = 0, disc = 0;
  for (int k = 0; k < 16; ++k) {
    bool known;
    const int at = moving_voxel(weight, kfid, base + k, kfids, n_map, &known);
    cnt += at >= 0 ? 1u : 0u;
    disc += (known && at < 0) ? 1u : 0u;
  }
  if (cnt) atomicAdd(&s_cnt, cnt);
  if (disc) atomicAdd(&s_disc, disc);
  __syncthreads();
  if (threadIdx.x == 0) {
    chunk_cnt[blockIdx.x] = s_cnt;
    if (s_disc) atomicAdd(reinterpret_cast<unsigned long long*>(ctl + 2), (unsigned long long)s_disc);
  }
}

__global__ __launch_bounds__(256) void deform_emit(Params P, const uint32_t* __restrict__ order_slot,
                                                   const int32_t* __restrict__ slot_ids, const float* __restrict__ weight,
                                                   const uint32_t* __restrict__ kfid, const uint32_t* __restrict__ kfids,
                                                   const float* __restrict__ Rt, int n_map, const uint32_t* __restrict__ chunk_off,
                                                   Directory ndir, int32_t* __restrict__ ncount, uint32_t* __restrict__ first,
                                                   uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t* __restrict__ ctl) {
  __shared__ uint32_t s_scan[256];
  const uint32_t slot = order_slot[blockIdx.x];
  const uint32_t base = slot * (uint32_t)kChunkVox + threadIdx.x * 16u;
  int at[16];
  uint32_t cnt = 0;
  for (int k = 0; k < 16; ++k) {
    bool known;
    at[k] = moving_voxel(weight, kfid, base + k, kfids, n_map, &known);
    cnt += at[k] >= 0 ? 1u : 0u;
  }
  s_scan[threadIdx.x] = cnt;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const uint32_t add = threadIdx.x >= (unsigned)d ? s_scan[threadIdx.x - d] : 0u;
    __syncthreads();
    s_scan[threadIdx.x] += add;
    __syncthreads();
  }
  uint32_t rec = chunk_off[blockIdx.x] + s_scan[threadIdx.x] - cnt;
  const int cx = slot_ids[3 * slot], cy = slot_ids[3 * slot + 1], cz = slot_ids[3 * slot + 2];
  const float res = P.resolution, inv_res = 1.f / res;   // ChunkManager.cpp:67
  const float ox = (float)(16 * cx) * res, oy = (float)(16 * cy) * res, oz = (float)(16 * cz) * res;   // Chunk.cpp:48
  for (int k = 0; k < 16; ++k) {
    if (at[k] < 0) continue;
    const uint32_t v = threadIdx.x * 16u + k;
    const int lx = v & 15, ly = (v >> 4) & 15, lz = v >> 8;
    const float px = ((float)lx * res + P.half_voxel) + ox;   // centroids[voxelID] + origin  :972
    const float py = ((float)ly * res + P.half_voxel) + oy;
    const float pz = ((float)lz * res + P.half_voxel) + oz;
    const float* R = Rt + 12 * (size_t)at[k];
    float np[3];
    xform(R, R + 9, px, py, pz, np);                           // Rt.R * pos + Rt.t  :973
    const int nx = (int)floorf(np[0] * P.rounding), ny = (int)floorf(np[1] * P.rounding), nz = (int)floorf(np[2] * P.rounding);
    const int gx = (int)floorf(np[0] * inv_res), gy = (int)floorf(np[1] * inv_res), gz = (int)floorf(np[2] * inv_res);
    const int qx = gx - nx * 16, qy = gy - ny * 16, qz = gz - nz * 16;   // Chunk.cpp:101-105
    const int nv = (qz * 16 + qy) * 16 + qx;                            // Chunk.h:90-93
    const int nslot = dir_find_or_insert(ndir, nx, ny, nz, ncount, ctl + 1);   // :981-990 (before the voxel is indexed)
    uint32_t key = 0xFFFFFFFFu;
    if (nslot >= 0) {
      atomicMin(&first[nslot], rec);
      if (nv >= 0 && nv < kChunkVox) key = (uint32_t)nslot * (uint32_t)kChunkVox + (uint32_t)nv;
      else atomicAdd(reinterpret_cast<unsigned long long*>(ctl + 4), 1ull);   // undefined behaviour in the reference
    }
    keys[rec] = key;
    vals[rec] = base + k;
    ++rec;
  }
}

__global__ void deform_init_planes(float* __restrict__ sdf, float* __restrict__ weight, uint32_t* __restrict__ kfid,
                                   uint32_t* __restrict__ rgbw, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  sdf[i] = 99999.0f;
  weight[i] = 0.f;
  kfid[i] = 0u;
  rgbw[i] = 0u;
}

// ColorVoxel::Integrate(r, g, b, 1) (ColorVoxel.h:68-89) on r | g << 8 | b << 16 | weight << 24.
__device__ __forceinline__ uint32_t colour_integrate_one(uint32_t p, uint32_t q) {
  const uint32_t cw = p >> 24;
  if (cw >= 254u) return p;
  const float den = (float)(1u + cw);
  uint32_t out = (cw + 1u) << 24;
  for (int k = 0; k < 3; ++k) {
    const float old = (float)((p >> (8 * k)) & 255u), in = (float)((q >> (8 * k)) & 255u);
    float v = ((float)cw * old + in) / den;
    v = v > 0.0f ? v : 0.0f;
    v = v < 255.0f ? v : 255.0f;
    out |= (uint32_t)(uint8_t)v << (8 * k);
  }
  return out;
}

// One thread per run of equal keys in the sorted records (claimants of one new voxel, in sequence).
__global__ __launch_bounds__(256) void deform_fold(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t n,
                                                   const float* __restrict__ sdf, const float* __restrict__ weight,
                                                   const uint32_t* __restrict__ kfid, const uint32_t* __restrict__ rgbw,
                                                   float* __restrict__ nsdf, float* __restrict__ nweight,
                                                   uint32_t* __restrict__ nkfid, uint32_t* __restrict__ nrgbw) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint32_t key = keys[j];
  if (key == 0xFFFFFFFFu || (j > 0 && keys[j - 1] == key)) return;
  uint32_t src = vals[j];
  float s = sdf[src], w = weight[src];                         // newDistVoxel = distVoxel  :1000-1001
  uint32_t kf = kfid[src], c = rgbw[src];
  for (uint32_t i = j + 1; i < n && keys[i] == key; ++i) {
    src = vals[i];
    const float us = sdf[src], uw = weight[src];
    if ((double)w <= 1e-15) {                                  // (:998 tests the NEW voxel each time)
      s = us; w = uw; kf = kfid[src]; c = rgbw[src];
      continue;
    }
    dist_update(s, w, uw * us, uw);                            // newDistVoxel.Integrate(sdf, weight)  :1006
    kf = kfid[src];
    c = colour_integrate_one(c, rgbw[src]);                    // :1009
  }
  nsdf[key] = s; nweight[key] = w; nkfid[key] = kf; nrgbw[key] = c;
}

__global__ void deform_mesh_kernel(float* __restrict__ vertices, float* __restrict__ normals, const uint32_t* __restrict__ vkfid,
                                   int n, const uint32_t* __restrict__ kfids, const float* __restrict__ Rt, int n_map) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int at = find_transform(kfids, n_map, vkfid[i]);
  if (at < 0) return;                                          // "jumping point"  :1038-1045
  const float* R = Rt + 12 * (size_t)at;
  float v[3];
  xform(R, R + 9, vertices[3 * (size_t)i], vertices[3 * (size_t)i + 1], vertices[3 * (size_t)i + 2], v);   // :1051
  const float nx = normals[3 * (size_t)i], ny = normals[3 * (size_t)i + 1], nz = normals[3 * (size_t)i + 2];
  for (int k = 0; k < 3; ++k) {
    vertices[3 * (size_t)i + k] = v[k];
    normals[3 * (size_t)i + k] = sum3(R[3 * k] * nx, R[3 * k + 1] * ny, R[3 * k + 2] * nz);                 // :1052
  }
}

ChunkIdKey unpack_chunk_key(unsigned long long key) {
  return ChunkIdKey{(int32_t)((key >> 42) & 0x1FFFFFu) - kCoordBias, (int32_t)((key >> 21) & 0x1FFFFFu) - kCoordBias,
                    (int32_t)(key & 0x1FFFFFu) - kCoordBias};
}

}  // namespace

// Before the integrate proper: which chunks will the call visit that the map does not have, and in which order.
static int deform_track_begin(plvs_tsdf_chisel* h, const float* d_xyz, const float* d_normals, int n, int nclouds,
                              const float* d_Twc, hipStream_t s) {
  ChiselDeformState* st = h->dfm;
  PLVS_REQUIRE(nclouds == 1, "a map with deform enabled takes one cloud per integrate call (the reference's chunk order is per call)");
  PLVS_HIP_TRY(h->poses.reserve(1));
  hipLaunchKernelGGL(pose_prep, dim3(1), dim3(64), 0, s, d_Twc, 1, h->poses.p);
  PLVS_HIP_TRY(hipMemsetAsync(st->d_n, 0, 2 * sizeof(uint32_t), s));
  if (d_normals != nullptr)
    hipLaunchKernelGGL(track_visits<true>, dim3(ceil_div((size_t)n, 256)), dim3(256), 0, s, h->P, d_xyz, d_normals, n, h->poses.p,
                       h->dir, st->vkeys, st->vseq, st->vmask, st->d_n + 1);
  else
    hipLaunchKernelGGL(track_visits<false>, dim3(ceil_div((size_t)n, 256)), dim3(256), 0, s, h->P, d_xyz, d_normals, n, h->poses.p,
                       h->dir, st->vkeys, st->vseq, st->vmask, st->d_n + 1);
  PLVS_KERNEL_CHECK();
  const uint32_t cap = st->vmask + 1;
  PLVS_HIP_TRY(st->list.reserve(2 * (size_t)cap));
  hipLaunchKernelGGL(track_collect, dim3(ceil_div((size_t)cap, 256)), dim3(256), 0, s, st->vkeys, st->vseq, cap, st->list.p, st->d_n);
  PLVS_KERNEL_CHECK();
  uint32_t ctl[2];
  PLVS_HIP_TRY(hipMemcpyAsync(ctl, st->d_n, sizeof(ctl), hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  if (ctl[1]) {
    h->poisoned = true;
    plvs::set_error("deform tracking: the call visits more new chunks than the visit set holds (raise max_chunks)");
    return PLVS_ERR_CAPACITY;
  }
  std::vector<unsigned long long> list(2 * (size_t)ctl[0]);
  if (ctl[0]) PLVS_HIP_TRY(hipMemcpy(list.data(), st->list.p, list.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::vector<std::pair<unsigned long long, unsigned long long>> byseq(ctl[0]);
  for (uint32_t i = 0; i < ctl[0]; ++i) byseq[i] = {list[2 * (size_t)i + 1], list[2 * (size_t)i]};
  std::sort(byseq.begin(), byseq.end());
  st->fresh.clear();
  for (const auto& e : byseq) {
    const ChunkIdKey k = unpack_chunk_key(e.second);
    if (st->chunks.find(k) == st->chunks.end()) {
      st->chunks.insert(std::make_pair(k, true));   // CreateChunk, ChunkManager.h:99
      st->fresh.push_back(k);
    }
  }
  return PLVS_OK;
}

// After it: the garbage collection (Chisel.cpp:574-585) — fresh chunks no voxel update reached leave the container.
static int deform_track_end(plvs_tsdf_chisel* h, hipStream_t s) {
  ChiselDeformState* st = h->dfm;
  const int n = (int)st->fresh.size();
  if (n == 0) return PLVS_OK;
  PLVS_HIP_TRY(st->ids.reserve(3 * (size_t)n));
  PLVS_HIP_TRY(st->found.reserve((size_t)n));
  PLVS_HIP_TRY(hipMemcpyAsync(st->ids.p, st->fresh.data(), 3 * (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(lookup_slots, dim3(ceil_div((size_t)n, 256)), dim3(256), 0, s, h->dir, st->ids.p, n, st->found.p);
  PLVS_KERNEL_CHECK();
  std::vector<uint32_t> slot((size_t)n);
  PLVS_HIP_TRY(hipMemcpyAsync(slot.data(), st->found.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  for (int i = 0; i < n; ++i)
    if (slot[i] == 0xFFFFFFFFu) st->chunks.erase(st->fresh[i]);
  st->fresh.clear();
  return PLVS_OK;
}

static void deform_state_clear(plvs_tsdf_chisel* h) {
  if (!h->dfm) return;
  h->dfm->chunks.clear();   // (ChunkManager::Reset: chunks.clear(); the bucket array stays, as in the reference)
  h->dfm->fresh.clear();
}
static void deform_state_free(plvs_tsdf_chisel* h) {
  delete h->dfm;
  h->dfm = nullptr;
}

static void deform_note_created(plvs_tsdf_chisel* h, int cx, int cy, int cz) {   // upload_chunk on a tracked map
  const ChunkIdKey k{cx, cy, cz};
  if (h->dfm->chunks.find(k) == h->dfm->chunks.end()) h->dfm->chunks.insert(std::make_pair(k, true));
}

extern "C" {

int plvs_hip_tsdf_chisel_enable_deform(plvs_tsdf_chisel* h) {
  PLVS_REQUIRE(h, "null handle");
  PLVS_REQUIRE(!h->poisoned, "handle is in a failed state (clear it)");
  if (h->dfm) return PLVS_OK;
  PLVS_REQUIRE(std::max(1, h->prm.shard_count) == 1, "deform is not available on a sharded map");
  PLVS_REQUIRE(h->num_chunks == 0, "enable deform on an empty map: the chunk order is the map's whole history");
  PLVS_REQUIRE(h->prm.max_chunks < (1 << 20), "deform addresses new voxels with 32 bits: max_chunks < 2^20");
  ChiselDeformState* st = new ChiselDeformState;
  const size_t cap = (size_t)h->dir.mask + 1;
  st->vmask = h->dir.mask;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&st->vkeys), cap * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&st->vseq), cap * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&st->d_n), 8 * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&st->d_ncount), 2 * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&st->ndir.keys), cap * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&st->ndir.slots), cap * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&st->ndir.slot_ids), 3 * (size_t)h->prm.max_chunks * sizeof(int32_t));
  if (e == hipSuccess) e = hipMemset(st->vkeys, 0xFF, cap * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemset(st->vseq, 0xFF, cap * sizeof(unsigned long long));
  if (e != hipSuccess) {
    delete st;
    plvs::set_error("enable_deform: %s", hipGetErrorString(e));
    return PLVS_ERR_HIP;
  }
  st->ndir.mask = h->dir.mask;
  st->ndir.max_blocks = h->dir.max_blocks;
  h->dfm = st;
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_chunk_order(plvs_tsdf_chisel* h, int32_t* ids_xyz, int cap, int* n) {
  PLVS_REQUIRE(h && n, "null argument");
  PLVS_FLUSH_QUEUE(h);
  PLVS_REQUIRE(h->dfm, "deform is not enabled on this map");
  *n = (int)h->dfm->chunks.size();
  PLVS_REQUIRE(cap >= *n && (ids_xyz || *n == 0), "id buffer too small");
  size_t k = 0;
  for (const auto& kv : h->dfm->chunks) {
    ids_xyz[3 * k] = kv.first.x; ids_xyz[3 * k + 1] = kv.first.y; ids_xyz[3 * k + 2] = kv.first.z;
    ++k;
  }
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_deform(plvs_tsdf_chisel* h, const uint32_t* kfids, const float* Rt, int n_map,
                                plvs_tsdf_deform_stats* out) {
  PLVS_REQUIRE(h, "null handle");
  PLVS_FLUSH_QUEUE(h);
  PLVS_REQUIRE(!h->poisoned, "handle is in a failed state (clear it)");
  PLVS_REQUIRE(h->dfm, "deform is not enabled on this map (plvs_hip_tsdf_chisel_enable_deform on the empty map)");
  PLVS_REQUIRE(n_map >= 0 && (n_map == 0 || (kfids && Rt)), "bad deformation map");
  for (int i = 1; i < n_map; ++i) PLVS_REQUIRE(kfids[i] > kfids[i - 1], "kfids must be strictly increasing");
  ChiselDeformState* st = h->dfm;
  hipStream_t s = nullptr;
  int rc = halo_drop(h, s);
  if (rc != PLVS_OK) return rc;
  const int norder = (int)st->chunks.size();
  PLVS_REQUIRE(norder == h->num_chunks, "the tracked chunk order does not cover the map (chunks created behind its back)");
  plvs_tsdf_deform_stats res{};
  h->stats = plvs_tsdf_stats{};
  h->last_updated = 0;
  const size_t cap = (size_t)h->dir.mask + 1;
  // the old chunks in container order -> pool slots
  std::vector<int32_t> order(3 * (size_t)norder + 3);
  {
    size_t k = 0;
    for (const auto& kv : st->chunks) { order[3 * k] = kv.first.x; order[3 * k + 1] = kv.first.y; order[3 * k + 2] = kv.first.z; ++k; }
  }
  PLVS_HIP_TRY(st->ids.reserve(3 * (size_t)norder + 3));
  PLVS_HIP_TRY(st->order_slot.reserve((size_t)norder + 1));
  PLVS_HIP_TRY(st->chunk_cnt.reserve((size_t)norder + 1));
  PLVS_HIP_TRY(st->scratch.reserve(scan_scratch_words((size_t)norder + 1)));
  PLVS_HIP_TRY(st->kf.reserve((size_t)n_map + 1));
  PLVS_HIP_TRY(st->rt.reserve(12 * (size_t)n_map + 12));
  PLVS_HIP_TRY(hipMemsetAsync(st->d_n, 0, 8 * sizeof(uint32_t), s));
  uint32_t nrec = 0;
  if (norder > 0) {
    PLVS_HIP_TRY(hipMemcpyAsync(st->ids.p, order.data(), 3 * (size_t)norder * sizeof(int32_t), hipMemcpyHostToDevice, s));
    if (n_map) {
      PLVS_HIP_TRY(hipMemcpyAsync(st->kf.p, kfids, (size_t)n_map * sizeof(uint32_t), hipMemcpyHostToDevice, s));
      PLVS_HIP_TRY(hipMemcpyAsync(st->rt.p, Rt, 12 * (size_t)n_map * sizeof(float), hipMemcpyHostToDevice, s));
    }
    hipLaunchKernelGGL(lookup_slots, dim3(ceil_div((size_t)norder, 256)), dim3(256), 0, s, h->dir, st->ids.p, norder, st->order_slot.p);
    PLVS_KERNEL_CHECK();
    std::vector<uint32_t> slots((size_t)norder);
    PLVS_HIP_TRY(hipMemcpyAsync(slots.data(), st->order_slot.p, (size_t)norder * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    PLVS_HIP_TRY(hipStreamSynchronize(s));
    for (uint32_t v : slots) PLVS_REQUIRE(v != 0xFFFFFFFFu, "the tracked chunk order names a chunk the map does not have");
    hipLaunchKernelGGL(deform_count, dim3(norder), dim3(256), 0, s, st->order_slot.p, h->weight, h->kfid, st->kf.p, n_map,
                       st->chunk_cnt.p, st->d_n);
    PLVS_KERNEL_CHECK();
    PLVS_HIP_TRY(exclusive_scan_u32(st->chunk_cnt.p, st->chunk_cnt.p, (size_t)norder, st->d_n, st->scratch.p, s));
    PLVS_HIP_TRY(hipMemcpyAsync(&nrec, st->d_n, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    PLVS_HIP_TRY(hipStreamSynchronize(s));
  }
  // the new directory
  PLVS_HIP_TRY(hipMemsetAsync(st->ndir.keys, 0xFF, cap * sizeof(unsigned long long), s));
  PLVS_HIP_TRY(hipMemsetAsync(st->ndir.slots, 0xFF, cap * sizeof(int32_t), s));
  PLVS_HIP_TRY(hipMemsetAsync(st->d_ncount, 0, 2 * sizeof(int32_t), s));
  PLVS_HIP_TRY(st->first.reserve((size_t)h->prm.max_chunks));
  PLVS_HIP_TRY(hipMemsetAsync(st->first.p, 0xFF, (size_t)h->prm.max_chunks * sizeof(uint32_t), s));
  int nnew = 0;
  bool in_second = false;
  if (nrec > 0) {
    PLVS_HIP_TRY(st->key0.reserve(nrec)); PLVS_HIP_TRY(st->key1.reserve(nrec));
    PLVS_HIP_TRY(st->val0.reserve(nrec)); PLVS_HIP_TRY(st->val1.reserve(nrec));
    hipLaunchKernelGGL(deform_emit, dim3(norder), dim3(256), 0, s, h->P, st->order_slot.p, h->dir.slot_ids, h->weight, h->kfid,
                       st->kf.p, st->rt.p, n_map, st->chunk_cnt.p, st->ndir, st->d_ncount, st->first.p, st->key0.p, st->val0.p,
                       st->d_n);
    PLVS_KERNEL_CHECK();
    uint32_t ctl[8];
    PLVS_HIP_TRY(hipMemcpyAsync(ctl, st->d_n, sizeof(ctl), hipMemcpyDeviceToHost, s));
    PLVS_HIP_TRY(hipMemcpyAsync(&nnew, st->d_ncount, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    PLVS_HIP_TRY(hipStreamSynchronize(s));
    if (ctl[1] || nnew > h->prm.max_chunks) {   // the old map is untouched
      plvs::set_error("deform: the deformed map needs %d chunks (max_chunks %d)%s", nnew, h->prm.max_chunks,
                      (ctl[1] & kErrCoordRange) ? "; chunk id out of range" : "");
      return PLVS_ERR_CAPACITY;
    }
    res.discarded = (int64_t)(((unsigned long long)ctl[3] << 32) | ctl[2]);
    res.undefined = (int64_t)(((unsigned long long)ctl[5] << 32) | ctl[4]);
    PLVS_HIP_TRY(st->scratch.reserve(radix_scratch_words(nrec)));
    PLVS_HIP_TRY(radix_sort_pairs(st->key0.p, st->val0.p, st->key1.p, st->val1.p, nrec, 0, 32, st->scratch.p, s, &in_second));
  } else if (norder > 0) {
    uint32_t ctl[8];
    PLVS_HIP_TRY(hipMemcpy(ctl, st->d_n, sizeof(ctl), hipMemcpyDeviceToHost));
    res.discarded = (int64_t)(((unsigned long long)ctl[3] << 32) | ctl[2]);
  }
  const size_t nvox_new = (size_t)nnew * kChunkVox;
  if (nnew > 0) {
    PLVS_HIP_TRY(st->nsdf.reserve(nvox_new)); PLVS_HIP_TRY(st->nweight.reserve(nvox_new));
    PLVS_HIP_TRY(st->nkfid.reserve(nvox_new)); PLVS_HIP_TRY(st->nrgbw.reserve(nvox_new));
    hipLaunchKernelGGL(deform_init_planes, dim3(ceil_div(nvox_new, 256)), dim3(256), 0, s, st->nsdf.p, st->nweight.p, st->nkfid.p,
                       st->nrgbw.p, nvox_new);
    PLVS_KERNEL_CHECK();
    hipLaunchKernelGGL(deform_fold, dim3(ceil_div((size_t)nrec, 256)), dim3(256), 0, s, in_second ? st->key1.p : st->key0.p,
                       in_second ? st->val1.p : st->val0.p, nrec, h->sdf, h->weight, h->kfid, h->rgbw, st->nsdf.p, st->nweight.p,
                       st->nkfid.p, st->nrgbw.p);
    PLVS_KERNEL_CHECK();
  }
  // chunks.swap(newChunks)  :1015 — the old slots go back to the initial state, the new chunks take slots 0 .. nnew-1
  const size_t nvox_old = (size_t)h->num_chunks * kChunkVox;
  if (nvox_old) {
    hipLaunchKernelGGL(deform_init_planes, dim3(ceil_div(nvox_old, 256)), dim3(256), 0, s, h->sdf, h->weight, h->kfid, h->rgbw, nvox_old);
    PLVS_KERNEL_CHECK();
  }
  PLVS_HIP_TRY(hipMemcpyAsync(h->dir.keys, st->ndir.keys, cap * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s));
  PLVS_HIP_TRY(hipMemcpyAsync(h->dir.slots, st->ndir.slots, cap * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  if (nnew > 0) {
    PLVS_HIP_TRY(hipMemcpyAsync(h->dir.slot_ids, st->ndir.slot_ids, 3 * (size_t)nnew * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    PLVS_HIP_TRY(hipMemcpyAsync(h->sdf, st->nsdf.p, nvox_new * sizeof(float), hipMemcpyDeviceToDevice, s));
    PLVS_HIP_TRY(hipMemcpyAsync(h->weight, st->nweight.p, nvox_new * sizeof(float), hipMemcpyDeviceToDevice, s));
    PLVS_HIP_TRY(hipMemcpyAsync(h->kfid, st->nkfid.p, nvox_new * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    PLVS_HIP_TRY(hipMemcpyAsync(h->rgbw, st->nrgbw.p, nvox_new * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
  }
  PLVS_HIP_TRY(hipMemcpyAsync(&h->d_ctr->num_chunks, st->d_ncount, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  // the container after the swap: a fresh map filled in first-claim order
  std::vector<uint32_t> first((size_t)nnew);
  std::vector<int32_t> nids(3 * (size_t)nnew + 3);
  if (nnew > 0) {
    PLVS_HIP_TRY(hipMemcpyAsync(first.data(), st->first.p, (size_t)nnew * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    PLVS_HIP_TRY(hipMemcpyAsync(nids.data(), st->ndir.slot_ids, 3 * (size_t)nnew * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  }
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  std::vector<std::pair<uint32_t, int>> claim((size_t)nnew);
  for (int i = 0; i < nnew; ++i) claim[i] = {first[i], i};
  std::sort(claim.begin(), claim.end());
  ChunkOrder next;
  for (const auto& c : claim)
    next.insert(std::make_pair(ChunkIdKey{nids[3 * (size_t)c.second], nids[3 * (size_t)c.second + 1], nids[3 * (size_t)c.second + 2]}, true));
  st->chunks.swap(next);
  st->fresh.clear();
  h->num_chunks = nnew;
  res.new_chunks = nnew;
  res.moved = (int64_t)nrec;
  if (out) *out = res;
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_deform_mesh_dev(float* d_vertices, float* d_normals, const uint32_t* d_vertex_kfid, int n,
                                         const uint32_t* d_kfids, const float* d_Rt, int n_map, void* stream) {
  PLVS_REQUIRE(n >= 0 && n_map >= 0, "bad sizes");
  if (n == 0 || n_map == 0) return PLVS_OK;
  PLVS_REQUIRE(d_vertices && d_normals && d_vertex_kfid && d_kfids && d_Rt, "null device pointer");
  hipLaunchKernelGGL(deform_mesh_kernel, dim3(ceil_div((size_t)n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), d_vertices,
                     d_normals, d_vertex_kfid, n, d_kfids, d_Rt, n_map);
  PLVS_KERNEL_CHECK();
  return PLVS_OK;
}

int plvs_hip_tsdf_chisel_deform_mesh(float* vertices, float* normals, const uint32_t* vertex_kfid, int n, const uint32_t* kfids,
                                     const float* Rt, int n_map) {
  PLVS_REQUIRE(n >= 0 && n_map >= 0, "bad sizes");
  if (n == 0 || n_map == 0) return PLVS_OK;
  PLVS_REQUIRE(vertices && normals && vertex_kfid && kfids && Rt, "null argument");
  for (int i = 1; i < n_map; ++i) PLVS_REQUIRE(kfids[i] > kfids[i - 1], "kfids must be strictly increasing");
  DevBuf<float> v, nr, rt;
  DevBuf<uint32_t> vk, kf;
  int rc = PLVS_OK;
  hipError_t e = v.reserve(3 * (size_t)n);
  if (e == hipSuccess) e = nr.reserve(3 * (size_t)n);
  if (e == hipSuccess) e = vk.reserve((size_t)n);
  if (e == hipSuccess) e = kf.reserve((size_t)n_map);
  if (e == hipSuccess) e = rt.reserve(12 * (size_t)n_map);
  if (e == hipSuccess) e = hipMemcpy(v.p, vertices, 3 * (size_t)n * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(nr.p, normals, 3 * (size_t)n * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(vk.p, vertex_kfid, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(kf.p, kfids, (size_t)n_map * sizeof(uint32_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(rt.p, Rt, 12 * (size_t)n_map * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    rc = plvs_hip_tsdf_chisel_deform_mesh_dev(v.p, nr.p, vk.p, n, kf.p, rt.p, n_map, nullptr);
    if (rc == PLVS_OK) e = hipMemcpy(vertices, v.p, 3 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
    if (rc == PLVS_OK && e == hipSuccess) e = hipMemcpy(normals, nr.p, 3 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
  }
  v.release(); nr.release(); vk.release(); kf.release(); rt.release();
  if (e != hipSuccess) {
    plvs::set_error("deform_mesh: %s", hipGetErrorString(e));
    return PLVS_ERR_HIP;
  }
  return rc;
}

}  // extern "C"
