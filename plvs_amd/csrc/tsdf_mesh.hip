// Surface extraction from the chisel map: the meshes PointCloudMapChisel::UpdateMap turns into
// PLVS's output cloud.
//
// Replaces ChunkManager::RecomputeMesh (Thirdparty/open_chisel/src/ChunkManager.cpp:116-170) for a
// list of chunks — Chisel::UpdateMeshes -> RecomputeMeshesParallel(meshesToUpdate) (Chisel.cpp:57-65,
// ChunkManager.cpp:202-222):
//   GenerateMesh                 :577-660   marching cubes over the chunk, USE_KFID_MESHING 1 /
//                                           USE_KFID_VERTICES 0 (:39-40): one kfid per cube, corner 0's
//   ColorizeMesh                 :860-872   InterpolateColor :718-805, Chunk::GetColorAt Chunk.cpp:137
//   ComputeNormalsFromGradients  :840-858   GetSDFAndGradient :663-690, GetSDF :692-716
//   MarchingCubes::MeshCube / InterpolateVertex   include/open_chisel/marching_cubes/MarchingCubes.h:110-245
//
// The reference walks the 4096 voxels of a chunk in a fixed order (15^3 interior, then the max-X,
// max-Y and max-Z planes) and appends up to five triangles per voxel.  Here:
//   mesh_count   one thread per (chunk, voxel in that order): number of vertices it emits
//   scan         exclusive scan of the counts = the position of every voxel's first vertex
//   mesh_emit    same thread layout: vertices, triangle normals, kfids at those positions
//   mesh_shade   one thread per vertex: interpolated colour and gradient normal (they read the map
//                through the chunk directory, neighbours included)
// so the vertex order inside a chunk and the order of the chunks are the reference's / the caller's.
// The map stays in HBM; only the finished mesh crosses PCIe.  Arithmetic follows the reference
// expression by expression (quirks included: the flat-edge vertex, voxel indices used as metres in the colour look-up, the
// linear-id range check), no FMA contraction.
#include <vector>

#include "common.hpp"
#include "device_utils.hpp"
#include "tsdf_chisel_core.hpp"
#include "tsdf_chisel_view.hpp"

namespace {

using plvs::tsdf::ChiselMapView;
using plvs::tsdf::dir_find;

constexpr int kChunkVox = 4096;
constexpr int kMeshThreads = 256;

__constant__ int8_t c_triangle_table[256 * 16] = {
#include "mc_table.inc"
};
__constant__ int8_t c_edge_pairs[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6},
                                           {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};

// cubeIndexOffsets, ChunkManager.cpp:84-86
__device__ __forceinline__ int cube_dx(int i) { return (0x66 >> i) & 1; }   // 0 1 1 0 0 1 1 0
__device__ __forceinline__ int cube_dy(int i) { return (0xCC >> i) & 1; }   // 0 0 1 1 0 0 1 1
__device__ __forceinline__ int cube_dz(int i) { return (0xF0 >> i) & 1; }   // 0 0 0 0 1 1 1 1

// k-th voxel of GenerateMesh's walk (:590-659).
__device__ __forceinline__ void walk_voxel(int k, int* x, int* y, int* z) {
  if (k < 3375) {            // interior: z, y, x < 15
    *x = k % 15;
    *y = (k / 15) % 15;
    *z = k / 225;
  } else if (k < 3375 + 240) {   // max-X plane: z < 15, y < 16
    k -= 3375;
    *x = 15;
    *y = k % 16;
    *z = k / 16;
  } else if (k < 3375 + 240 + 225) {   // max-Y plane: z < 15, x < 15
    k -= 3375 + 240;
    *x = k % 15;
    *y = 15;
    *z = k / 15;
  } else {                    // max-Z plane: y < 16, x < 16
    k -= 3375 + 240 + 225;
    *x = k % 16;
    *y = k / 16;
    *z = 15;
  }
}

// Chunk id -> pool slot.  On a sharded map a chunk of another rank is looked for among the imported ghosts; one that
// has not been brought in yet is noted in the miss set (the caller fetches it and meshes again).
__device__ __forceinline__ int find_chunk(const ChiselMapView& m, int cx, int cy, int cz) {
  const int s = dir_find(m.dir, cx, cy, cz);
  if (s >= 0 || m.shard_count <= 1) return s;
  if (plvs::chisel::shard_of(plvs::chisel::chunk_hash(cx, cy, cz), m.shard_count) == m.shard_rank) return -1;
  if (m.ghost.keys != nullptr) {
    const int g = dir_find(m.ghost, cx, cy, cz);
    if (g >= 0) return g;
    if (g == plvs::tsdf::kGhostAbsent) return -1;
  }
  unsigned long long key;
  if (!plvs::tsdf::pack_block(cx, cy, cz, &key)) return -1;
  uint32_t hsh = plvs::tsdf::dir_hash(cx, cy, cz, m.miss_mask);
  for (uint32_t probe = 0; probe <= m.miss_mask; ++probe) {
    unsigned long long cur = m.miss_keys[hsh];
    if (cur == key) break;
    if (cur == plvs::tsdf::kEmptyKey) {
      cur = atomicCAS(&m.miss_keys[hsh], plvs::tsdf::kEmptyKey, key);
      if (cur == plvs::tsdf::kEmptyKey) {
        const uint32_t at = atomicAdd(m.miss_count, 1u);
        if (at < m.miss_cap) {
          m.miss_ids[3 * at] = cx;
          m.miss_ids[3 * at + 1] = cy;
          m.miss_ids[3 * at + 2] = cz;
        }
        break;
      }
      if (cur == key) break;
    }
    hsh = (hsh + 1) & m.miss_mask;
  }
  return -1;
}

struct Cube {
  float sdf[8];
  uint32_t kfid;
  int index;   // vertex configuration; 0 = nothing to emit (also: an unobserved corner)
};

// ExtractInsideVoxelMeshKfid / ExtractBorderVoxelMeshKfid (:438-575): the eight corners, through
// the neighbour chunks where the cube leaves this one.
__device__ __forceinline__ void load_cube(const ChiselMapView& m, int slot, int cx, int cy, int cz, int x,
                                          int y, int z, Cube* c) {
  c->index = 0;
  c->kfid = 0;
  int nslot[8];   // neighbour slot per (ox, oy, oz) combination, looked up lazily
#pragma unroll
  for (int i = 0; i < 8; ++i) nslot[i] = -2;
  nslot[0] = slot;
  int index = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int vx = x + cube_dx(i), vy = y + cube_dy(i), vz = z + cube_dz(i);
    const int ox = vx >> 4, oy = vy >> 4, oz = vz >> 4;
    vx &= 15;
    vy &= 15;
    vz &= 15;
    const int which = ox | (oy << 1) | (oz << 2);
    int s = slot;
    if (which != 0) {
      int cached = -2;
#pragma unroll
      for (int j = 1; j < 8; ++j)
        if (j == which) cached = nslot[j];
      if (cached == -2) {
        cached = find_chunk(m, cx + ox, cy + oy, cz + oz);
#pragma unroll
        for (int j = 1; j < 8; ++j)
          if (j == which) nslot[j] = cached;
      }
      s = cached;
      if (s < 0) return;   // allNeighborsObserved = false
    }
    const size_t id = (size_t)s * kChunkVox + (size_t)((vz * 16 + vy) * 16 + vx);
    if ((double)m.weight[id] <= 1e-15) return;
    const float d = m.sdf[id];
    c->sdf[i] = d;
    if (i == 0) c->kfid = m.kfid[id];
    index |= (d < 0) ? (1 << i) : 0;
  }
  c->index = index;
}

__device__ __forceinline__ int table_vertices(int index) {
  int n = 0;
  while (n < 15 && c_triangle_table[index * 16 + n] != -1) n += 3;
  return n;
}

__global__ __launch_bounds__(kMeshThreads) void mesh_count(ChiselMapView m, const int32_t* __restrict__ ids,
                                                          const int32_t* __restrict__ slots, int nchunks,
                                                          uint32_t* __restrict__ counts) {
  const int g = blockIdx.x * kMeshThreads + threadIdx.x;
  if (g >= nchunks * kChunkVox) return;
  const int c = g / kChunkVox, k = g % kChunkVox;
  const int slot = slots[c];
  uint32_t n = 0;
  if (slot >= 0) {
    int x, y, z;
    walk_voxel(k, &x, &y, &z);
    Cube cube;
    load_cube(m, slot, ids[3 * c], ids[3 * c + 1], ids[3 * c + 2], x, y, z, &cube);
    n = (uint32_t)table_vertices(cube.index);
  }
  counts[g] = n;
}

// Chunk ids -> pool slots (-1 = the chunk does not exist: RecomputeMesh returns early, :126-130).
__global__ void mesh_slots(ChiselMapView m, const int32_t* __restrict__ ids, int nchunks, int32_t* __restrict__ slots) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < nchunks) slots[c] = dir_find(m.dir, ids[3 * c], ids[3 * c + 1], ids[3 * c + 2]);
}

__global__ __launch_bounds__(kMeshThreads) void mesh_emit(ChiselMapView m, const int32_t* __restrict__ ids,
                                                         const int32_t* __restrict__ slots, int nchunks,
                                                         const uint32_t* __restrict__ first, uint32_t cap,
                                                         float* __restrict__ vertices, float* __restrict__ normals,
                                                         uint32_t* __restrict__ kfids) {
  const int g = blockIdx.x * kMeshThreads + threadIdx.x;
  if (g >= nchunks * kChunkVox) return;
  const int c = g / kChunkVox, k = g % kChunkVox;
  const int slot = slots[c];
  if (slot < 0) return;
  int x, y, z;
  walk_voxel(k, &x, &y, &z);
  Cube cube;
  const int cx = ids[3 * c], cy = ids[3 * c + 1], cz = ids[3 * c + 2];
  load_cube(m, slot, cx, cy, cz, x, y, z, &cube);
  if (cube.index == 0) return;
  const float res = m.resolution;
  const float half = res * 0.5f;   // halfVoxel, ChunkManager.cpp:68
  // centroids[i] + chunk->GetOrigin()  (:78, :601; Chunk.cpp:48)
  const float bx = ((float)x * res + half) + (float)(16 * cx) * res;
  const float by = ((float)y * res + half) + (float)(16 * cy) * res;
  const float bz = ((float)z * res + half) + (float)(16 * cz) * res;
  // InterpolateEdgeVertices (MarchingCubes.h:206-220)
  float ex[12], ey[12], ez[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) {
    const int e0 = c_edge_pairs[e][0], e1 = c_edge_pairs[e][1];
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j == e0) s0 = cube.sdf[j];
      if (j == e1) s1 = cube.sdf[j];
    }
    ex[e] = ey[e] = ez[e] = 0.f;
    if ((s0 < 0 && s1 >= 0) || (s0 >= 0 && s1 < 0)) {
      const float x0 = bx + (float)cube_dx(e0) * res, y0 = by + (float)cube_dy(e0) * res, z0 = bz + (float)cube_dz(e0) * res;
      const float x1 = bx + (float)cube_dx(e1) * res, y1 = by + (float)cube_dy(e1) * res, z1 = bz + (float)cube_dz(e1) * res;
      const float diff = s0 - s1;
      if (fabsf(diff) < 1e-6f) {   // InterpolateVertex :224-235: vertex1 + 0.5f * vertex2
        ex[e] = x0 + 0.5f * x1;
        ey[e] = y0 + 0.5f * y1;
        ez[e] = z0 + 0.5f * z1;
      } else {
        const float t = s0 / diff;
        ex[e] = x0 + t * (x1 - x0);
        ey[e] = y0 + t * (y1 - y0);
        ez[e] = z0 + t * (z1 - z0);
      }
    }
  }
  uint32_t pos = first[g];
  for (int col = 0; col < 15 && c_triangle_table[cube.index * 16 + col] != -1; col += 3, pos += 3) {
    if (pos + 3 > cap) return;
    float p[3][3];
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const int e = c_triangle_table[cube.index * 16 + col + 2 - v];   // emitted as col+2, col+1, col
      float px = 0.f, py = 0.f, pz = 0.f;
#pragma unroll
      for (int j = 0; j < 12; ++j)
        if (j == e) {
          px = ex[j];
          py = ey[j];
          pz = ez[j];
        }
      p[v][0] = px;
      p[v][1] = py;
      p[v][2] = pz;
    }
    const float ax = p[1][0] - p[0][0], ay = p[1][1] - p[0][1], az = p[1][2] - p[0][2];
    const float bxx = p[2][0] - p[0][0], byy = p[2][1] - p[0][1], bzz = p[2][2] - p[0][2];
    float nx = ay * bzz - az * byy, ny = az * bxx - ax * bzz, nz = ax * byy - ay * bxx;
    const float zz = nx * nx + (ny * ny + nz * nz);
    if (zz > 0.0f) {   // normalized()
      const float s = sqrtf(zz);
      nx /= s;
      ny /= s;
      nz /= s;
    }
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const size_t o = 3 * (size_t)(pos + v);
      vertices[o] = p[v][0];
      vertices[o + 1] = p[v][1];
      vertices[o + 2] = p[v][2];
      normals[o] = nx;
      normals[o + 1] = ny;
      normals[o + 2] = nz;
      kfids[pos + v] = cube.kfid;
    }
  }
}

// GetChunkAt + Chunk::GetVoxelID(rel) as GetColorVoxel / GetSDF use them (:692-716, :818-836): the
// linear voxel id of `pos` inside the chunk that contains it, or -1.
__device__ __forceinline__ long voxel_at(const ChiselMapView& m, float px, float py, float pz, float rounding,
                                         float inv) {
  const int cx = (int)floorf(px * rounding), cy = (int)floorf(py * rounding), cz = (int)floorf(pz * rounding);
  const int slot = find_chunk(m, cx, cy, cz);
  if (slot < 0) return -1;
  const float res = m.resolution;
  const int vx = (int)floorf((px - (float)(16 * cx) * res) * inv);
  const int vy = (int)floorf((py - (float)(16 * cy) * res) * inv);
  const int vz = (int)floorf((pz - (float)(16 * cz) * res) * inv);
  const int id = (vz * 16 + vy) * 16 + vx;   // only the LINEAR id is range-checked there
  if (id < 0 || id >= kChunkVox) return -1;
  return (long)slot * kChunkVox + id;
}

__device__ __forceinline__ bool sdf_at(const ChiselMapView& m, float px, float py, float pz, float rounding,
                                       float inv, double* dist) {
  const long id = voxel_at(m, px, py, pz, rounding, inv);
  if (id < 0) return false;
  if (!((double)m.weight[id] > 1e-12)) return false;
  *dist = (double)m.sdf[id];
  return true;
}

__global__ __launch_bounds__(kMeshThreads) void mesh_shade(ChiselMapView m, uint32_t nverts,
                                                          const float* __restrict__ vertices,
                                                          float* __restrict__ normals, float* __restrict__ colors) {
  const uint32_t i = blockIdx.x * kMeshThreads + threadIdx.x;
  if (i >= nverts) return;
  const float res = m.resolution;
  const float inv = 1.f / res;                      // invVoxelResolutionMeters, ChunkManager.cpp:67
  const float rounding = 1.0f / (16 * res);         // roundingFactorX, :91
  const float x = vertices[3 * (size_t)i], y = vertices[3 * (size_t)i + 1], z = vertices[3 * (size_t)i + 2];

  // ---- InterpolateColor (:718-805).  The eight look-ups are made at the voxel INDICES used as
  // positions in metres, as the reference does.
  {
    const int x_0 = (int)floorf(x * inv), y_0 = (int)floorf(y * inv), z_0 = (int)floorf(z * inv);
    const int x_1 = x_0 + 1, y_1 = y_0 + 1, z_1 = z_0 + 1;
    long v[8];   // 000 001 011 111 110 100 010 101
    v[0] = voxel_at(m, (float)x_0, (float)y_0, (float)z_0, rounding, inv);
    v[1] = voxel_at(m, (float)x_0, (float)y_0, (float)z_1, rounding, inv);
    v[2] = voxel_at(m, (float)x_0, (float)y_1, (float)z_1, rounding, inv);
    v[3] = voxel_at(m, (float)x_1, (float)y_1, (float)z_1, rounding, inv);
    v[4] = voxel_at(m, (float)x_1, (float)y_1, (float)z_0, rounding, inv);
    v[5] = voxel_at(m, (float)x_1, (float)y_0, (float)z_0, rounding, inv);
    v[6] = voxel_at(m, (float)x_0, (float)y_1, (float)z_0, rounding, inv);
    v[7] = voxel_at(m, (float)x_1, (float)y_0, (float)z_1, rounding, inv);
    bool all = true;
#pragma unroll
    for (int j = 0; j < 8; ++j) all = all && v[j] >= 0;
    float cr = 0.f, cg = 0.f, cb = 0.f;
    if (!all) {
      // chunk->GetColorAt(colorPos), Chunk.cpp:137-155
      const int cx = (int)floorf(x * rounding), cy = (int)floorf(y * rounding), cz = (int)floorf(z * rounding);
      const int slot = find_chunk(m, cx, cy, cz);
      if (slot >= 0) {
        const float ox = (float)(16 * cx) * res, oy = (float)(16 * cy) * res, oz = (float)(16 * cz) * res;
        const float size = 16.0f * res;
        if (x >= ox && y >= oy && z >= oz && x <= ox + size && y <= oy + size && z <= oz + size) {
          const int ix = (int)((x - ox) * inv), iy = (int)((y - oy) * inv), iz = (int)((z - oz) * inv);
          if (ix >= 0 && ix < 16 && iy >= 0 && iy < 16 && iz >= 0 && iz < 16) {
            const uint32_t col = m.rgbw[(size_t)slot * kChunkVox + (size_t)((iz * 16 + iy) * 16 + ix)];
            const float invMaxVal = 1.f / 255.0f;
            cr = (float)(col & 255u) * invMaxVal;
            cg = (float)((col >> 8) & 255u) * invMaxVal;
            cb = (float)((col >> 16) & 255u) * invMaxVal;
          }
        }
      }
    } else {
      uint32_t w[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = m.rgbw[v[j]];
      const float xd = (x - (float)x_0) / (float)(x_1 - x_0);
      const float yd = (y - (float)y_0) / (float)(y_1 - y_0);
      const float zd = (z - (float)z_0) / (float)(z_1 - z_0);
      float out[3];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const int sh = 8 * ch;
        const float f000 = (float)((w[0] >> sh) & 255u), f001 = (float)((w[1] >> sh) & 255u);
        const float f011 = (float)((w[2] >> sh) & 255u), f111 = (float)((w[3] >> sh) & 255u);
        const float f110 = (float)((w[4] >> sh) & 255u), f100 = (float)((w[5] >> sh) & 255u);
        const float f010 = (float)((w[6] >> sh) & 255u), f101 = (float)((w[7] >> sh) & 255u);
        const float c_00 = f000 * (1 - xd) + f100 * xd;
        const float c_10 = f010 * (1 - xd) + f110 * xd;
        const float c_01 = f001 * (1 - xd) + f101 * xd;
        const float c_11 = f011 * (1 - xd) + f111 * xd;
        const float c_0 = c_00 * (1 - yd) + c_10 * yd;
        const float c_1 = c_01 * (1 - yd) + c_11 * yd;
        const float cc = c_0 * (1 - zd) + c_1 * zd;
        out[ch] = cc / 255.0f;
      }
      cr = out[0];
      cg = out[1];
      cb = out[2];
    }
    colors[3 * (size_t)i] = cr;
    colors[3 * (size_t)i + 1] = cg;
    colors[3 * (size_t)i + 2] = cb;
  }

  // ---- ComputeNormalsFromGradients (:840-858)
  {
    const float half = 0.5f * res;   // halfVoxelResolutionMeters, :89
    const float fx = floorf(x * inv) * res + half, fy = floorf(y * inv) * res + half, fz = floorf(z * inv) * res + half;
    double d0, xp, yp, zp, xm, ym, zm;
    if (!sdf_at(m, fx, fy, fz, rounding, inv, &d0)) return;
    if (!sdf_at(m, fx + res, fy, fz, rounding, inv, &xp)) return;
    if (!sdf_at(m, fx, fy + res, fz, rounding, inv, &yp)) return;
    if (!sdf_at(m, fx, fy, fz + res, rounding, inv, &zp)) return;
    if (!sdf_at(m, fx - res, fy, fz, rounding, inv, &xm)) return;
    if (!sdf_at(m, fx, fy - res, fz, rounding, inv, &ym)) return;
    if (!sdf_at(m, fx, fy, fz - res, rounding, inv, &zm)) return;
    float gx = (float)(xp - xm), gy = (float)(yp - ym), gz = (float)(zp - zm);
    const float zz = gx * gx + (gy * gy + gz * gz);
    if (zz > 0.0f) {   // grad->normalize()
      const float s = sqrtf(zz);
      gx /= s;
      gy /= s;
      gz /= s;
    }
    const float mag = sqrtf(gx * gx + (gy * gy + gz * gz));
    if ((double)mag > 1e-12) {
      const float r = 1.0f / mag;
      normals[3 * (size_t)i] = gx * r;
      normals[3 * (size_t)i + 1] = gy * r;
      normals[3 * (size_t)i + 2] = gz * r;
    }
  }
}

// Scratch of the meshing calls, kept with the map (ChiselMapView::ext) so that a call per keyframe does not
// allocate: the buffers only grow.
struct MeshScratch {
  plvs::DevBuf<int32_t> ids, slots;
  plvs::DevBuf<uint32_t> counts, first, scan, total, kfids;
  plvs::DevBuf<float> vertices, normals, colors;
  static void destroy(void* p) {
    MeshScratch* s = static_cast<MeshScratch*>(p);
    s->ids.release(); s->slots.release(); s->counts.release(); s->first.release(); s->scan.release();
    s->total.release(); s->kfids.release(); s->vertices.release(); s->normals.release(); s->colors.release();
    delete s;
  }
};

}  // namespace

// probe: run every stage on the device and report only how many foreign chunks were missed (no host outputs, no
// capacity limit) — what settles the halo of a sharded map before the sizing and the filling call.
static int mesh_run(plvs_tsdf_chisel* h, const int32_t* chunk_ids_xyz, int nchunks, float* vertices, float* normals,
                    float* colors, uint32_t* kfids, int capacity, int32_t* chunk_first, int* nvertices, bool probe,
                    uint32_t* missing_out) {
  PLVS_REQUIRE(h != nullptr && nvertices != nullptr, "null handle / nvertices");
  PLVS_REQUIRE(nchunks >= 0 && capacity >= 0, "negative size");
  *nvertices = 0;
  if (chunk_first != nullptr)
    for (int c = 0; c <= nchunks; ++c) chunk_first[c] = 0;
  ChiselMapView m;
  if (!plvs::tsdf::chisel_map_view(h, &m)) {   // (also empties the miss set of a sharded map)
    plvs::set_error("mesh_chunks: the map handle is unusable");
    return PLVS_ERR_INVALID_ARG;
  }
  if (nchunks == 0) return PLVS_OK;
  PLVS_REQUIRE(chunk_ids_xyz != nullptr && chunk_first != nullptr, "null chunk list / chunk_first");
  PLVS_REQUIRE(nchunks <= (1 << 18), "too many chunks in one call");

  if (*m.ext == nullptr) {
    *m.ext = new MeshScratch();
    *m.ext_free = &MeshScratch::destroy;
  }
  MeshScratch& sc = *static_cast<MeshScratch*>(*m.ext);
  const size_t nvox = (size_t)nchunks * kChunkVox;
  hipStream_t s = nullptr;   // the map's calls are synchronous on return; the default stream orders after them
  PLVS_HIP_TRY(sc.ids.reserve(3 * (size_t)nchunks));
  PLVS_HIP_TRY(sc.slots.reserve((size_t)nchunks));
  PLVS_HIP_TRY(sc.counts.reserve(nvox));
  PLVS_HIP_TRY(sc.first.reserve(nvox));
  PLVS_HIP_TRY(sc.scan.reserve(plvs::scan_scratch_words(nvox)));
  PLVS_HIP_TRY(sc.total.reserve(1));
  PLVS_HIP_TRY(hipMemcpyAsync(sc.ids.p, chunk_ids_xyz, sizeof(int32_t) * 3 * (size_t)nchunks, hipMemcpyHostToDevice, s));
  mesh_slots<<<plvs::ceil_div((size_t)nchunks, 256), 256, 0, s>>>(m, sc.ids.p, nchunks, sc.slots.p);
  mesh_count<<<plvs::ceil_div(nvox, kMeshThreads), kMeshThreads, 0, s>>>(m, sc.ids.p, sc.slots.p, nchunks, sc.counts.p);
  PLVS_KERNEL_CHECK();
  PLVS_HIP_TRY(plvs::exclusive_scan_u32(sc.counts.p, sc.first.p, nvox, sc.total.p, sc.scan.p, s));
  uint32_t total = 0, missing = 0;
  std::vector<uint32_t> firsts((size_t)nchunks);
  PLVS_HIP_TRY(hipMemcpyAsync(&total, sc.total.p, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  if (m.miss_count != nullptr) PLVS_HIP_TRY(hipMemcpyAsync(&missing, m.miss_count, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipMemcpy2DAsync(firsts.data(), sizeof(uint32_t), sc.first.p, sizeof(uint32_t) * kChunkVox, sizeof(uint32_t),
                                (size_t)nchunks, hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  *nvertices = (int)total;
  for (int c = 0; c < nchunks; ++c) chunk_first[c] = (int32_t)firsts[(size_t)c];
  chunk_first[nchunks] = (int32_t)total;
  if (missing_out != nullptr) *missing_out = missing;
  if (missing > 0) {
    plvs::set_error("mesh_chunks: %u chunks of other ranks are needed (plvs_hip_tsdf_chisel_halo_missing lists them)", missing);
    return PLVS_ERR_HALO;
  }
  if (!probe && total > (uint32_t)capacity) {
    plvs::set_error("mesh_chunks: %u vertices exceed the capacity %d (call again with room for *nvertices)", total, capacity);
    return PLVS_ERR_CAPACITY;
  }
  if (total == 0) return PLVS_OK;
  PLVS_REQUIRE(probe || (vertices && normals && colors && kfids), "null output array");
  PLVS_HIP_TRY(sc.vertices.reserve(3 * (size_t)total));
  PLVS_HIP_TRY(sc.normals.reserve(3 * (size_t)total));
  PLVS_HIP_TRY(sc.colors.reserve(3 * (size_t)total));
  PLVS_HIP_TRY(sc.kfids.reserve((size_t)total));
  mesh_emit<<<plvs::ceil_div(nvox, kMeshThreads), kMeshThreads, 0, s>>>(m, sc.ids.p, sc.slots.p, nchunks, sc.first.p, total,
                                                                       sc.vertices.p, sc.normals.p, sc.kfids.p);
  mesh_shade<<<plvs::ceil_div((size_t)total, kMeshThreads), kMeshThreads, 0, s>>>(m, total, sc.vertices.p, sc.normals.p,
                                                                                 sc.colors.p);
  PLVS_KERNEL_CHECK();
  if (!probe) {
    PLVS_HIP_TRY(hipMemcpyAsync(vertices, sc.vertices.p, sizeof(float) * 3 * (size_t)total, hipMemcpyDeviceToHost, s));
    PLVS_HIP_TRY(hipMemcpyAsync(normals, sc.normals.p, sizeof(float) * 3 * (size_t)total, hipMemcpyDeviceToHost, s));
    PLVS_HIP_TRY(hipMemcpyAsync(colors, sc.colors.p, sizeof(float) * 3 * (size_t)total, hipMemcpyDeviceToHost, s));
    PLVS_HIP_TRY(hipMemcpyAsync(kfids, sc.kfids.p, sizeof(uint32_t) * (size_t)total, hipMemcpyDeviceToHost, s));
  }
  if (m.miss_count != nullptr) PLVS_HIP_TRY(hipMemcpyAsync(&missing, m.miss_count, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  if (missing_out != nullptr) *missing_out = missing;
  if (missing > 0) {   // the colour / gradient look-ups of the new vertices reached chunks that are not here yet
    plvs::set_error("mesh_chunks: %u chunks of other ranks are needed (plvs_hip_tsdf_chisel_halo_missing lists them)", missing);
    return PLVS_ERR_HALO;
  }
  return PLVS_OK;
}

extern "C" int plvs_hip_tsdf_chisel_mesh_chunks(plvs_tsdf_chisel* h, const int32_t* chunk_ids_xyz, int nchunks,
                                                float* vertices, float* normals, float* colors, uint32_t* kfids,
                                                int capacity, int32_t* chunk_first, int* nvertices) {
  if (h) {   // (queued clouds are part of the map that is meshed)
    const int rc = plvs_hip_tsdf_chisel_flush(h);
    if (rc != PLVS_OK) return rc;
  }
  return mesh_run(h, chunk_ids_xyz, nchunks, vertices, normals, colors, kfids, capacity, chunk_first, nvertices, false,
                  nullptr);
}

extern "C" int plvs_hip_tsdf_chisel_mesh_probe(plvs_tsdf_chisel* h, const int32_t* chunk_ids_xyz, int nchunks,
                                               int* nmissing) {
  PLVS_REQUIRE(nmissing != nullptr, "null nmissing");
  if (h) {
    const int rc = plvs_hip_tsdf_chisel_flush(h);
    if (rc != PLVS_OK) return rc;
  }
  *nmissing = 0;
  std::vector<int32_t> first((size_t)(nchunks > 0 ? nchunks : 0) + 1);
  int nv = 0;
  uint32_t missing = 0;
  const int rc = mesh_run(h, chunk_ids_xyz, nchunks, nullptr, nullptr, nullptr, nullptr, 0, first.data(), &nv, true, &missing);
  *nmissing = (int)missing;
  return rc == PLVS_ERR_HALO ? PLVS_OK : rc;
}
