// Surface extraction from the voxblox map: the meshes PointCloudMapVoxblox::UpdateMap turns into PLVS's
// output cloud.
//
// Replaces MeshIntegrator<TsdfVoxel>::updateMeshForBlock
// (Thirdparty/voxblox/include/voxblox/mesh/mesh_integrator.h:231-251) for a list of blocks — what
// TsdfServer::updateMesh (Thirdparty/voxblox_server/src/tsdf_server.cc:775-787) does through
// generateMesh(only_mesh_updated_blocks, clear_updated_flag) (:111-139) for the blocks integrated into since the last call,
// when PointCloudMapVoxblox::UpdateMap calls it (src/PointCloudMapVoxblox.cc:168):
//   extractBlockMesh                      :165-229   the walk over the block
//   extractMeshInsideBlock / OnBorder     :253-346   eight corners, +x / +y / +z neighbour blocks, getSdfIfValid
//   MarchingCubes::meshCube               mesh/marching_cubes.h:66-102, interpolateEdgeVertices / interpolateVertex :117-153
//   updateMeshColor                       :348-368   nearest voxel of the block itself
//
// The reference walks the 4096 voxels of a block in a fixed order (15^3 interior with x outermost, then the max-X,
// max-Y and max-Z planes) and appends up to five triangles per voxel.  Here, as for the chisel map (tsdf_mesh.hip):
//   vmesh_count   one thread per (block, voxel in that order): number of vertices it emits
//   scan          exclusive scan of the counts = the position of every voxel's first vertex
//   vmesh_emit    same thread layout: vertices, triangle normals and vertex colours at those positions
// so the vertex order inside a block and the order of the blocks are the reference's / the caller's.  The map
// stays in HBM; only the finished mesh crosses PCIe.  Arithmetic follows the reference expression by expression
// (voxel centres through double, as common.h:179-184 computes them), no FMA contraction.
#include <vector>

#include "common.hpp"
#include "device_utils.hpp"
#include "tsdf_voxblox_view.hpp"

namespace {

using plvs::tsdf::dir_find;
using plvs::vbx::VoxbloxMapView;

constexpr int kBlockVox = 4096;
constexpr int kMeshThreads = 256;
constexpr float kMinWeight = 1e-4f;   // MeshIntegratorConfig::min_weight (mesh_integrator.h:49; tsdf_server.cc:323-328 keeps it)

__constant__ int8_t c_vb_triangle_table[256 * 16] = {
#include "mc_table.inc"
};
__constant__ int8_t c_vb_edge_pairs[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6},
                                              {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};

// cube_index_offsets_, mesh_integrator.h:100-102
__device__ __forceinline__ int cube_dx(int i) { return (0x66 >> i) & 1; }   // 0 1 1 0 0 1 1 0
__device__ __forceinline__ int cube_dy(int i) { return (0xCC >> i) & 1; }   // 0 0 1 1 0 0 1 1
__device__ __forceinline__ int cube_dz(int i) { return (0xF0 >> i) & 1; }   // 0 0 0 0 1 1 1 1

// k-th voxel of extractBlockMesh's walk (:172-228).
__device__ __forceinline__ void walk_voxel(int k, int* x, int* y, int* z) {
  if (k < 3375) {                       // interior: x, y, z < 15, z innermost
    *x = k / 225;
    *y = (k / 15) % 15;
    *z = k % 15;
  } else if (k < 3375 + 256) {          // max-X plane: z < 16 outer, y < 16 inner
    k -= 3375;
    *x = 15;
    *y = k % 16;
    *z = k / 16;
  } else if (k < 3375 + 256 + 240) {    // max-Y plane: z < 16 outer, x < 15 inner
    k -= 3375 + 256;
    *x = k % 15;
    *y = 15;
    *z = k / 15;
  } else {                              // max-Z plane: y < 15 outer, x < 15 inner
    k -= 3375 + 256 + 240;
    *x = k % 15;
    *y = k / 15;
    *z = 15;
  }
}

struct Cube {
  float sdf[8];
  int index;   // vertex configuration; 0 = nothing to emit (also: an unobserved corner)
};

// Block id -> pool slot: the rank's own blocks, then the ghosts of a sharded map's halo.
__device__ __forceinline__ int find_block(const VoxbloxMapView& m, int bx, int by, int bz) {
  int s = dir_find(m.dir, bx, by, bz);
  if (s >= m.visible_blocks) s = -1;   // (a block that has not joined the layer yet)
  if (s >= 0 || m.ghost.keys == nullptr) return s;
  return dir_find(m.ghost, bx, by, bz);
}

__device__ __forceinline__ void load_cube(const VoxbloxMapView& m, int slot, int bx, int by, int bz, int x, int y,
                                          int z, Cube* c) {
  c->index = 0;
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
        cached = find_block(m, bx + ox, by + oy, bz + oz);
#pragma unroll
        for (int j = 1; j < 8; ++j)
          if (j == which) nslot[j] = cached;
      }
      s = cached;
      if (s < 0) return;   // the neighbour block does not exist: all_neighbors_observed = false
    }
    const size_t id = (size_t)s * kBlockVox + (size_t)(vx + 16 * (vy + 16 * vz));
    if (m.weight[id] <= kMinWeight) return;   // getSdfIfValid, utils/meshing_utils.h:15-23
    const float d = m.distance[id];
    c->sdf[i] = d;
    index |= (d < 0) ? (1 << i) : 0;
  }
  c->index = index;
}

__device__ __forceinline__ int table_vertices(int index) {
  int n = 0;
  while (n < 15 && c_vb_triangle_table[index * 16 + n] != -1) n += 3;
  return n;
}

// Block ids -> pool slots (-1 = the block does not exist: updateMeshForBlock returns with the mesh cleared, :239-243).
__global__ void vmesh_slots(VoxbloxMapView m, const int32_t* __restrict__ ids, int nblocks, int32_t* __restrict__ slots) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < nblocks) {
    const int s = dir_find(m.dir, ids[3 * c], ids[3 * c + 1], ids[3 * c + 2]);
    slots[c] = s < m.visible_blocks ? s : -1;
  }
}

__global__ __launch_bounds__(kMeshThreads) void vmesh_count(VoxbloxMapView m, const int32_t* __restrict__ ids,
                                                           const int32_t* __restrict__ slots, int nblocks,
                                                           uint32_t* __restrict__ counts) {
  const int g = blockIdx.x * kMeshThreads + threadIdx.x;
  if (g >= nblocks * kBlockVox) return;
  const int c = g / kBlockVox, k = g % kBlockVox;
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

__global__ __launch_bounds__(kMeshThreads) void vmesh_emit(VoxbloxMapView m, const int32_t* __restrict__ ids,
                                                          const int32_t* __restrict__ slots, int nblocks,
                                                          const uint32_t* __restrict__ first, uint32_t cap,
                                                          float* __restrict__ vertices, float* __restrict__ normals,
                                                          uint32_t* __restrict__ colors) {
  const int g = blockIdx.x * kMeshThreads + threadIdx.x;
  if (g >= nblocks * kBlockVox) return;
  const int c = g / kBlockVox, k = g % kBlockVox;
  const int slot = slots[c];
  if (slot < 0) return;
  int x, y, z;
  walk_voxel(k, &x, &y, &z);
  Cube cube;
  const int bx = ids[3 * c], by = ids[3 * c + 1], bz = ids[3 * c + 2];
  load_cube(m, slot, bx, by, bz, x, y, z, &cube);
  if (cube.index == 0) return;
  const float vs = m.voxel_size;
  const float block_size = vs * 16.0f;   // Layer::block_size_ = voxel_size_ * voxels_per_side_ (layer.h:34)
  // Block::origin_ = index * block_size (layer.h:122-126); the voxel centre origin_ + (index + 0.5) * voxel_size with
  // the bracket and the product in double (common.h:179-184)
  const float ox = (float)bx * block_size, oy = (float)by * block_size, oz = (float)bz * block_size;
  const float cx = ox + (float)(((double)(float)x + 0.5) * (double)vs);
  const float cy = oy + (float)(((double)(float)y + 0.5) * (double)vs);
  const float cz = oz + (float)(((double)(float)z + 0.5) * (double)vs);
  // interpolateEdgeVertices (marching_cubes.h:117-134)
  float ex[12], ey[12], ez[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) {
    const int e0 = c_vb_edge_pairs[e][0], e1 = c_vb_edge_pairs[e][1];
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j == e0) s0 = cube.sdf[j];
      if (j == e1) s1 = cube.sdf[j];
    }
    ex[e] = ey[e] = ez[e] = 0.f;
    if ((s0 < 0 && s1 >= 0) || (s0 >= 0 && s1 < 0)) {
      const float x0 = cx + (float)cube_dx(e0) * vs, y0 = cy + (float)cube_dy(e0) * vs, z0 = cz + (float)cube_dz(e0) * vs;
      const float x1 = cx + (float)cube_dx(e1) * vs, y1 = cy + (float)cube_dy(e1) * vs, z1 = cz + (float)cube_dz(e1) * vs;
      const float diff = s0 - s1;
      if (fabsf(diff) >= 1e-6f) {   // interpolateVertex :138-153
        const float t = s0 / diff;
        ex[e] = x0 + t * (x1 - x0);
        ey[e] = y0 + t * (y1 - y0);
        ez[e] = z0 + t * (z1 - z0);
      } else {
        ex[e] = 0.5f * (x0 + x1);
        ey[e] = 0.5f * (y0 + y1);
        ez[e] = 0.5f * (z0 + z1);
      }
    }
  }
  uint32_t pos = first[g];
  for (int col = 0; col < 15 && c_vb_triangle_table[cube.index * 16 + col] != -1; col += 3, pos += 3) {
    if (pos + 3 > cap) return;
    float p[3][3];
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const int e = c_vb_triangle_table[cube.index * 16 + col + 2 - v];   // emitted as col+2, col+1, col
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
      // updateMeshColor (:348-368): computeVoxelIndexFromCoordinates (core/block.h:60-70) clamps into this block,
      // so the look-up never leaves it; an unobserved voxel leaves Color() = 0, 0, 0, 0
      int ix = (int)floorf((p[v][0] - ox) * m.voxel_size_inv + 1e-6f);
      int iy = (int)floorf((p[v][1] - oy) * m.voxel_size_inv + 1e-6f);
      int iz = (int)floorf((p[v][2] - oz) * m.voxel_size_inv + 1e-6f);
      ix = max(min(ix, 15), 0);
      iy = max(min(iy, 15), 0);
      iz = max(min(iz, 15), 0);
      const size_t id = (size_t)slot * kBlockVox + (size_t)(ix + 16 * (iy + 16 * iz));
      colors[pos + v] = (m.weight[id] <= kMinWeight) ? 0u : m.rgba[id];
    }
  }
}

// Scratch of the meshing calls, kept with the map (VoxbloxMapView::ext): the buffers only grow.
struct MeshScratch {
  plvs::DevBuf<int32_t> ids, slots;
  plvs::DevBuf<uint32_t> counts, first, scan, total, colors;
  plvs::DevBuf<float> vertices, normals;
  static void destroy(void* p) {
    MeshScratch* s = static_cast<MeshScratch*>(p);
    s->ids.release(); s->slots.release(); s->counts.release(); s->first.release(); s->scan.release();
    s->total.release(); s->colors.release(); s->vertices.release(); s->normals.release();
    delete s;
  }
};

}  // namespace

extern "C" int plvs_hip_tsdf_voxblox_mesh_blocks(plvs_tsdf_voxblox* h, const int32_t* block_ids_xyz, int nblocks,
                                                 float* vertices, float* normals, uint8_t* colors_rgba, int capacity,
                                                 int32_t* block_first, int* nvertices) {
  PLVS_REQUIRE(h != nullptr && nvertices != nullptr, "null handle / nvertices");
  PLVS_REQUIRE(nblocks >= 0 && capacity >= 0, "negative size");
  *nvertices = 0;
  if (block_first != nullptr)
    for (int c = 0; c <= nblocks; ++c) block_first[c] = 0;
  if (nblocks == 0) return PLVS_OK;
  PLVS_REQUIRE(block_ids_xyz != nullptr && block_first != nullptr, "null block list / block_first");
  PLVS_REQUIRE(nblocks <= (1 << 18), "too many blocks in one call");
  VoxbloxMapView m;
  if (!plvs::vbx::voxblox_map_view(h, &m)) {
    plvs::set_error("mesh_blocks: the map handle is unusable");
    return PLVS_ERR_INVALID_ARG;
  }

  if (*m.ext == nullptr) {
    *m.ext = new MeshScratch();
    *m.ext_free = &MeshScratch::destroy;
  }
  MeshScratch& sc = *static_cast<MeshScratch*>(*m.ext);
  const size_t nvox = (size_t)nblocks * kBlockVox;
  hipStream_t s = nullptr;   // the map's calls are synchronous on return; the default stream orders after them
  PLVS_HIP_TRY(sc.ids.reserve(3 * (size_t)nblocks));
  PLVS_HIP_TRY(sc.slots.reserve((size_t)nblocks));
  PLVS_HIP_TRY(sc.counts.reserve(nvox));
  PLVS_HIP_TRY(sc.first.reserve(nvox));
  PLVS_HIP_TRY(sc.scan.reserve(plvs::scan_scratch_words(nvox)));
  PLVS_HIP_TRY(sc.total.reserve(1));
  PLVS_HIP_TRY(hipMemcpyAsync(sc.ids.p, block_ids_xyz, sizeof(int32_t) * 3 * (size_t)nblocks, hipMemcpyHostToDevice, s));
  vmesh_slots<<<plvs::ceil_div((size_t)nblocks, 256), 256, 0, s>>>(m, sc.ids.p, nblocks, sc.slots.p);
  vmesh_count<<<plvs::ceil_div(nvox, kMeshThreads), kMeshThreads, 0, s>>>(m, sc.ids.p, sc.slots.p, nblocks, sc.counts.p);
  PLVS_KERNEL_CHECK();
  PLVS_HIP_TRY(plvs::exclusive_scan_u32(sc.counts.p, sc.first.p, nvox, sc.total.p, sc.scan.p, s));
  uint32_t total = 0;
  std::vector<uint32_t> firsts((size_t)nblocks);
  PLVS_HIP_TRY(hipMemcpyAsync(&total, sc.total.p, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipMemcpy2DAsync(firsts.data(), sizeof(uint32_t), sc.first.p, sizeof(uint32_t) * kBlockVox, sizeof(uint32_t),
                                (size_t)nblocks, hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  *nvertices = (int)total;
  for (int c = 0; c < nblocks; ++c) block_first[c] = (int32_t)firsts[(size_t)c];
  block_first[nblocks] = (int32_t)total;
  if (total > (uint32_t)capacity) {
    plvs::set_error("mesh_blocks: %u vertices exceed the capacity %d (call again with room for *nvertices)", total, capacity);
    return PLVS_ERR_CAPACITY;
  }
  if (total == 0) return PLVS_OK;
  PLVS_REQUIRE(vertices && normals && colors_rgba, "null output array");
  PLVS_HIP_TRY(sc.vertices.reserve(3 * (size_t)total));
  PLVS_HIP_TRY(sc.normals.reserve(3 * (size_t)total));
  PLVS_HIP_TRY(sc.colors.reserve((size_t)total));
  vmesh_emit<<<plvs::ceil_div(nvox, kMeshThreads), kMeshThreads, 0, s>>>(m, sc.ids.p, sc.slots.p, nblocks, sc.first.p, total,
                                                                        sc.vertices.p, sc.normals.p, sc.colors.p);
  PLVS_KERNEL_CHECK();
  PLVS_HIP_TRY(hipMemcpyAsync(vertices, sc.vertices.p, sizeof(float) * 3 * (size_t)total, hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipMemcpyAsync(normals, sc.normals.p, sizeof(float) * 3 * (size_t)total, hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipMemcpyAsync(colors_rgba, sc.colors.p, sizeof(uint32_t) * (size_t)total, hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  return PLVS_OK;
}
