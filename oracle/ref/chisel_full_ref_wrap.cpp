// C entry points over the WHOLE open_chisel library of the reference — all sixteen sources under
// Thirdparty/open_chisel/src compiled where they lie against the Eigen stand-in of oracle/ref/eigen_full
// (oracle/ref/Makefile -> oracle/_ref/libchisel_full_ref.so).  tests/test_oracle_pinned.py integrates the same clouds
// through chisel::Chisel here and through the restatement in oracle/tsdf_chisel.c and compares every voxel of every
// chunk, and every vertex of every chunk mesh: the map-level control flow (chunk creation / garbage collection, carving
// with the depth image, ChunkManager::RecomputeMesh) then comes from the reference's source, not from a reading of it.
//
// What stands between these calls and PointCloudMapChisel is chisel_server, which needs PCL and is not compiled; the
// few lines of it on this path are repeated here and cited:
//   ChiselServer::SetupProjectionIntegrator   ChiselServer.cpp:623-630
//   ChiselServer::SetDepthCameraInfo          ChiselServer.cpp:444-450 (ToChiselCamera: intrinsics, width, height)
//   PclPointCloudToChisel                     Conversions.h:107-121 (colour bytes * (1.0f / 255.0f), kfid)
//   ChiselServer::IntegrateLastPointCloud     ChiselServer.cpp:664-704
//   ChiselServer::IntegrateWorldPointCloud    ChiselServer.cpp:587-615
// TEST INFRASTRUCTURE ONLY: nothing under plvs_amd/ may call into this file.
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include <open_chisel/Chisel.h>
#include <open_chisel/ProjectionIntegrator.h>
#include <open_chisel/camera/DepthImage.h>
#include <open_chisel/camera/PinholeCamera.h>
#include <open_chisel/pointcloud/PointCloud.h>
#include <open_chisel/truncation/QuadraticTruncator.h>
#include <open_chisel/weighting/ConstantWeighter.h>

namespace {

struct FullRef {
  std::unique_ptr<chisel::Chisel> map;
  chisel::ProjectionIntegrator integrator;
  chisel::PinholeCamera camera;
  float far_plane;
};

chisel::Transform pose_of(const float* Twc) {   // 3x4 row-major [R|t]
  chisel::Transform T;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) T.linear()(r, c) = Twc[4 * r + c];
    T.translation()(r) = Twc[4 * r + 3];
  }
  return T;
}

void fill_cloud(chisel::PointCloud* cloud, const float* xyz, const uint8_t* rgb, const uint32_t* kfid,
                const float* normals, int n) {
  const float byteToFloat = 1.0f / 255.0f;
  for (int i = 0; i < n; ++i) {
    cloud->AddPoint(chisel::Vec3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
    cloud->AddColor(chisel::Vec3(rgb[3 * i] * byteToFloat, rgb[3 * i + 1] * byteToFloat, rgb[3 * i + 2] * byteToFloat));
    if (normals) cloud->AddNormal(chisel::Vec3(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]));
    cloud->GetMutableKfids().push_back(kfid ? kfid[i] : 0u);
  }
}

}  // namespace

extern "C" {

void* ref_chisel_full_create(float resolution, float tq, float tl, float tc, float ts, float weight, int carving,
                             float carving_dist, float fx, float fy, float cx, float cy, int width, int height,
                             float near_plane, float far_plane) {
  FullRef* h = new FullRef;
  h->map.reset(new chisel::Chisel(Eigen::Vector3i(16, 16, 16), resolution, true));
  h->integrator.SetCentroids(h->map->GetChunkManager().GetCentroids());
  h->integrator.SetTruncator(chisel::TruncatorPtr(new chisel::QuadraticTruncator(tq, tl, tc, ts)));
  h->integrator.SetWeighter(chisel::WeighterPtr(new chisel::ConstantWeighter(weight)));
  h->integrator.SetCarvingDist(carving_dist);
  h->integrator.SetCarvingEnabled(carving != 0);
  chisel::Intrinsics in;
  in.SetFx(fx); in.SetFy(fy); in.SetCx(cx); in.SetCy(cy);
  h->camera.SetIntrinsics(in);
  h->camera.SetWidth(width);
  h->camera.SetHeight(height);
  h->camera.SetNearPlane(near_plane);
  h->camera.SetFarPlane(far_plane);
  h->far_plane = far_plane;
  return h;
}

void ref_chisel_full_destroy(void* p) { delete static_cast<FullRef*>(p); }

// IntegratePointCloudWidthDepth<float>.  depth: height x width floats or null (null only with carving off).
void ref_chisel_full_integrate(void* p, const float* xyz, const uint8_t* rgb, const uint32_t* kfid, int n,
                               const float* Twc, const float* depth) {
  FullRef* h = static_cast<FullRef*>(p);
  chisel::PointCloud cloud;
  fill_cloud(&cloud, xyz, rgb, kfid, nullptr, n);
  std::shared_ptr<chisel::DepthImage<float> > img;
  if (depth) {
    img.reset(new chisel::DepthImage<float>(h->camera.GetWidth(), h->camera.GetHeight()));
    std::memcpy(img->GetMutableData(), depth, sizeof(float) * (size_t)h->camera.GetWidth() * h->camera.GetHeight());
  }
  h->map->IntegratePointCloudWidthDepth<float>(h->integrator, cloud, pose_of(Twc), img, h->camera, h->far_plane);
}

void ref_chisel_full_integrate_world_normals(void* p, const float* xyz, const uint8_t* rgb, const uint32_t* kfid,
                                             const float* normals, int n, const float* Twc) {
  FullRef* h = static_cast<FullRef*>(p);
  chisel::PointCloud cloud;
  fill_cloud(&cloud, xyz, rgb, kfid, normals, n);
  h->map->IntegrateWorldPointCloudWithNormals(h->integrator, cloud, pose_of(Twc), h->far_plane);
}

int ref_chisel_full_num_chunks(void* p) {
  return (int)static_cast<FullRef*>(p)->map->GetChunkManager().GetChunks().size();
}

void ref_chisel_full_chunk_ids(void* p, int32_t* ids) {
  size_t k = 0;
  for (const auto& kv : static_cast<FullRef*>(p)->map->GetChunkManager().GetChunks()) {
    ids[3 * k] = kv.first(0); ids[3 * k + 1] = kv.first(1); ids[3 * k + 2] = kv.first(2);
    ++k;
  }
}

int ref_chisel_full_get_chunk(void* p, int cx, int cy, int cz, float* sdf, float* weight, uint32_t* kfid,
                              uint32_t* rgbw) {
  const chisel::ChunkManager& cm = static_cast<FullRef*>(p)->map->GetChunkManager();
  const chisel::ChunkID id(cx, cy, cz);
  if (!cm.HasChunk(id)) return 0;
  chisel::ChunkPtr chunk = cm.GetChunk(id);
  for (int i = 0; i < 4096; ++i) {
    const chisel::DistVoxel& d = chunk->GetDistVoxel(i);
    const chisel::ColorVoxel& c = chunk->GetColorVoxel(i);
    sdf[i] = d.GetSDF();
    weight[i] = d.GetWeight();
    kfid[i] = d.GetKfid();
    rgbw[i] = (uint32_t)c.GetRed() | (uint32_t)c.GetGreen() << 8 | (uint32_t)c.GetBlue() << 16 |
              (uint32_t)c.GetWeight() << 24;
  }
  return 1;
}

// Chisel::UpdateMeshes (Chisel.cpp:57-65): RecomputeMeshes over the chunks the integrates marked.
void ref_chisel_full_update_meshes(void* p) { static_cast<FullRef*>(p)->map->UpdateMeshes(); }

// The mesh ChunkManager holds for one chunk after UpdateMeshes: returns its vertex count (0 if none), writing up to
// cap vertices / normals / colours (n x 3 floats) and kfids.
int ref_chisel_full_mesh_chunk(void* p, int cx, int cy, int cz, float* vertices, float* normals, float* colors,
                               uint32_t* kfids, int cap) {
  const chisel::ChunkManager& cm = static_cast<FullRef*>(p)->map->GetChunkManager();
  const chisel::ChunkID id(cx, cy, cz);
  if (!cm.HasMesh(id)) return 0;
  const chisel::MeshPtr& m = cm.GetMesh(id);
  const int n = (int)m->vertices.size();
  for (int i = 0; i < n && i < cap; ++i)
    for (int k = 0; k < 3; ++k) {
      vertices[3 * i + k] = m->vertices[i](k);
      normals[3 * i + k] = m->normals[i](k);
      colors[3 * i + k] = m->colors[i](k);
    }
  for (int i = 0; i < n && i < cap && i < (int)m->kfids.size(); ++i) kfids[i] = m->kfids[i];
  return n;
}

// Chisel::Deform (Chisel.cpp:588-591 -> ChunkManager::Deform): kfids n, Rt n x 12 floats (R row-major, then t), as
// PointCloudMapChisel::OnMapChange fills the MapKfidRt.
void ref_chisel_full_deform(void* p, const uint32_t* kfids, const float* Rt, int n) {
  chisel::MapKfidRt map;
  for (int i = 0; i < n; ++i) {
    chisel::TransformRt T;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) T.R(r, c) = Rt[12 * i + 3 * r + c];
      T.t(r) = Rt[12 * i + 9 + r];
    }
    map[kfids[i]] = T;
  }
  static_cast<FullRef*>(p)->map->Deform(map);
}

}  // extern "C"
