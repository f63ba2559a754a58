// Host compile of the DEVICE arithmetic header (plvs_amd/csrc/tsdf_chisel_core.hpp)
// driven sequentially — a CPU unit check that the code the kernels execute
// (ray set-up, Amanatides-Woo cursor, visit resolution, voxel update) agrees
// bit-for-bit with the oracle.  Test infrastructure only: never loaded by the
// product path (which has no CPU fallback).
#include <cstdint>
#include <cstring>
#include <map>
#include <tuple>
#include <vector>

#include "../../plvs_amd/csrc/tsdf_chisel_core.hpp"
#include "../../plvs_amd/csrc/tsdf_voxblox_core.hpp"

using namespace plvs::chisel;

struct HostChunk {
  std::vector<float> sdf, w;
  std::vector<uint32_t> kfid, rgbw;
  HostChunk() : sdf(kChunkVox, 99999.0f), w(kChunkVox, 0.0f), kfid(kChunkVox, 0), rgbw(kChunkVox, 0) {}
};

struct HostMap {
  Params P;
  std::map<std::tuple<int, int, int>, HostChunk> chunks;
  long long visits = 0, fallbacks = 0;
};

extern "C" {

HostMap* hostcore_create(float resolution, float tq, float tl, float tc, float ts, float weight,
                         int shard_rank, int shard_count) {
  HostMap* m = new HostMap();
  Params& P = m->P;
  P.resolution = resolution;
  P.round_to_voxel = 1.0f / resolution;
  P.half_voxel = resolution * 0.5f;
  P.rounding = 1.0f / ((float)16 * resolution);
  P.diag = (float)(2.0 * std::sqrt((double)3.0f) * (double)resolution);
  P.tq = tq; P.tl = tl; P.tc = tc; P.ts = ts; P.weight = weight;
  P.shard_rank = shard_rank;
  P.shard_count = shard_count < 1 ? 1 : shard_count;
  return m;
}

void hostcore_destroy(HostMap* m) { delete m; }

void hostcore_integrate(HostMap* m, const float* xyz, const uint8_t* rgb, const uint32_t* kfid,
                        int n, const float* Twc) {
  Pose pose;
  make_pose(Twc, &pose);
  m->visits = 0;
  for (int i = 0; i < n; ++i) {
    Ray ray;
    if (!make_ray(m->P, pose, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], &ray)) continue;
    RayCursor cur;
    OwnerCache owner;
    ray_begin(ray, &cur);
    int vx, vy, vz;
    while (ray_next(&cur, &vx, &vy, &vz)) {
      Visit v;
      if (!resolve_visit(m->P, pose, ray, vx, vy, vz, &v, &owner)) continue;
      HostChunk& c = m->chunks[std::make_tuple(v.cx, v.cy, v.cz)];
      // what apply_runs recomputes from (voxel, point)
      const int lx = v.vid & 15, ly = (v.vid >> 4) & 15, lz = v.vid >> 8;
      const float c0 = (float)(v.cx * 16 + lx) * m->P.resolution + m->P.half_voxel;
      const float c1 = (float)(v.cy * 16 + ly) * m->P.resolution + m->P.half_voxel;
      const float c2 = (float)(v.cz * 16 + lz) * m->P.resolution + m->P.half_voxel;
      const float depth = xyz[3 * i + 2];
      const float tr = truncation_of(m->P, depth);
      const float u = signed_dist(pose, depth, c0, c1, c2);
      const float wu = m->P.weight / (2.0f * tr);
      // the chain kernel's form of DistVoxel::Integrate: reciprocal form, plain division
      // when an operand is outside its exact range
      float s_new = c.sdf[v.vid], w_new = c.w[v.vid];
      float amin = 1.0f, amax = 1.0f;
      const float wn = wu + w_new;
      dist_update_rcp(s_new, w_new, wu * u, wn, 1.0f / wn, amin, amax);
      if (dist_update_rcp_exact(amin, amax, wn, wn)) {
        c.sdf[v.vid] = s_new;
        c.w[v.vid] = w_new;
      } else {
        dist_update(c.sdf[v.vid], c.w[v.vid], wu * u, wu);
        m->fallbacks++;
      }
      c.kfid[v.vid] = kfid ? kfid[i] : 0u;
      colour_update(c.rgbw[v.vid], colour_roundtrip(rgb[3 * i]), colour_roundtrip(rgb[3 * i + 1]),
                    colour_roundtrip(rgb[3 * i + 2]));
      m->visits++;
    }
  }
}

long long hostcore_last_visits(HostMap* m) { return m->visits; }
long long hostcore_fallbacks(HostMap* m) { return m->fallbacks; }
int hostcore_num_chunks(HostMap* m) { return (int)m->chunks.size(); }
void hostcore_chunk_ids(HostMap* m, int32_t* ids) {
  int k = 0;
  for (auto& kv : m->chunks) {
    ids[3 * k] = std::get<0>(kv.first);
    ids[3 * k + 1] = std::get<1>(kv.first);
    ids[3 * k + 2] = std::get<2>(kv.first);
    ++k;
  }
}
int hostcore_get_chunk(HostMap* m, int cx, int cy, int cz, float* sdf, float* w, uint32_t* kfid,
                       uint32_t* rgbw) {
  auto it = m->chunks.find(std::make_tuple(cx, cy, cz));
  if (it == m->chunks.end()) return 0;
  memcpy(sdf, it->second.sdf.data(), kChunkVox * 4);
  memcpy(w, it->second.w.data(), kChunkVox * 4);
  memcpy(kfid, it->second.kfid.data(), kChunkVox * 4);
  memcpy(rgbw, it->second.rgbw.data(), kChunkVox * 4);
  return 1;
}
}

// ---------------------------------------------------------------- sharding checks
extern "C" {

int hostcore_shard_of(unsigned long long h, int count) { return shard_of((uint64_t)h, count); }

// For every point: does the cheap cull agree with the walk?  Returns the number of rays whose walk
// emits a visit this rank owns although walk_may_touch_owned said it could not (must be 0);
// *culled = rays the cull rejects, *owned_rays = rays with at least one owned visit.
long long hostcore_cull_violations(HostMap* m, const float* xyz, int n, const float* Twc, long long* culled,
                                   long long* owned_rays) {
  Pose pose;
  make_pose(Twc, &pose);
  long long bad = 0;
  *culled = 0;
  *owned_rays = 0;
  for (int i = 0; i < n; ++i) {
    Ray ray;
    if (!make_ray(m->P, pose, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], &ray)) continue;
    if (!ray_in_coord_range(ray)) continue;
    const bool may = walk_may_touch_owned(m->P, ray);
    RayCursor cur;
    OwnerCache owner;
    ray_begin(ray, &cur);
    int vx, vy, vz;
    bool any = false;
    while (ray_next(&cur, &vx, &vy, &vz)) {
      Visit v;
      any = resolve_visit(m->P, pose, ray, vx, vy, vz, &v, &owner) || any;
    }
    *culled += may ? 0 : 1;
    *owned_rays += any ? 1 : 0;
    bad += (any && !may) ? 1 : 0;
  }
  return bad;
}

// The reference's float chunk lookup (ChunkManager.h:192-198 on the voxel centre, Chisel.cpp:505-511)
// against the integer form the kernels use; returns the number of coordinates where they differ.
long long hostcore_chunk_id_mismatches(float resolution, const int32_t* v, int n) {
  const float half = resolution * 0.5f, rounding = 1.0f / ((float)16 * resolution);
  long long bad = 0;
  for (int i = 0; i < n; ++i) {
    const float centre = (float)v[i] * resolution + half;
    const int ref = (int)std::floor(centre * rounding);
    bad += (ref != (v[i] >> 4)) ? 1 : 0;
  }
  return bad;
}
}

// ---------------------------------------------------------------- voxblox
struct HostVBlock {
  std::vector<float> d, w;
  std::vector<uint32_t> c;
  HostVBlock() : d(plvs::vbx::kBlockVox, 0.0f), w(plvs::vbx::kBlockVox, 0.0f), c(plvs::vbx::kBlockVox, 0) {}
};
struct HostVMap {
  plvs::vbx::Params P;
  std::map<std::tuple<int, int, int>, HostVBlock> blocks;
  long long visits = 0, fallbacks = 0;
};

extern "C" {
HostVMap* hostvbx_create(float voxel_size, float truncation, float max_weight, float min_ray,
                         float max_ray, int carving, int shard_rank, int shard_count) {
  HostVMap* m = new HostVMap();
  plvs::vbx::Params& P = m->P;
  P.voxel_size = voxel_size;
  P.voxel_size_inv = (float)(1.0 / voxel_size);
  P.vps_inv = (float)(1.0 / 16);
  P.truncation = truncation; P.max_weight = max_weight; P.min_ray = min_ray; P.max_ray = max_ray;
  P.carving = carving ? 1 : 0; P.allow_clear = P.carving;
  P.shard_rank = shard_rank; P.shard_count = shard_count < 1 ? 1 : shard_count;
  return m;
}
void hostvbx_destroy(HostVMap* m) { delete m; }
void hostvbx_integrate(HostVMap* m, const float* xyz, const uint8_t* rgba, int n, const float* Twc) {
  namespace vbx = plvs::vbx;
  using vbx::mixed_index; using vbx::make_ray; using vbx::ray_step; using vbx::block_of;
  using vbx::visit_operands; using vbx::voxel_fold;
  vbx::PoseRt pose;
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) pose.R[3 * i + j] = Twc[4 * i + j]; pose.t[i] = Twc[4 * i + 3]; }
  vbx::quat_from_matrix(pose.R, pose.q);   // as load_pose (tsdf_voxblox.hip) does
  m->visits = 0;
  for (int seq = 0; seq < n; ++seq) {
    const int p = (int)mixed_index((uint32_t)seq, (uint32_t)n);
    vbx::Ray ray;
    if (!make_ray(m->P, pose, xyz[3 * p], xyz[3 * p + 1], xyz[3 * p + 2], &ray)) continue;
    uint32_t colour;
    memcpy(&colour, rgba + 4 * (size_t)p, 4);
    for (int s = 0; s <= ray.steps; ++s) {
      int g[3], b[3], vid;
      ray_step(&ray, g);
      if (!block_of(m->P, g, b, &vid)) continue;
      HostVBlock& blk = m->blocks[std::make_tuple(b[0], b[1], b[2])];
      // what vb_expand recomputes from (voxel, point)
      const int gg[3] = {b[0] * 16 + (vid & 15), b[1] * 16 + ((vid >> 4) & 15), b[2] * 16 + (vid >> 8)};
      float sdf, uw;
      visit_operands(m->P, pose.t, ray.pG, gg, ray.weight, &sdf, &uw);
      voxel_fold(m->P, blk.d[vid], blk.w[vid], blk.c[vid], sdf, uw, colour);
      m->visits++;
    }
  }
}
long long hostvbx_last_visits(HostVMap* m) { return m->visits; }
int hostvbx_num_chunks(HostVMap* m) { return (int)m->blocks.size(); }
void hostvbx_chunk_ids(HostVMap* m, int32_t* ids) {
  int k = 0;
  for (auto& kv : m->blocks) {
    ids[3 * k] = std::get<0>(kv.first); ids[3 * k + 1] = std::get<1>(kv.first); ids[3 * k + 2] = std::get<2>(kv.first);
    ++k;
  }
}
int hostvbx_get_chunk(HostVMap* m, int bx, int by, int bz, float* d, float* w, uint32_t* c) {
  auto it = m->blocks.find(std::make_tuple(bx, by, bz));
  if (it == m->blocks.end()) return 0;
  memcpy(d, it->second.d.data(), plvs::vbx::kBlockVox * 4);
  memcpy(w, it->second.w.data(), plvs::vbx::kBlockVox * 4);
  memcpy(c, it->second.c.data(), plvs::vbx::kBlockVox * 4);
  return 1;
}
}
