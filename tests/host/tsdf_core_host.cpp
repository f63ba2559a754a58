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

using namespace plvs::chisel;

struct HostChunk {
  std::vector<float> sdf, w;
  std::vector<uint32_t> kfid, rgbw;
  HostChunk() : sdf(kChunkVox, 99999.0f), w(kChunkVox, 0.0f), kfid(kChunkVox, 0), rgbw(kChunkVox, 0) {}
};

struct HostMap {
  Params P;
  std::map<std::tuple<int, int, int>, HostChunk> chunks;
  long long visits = 0;
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
    ray_begin(ray, &cur);
    int vx, vy, vz;
    while (ray_next(&cur, &vx, &vy, &vz)) {
      Visit v;
      if (!resolve_visit(m->P, pose, ray, vx, vy, vz, &v)) continue;
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
      apply_update(c.sdf[v.vid], c.w[v.vid], c.kfid[v.vid], c.rgbw[v.vid], u, wu,
                   kfid ? kfid[i] : 0u, colour_roundtrip(rgb[3 * i]), colour_roundtrip(rgb[3 * i + 1]),
                   colour_roundtrip(rgb[3 * i + 2]));
      m->visits++;
    }
  }
}

long long hostcore_last_visits(HostMap* m) { return m->visits; }
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
