// Per-ray and per-voxel arithmetic of voxblox's "simple" TSDF integrator,
// shared by the kernels (and compilable on the host for unit checks).
//
// Reference: Thirdparty/voxblox/src/integrator/tsdf_integrator.cc:85-103
// (isPointValid), :173-264 (updateTsdfVoxel, computeDistance, getVoxelWeight),
// :291-327 (SimpleTsdfIntegrator::integrateFunction);
// src/integrator/integrator_utils.cc:33-44 (mixed visiting order), :137-235
// (RayCaster); include/voxblox/core/common.h:95-125, 140-222.
// Built with -ffp-contract=off; 3-term reductions are a0 + (a1 + a2) as Eigen
// evaluates them.
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define PLVS_HD __host__ __device__ __forceinline__
#else
#define PLVS_HD inline
#endif

namespace plvs {
namespace vbx {

constexpr int kBlockVox = 4096;  // 16^3, linear index x + 16*(y + 16*z) (block_inl.h)

struct Params {
  float voxel_size, voxel_size_inv, vps_inv;
  float truncation, max_weight, min_ray, max_ray;
  int carving, allow_clear;
  int shard_rank, shard_count;
};

struct PoseRt {
  float R[9], t[3];
  float q[4];   // w, x, y, z: the rotation as the reference's kindr transformation holds it
};

// Eigen 3.3's rotation matrix -> quaternion assignment (Quaternion.h, quaternionbase_assign_impl<Other,3,3>), which
// kindr::minimal::QuatTransformationTemplate(TransformationMatrix) runs on the pose (tsdf_server.cc:484-486).
PLVS_HD void quat_from_matrix(const float m[9], float q[4]) {
  const float tr = m[0] + (m[4] + m[8]);
  if (tr > 0.0f) {
    float s = sqrtf(tr + 1.0f);
    q[0] = 0.5f * s;
    s = 0.5f / s;
    q[1] = (m[7] - m[5]) * s;
    q[2] = (m[2] - m[6]) * s;
    q[3] = (m[3] - m[1]) * s;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    float s = sqrtf(m[4 * i] - m[4 * j] - m[4 * k] + 1.0f);
    float v[3];
    v[i] = 0.5f * s;
    s = 0.5f / s;
    q[0] = (m[3 * k + j] - m[3 * j + k]) * s;
    v[j] = (m[3 * j + i] + m[3 * i + j]) * s;
    v[k] = (m[3 * k + i] + m[3 * i + k]) * s;
    q[1] = v[0]; q[2] = v[1]; q[3] = v[2];
  }
}

// T_G_C * point_C = q.rotate(p) + t (minkindr quat-transformation-inl.h:159-162; Eigen's
// QuaternionBase::_transformVector: uv = 2 (q.vec x v); (v + w uv) + q.vec x uv).
PLVS_HD void quat_transform(const PoseRt& pose, float vx, float vy, float vz, float out[3]) {
  const float w = pose.q[0], x = pose.q[1], y = pose.q[2], z = pose.q[3];
  float u0 = y * vz - z * vy, u1 = z * vx - x * vz, u2 = x * vy - y * vx;
  u0 += u0; u1 += u1; u2 += u2;
  const float c0 = y * u2 - z * u1, c1 = z * u0 - x * u2, c2 = x * u1 - y * u0;
  out[0] = ((vx + w * u0) + c0) + pose.t[0];
  out[1] = ((vy + w * u1) + c1) + pose.t[1];
  out[2] = ((vz + w * u2) + c2) + pose.t[2];
}

PLVS_HD float vsum3(float a, float b, float c) { return a + (b + c); }

// ThreadSafeIndex::getMixedIndex, step_size_ = 1024.
PLVS_HD uint32_t mixed_index(uint32_t base_idx, uint32_t number_of_points) {
  const uint32_t step = 1024u, groups = number_of_points / step;
  if (groups * step <= base_idx) return base_idx;
  return (base_idx % groups) * step + base_idx / groups;
}

struct Ray {
  int cur[3], sgn[3], steps;  // the ray emits steps + 1 voxels
  float t_next[3], t_step[3];
  float pG[3];
  float weight;               // getVoxelWeight(point_C)
};

PLVS_HD void ray_setup(const Params& P, const PoseRt& pose, bool clearing, Ray* r, bool from_end = false);

// isPointValid + RayCaster set-up.  Returns false when the point is skipped.
// from_end: the same two end points cast the other way round (RayCaster(..., cast_from_origin = false),
// integrator_utils.cc:164-168: what FastTsdfIntegrator asks for).
PLVS_HD bool make_ray(const Params& P, const PoseRt& pose, float px, float py, float pz, Ray* r, bool from_end = false) {
  const float ray_distance = sqrtf(vsum3(px * px, py * py, pz * pz));
  bool clearing;
  if (ray_distance < P.min_ray) return false;
  else if (ray_distance > P.max_ray) {
    if (!P.allow_clear) return false;
    clearing = true;
  } else
    clearing = false;
  quat_transform(pose, px, py, pz, r->pG);
  ray_setup(P, pose, clearing, r, from_end);
  r->weight = fabsf(pz) > 1e-6f ? 1.0f / (pz * pz) : 0.0f;  // use_const_weight = false
  return true;
}

// MergedTsdfIntegrator::integrateVoxel's ray (tsdf_integrator.cc:418-424): RayCaster(origin, T_G_C * merged_point_C,
// clearing_ray, ...) with the clearing flag of the bundle — no validity test of its own (bundleRays made it per point).
PLVS_HD void make_ray_merged(const Params& P, const PoseRt& pose, float px, float py, float pz, bool clearing, Ray* r) {
  quat_transform(pose, px, py, pz, r->pG);
  ray_setup(P, pose, clearing, r);
  r->weight = 0.0f;   // (the bundle's merged weight is an input of its own)
}

// RayCaster's constructor from origin and r->pG (integrator_utils.cc:137-167) + setupRayCaster (:196-235).
PLVS_HD void ray_setup(const Params& P, const PoseRt& pose, bool clearing, Ray* r, bool from_end) {
  const float* o = pose.t;
  const float d0 = r->pG[0] - o[0], d1 = r->pG[1] - o[1], d2 = r->pG[2] - o[2];
  const float z2 = vsum3(d0 * d0, d1 * d1, d2 * d2);
  const float dn = sqrtf(z2);
  float u[3] = {d0, d1, d2};
  if (z2 > 0.0f) { u[0] = d0 / dn; u[1] = d1 / dn; u[2] = d2 / dn; }
  float rs[3], re[3];
  if (clearing) {
    float tmp = dn - P.truncation;
    tmp = (tmp < 0.0f) ? 0.0f : tmp;                     // std::max(x, 0)
    const float len = (P.max_ray < tmp) ? P.max_ray : tmp;  // std::min(x, max_ray)
    for (int k = 0; k < 3; ++k) {
      re[k] = o[k] + u[k] * len;
      rs[k] = P.carving ? o[k] : re[k];
    }
  } else {
    for (int k = 0; k < 3; ++k) {
      re[k] = r->pG[k] + u[k] * P.truncation;
      rs[k] = P.carving ? o[k] : (r->pG[k] - u[k] * P.truncation);
    }
  }
  r->steps = 0;
  for (int k = 0; k < 3; ++k) {
    // (cast_from_origin = false: setupRayCaster(end_scaled, start_scaled))
    const float ss = (from_end ? re[k] : rs[k]) * P.voxel_size_inv, es = (from_end ? rs[k] : re[k]) * P.voxel_size_inv;
    r->cur[k] = (int)floorf(ss + 1e-6f);
    const int endk = (int)floorf(es + 1e-6f);
    const int dk = endk - r->cur[k];
    r->steps += dk < 0 ? -dk : dk;
    const float ray = es - ss;
    r->sgn[k] = (ray == 0) ? 0 : (ray < 0 ? -1 : 1);
    const float corrected = (float)(r->sgn[k] > 0 ? r->sgn[k] : 0);
    const float shifted = ss - (float)r->cur[k];
    r->t_next[k] = (corrected - shifted) / ray;
    r->t_step[k] = (float)r->sgn[k] / ray;
  }
}

// The world-cloud-with-normals flavour (TsdfIntegratorBase::integrateWorlPointCloud, tsdf_integrator.cc:35-82: what
// PointCloudMapVoxblox::LoadMap integrates the saved cloud through): the ray point + n * truncation ->
// point - n * truncation with n = T_G_C * normalized(normal) (the full transformation, as written there), through
// RayCaster(start_scaled, end_scaled); ray_start takes the sensor origin's place in updateTsdfVoxel, weight 1.
PLVS_HD void make_ray_world(const Params& P, const PoseRt& pose, float px, float py, float pz, float nx, float ny,
                            float nz, Ray* r, float ray_start[3]) {
  const float z2 = vsum3(nx * nx, ny * ny, nz * nz);
  if (z2 > 0.0f) {
    const float l = sqrtf(z2);
    nx /= l; ny /= l; nz /= l;
  }
  float nG[3];
  quat_transform(pose, px, py, pz, r->pG);
  quat_transform(pose, nx, ny, nz, nG);
  r->steps = 0;
  for (int k = 0; k < 3; ++k) {
    ray_start[k] = r->pG[k] + nG[k] * P.truncation;
    const float re = r->pG[k] - nG[k] * P.truncation;
    const float ss = ray_start[k] * P.voxel_size_inv, es = re * P.voxel_size_inv;
    r->cur[k] = (int)floorf(ss + 1e-6f);
    const int endk = (int)floorf(es + 1e-6f);
    const int dk = endk - r->cur[k];
    r->steps += dk < 0 ? -dk : dk;
    const float ray = es - ss;
    r->sgn[k] = (ray == 0) ? 0 : (ray < 0 ? -1 : 1);
    const float corrected = (float)(r->sgn[k] > 0 ? r->sgn[k] : 0);
    const float shifted = ss - (float)r->cur[k];
    r->t_next[k] = (corrected - shifted) / ray;
    r->t_step[k] = (float)r->sgn[k] / ray;
  }
  r->weight = 1.0f;
}

// RayCaster::nextRayIndex: returns the current voxel and advances.
PLVS_HD void ray_step(Ray* r, int g[3]) {
  g[0] = r->cur[0]; g[1] = r->cur[1]; g[2] = r->cur[2];
  // Eigen minCoeff(&idx): the first coefficient unless a later one is strictly smaller.  Written with selects on
  // constant indices: an index computed at run time would send the whole Ray to scratch / LDS.
  const bool y_lt = r->t_next[1] < r->t_next[0];
  const float m01 = y_lt ? r->t_next[1] : r->t_next[0];
  const bool z_lt = r->t_next[2] < m01;
  const bool s0 = !y_lt && !z_lt, s1 = y_lt && !z_lt;
  r->cur[0] += s0 ? r->sgn[0] : 0;
  r->cur[1] += s1 ? r->sgn[1] : 0;
  r->cur[2] += z_lt ? r->sgn[2] : 0;
  r->t_next[0] = s0 ? r->t_next[0] + r->t_step[0] : r->t_next[0];
  r->t_next[1] = s1 ? r->t_next[1] + r->t_step[1] : r->t_next[1];
  r->t_next[2] = z_lt ? r->t_next[2] + r->t_step[2] : r->t_next[2];
}

PLVS_HD uint64_t owner_hash(int x, int y, int z) {
  return ((uint64_t)(int64_t)x * 73856093ull) ^ ((uint64_t)(int64_t)y * 19349663ull) ^
         ((uint64_t)(int64_t)z * 83492791ull);
}

// hash % count without a 64-bit division (count = number of GPUs: small, usually a power of two).
PLVS_HD int shard_of(uint64_t h, int count) {
  const uint32_t c = (uint32_t)count;
  if ((c & (c - 1u)) == 0u) return (int)((uint32_t)h & (c - 1u));
  const uint32_t hi = (uint32_t)(h >> 32) % c, lo = (uint32_t)h % c;
  const uint32_t two32 = ((0xFFFFFFFFu % c) + 1u) % c;   // 2^32 mod c
  return (int)(((unsigned long long)hi * two32 + lo) % c);
}

// getBlockIndexFromGlobalVoxelIndex + getLocalFromGlobalVoxelIndex.  Returns false
// when the block belongs to another shard.
PLVS_HD bool block_of(const Params& P, const int g[3], int b[3], int* vid) {
  b[0] = (int)floorf((float)g[0] * P.vps_inv);
  b[1] = (int)floorf((float)g[1] * P.vps_inv);
  b[2] = (int)floorf((float)g[2] * P.vps_inv);
  if (P.shard_count > 1 && shard_of(owner_hash(b[0], b[1], b[2]), P.shard_count) != P.shard_rank)
    return false;
  const uint32_t off = 1u << 31;
  const int lx = (int)(((uint32_t)g[0] + off) & 15u), ly = (int)(((uint32_t)g[1] + off) & 15u),
            lz = (int)(((uint32_t)g[2] + off) & 15u);
  *vid = lx + 16 * (ly + lz * 16);
  return true;
}

// The order-independent part of updateTsdfVoxel: sdf and the (drop-off) weight.
PLVS_HD void visit_operands(const Params& P, const float* origin, const float* pG, const int g[3],
                            float weight, float* sdf_out, float* uw_out) {
  const float c0 = ((float)g[0] + 0.5f) * P.voxel_size, c1 = ((float)g[1] + 0.5f) * P.voxel_size,
              c2 = ((float)g[2] + 0.5f) * P.voxel_size;
  const float a0 = c0 - origin[0], a1 = c1 - origin[1], a2 = c2 - origin[2];
  const float b0 = pG[0] - origin[0], b1 = pG[1] - origin[1], b2 = pG[2] - origin[2];
  const float dist_G = sqrtf(vsum3(b0 * b0, b1 * b1, b2 * b2));
  const float dist_G_V = vsum3(a0 * b0, a1 * b1, a2 * b2) / dist_G;
  const float sdf = dist_G - dist_G_V;
  float uw = weight;
  if (sdf < -P.voxel_size) {  // use_weight_dropoff
    uw = weight * (P.truncation + sdf) / (P.truncation - P.voxel_size);
    uw = (uw < 0.0f) ? 0.0f : uw;  // std::max(uw, 0)
  }
  *sdf_out = sdf;
  *uw_out = uw;
}

// Color::blendTwoColors on r | g<<8 | b<<16 | a<<24.
PLVS_HD uint32_t blend_colours(uint32_t c1, float w1, uint32_t c2, float w2) {
  const float total = w1 + w2;
  w1 /= total;
  w2 /= total;
  uint32_t out = 0;
  for (int k = 0; k < 4; ++k) {
    const float a = (float)((c1 >> (8 * k)) & 255u), b = (float)((c2 >> (8 * k)) & 255u);
    out |= ((uint32_t)(uint8_t)roundf(a * w1 + b * w2)) << (8 * k);
  }
  return out;
}

// The order-dependent fold of updateTsdfVoxel.
PLVS_HD void voxel_fold(const Params& P, float& D, float& W, uint32_t& C, float sdf, float uw,
                        uint32_t colour) {
  const float new_weight = W + uw;
  if (new_weight < 1e-6f) return;
  const float new_sdf = (sdf * uw + D * W) / new_weight;
  if (fabsf(sdf) < P.truncation) C = blend_colours(C, W, colour, uw);
  // std::min(trunc, x) / std::max(-trunc, x) / std::min(max_weight, x), argument order kept
  D = (new_sdf > 0.0f) ? ((new_sdf < P.truncation) ? new_sdf : P.truncation)
                       : ((-P.truncation < new_sdf) ? new_sdf : -P.truncation);
  W = (new_weight < P.max_weight) ? new_weight : P.max_weight;
}

}  // namespace vbx
}  // namespace plvs
