// Per-ray and per-voxel arithmetic of the open_chisel point-cloud integrate,
// shared by every TSDF kernel (and compilable on the host for unit checks of
// the device logic).  Evaluation order is part of the contract: results must
// round exactly like Chisel.cpp / Raycast.cpp built against Eigen 3.3 (3-term
// reductions are a0 + (a1 + a2); an affine transform of a point is t + R p),
// and the translation unit is built with -ffp-contract=off.
//
// Reference: Thirdparty/open_chisel/src/Chisel.cpp:442-550,
//            Thirdparty/open_chisel/src/geometry/Raycast.cpp:6-182,
//            include/open_chisel/ChunkManager.h:42-54, 192-206.
#pragma once
#include <climits>
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define PLVS_HD __host__ __device__ __forceinline__
#else
#define PLVS_HD inline
#endif

namespace plvs {
namespace chisel {

constexpr int kChunkVox = 4096;        // 16^3 voxels, id = (z*16+y)*16+x  (Chunk.h:90-93)
constexpr int kRayStepGuard = 1 << 16; // the reference loop is unbounded

struct Pose {        // one per cloud
  float R[9], t[3];  // camera -> world (Twc)
  float Ri[9], ti[3];// Transform::inverse() of it
};

struct Params {
  float resolution, round_to_voxel, half_voxel, rounding, diag;
  float tq, tl, tc, ts, weight;
  int shard_rank, shard_count;
};

PLVS_HD float sum3(float a, float b, float c) { return a + (b + c); }
PLVS_HD float sqnorm3(float a, float b, float c) { return sum3(a * a, b * b, c * c); }

PLVS_HD void xform(const float* R, const float* t, float px, float py, float pz, float* out) {
  out[0] = t[0] + sum3(R[0] * px, R[1] * py, R[2] * pz);
  out[1] = t[1] + sum3(R[3] * px, R[4] * py, R[5] * pz);
  out[2] = t[2] + sum3(R[6] * px, R[7] * py, R[8] * pz);
}

PLVS_HD float cof3(const float* m, int i, int j) {
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m[3 * i1 + j1] * m[3 * i2 + j2] - m[3 * i1 + j2] * m[3 * i2 + j1];
}

// Eigen Transform<float,3,Affine>::inverse(): cofactor inverse, t' = (-R^-1) t.
PLVS_HD void make_pose(const float* Twc /*3x4 row-major*/, Pose* p) {
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) p->R[3 * i + j] = Twc[4 * i + j];
    p->t[i] = Twc[4 * i + 3];
  }
  const float* R = p->R;
  const float c00 = cof3(R, 0, 0), c10 = cof3(R, 1, 0), c20 = cof3(R, 2, 0);
  const float det = sum3(c00 * R[0], c10 * R[3], c20 * R[6]);
  const float invdet = 1.0f / det;
  float* Ri = p->Ri;
  Ri[0] = c00 * invdet; Ri[1] = c10 * invdet; Ri[2] = c20 * invdet;
  Ri[3] = cof3(R, 0, 1) * invdet; Ri[4] = cof3(R, 1, 1) * invdet; Ri[5] = cof3(R, 2, 1) * invdet;
  Ri[6] = cof3(R, 0, 2) * invdet; Ri[7] = cof3(R, 1, 2) * invdet; Ri[8] = cof3(R, 2, 2) * invdet;
  for (int i = 0; i < 3; ++i)
    p->ti[i] = sum3((-Ri[3 * i + 0]) * p->t[0], (-Ri[3 * i + 1]) * p->t[1], (-Ri[3 * i + 2]) * p->t[2]);
}

// Raycast.cpp:12-15: fmod(fmod(v,1)+1, 1) evaluated in double (the unqualified
// call picks the double overload), returned as float.  fmod(a,1) == a - trunc(a)
// exactly for the magnitudes involved, which keeps this off the slow ocml fmod.
PLVS_HD float rc_mod1(float value) {
  double r = (double)value;
  r = r - trunc(r);
  r = r + 1.0;
  r = r - trunc(r);
  return (float)r;
}
// Raycast.cpp:17-30
PLVS_HD float rc_intbound(float s, float ds) {
  if (ds < 0) { s = -s; ds = -ds; }
  s = rc_mod1(s);
  return (1 - s) / ds;
}
PLVS_HD int rc_signum(int x) { return (x > 0) ? 1 : ((x < 0) ? -1 : 0); }

// QuadraticTruncator::GetTruncationDistance floored by 2*sqrt(3)*res (Chisel.cpp:479).
PLVS_HD float truncation_of(const Params& P, float depth) {
  const float q = (P.tq * depth * depth + P.tl * depth + P.tc) * P.ts;
  return q > P.diag ? q : P.diag;
}

// Everything a ray needs (Chisel.cpp:472-488).
struct Ray {
  float start[3], end[3];
  float depth, truncation;
};

PLVS_HD bool make_ray(const Params& P, const Pose& pose, float px, float py, float pz, Ray* r) {
  if (pz < 0.01f) return false;  // Chisel.cpp:475
  float w[3];
  xform(pose.R, pose.t, px, py, pz, w);
  const float v0 = w[0] - pose.t[0], v1 = w[1] - pose.t[1], v2 = w[2] - pose.t[2];
  const float z2 = sqnorm3(v0, v1, v2);
  float d0 = v0, d1 = v1, d2 = v2;
  if (z2 > 0.0f) {
    const float nrm = sqrtf(z2);
    d0 = v0 / nrm; d1 = v1 / nrm; d2 = v2 / nrm;
  }
  const float tr = truncation_of(P, pz);
  const float d[3] = {d0, d1, d2};
  for (int k = 0; k < 3; ++k) {
    const float swp = w[k] * P.round_to_voxel;
    const float sdt = d[k] * tr * P.round_to_voxel;
    r->start[k] = swp - sdt;
    r->end[k] = swp + sdt;
  }
  r->depth = pz;
  r->truncation = tr;
  return true;
}

// The "world cloud with normals" flavour (Chisel::IntegrateWorldPointCloudWithNormals, Chisel.cpp:238-376: what
// PointCloudMapChisel::LoadMap integrates the saved map through): the segment point -/+ 4 voxels along the point's
// normal, no depth test, constant truncation.
struct RayN {
  float wp[3], dir[3];   // world point, unit normal
};

PLVS_HD void make_ray_normal(const Params& P, const Pose& pose, float px, float py, float pz, float nx, float ny,
                             float nz, Ray* r, RayN* a) {
  xform(pose.R, pose.t, px, py, pz, a->wp);                 // cameraPose * point, :276
  const float z2 = sqnorm3(nx, ny, nz);
  a->dir[0] = nx; a->dir[1] = ny; a->dir[2] = nz;
  if (z2 > 0.0f) {                                          // normal.normalized(), :282
    const float nrm = sqrtf(z2);
    a->dir[0] = nx / nrm; a->dir[1] = ny / nrm; a->dir[2] = nz / nrm;
  }
  const float tr = 4 * P.resolution;                        // :266
  for (int k = 0; k < 3; ++k) {
    const float swp = a->wp[k] * P.round_to_voxel;          // :288
    const float sdt = a->dir[k] * tr * P.round_to_voxel;    // :289
    r->start[k] = swp - sdt;
    r->end[k] = swp + sdt;
  }
  r->depth = 0.f;
  r->truncation = tr;
}

// ChunkHasher (ChunkManager.h:42-54), size_t arithmetic; used only for sharding.
PLVS_HD uint64_t chunk_hash(int x, int y, int z) {
  return ((uint64_t)(int64_t)x * 73856093ull) ^ ((uint64_t)(int64_t)y * 19349663ull) ^
         ((uint64_t)(int64_t)z * 83492791ull);
}

// hash % count without a 64-bit division (count is the number of GPUs: small, usually a power of two).
PLVS_HD int shard_of(uint64_t h, int count) {
  const uint32_t c = (uint32_t)count;
  if ((c & (c - 1u)) == 0u) return (int)((uint32_t)h & (c - 1u));
  const uint32_t hi = (uint32_t)(h >> 32) % c, lo = (uint32_t)h % c;
  const uint32_t two32 = ((0xFFFFFFFFu % c) + 1u) % c;   // 2^32 mod c
  return (int)(((unsigned long long)hi * two32 + lo) % c);
}

// Ownership of the chunk a walk is in, re-evaluated only when the chunk changes (a ray crosses one or
// two chunk boundaries in a dozen steps).
struct OwnerCache {
  int cx = INT_MIN, cy = 0, cz = 0;
  bool owned = false;
};
PLVS_HD bool chunk_owned(const Params& P, int cx, int cy, int cz, OwnerCache* oc) {
  if (P.shard_count <= 1) return true;
  if (cx != oc->cx || cy != oc->cy || cz != oc->cz) {
    oc->cx = cx; oc->cy = cy; oc->cz = cz;
    oc->owned = shard_of(chunk_hash(cx, cy, cz), P.shard_count) == P.shard_rank;
  }
  return oc->owned;
}

// What one raycast voxel resolves to (Chisel.cpp:505-531).
struct Visit {
  int cx, cy, cz;  // chunk id
  int vid;         // voxel id inside the chunk
  float u;         // signed distance along the ray
};

PLVS_HD float signed_dist(const Pose& pose, float depth, float c0, float c1, float c2) {
  float cc[3];
  xform(pose.Ri, pose.ti, c0, c1, c2, cc);
  const float length = sqrtf(sqnorm3(cc[0], cc[1], cc[2]));
  return length * (depth / cc[2] - 1);
}

// |voxel coordinate| below which the reference's float chunk lookup (GetIDAt on the voxel centre,
// ChunkManager.h:192-198: floor(((v * res + res/2)) * (1 / (16 res)))) provably equals the integer
// v >> 4: the exact value (v + 0.5) / 16 is at least 1/32 away from every integer and the three
// roundings perturb it by less than 2^-22 relative, i.e. by less than 1/32 while |v| < 2^21.  Rays
// that leave this range (50 km at 5 cm) make the call fail with kErrCoordRange instead of diverging.
constexpr float kVoxelCoordLimit = 1048576.0f - 128.0f;   // the walk can overshoot a ray's ends by a few voxels

PLVS_HD bool ray_in_coord_range(const Ray& r) {
  bool ok = true;
  for (int k = 0; k < 3; ++k)
    ok = ok && (fabsf(r.start[k]) < kVoxelCoordLimit) && (fabsf(r.end[k]) < kVoxelCoordLimit);
  return ok;
}

// Returns true when the voxel takes an update (owned, |u| < truncation).  Chunk id and local voxel
// id come from the integer voxel coordinates (see kVoxelCoordLimit).
PLVS_HD bool resolve_visit(const Params& P, const Pose& pose, const Ray& ray, int vx, int vy, int vz,
                           Visit* v, OwnerCache* oc) {
  v->cx = vx >> 4;
  v->cy = vy >> 4;
  v->cz = vz >> 4;
  if (!chunk_owned(P, v->cx, v->cy, v->cz, oc)) return false;
  v->vid = ((vz & 15) * 16 + (vy & 15)) * 16 + (vx & 15);
  const float c0 = (float)vx * P.resolution + P.half_voxel;
  const float c1 = (float)vy * P.resolution + P.half_voxel;
  const float c2 = (float)vz * P.resolution + P.half_voxel;
  v->u = signed_dist(pose, ray.depth, c0, c1, c2);
  return fabsf(v->u) < ray.truncation;
}

// The same for the normals flavour: u = (centre - worldPoint) . dir (Chisel.cpp:329).
PLVS_HD bool resolve_visit_normal(const Params& P, const RayN& a, const Ray& ray, int vx, int vy, int vz, Visit* v,
                                  OwnerCache* oc) {
  v->cx = vx >> 4;
  v->cy = vy >> 4;
  v->cz = vz >> 4;
  if (!chunk_owned(P, v->cx, v->cy, v->cz, oc)) return false;
  v->vid = ((vz & 15) * 16 + (vy & 15)) * 16 + (vx & 15);
  const float c0 = (float)vx * P.resolution + P.half_voxel;
  const float c1 = (float)vy * P.resolution + P.half_voxel;
  const float c2 = (float)vz * P.resolution + P.half_voxel;
  v->u = sum3((c0 - a.wp[0]) * a.dir[0], (c1 - a.wp[1]) * a.dir[1], (c2 - a.wp[2]) * a.dir[2]);
  return fabsf(v->u) < ray.truncation;
}

// chisel::Raycast as a resumable cursor (Raycast.cpp:65-182).
struct RayCursor {
  int x, y, z, endX, endY, endZ, stepX, stepY, stepZ;
  float tMaxX, tMaxY, tMaxZ, tDeltaX, tDeltaY, tDeltaZ, maxDist;
  float sx, sy, sz;
  int guard;
  bool done;
};

PLVS_HD void ray_begin(const Ray& r, RayCursor* c) {
  c->sx = r.start[0]; c->sy = r.start[1]; c->sz = r.start[2];
  c->x = (int)floorf(r.start[0]); c->y = (int)floorf(r.start[1]); c->z = (int)floorf(r.start[2]);
  c->endX = (int)floorf(r.end[0]); c->endY = (int)floorf(r.end[1]); c->endZ = (int)floorf(r.end[2]);
  c->maxDist = sqnorm3(r.end[0] - r.start[0], r.end[1] - r.start[1], r.end[2] - r.start[2]);
  const float dx = (float)(c->endX - c->x), dy = (float)(c->endY - c->y), dz = (float)(c->endZ - c->z);
  c->stepX = rc_signum((int)dx); c->stepY = rc_signum((int)dy); c->stepZ = rc_signum((int)dz);
  c->tMaxX = rc_intbound(r.start[0], dx);
  c->tMaxY = rc_intbound(r.start[1], dy);
  c->tMaxZ = rc_intbound(r.start[2], dz);
  c->tDeltaX = ((float)c->stepX) / dx;
  c->tDeltaY = ((float)c->stepY) / dy;
  c->tDeltaZ = ((float)c->stepZ) / dz;
  c->guard = 0;
  c->done = (c->stepX == 0 && c->stepY == 0 && c->stepZ == 0);
}

// Sharded maps: can the walk that starts at `c` emit a voxel of a chunk this rank owns?  Conservative
// and cheap (no walk).  Along an axis the walk only moves in the direction of its step; every emitted
// voxel but the last is within sqrt(maxDist) of the start point (the loop's own stop test) and the
// last is one step further, so an axis advances at most L = ceil(sqrt(maxDist)) + 1 voxels (+1 below
// for the rounding of the square root).  That box covers at most a few chunks.
PLVS_HD bool walk_may_touch_owned(const Params& P, const Ray& r) {
  // start voxel, step signs and stop distance exactly as ray_begin derives them (without its divisions)
  const int x = (int)floorf(r.start[0]), y = (int)floorf(r.start[1]), z = (int)floorf(r.start[2]);
  const int sx = rc_signum((int)floorf(r.end[0]) - x), sy = rc_signum((int)floorf(r.end[1]) - y),
            sz = rc_signum((int)floorf(r.end[2]) - z);
  const float maxDist = sqnorm3(r.end[0] - r.start[0], r.end[1] - r.start[1], r.end[2] - r.start[2]);
  const int L = (int)ceilf(sqrtf(maxDist)) + 2;
  const int x1 = x + sx * L, y1 = y + sy * L, z1 = z + sz * L;
  const int cx0 = (x < x1 ? x : x1) >> 4, cx1 = (x < x1 ? x1 : x) >> 4;
  const int cy0 = (y < y1 ? y : y1) >> 4, cy1 = (y < y1 ? y1 : y) >> 4;
  const int cz0 = (z < z1 ? z : z1) >> 4, cz1 = (z < z1 ? z1 : z) >> 4;
  for (int cz = cz0; cz <= cz1; ++cz)
    for (int cy = cy0; cy <= cy1; ++cy)
      for (int cx = cx0; cx <= cx1; ++cx)
        if (shard_of(chunk_hash(cx, cy, cz), P.shard_count) == P.shard_rank) return true;
  return false;
}

// Emits the current voxel into (vx,vy,vz) and advances; false when the ray is exhausted.
// Straight-line form of the loop body of Raycast.cpp:115-180 (selects instead of the if / else
// ladder: on the GPU the ladder costs more exec-mask bookkeeping than arithmetic); the
// comparisons, their order and the one addition per step are the reference's.
PLVS_HD bool ray_next(RayCursor* c, int* vx, int* vy, int* vz) {
  if (c->done) return false;
  *vx = c->x; *vy = c->y; *vz = c->z;
  // (the reference's bounds test against -/+INT_MAX only rejects INT_MAX itself)
  const float d = sqnorm3((float)c->x - c->sx, (float)c->y - c->sy, (float)c->z - c->sz);
  const bool stop = (d > c->maxDist) || ((c->x == c->endX) && (c->y == c->endY) && (c->z == c->endZ));
  const bool x_lt_y = c->tMaxX < c->tMaxY, x_lt_z = c->tMaxX < c->tMaxZ, y_lt_z = c->tMaxY < c->tMaxZ;
  // (lane-mask logic: pure comparisons, no side effects — the compiler keeps them as mask operations)
  const bool go_x = !stop && x_lt_y && x_lt_z;
  const bool go_y = !stop && !x_lt_y && y_lt_z;
  const bool go_z = !stop && ((x_lt_y && !x_lt_z) || (!x_lt_y && !y_lt_z));
  c->x += go_x ? c->stepX : 0;
  c->y += go_y ? c->stepY : 0;
  c->z += go_z ? c->stepZ : 0;
  c->tMaxX = go_x ? c->tMaxX + c->tDeltaX : c->tMaxX;
  c->tMaxY = go_y ? c->tMaxY + c->tDeltaY : c->tMaxY;
  c->tMaxZ = go_z ? c->tMaxZ + c->tDeltaZ : c->tMaxZ;
  c->guard += stop ? 0 : 1;
  c->done = stop | (c->guard >= kRayStepGuard);
  return true;
}

// DistVoxel::Integrate (DistVoxel.h:91-99) with the product w_u*u already rounded.
PLVS_HD void dist_update(float& sdf, float& w, float wu_times_u, float wu) {
  const float oldSDF = sdf, oldW = w;
  sdf = (oldW * oldSDF + wu_times_u) / (wu + oldW);
  w = oldW + wu;
}

// dist_update with the division taken off the sdf -> sdf dependency: the divisor
// (the new weight) does not depend on the running sdf, so y = RN(1/wn) is computed
// ahead of the chain and the quotient is recovered exactly from it,
//   q = RN(a*y), r = a - q*wn (exact in one fma), sdf = RN(q + r*y) = RN(a / wn)
// (the classic correction step of reciprocal-based division: with a correctly
// rounded reciprocal and no under/overflow in q, r it yields the correctly rounded
// quotient).  wn = wu + w (== w + wu) and y = RN(1/wn) are passed in; amin / amax collect the
// range of |a|.  Outside dist_update_rcp_exact's ranges the caller redoes the step(s) with
// dist_update.
PLVS_HD void dist_update_rcp(float& sdf, float& w, float wu_times_u, float wn, float y, float& amin,
                             float& amax) {
  const float a = w * sdf + wu_times_u;
  const float q = a * y;
  const float r = fmaf(-q, wn, a);
  sdf = fmaf(r, y, q);
  w = wn;
  amin = fminf(amin, fabsf(a));
  amax = fmaxf(amax, fabsf(a));
}
// the operand ranges in which dist_update_rcp is exact
PLVS_HD bool dist_update_rcp_exact(float amin, float amax, float wn_min, float wn_max) {
  return (amin >= 0x1p-60f) && (amax <= 0x1p60f) && (wn_min >= 0x1p-20f) && (wn_max <= 0x1p40f);
}

// ColorVoxel::IntegrateSimple(r,g,b,1) (ColorVoxel.h:91-110) on the packed
// r | g<<8 | b<<16 | weight<<24 word; a no-op once the weight reaches 254.
PLVS_HD void colour_update(uint32_t& rgbw, uint32_t r, uint32_t g, uint32_t b) {
  const uint32_t cw = rgbw >> 24;
  if (cw >= 254u) return;
  const float inv = 1.f / (float)(1u + cw);
  const uint32_t red = (uint32_t)(uint8_t)((float)(cw * (rgbw & 255u) + r) * inv);
  const uint32_t green = (uint32_t)(uint8_t)((float)(cw * ((rgbw >> 8) & 255u) + g) * inv);
  const uint32_t blue = (uint32_t)(uint8_t)((float)(cw * ((rgbw >> 16) & 255u) + b) * inv);
  rgbw = red | (green << 8) | (blue << 16) | ((cw + 1u) << 24);
}

// DistVoxel::Integrate + SetKfid + ColorVoxel::IntegrateSimple for one visit.
PLVS_HD void apply_update(float& sdf, float& w, uint32_t& kfid, uint32_t& rgbw, float u, float wu,
                          uint32_t new_kfid, uint32_t r, uint32_t g, uint32_t b) {
  dist_update(sdf, w, wu * u, wu);
  kfid = new_kfid;
  colour_update(rgbw, r, g, b);
}

// Conversions.h:118-121 (u8 * 1/255) followed by Chisel.cpp:536 ((uint8_t)(c * 255.0f)).
PLVS_HD uint32_t colour_roundtrip(uint32_t c) {
  const float byteToFloat = 1.0f / 255.0f;
  return (uint32_t)(uint8_t)(((float)c * byteToFloat) * 255.0f);
}

}  // namespace chisel
}  // namespace plvs
