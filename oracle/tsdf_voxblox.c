/*
 * oracle/tsdf_voxblox.c — CPU restatement of the voxblox TSDF integrate that
 * sits behind PointCloudMapVoxblox::InsertCloud ("simple" integrator, one
 * thread: the deterministic order of the reference).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under plvs_amd/ may call into this file.
 *
 * Parity status: PINNED by the reference's own sources, end to end but for the pose conversion.  voxblox's gtests
 * never call integratePointCloud and are disabled in the build, and the library as a whole cannot be
 * built here (Eigen3, glog, protobuf, kindr absent) — but its integrator sources compile UNMODIFIED against small
 * stand-ins for those headers (oracle/ref/vbx_shim -> oracle/_ref/libvoxblox_ref.so): src/integrator/tsdf_integrator.cc
 * (TsdfIntegratorBase::updateTsdfVoxel / computeDistance / getVoxelWeight / isPointValid /
 * allocateStorageAndGetVoxelPtr / integrateWorlPointCloud, SimpleTsdfIntegrator, MergedTsdfIntegrator) with Layer and
 * Block from the core headers, src/integrator/integrator_utils.cc (RayCaster, ThreadSafeIndex), mesh/marching_cubes.{h,cc}.
 * tests/test_oracle_pinned.py runs WHOLE CLOUDS (three key frames, 5 and 10 cm, carving off / on with points beyond
 * the 5 m ray limit; posed world clouds with un-normalised and zero normals) through the reference's Simple and Merged
 * integrators and integrateWorlPointCloud and through this file (+ tsdf_voxblox_merged.cpp) and compares every voxel
 * of every block bit for bit; the pieces (ray caster, visiting order, colour blend, indices, hash, map iteration
 * order, meshCube) are also compared one by one.  What the stand-ins encode rather than pin: Eigen's evaluation order
 * for sums of three, and the pose — the kindr stand-in applies a quaternion it is handed (computed by
 * quat_from_matrix below, restated from minkindr + Eigen 3.3's Quaternion.h), it does not convert the matrix.
 * One behaviour of the reference is NOT reproduced: integrateWorlPointCloud never calls updateLayerWithStoredBlocks,
 * so the blocks it creates stay in the integrator's temp_block_map_ and join the layer only with the next
 * integratePointCloud call (PointCloudMapVoxblox::LoadMap's UpdateMap therefore meshes none of them); here, and on the
 * device, they are in the map at once — the voxel payloads are identical.
 * The pose goes through kindr's quaternion as in the reference (quat_from_matrix / quat_transform below, restated
 * from minkindr — present in the tree — and Eigen 3.3's Quaternion.h, which is not).  The multi-threaded reference is itself
 * order-nondeterministic (per-voxel mutexes, ThreadSafeIndex); the oracle is the
 * integrator_threads = 1 schedule.  The "fast" integrator (PLVS's YAML default,
 * tsdf_integrator.cc:505-605) is racy at integrator_threads > 1 and lossy by design (two approximate hash sets); its
 * ONE-thread schedule is deterministic and restated here (oracle_voxblox_integrate_fast) twice over: with the reference's
 * own approximate sets (pinned bit for bit by the compiled FastTsdfIntegrator, tests/test_oracle_pinned.py) and with
 * collision-free sets — the same algorithm without the sets' losses, which is what the device builds.
 *
 * Follows (paths relative to the PLVS tree):
 *   src/PointCloudMapVoxblox.cc:48-99                      config + InsertCloud
 *   Thirdparty/voxblox_server/src/tsdf_server.cc:476-530   insertPointCloud (finite filter, colours)
 *   Thirdparty/voxblox/src/integrator/tsdf_integrator.cc
 *        :9-31    constructor (inverse sizes, allow_clear needs carving)
 *        :85-103  isPointValid        :114-157 allocateStorageAndGetVoxelPtr
 *        :173-232 updateTsdfVoxel     :240-253 computeDistance
 *        :255-264 getVoxelWeight      :266-327 SimpleTsdfIntegrator
 *   Thirdparty/voxblox/src/integrator/integrator_utils.cc
 *        :11-44   ThreadSafeIndex (mixed visiting order)
 *        :137-235 RayCaster
 *   Thirdparty/voxblox/include/voxblox/core/common.h:95-125 (Color::blendTwoColors),
 *        :140-222 (grid helpers, kEpsilon, signum) ; core/voxel.h:12-18 ; core/block_hash.h:15-26
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define VPS 16
#define BLOCK_VOX 4096

typedef struct {
  int32_t id[3];
  int used;
  int pending;     /* created by integrateWorlPointCloud, still in temp_block_map_ (only with defer_world_blocks) */
  float* distance; /* TsdfVoxel::distance, init 0 */
  float* weight;   /* TsdfVoxel::weight, init 0   */
  uint32_t* rgba;  /* Color r | g<<8 | b<<16 | a<<24, init 0 */
} vblock_t;

typedef struct oracle_voxblox {
  float voxel_size, voxel_size_inv, voxels_per_side_inv;
  float truncation, max_weight, min_ray, max_ray;
  int carving, allow_clear, use_const_weight, use_weight_dropoff;
  vblock_t* tab;
  size_t cap, count;
  int shard_rank, shard_count;
  int64_t last_visits;
  int defer_world_blocks, in_world_call;
  /* FastTsdfIntegrator's two ApproxHashSet<20, 10000> (allocated by the first fast call that asks for them) */
  uint64_t *approx_start, *approx_seen;
  size_t approx_offset;
} oracle_voxblox;

static size_t vb_hash(const int32_t id[3]) {
  /* AnyIndexHash (block_hash.h:21-24) */
  return ((size_t)(unsigned int)id[0] * 73856093u) ^ ((size_t)(int64_t)id[1] * 19349663u) ^
         ((size_t)(int64_t)id[2] * 83492791u);
}
/* sharding uses the same three-prime hash as the chisel path (sign-extended x) so
 * that one owner function serves both back ends */
static size_t owner_hash(const int32_t id[3]) {
  return ((size_t)(int64_t)id[0] * 73856093u) ^ ((size_t)(int64_t)id[1] * 19349663u) ^
         ((size_t)(int64_t)id[2] * 83492791u);
}

static vblock_t* vtab_find(vblock_t* tab, size_t cap, const int32_t id[3], int* found) {
  size_t h = vb_hash(id) & (cap - 1);
  for (;;) {
    vblock_t* c = &tab[h];
    if (!c->used) { *found = 0; return c; }
    if (c->id[0] == id[0] && c->id[1] == id[1] && c->id[2] == id[2]) { *found = 1; return c; }
    h = (h + 1) & (cap - 1);
  }
}

static void vtab_grow(oracle_voxblox* o) {
  size_t ncap = o->cap * 2;
  vblock_t* nt = (vblock_t*)calloc(ncap, sizeof(vblock_t));
  for (size_t i = 0; i < o->cap; i++)
    if (o->tab[i].used) {
      int f;
      *vtab_find(nt, ncap, o->tab[i].id, &f) = o->tab[i];
    }
  free(o->tab);
  o->tab = nt;
  o->cap = ncap;
}

static void vb_publish_pending(oracle_voxblox* o);

static vblock_t* vblock_get(oracle_voxblox* o, const int32_t id[3]) {
  int found;
  vblock_t* b = vtab_find(o->tab, o->cap, id, &found);
  if (found) return b;
  if ((o->count + 1) * 2 > o->cap) {
    vtab_grow(o);
    b = vtab_find(o->tab, o->cap, id, &found);
  }
  memcpy(b->id, id, sizeof(b->id));
  b->used = 1;
  b->pending = o->in_world_call && o->defer_world_blocks;
  b->distance = (float*)calloc(BLOCK_VOX, sizeof(float));
  b->weight = (float*)calloc(BLOCK_VOX, sizeof(float));
  b->rgba = (uint32_t*)calloc(BLOCK_VOX, sizeof(uint32_t));
  o->count++;
  return b;
}

/* PointCloudMapVoxblox.cc:52-71 passes truncation 0.1, max_weight 1e4, ray length
 * 0.1..5, weight drop-off on, constant weight off, allow_clear on. */
oracle_voxblox* oracle_voxblox_create(float voxel_size, float truncation, float max_weight,
                                      float min_ray, float max_ray, int carving, int shard_rank,
                                      int shard_count) {
  oracle_voxblox* o = (oracle_voxblox*)calloc(1, sizeof(*o));
  o->voxel_size = voxel_size;
  o->voxel_size_inv = (float)(1.0 / voxel_size);        /* tsdf_integrator.cc:17 */
  o->voxels_per_side_inv = (float)(1.0 / VPS);          /* :19 */
  o->truncation = truncation;
  o->max_weight = max_weight;
  o->min_ray = min_ray;
  o->max_ray = max_ray;
  o->carving = carving;
  o->allow_clear = carving ? 1 : 0;                     /* :26-28 */
  o->use_const_weight = 0;
  o->use_weight_dropoff = 1;
  o->cap = 1024;
  o->tab = (vblock_t*)calloc(o->cap, sizeof(vblock_t));
  o->shard_rank = shard_rank;
  o->shard_count = shard_count;
  return o;
}

void oracle_voxblox_clear(oracle_voxblox* o) {
  for (size_t i = 0; i < o->cap; i++)
    if (o->tab[i].used) { free(o->tab[i].distance); free(o->tab[i].weight); free(o->tab[i].rgba); }
  memset(o->tab, 0, o->cap * sizeof(vblock_t));
  o->count = 0;
}

void oracle_voxblox_destroy(oracle_voxblox* o) {
  if (o) { free(o->approx_start); free(o->approx_seen); o->approx_start = o->approx_seen = NULL; }
  if (!o) return;
  oracle_voxblox_clear(o);
  free(o->tab);
  free(o);
}

static float sum3(float a, float b, float c) { return a + (b + c); }
static float norm3(const float v[3]) { return sqrtf(sum3(v[0] * v[0], v[1] * v[1], v[2] * v[2])); }
static int vb_signum(float x) { return (x == 0) ? 0 : x < 0 ? -1 : 1; }

/* ThreadSafeIndex::getMixedIndex (integrator_utils.cc:33-44), step_size_ = 1024 */
static size_t mixed_index(size_t base_idx, size_t number_of_points) {
  const size_t step = 1024, groups = number_of_points / step;
  if (groups * step <= base_idx) return base_idx;
  return (base_idx % groups) * step + base_idx / groups;
}

/* Color::blendTwoColors (common.h:106-124) on the packed word. */
static uint32_t blend(uint32_t c1, float w1, uint32_t c2, float w2) {
  const float total = w1 + w2;
  w1 /= total;
  w2 /= total;
  uint32_t out = 0;
  for (int k = 0; k < 4; k++) {
    const uint8_t a = (uint8_t)(c1 >> (8 * k)), b = (uint8_t)(c2 >> (8 * k));
    out |= (uint32_t)(uint8_t)round((double)(a * w1 + b * w2)) << (8 * k);
  }
  return out;
}

/* updateTsdfVoxel (tsdf_integrator.cc:173-232) */
static void update_voxel(const oracle_voxblox* o, const float origin[3], const float pG[3],
                         const int32_t g[3], uint32_t color, float weight, float* distance,
                         float* vweight, uint32_t* vcolor) {
  const float center[3] = {((float)g[0] + 0.5f) * o->voxel_size, ((float)g[1] + 0.5f) * o->voxel_size,
                           ((float)g[2] + 0.5f) * o->voxel_size};
  /* computeDistance (:240-253) */
  const float vvo[3] = {center[0] - origin[0], center[1] - origin[1], center[2] - origin[2]};
  const float vpo[3] = {pG[0] - origin[0], pG[1] - origin[1], pG[2] - origin[2]};
  const float dist_G = norm3(vpo);
  const float dist_G_V = sum3(vvo[0] * vpo[0], vvo[1] * vpo[1], vvo[2] * vpo[2]) / dist_G;
  const float sdf = dist_G - dist_G_V;
  float updated_weight = weight;
  const float dropoff_epsilon = o->voxel_size;
  if (o->use_weight_dropoff && sdf < -dropoff_epsilon) {
    updated_weight = weight * (o->truncation + sdf) / (o->truncation - dropoff_epsilon);
    updated_weight = (updated_weight < 0.0f) ? 0.0f : updated_weight; /* std::max(uw, 0.0f) */
  }
  const float new_weight = *vweight + updated_weight;
  if (new_weight < 1e-6f) return;
  const float new_sdf = (sdf * updated_weight + *distance * *vweight) / new_weight;
  if (fabsf(sdf) < o->truncation) *vcolor = blend(*vcolor, *vweight, color, updated_weight);
  /* std::min(trunc, x) = (x < trunc) ? x : trunc ; std::max(-trunc, x) = (-trunc < x) ? x : -trunc */
  *distance = (new_sdf > 0.0) ? ((new_sdf < o->truncation) ? new_sdf : o->truncation)
                              : ((-o->truncation < new_sdf) ? new_sdf : -o->truncation);
  *vweight = (new_weight < o->max_weight) ? new_weight : o->max_weight;
}

/* RayCaster (integrator_utils.cc:137-235): the constructor SimpleTsdfIntegrator uses + setupRayCaster, nextRayIndex. */
typedef struct {
  int32_t cur[3], sgn[3];
  float t_next[3], t_step[3];
  int steps; /* ray_length_in_steps_: the ray emits steps + 1 voxels */
} vb_ray;

static void vb_ray_setup_scaled(const float ss[3], const float es[3], vb_ray* r);

static void vb_ray_setup(const float origin[3], const float pG[3], int is_clearing, int carving, float max_ray,
                         float voxel_size_inv, float truncation, vb_ray* r) {
  const float d[3] = {pG[0] - origin[0], pG[1] - origin[1], pG[2] - origin[2]};
  const float dn = norm3(d);
  float unit[3] = {d[0], d[1], d[2]};
  if (sum3(d[0] * d[0], d[1] * d[1], d[2] * d[2]) > 0.0f) { unit[0] = d[0] / dn; unit[1] = d[1] / dn; unit[2] = d[2] / dn; }
  float ray_start[3], ray_end[3];
  if (is_clearing) {
    float ray_length = dn;
    float tmp = ray_length - truncation;
    tmp = (tmp < 0.0f) ? 0.0f : tmp;                     /* std::max(x, 0) */
    ray_length = (max_ray < tmp) ? max_ray : tmp;        /* std::min(x, max_ray) */
    for (int k = 0; k < 3; k++) {
      ray_end[k] = origin[k] + unit[k] * ray_length;
      ray_start[k] = carving ? origin[k] : ray_end[k];
    }
  } else {
    for (int k = 0; k < 3; k++) {
      ray_end[k] = pG[k] + unit[k] * truncation;
      ray_start[k] = carving ? origin[k] : (pG[k] - unit[k] * truncation);
    }
  }
  float ss[3], es[3];
  for (int k = 0; k < 3; k++) { ss[k] = ray_start[k] * voxel_size_inv; es[k] = ray_end[k] * voxel_size_inv; }
  vb_ray_setup_scaled(ss, es, r);
}

/* setupRayCaster (:196-235); also RayCaster(start_scaled, end_scaled) (:169-173) */
static void vb_ray_setup_scaled(const float ss[3], const float es[3], vb_ray* r) {
  int32_t endi[3];
  r->steps = 0;
  for (int k = 0; k < 3; k++) {
    r->cur[k] = (int32_t)floorf(ss[k] + 1e-6f);          /* getGridIndexFromPoint (common.h:147-151) */
    endi[k] = (int32_t)floorf(es[k] + 1e-6f);
    r->steps += abs(endi[k] - r->cur[k]);
    const float rs = es[k] - ss[k];
    r->sgn[k] = vb_signum(rs);
    const float corrected = (float)(r->sgn[k] > 0 ? r->sgn[k] : 0);
    const float shifted = ss[k] - (float)r->cur[k];
    const float dist = corrected - shifted;
    r->t_next[k] = dist / rs;          /* the `abs(x) < 0.0 ? 2.0 :` guards can never fire */
    r->t_step[k] = (float)r->sgn[k] / rs;
  }
}

static void vb_ray_next(vb_ray* r, int32_t g[3]) {
  g[0] = r->cur[0]; g[1] = r->cur[1]; g[2] = r->cur[2];
  int mi = 0; /* Eigen minCoeff(&idx): first coefficient, replaced only by a strictly smaller one */
  if (r->t_next[1] < r->t_next[mi]) mi = 1;
  if (r->t_next[2] < r->t_next[mi]) mi = 2;
  r->cur[mi] += r->sgn[mi];
  r->t_next[mi] += r->t_step[mi];
}

/* getBlockIndexFromGlobalVoxelIndex (common.h:175-186), getLocalFromGlobalVoxelIndex (:190-198), 16 voxels per side */
static void block_index(const int32_t g[3], float voxels_per_side_inv, int32_t bid[3]) {
  for (int k = 0; k < 3; k++) bid[k] = (int32_t)floorf((float)g[k] * voxels_per_side_inv);
}
static void local_index(const int32_t g[3], int32_t l[3]) {
  const uint32_t off = 1u << 31;
  for (int k = 0; k < 3; k++) l[k] = (int32_t)(((uint32_t)g[k] + off) & 15u);
}

/* T_G_C = kindr::minimal::QuatTransformationTemplate<float>(Twc.matrix()) (tsdf_server.cc:484-486): the rotation
 * block becomes an Eigen::Quaternionf (minkindr/.../quat-transformation-inl.h:72-75, rotation-quaternion-inl.h:112-116),
 * by Eigen 3.3's matrix -> quaternion assignment (Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl<Other,3,3>:
 * Shoemake's algorithm).  q = {w, x, y, z}. */
static void quat_from_matrix(const float m[9], float q[4]) {
  const float tr = m[0] + (m[4] + m[8]);   /* trace() = diagonal().sum(): the length-3 unrolled redux a0 + (a1 + a2) */
  if (tr > 0.0f) {
    float s = sqrtf(tr + 1.0f);
    q[0] = 0.5f * s;
    s = 0.5f / s;
    q[1] = (m[7] - m[5]) * s;                                 /* (m21 - m12) */
    q[2] = (m[2] - m[6]) * s;                                 /* (m02 - m20) */
    q[3] = (m[3] - m[1]) * s;                                 /* (m10 - m01) */
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    float s = sqrtf(m[4 * i] - m[4 * j] - m[4 * k] + 1.0f);
    q[1 + i] = 0.5f * s;
    s = 0.5f / s;
    q[0] = (m[3 * k + j] - m[3 * j + k]) * s;
    q[1 + j] = (m[3 * j + i] + m[3 * i + j]) * s;
    q[1 + k] = (m[3 * k + i] + m[3 * i + k]) * s;
  }
}

/* T * p = q.rotate(p) + t (quat-transformation-inl.h:159-162; rotate = Eigen's Quaternion * Vector3,
 * QuaternionBase::_transformVector: uv = 2 (q.vec x v);  v + w uv + q.vec x uv, evaluated coefficient-wise as
 * (v + w * uv) + cross). */
static void quat_transform(const float q[4], const float t[3], const float v[3], float out[3]) {
  const float w = q[0], x = q[1], y = q[2], z = q[3];
  float uv[3] = {y * v[2] - z * v[1], z * v[0] - x * v[2], x * v[1] - y * v[0]};
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  const float c[3] = {y * uv[2] - z * uv[1], z * uv[0] - x * uv[2], x * uv[1] - y * uv[0]};
  for (int k = 0; k < 3; k++) out[k] = ((v[k] + w * uv[k]) + c[k]) + t[k];
}

/* SimpleTsdfIntegrator::integratePointCloud with integrator_threads = 1.
 * Twc: 3x4 row-major [R|t]; rgba: n x 4 u8 (r,g,b,a members of the pcl point). */
void oracle_voxblox_integrate(oracle_voxblox* o, const float* xyz, const uint8_t* rgba, int n,
                              const float* Twc) {
  float R[9], t[3];
  vb_publish_pending(o);   /* updateLayerWithStoredBlocks, tsdf_integrator.cc:306 */
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) R[3 * i + j] = Twc[4 * i + j];
    t[i] = Twc[4 * i + 3];
  }
  float q[4];
  quat_from_matrix(R, q);
  /* tsdf_server.cc:509-527: drop non-finite points, keep order */
  int* keep = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  size_t m = 0;
  for (int i = 0; i < n; i++)
    if (isfinite(xyz[3 * i]) && isfinite(xyz[3 * i + 1]) && isfinite(xyz[3 * i + 2])) keep[m++] = i;
  int64_t visits = 0;
  vblock_t* last_block = NULL;
  int32_t last_bid[3] = {0, 0, 0};
  for (size_t seq = 0; seq < m; seq++) {
    const int pi = keep[mixed_index(seq, m)];
    const float* pC = xyz + 3 * (size_t)pi;
    uint32_t color;
    memcpy(&color, rgba + 4 * (size_t)pi, 4);
    /* isPointValid (:85-103), freespace_points = false */
    const float ray_distance = norm3(pC);
    int is_clearing;
    if (ray_distance < o->min_ray) continue;
    else if (ray_distance > o->max_ray) {
      if (o->allow_clear) is_clearing = 1;
      else continue;
    } else
      is_clearing = 0;
    const float* origin = t;
    float pG[3];
    quat_transform(q, t, pC, pG);                            /* T_G_C * point_C (tsdf_integrator.cc:289) */
    vb_ray ray;
    vb_ray_setup(origin, pG, is_clearing, o->carving, o->max_ray, o->voxel_size_inv, o->truncation, &ray);
    const float weight = o->use_const_weight ? 1.0f : (fabsf(pC[2]) > 1e-6f ? 1.0f / (pC[2] * pC[2]) : 0.0f);
    /* nextRayIndex (:180-194): emits ray_length_in_steps + 1 voxels */
    for (int step = 0; step <= ray.steps; step++) {
      int32_t g[3];
      vb_ray_next(&ray, g);
      /* allocateStorageAndGetVoxelPtr (:114-157) */
      int32_t bid[3];
      block_index(g, o->voxels_per_side_inv, bid);
      if (o->shard_count > 1 && (int)(owner_hash(bid) % (size_t)o->shard_count) != o->shard_rank) continue;
      if (!last_block || last_bid[0] != bid[0] || last_bid[1] != bid[1] || last_bid[2] != bid[2]) {
        last_block = vblock_get(o, bid); /* may rehash: never keep a pointer across another lookup */
        memcpy(last_bid, bid, sizeof(last_bid));
      }
      int32_t l[3];
      local_index(g, l);
      const int vid = l[0] + VPS * (l[1] + l[2] * VPS);
      update_voxel(o, origin, pG, g, color, weight, &last_block->distance[vid], &last_block->weight[vid],
                   &last_block->rgba[vid]);
      visits++;
    }
  }
  free(keep);
  o->last_visits = visits;
}

/* ---- FastTsdfIntegrator (tsdf_integrator.cc:505-605), integrator_threads = 1, max_integration_time_s = its default
 * (no limit), start_voxel_subsampling_factor 2, max_consecutive_ray_collisions 2, clear_checks_every_n_frames 1 as
 * PointCloudMapVoxblox.cc:70-72 sets them.  Per point in the mixed order: a ray starts only where no earlier ray of THIS
 * scan started (a set of voxels of half the voxel size); it is cast from the surface end towards the sensor
 * (cast_from_origin = false: setupRayCaster(end_scaled, start_scaled), integrator_utils.cc:164-168) and stops at the
 * third voxel in a row that an earlier ray of the scan has gone through; every voxel before that takes updateTsdfVoxel.
 * The two sets:
 *   approx_sets = 1  ApproxHashSet<20, 10000> (utils/approx_hash_array.h:66-160) as one thread sees it: slot
 *                    (hash & 0xFFFFF) + offset holds the full hash of whatever index came last; "reset" = offset + 1
 *                    (a stale entry can never match: its slot moved), everything zeroed every 10 000 scans;
 *   approx_sets = 0  collision-free sets: the algorithm without the sets' false negatives / positives. */
typedef struct { int32_t* key; uint8_t* used; size_t cap, n; } vb_idxset;
static void idxset_init(vb_idxset* s, size_t cap) {
  s->cap = cap; s->n = 0;
  s->key = (int32_t*)malloc(sizeof(int32_t) * 3 * cap);
  s->used = (uint8_t*)calloc(cap, 1);
}
static int idxset_insert_raw(vb_idxset* s, const int32_t g[3]) {   /* 1 = new */
  size_t h = (vb_hash(g) * 0x9E3779B97F4A7C15ull >> 20) & (s->cap - 1);
  for (;;) {
    if (!s->used[h]) {
      s->used[h] = 1;
      memcpy(s->key + 3 * h, g, 3 * sizeof(int32_t));
      s->n++;
      return 1;
    }
    if (s->key[3 * h] == g[0] && s->key[3 * h + 1] == g[1] && s->key[3 * h + 2] == g[2]) return 0;
    h = (h + 1) & (s->cap - 1);
  }
}
static int idxset_insert(vb_idxset* s, const int32_t g[3]) {
  if ((s->n + 1) * 2 > s->cap) {
    vb_idxset t;
    idxset_init(&t, s->cap * 2);
    for (size_t i = 0; i < s->cap; i++)
      if (s->used[i]) idxset_insert_raw(&t, s->key + 3 * i);
    free(s->key); free(s->used);
    *s = t;
  }
  return idxset_insert_raw(s, g);
}
#define VB_APPROX_BITS 20
#define VB_APPROX_RESET 10000
static void approx_clear(uint64_t* a) {
  memset(a, 0, sizeof(uint64_t) * (((size_t)1 << VB_APPROX_BITS) + VB_APPROX_RESET));
  a[0] = UINT64_MAX;   /* pseudo_set_[offset_ = 0] = max (approx_hash_array.h:78, 148) */
}
static int approx_replace(uint64_t* a, size_t offset, const int32_t g[3]) {   /* replaceHash (:105-114): 1 = new */
  const uint64_t hash = (uint64_t)vb_hash(g);
  uint64_t* slot = a + (hash & (((uint64_t)1 << VB_APPROX_BITS) - 1)) + offset;
  if (*slot == hash) return 0;
  *slot = hash;
  return 1;
}

static int32_t* g_fast_updates_out = NULL;   /* (test hook: updates per mixed position of the next fast call) */
void oracle_voxblox_fast_record_updates(int32_t* out) { g_fast_updates_out = out; }
void oracle_voxblox_integrate_fast(oracle_voxblox* o, const float* xyz, const uint8_t* rgba, int n, const float* Twc,
                                   int approx_sets) {
  float R[9], t[3];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) R[3 * i + j] = Twc[4 * i + j];
    t[i] = Twc[4 * i + 3];
  }
  float q[4];
  quat_from_matrix(R, q);
  vb_publish_pending(o);   /* (as oracle_voxblox_integrate) */
  /* integratePointCloud (:575-605): the sets are "reset" at the start of every scan */
  vb_idxset start_set, seen_set;
  if (approx_sets) {
    const size_t words = ((size_t)1 << VB_APPROX_BITS) + VB_APPROX_RESET;
    if (!o->approx_start) {
      o->approx_start = (uint64_t*)malloc(sizeof(uint64_t) * words);
      o->approx_seen = (uint64_t*)malloc(sizeof(uint64_t) * words);
      approx_clear(o->approx_start);
      approx_clear(o->approx_seen);
      o->approx_offset = 0;
    }
    if (++o->approx_offset >= VB_APPROX_RESET) {   /* resetApproxSet (:141-150), both sets in step */
      approx_clear(o->approx_start);
      approx_clear(o->approx_seen);
      o->approx_offset = 0;
    }
  } else {
    idxset_init(&start_set, 1 << 12);
    idxset_init(&seen_set, 1 << 14);
  }
  int* keep = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  size_t m = 0;
  for (int i = 0; i < n; i++)   /* tsdf_server.cc:509-527 */
    if (isfinite(xyz[3 * i]) && isfinite(xyz[3 * i + 1]) && isfinite(xyz[3 * i + 2])) keep[m++] = i;
  int64_t visits = 0;
  const float start_inv = 2.0f * o->voxel_size_inv;   /* config_.start_voxel_subsampling_factor * voxel_size_inv_ (:540) */
  vblock_t* last_block = NULL;
  int32_t last_bid[3] = {0, 0, 0};
  for (size_t seq = 0; seq < m; seq++) {
    const int pi = keep[mixed_index(seq, m)];
    const float* pC = xyz + 3 * (size_t)pi;
    uint32_t color;
    memcpy(&color, rgba + 4 * (size_t)pi, 4);
    const float ray_distance = norm3(pC);
    int is_clearing;
    if (ray_distance < o->min_ray) continue;
    else if (ray_distance > o->max_ray) {
      if (o->allow_clear) is_clearing = 1;
      else continue;
    } else
      is_clearing = 0;
    const float* origin = t;
    float pG[3];
    quat_transform(q, t, pC, pG);
    int32_t sv[3];
    for (int k = 0; k < 3; k++) sv[k] = (int32_t)floorf(pG[k] * start_inv + 1e-6f);   /* getGridIndexFromPoint(p, inv) */
    if (!(approx_sets ? approx_replace(o->approx_start, o->approx_offset, sv) : idxset_insert(&start_set, sv))) continue;
    /* RayCaster(..., cast_from_origin = false): the same two end points, cast the other way round */
    vb_ray ray;
    {
      const float d[3] = {pG[0] - origin[0], pG[1] - origin[1], pG[2] - origin[2]};
      const float dn = norm3(d);
      float unit[3] = {d[0], d[1], d[2]};
      if (sum3(d[0] * d[0], d[1] * d[1], d[2] * d[2]) > 0.0f) { unit[0] = d[0] / dn; unit[1] = d[1] / dn; unit[2] = d[2] / dn; }
      float rs[3], re[3], ss[3], es[3];
      if (is_clearing) {
        float tmp = dn - o->truncation;
        tmp = (tmp < 0.0f) ? 0.0f : tmp;
        const float len = (o->max_ray < tmp) ? o->max_ray : tmp;
        for (int k = 0; k < 3; k++) { re[k] = origin[k] + unit[k] * len; rs[k] = o->carving ? origin[k] : re[k]; }
      } else {
        for (int k = 0; k < 3; k++) {
          re[k] = pG[k] + unit[k] * o->truncation;
          rs[k] = o->carving ? origin[k] : (pG[k] - unit[k] * o->truncation);
        }
      }
      for (int k = 0; k < 3; k++) { ss[k] = rs[k] * o->voxel_size_inv; es[k] = re[k] * o->voxel_size_inv; }
      vb_ray_setup_scaled(es, ss, &ray);
    }
    const float weight = o->use_const_weight ? 1.0f : (fabsf(pC[2]) > 1e-6f ? 1.0f / (pC[2] * pC[2]) : 0.0f);
    int64_t collisions = 0;
    const int64_t visits_before = visits;
    for (int step = 0; step <= ray.steps; step++) {
      int32_t g[3];
      vb_ray_next(&ray, g);
      if (!(approx_sets ? approx_replace(o->approx_seen, o->approx_offset, g) : idxset_insert(&seen_set, g))) ++collisions;
      else collisions = 0;
      if (collisions > 2) break;   /* max_consecutive_ray_collisions */
      int32_t bid[3];
      block_index(g, o->voxels_per_side_inv, bid);
      if (o->shard_count > 1 && (int)(owner_hash(bid) % (size_t)o->shard_count) != o->shard_rank) continue;
      if (!last_block || last_bid[0] != bid[0] || last_bid[1] != bid[1] || last_bid[2] != bid[2]) {
        last_block = vblock_get(o, bid);
        memcpy(last_bid, bid, sizeof(last_bid));
      }
      int32_t l[3];
      local_index(g, l);
      const int vid = l[0] + VPS * (l[1] + l[2] * VPS);
      update_voxel(o, origin, pG, g, color, weight, &last_block->distance[vid], &last_block->weight[vid],
                   &last_block->rgba[vid]);
      visits++;
    }
    if (g_fast_updates_out) g_fast_updates_out[seq] = (int32_t)(visits - visits_before);
  }
  g_fast_updates_out = NULL;
  free(keep);
  if (!approx_sets) { free(start_set.key); free(start_set.used); free(seen_set.key); free(seen_set.used); }
  vb_publish_pending(o);   /* updateLayerWithStoredBlocks (:601) */
  o->last_visits = visits;
}

/* ---- A model of how the device reaches the one-thread schedule of the fast integrator without running it ray after ray
 * (test infrastructure like the rest of this file; tests/test_tsdf_voxblox_fast.py checks it against the loop above).
 * A ray's fate depends on the set queries of the rays before it; the set is a table of slots that remember the LAST
 * index asked for, so "was index x seen?" = "does the previous query of x's slot, in (ray, step) order, carry x's
 * hash?".  Given how many queries every ray makes (Q), all answers follow from one stable sort of the queries by slot;
 * from the answers every ray finds its own stopping point; repeat until no Q changes.  Ray r's answers depend only on
 * rays < r and its own earlier steps, so the first ray is right after one round, the fixed point is unique and it is the
 * sequential schedule.  Returns the number of rounds; Q_out[seq] = queries of the ray at mixed position seq (0: no ray),
 * L_out[seq] = voxels it updates. */
typedef struct { uint32_t slot; uint32_t q; } vb_query_key;
static int vb_query_cmp(const void* a, const void* b) {
  const vb_query_key *x = (const vb_query_key*)a, *y = (const vb_query_key*)b;
  if (x->slot != y->slot) return x->slot < y->slot ? -1 : 1;
  return x->q < y->q ? -1 : (x->q > y->q ? 1 : 0);   /* (q grows with (ray, step): the stable order) */
}
int oracle_voxblox_fast_model(const oracle_voxblox* o, const float* xyz, int n, const float* Twc, int first_window,
                              int32_t* Q_out, int32_t* L_out) {
  float R[9], t[3], q[4];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) R[3 * i + j] = Twc[4 * i + j];
    t[i] = Twc[4 * i + 3];
  }
  quat_from_matrix(R, q);
  const uint64_t mask = ((uint64_t)1 << VB_APPROX_BITS) - 1;
  const float start_inv = 2.0f * o->voxel_size_inv;
  /* the start set: every valid point asks once, in the mixed order */
  vb_ray* rays = (vb_ray*)malloc(sizeof(vb_ray) * (size_t)(n > 0 ? n : 1));
  uint64_t* sh = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(n > 0 ? n : 1));
  vb_query_key* sk = (vb_query_key*)malloc(sizeof(vb_query_key) * (size_t)(n > 0 ? n : 1));
  int ns = 0;
  for (int seq = 0; seq < n; seq++) {
    Q_out[seq] = 0;
    L_out[seq] = 0;
    const int pi = (int)mixed_index((size_t)seq, (size_t)n);
    const float* pC = xyz + 3 * (size_t)pi;
    const float ray_distance = norm3(pC);
    int is_clearing;
    if (ray_distance < o->min_ray) continue;
    else if (ray_distance > o->max_ray) {
      if (o->allow_clear) is_clearing = 1;
      else continue;
    } else
      is_clearing = 0;
    float pG[3];
    quat_transform(q, t, pC, pG);
    int32_t sv[3];
    for (int k = 0; k < 3; k++) sv[k] = (int32_t)floorf(pG[k] * start_inv + 1e-6f);
    sh[seq] = (uint64_t)vb_hash(sv);
    sk[ns].slot = (uint32_t)(sh[seq] & mask);
    sk[ns].q = (uint32_t)seq;
    ns++;
    {
      const float d[3] = {pG[0] - t[0], pG[1] - t[1], pG[2] - t[2]};
      const float dn = norm3(d);
      float unit[3] = {d[0], d[1], d[2]};
      if (sum3(d[0] * d[0], d[1] * d[1], d[2] * d[2]) > 0.0f) { unit[0] = d[0] / dn; unit[1] = d[1] / dn; unit[2] = d[2] / dn; }
      float rs[3], re[3], ss[3], es[3];
      if (is_clearing) {
        float tmp = dn - o->truncation;
        tmp = (tmp < 0.0f) ? 0.0f : tmp;
        const float len = (o->max_ray < tmp) ? o->max_ray : tmp;
        for (int k = 0; k < 3; k++) { re[k] = t[k] + unit[k] * len; rs[k] = o->carving ? t[k] : re[k]; }
      } else {
        for (int k = 0; k < 3; k++) {
          re[k] = pG[k] + unit[k] * o->truncation;
          rs[k] = o->carving ? t[k] : (pG[k] - unit[k] * o->truncation);
        }
      }
      for (int k = 0; k < 3; k++) { ss[k] = rs[k] * o->voxel_size_inv; es[k] = re[k] * o->voxel_size_inv; }
      vb_ray_setup_scaled(es, ss, &rays[seq]);
    }
    Q_out[seq] = -1;   /* valid; alive or not is decided below */
  }
  qsort(sk, (size_t)ns, sizeof(vb_query_key), vb_query_cmp);
  for (int p = 0; p < ns; p++) {
    const int seq = (int)sk[p].q;
    const int dup = p > 0 && sk[p - 1].slot == sk[p].slot && sh[sk[p - 1].q] == sh[seq];
    Q_out[seq] = dup ? 0 : (rays[seq].steps + 1 < first_window ? rays[seq].steps + 1 : first_window);
  }
  /* the observed set: rounds */
  int rounds = 0;
  size_t cap = 0;
  vb_query_key* qk = NULL;
  uint64_t* qh = NULL;
  uint8_t* seen = NULL;
  for (;;) {
    rounds++;
    size_t M = 0;
    for (int seq = 0; seq < n; seq++) M += (size_t)Q_out[seq];
    if (M > cap) {
      cap = M + M / 2 + 16;
      qk = (vb_query_key*)realloc(qk, sizeof(vb_query_key) * cap);
      qh = (uint64_t*)realloc(qh, sizeof(uint64_t) * cap);
      seen = (uint8_t*)realloc(seen, cap);
    }
    size_t at = 0;
    for (int seq = 0; seq < n; seq++) {
      vb_ray r = rays[seq];
      for (int s = 0; s < Q_out[seq]; s++) {
        int32_t g[3];
        vb_ray_next(&r, g);
        qh[at] = (uint64_t)vb_hash(g);
        qk[at].slot = (uint32_t)(qh[at] & mask);
        qk[at].q = (uint32_t)at;
        at++;
      }
    }
    qsort(qk, M, sizeof(vb_query_key), vb_query_cmp);
    for (size_t p = 0; p < M; p++)
      seen[qk[p].q] = (uint8_t)(p > 0 && qk[p - 1].slot == qk[p].slot && qh[qk[p - 1].q] == qh[qk[p].q]);
    int changed = 0;
    at = 0;
    for (int seq = 0; seq < n; seq++) {
      const int Q = Q_out[seq], full = Q ? rays[seq].steps + 1 : 0;
      int collisions = 0, newQ = Q, L = Q;
      for (int s = 0; s < Q; s++) {
        collisions = seen[at + (size_t)s] ? collisions + 1 : 0;
        if (collisions > 2) { newQ = s + 1; L = s; break; }
      }
      if (newQ == Q && L == Q && Q < full) newQ = full;   /* no stop inside the window: look at the whole ray next round */
      at += (size_t)Q;
      if (newQ != Q) changed = 1;
      Q_out[seq] = newQ;
      L_out[seq] = L;
    }
    if (!changed) break;
  }
  free(rays); free(sh); free(sk); free(qk); free(qh); free(seen);
  return rounds;
}

/* ---- MergedTsdfIntegrator (tsdf_integrator.cc:329-492), integrator_threads = 1.  The bundling — points grouped by
 * the voxel they end in, in the iteration order of the reference's std::unordered_map — is oracle/tsdf_voxblox_merged.cpp
 * (it IS a std::unordered_map with the reference's hash); the pieces it needs from this file: */

/* isPointValid (:85-103), freespace_points = false: 0 = skipped, 1 = normal ray, 2 = clearing ray */
int oracle_voxblox_point_kind(const oracle_voxblox* o, const float* pC) {
  const float ray_distance = norm3(pC);
  if (ray_distance < o->min_ray) return 0;
  if (ray_distance > o->max_ray) return o->allow_clear ? 2 : 0;
  return 1;
}

/* getGridIndexFromPoint(T_G_C * point_C, voxel_size_inv) (:372-374; common.h:152-157) */
void oracle_voxblox_point_voxel(const oracle_voxblox* o, const float* Twc, const float* pC, int32_t* g) {
  float R[9], t[3], q[4], pG[3];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) R[3 * i + j] = Twc[4 * i + j];
    t[i] = Twc[4 * i + 3];
  }
  quat_from_matrix(R, q);
  quat_transform(q, t, pC, pG);
  for (int k = 0; k < 3; k++) g[k] = (int32_t)floorf(pG[k] * o->voxel_size_inv + 1e-6f);
}

/* getVoxelWeight (:255-264), use_const_weight = false */
float oracle_voxblox_point_weight(const float* pC) { return fabsf(pC[2]) > 1e-6f ? 1.0f / (pC[2] * pC[2]) : 0.0f; }

/* The ray part of integrateVoxel (:418-446) for n merged bundles, in the given order: RayCaster(origin,
 * T_G_C * merged_point_C, clearing_ray, ...) and updateTsdfVoxel with the merged colour and weight.  enable_anti_grazing
 * is off (src/PointCloudMapVoxblox.cc:67). */
void oracle_voxblox_integrate_bundles(oracle_voxblox* o, const float* merged_C, const uint32_t* colours, const float* weights,
                                      const uint8_t* clearing, int n, const float* Twc) {
  float R[9], t[3], q[4];
  vb_publish_pending(o);   /* updateLayerWithStoredBlocks, tsdf_integrator.cc:343 */
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) R[3 * i + j] = Twc[4 * i + j];
    t[i] = Twc[4 * i + 3];
  }
  quat_from_matrix(R, q);
  int64_t visits = 0;
  vblock_t* last_block = NULL;
  int32_t last_bid[3] = {0, 0, 0};
  for (int b = 0; b < n; b++) {
    float pG[3];
    quat_transform(q, t, merged_C + 3 * (size_t)b, pG);
    vb_ray ray;
    vb_ray_setup(t, pG, clearing[b] != 0, o->carving, o->max_ray, o->voxel_size_inv, o->truncation, &ray);
    for (int step = 0; step <= ray.steps; step++) {
      int32_t g[3];
      vb_ray_next(&ray, g);
      int32_t bid[3];
      block_index(g, o->voxels_per_side_inv, bid);
      if (o->shard_count > 1 && (int)(owner_hash(bid) % (size_t)o->shard_count) != o->shard_rank) continue;
      if (!last_block || last_bid[0] != bid[0] || last_bid[1] != bid[1] || last_bid[2] != bid[2]) {
        last_block = vblock_get(o, bid);
        memcpy(last_bid, bid, sizeof(last_bid));
      }
      int32_t l[3];
      local_index(g, l);
      const int vid = l[0] + VPS * (l[1] + l[2] * VPS);
      update_voxel(o, t, pG, g, colours[b], weights[b], &last_block->distance[vid], &last_block->weight[vid],
                   &last_block->rgba[vid]);
      visits++;
    }
  }
  o->last_visits = visits;
}

/* TsdfIntegratorBase::integrateWorlPointCloud (tsdf_integrator.cc:35-82), what PointCloudMapVoxblox::LoadMap feeds the
 * saved cloud through (src/PointCloudMapVoxblox.cc:233-258 -> TsdfServer::insertWorldPointCloud, tsdf_server.cc:577-660,
 * T = identity there): points in CLOUD order (no ThreadSafeIndex), each casting point + normal * truncation ->
 * point - normal * truncation, every voxel updated with weight 1 and ray_start in the role of the sensor origin.  No
 * isPointValid test.  The normal goes through T_G_C * normal_C — the full transformation, translation included, as the
 * reference writes it.  normals: n x 3.  Restated by reading (RayCaster, blend, indices are the pinned pieces). */
void oracle_voxblox_integrate_world_normals(oracle_voxblox* o, const float* xyz, const uint8_t* rgba, const float* normals,
                                            int n, const float* Twc) {
  float R[9], t[3];
  o->in_world_call = 1;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) R[3 * i + j] = Twc[4 * i + j];
    t[i] = Twc[4 * i + 3];
  }
  float q[4];
  quat_from_matrix(R, q);
  int64_t visits = 0;
  vblock_t* last_block = NULL;
  int32_t last_bid[3] = {0, 0, 0};
  for (int pi = 0; pi < n; pi++) {
    const float* pC = xyz + 3 * (size_t)pi;
    if (!(isfinite(pC[0]) && isfinite(pC[1]) && isfinite(pC[2]))) continue;   /* tsdf_server.cc:617-622 */
    uint32_t color;
    memcpy(&color, rgba + 4 * (size_t)pi, 4);
    const float* nC = normals + 3 * (size_t)pi;
    float nn[3] = {nC[0], nC[1], nC[2]};
    const float z2 = sum3(nC[0] * nC[0], nC[1] * nC[1], nC[2] * nC[2]);
    if (z2 > 0.0f) { const float l = sqrtf(z2); nn[0] = nC[0] / l; nn[1] = nC[1] / l; nn[2] = nC[2] / l; }   /* :44 */
    float pG[3], nG[3];
    quat_transform(q, t, pC, pG);                              /* :47 */
    quat_transform(q, t, nn, nG);                              /* :48 */
    float ray_start[3], ray_end[3], ss[3], es[3];
    for (int k = 0; k < 3; k++) {
      ray_start[k] = pG[k] + nG[k] * o->truncation;            /* :52 */
      ray_end[k] = pG[k] - nG[k] * o->truncation;              /* :53 */
      ss[k] = ray_start[k] * o->voxel_size_inv;                /* :55-56 */
      es[k] = ray_end[k] * o->voxel_size_inv;
    }
    vb_ray ray;
    vb_ray_setup_scaled(ss, es, &ray);                         /* RayCaster(start_scaled, end_scaled), :62 */
    for (int step = 0; step <= ray.steps; step++) {
      int32_t g[3];
      vb_ray_next(&ray, g);
      int32_t bid[3];
      block_index(g, o->voxels_per_side_inv, bid);
      if (o->shard_count > 1 && (int)(owner_hash(bid) % (size_t)o->shard_count) != o->shard_rank) continue;
      if (!last_block || last_bid[0] != bid[0] || last_bid[1] != bid[1] || last_bid[2] != bid[2]) {
        last_block = vblock_get(o, bid);
        memcpy(last_bid, bid, sizeof(last_bid));
      }
      int32_t l[3];
      local_index(g, l);
      const int vid = l[0] + VPS * (l[1] + l[2] * VPS);
      update_voxel(o, ray_start, pG, g, color, 1.0f, &last_block->distance[vid], &last_block->weight[vid],
                   &last_block->rgba[vid]);                    /* :78 */
      visits++;
    }
  }
  o->last_visits = visits;
  o->in_world_call = 0;
}

int64_t oracle_voxblox_last_visits(const oracle_voxblox* o) { return o->last_visits; }
/* The layer's blocks: with defer_world_blocks, the ones integrateWorlPointCloud created are not among them until the
 * next integratePointCloud (updateLayerWithStoredBlocks, tsdf_integrator.cc:306 / :343; never called by :35-82). */
int oracle_voxblox_num_chunks(const oracle_voxblox* o) {
  int k = 0;
  for (size_t i = 0; i < o->cap; i++) k += o->tab[i].used && !o->tab[i].pending;
  return k;
}
void oracle_voxblox_chunk_ids(const oracle_voxblox* o, int32_t* ids) {
  size_t k = 0;
  for (size_t i = 0; i < o->cap; i++)
    if (o->tab[i].used && !o->tab[i].pending) { memcpy(ids + 3 * k, o->tab[i].id, 3 * sizeof(int32_t)); k++; }
}
static void vb_publish_pending(oracle_voxblox* o) {
  for (size_t i = 0; i < o->cap; i++) o->tab[i].pending = 0;
}
void oracle_voxblox_set_deferred_world_blocks(oracle_voxblox* o, int enable) {
  o->defer_world_blocks = enable != 0;
  if (!enable) vb_publish_pending(o);
}
int oracle_voxblox_get_chunk(const oracle_voxblox* o, int cx, int cy, int cz, float* distance,
                             float* weight, uint32_t* rgba) {
  const int32_t id[3] = {cx, cy, cz};
  int found;
  vblock_t* b = vtab_find(o->tab, o->cap, id, &found);
  if (!found || b->pending) return 0;
  memcpy(distance, b->distance, BLOCK_VOX * sizeof(float));
  memcpy(weight, b->weight, BLOCK_VOX * sizeof(float));
  memcpy(rgba, b->rgba, BLOCK_VOX * sizeof(uint32_t));
  return 1;
}

/* T_G_C * point_C alone (tests/test_tsdf_voxblox.py checks it against R p + t on rotations that take every branch
 * of quat_from_matrix). */
void oracle_voxblox_transform(const float* Twc, const float* p, float* out) {
  float R[9], t[3], q[4];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) R[3 * i + j] = Twc[4 * i + j];
    t[i] = Twc[4 * i + 3];
  }
  quat_from_matrix(R, q);
  quat_transform(q, t, p, out);
}

/* The pose as the kindr transformation holds it: (w, x, y, z) of quat_from_matrix — handed to the reference's own
 * integrators compiled into oracle/_ref (their kindr stand-in takes the quaternion, not the matrix). */
void oracle_voxblox_pose_quat(const float* Twc, float* q_wxyz) {
  float R[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R[3 * i + j] = Twc[4 * i + j];
  quat_from_matrix(R, q_wxyz);
}

/* ---- the pieces above one by one, for tests/test_oracle_pinned.py: checked there against voxblox's own
 * integrator_utils.cc / common.h / block_hash.h compiled into oracle/_ref/libvoxblox_ref.so (oracle/ref/). */
int oracle_voxblox_raycast(const float* origin, const float* point_G, int is_clearing, int carving, float max_ray,
                           float voxel_size_inv, float truncation, int32_t* out, int cap) {
  vb_ray r;
  vb_ray_setup(origin, point_G, is_clearing, carving, max_ray, voxel_size_inv, truncation, &r);
  for (int s = 0; s <= r.steps; s++) {
    int32_t g[3];
    vb_ray_next(&r, g);
    if (s < cap) { out[3 * s] = g[0]; out[3 * s + 1] = g[1]; out[3 * s + 2] = g[2]; }
  }
  return r.steps + 1;
}
void oracle_voxblox_mixed_order(int n, int64_t* out) {
  for (int i = 0; i < n; i++) out[i] = (int64_t)mixed_index((size_t)i, (size_t)n);
}
uint32_t oracle_voxblox_blend(uint32_t c1, float w1, uint32_t c2, float w2) { return blend(c1, w1, c2, w2); }
void oracle_voxblox_indices(const int32_t* g, int32_t* block, int32_t* local, uint64_t* hash) {
  block_index(g, (float)(1.0 / VPS), block);
  local_index(g, local);
  *hash = (uint64_t)vb_hash(block);
}

/* ---------------------------------------------------------------- surface extraction
 * MeshIntegrator<TsdfVoxel>::updateMeshForBlock (Thirdparty/voxblox/include/voxblox/mesh/mesh_integrator.h:231-251)
 * as TsdfServer::updateMesh (Thirdparty/voxblox_server/src/tsdf_server.cc:775-787) runs it on every block whose
 * updated() flag is set, PointCloudMapVoxblox::UpdateMap (src/PointCloudMapVoxblox.cc:160-179):
 *   extractBlockMesh            :165-229   the walk over the block: 15^3 interior (x outer, z inner), then the
 *                                          max-X (z, y), max-Y (z, x < 15) and max-Z (y < 15, x < 15) planes
 *   extractMeshInsideBlock / OnBorder  :253-346   the eight corners (neighbour blocks in +x/+y/+z), getSdfIfValid
 *                                          (utils/meshing_utils.h:15-23: weight <= min_weight is unobserved)
 *   MarchingCubes::meshCube     mesh/marching_cubes.h:66-102 (vertices col+2, col+1, col; flat triangle normal),
 *                               interpolateEdgeVertices :117-134, interpolateVertex :138-153
 *   updateMeshColor             :348-368   nearest voxel of THIS block (computeVoxelIndexFromCoordinates clamps
 *                                          into the block, core/block.h:60-70, so the neighbour branch is dead)
 * Block geometry: origin = index * block_size (layer.h:122-126, common.h:186-191), voxel centre =
 * origin + (index + 0.5) * voxel_size with the sum and product in double (common.h:179-184).
 * The marching-cubes table is the one open_chisel uses (mc_table.inc); tests/test_oracle_pinned.py compares it,
 * the edge pairs and meshCube itself with voxblox's marching_cubes.{h,cc} compiled into oracle/_ref.
 * Pinned: the reference's own MeshIntegrator::updateMeshForBlock (mesh_integrator.h with Layer / Block / Mesh, compiled
 * into oracle/_ref against the stand-ins) on maps its SimpleTsdfIntegrator built, against oracle_voxblox_mesh_block on
 * the oracle's copy — every block's vertices, normals and colours byte for byte (tests/test_oracle_pinned.py). */
static const int kVbTriangleTable[256 * 16] = {
#include "mc_table.inc"
};
static const int kVbEdgeIndexPairs[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6},
                                             {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
static const int kVbCubeIndexOffsets[3][8] = {{0, 1, 1, 0, 0, 1, 1, 0},   /* mesh_integrator.h:100-102 */
                                              {0, 0, 1, 1, 0, 0, 1, 1},
                                              {0, 0, 0, 0, 1, 1, 1, 1}};
#define VB_MESH_MIN_WEIGHT 1e-4f   /* MeshIntegratorConfig::min_weight, mesh_integrator.h:49 (tsdf_server.cc:323-328 keeps it) */

typedef struct {
  float* vertices; float* normals;
  int n, cap;
} vb_mesh_out;

static void vb_mesh_cube(const float cc[8][3], const float sdf[8], vb_mesh_out* m) {
  int index = 0;
  for (int i = 0; i < 8; i++) index |= (sdf[i] < 0) ? (1 << i) : 0;
  if (index == 0) return;
  float edge[12][3];
  memset(edge, 0, sizeof edge);
  for (int i = 0; i < 12; i++) {
    const int e0 = kVbEdgeIndexPairs[i][0], e1 = kVbEdgeIndexPairs[i][1];
    if ((sdf[e0] < 0 && sdf[e1] >= 0) || (sdf[e0] >= 0 && sdf[e1] < 0)) {
      const float diff = sdf[e0] - sdf[e1];
      if (fabsf(diff) >= 1e-6f) {
        const float t = sdf[e0] / diff;
        for (int k = 0; k < 3; k++) edge[i][k] = cc[e0][k] + t * (cc[e1][k] - cc[e0][k]);
      } else {
        for (int k = 0; k < 3; k++) edge[i][k] = 0.5f * (cc[e0][k] + cc[e1][k]);
      }
    }
  }
  const int* row = kVbTriangleTable + 16 * index;
  for (int col = 0; row[col] != -1; col += 3) {
    if (m->n + 3 > m->cap) { m->n += 3; continue; }   /* counted, not stored */
    float* p0 = m->vertices + 3 * (size_t)m->n;
    float *p1 = p0 + 3, *p2 = p0 + 6;
    for (int k = 0; k < 3; k++) {
      p0[k] = edge[row[col + 2]][k];
      p1[k] = edge[row[col + 1]][k];
      p2[k] = edge[row[col]][k];
    }
    const float px[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
    const float py[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
    float n[3] = {px[1] * py[2] - px[2] * py[1], px[2] * py[0] - px[0] * py[2], px[0] * py[1] - px[1] * py[0]};
    const float z = sum3(n[0] * n[0], n[1] * n[1], n[2] * n[2]);
    if (z > 0.0f) { const float s = sqrtf(z); n[0] /= s; n[1] /= s; n[2] /= s; }   /* normalized() */
    for (int v = 0; v < 3; v++)
      for (int k = 0; k < 3; k++) m->normals[3 * (size_t)(m->n + v) + k] = n[k];
    m->n += 3;
  }
}

static const vblock_t* vb_mesh_find(const oracle_voxblox* o, int x, int y, int z) {
  const int32_t id[3] = {x, y, z};
  int found = 0;
  vblock_t* b = vtab_find(o->tab, o->cap, id, &found);
  return found && !b->pending ? b : NULL;
}

static void vb_block_origin(const oracle_voxblox* o, const vblock_t* b, float org[3]) {
  const float block_size = o->voxel_size * (float)VPS;   /* layer.h:34 */
  for (int k = 0; k < 3; k++) org[k] = (float)b->id[k] * block_size;
}

static void vb_mesh_voxel(const oracle_voxblox* o, const vblock_t* blk, int x, int y, int z, vb_mesh_out* m) {
  float org[3], coords[3];
  vb_block_origin(o, blk, org);
  const int v[3] = {x, y, z};
  for (int k = 0; k < 3; k++) coords[k] = org[k] + (float)(((double)(float)v[k] + 0.5) * (double)o->voxel_size);
  float cc[8][3], sdf[8];
  for (int i = 0; i < 8; i++) {
    int c[3] = {x + kVbCubeIndexOffsets[0][i], y + kVbCubeIndexOffsets[1][i], z + kVbCubeIndexOffsets[2][i]};
    const vblock_t* src = blk;
    if (c[0] >= VPS || c[1] >= VPS || c[2] >= VPS) {   /* :303-337 (offsets are never negative) */
      int off[3] = {0, 0, 0};
      for (int j = 0; j < 3; j++)
        if (c[j] >= VPS) { off[j] = 1; c[j] -= VPS; }
      src = vb_mesh_find(o, blk->id[0] + off[0], blk->id[1] + off[1], blk->id[2] + off[2]);
      if (!src) return;
    }
    const int id = c[0] + VPS * (c[1] + c[2] * VPS);
    if (src->weight[id] <= VB_MESH_MIN_WEIGHT) return;
    sdf[i] = src->distance[id];
    for (int k = 0; k < 3; k++) cc[i][k] = coords[k] + (float)kVbCubeIndexOffsets[k][i] * o->voxel_size;
  }
  vb_mesh_cube(cc, sdf, m);
}

/* voxblox::Mesh of block (bx, by, bz) after updateMeshForBlock: vertices / normals n x 3 f32, colors n x 4 u8
 * (r, g, b, a).  Returns n (0 for a block that does not exist), writing at most `cap` vertices. */
int oracle_voxblox_mesh_block(const oracle_voxblox* o, int bx, int by, int bz, float* vertices, float* normals,
                              uint8_t* colors, int cap) {
  const vblock_t* blk = vb_mesh_find(o, bx, by, bz);
  if (!blk) return 0;
  vb_mesh_out m = {vertices, normals, 0, cap};
  for (int x = 0; x < VPS - 1; x++)
    for (int y = 0; y < VPS - 1; y++)
      for (int z = 0; z < VPS - 1; z++) vb_mesh_voxel(o, blk, x, y, z, &m);
  for (int z = 0; z < VPS; z++)                                   /* max X plane */
    for (int y = 0; y < VPS; y++) vb_mesh_voxel(o, blk, VPS - 1, y, z, &m);
  for (int z = 0; z < VPS; z++)                                   /* max Y plane */
    for (int x = 0; x < VPS - 1; x++) vb_mesh_voxel(o, blk, x, VPS - 1, z, &m);
  for (int y = 0; y < VPS - 1; y++)                               /* max Z plane */
    for (int x = 0; x < VPS - 1; x++) vb_mesh_voxel(o, blk, x, y, VPS - 1, &m);
  const int n = m.n < cap ? m.n : cap;
  float org[3];
  vb_block_origin(o, blk, org);
  for (int i = 0; i < n; i++) {                                   /* updateMeshColor */
    int idx[3];
    for (int k = 0; k < 3; k++) {
      int g = (int)floorf((vertices[3 * (size_t)i + k] - org[k]) * o->voxel_size_inv + 1e-6f);
      idx[k] = g < 0 ? 0 : (g > VPS - 1 ? VPS - 1 : g);
    }
    const int id = idx[0] + VPS * (idx[1] + idx[2] * VPS);
    uint32_t c = 0;                                               /* Color(): r = g = b = a = 0 */
    if (!(blk->weight[id] <= VB_MESH_MIN_WEIGHT)) c = blk->rgba[id];
    for (int k = 0; k < 4; k++) colors[4 * (size_t)i + k] = (uint8_t)(c >> (8 * k));
  }
  return m.n;
}

/* getMeshAsPointcloud (voxblox_ros/tsdf_server.h:84-87 -> mesh_vis.h:272-318, ColorMode::kColor): every colour
 * channel goes through a float in [0, 1] and back (conversions.h:44-60). */
uint8_t oracle_voxblox_cloud_colour(uint8_t c) {
  const float msg = (float)((double)c / 255.0);
  return (uint8_t)((double)msg * 255.0);
}

/* Test hooks (no counterpart in the reference): one cube through meshCube; overwrite / create one block. */
int oracle_voxblox_mesh_cube(const float* corner_coords /* 8 x 3 */, const float* corner_sdf, float* vertices,
                             float* normals) {
  float cc[8][3];
  memcpy(cc, corner_coords, sizeof cc);
  vb_mesh_out m = {vertices, normals, 0, 15};
  vb_mesh_cube(cc, corner_sdf, &m);
  return m.n;
}
void oracle_voxblox_mc_tables(int* triangle_table, int* edge_index_pairs) {
  memcpy(triangle_table, kVbTriangleTable, sizeof(kVbTriangleTable));
  memcpy(edge_index_pairs, kVbEdgeIndexPairs, sizeof(kVbEdgeIndexPairs));
}
void oracle_voxblox_set_block(oracle_voxblox* o, int bx, int by, int bz, const float* distance, const float* weight,
                              const uint32_t* rgba) {
  const int32_t id[3] = {bx, by, bz};
  vblock_t* b = vblock_get(o, id);
  memcpy(b->distance, distance, BLOCK_VOX * sizeof(float));
  memcpy(b->weight, weight, BLOCK_VOX * sizeof(float));
  memcpy(b->rgba, rgba, BLOCK_VOX * sizeof(uint32_t));
}
