/*
 * oracle/tsdf_chisel.c — CPU restatement of the open_chisel point-cloud
 * integrate that sits behind PointCloudMapChisel::InsertCloud.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under plvs_amd/ may call into this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load it.
 *
 * Parity status: PINNED END TO END by the reference's own sources.  open_chisel ships no tests and no golden
 * vectors, and the real Eigen3 / PCL are absent here, but ALL SIXTEEN sources under
 * Thirdparty/open_chisel/src compile UNMODIFIED against an Eigen stand-in (oracle/ref/eigen_full/: fixed-size
 * vectors, 3x3 / 4x4 matrices, the affine Transform, Quaternion) into oracle/_ref/libchisel_full_ref.so.
 * tests/test_oracle_pinned_chisel_map.py drives chisel::Chisel there as ChiselServer does — several key frames
 * through IntegratePointCloudWidthDepth at three resolutions, the same with carving by depth images,
 * IntegrateWorldPointCloudWithNormals on posed clouds, UpdateMeshes — and compares EVERY voxel of every chunk
 * (sdf, weight, kfid, colour, colour weight) and every mesh vertex / normal / colour / kfid of every chunk with
 * this file's results, bit for bit.  Chunk creation / garbage collection, the frustum and CarveWithDepth, the
 * loop glue of Chisel.cpp:442-585 and ChunkManager::RecomputeMesh therefore come from the reference's source.
 * The pieces are also pinned one by one (oracle/_ref/libchisel_ref.so, tests/test_oracle_pinned.py):
 * Raycast.cpp on 40 000 rays incl. boundary starts, axis-aligned, diagonal, zero-length and 400-voxel rays;
 * DistVoxel::Integrate, ColorVoxel::Integrate / IntegrateSimple through saturation; QuadraticTruncator,
 * ConstantWeighter; the marching-cubes tables.
 * What the stand-in ENCODES rather than takes from Eigen, and so remains a reading (of Eigen 3.3, not of the
 * reference): the evaluation order of fixed-size expressions (3-vector reductions are a0 + (a1 + a2); an Affine
 * transform applied to a point is R*p + t, each row of R*p one such reduction; Transform::inverse() of an Affine
 * transform uses the cofactor inverse of the linear part).  The C-library overloads g++ picks for the
 * unqualified calls (fmod / sqrt resolve to the double versions) ARE the reference's, since its translation
 * units are what is compiled.  Not compiled: chisel_server (needs PCL); the handful of its lines on this path
 * (integrator set-up, colour bytes * 1/255) are repeated in oracle/ref/chisel_full_ref_wrap.cpp with citations.
 * Compiled with -ffp-contract=off so no FMA contraction changes a rounding.
 *
 * Follows (paths relative to the PLVS tree):
 *   src/PointCloudMapChisel.cc:76-98                  InsertCloud (pose -> chisel::Transform)
 *   Thirdparty/chisel_server/src/ChiselServer.cpp:50-69, 105-108, 561-586, 664-700
 *   Thirdparty/chisel_server/include/chisel_server/Conversions.h:97-127   colours = u8 * (1/255)
 *   Thirdparty/open_chisel/src/Chisel.cpp:442-585     IntegratePointCloudWidthDepth (cloud part)
 *   Thirdparty/open_chisel/src/geometry/Raycast.cpp:6-30, 65-182
 *   Thirdparty/open_chisel/include/open_chisel/ChunkManager.h:42-54, 161-206
 *   Thirdparty/open_chisel/src/ChunkManager.cpp:68, 89-93
 *   Thirdparty/open_chisel/src/Chunk.cpp:34-50, 96-105 ; include/open_chisel/Chunk.h:85-118
 *   Thirdparty/open_chisel/include/open_chisel/DistVoxel.h:91-99 ; src/DistVoxel.cpp:27-35
 *   Thirdparty/open_chisel/include/open_chisel/ColorVoxel.h:91-110
 *   Thirdparty/open_chisel/include/open_chisel/truncation/QuadraticTruncator.h:45-50
 *   Thirdparty/open_chisel/include/open_chisel/weighting/ConstantWeighter.h:43-46
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define CHUNK_VOX 4096
#define RAY_STEP_GUARD (1 << 16) /* the reference loop is unbounded; never reached on finite input */

typedef struct {
  int32_t id[3];
  int used;
  float* sdf;     /* DistVoxel::sdf, init 99999 */
  float* weight;  /* DistVoxel::weight, init 0  */
  uint32_t* kfid; /* DistVoxel::kfid, init 0    */
  uint32_t* rgbw; /* ColorVoxel r | g<<8 | b<<16 | weight<<24, init 0 */
  /* the EXACT mean of the same visits, kept beside the reference's f32 running mean when oracle_chisel_track_exact is
   * on: sum of w_u * u and sum of w_u in double (the membership |u| < truncation and every u are the reference's own
   * f32 values).  A checker of both the reference's sequential f32 drift and the order-free mode's rounding. */
  double* x_wu;
  double* x_w;
} chunk_t;

typedef struct oracle_chisel {
  float resolution;
  float tq, tl, tc, ts; /* QuadraticTruncator */
  float weight;         /* ConstantWeighter */
  float half_voxel;     /* ChunkManager.cpp:68  res * 0.5f */
  float rounding;       /* ChunkManager.cpp:91  1.0f / (16 * res) */
  chunk_t* tab;
  size_t cap, count;
  int shard_rank, shard_count;
  int64_t last_visits;
  int32_t last_new, last_updated;
  /* every raycast voxel's chunk id, in visiting order (GetOrCreateChunkAt is called for each, Chisel.cpp:505):
   * what oracle/tsdf_chisel_deform.cpp needs to keep the reference's std::unordered_map order */
  int track_exact;
  void (*visit_hook)(void* ctx, const int32_t id[3]);
  void* visit_hook_ctx;
  int32_t hook_last[3];
  int hook_has_last;
} oracle_chisel;

/* ChunkHasher (ChunkManager.h:42-54): size_t arithmetic on the int coordinates. */
static size_t chunk_hash(const int32_t id[3]) {
  return ((size_t)(int64_t)id[0] * 73856093u) ^ ((size_t)(int64_t)id[1] * 19349663u) ^
         ((size_t)(int64_t)id[2] * 83492791u);
}

static chunk_t* tab_find(chunk_t* tab, size_t cap, const int32_t id[3], int* found) {
  size_t h = chunk_hash(id) & (cap - 1);
  for (;;) {
    chunk_t* c = &tab[h];
    if (!c->used) { *found = 0; return c; }
    if (c->id[0] == id[0] && c->id[1] == id[1] && c->id[2] == id[2]) { *found = 1; return c; }
    h = (h + 1) & (cap - 1);
  }
}

static void tab_grow(oracle_chisel* o) {
  size_t ncap = o->cap * 2;
  chunk_t* nt = (chunk_t*)calloc(ncap, sizeof(chunk_t));
  for (size_t i = 0; i < o->cap; i++)
    if (o->tab[i].used) {
      int f;
      *tab_find(nt, ncap, o->tab[i].id, &f) = o->tab[i];
    }
  free(o->tab);
  o->tab = nt;
  o->cap = ncap;
}

/* Chunk::Chunk (Chunk.cpp:34-50) + DistVoxel()/ColorVoxel() defaults. */
static chunk_t* chunk_create(oracle_chisel* o, const int32_t id[3]) {
  if ((o->count + 1) * 2 > o->cap) tab_grow(o);
  int f;
  chunk_t* c = tab_find(o->tab, o->cap, id, &f);
  memcpy(c->id, id, sizeof(c->id));
  c->used = 1;
  c->sdf = (float*)malloc(CHUNK_VOX * sizeof(float));
  c->weight = (float*)calloc(CHUNK_VOX, sizeof(float));
  c->kfid = (uint32_t*)calloc(CHUNK_VOX, sizeof(uint32_t));
  c->rgbw = (uint32_t*)calloc(CHUNK_VOX, sizeof(uint32_t));
  for (int i = 0; i < CHUNK_VOX; i++) c->sdf[i] = 99999.0f;
  c->x_wu = c->x_w = NULL;
  if (o->track_exact) {
    c->x_wu = (double*)calloc(CHUNK_VOX, sizeof(double));
    c->x_w = (double*)calloc(CHUNK_VOX, sizeof(double));
  }
  o->count++;
  return c;
}

oracle_chisel* oracle_chisel_create(float resolution, float tq, float tl, float tc, float ts,
                                    float weight, int shard_rank, int shard_count) {
  oracle_chisel* o = (oracle_chisel*)calloc(1, sizeof(*o));
  o->resolution = resolution;
  o->tq = tq; o->tl = tl; o->tc = tc; o->ts = ts;
  o->weight = weight;
  o->half_voxel = resolution * 0.5f;
  o->rounding = 1.0f / ((float)16 * resolution);
  o->cap = 1024;
  o->tab = (chunk_t*)calloc(o->cap, sizeof(chunk_t));
  o->shard_rank = shard_rank;
  o->shard_count = shard_count;
  return o;
}

void oracle_chisel_clear(oracle_chisel* o) {
  for (size_t i = 0; i < o->cap; i++)
    if (o->tab[i].used) {
      free(o->tab[i].sdf); free(o->tab[i].weight); free(o->tab[i].kfid); free(o->tab[i].rgbw);
      free(o->tab[i].x_wu); free(o->tab[i].x_w);
    }
  memset(o->tab, 0, o->cap * sizeof(chunk_t));
  o->count = 0;
}

void oracle_chisel_destroy(oracle_chisel* o) {
  if (!o) return;
  oracle_chisel_clear(o);
  free(o->tab);
  free(o);
}

/* ---- small float helpers in Eigen's evaluation order ---------------------- */
static float sum3(float a, float b, float c) { return a + (b + c); } /* redux of 3: a0 + (a1 + a2) */
static float sqnorm3(const float v[3]) { return sum3(v[0] * v[0], v[1] * v[1], v[2] * v[2]); }

/* Affine3f * Vector3f  ->  t + R p, R p row-wise as a 3-term reduction. */
static void xform(const float R[9], const float t[3], const float p[3], float out[3]) {
  for (int i = 0; i < 3; i++)
    out[i] = t[i] + sum3(R[3 * i + 0] * p[0], R[3 * i + 1] * p[1], R[3 * i + 2] * p[2]);
}

static float cof(const float m[9], int i, int j) {
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m[3 * i1 + j1] * m[3 * i2 + j2] - m[3 * i1 + j2] * m[3 * i2 + j1];
}

/* Transform<float,3,Affine>::inverse(): cofactor inverse of the linear part,
 * translation = (-Rinv) * t  (Chisel.cpp:451 `cameraPose.inverse()`). */
void oracle_affine_inverse(const float R[9], const float t[3], float Ri[9], float ti[3]) {
  const float c00 = cof(R, 0, 0), c10 = cof(R, 1, 0), c20 = cof(R, 2, 0);
  const float det = sum3(c00 * R[0], c10 * R[3], c20 * R[6]);
  const float invdet = 1.0f / det;
  Ri[0] = c00 * invdet; Ri[1] = c10 * invdet; Ri[2] = c20 * invdet;
  Ri[3] = cof(R, 0, 1) * invdet; Ri[4] = cof(R, 1, 1) * invdet; Ri[5] = cof(R, 2, 1) * invdet;
  Ri[6] = cof(R, 0, 2) * invdet; Ri[7] = cof(R, 1, 2) * invdet; Ri[8] = cof(R, 2, 2) * invdet;
  for (int i = 0; i < 3; i++)
    ti[i] = sum3((-Ri[3 * i + 0]) * t[0], (-Ri[3 * i + 1]) * t[1], (-Ri[3 * i + 2]) * t[2]);
}

/* Raycast.cpp:12-15 mod(): unqualified fmod on floats -> the double overload. */
static float rc_mod(float value, float modulus) {
  return (float)fmod(fmod((double)value, (double)modulus) + (double)modulus, (double)modulus);
}
/* Raycast.cpp:17-30 */
static float rc_intbound(float s, float ds) {
  if (ds < 0) return rc_intbound(-s, -ds);
  s = rc_mod(s, 1);
  return (1 - s) / ds;
}
static int rc_signum(int x) { return (x > 0) ? 1 : ((x < 0) ? -1 : 0); }

typedef void (*visit_fn)(void* ctx, int x, int y, int z);

/* chisel::Raycast (Raycast.cpp:65-182); min/max are -/+INT_MAX at the call site
 * (Chisel.cpp:452-453). */
static void raycast(const float start[3], const float end[3], visit_fn fn, void* ctx) {
  int x = (int)floorf(start[0]), y = (int)floorf(start[1]), z = (int)floorf(start[2]);
  const int endX = (int)floorf(end[0]), endY = (int)floorf(end[1]), endZ = (int)floorf(end[2]);
  const float direction[3] = {end[0] - start[0], end[1] - start[1], end[2] - start[2]};
  const float maxDist = sqnorm3(direction);
  const float dx = (float)(endX - x), dy = (float)(endY - y), dz = (float)(endZ - z);
  const int stepX = rc_signum((int)dx), stepY = rc_signum((int)dy), stepZ = rc_signum((int)dz);
  float tMaxX = rc_intbound(start[0], dx), tMaxY = rc_intbound(start[1], dy),
        tMaxZ = rc_intbound(start[2], dz);
  const float tDeltaX = ((float)stepX) / dx, tDeltaY = ((float)stepY) / dy,
              tDeltaZ = ((float)stepZ) / dz;
  if (stepX == 0 && stepY == 0 && stepZ == 0) return;
  const int lo = -INT_MAX, hi = INT_MAX;
  for (int guard = 0; guard < RAY_STEP_GUARD; guard++) {
    if (x >= lo && x < hi && y >= lo && y < hi && z >= lo && z < hi) {
      fn(ctx, x, y, z);
      const float d[3] = {(float)x - start[0], (float)y - start[1], (float)z - start[2]};
      if (sqnorm3(d) > maxDist) return;
    }
    if (x == endX && y == endY && z == endZ) break;
    if (tMaxX < tMaxY) {
      if (tMaxX < tMaxZ) { x += stepX; tMaxX += tDeltaX; }
      else               { z += stepZ; tMaxZ += tDeltaZ; }
    } else {
      if (tMaxY < tMaxZ) { y += stepY; tMaxY += tDeltaY; }
      else               { z += stepZ; tMaxZ += tDeltaZ; }
    }
  }
}

typedef struct {
  oracle_chisel* o;
  const float* Ri; const float* ti; /* inverse pose */
  float depth, truncation;
  const float* wp; const float* dir; /* non-null: the normals flavour (world point, unit normal) */
  uint8_t r, g, b;
  uint32_t kfid;
  chunk_t* last; /* small lookup cache, no semantic effect */
  int64_t visits;
  int32_t newc, updc;
} visit_ctx;

/* DistVoxel::Integrate (DistVoxel.h:91-99) */
static void dist_integrate(float* sdf, float* w, float distUpdate, float weightUpdate) {
  const float oldSDF = *sdf, oldWeight = *w;
  *sdf = (oldWeight * oldSDF + weightUpdate * distUpdate) / (weightUpdate + oldWeight);
  *w = oldWeight + weightUpdate;
}

/* ColorVoxel::IntegrateSimple (ColorVoxel.h:91-110) on r | g << 8 | b << 16 | weight << 24 */
static uint32_t colour_integrate_simple(uint32_t p, uint8_t r, uint8_t g, uint8_t b, uint8_t wu) {
  uint8_t red = p & 255, green = (p >> 8) & 255, blue = (p >> 16) & 255, cw = p >> 24;
  if (cw >= 255 - wu) return p;
  const float inv = 1.f / (float)(wu + cw);
  red = (uint8_t)((float)(cw * red + wu * r) * inv);
  green = (uint8_t)((float)(cw * green + wu * g) * inv);
  blue = (uint8_t)((float)(cw * blue + wu * b) * inv);
  cw = (uint8_t)(cw + wu);
  return (uint32_t)red | ((uint32_t)green << 8) | ((uint32_t)blue << 16) | ((uint32_t)cw << 24);
}

/* ColorVoxel::Integrate (ColorVoxel.h:68-89): the flavour IntegrateWorldPointCloudWithNormals calls — a true
 * division and Saturate where IntegrateSimple multiplies by a rounded reciprocal. */
static uint32_t colour_integrate(uint32_t p, uint8_t r, uint8_t g, uint8_t b, uint8_t wu) {
  uint8_t red = p & 255, green = (p >> 8) & 255, blue = (p >> 16) & 255, cw = p >> 24;
  if (cw >= 255 - wu) return p;
  const float den = (float)(wu + cw);
  const uint8_t in[3] = {r, g, b};
  uint8_t* ch[3] = {&red, &green, &blue};
  for (int k = 0; k < 3; k++) {
    float v = ((float)cw * (float)*ch[k] + (float)(wu * in[k])) / den;
    v = v > 0.0f ? v : 0.0f;            /* Saturate: std::min(std::max(value, 0.0f), 255.0f) */
    v = v < 255.0f ? v : 255.0f;
    *ch[k] = (uint8_t)v;
  }
  cw = (uint8_t)(cw + wu);
  return (uint32_t)red | ((uint32_t)green << 8) | ((uint32_t)blue << 16) | ((uint32_t)cw << 24);
}

/* QuadraticTruncator::GetTruncationDistance (QuadraticTruncator.h:49), ConstantWeighter::GetWeight (ConstantWeighter.h:45) */
static float quadratic_truncation(float q, float l, float c, float s, float reading) {
  return (q * reading * reading + l * reading + c) * s;
}
static float constant_weight(float weight, float truncation) { return weight / (2.0f * truncation); }

static int owned(const oracle_chisel* o, const int32_t id[3]) {
  if (o->shard_count <= 1) return 1;
  return (int)(chunk_hash(id) % (size_t)o->shard_count) == o->shard_rank;
}

/* Body of the per-voxel loop, Chisel.cpp:503-549. */
static void visit(void* vctx, int vx, int vy, int vz) {
  visit_ctx* c = (visit_ctx*)vctx;
  oracle_chisel* o = c->o;
  /* ChunkManager::GetCentroid (ChunkManager.h:203-206) */
  const float center[3] = {(float)vx * o->resolution + o->half_voxel,
                           (float)vy * o->resolution + o->half_voxel,
                           (float)vz * o->resolution + o->half_voxel};
  /* GetIDAt (ChunkManager.h:192-201) */
  const int32_t id[3] = {(int32_t)floorf(center[0] * o->rounding),
                         (int32_t)floorf(center[1] * o->rounding),
                         (int32_t)floorf(center[2] * o->rounding)};
  if (o->visit_hook && !(o->hook_has_last && o->hook_last[0] == id[0] && o->hook_last[1] == id[1] && o->hook_last[2] == id[2])) {
    o->visit_hook(o->visit_hook_ctx, id);   /* (repeats of one id in a row are dropped: the hook only looks the id up) */
    memcpy(o->hook_last, id, sizeof(id));
    o->hook_has_last = 1;
  }
  if (!owned(o, id)) return; /* multi-GPU shard filter (not in the reference) */
  /* Chunk::GetLocalVoxelIDFromGlobal + IsCoordValid(VoxelID) (Chunk.cpp:96-105, Chunk.h:90-118) */
  const int lx = vx - id[0] * 16, ly = vy - id[1] * 16, lz = vz - id[2] * 16;
  const int vid = (lz * 16 + ly) * 16 + lx;
  if (!(vid >= 0 && vid < CHUNK_VOX)) return;

  float u;
  if (c->dir) {                                            /* IntegrateWorldPointCloudWithNormals, Chisel.cpp:329 */
    const float d[3] = {center[0] - c->wp[0], center[1] - c->wp[1], center[2] - c->wp[2]};
    u = d[0] * c->dir[0] + (d[1] * c->dir[1] + d[2] * c->dir[2]);   /* (center - worldPoint).dot(dir) */
  } else {
    float cc[3];
    xform(c->Ri, c->ti, center, cc);                       /* inversePose * center  :525 */
    const float length = sqrtf(sqnorm3(cc));               /* :526 */
    u = length * (c->depth / cc[2] - 1);                   /* :527 */
  }
  const float weight = constant_weight(o->weight, c->truncation);
  if (!(fabs((double)u) < (double)c->truncation)) return;  /* :531 */

  /* A chunk created by GetOrCreateChunkAt but never updated is garbage-collected
   * at the end of the call (:574-583), so creation can be deferred to the first
   * update without changing the resulting map. */
  chunk_t* ch = c->last;
  if (!ch || ch->id[0] != id[0] || ch->id[1] != id[1] || ch->id[2] != id[2]) {
    int found;
    ch = tab_find(o->tab, o->cap, id, &found);
    if (!found) { ch = chunk_create(o, id); c->newc++; }
    c->last = ch;
  }
  dist_integrate(&ch->sdf[vid], &ch->weight[vid], u, weight);
  if (ch->x_w) {   /* the exact mean of the same visits (products and sums in double) */
    ch->x_wu[vid] += (double)weight * (double)u;
    ch->x_w[vid] += (double)weight;
  }
  ch->kfid[vid] = c->kfid; /* SetKfid, USE_KFID_INTEGRATION 0 */
  ch->rgbw[vid] = c->dir ? colour_integrate(ch->rgbw[vid], c->r, c->g, c->b, 1)          /* :338 */
                         : colour_integrate_simple(ch->rgbw[vid], c->r, c->g, c->b, 1);  /* :536 */
  c->visits++;
}

/* Chisel::IntegratePointCloudWidthDepth, point-cloud part (Chisel.cpp:442-585).
 * Twc: 3x4 row-major [R|t]. */
void oracle_chisel_integrate(oracle_chisel* o, const float* xyz, const uint8_t* rgb,
                             const uint32_t* kfid, int n, const float* Twc) {
  float R[9], t[3], Ri[9], ti[3];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) R[3 * i + j] = Twc[4 * i + j];
    t[i] = Twc[4 * i + 3];
  }
  oracle_affine_inverse(R, t, Ri, ti);
  const float resolution = o->resolution;
  const float roundToVoxel = 1.0f / resolution;                             /* :444 */
  const float diag = (float)(2.0 * sqrt((double)3.0f) * (double)resolution); /* :447 */
  const float byteToFloat = 1.0f / 255.0f;                                  /* Conversions.h:107 */
  visit_ctx c;
  memset(&c, 0, sizeof(c));
  o->hook_has_last = 0;
  c.o = o; c.Ri = Ri; c.ti = ti;
  const size_t before = o->count;
  /* updated-chunk count: chunks whose any voxel passed; tracked through a per-call mark */
  for (int i = 0; i < n; i++) {
    const float* point = xyz + 3 * (size_t)i;
    float worldPoint[3];
    xform(R, t, point, worldPoint);                                         /* :472 */
    const float depth = point[2];
    if (depth < 0.01f) continue;                                            /* :475 */
    float v[3] = {worldPoint[0] - t[0], worldPoint[1] - t[1], worldPoint[2] - t[2]};
    const float z2 = sqnorm3(v);
    float dir[3] = {v[0], v[1], v[2]};
    if (z2 > 0.0f) {                                                        /* normalized() */
      const float nrm = sqrtf(z2);
      dir[0] = v[0] / nrm; dir[1] = v[1] / nrm; dir[2] = v[2] / nrm;
    }
    const float trunc_q = quadratic_truncation(o->tq, o->tl, o->tc, o->ts, depth);
    const float truncation = trunc_q > diag ? trunc_q : diag;               /* std::max, :479 */
    float start[3], end[3];
    for (int k = 0; k < 3; k++) {
      const float swp = worldPoint[k] * roundToVoxel;                       /* :482 */
      const float sdt = dir[k] * truncation * roundToVoxel;                 /* :483 */
      start[k] = swp - sdt;                                                 /* :486 */
      end[k] = swp + sdt;                                                   /* :488 */
    }
    c.depth = depth; c.truncation = truncation;
    /* Conversions.h:118-121 then Chisel.cpp:536 */
    c.r = (uint8_t)(((float)rgb[3 * (size_t)i + 0] * byteToFloat) * 255.0f);
    c.g = (uint8_t)(((float)rgb[3 * (size_t)i + 1] * byteToFloat) * 255.0f);
    c.b = (uint8_t)(((float)rgb[3 * (size_t)i + 2] * byteToFloat) * 255.0f);
    c.kfid = kfid ? kfid[i] : 0;
    raycast(start, end, visit, &c);
  }
  o->last_visits = c.visits;
  o->last_new = (int32_t)(o->count - before);
}

/* Chisel::IntegrateWorldPointCloudWithNormals (Chisel.cpp:238-376), what PointCloudMapChisel::LoadMap feeds the
 * saved map cloud through (src/PointCloudMapChisel.cc:527-546 -> ChiselServer::IntegrateWorldPointCloud,
 * ChiselServer.cpp:587-615; Twc = identity there): every point casts the segment point -/+ 4 voxels along its
 * NORMAL, u = (centre - point) . normal, constant truncation 4 * resolution, no depth test, no carving,
 * ColorVoxel::Integrate.  normals: n x 3.  Pinned against the compiled Chisel.cpp
 * (tests/test_oracle_pinned_chisel_map.py). */
void oracle_chisel_integrate_world_normals(oracle_chisel* o, const float* xyz, const uint8_t* rgb, const uint32_t* kfid,
                                           const float* normals, int n, const float* Twc) {
  float R[9], t[3];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) R[3 * i + j] = Twc[4 * i + j];
    t[i] = Twc[4 * i + 3];
  }
  const float resolution = o->resolution;
  const float roundToVoxel = 1.0f / resolution;                             /* :244 */
  const float truncation = 4 * resolution;                                  /* :266 */
  const float byteToFloat = 1.0f / 255.0f;                                  /* Conversions.h:158 */
  visit_ctx c;
  memset(&c, 0, sizeof(c));
  o->hook_has_last = 0;
  c.o = o;
  const size_t before = o->count;
  for (int i = 0; i < n; i++) {
    float worldPoint[3];
    xform(R, t, xyz + 3 * (size_t)i, worldPoint);                           /* :276 */
    const float* nrm = normals + 3 * (size_t)i;
    const float z2 = sqnorm3(nrm);
    float dir[3] = {nrm[0], nrm[1], nrm[2]};
    if (z2 > 0.0f) {                                                        /* normalized(), :282 */
      const float l = sqrtf(z2);
      dir[0] = nrm[0] / l; dir[1] = nrm[1] / l; dir[2] = nrm[2] / l;
    }
    float start[3], end[3];
    for (int k = 0; k < 3; k++) {
      const float swp = worldPoint[k] * roundToVoxel;                       /* :288 */
      const float sdt = dir[k] * truncation * roundToVoxel;                 /* :289 */
      start[k] = swp - sdt;                                                 /* :291 */
      end[k] = swp + sdt;                                                   /* :293 */
    }
    c.truncation = truncation;
    c.wp = worldPoint; c.dir = dir;
    c.r = (uint8_t)(((float)rgb[3 * (size_t)i + 0] * byteToFloat) * 255.0f);   /* Conversions.h:177-179, Chisel.cpp:338 */
    c.g = (uint8_t)(((float)rgb[3 * (size_t)i + 1] * byteToFloat) * 255.0f);
    c.b = (uint8_t)(((float)rgb[3 * (size_t)i + 2] * byteToFloat) * 255.0f);
    c.kfid = kfid ? kfid[i] : 0;
    raycast(start, end, visit, &c);
  }
  o->last_visits = c.visits;
  o->last_new = (int32_t)(o->count - before);
}

int64_t oracle_chisel_last_visits(const oracle_chisel* o) { return o->last_visits; }
int oracle_chisel_last_new_chunks(const oracle_chisel* o) { return o->last_new; }
int oracle_chisel_num_chunks(const oracle_chisel* o) { return (int)o->count; }

void oracle_chisel_chunk_ids(const oracle_chisel* o, int32_t* ids) {
  size_t k = 0;
  for (size_t i = 0; i < o->cap; i++)
    if (o->tab[i].used) { memcpy(ids + 3 * k, o->tab[i].id, 3 * sizeof(int32_t)); k++; }
}

/* Turn the exact (double) accumulators on: for maps that start empty (chunks created afterwards carry them). */
void oracle_chisel_track_exact(oracle_chisel* o, int on) { o->track_exact = on; }

/* sdf = sum(w_u u) / sum(w_u) and W = sum(w_u) of every voxel of the chunk in double; 0 where the voxel is unknown.
 * Returns 0 if the chunk does not exist or carries no exact accumulators. */
int oracle_chisel_get_chunk_exact(const oracle_chisel* o, int cx, int cy, int cz, double* sdf, double* weight) {
  const int32_t id[3] = {cx, cy, cz};
  int found;
  chunk_t* c = tab_find(o->tab, o->cap, id, &found);
  if (!found || !c->x_w) return 0;
  for (int i = 0; i < CHUNK_VOX; i++) {
    weight[i] = c->x_w[i];
    sdf[i] = c->x_w[i] > 0 ? c->x_wu[i] / c->x_w[i] : 0.0;
  }
  return 1;
}

int oracle_chisel_get_chunk(const oracle_chisel* o, int cx, int cy, int cz, float* sdf,
                            float* weight, uint32_t* kfid, uint32_t* rgbw) {
  const int32_t id[3] = {cx, cy, cz};
  int found;
  chunk_t* c = tab_find(o->tab, o->cap, id, &found);
  if (!found) return 0;
  memcpy(sdf, c->sdf, CHUNK_VOX * sizeof(float));
  memcpy(weight, c->weight, CHUNK_VOX * sizeof(float));
  memcpy(kfid, c->kfid, CHUNK_VOX * sizeof(uint32_t));
  memcpy(rgbw, c->rgbw, CHUNK_VOX * sizeof(uint32_t));
  return 1;
}

/* ------------------------------------------------------------------ carving
 * Chisel::IntegratePointCloudWidthDepth, the part before the point cloud (Chisel.cpp:394-438):
 * PinholeCamera::SetupFrustum (PinholeCamera.cpp:55-59) -> Frustum::SetFromParams /
 * SetFromVectors (Frustum.cpp:150-196) -> ChunkManager::GetChunkIDsIntersecting(frustum)
 * (ChunkManager.cpp:241-271, Frustum::Intersects Frustum.cpp:40-78, Plane(a,b,c) Plane.cpp:46-54)
 * -> ProjectionIntegrator::CarveWithDepth (ProjectionIntegrator.h:271-338) for every
 * existing chunk of the list.  Restated literally, including what looks unintended in the
 * reference: SetupFrustum passes fy for fx, Plane keeps the distance of the UNnormalised
 * normal, and Intersects accepts a box as soon as ONE plane has its far vertex in front.
 * atan2 / tan resolve to the double overloads (unqualified calls on floats without
 * <math.h>'s C++ wrappers); Eigen reductions of three terms are a0 + (a1 + a2). */
typedef struct { float n[3]; float d; } plane_t;

static void vsub(const float a[3], const float b[3], float o[3]) { for (int i = 0; i < 3; i++) o[i] = a[i] - b[i]; }
static float vdot(const float a[3], const float b[3]) { return sum3(a[0] * b[0], a[1] * b[1], a[2] * b[2]); }

static plane_t plane_from(const float a[3], const float b[3], const float c[3]) {
  float ab[3], ac[3], cr[3];
  vsub(b, a, ab);
  vsub(c, a, ac);
  cr[0] = ab[1] * ac[2] - ab[2] * ac[1];
  cr[1] = ab[2] * ac[0] - ab[0] * ac[2];
  cr[2] = ab[0] * ac[1] - ab[1] * ac[0];
  plane_t p;
  const float z = sqnorm3(cr);
  for (int i = 0; i < 3; i++) p.n[i] = (z > 0.0f) ? cr[i] / sqrtf(z) : cr[i];   /* Eigen normalized() */
  p.d = -vdot(cr, a);
  return p;
}

/* planes[6] = far, near, top, bottom, left, right (the order Intersects walks them);
 * lo / hi = ComputeBoundingBox of the 8 corners. */
static void chisel_frustum(const float R[9], const float t[3], float near_d, float far_d, float fy, float cy,
                           float width, float height, plane_t planes[6], float lo[3], float hi[3]) {
  float right[3], up[3], fwd[3];
  for (int i = 0; i < 3; i++) { right[i] = R[3 * i + 0]; up[i] = -R[3 * i + 1]; fwd[i] = R[3 * i + 2]; }
  const float fx = fy;                                            /* SetupFrustum hands fy twice */
  const float aspect = (fx * width) / (fy * height);
  const float fov = (float)(atan2((double)cy, (double)fy) + atan2((double)(height - cy), (double)fy));
  const float tang = (float)tan((double)(fov / 2));
  const float hf = tang * far_d, wf = hf * aspect, hn = tang * near_d, wn = hn * aspect;
  float fc[3], nc[3], c[8][3];
  for (int i = 0; i < 3; i++) { fc[i] = t[i] + fwd[i] * far_d; nc[i] = t[i] + fwd[i] * near_d; }
  float *ftl = c[0], *ftr = c[1], *fbl = c[2], *fbr = c[3], *nbr = c[4], *ntl = c[5], *ntr = c[6], *nbl = c[7];
  for (int i = 0; i < 3; i++) {
    ftl[i] = fc[i] + (up[i] * hf) - (right[i] * wf);
    ftr[i] = fc[i] + (up[i] * hf) + (right[i] * wf);
    fbl[i] = fc[i] - (up[i] * hf) - (right[i] * wf);
    fbr[i] = fc[i] - (up[i] * hf) + (right[i] * wf);
    ntl[i] = nc[i] + (up[i] * hn) - (right[i] * wn);
    ntr[i] = nc[i] + (up[i] * hn) + (right[i] * wn);
    nbl[i] = nc[i] - (up[i] * hn) - (right[i] * wn);
    nbr[i] = nc[i] - (up[i] * hn) + (right[i] * wn);
  }
  planes[0] = plane_from(ftr, ftl, fbr);   /* far */
  planes[1] = plane_from(nbl, ntl, nbr);   /* near */
  planes[2] = plane_from(ntl, ftl, ntr);   /* top */
  planes[3] = plane_from(nbr, fbl, nbl);   /* bottom */
  planes[4] = plane_from(ftl, ntl, fbl);   /* left */
  planes[5] = plane_from(ntr, ftr, nbr);   /* right */
  for (int i = 0; i < 3; i++) { lo[i] = 3.402823466e+38f; hi[i] = -3.402823466e+38f; }
  for (int k = 0; k < 8; k++)
    for (int i = 0; i < 3; i++) {
      lo[i] = (c[k][i] < lo[i]) ? c[k][i] : lo[i];   /* std::min<float>(tempMin, corner) */
      hi[i] = (hi[i] < c[k][i]) ? c[k][i] : hi[i];   /* std::max<float>(tempMax, corner) */
    }
}

static int frustum_intersects(const plane_t planes[6], const float bmin[3], const float bmax[3]) {
  for (int p = 0; p < 6; p++) {
    float v[3];
    for (int i = 0; i < 3; i++) v[i] = (planes[p].n[i] < 0.0f) ? bmin[i] : bmax[i];
    if (vdot(v, planes[p].n) + planes[p].d > 0.0f) return 1;
  }
  return 0;
}

/* depth: height rows of width floats (NaN = no measurement).  Returns the number of carved chunks
 * ("carved in N chunks"); ids of those chunks to carved_ids (3 ints each) if not NULL. */
int oracle_chisel_carve(oracle_chisel* o, const float* depth, int width, int height, float fx, float fy, float cx,
                        float cy, float near_d, float far_d, const float* Twc, float carving_dist,
                        int32_t* carved_ids) {
  float R[9], t[3];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) R[3 * i + j] = Twc[4 * i + j];
    t[i] = Twc[4 * i + 3];
  }
  plane_t planes[6];
  float lo[3], hi[3];
  chisel_frustum(R, t, near_d, far_d, fy, cy, (float)width, (float)height, planes, lo, hi);
  const float res = o->resolution;
  const float diag = (float)(2.0 * sqrt((double)3.0f) * (double)res);
  int32_t min_id[3], max_id[3];
  for (int i = 0; i < 3; i++) {                     /* GetIDAt, ChunkManager.h:192-198 */
    min_id[i] = (int32_t)floorf(lo[i] * o->rounding);
    max_id[i] = (int32_t)floorf(hi[i] * o->rounding) + 1;
  }
  int carved = 0;
  for (int x = min_id[0] - 1; x <= max_id[0] + 1; x++)
    for (int y = min_id[1] - 1; y <= max_id[1] + 1; y++)
      for (int z = min_id[2] - 1; z <= max_id[2] + 1; z++) {
        const float bmin[3] = {(float)(x * 16) * res, (float)(y * 16) * res, (float)(z * 16) * res};
        const float bmax[3] = {bmin[0] + 16.0f * res, bmin[1] + 16.0f * res, bmin[2] + 16.0f * res};
        if (!frustum_intersects(planes, bmin, bmax)) continue;
        const int32_t id[3] = {x, y, z};
        int found = 0;
        chunk_t* ch = tab_find(o->tab, o->cap, id, &found);
        if (!found) continue;
        /* ---- CarveWithDepth */
        const float origin[3] = {(float)(16 * x) * res, (float)(16 * y) * res, (float)(16 * z) * res};   /* Chunk.cpp:48 */
        int updated = 0;
        for (int i = 0; i < CHUNK_VOX; i++) {
          if ((double)ch->weight[i] <= 1e-15) continue;
          const int lx = i & 15, ly = (i >> 4) & 15, lz = i >> 8;
          const float cen[3] = {((float)lx * res + o->half_voxel) + origin[0],        /* centroids[i] + origin */
                                ((float)ly * res + o->half_voxel) + origin[1],
                                ((float)lz * res + o->half_voxel) + origin[2]};
          float dv[3], pc[3];
          vsub(cen, t, dv);
          for (int r = 0; r < 3; r++) pc[r] = sum3(R[0 + r] * dv[0], R[3 + r] * dv[1], R[6 + r] * dv[2]);   /* Rcw = R^T */
          const float inv_z = 1.0f / pc[2];
          const float u = fx * pc[0] * inv_z + cx, v = fy * pc[1] * inv_z + cy;
          if (pc[2] < 0 || !(u >= 0 && v >= 0 && u < (float)width && v < (float)height)) continue;
          const float d = depth[(size_t)(int)v * (size_t)width + (size_t)(int)u];
          if (isnan(d)) continue;
          const float q = (o->tq * d * d + o->tl * d + o->tc) * o->ts;
          const float trunc = q > diag ? q : diag;      /* std::max(truncator(depth), diag) */
          const float surface = d - pc[2];
          if (surface > trunc + carving_dist) {
            if ((double)ch->sdf[i] < 1e-5) {
              ch->sdf[i] = 99999.0f;                    /* DistVoxel::Reset */
              ch->weight[i] = 0.0f;
              ch->kfid[i] = 0;
              updated = 1;
            }
          }
        }
        if (updated) {
          if (carved_ids) { carved_ids[3 * carved] = x; carved_ids[3 * carved + 1] = y; carved_ids[3 * carved + 2] = z; }
          carved++;
        }
      }
  return carved;
}

/* ===========================================================================================
 * Surface extraction of one chunk — ChunkManager::RecomputeMesh (src/ChunkManager.cpp:116-170):
 *   GenerateMesh (:577-660) with USE_KFID_MESHING 1 / USE_KFID_VERTICES 0 (:39-40):
 *     ExtractInsideVoxelMeshKfid (:438-475), ExtractBorderVoxelMeshKfid (:477-575),
 *     MarchingCubes::MeshCube(coords, sdf, kfid, ...) + InterpolateEdgeVertices / InterpolateVertex
 *     (include/open_chisel/marching_cubes/MarchingCubes.h:110-143, 206-245), triangleTable /
 *     edgeIndexPairs (src/marching_cubes/MarchingCubes.cpp:32-305);
 *   ColorizeMesh (:860-872) -> InterpolateColor (:718-805), Chunk::GetColorAt (Chunk.cpp:137-155);
 *   ComputeNormalsFromGradients (:840-858) -> GetSDFAndGradient (:663-690), GetSDF (:692-716).
 * Called from Chisel::UpdateMeshes (Chisel.cpp:57-65) <- ChiselServer::UpdateMesh <-
 * PointCloudMapChisel::UpdateMap (src/PointCloudMapChisel.cc:228-246).
 * Quirks kept: InterpolateVertex returns v1 + 0.5 v2 on a flat edge; InterpolateColor passes voxel
 * INDICES to a function that expects metres; GetSDF accepts any linear id in [0, 4096).
 * Pinned: ChunkManager.cpp compiled into oracle/_ref/libchisel_full_ref.so gives the same meshes bit for bit
 * (tests/test_oracle_pinned_chisel_map.py). */
static const int kTriangleTable[256 * 16] = {
#include "mc_table.inc"
};
static const int kEdgeIndexPairs[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6},
                                           {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
static const int kCubeIndexOffsets[3][8] = {{0, 1, 1, 0, 0, 1, 1, 0},   /* ChunkManager.cpp:84-86 */
                                            {0, 0, 1, 1, 0, 0, 1, 1},
                                            {0, 0, 0, 0, 1, 1, 1, 1}};

typedef struct {
  float* vertices; float* normals; uint32_t* kfids;
  int n, cap;
} mesh_out;

static const chunk_t* mesh_find(const oracle_chisel* o, int x, int y, int z) {
  const int32_t id[3] = {x, y, z};
  int found = 0;
  chunk_t* c = tab_find(o->tab, o->cap, id, &found);
  return found ? c : NULL;
}
static const chunk_t* mesh_chunk_at(const oracle_chisel* o, const float pos[3]) {   /* GetChunkAt, ChunkManager.h:141 */
  return mesh_find(o, (int)floorf(pos[0] * o->rounding), (int)floorf(pos[1] * o->rounding),
                   (int)floorf(pos[2] * o->rounding));
}
static void mesh_origin(const oracle_chisel* o, const chunk_t* c, float org[3]) {    /* Chunk.cpp:48 */
  for (int i = 0; i < 3; i++) org[i] = (float)(16 * c->id[i]) * o->resolution;
}

/* MarchingCubes::MeshCube(vertexCoords, vertexSDF, const uint32_t& vertexKfid, ...) */
static void mesh_cube(const float cc[8][3], const float sdf[8], uint32_t kfid, mesh_out* m) {
  int index = 0;
  for (int i = 0; i < 8; i++) index |= (sdf[i] < 0) ? (1 << i) : 0;
  if (index == 0) return;
  float edge[12][3];
  memset(edge, 0, sizeof edge);
  for (int i = 0; i < 12; i++) {
    const int e0 = kEdgeIndexPairs[i][0], e1 = kEdgeIndexPairs[i][1];
    if ((sdf[e0] < 0 && sdf[e1] >= 0) || (sdf[e0] >= 0 && sdf[e1] < 0)) {
      const float minDiff = 1e-6f;
      const float sdfDiff = sdf[e0] - sdf[e1];
      if (fabsf(sdfDiff) < minDiff) {
        for (int k = 0; k < 3; k++) edge[i][k] = cc[e0][k] + 0.5f * cc[e1][k];
      } else {
        const float t = sdf[e0] / sdfDiff;
        for (int k = 0; k < 3; k++) edge[i][k] = cc[e0][k] + t * (cc[e1][k] - cc[e0][k]);
      }
    }
  }
  const int* row = kTriangleTable + 16 * index;
  for (int col = 0; row[col] != -1; col += 3) {
    if (m->n + 3 > m->cap) { m->n += 3; continue; }   /* counted, not stored */
    float* p0 = m->vertices + 3 * (size_t)m->n;
    float *p1 = p0 + 3, *p2 = p0 + 6;
    for (int k = 0; k < 3; k++) {
      p0[k] = edge[row[col + 2]][k];
      p1[k] = edge[row[col + 1]][k];
      p2[k] = edge[row[col]][k];
    }
    float px[3], py[3], n[3];
    vsub(p1, p0, px);
    vsub(p2, p0, py);
    n[0] = px[1] * py[2] - px[2] * py[1];
    n[1] = px[2] * py[0] - px[0] * py[2];
    n[2] = px[0] * py[1] - px[1] * py[0];
    const float z = sqnorm3(n);
    if (z > 0.0f) { const float s = sqrtf(z); n[0] /= s; n[1] /= s; n[2] /= s; }   /* normalized() */
    for (int v = 0; v < 3; v++) {
      for (int k = 0; k < 3; k++) m->normals[3 * (size_t)(m->n + v) + k] = n[k];
      m->kfids[m->n + v] = kfid;
    }
    m->n += 3;
  }
}

/* ExtractInsideVoxelMeshKfid / ExtractBorderVoxelMeshKfid: the border variant with every corner
 * inside the chunk reduces to the inside variant, so one function serves both loops. */
static void mesh_voxel(const oracle_chisel* o, const chunk_t* ch, int x, int y, int z, mesh_out* m) {
  float org[3];
  mesh_origin(o, ch, org);
  const float res = o->resolution;
  /* centroids[i] + chunk->GetOrigin(), ChunkManager.cpp:78, :601 */
  const float coords[3] = {((float)x * res + o->half_voxel) + org[0], ((float)y * res + o->half_voxel) + org[1],
                           ((float)z * res + o->half_voxel) + org[2]};
  float cc[8][3], sdf[8];
  uint32_t kfid = 0;
  for (int i = 0; i < 8; i++) {
    int c[3] = {x + kCubeIndexOffsets[0][i], y + kCubeIndexOffsets[1][i], z + kCubeIndexOffsets[2][i]};
    const chunk_t* src = ch;
    if (c[0] > 15 || c[1] > 15 || c[2] > 15) {                   /* :492-527 (offsets are never negative) */
      int off[3] = {0, 0, 0};
      for (int j = 0; j < 3; j++)
        if (c[j] >= 16) { off[j] = 1; c[j] = 0; }
      src = mesh_find(o, ch->id[0] + off[0], ch->id[1] + off[1], ch->id[2] + off[2]);
      if (!src) return;                                           /* allNeighborsObserved = false */
    }
    const int id = (c[2] * 16 + c[1]) * 16 + c[0];
    if (src->weight[id] <= 1e-15) return;                         /* double comparison, :462, :499, :543 */
    for (int k = 0; k < 3; k++) cc[i][k] = coords[k] + (float)kCubeIndexOffsets[k][i] * res;
    sdf[i] = src->sdf[id];
    if (i == 0) kfid = src->kfid[id];
  }
  mesh_cube(cc, sdf, kfid, m);
}

static const uint32_t* mesh_color_voxel(const oracle_chisel* o, float px, float py, float pz) {   /* GetColorVoxel, :818 */
  const float pos[3] = {px, py, pz};
  const chunk_t* c = mesh_chunk_at(o, pos);
  if (!c) return NULL;
  float org[3];
  mesh_origin(o, c, org);
  const float inv = 1.0f / o->resolution;                         /* Chunk.cpp:38 */
  const int vx = (int)floorf((pos[0] - org[0]) * inv), vy = (int)floorf((pos[1] - org[1]) * inv),
            vz = (int)floorf((pos[2] - org[2]) * inv);
  const int id = (vz * 16 + vy) * 16 + vx;
  return (id >= 0 && id < CHUNK_VOX) ? &c->rgbw[id] : NULL;
}

static void mesh_interpolate_color(const oracle_chisel* o, const float p[3], float out[3]) {   /* :718-805 */
  const float inv = 1.f / o->resolution;                          /* ChunkManager.cpp:67 */
  const float x = p[0], y = p[1], z = p[2];
  const int x_0 = (int)floorf(x * inv), y_0 = (int)floorf(y * inv), z_0 = (int)floorf(z * inv);
  const int x_1 = x_0 + 1, y_1 = y_0 + 1, z_1 = z_0 + 1;
  const uint32_t* v_000 = mesh_color_voxel(o, (float)x_0, (float)y_0, (float)z_0);
  const uint32_t* v_001 = mesh_color_voxel(o, (float)x_0, (float)y_0, (float)z_1);
  const uint32_t* v_011 = mesh_color_voxel(o, (float)x_0, (float)y_1, (float)z_1);
  const uint32_t* v_111 = mesh_color_voxel(o, (float)x_1, (float)y_1, (float)z_1);
  const uint32_t* v_110 = mesh_color_voxel(o, (float)x_1, (float)y_1, (float)z_0);
  const uint32_t* v_100 = mesh_color_voxel(o, (float)x_1, (float)y_0, (float)z_0);
  const uint32_t* v_010 = mesh_color_voxel(o, (float)x_0, (float)y_1, (float)z_0);
  const uint32_t* v_101 = mesh_color_voxel(o, (float)x_1, (float)y_0, (float)z_1);
  if (!v_000 || !v_001 || !v_011 || !v_111 || !v_110 || !v_100 || !v_010 || !v_101) {
    out[0] = out[1] = out[2] = 0.0f;
    const chunk_t* c = mesh_chunk_at(o, p);
    if (!c) return;
    float org[3];
    mesh_origin(o, c, org);                                        /* Chunk::GetColorAt, Chunk.cpp:137-155 */
    const float size = 16.0f * o->resolution;
    for (int k = 0; k < 3; k++)
      if (!(p[k] >= org[k] && p[k] <= org[k] + size)) return;
    const float cinv = 1.0f / o->resolution;
    const int cx = (int)((p[0] - org[0]) * cinv), cy = (int)((p[1] - org[1]) * cinv), cz = (int)((p[2] - org[2]) * cinv);
    if (cx < 0 || cx >= 16 || cy < 0 || cy >= 16 || cz < 0 || cz >= 16) return;
    const uint32_t col = c->rgbw[(cz * 16 + cy) * 16 + cx];
    const float invMaxVal = 1.f / 255.0f;
    out[0] = (float)(col & 255u) * invMaxVal;
    out[1] = (float)((col >> 8) & 255u) * invMaxVal;
    out[2] = (float)((col >> 16) & 255u) * invMaxVal;
    return;
  }
  const float xd = (x - (float)x_0) / (float)(x_1 - x_0);
  const float yd = (y - (float)y_0) / (float)(y_1 - y_0);
  const float zd = (z - (float)z_0) / (float)(z_1 - z_0);
  for (int ch = 0; ch < 3; ch++) {
    const int sh = 8 * ch;
#define CH(v) ((float)(((*(v)) >> sh) & 255u))
    const float c_00 = CH(v_000) * (1 - xd) + CH(v_100) * xd;
    const float c_10 = CH(v_010) * (1 - xd) + CH(v_110) * xd;
    const float c_01 = CH(v_001) * (1 - xd) + CH(v_101) * xd;
    const float c_11 = CH(v_011) * (1 - xd) + CH(v_111) * xd;
#undef CH
    const float c_0 = c_00 * (1 - yd) + c_10 * yd;
    const float c_1 = c_01 * (1 - yd) + c_11 * yd;
    const float c = c_0 * (1 - zd) + c_1 * zd;
    out[ch] = c / 255.0f;
  }
}

static int mesh_get_sdf(const oracle_chisel* o, const float posf[3], double* dist) {   /* GetSDF, :692-716 */
  const chunk_t* c = mesh_chunk_at(o, posf);
  if (!c) return 0;
  float org[3];
  mesh_origin(o, c, org);
  const float inv = 1.0f / o->resolution;
  const int vx = (int)floorf((posf[0] - org[0]) * inv), vy = (int)floorf((posf[1] - org[1]) * inv),
            vz = (int)floorf((posf[2] - org[2]) * inv);
  const int id = (vz * 16 + vy) * 16 + vx;
  if (id >= 0 && id < CHUNK_VOX && c->weight[id] > 1e-12) {
    *dist = c->sdf[id];
    return 1;
  }
  return 0;
}

static void mesh_gradient_normal(const oracle_chisel* o, const float pos[3], float* normal) {   /* :840-858 */
  const float res = o->resolution, inv = 1.f / res, half = 0.5f * res;   /* ChunkManager.cpp:67, :89 */
  const float posf[3] = {floorf(pos[0] * inv) * res + half, floorf(pos[1] * inv) * res + half,
                         floorf(pos[2] * inv) * res + half};
  double dist, dp[3], dm[3];
  if (!mesh_get_sdf(o, posf, &dist)) return;
  for (int k = 0; k < 3; k++) {                                    /* +x, +y, +z */
    float q[3] = {posf[0], posf[1], posf[2]};
    q[k] = posf[k] + res;
    if (!mesh_get_sdf(o, q, &dp[k])) return;
  }
  for (int k = 0; k < 3; k++) {                                    /* -x, -y, -z */
    float q[3] = {posf[0], posf[1], posf[2]};
    q[k] = posf[k] - res;
    if (!mesh_get_sdf(o, q, &dm[k])) return;
  }
  float g[3] = {(float)(dp[0] - dm[0]), (float)(dp[1] - dm[1]), (float)(dp[2] - dm[2])};
  const float z = sqnorm3(g);
  if (z > 0.0f) { const float s = sqrtf(z); g[0] /= s; g[1] /= s; g[2] /= s; }   /* grad->normalize() */
  const float mag = sqrtf(sqnorm3(g));
  if (mag > 1e-12) {
    const float r = 1.0f / mag;
    normal[0] = g[0] * r; normal[1] = g[1] * r; normal[2] = g[2] * r;
  }
}

/* Returns the number of vertices of chunk (cx, cy, cz) (0 if the chunk does not exist), writing up to
 * `cap` of them: vertices / normals / colors n x 3 f32, kfids n u32 — chisel::Mesh after RecomputeMesh. */
int oracle_chisel_mesh_chunk(const oracle_chisel* o, int cx, int cy, int cz, float* vertices, float* normals,
                             float* colors, uint32_t* kfids, int cap) {
  const chunk_t* ch = mesh_find(o, cx, cy, cz);
  if (!ch) return 0;
  mesh_out m = {vertices, normals, kfids, 0, cap};
  for (int z = 0; z < 15; z++)                                     /* :590-606 */
    for (int y = 0; y < 15; y++)
      for (int x = 0; x < 15; x++) mesh_voxel(o, ch, x, y, z, &m);
  for (int z = 0; z < 15; z++)                                     /* max X plane, :611-625 */
    for (int y = 0; y < 16; y++) mesh_voxel(o, ch, 15, y, z, &m);
  for (int z = 0; z < 15; z++)                                     /* max Y plane, :628-642 */
    for (int x = 0; x < 15; x++) mesh_voxel(o, ch, x, 15, z, &m);
  for (int y = 0; y < 16; y++)                                     /* max Z plane, :645-659 */
    for (int x = 0; x < 16; x++) mesh_voxel(o, ch, x, y, 15, &m);
  const int n = m.n < cap ? m.n : cap;
  for (int i = 0; i < n; i++) mesh_interpolate_color(o, vertices + 3 * (size_t)i, colors + 3 * (size_t)i);   /* :158-161 */
  for (int i = 0; i < n; i++) mesh_gradient_normal(o, vertices + 3 * (size_t)i, normals + 3 * (size_t)i);    /* :163 */
  return m.n;
}

/* Test hook (no counterpart in the reference): overwrite / create one chunk with given voxel payloads, so
 * that the meshing restatement can be checked on analytic distance fields. */
void oracle_chisel_set_chunk(oracle_chisel* o, int cx, int cy, int cz, const float* sdf, const float* weight,
                             const uint32_t* kfid, const uint32_t* rgbw) {
  const int32_t id[3] = {cx, cy, cz};
  int found = 0;
  chunk_t* c = tab_find(o->tab, o->cap, id, &found);
  if (!found) c = chunk_create(o, id);
  memcpy(c->sdf, sdf, CHUNK_VOX * sizeof(float));
  memcpy(c->weight, weight, CHUNK_VOX * sizeof(float));
  memcpy(c->kfid, kfid, CHUNK_VOX * sizeof(uint32_t));
  memcpy(c->rgbw, rgbw, CHUNK_VOX * sizeof(uint32_t));
}

/* ===========================================================================================
 * ChunkManager::Deform, the chunk part (src/ChunkManager.cpp:918-1017), behind Chisel::Deform (Chisel.cpp:588-591) <-
 * ChiselServer::Deform (ChiselServer.cpp:617-621) <- PointCloudMapChisel::OnMapChange (src/PointCloudMapChisel.cc):
 * every known voxel (weight > 1e-15) whose kfid has an entry in the deformation map moves to
 * newPos = R * centre + t; the first voxel to land in a new voxel is copied, later ones are merged with
 * DistVoxel::Integrate(sdf, weight), SetKfid and ColorVoxel::Integrate(r, g, b, 1).  Old chunks are walked in the
 * iteration order of the reference's std::unordered_map `chunks`, which the CALLER supplies as `order` (n_order ids;
 * oracle/tsdf_chisel_deform.cpp keeps the real container); voxels of a chunk in id order.  new_order receives the ids
 * of the new chunks in the order the reference inserts them into `newChunks` (first claim).
 * The new chunk comes from floor(newPos * 1/(16 res)) and the voxel from floor(newPos * 1/res) - 16 * chunk: two
 * roundings that can disagree by one voxel at a chunk face.  The reference then indexes with the linear id
 * (z * 16 + y) * 16 + x of the out-of-range local coordinates: inside [0, 4096) that is a (wrong but well-defined)
 * voxel and is reproduced; outside it is undefined behaviour in the reference — skipped here and counted in
 * stats[1].  stats[0] = voxels discarded because their kfid has no transformation ("num discarded voxels").
 * Rt: n_map x 12 floats (R row-major, then t); kfids strictly increasing.  Pinned against the compiled
 * ChunkManager.cpp (tests/test_oracle_pinned_chisel_map.py). */
int oracle_chisel_deform(oracle_chisel* o, const int32_t* order, int n_order, const uint32_t* kfids, const float* Rt,
                         int n_map, int32_t* new_order, int new_cap, int64_t* stats) {
  oracle_chisel nw = *o;
  nw.cap = 1024; nw.count = 0;
  nw.tab = (chunk_t*)calloc(nw.cap, sizeof(chunk_t));
  const float res = o->resolution, inv_res = 1.f / res;          /* ChunkManager.cpp:67 */
  int n_new = 0;
  int64_t discarded = 0, undefined = 0;
  for (int c = 0; c < n_order; c++) {
    const int32_t* cid = order + 3 * (size_t)c;
    int found = 0;
    chunk_t* ch = tab_find(o->tab, o->cap, cid, &found);
    if (!found) continue;                                        /* (a caller error; the order lists existing chunks) */
    const float origin[3] = {(float)(16 * cid[0]) * res, (float)(16 * cid[1]) * res, (float)(16 * cid[2]) * res};
    for (int v = 0; v < CHUNK_VOX; v++) {
      if ((double)ch->weight[v] <= 1e-15) continue;              /* :953 */
      const uint32_t kf = ch->kfid[v];
      int lo = 0, hi = n_map - 1, at = -1;
      while (lo <= hi) { const int mid = (lo + hi) / 2; if (kfids[mid] == kf) { at = mid; break; } if (kfids[mid] < kf) lo = mid + 1; else hi = mid - 1; }
      if (at < 0) { discarded++; continue; }                     /* :960-966 */
      const float* R = Rt + 12 * (size_t)at;
      const int lx = v & 15, ly = (v >> 4) & 15, lz = v >> 8;
      const float pos[3] = {((float)lx * res + o->half_voxel) + origin[0],       /* centroids[voxelID] + origin  :972 */
                            ((float)ly * res + o->half_voxel) + origin[1],
                            ((float)lz * res + o->half_voxel) + origin[2]};
      float np[3];
      xform(R, R + 9, pos, np);                                  /* Rt.R * pos + Rt.t  :973 */
      const int32_t nid[3] = {(int32_t)floorf(np[0] * o->rounding), (int32_t)floorf(np[1] * o->rounding),
                              (int32_t)floorf(np[2] * o->rounding)};             /* GetIDAt  :977 */
      const int gx = (int)floorf(np[0] * inv_res), gy = (int)floorf(np[1] * inv_res), gz = (int)floorf(np[2] * inv_res);
      const int nx = gx - nid[0] * 16, ny = gy - nid[1] * 16, nz = gz - nid[2] * 16;   /* Chunk.cpp:101-105 */
      const int nv = (nz * 16 + ny) * 16 + nx;                   /* Chunk.h:90-93 */
      int f2 = 0;
      chunk_t* nc = tab_find(nw.tab, nw.cap, nid, &f2);
      if (!f2) {                                                 /* :981-990 (the chunk is created before the voxel is indexed) */
        nc = chunk_create(&nw, nid);
        if (n_new < new_cap) memcpy(new_order + 3 * (size_t)n_new, nid, sizeof(nid));
        n_new++;
      }
      if (nv < 0 || nv >= CHUNK_VOX) { undefined++; continue; }
      if ((double)nc->weight[nv] <= 1e-15) {                     /* :998-1003 */
        nc->sdf[nv] = ch->sdf[v]; nc->weight[nv] = ch->weight[v]; nc->kfid[nv] = kf; nc->rgbw[nv] = ch->rgbw[v];
      } else {                                                   /* :1004-1011 */
        dist_integrate(&nc->sdf[nv], &nc->weight[nv], ch->sdf[v], ch->weight[v]);
        nc->kfid[nv] = kf;
        const uint32_t p = ch->rgbw[v];
        nc->rgbw[nv] = colour_integrate(nc->rgbw[nv], (uint8_t)p, (uint8_t)(p >> 8), (uint8_t)(p >> 16), 1);
      }
    }
  }
  oracle_chisel_clear(o);
  free(o->tab);
  o->tab = nw.tab; o->cap = nw.cap; o->count = nw.count;         /* chunks.swap(newChunks)  :1015 */
  if (stats) { stats[0] = discarded; stats[1] = undefined; }
  return n_new;
}

/* The mesh part of ChunkManager::Deform (:1020-1051): vertex = R * vertex + t, normal = R * normal for the vertices
 * whose kfid has a transformation; the others stay. */
void oracle_chisel_deform_mesh(float* vertices, float* normals, const uint32_t* vkfid, int n, const uint32_t* kfids,
                               const float* Rt, int n_map) {
  for (int i = 0; i < n; i++) {
    int lo = 0, hi = n_map - 1, at = -1;
    while (lo <= hi) { const int mid = (lo + hi) / 2; if (kfids[mid] == vkfid[i]) { at = mid; break; } if (kfids[mid] < vkfid[i]) lo = mid + 1; else hi = mid - 1; }
    if (at < 0) continue;
    const float* R = Rt + 12 * (size_t)at;
    float v[3], nr[3];
    xform(R, R + 9, vertices + 3 * (size_t)i, v);
    for (int k = 0; k < 3; k++) nr[k] = sum3(R[3 * k] * normals[3 * (size_t)i], R[3 * k + 1] * normals[3 * (size_t)i + 1], R[3 * k + 2] * normals[3 * (size_t)i + 2]);
    memcpy(vertices + 3 * (size_t)i, v, sizeof(v));
    memcpy(normals + 3 * (size_t)i, nr, sizeof(nr));
  }
}

void oracle_chisel_set_visit_hook(oracle_chisel* o, void (*hook)(void*, const int32_t*), void* ctx) {
  o->visit_hook = hook; o->visit_hook_ctx = ctx; o->hook_has_last = 0;
}
int oracle_chisel_has_chunk(const oracle_chisel* o, int cx, int cy, int cz) {
  const int32_t id[3] = {cx, cy, cz};
  int found = 0;
  tab_find(o->tab, o->cap, id, &found);
  return found;
}

/* ---- the pieces above, one by one, for tests/test_oracle_pinned.py: checked there against the reference's own
 * Raycast.cpp / DistVoxel.h / ColorVoxel.h / QuadraticTruncator.h / ConstantWeighter.h compiled into
 * oracle/_ref/libchisel_ref.so (oracle/ref/). */
typedef struct { int32_t* out; int cap, n; } rc_list;
static void rc_collect(void* ctx, int x, int y, int z) {
  rc_list* l = (rc_list*)ctx;
  if (l->n < l->cap) { l->out[3 * l->n] = x; l->out[3 * l->n + 1] = y; l->out[3 * l->n + 2] = z; }
  l->n++;
}
int oracle_chisel_raycast(const float* start, const float* end, int32_t* out, int cap) {
  rc_list l = {out, cap, 0};
  raycast(start, end, rc_collect, &l);
  return l.n;
}
void oracle_chisel_dist_integrate(float* sdf, float* weight, float dist_update, float weight_update) {
  dist_integrate(sdf, weight, dist_update, weight_update);
}
void oracle_chisel_colour_integrate_simple(uint8_t* rgbw, uint8_t r, uint8_t g, uint8_t b, uint8_t weight_update) {
  const uint32_t p = colour_integrate_simple((uint32_t)rgbw[0] | ((uint32_t)rgbw[1] << 8) | ((uint32_t)rgbw[2] << 16) |
                                             ((uint32_t)rgbw[3] << 24), r, g, b, weight_update);
  rgbw[0] = p & 255; rgbw[1] = (p >> 8) & 255; rgbw[2] = (p >> 16) & 255; rgbw[3] = p >> 24;
}
void oracle_chisel_colour_integrate(uint8_t* rgbw, uint8_t r, uint8_t g, uint8_t b, uint8_t weight_update) {
  const uint32_t p = colour_integrate((uint32_t)rgbw[0] | ((uint32_t)rgbw[1] << 8) | ((uint32_t)rgbw[2] << 16) |
                                      ((uint32_t)rgbw[3] << 24), r, g, b, weight_update);
  rgbw[0] = p & 255; rgbw[1] = (p >> 8) & 255; rgbw[2] = (p >> 16) & 255; rgbw[3] = p >> 24;
}
float oracle_chisel_diag(float resolution) { return (float)(2.0 * sqrt((double)3.0f) * (double)resolution); }   /* as oracle_chisel_integrate computes it */
float oracle_chisel_truncation(float q, float l, float c, float s, float reading) { return quadratic_truncation(q, l, c, s, reading); }
float oracle_chisel_weight(float weight, float surface_dist, float truncation) { (void)surface_dist; return constant_weight(weight, truncation); }
void oracle_chisel_mc_tables(int* triangle_table, int* edge_index_pairs) {
  memcpy(triangle_table, kTriangleTable, sizeof(kTriangleTable));
  memcpy(edge_index_pairs, kEdgeIndexPairs, sizeof(kEdgeIndexPairs));
}
