/* ORACLE — test infrastructure only.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may call this; the product (plvs_amd/) never does.
 *
 * CPU restatement of the dense stereo path PLVS runs through libsgm:
 *   PointCloudKeyFrame::ProcessStereoLibsgm          src/PointCloudKeyFrame.cc:435-481
 *     sgm::StereoSGM(w, h, 64, 8, 8, HOST2HOST), P1 = 10, P2 = 120, uniqueness = 0.95 (libsgm.h:62-70)
 *   StereoSGM::execute                               Thirdparty/libsgm/src/stereo_sgm.cpp:133-181
 *     census_transform_kernel        census_transform.cu:33-101    9 x 7 centre-symmetric census, 31 bits
 *     DynamicProgramming::update     path_aggregation_common.hpp:45-92   the SGM recurrence, 8 paths
 *       aggregate_{vertical,horizontal,oblique}_path_kernel (start / border rules of the paths)
 *     winner_takes_all_kernel        winner_takes_all.cu:109-236   sum of the 8 paths, best two, uniqueness
 *     median_kernel_3x3_8u[_v4]      median_filter.cu:95-189       interior pixels only
 *     check_consistency_kernel       check_consistency.cu:22-37    on the (w/16*16) x (h/16*16) region only
 * The reference is CUDA; every stage is integer arithmetic (one float comparison in the uniqueness test),
 * so the result does not depend on how the work is split into threads.
 *
 * One point the reference leaves undefined: the census feature of the 4 / 3-pixel image border is never
 * written (census_transform.cu:71) and the buffer is not cleared (device_buffer.hpp) — taken as 0 here.
 * Parity unpinned: the reference holds no stored output for libsgm (the stored disparities under
 * Thirdparty/libelas-gpu/GPU_test are libelas').
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { MAX_DISPARITY = 64, NUM_PATHS = 8 };

static void census(const uint8_t* src, int w, int h, uint32_t* dst) {
  memset(dst, 0, sizeof(uint32_t) * (size_t)w * h);
  for (int y = 3; y < h - 3; y++)
    for (int x = 4; x < w - 4; x++) {
      uint32_t f = 0;
      for (int dy = -3; dy < 0; dy++)
        for (int dx = -4; dx <= 4; dx++) {
          const uint8_t a = src[(size_t)(y + dy) * w + (x + dx)], b = src[(size_t)(y - dy) * w + (x - dx)];
          f = (f << 1) | (uint32_t)(a > b);
        }
      for (int dx = -4; dx < 0; dx++) {
        const uint8_t a = src[(size_t)y * w + (x + dx)], b = src[(size_t)y * w + (x - dx)];
        f = (f << 1) | (uint32_t)(a > b);
      }
      dst[(size_t)y * w + x] = f;
    }
}

static int popc(uint32_t v) { return __builtin_popcount(v); }

typedef struct {
  uint32_t dp[MAX_DISPARITY];
  uint32_t last_min;
} dp_state;

/* DynamicProgramming::update for one pixel of a path; writes the new costs as bytes. */
static void dp_update(dp_state* s, const uint32_t* left, const uint32_t* right, int w, int x, int y, uint32_t p1,
                      uint32_t p2, uint8_t* dest) {
  const uint32_t fl = left[(size_t)y * w + x];
  uint32_t out[MAX_DISPARITY], mn = 0xffffffffu;
  for (int d = 0; d < MAX_DISPARITY; d++) {
    const uint32_t fr = (x - d >= 0) ? right[(size_t)y * w + (x - d)] : 0u;
    uint32_t o = s->dp[d] - s->last_min;
    if (o > p2) o = p2;
    if (d > 0) { const uint32_t t = s->dp[d - 1] - s->last_min + p1; if (t < o) o = t; }
    if (d + 1 < MAX_DISPARITY) { const uint32_t t = s->dp[d + 1] - s->last_min + p1; if (t < o) o = t; }
    out[d] = o + (uint32_t)popc(fl ^ fr);
    if (out[d] < mn) mn = out[d];
  }
  memcpy(s->dp, out, sizeof out);
  s->last_min = mn;
  uint8_t* q = dest + ((size_t)y * w + x) * MAX_DISPARITY;
  for (int d = 0; d < MAX_DISPARITY; d++) q[d] = (uint8_t)out[d];
}

/* One path direction (dx, dy): every path starts with dp = 0, last_min = 0 where it enters the image. */
static void aggregate(const uint32_t* left, const uint32_t* right, int w, int h, int dx, int dy, uint32_t p1,
                      uint32_t p2, uint8_t* dest) {
  dp_state s;
  if (dy == 0) {                                        /* horizontal_path_aggregation.cu */
    for (int y = 0; y < h; y++) {
      memset(&s, 0, sizeof s);
      for (int i = 0; i < w; i++) dp_update(&s, left, right, w, dx > 0 ? i : w - 1 - i, y, p1, p2, dest);
    }
  } else if (dx == 0) {                                 /* vertical_path_aggregation.cu */
    for (int x = 0; x < w; x++) {
      memset(&s, 0, sizeof s);
      for (int i = 0; i < h; i++) dp_update(&s, left, right, w, x, dy > 0 ? i : h - 1 - i, p1, p2, dest);
    }
  } else {                                              /* oblique_path_aggregation.cu: one path per diagonal */
    for (int x0 = -(h - 1); x0 < w + (h - 1); x0++) {
      memset(&s, 0, sizeof s);
      for (int i = 0; i < h; i++) {
        const int y = dy > 0 ? i : h - 1 - i;
        const int x = x0 + i * dx;
        if (x >= 0 && x < w) dp_update(&s, left, right, w, x, y, p1, p2, dest);
      }
    }
  }
}

static uint32_t compute_disparity(uint32_t v0, uint32_t v1, float uniqueness) {   /* winner_takes_all.cu:94-107 */
  const float cost0 = (float)(v0 >> 16), cost1 = (float)(v1 >> 16);
  const int disp0 = (int)(v0 & 0xffffu), disp1 = (int)(v1 & 0xffffu);
  if (cost1 * uniqueness >= cost0) return (uint32_t)disp0;
  if (abs(disp1 - disp0) <= 1) return (uint32_t)disp0;
  return 0;
}

static void push_top2(uint32_t* v0, uint32_t* v1, uint32_t x) {
  const uint32_t y = x > *v0 ? x : *v0;
  if (x < *v0) *v0 = x;
  if (y < *v1) *v1 = y;
}

static void winner_takes_all(const uint8_t* cost, int w, int h, float uniqueness, uint8_t* left_disp,
                             uint8_t* right_disp, uint16_t* sum) {
  const size_t step = (size_t)w * h * MAX_DISPARITY;
  for (size_t i = 0; i < step; i++) {
    uint32_t s = 0;
    for (int p = 0; p < NUM_PATHS; p++) s += cost[p * step + i];
    sum[i] = (uint16_t)s;
  }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      uint32_t v0 = 0xffffffffu, v1 = 0xffffffffu;
      for (int d = 0; d < MAX_DISPARITY; d++)
        push_top2(&v0, &v1, ((uint32_t)sum[((size_t)y * w + x) * MAX_DISPARITY + d] << 16) | (uint32_t)d);
      left_disp[(size_t)y * w + x] = (uint8_t)compute_disparity(v0, v1, uniqueness);
      /* right image, pixel p = x: the costs of the left pixels p + d at disparity d (winner_takes_all.cu:196-236) */
      v0 = v1 = 0xffffffffu;
      for (int d = 0; d < MAX_DISPARITY && x + d < w; d++)
        push_top2(&v0, &v1, ((uint32_t)sum[((size_t)y * w + (x + d)) * MAX_DISPARITY + d] << 16) | (uint32_t)d);
      right_disp[(size_t)y * w + x] = (uint8_t)compute_disparity(v0, v1, uniqueness);
    }
}

static int cmp_u8(const void* a, const void* b) { return (int)*(const uint8_t*)a - (int)*(const uint8_t*)b; }

static void median3x3(const uint8_t* src, int w, int h, uint8_t* dst) {   /* dst border stays 0 (cudaMemset at allocation) */
  memset(dst, 0, (size_t)w * h);
  for (int y = 1; y < h - 1; y++)
    for (int x = 1; x < w - 1; x++) {
      uint8_t b[9];
      for (int i = 0; i < 9; i++) b[i] = src[(size_t)(y - 1 + i / 3) * w + (x - 1 + i % 3)];
      qsort(b, 9, 1, cmp_u8);
      dst[(size_t)y * w + x] = b[4];
    }
}

/* left / right: w x h u8 images; disparity: w x h u8 (0 = invalid).  Optional stage outputs for the tests
 * (any may be NULL): census_left, census_right (w*h u32), cost_sum (w*h*64 u16), raw_left, raw_right
 * (before the median), median_left, median_right. */
void oracle_sgm(const uint8_t* left, const uint8_t* right, int w, int h, int p1, int p2, float uniqueness,
                uint8_t* disparity, uint32_t* census_left, uint32_t* census_right, uint16_t* cost_sum,
                uint8_t* raw_left, uint8_t* raw_right, uint8_t* median_left, uint8_t* median_right) {
  const size_t n = (size_t)w * h;
  uint32_t* cl = (uint32_t*)malloc(sizeof(uint32_t) * n);
  uint32_t* cr = (uint32_t*)malloc(sizeof(uint32_t) * n);
  census(left, w, h, cl);
  census(right, w, h, cr);
  uint8_t* cost = (uint8_t*)malloc(n * MAX_DISPARITY * NUM_PATHS);
  const size_t step = n * MAX_DISPARITY;
  static const int dirs[NUM_PATHS][2] = {{0, 1}, {0, -1}, {1, 0}, {-1, 0}, {1, 1}, {-1, 1}, {-1, -1}, {1, -1}};  /* path_aggregation.cu:56-79 */
  for (int p = 0; p < NUM_PATHS; p++) aggregate(cl, cr, w, h, dirs[p][0], dirs[p][1], (uint32_t)p1, (uint32_t)p2, cost + p * step);
  uint16_t* sum = (uint16_t*)malloc(sizeof(uint16_t) * step);
  uint8_t* dl = (uint8_t*)malloc(n);
  uint8_t* dr = (uint8_t*)malloc(n);
  uint8_t* ml = (uint8_t*)malloc(n);
  uint8_t* mr = (uint8_t*)malloc(n);
  winner_takes_all(cost, w, h, uniqueness, dl, dr, sum);
  median3x3(dl, w, h, ml);
  median3x3(dr, w, h, mr);
  memcpy(disparity, ml, n);
  const int cw = (w / 16) * 16, ch = (h / 16) * 16;                        /* grid = (w / 16, h / 16) blocks of 16 x 16 */
  for (int i = 0; i < ch; i++)
    for (int j = 0; j < cw; j++) {
      const int d = ml[(size_t)i * w + j], k = j - d;
      if (left[(size_t)i * w + j] == 0 || d <= 0 || (k >= 0 && k < w && abs((int)mr[(size_t)i * w + k] - d) > 1))
        disparity[(size_t)i * w + j] = 0;
    }
  if (census_left) memcpy(census_left, cl, sizeof(uint32_t) * n);
  if (census_right) memcpy(census_right, cr, sizeof(uint32_t) * n);
  if (cost_sum) memcpy(cost_sum, sum, sizeof(uint16_t) * step);
  if (raw_left) memcpy(raw_left, dl, n);
  if (raw_right) memcpy(raw_right, dr, n);
  if (median_left) memcpy(median_left, ml, n);
  if (median_right) memcpy(median_right, mr, n);
  free(cl); free(cr); free(cost); free(sum); free(dl); free(dr); free(ml); free(mr);
}
