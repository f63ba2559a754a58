/* TEST INFRASTRUCTURE ONLY — CPU restatement of the two methods of libelas that the reference's own accelerated
 * build moves to the GPU (class ElasGPU : public Elas, Thirdparty/libelas-gpu/GPU/elas_gpu.h:41-45):
 *
 *   Elas::computeDisparity   Thirdparty/libelas-gpu/CPU/elas.cpp:840-968  (findMatch :739-837,
 *                            updatePosteriorMinimum :717-737)
 *   Elas::adaptiveMean       Thirdparty/libelas-gpu/CPU/elas.cpp:1349-1572
 *
 * PLVS reaches them through PointCloudKeyFrame::ProcessStereoLibelas (src/PointCloudKeyFrame.cc:335-432) ->
 * libelas::ElasInterface::process -> Elas::process (elas.cpp:36-159).  Everything else of Elas::process (descriptors,
 * support matches, Delaunay triangulation, planes, grid, left/right check, speckles, gap interpolation, median) stays
 * the reference's host code, exactly as in its GPU build.
 *
 * Plain C, no SSE: _mm_sad_epu8 of two 16-byte descriptors + the two extracts is the sum of the 16 absolute byte
 * differences; the 4- / 8-float registers of adaptiveMean are written out slot by slot IN THE REGISTER'S ORDER (the
 * float sums are not associative).  Pinned: tests/test_elas.py compares both functions with the reference's compiled
 * methods (oracle/_ref/libelas_ref.so, oracle/ref/elas_ref_wrap.cpp) on the arguments the reference pipeline itself
 * produces for real stereo pairs.  Nothing under plvs_amd/ may call into this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int32_t subsampling;   /* Elas::Parameters, elas.h:62-90: the fields the two methods read */
  int32_t grid_size;
  int32_t match_texture;
  float beta, gamma, sigma, sradius;
} oracle_elas_params;

typedef struct { int32_t u, v, d; } elas_support_pt;                                   /* elas.h:178-183 */
typedef struct { int32_t c1, c2, c3; float t1a, t1b, t1c, t2a, t2b, t2c; } elas_triangle;   /* elas.h:185-190 */

static int32_t imin(int32_t a, int32_t b) { return a < b ? a : b; }
static int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }

/* (uint32_t)(float) assigned to an int32_t (elas.cpp:927-928, :947-948): on x86-64 the conversion goes through the
 * 64-bit truncation and keeps the low 32 bits, so a negative product comes back as the negative integer */
static int32_t trunc_u32_as_i32(float x) { return (int32_t)(uint32_t)(int64_t)x; }

static int32_t sad16(const uint8_t* a, const uint8_t* b) {
  int32_t s = 0;
  for (int i = 0; i < 16; ++i) s += abs((int32_t)a[i] - (int32_t)b[i]);
  return s;
}

/* Elas::findMatch (elas.cpp:739-837) */
static void find_match(const oracle_elas_params* p, int32_t width, int32_t height, int32_t u, int32_t v, float plane_a, float plane_b,
                       float plane_c, const int32_t* disparity_grid, const int32_t* grid_dims, const uint8_t* I1_desc,
                       const uint8_t* I2_desc, const int32_t* P, int32_t plane_radius, int valid, int right_image, float* D) {
  const int32_t disp_num = grid_dims[0] - 1;
  const int32_t window_size = 2;
  const uint32_t d_addr = p->subsampling ? (uint32_t)((v / 2) * (width / 2) + u / 2) : (uint32_t)(v * width + u);
  if (u < window_size || u >= width - window_size) return;
  const int32_t line_offset = 16 * width * imax(imin(v, height - 3), 2);
  const uint8_t* I1_line = (right_image ? I2_desc : I1_desc) + line_offset;
  const uint8_t* I2_line = (right_image ? I1_desc : I2_desc) + line_offset;
  const uint8_t* I1_block = I1_line + 16 * u;
  int32_t sum = 0;
  for (int i = 0; i < 16; ++i) sum += abs((int32_t)I1_block[i] - 128);
  if (sum < p->match_texture) return;
  const int32_t d_plane = (int32_t)(plane_a * (float)u + plane_b * (float)v + plane_c);
  const int32_t d_plane_min = imax(d_plane - plane_radius, 0);
  const int32_t d_plane_max = imin(d_plane + plane_radius, disp_num - 1);
  const int32_t grid_x = (int32_t)floorf((float)u / (float)p->grid_size);
  const int32_t grid_y = (int32_t)floorf((float)v / (float)p->grid_size);
  const uint32_t grid_addr = (uint32_t)((grid_y * grid_dims[1] + grid_x) * grid_dims[0]);   /* getAddressOffsetGrid, d = 0 */
  const int32_t num_grid = disparity_grid[grid_addr];
  const int32_t* d_grid = disparity_grid + grid_addr + 1;
  int32_t min_val = 10000, min_d = -1;
  const int32_t sign = right_image ? 1 : -1;
  for (int32_t i = 0; i < num_grid; ++i) {
    const int32_t d_curr = d_grid[i];
    if (d_curr < d_plane_min || d_curr > d_plane_max) {
      const int32_t u_warp = u + sign * d_curr;
      if (u_warp < window_size || u_warp >= width - window_size) continue;
      const int32_t val = sad16(I1_block, I2_line + 16 * u_warp);
      if (val < min_val) { min_val = val; min_d = d_curr; }
    }
  }
  for (int32_t d_curr = d_plane_min; d_curr <= d_plane_max; ++d_curr) {
    const int32_t u_warp = u + sign * d_curr;
    if (u_warp < window_size || u_warp >= width - window_size) continue;
    const int32_t val = sad16(I1_block, I2_line + 16 * u_warp) + (valid ? P[abs(d_curr - d_plane)] : 0);
    if (val < min_val) { min_val = val; min_d = d_curr; }
  }
  D[d_addr] = min_d >= 0 ? (float)min_d : -1.0f;
}

/* Elas::computeDisparity (elas.cpp:840-968) */
void oracle_elas_compute_disparity(const oracle_elas_params* p, const elas_support_pt* p_support, int32_t n_support,
                                   const elas_triangle* tri, int32_t n_tri, const int32_t* disparity_grid, const int32_t* grid_dims,
                                   const uint8_t* I1_desc, const uint8_t* I2_desc, int32_t width, int32_t height, int32_t right_image,
                                   float* D) {
  (void)n_support;
  const int32_t disp_num = grid_dims[0] - 1;
  const int32_t npix = p->subsampling ? (width / 2) * (height / 2) : width * height;
  for (int32_t i = 0; i < npix; ++i) D[i] = -10.0f;
  const float two_sigma_squared = 2 * p->sigma * p->sigma;
  int32_t* P = (int32_t*)malloc(sizeof(int32_t) * (size_t)(disp_num > 0 ? disp_num : 1));
  /* (elas.cpp:861-864 under `using namespace std` with <math.h>: exp / log / ceil of float arguments bind to the float
   * overloads, the whole expression is evaluated in float) */
  for (int32_t delta_d = 0; delta_d < disp_num; ++delta_d)
    P[delta_d] = (int32_t)((-logf(p->gamma + expf((float)(-delta_d * delta_d) / two_sigma_squared)) + logf(p->gamma)) / p->beta);
  const int32_t plane_radius = (int32_t)fmaxf(ceilf(p->sigma * p->sradius), 2.0f);
  for (int32_t i = 0; i < n_tri; ++i) {
    float plane_a, plane_b, plane_c, plane_d;
    if (!right_image) { plane_a = tri[i].t1a; plane_b = tri[i].t1b; plane_c = tri[i].t1c; plane_d = tri[i].t2a; }
    else              { plane_a = tri[i].t2a; plane_b = tri[i].t2b; plane_c = tri[i].t2c; plane_d = tri[i].t1a; }
    const int32_t c[3] = {tri[i].c1, tri[i].c2, tri[i].c3};
    float tri_u[3], tri_v[3];
    for (int k = 0; k < 3; ++k) {
      tri_u[k] = right_image ? (float)(p_support[c[k]].u - p_support[c[k]].d) : (float)p_support[c[k]].u;
      tri_v[k] = (float)p_support[c[k]].v;
    }
    for (int j = 0; j < 3; ++j)
      for (int k = 0; k < j; ++k)
        if (tri_u[k] > tri_u[j]) {
          const float tu = tri_u[j]; tri_u[j] = tri_u[k]; tri_u[k] = tu;
          const float tv = tri_v[j]; tri_v[j] = tri_v[k]; tri_v[k] = tv;
        }
    const float A_u = tri_u[0], A_v = tri_v[0], B_u = tri_u[1], B_v = tri_v[1], C_u = tri_u[2], C_v = tri_v[2];
    float AB_a = 0, AC_a = 0, BC_a = 0;
    if ((int32_t)A_u != (int32_t)B_u) AB_a = (A_v - B_v) / (A_u - B_u);
    if ((int32_t)A_u != (int32_t)C_u) AC_a = (A_v - C_v) / (A_u - C_u);
    if ((int32_t)B_u != (int32_t)C_u) BC_a = (B_v - C_v) / (B_u - C_u);
    const float AB_b = A_v - AB_a * A_u, AC_b = A_v - AC_a * A_u, BC_b = B_v - BC_a * B_u;
    const int valid = fabs(plane_a) < 0.7 && fabs(plane_d) < 0.7;
    for (int part = 0; part < 2; ++part) {
      const float lo_u = part ? B_u : A_u, hi_u = part ? C_u : B_u, e_a = part ? BC_a : AB_a, e_b = part ? BC_b : AB_b;
      if ((int32_t)lo_u == (int32_t)hi_u) continue;
      for (int32_t u = imax((int32_t)lo_u, 0); u < imin((int32_t)hi_u, width); ++u) {
        if (p->subsampling && u % 2 != 0) continue;
        const int32_t v_1 = trunc_u32_as_i32(AC_a * (float)u + AC_b);
        const int32_t v_2 = trunc_u32_as_i32(e_a * (float)u + e_b);
        for (int32_t v = imin(v_1, v_2); v < imax(v_1, v_2); ++v)
          if (!p->subsampling || v % 2 == 0)
            find_match(p, width, height, u, v, plane_a, plane_b, plane_c, disparity_grid, grid_dims, I1_desc, I2_desc, P,
                       plane_radius, valid, right_image, D);
      }
    }
  }
  free(P);
}

/* one output of Elas::adaptiveMean's filters: the window's values by REGISTER SLOT (slot = pixel index mod n), the
 * weights max(0, 4 - |val - centre|) — with subsampling the "absolute value" is the reference's
 * _mm_and_ps(x, _mm_set1_ps(0x7FFFFFFF)) (elas.cpp:1379, :1402): the mask is the FLOAT 2147483648.0f = 0x4F000000, so
 * it keeps four exponent bits and nothing else — summed in the order of the four-lane registers */
static int mean_of_window(const float* val, int n, float centre, int masked_abs, float* out) {
  float weight[8], factor[8];
  for (int s = 0; s < n; ++s) {
    float w = val[s] - centre;
    if (masked_abs) {
      uint32_t b;
      memcpy(&b, &w, 4);
      b &= 0x4F000000u;
      memcpy(&w, &b, 4);
    } else {
      const float neg = 0.0f - w;
      w = (neg > w) ? neg : w;   /* _mm_max_ps(0 - x, x) */
    }
    w = 4.0f - w;
    w = (0.0f > w) ? 0.0f : w;   /* _mm_max_ps(0, x) */
    weight[s] = w;
    factor[s] = val[s] * w;
  }
  if (n == 8)
    for (int s = 0; s < 4; ++s) {
      weight[s] = weight[s] + weight[s + 4];
      factor[s] = factor[s] + factor[s + 4];
    }
  const float weight_sum = weight[0] + weight[1] + weight[2] + weight[3];
  const float factor_sum = factor[0] + factor[1] + factor[2] + factor[3];
  if (weight_sum > 0) {
    const float d = factor_sum / weight_sum;
    if (d >= 0) {
      *out = d;
      return 1;
    }
  }
  return 0;
}

/* Elas::adaptiveMean (elas.cpp:1349-1572).  width x height: the IMAGE size (D is half of it with subsampling).
 * The scratch image D_tmp is uninitialised malloc memory in the reference wherever the horizontal pass does not write
 * (rows 0-2 and the last three, the outermost columns) and the vertical pass reads some of it: 0.0 here, the content
 * of fresh pages (what a large malloc returns in a fresh process; oracle/ref/elas_ref_wrap.cpp keeps the compiled
 * reference on such pages). */
void oracle_elas_adaptive_mean(float* D, int32_t width, int32_t height, int32_t subsampling) {
  const int32_t W = subsampling ? width / 2 : width, H = subsampling ? height / 2 : height;
  const size_t n = (size_t)W * (size_t)H;
  float* D_copy = (float*)malloc(n * sizeof(float));
  float* D_tmp = (float*)calloc(n, sizeof(float));
  memcpy(D_copy, D, n * sizeof(float));
  for (size_t i = 0; i < n; ++i)
    if (D[i] < 0) { D_copy[i] = -10.0f; D_tmp[i] = -10.0f; }
  const int taps = subsampling ? 4 : 8, back = subsampling ? 1 : 3;
  float val[8];
  /* horizontal: the window ends at u, its output is `back` pixels behind */
  for (int32_t v = 3; v < H - 3; ++v) {
    for (int32_t u = 0; u < taps - 1; ++u) val[u] = D_copy[v * W + u];
    for (int32_t u = taps - 1; u < W; ++u) {
      const float centre = D_copy[v * W + (u - back)];
      val[u % taps] = D_copy[v * W + u];
      float d;
      if (mean_of_window(val, taps, centre, subsampling, &d)) D_tmp[v * W + (u - back)] = d;
    }
  }
  /* vertical */
  for (int32_t u = 3; u < W - 3; ++u) {
    for (int32_t v = 0; v < taps - 1; ++v) val[v] = D_tmp[v * W + u];
    for (int32_t v = taps - 1; v < H; ++v) {
      const float centre = D_tmp[(v - back) * W + u];
      val[v % taps] = D_tmp[v * W + u];
      float d;
      if (mean_of_window(val, taps, centre, subsampling, &d)) D[(v - back) * W + u] = d;
    }
  }
  free(D_copy);
  free(D_tmp);
}

/* ------------------------------------------------------------------ the candidate grid of Elas::computeSupportMatches
 * (elas.cpp:416-489, computeMatchingDisparity :296-410): for every point of the regular grid the forward match in the
 * other image, confirmed by the backward match from there.  What follows in computeSupportMatches —
 * removeInconsistentSupportPoints, removeRedundantSupportPoints, the conversion to a vector, addCornerSupportPoints —
 * works on this small grid and stays the reference's host code. */
typedef struct {
  int32_t subsampling, candidate_stepsize, disp_min, disp_max, support_texture, lr_threshold;
  float support_threshold;
} oracle_elas_support_params;

static int16_t matching_disparity(const oracle_elas_support_params* p, int32_t width, int32_t height, int32_t u, int32_t v,
                                  const uint8_t* I1_desc, const uint8_t* I2_desc, int right_image) {
  const int32_t u_step = 2, v_step = 2, window_size = 3;
  const int32_t off[4] = {-16 * u_step - 16 * width * v_step, +16 * u_step - 16 * width * v_step,
                          -16 * u_step + 16 * width * v_step, +16 * u_step + 16 * width * v_step};
  if (!(u >= window_size + u_step && u <= width - window_size - 1 - u_step && v >= window_size + v_step &&
        v <= height - window_size - 1 - v_step))
    return -1;
  const int32_t line_offset = 16 * width * v;
  const uint8_t* I1_line = (right_image ? I2_desc : I1_desc) + line_offset;
  const uint8_t* I2_line = (right_image ? I1_desc : I2_desc) + line_offset;
  const uint8_t* I1_block = I1_line + 16 * u;
  int32_t sum = 0;
  for (int i = 0; i < 16; ++i) sum += abs((int32_t)I1_block[i] - 128);
  if (sum < p->support_texture) return -1;
  int16_t min_1_E = 32767, min_1_d = -1, min_2_E = 32767, min_2_d = -1;
  const int32_t disp_min_valid = imax(p->disp_min, 0);
  const int32_t disp_max_valid = right_image ? imin(p->disp_max, width - u - window_size - u_step)
                                             : imin(p->disp_max, u - window_size - u_step);
  if (disp_max_valid - disp_min_valid < 10) return -1;
  for (int16_t d = (int16_t)disp_min_valid; d <= disp_max_valid; ++d) {
    const int32_t u_warp = right_image ? u + d : u - d;
    const uint8_t* I2_block = I2_line + 16 * u_warp;
    sum = 0;
    for (int k = 0; k < 4; ++k) sum += sad16(I1_block + off[k], I2_block + off[k]);
    if (sum < min_1_E) {
      min_2_E = min_1_E;
      min_2_d = min_1_d;
      min_1_E = (int16_t)sum;
      min_1_d = d;
    } else if (sum < min_2_E) {
      min_2_E = (int16_t)sum;
      min_2_d = d;
    }
  }
  if (min_1_d >= 0 && min_2_d >= 0 && (float)min_1_E < p->support_threshold * (float)min_2_E) return min_1_d;
  return -1;
}

/* D_can: D_can_width x D_can_height int16 (the counts of elas.cpp:425-428), zero-initialised by the caller as the
 * reference's calloc leaves it: row 0 and column 0 are never written. */
void oracle_elas_support_candidates(const oracle_elas_support_params* p, const uint8_t* I1_desc, const uint8_t* I2_desc,
                                    int32_t width, int32_t height, int16_t* D_can) {
  int32_t step = p->candidate_stepsize;
  if (p->subsampling) step += step % 2;
  int32_t W = 0, H = 0;
  for (int32_t u = 0; u < width; u += step) ++W;
  for (int32_t v = 0; v < height; v += step) ++H;
  for (int32_t u_can = 1; u_can < W; ++u_can) {
    const int32_t u = u_can * step;
    for (int32_t v_can = 1; v_can < H; ++v_can) {
      const int32_t v = v_can * step;
      D_can[v_can * W + u_can] = -1;
      const int16_t d = matching_disparity(p, width, height, u, v, I1_desc, I2_desc, 0);
      if (d >= 0) {
        const int16_t d2 = matching_disparity(p, width, height, u - d, v, I1_desc, I2_desc, 1);
        if (d2 >= 0 && abs(d - d2) <= p->lr_threshold) D_can[v_can * W + u_can] = d;
      }
    }
  }
}

/* ------------------------------------------------------------------ the post-processing between computeDisparity and
 * adaptiveMean: Elas::leftRightConsistencyCheck (elas.cpp:971-1040), Elas::removeSmallSegments (:1043-1160),
 * Elas::gapInterpolation (:1163-1347).  W x H below is the DISPARITY MAP's size (half the image's with subsampling). */

/* both maps against each other; invalid = -10 */
void oracle_elas_left_right_check(float* D1, float* D2, int32_t W, int32_t H, int32_t subsampling, int32_t lr_threshold) {
  const size_t n = (size_t)W * H;
  float* c1 = (float*)malloc(n * sizeof(float));
  float* c2 = (float*)malloc(n * sizeof(float));
  memcpy(c1, D1, n * sizeof(float));
  memcpy(c2, D2, n * sizeof(float));
  for (int32_t u = 0; u < W; ++u)
    for (int32_t v = 0; v < H; ++v) {
      const size_t addr = (size_t)v * W + u;
      const float d1 = c1[addr], d2 = c2[addr];
      const float u_warp_1 = subsampling ? (float)u - d1 / 2 : (float)u - d1;
      const float u_warp_2 = subsampling ? (float)u + d2 / 2 : (float)u + d2;
      if (d1 >= 0 && u_warp_1 >= 0 && u_warp_1 < W) {
        if (fabs(c2[(size_t)v * W + (int32_t)u_warp_1] - d1) > lr_threshold) D1[addr] = -10;
      } else {
        D1[addr] = -10;
      }
      if (d2 >= 0 && u_warp_2 >= 0 && u_warp_2 < W) {
        if (fabs(c1[(size_t)v * W + (int32_t)u_warp_2] - d2) > lr_threshold) D2[addr] = -10;
      } else {
        D2[addr] = -10;
      }
    }
  free(c1);
  free(c2);
}

/* segments of 4-connected pixels whose neighbouring disparities differ by at most speckle_sim_threshold, grown in the
 * reference's order; segments smaller than the speckle size are invalidated.  speckle_size: Parameters::speckle_size
 * (the reference shrinks it to sqrt(size) * 2 with subsampling, elas.cpp:1051). */
void oracle_elas_remove_small_segments(float* D, int32_t W, int32_t H, int32_t subsampling, int32_t speckle_size,
                                       float speckle_sim_threshold) {
  int32_t D_speckle_size = speckle_size;
  if (subsampling) D_speckle_size = (int32_t)(sqrtf((float)speckle_size) * 2);
  const size_t n = (size_t)W * H;
  int32_t* done = (int32_t*)calloc(n, sizeof(int32_t));
  int32_t* list_u = (int32_t*)calloc(n, sizeof(int32_t));
  int32_t* list_v = (int32_t*)calloc(n, sizeof(int32_t));
  for (int32_t u = 0; u < W; ++u)
    for (int32_t v = 0; v < H; ++v) {
      if (done[(size_t)v * W + u] != 0) continue;
      list_u[0] = u;
      list_v[0] = v;
      int32_t count = 1, curr = 0;
      while (curr < count) {
        const int32_t uc = list_u[curr], vc = list_v[curr];
        const size_t addr_curr = (size_t)vc * W + uc;
        const int32_t nu[4] = {uc - 1, uc + 1, uc, uc}, nv[4] = {vc, vc, vc - 1, vc + 1};
        for (int i = 0; i < 4; ++i) {
          if (nu[i] >= 0 && nv[i] >= 0 && nu[i] < W && nv[i] < H) {
            const size_t addr_n = (size_t)nv[i] * W + nu[i];
            if (done[addr_n] == 0 && D[addr_n] >= 0 && fabs(D[addr_curr] - D[addr_n]) <= speckle_sim_threshold) {
              list_u[count] = nu[i];
              list_v[count] = nv[i];
              ++count;
              done[addr_n] = 1;
            }
          }
        }
        ++curr;
        done[addr_curr] = 1;
      }
      if (count < D_speckle_size)
        for (int32_t i = 0; i < count; ++i) D[(size_t)list_v[i] * W + list_u[i]] = -10;
    }
  free(done);
  free(list_u);
  free(list_v);
}

/* one line of gapInterpolation: D[0], D[stride], ... D[(len-1) * stride] */
static void gap_line(float* D, int32_t len, size_t stride, int32_t gap_width, int32_t add_corners) {
  const float discon_threshold = 3.0f;
  int32_t count = 0;
  for (int32_t i = 0; i < len; ++i) {
    if (D[(size_t)i * stride] >= 0) {
      if (count >= 1 && count <= gap_width) {
        const int32_t first = i - count, last = i - 1;
        if (first > 0 && last < len - 1) {
          const float d1 = D[(size_t)(first - 1) * stride], d2 = D[(size_t)(last + 1) * stride];
          const float d_ipol = (fabs(d1 - d2) < discon_threshold) ? (d1 + d2) / 2 : (d1 < d2 ? d1 : d2);   /* std::min(d1, d2) */
          for (int32_t k = first; k <= last; ++k) D[(size_t)k * stride] = d_ipol;
        }
      }
      count = 0;
    } else {
      ++count;
    }
  }
  if (add_corners) {
    for (int32_t i = 0; i < len; ++i)
      if (D[(size_t)i * stride] >= 0) {
        for (int32_t k = imax(i - gap_width, 0); k < i; ++k) D[(size_t)k * stride] = D[(size_t)i * stride];
        break;
      }
    for (int32_t i = len - 1; i >= 0; --i)
      if (D[(size_t)i * stride] >= 0) {
        for (int32_t k = i; k <= imin(i + gap_width, len - 1); ++k) D[(size_t)k * stride] = D[(size_t)i * stride];
        break;
      }
  }
}

void oracle_elas_gap_interpolation(float* D, int32_t W, int32_t H, int32_t subsampling, int32_t ipol_gap_width, int32_t add_corners) {
  const int32_t gap = subsampling ? ipol_gap_width / 2 + 1 : ipol_gap_width;
  for (int32_t v = 0; v < H; ++v) gap_line(D + (size_t)v * W, W, 1, gap, add_corners);
  for (int32_t u = 0; u < W; ++u) gap_line(D + u, H, (size_t)W, gap, add_corners);
}

/* ------------------------------------------------------------------ Descriptor (descriptor.cpp:30-131) on an image as
 * Elas::process hands it over (elas.cpp:39-57: rows copied into a zeroed buffer whose line length bpl is the width rounded
 * up to 16): filter::sobel3x3 (filter.cpp:410-418) and the 16 samples per pixel.
 *
 * sobel3x3 works on the buffer as ONE flat array: a column pass over whole lines (rows 1 .. h-2), then row filters that
 * run across the line ends and store at an offset of one — convolve_101_row_3x3_16bit (:229-271) with a scalar tail that
 * does not saturate, convolve_121_row_3x3_16bit (:178-224) without a tail.  What they never write (first element, the last
 * few, rows 0 and h-1 of the temporaries) is read as zero, the content of fresh pages (oracle/ref/elas_zero_malloc.h pins
 * the compiled reference to the same).  desc: 16 * width * height bytes, zero where the reference writes nothing
 * (a border of three pixels, rows 0-3 and every odd row at half resolution). */
static uint8_t satu8(int32_t x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

void oracle_elas_descriptor(const uint8_t* img, int32_t width, int32_t height, int32_t stride, int32_t half_resolution,
                            uint8_t* desc) {
  const int32_t w = width + 15 - (width - 1) % 16, h = height;   /* bpl, elas.cpp:41 */
  const size_t n = (size_t)w * h;
  uint8_t* I = (uint8_t*)calloc(n, 1);
  int16_t* tv = (int16_t*)calloc(n, sizeof(int16_t));
  int16_t* th = (int16_t*)calloc(n, sizeof(int16_t));
  uint8_t* du = (uint8_t*)calloc(n, 1);
  uint8_t* dv = (uint8_t*)calloc(n, 1);
  for (int32_t v = 0; v < h; ++v) memcpy(I + (size_t)v * w, img + (size_t)v * stride, (size_t)width);
  /* convolve_cols_3x3 (filter.cpp:374-407): out_v = (1, 2, 1) down the column, out_h = (1, 0, -1) */
  for (int32_t r = 1; r + 1 < h; ++r)
    for (int32_t c = 0; c < w; ++c) {
      const int32_t a = I[(size_t)(r - 1) * w + c], b = I[(size_t)r * w + c], d = I[(size_t)(r + 1) * w + c];
      tv[(size_t)r * w + c] = (int16_t)(a + 2 * b + d);
      th[(size_t)r * w + c] = (int16_t)(a - d);
    }
  const size_t blocked = (n - 2) / 16 * 16;   /* elements the vector loops of the row filters cover */
  for (size_t j = 0; j < blocked; ++j) {
    du[j + 1] = satu8((((int16_t)(tv[j] - tv[j + 2])) >> 2) + 128);
    dv[j + 1] = satu8((((int16_t)(th[j] + 2 * th[j + 1] + th[j + 2])) >> 2) + 128);
  }
  for (size_t j = blocked; j + 2 < n; ++j) du[j + 1] = (uint8_t)(((tv[j] - tv[j + 2]) >> 2) + 128);   /* the scalar tail of the 101 filter */
  memset(desc, 0, (size_t)16 * width * height);
  for (int32_t v = half_resolution ? 4 : 3; v < height - 3; v += half_resolution ? 2 : 1)
    for (int32_t u = 3; u < width - 3; ++u) {
      uint8_t* o = desc + ((size_t)v * width + u) * 16;
      const uint8_t *u0 = du + (size_t)(v - 2) * w + u, *u1 = du + (size_t)(v - 1) * w + u, *u2 = du + (size_t)v * w + u,
                    *u3 = du + (size_t)(v + 1) * w + u, *u4 = du + (size_t)(v + 2) * w + u;
      const uint8_t *v1 = dv + (size_t)(v - 1) * w + u, *v2 = dv + (size_t)v * w + u, *v3 = dv + (size_t)(v + 1) * w + u;
      o[0] = u0[0]; o[1] = u1[-2]; o[2] = u1[0]; o[3] = u1[2]; o[4] = u2[-1]; o[5] = u2[0]; o[6] = u2[0]; o[7] = u2[1];
      o[8] = u3[-2]; o[9] = u3[0]; o[10] = u3[2]; o[11] = u4[0]; o[12] = v1[0]; o[13] = v2[-1]; o[14] = v2[1]; o[15] = v3[0];
    }
  free(I); free(tv); free(th); free(du); free(dv);
}
