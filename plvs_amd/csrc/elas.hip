// libelas on the device.  First the two methods the reference's own accelerated build overrides —
//   ElasGPU::computeDisparity, ElasGPU::adaptiveMean   (Thirdparty/libelas-gpu/GPU/elas_gpu.h:41-45),
// i.e. Elas::computeDisparity (CPU/elas.cpp:840-968 with findMatch :739-837) and Elas::adaptiveMean (:1349-1572) — then
// the other virtual stages between the descriptors and the result: the candidate loop of computeSupportMatches (:434-456
// with computeMatchingDisparity :296-410), leftRightConsistencyCheck (:971-1040), removeSmallSegments (:1043-1160),
// gapInterpolation (:1163-1347).  All reached from PointCloudKeyFrame::ProcessStereoLibelas (src/PointCloudKeyFrame.cc:335-432)
// through libelas::ElasInterface::process -> Elas::process (elas.cpp:36-159).  Descriptors, the support filters, the Delaunay
// triangulation, planes and grid stay the caller's host code.  Every entry point takes the host pointers of the method it
// replaces and returns results bit-identical to it.
//
// computeDisparity rasterises the triangles one after the other and lets later ones overwrite earlier ones on shared
// pixels; whether a pixel is written at all depends on the pixel alone (its column and texture), the value on the plane
// of the LAST triangle that covers it.  So: (1) elas_owner — a wave per triangle walks the triangle's columns exactly as
// the reference does (same float expressions, same float -> integer conversions) and leaves max(triangle index) per
// pixel; (2) elas_match — a thread per pixel runs findMatch against its owner's plane: the grid candidates, then the
// plane's disparity range with the prior, 16-byte descriptor SADs (v_sad_u8).  adaptiveMean's two filter passes have
// a fixed window per output pixel: a thread per pixel, the window's values summed in the order the reference's
// four-lane registers impose (slot = pixel index mod 4 / 8).  The other stages are described at their kernels.
#include <cmath>
#include <vector>

#include "common.hpp"

namespace {

using plvs::ceil_div;

struct Support { int32_t u, v, d; };                                                   // Elas::support_pt, elas.h:178-183
struct Triangle { int32_t c1, c2, c3; float t1a, t1b, t1c, t2a, t2b, t2c; };           // Elas::triangle, elas.h:185-190
static_assert(sizeof(Support) == 12 && sizeof(Triangle) == 36, "record layouts of the C ABI");

struct MatchParams {
  int32_t width, height, subsampling, right_image;
  int32_t grid_size, match_texture, plane_radius, disp_num;
  int32_t grid_w, grid_stride;   // grid_dims[1], grid_dims[0]
};

// (uint32_t)(float) stored into an int32_t (elas.cpp:927-928, :947-948) as x86-64 evaluates it: the 64-bit truncation's
// low word — a negative product comes back as the negative integer, not as 0
__device__ __forceinline__ int32_t trunc_u32_as_i32(float x) { return (int32_t)(uint32_t)(long long)x; }

// the triangle as computeDisparity sees it: sorted corners, the three edge lines, the plane and its validity
struct TriSetup {
  float A_u, B_u, C_u;
  float AB_a, AB_b, AC_a, AC_b, BC_a, BC_b;
  float plane_a, plane_b, plane_c;
  bool valid;
};

__device__ __forceinline__ TriSetup tri_setup(const Triangle& t, const Support* __restrict__ sup, bool right_image) {
  TriSetup s;
  float plane_d;
  if (!right_image) { s.plane_a = t.t1a; s.plane_b = t.t1b; s.plane_c = t.t1c; plane_d = t.t2a; }
  else              { s.plane_a = t.t2a; s.plane_b = t.t2b; s.plane_c = t.t2c; plane_d = t.t1a; }
  const Support p1 = sup[t.c1], p2 = sup[t.c2], p3 = sup[t.c3];
  float u0 = right_image ? (float)(p1.u - p1.d) : (float)p1.u, v0 = (float)p1.v;
  float u1 = right_image ? (float)(p2.u - p2.d) : (float)p2.u, v1 = (float)p2.v;
  float u2 = right_image ? (float)(p3.u - p3.d) : (float)p3.u, v2 = (float)p3.v;
  // (the reference's exchange sort over j = 0..2, k < j: compares (0,1), then (0,2), then (1,2))
#define SWAP_IF(ua, va, ub, vb) if (ua > ub) { const float tu = ub; ub = ua; ua = tu; const float tv = vb; vb = va; va = tv; }
  SWAP_IF(u0, v0, u1, v1)
  SWAP_IF(u0, v0, u2, v2)
  SWAP_IF(u1, v1, u2, v2)
#undef SWAP_IF
  s.A_u = u0; s.B_u = u1; s.C_u = u2;
  s.AB_a = 0.f; s.AC_a = 0.f; s.BC_a = 0.f;
  if ((int32_t)u0 != (int32_t)u1) s.AB_a = (v0 - v1) / (u0 - u1);
  if ((int32_t)u0 != (int32_t)u2) s.AC_a = (v0 - v2) / (u0 - u2);
  if ((int32_t)u1 != (int32_t)u2) s.BC_a = (v1 - v2) / (u1 - u2);
  s.AB_b = v0 - s.AB_a * u0;
  s.AC_b = v0 - s.AC_a * u0;
  s.BC_b = v1 - s.BC_a * u1;
  s.valid = fabs((double)s.plane_a) < 0.7 && fabs((double)plane_d) < 0.7;
  return s;
}

// (1) the last triangle over every pixel: owner[pixel] = index + 1.  One wave per triangle, a lane per column.
__global__ __launch_bounds__(64) void elas_owner(MatchParams P, const Triangle* __restrict__ tri, int ntri,
                                                 const Support* __restrict__ sup, uint32_t* __restrict__ owner) {
  const int i = blockIdx.x;
  if (i >= ntri) return;
  const TriSetup s = tri_setup(tri[i], sup, P.right_image != 0);
  const int ow = P.subsampling ? P.width / 2 : P.width;
  for (int part = 0; part < 2; ++part) {
    const float lo_u = part ? s.B_u : s.A_u, hi_u = part ? s.C_u : s.B_u;
    const float e_a = part ? s.BC_a : s.AB_a, e_b = part ? s.BC_b : s.AB_b;
    if ((int32_t)lo_u == (int32_t)hi_u) continue;
    const int u_begin = max((int32_t)lo_u, 0), u_end = min((int32_t)hi_u, P.width);
    for (int u = u_begin + (int)threadIdx.x; u < u_end; u += 64) {
      if (P.subsampling && (u % 2) != 0) continue;
      if (u < 2 || u >= P.width - 2) continue;   // (findMatch returns before it writes: elas.cpp:753-754)
      const int32_t v_1 = trunc_u32_as_i32(s.AC_a * (float)u + s.AC_b);
      const int32_t v_2 = trunc_u32_as_i32(e_a * (float)u + e_b);
      // (rows outside the image would be writes outside D in the reference: its triangles never produce them)
      // (with subsampling the owner map has height / 2 rows: an odd height's last row has none)
      const int v_begin = max(min(v_1, v_2), 0), v_end = min(max(v_1, v_2), P.subsampling ? 2 * (P.height / 2) : P.height);
      for (int v = v_begin; v < v_end; ++v)
        if (!P.subsampling || (v % 2) == 0) {
          const size_t at = P.subsampling ? (size_t)(v / 2) * ow + u / 2 : (size_t)v * ow + u;
          atomicMax(&owner[at], (uint32_t)i + 1u);
        }
    }
  }
}

__device__ __forceinline__ int32_t sad16(const uint4 a, const uint4 b) {
  uint32_t s = __builtin_amdgcn_sad_u8(a.x, b.x, 0u);
  s = __builtin_amdgcn_sad_u8(a.y, b.y, s);
  s = __builtin_amdgcn_sad_u8(a.z, b.z, s);
  return (int32_t)__builtin_amdgcn_sad_u8(a.w, b.w, s);
}

// (2) Elas::findMatch for every pixel some triangle covers
__global__ __launch_bounds__(256) void elas_match(MatchParams P, const Triangle* __restrict__ tri, const Support* __restrict__ sup,
                                                  const uint32_t* __restrict__ owner, const int32_t* __restrict__ grid,
                                                  const int32_t* __restrict__ prior, const uint4* __restrict__ desc1,
                                                  const uint4* __restrict__ desc2, float* __restrict__ D) {
  const int ow = P.subsampling ? P.width / 2 : P.width, oh = P.subsampling ? P.height / 2 : P.height;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= ow || y >= oh) return;
  const size_t at = (size_t)y * ow + x;
  float out = -10.0f;   // (computeDisparity's initial value: no triangle, a column at the border, too little texture)
  const uint32_t own = owner[at];
  const int u = P.subsampling ? 2 * x : x, v = P.subsampling ? 2 * y : y;
  const int window_size = 2;
  if (own != 0u && !(u < window_size || u >= P.width - window_size)) {
    const TriSetup s = tri_setup(tri[own - 1u], sup, P.right_image != 0);
    const size_t line = (size_t)P.width * (size_t)max(min(v, P.height - 3), 2);
    const uint4* I1_line = (P.right_image ? desc2 : desc1) + line;
    const uint4* I2_line = (P.right_image ? desc1 : desc2) + line;
    const uint4 block = I1_line[u];
    const int32_t texture = sad16(block, make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u));
    if (texture >= P.match_texture) {
      const int32_t d_plane = (int32_t)(s.plane_a * (float)u + s.plane_b * (float)v + s.plane_c);
      const int32_t d_plane_min = max(d_plane - P.plane_radius, 0);
      const int32_t d_plane_max = min(d_plane + P.plane_radius, P.disp_num - 1);
      const int32_t grid_x = (int32_t)floorf((float)u / (float)P.grid_size);
      const int32_t grid_y = (int32_t)floorf((float)v / (float)P.grid_size);
      const int32_t* cell = grid + (size_t)(grid_y * P.grid_w + grid_x) * (size_t)P.grid_stride;
      const int32_t num_grid = min(cell[0], P.grid_stride - 1);   // (createGrid never writes more; a corrupt count stays inside the cell)
      int32_t min_val = 10000, min_d = -1;
      const int32_t sign = P.right_image ? 1 : -1;
      for (int32_t k = 0; k < num_grid; ++k) {
        const int32_t d_curr = cell[1 + k];
        if (d_curr < d_plane_min || d_curr > d_plane_max) {
          const int32_t u_warp = u + sign * d_curr;
          if (u_warp < window_size || u_warp >= P.width - window_size) continue;
          const int32_t val = sad16(block, I2_line[u_warp]);
          if (val < min_val) { min_val = val; min_d = d_curr; }
        }
      }
      for (int32_t d_curr = d_plane_min; d_curr <= d_plane_max; ++d_curr) {
        const int32_t u_warp = u + sign * d_curr;
        if (u_warp < window_size || u_warp >= P.width - window_size) continue;
        const int32_t val = sad16(block, I2_line[u_warp]) + (s.valid ? prior[abs(d_curr - d_plane)] : 0);
        if (val < min_val) { min_val = val; min_d = d_curr; }
      }
      out = min_d >= 0 ? (float)min_d : -1.0f;
    }
  }
  D[at] = out;
}

// ------------------------------------------------------------------ the descriptor images
// libelas::Descriptor (descriptor.cpp:30-131) on the zero-padded image buffer Elas::process builds (line length bpl = the
// width rounded up to 16, elas.cpp:41-57).  filter::sobel3x3 (filter.cpp:410-418) treats that buffer as ONE flat array: a
// (1,2,1) / (1,0,-1) pass down the columns of rows 1 .. h-2 into 16-bit temporaries, then (1,0,-1) / (1,2,1) across the
// flat array — over the line ends — stored one element further on, saturated to a byte after >> 2 and + 128; the last
// elements go through the 101 filter's unsaturated scalar tail, the 121 filter has none.  What the reference never writes
// reads as zero (fresh pages).
// (the padded buffer is not materialised: the image is read as the caller laid it out, columns beyond the width are 0)
struct PaddedImage {
  const uint8_t* raw;
  int width, stride, w, h;   // w = bpl
  __device__ __forceinline__ int32_t at(int r, int c) const { return c < width ? (int32_t)raw[(size_t)r * stride + c] : 0; }
};
__device__ __forceinline__ int32_t sobel_tv(const PaddedImage& I, uint32_t q) {
  const int r = (int)(q / (uint32_t)I.w), c = (int)(q - (uint32_t)r * (uint32_t)I.w);
  return (r >= 1 && r + 1 < I.h) ? I.at(r - 1, c) + 2 * I.at(r, c) + I.at(r + 1, c) : 0;
}
__device__ __forceinline__ int32_t sobel_th(const PaddedImage& I, uint32_t q) {
  const int r = (int)(q / (uint32_t)I.w), c = (int)(q - (uint32_t)r * (uint32_t)I.w);
  return (r >= 1 && r + 1 < I.h) ? I.at(r - 1, c) - I.at(r + 1, c) : 0;
}
__device__ __forceinline__ uint8_t satu8(int32_t x) { return (uint8_t)min(max(x, 0), 255); }

__global__ __launch_bounds__(256) void elas_sobel(PaddedImage I, uint8_t* __restrict__ du, uint8_t* __restrict__ dv) {
  const uint32_t n = (uint32_t)I.w * (uint32_t)I.h, p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t blocked = (n - 2u) / 16u * 16u;   // the elements the row filters' vector loops cover
  uint8_t u = 0, v = 0;
  if (p >= 1u) {
    const uint32_t j = p - 1u;
    if (j < blocked) {
      u = satu8(((int32_t)(int16_t)(sobel_tv(I, j) - sobel_tv(I, j + 2u)) >> 2) + 128);
      v = satu8(((int32_t)(int16_t)(sobel_th(I, j) + 2 * sobel_th(I, j + 1u) + sobel_th(I, j + 2u)) >> 2) + 128);
    } else if (j + 2u < n) {
      u = (uint8_t)(((sobel_tv(I, j) - sobel_tv(I, j + 2u)) >> 2) + 128);
    }
  }
  du[p] = u;
  dv[p] = v;
}

// the 16 samples of a pixel (descriptor.cpp:75-92 / :112-129): twelve of du on a rhombus, four of dv on a cross
__global__ __launch_bounds__(256) void elas_describe(const uint8_t* __restrict__ du, const uint8_t* __restrict__ dv, int w, int width,
                                                     int height, int half_resolution, uint4* __restrict__ desc) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y;
  if (u >= width || v >= height) return;
  uint4 out = make_uint4(0u, 0u, 0u, 0u);
  const bool row_ok = half_resolution ? (v >= 4 && v < height - 3 && (v % 2) == 0) : (v >= 3 && v < height - 3);
  if (row_ok && u >= 3 && u < width - 3) {
    const uint8_t *u0 = du + (size_t)(v - 2) * w + u, *u1 = du + (size_t)(v - 1) * w + u, *u2 = du + (size_t)v * w + u,
                  *u3 = du + (size_t)(v + 1) * w + u, *u4 = du + (size_t)(v + 2) * w + u;
    const uint8_t *v1 = dv + (size_t)(v - 1) * w + u, *v2 = dv + (size_t)v * w + u, *v3 = dv + (size_t)(v + 1) * w + u;
#define B4(a, b, c, d) ((uint32_t)(a) | ((uint32_t)(b) << 8) | ((uint32_t)(c) << 16) | ((uint32_t)(d) << 24))
    out.x = B4(u0[0], u1[-2], u1[0], u1[2]);
    out.y = B4(u2[-1], u2[0], u2[0], u2[1]);
    out.z = B4(u3[-2], u3[0], u3[2], u4[0]);
    out.w = B4(v1[0], v2[-1], v2[1], v3[0]);
#undef B4
  }
  desc[(size_t)v * width + u] = out;
}

// ------------------------------------------------------------------ the candidate grid of computeSupportMatches
// Elas::computeMatchingDisparity (elas.cpp:296-410) for one grid point: the energy of a disparity is the SAD of four
// descriptors around the point (two columns / two rows away); the best disparity must beat the second smallest energy by
// the support ratio.  A wave per candidate, a lane per disparity: the sequential "best and second best" of the reference
// is the lexicographic minimum of (energy, disparity) and the second smallest energy of all — both order-free.
struct SupportParams {
  int32_t width, height, step, can_w, can_h;
  int32_t disp_min, disp_max, support_texture, lr_threshold;
  float support_threshold;
};

__device__ __forceinline__ int matching_disparity(const SupportParams& P, int u, int v, const uint4* __restrict__ desc1,
                                                  const uint4* __restrict__ desc2, bool right_image, int lane) {
  const int u_step = 2, v_step = 2, window_size = 3;
  if (!(u >= window_size + u_step && u <= P.width - window_size - 1 - u_step && v >= window_size + v_step &&
        v <= P.height - window_size - 1 - v_step))
    return -1;
  const uint4* I1_line = (right_image ? desc2 : desc1) + (size_t)P.width * v;
  const uint4* I2_line = (right_image ? desc1 : desc2) + (size_t)P.width * v;
  const int32_t texture = sad16(I1_line[u], make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u));
  if (texture < P.support_texture) return -1;
  const int disp_min_valid = max(P.disp_min, 0);
  const int disp_max_valid = right_image ? min(P.disp_max, P.width - u - window_size - u_step)
                                         : min(P.disp_max, u - window_size - u_step);
  if (disp_max_valid - disp_min_valid < 10) return -1;
  const ptrdiff_t row = (ptrdiff_t)P.width * v_step;
  const uint4 b0 = I1_line[u - u_step - row], b1 = I1_line[u + u_step - row], b2 = I1_line[u - u_step + row],
              b3 = I1_line[u + u_step + row];
  int32_t e1 = 32767, d1 = 0x7FFFFFFF, e2 = 32767;   // this lane's best (energy, disparity) and second smallest energy
  for (int d = disp_min_valid + lane; d <= disp_max_valid; d += 64) {
    const int u_warp = right_image ? u + d : u - d;
    const int32_t sum = sad16(b0, I2_line[u_warp - u_step - row]) + sad16(b1, I2_line[u_warp + u_step - row]) +
                        sad16(b2, I2_line[u_warp - u_step + row]) + sad16(b3, I2_line[u_warp + u_step + row]);
    if (sum < e1) {          // (a lane meets its disparities in ascending order: the reference's own update)
      e2 = e1;
      e1 = sum;
      d1 = d;
    } else if (sum < e2) {
      e2 = sum;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int32_t oe1 = __shfl_xor(e1, off), od1 = __shfl_xor(d1, off), oe2 = __shfl_xor(e2, off);
    const bool other_wins = oe1 < e1 || (oe1 == e1 && od1 < d1);
    const int32_t loser = other_wins ? e1 : oe1;
    e2 = min(min(e2, oe2), loser);
    e1 = other_wins ? oe1 : e1;
    d1 = other_wins ? od1 : d1;
  }
  // (at least eleven disparities were tried: a second best always exists)
  return ((float)e1 < P.support_threshold * (float)e2) ? d1 : -1;
}

__global__ __launch_bounds__(64) void elas_support_candidates(SupportParams P, const uint4* __restrict__ desc1,
                                                              const uint4* __restrict__ desc2, int16_t* __restrict__ D_can) {
  const int u_can = 1 + (int)blockIdx.x, v_can = 1 + (int)blockIdx.y, lane = threadIdx.x;
  if (u_can >= P.can_w || v_can >= P.can_h) return;
  const int u = u_can * P.step, v = v_can * P.step;
  int out = -1;
  const int d = matching_disparity(P, u, v, desc1, desc2, false, lane);   // (wave-uniform)
  if (d >= 0) {
    const int d2 = matching_disparity(P, u - d, v, desc1, desc2, true, lane);
    if (d2 >= 0 && abs(d - d2) <= P.lr_threshold) out = d;
  }
  if (lane == 0) D_can[(size_t)v_can * P.can_w + u_can] = (int16_t)out;
}

// ------------------------------------------------------------------ adaptiveMean
// one output of either pass (elas.cpp:1389-1411, :1467-1503): `val` by register slot; the "absolute value" of the
// subsampling branch is the reference's and-mask with the FLOAT 2147483648.0f = 0x4F000000 (elas.cpp:1379)
template <int kTaps>
__device__ __forceinline__ bool mean_of_window(const float* val, float centre, float* out) {
  float weight[kTaps], factor[kTaps];
#pragma unroll
  for (int s = 0; s < kTaps; ++s) {
    float w = val[s] - centre;
    if (kTaps == 4) {
      w = __uint_as_float(__float_as_uint(w) & 0x4F000000u);
    } else {
      const float neg = 0.0f - w;
      w = (neg > w) ? neg : w;
    }
    w = 4.0f - w;
    w = (0.0f > w) ? 0.0f : w;
    weight[s] = w;
    factor[s] = val[s] * w;
  }
  if (kTaps == 8) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      weight[s] = weight[s] + weight[s + 4];
      factor[s] = factor[s] + factor[s + 4];
    }
  }
  const float weight_sum = weight[0] + weight[1] + weight[2] + weight[3];
  const float factor_sum = factor[0] + factor[1] + factor[2] + factor[3];
  if (weight_sum > 0) {
    const float d = factor_sum / weight_sum;
    if (d >= 0) {
      *out = d;
      return true;
    }
  }
  return false;
}

// D -> D_copy (negatives -> -10) and the initial D_tmp (-10 there, 0 elsewhere: the reference leaves D_tmp's valid pixels
// unwritten until the horizontal pass; where that pass never writes, the vertical one reads what malloc returned —
// zero pages in a fresh process, the value the parity tests pin the compiled reference to)
__global__ void mean_prepare(const float* __restrict__ D, size_t n, float* __restrict__ D_copy, float* __restrict__ D_tmp) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float d = D[i];
  D_copy[i] = d < 0 ? -10.0f : d;
  D_tmp[i] = d < 0 ? -10.0f : 0.0f;
}

// kVertical = false: D_copy -> D_tmp along rows; true: D_tmp -> D along columns.  The window ends `back` pixels behind
// its last pixel's output: window = [c + back - kTaps + 1, c + back] for the output at c.
template <int kTaps, bool kVertical>
__global__ __launch_bounds__(256) void mean_pass(const float* __restrict__ src, float* __restrict__ dst, int W, int H) {
  constexpr int back = kTaps == 4 ? 1 : 3;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W || y >= H) return;
  const int c = kVertical ? y : x, len = kVertical ? H : W;
  // the other coordinate's range (elas.cpp:1383 / :1413: rows 3 .. H-4 for the horizontal pass, columns 3 .. W-4 for the vertical)
  if (kVertical ? (x < 3 || x >= W - 3) : (y < 3 || y >= H - 3)) return;
  const int last = c + back;                       // the loop index (u or v) at which this output is produced
  if (last < kTaps - 1 || last >= len) return;
  float val[kTaps];
#pragma unroll
  for (int k = 0; k < kTaps; ++k) {
    const int i = last - k;                        // a pixel of the window ...
    const float vk = kVertical ? src[(size_t)i * W + x] : src[(size_t)y * W + i];
    // ... sits in slot i mod kTaps; write through a select chain (kTaps is 4 or 8: no dynamic indexing)
#pragma unroll
    for (int s = 0; s < kTaps; ++s)
      if ((i % kTaps) == s) val[s] = vk;
  }
  const float centre = kVertical ? src[(size_t)c * W + x] : src[(size_t)y * W + c];
  float d;
  if (mean_of_window<kTaps>(val, centre, &d)) dst[kVertical ? (size_t)c * W + x : (size_t)y * W + c] = d;
}

// ------------------------------------------------------------------ the post-processing between computeDisparity and
// adaptiveMean (elas.cpp:971-1347); W x H is the disparity map's size
// Elas::leftRightConsistencyCheck: every pixel of either map against the other map at its warped column
__global__ __launch_bounds__(256) void lr_check(const float* __restrict__ in1, const float* __restrict__ in2, float* __restrict__ out1,
                                                float* __restrict__ out2, int W, int H, int subsampling, int lr_threshold) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y;
  if (u >= W || v >= H) return;
  const size_t addr = (size_t)v * W + u;
  const float d1 = in1[addr], d2 = in2[addr];
  const float u_warp_1 = subsampling ? (float)u - d1 / 2 : (float)u - d1;
  const float u_warp_2 = subsampling ? (float)u + d2 / 2 : (float)u + d2;
  float o1 = -10.0f, o2 = -10.0f;
  if (d1 >= 0 && u_warp_1 >= 0 && u_warp_1 < W && !(fabsf(in2[(size_t)v * W + (int32_t)u_warp_1] - d1) > (float)lr_threshold)) o1 = d1;
  if (d2 >= 0 && u_warp_2 >= 0 && u_warp_2 < W && !(fabsf(in1[(size_t)v * W + (int32_t)u_warp_2] - d2) > (float)lr_threshold)) o2 = d2;
  out1[addr] = o1;
  out2[addr] = o2;
}

// Elas::removeSmallSegments: the reference grows 4-connected segments of valid pixels whose neighbouring disparities differ
// by at most the similarity threshold and invalidates the small ones.  With every invalid pixel at -10 (as the left/right
// check leaves them) and a threshold below 10 the segments are the connected components of a symmetric relation: a
// union-find over the right / lower neighbour pairs gives the same partition whatever the order.
__device__ __forceinline__ uint32_t seg_find(const uint32_t* parent, uint32_t x) {
  uint32_t p = parent[x];
  while (p != x) {
    x = p;
    p = parent[x];
  }
  return x;
}
__device__ __forceinline__ void seg_union(uint32_t* parent, uint32_t a, uint32_t b) {
  for (;;) {
    a = seg_find(parent, a);
    b = seg_find(parent, b);
    if (a == b) return;
    if (a < b) { const uint32_t t = a; a = b; b = t; }
    const uint32_t old = atomicMin(&parent[a], b);   // (a was a root: hang it under the smaller root)
    if (old == a) return;
    a = old;                                         // (somebody moved it meanwhile: go on from there)
  }
}
// The horizontal part of the relation needs no atomics: a workgroup per row finds the start of every pixel's run of
// mutually similar valid neighbours (a max-scan of the break positions) and makes it the pixel's parent.  Only the
// vertical edges are left for the union-find, and of those only the first of each stretch in which the pair above and
// the pair below both continue their runs (the others would join the same two runs again).
__device__ __forceinline__ bool seg_similar(float a, float b, float sim) { return a >= 0 && b >= 0 && fabsf(a - b) <= sim; }

__global__ __launch_bounds__(256) void seg_rows(const float* __restrict__ D, uint32_t* __restrict__ parent, uint32_t* __restrict__ size,
                                                uint32_t* __restrict__ run_len, int W, float sim) {
  __shared__ int s_part[256];
  const int v = blockIdx.x, tid = threadIdx.x;
  const float* row = D + (size_t)v * W;
  const int per = (W + 255) / 256, begin = tid * per, end = min(begin + per, W);
  // the last break (a pixel that does not continue its left neighbour's run) at or before each pixel
  int last = -1;
  for (int u = begin; u < end; ++u)
    if (u == 0 || !seg_similar(row[u - 1], row[u], sim)) last = u;
  s_part[tid] = last;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {      // inclusive max-scan of the segments' last breaks
    const int mine = s_part[tid], other = tid >= off ? s_part[tid - off] : -1;
    __syncthreads();
    s_part[tid] = max(mine, other);
    __syncthreads();
  }
  int start = tid > 0 ? s_part[tid - 1] : -1;
  for (int u = begin; u < end; ++u) {
    if (u == 0 || !seg_similar(row[u - 1], row[u], sim)) start = u;
    parent[(size_t)v * W + u] = (uint32_t)((size_t)v * W + start);
    size[(size_t)v * W + u] = 0u;
    if (row[u] >= 0) atomicAdd(&run_len[(size_t)v * W + start], 1u);   // (run_len arrives zeroed; non-zero at run starts only)
  }
}

__global__ __launch_bounds__(256) void seg_link(const float* __restrict__ D, uint32_t* __restrict__ parent, int W, int H, float sim) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y;
  if (u >= W || v + 1 >= H) return;
  const size_t at = (size_t)v * W + u;
  const float d = D[at], b = D[at + W];
  if (!seg_similar(d, b, sim)) return;
  if (u > 0) {   // the pair to the left joins the same two runs: leave it to that one
    const float dl = D[at - 1], bl = D[at + W - 1];
    if (seg_similar(dl, bl, sim) && seg_similar(dl, d, sim) && seg_similar(bl, b, sim)) return;
  }
  seg_union(parent, (uint32_t)at, (uint32_t)(at + W));
}
// the unions leave chains as long as a segment is tall (a run's root hangs under the root of a run above it): pointer
// jumping halves them per round
__global__ void seg_jump(uint32_t* __restrict__ parent, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) parent[i] = parent[parent[i]];
}
// a segment's size: one addition per RUN onto its root (a pixel each would be hundreds of thousands of atomics on the
// one word of a large segment: 4 ms)
__global__ void seg_count(uint32_t* __restrict__ parent, const uint32_t* __restrict__ run_len, uint32_t* __restrict__ size, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t root = 0u;
  if (i < n) {
    root = seg_find(parent, (uint32_t)i);
    parent[i] = root;   // (only shortens this pixel's own path: the roots are final after seg_link)
  }
  // the runs of a wave mostly hang under a few roots: lanes with the same root add up first, one of them goes to memory
  uint32_t len = (i < n) ? run_len[i] : 0u;
  while (__any(len != 0u)) {
    const bool active = len != 0u;
    const unsigned long long act = __ballot(active);
    const int leader = __ffsll((long long)act) - 1;
    const uint32_t lead_root = (uint32_t)__shfl((int)root, leader);
    const bool same = active && root == lead_root;
    uint32_t sum = same ? len : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += (uint32_t)__shfl_xor((int)sum, off);
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&size[lead_root], sum);
    if (same) len = 0u;
  }
}
__global__ void seg_apply(float* __restrict__ D, const uint32_t* __restrict__ parent, const uint32_t* __restrict__ size, size_t n,
                          uint32_t speckle) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t count = (D[i] >= 0) ? size[parent[i]] : 1u;   // (an invalid pixel is a segment of its own)
  if (count < speckle) D[i] = -10.0f;
}

// Elas::gapInterpolation: one line (a row, then a column) at a time, exactly as the reference walks it
__device__ void gap_line(float* D, int len, size_t stride, int gap_width, bool add_corners) {
  const float discon_threshold = 3.0f;
  int count = 0;
  for (int i = 0; i < len; ++i) {
    if (D[(size_t)i * stride] >= 0) {
      if (count >= 1 && count <= gap_width) {
        const int first = i - count, last = i - 1;
        if (first > 0 && last < len - 1) {
          const float d1 = D[(size_t)(first - 1) * stride], d2 = D[(size_t)(last + 1) * stride];
          const float d_ipol = (fabsf(d1 - d2) < discon_threshold) ? (d1 + d2) / 2 : ((d2 < d1) ? d2 : d1);
          for (int k = first; k <= last; ++k) D[(size_t)k * stride] = d_ipol;
        }
      }
      count = 0;
    } else {
      ++count;
    }
  }
  if (add_corners) {
    for (int i = 0; i < len; ++i)
      if (D[(size_t)i * stride] >= 0) {
        const float d = D[(size_t)i * stride];
        for (int k = max(i - gap_width, 0); k < i; ++k) D[(size_t)k * stride] = d;
        break;
      }
    for (int i = len - 1; i >= 0; --i)
      if (D[(size_t)i * stride] >= 0) {
        const float d = D[(size_t)i * stride];
        for (int k = i; k <= min(i + gap_width, len - 1); ++k) D[(size_t)k * stride] = d;
        break;
      }
  }
}
__global__ void gap_rows(float* __restrict__ D, int W, int H, int gap_width, int add_corners) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < H) gap_line(D + (size_t)v * W, W, 1, gap_width, add_corners != 0);
}
__global__ void gap_cols(float* __restrict__ D, int W, int H, int gap_width, int add_corners) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u < W) gap_line(D + u, H, (size_t)W, gap_width, add_corners != 0);
}
// Narrow gaps without add_corners (PLVS's setting: three pixels, two with subsampling): a pixel per thread.  An invalid
// pixel is filled exactly when valid pixels lie within gap_width steps on both sides of it along the line and the run of
// invalid pixels between them is no longer than gap_width — the line walk above reaches the same decision and the same
// two end values.  Out of place (the pass reads validity next to the pixels it fills).
template <bool kCols>
__global__ __launch_bounds__(256) void gap_pass(const float* __restrict__ src, float* __restrict__ dst, int W, int H, int gap_width) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y;
  if (u >= W || v >= H) return;
  const size_t at = (size_t)v * W + u;
  const int i = kCols ? v : u, len = kCols ? H : W;
  const size_t stride = kCols ? (size_t)W : 1;
  const float* line = src + (kCols ? (size_t)u : (size_t)v * W);
  float out = src[at];
  if (!(out >= 0)) {
    int a = 0, b = 0;   // steps to the nearest valid pixel before / after (0: none within gap_width)
    for (int k = 1; k <= gap_width && i - k >= 0; ++k)
      if (line[(size_t)(i - k) * stride] >= 0) { a = k; break; }
    for (int k = 1; k <= gap_width && i + k < len; ++k)
      if (line[(size_t)(i + k) * stride] >= 0) { b = k; break; }
    if (a > 0 && b > 0 && a + b - 1 <= gap_width) {
      const float d1 = line[(size_t)(i - a) * stride], d2 = line[(size_t)(i + b) * stride];
      out = (fabsf(d1 - d2) < 3.0f) ? (d1 + d2) / 2 : ((d2 < d1) ? d2 : d1);
    }
  }
  dst[at] = out;
}

}  // namespace

struct plvs_elas {
  plvs_elas_params prm;
  hipStream_t stream = nullptr;
  plvs::DevBuf<uint8_t> desc1, desc2;
  plvs::DevBuf<Support> support;
  plvs::DevBuf<Triangle> tri;
  plvs::DevBuf<int32_t> grid, prior;
  plvs::DevBuf<uint32_t> owner;
  plvs::DevBuf<float> D, D_copy, D_tmp;
  plvs::DevBuf<int16_t> D_can;
  plvs::DevBuf<uint32_t> seg_parent, seg_size, seg_run;
  plvs::DevBuf<uint8_t> img, sob_u, sob_v;
  int desc_width = 0, desc_height = 0;   // the staged descriptor images' size (0: none)
  // the maps of the last computeDisparity calls, kept in HBM for plvs_hip_elas_postprocess ([0] left, [1] right)
  plvs::DevBuf<float> res[2];
  int res_w[2] = {0, 0}, res_h[2] = {0, 0};
};

namespace {
// ---- the post-processing stages on a map in HBM (in place), asynchronous on the handle's stream
int mean_core(plvs_elas* h, float* d, int W, int H) {
  hipStream_t s = h->stream;
  const bool sub = h->prm.subsampling != 0;
  const size_t n = (size_t)W * H;
  PLVS_HIP_TRY(h->D_copy.reserve(n));
  PLVS_HIP_TRY(h->D_tmp.reserve(n));
  hipLaunchKernelGGL(mean_prepare, dim3(ceil_div(n, 256)), dim3(256), 0, s, d, n, h->D_copy.p, h->D_tmp.p);
  const dim3 grid(ceil_div((size_t)W, 256), (unsigned)H), block(256);
  if (sub) {
    hipLaunchKernelGGL((mean_pass<4, false>), grid, block, 0, s, h->D_copy.p, h->D_tmp.p, W, H);
    hipLaunchKernelGGL((mean_pass<4, true>), grid, block, 0, s, h->D_tmp.p, d, W, H);
  } else {
    hipLaunchKernelGGL((mean_pass<8, false>), grid, block, 0, s, h->D_copy.p, h->D_tmp.p, W, H);
    hipLaunchKernelGGL((mean_pass<8, true>), grid, block, 0, s, h->D_tmp.p, d, W, H);
  }
  PLVS_KERNEL_CHECK();
  return PLVS_OK;
}
int lr_core(plvs_elas* h, float* d1, float* d2, int W, int H) {
  hipStream_t s = h->stream;
  const size_t n = (size_t)W * H;
  PLVS_HIP_TRY(h->D_tmp.reserve(2 * n));
  hipLaunchKernelGGL(lr_check, dim3(ceil_div((size_t)W, 256), (unsigned)H), dim3(256), 0, s, d1, d2, h->D_tmp.p, h->D_tmp.p + n,
                     W, H, h->prm.subsampling != 0 ? 1 : 0, h->prm.lr_threshold);
  PLVS_KERNEL_CHECK();
  PLVS_HIP_TRY(hipMemcpyAsync(d1, h->D_tmp.p, n * sizeof(float), hipMemcpyDeviceToDevice, s));
  PLVS_HIP_TRY(hipMemcpyAsync(d2, h->D_tmp.p + n, n * sizeof(float), hipMemcpyDeviceToDevice, s));
  return PLVS_OK;
}
int seg_core(plvs_elas* h, float* d, int W, int H) {
  PLVS_REQUIRE(h->prm.speckle_sim_threshold >= 0.0f && h->prm.speckle_sim_threshold < 10.0f,
               "speckle_sim_threshold must stay below the distance of the invalid marker (-10) to a valid disparity");
  hipStream_t s = h->stream;
  const size_t n = (size_t)W * H;
  PLVS_REQUIRE(n < 0xFFFFFFFFull, "image size");
  int32_t speckle = h->prm.speckle_size;
  if (h->prm.subsampling != 0) speckle = (int32_t)(std::sqrt((float)h->prm.speckle_size) * 2);   // elas.cpp:1051
  PLVS_HIP_TRY(h->seg_parent.reserve(n));
  PLVS_HIP_TRY(h->seg_size.reserve(n));
  PLVS_HIP_TRY(h->seg_run.reserve(n));
  PLVS_HIP_TRY(hipMemsetAsync(h->seg_run.p, 0, n * sizeof(uint32_t), s));
  const dim3 grid2(ceil_div((size_t)W, 256), (unsigned)H), block(256);
  hipLaunchKernelGGL(seg_rows, dim3((unsigned)H), block, 0, s, d, h->seg_parent.p, h->seg_size.p, h->seg_run.p, W,
                     h->prm.speckle_sim_threshold);
  hipLaunchKernelGGL(seg_link, grid2, block, 0, s, d, h->seg_parent.p, W, H, h->prm.speckle_sim_threshold);
  for (int span = 1; span < H; span *= 2)
    hipLaunchKernelGGL(seg_jump, dim3(ceil_div(n, 256)), block, 0, s, h->seg_parent.p, n);
  hipLaunchKernelGGL(seg_count, dim3(ceil_div(n, 256)), block, 0, s, h->seg_parent.p, h->seg_run.p, h->seg_size.p, n);
  hipLaunchKernelGGL(seg_apply, dim3(ceil_div(n, 256)), block, 0, s, d, h->seg_parent.p, h->seg_size.p, n,
                     (uint32_t)std::max(speckle, 0));
  PLVS_KERNEL_CHECK();
  return PLVS_OK;
}
int gap_core(plvs_elas* h, float* d, int W, int H) {
  hipStream_t s = h->stream;
  const size_t n = (size_t)W * H;
  const int gap = h->prm.subsampling != 0 ? h->prm.ipol_gap_width / 2 + 1 : h->prm.ipol_gap_width;   // elas.cpp:1172
  if (!h->prm.add_corners && gap <= 64) {
    PLVS_HIP_TRY(h->D_tmp.reserve(n));
    const dim3 grid(ceil_div((size_t)W, 256), (unsigned)H), block(256);
    hipLaunchKernelGGL((gap_pass<false>), grid, block, 0, s, d, h->D_tmp.p, W, H, gap);
    hipLaunchKernelGGL((gap_pass<true>), grid, block, 0, s, h->D_tmp.p, d, W, H, gap);
  } else {   // (MIDDLEBURY: gaps of any width, the corner fill — the reference's walk, a line per thread)
    hipLaunchKernelGGL(gap_rows, dim3(ceil_div((size_t)H, 64)), dim3(64), 0, s, d, W, H, gap, h->prm.add_corners);
    hipLaunchKernelGGL(gap_cols, dim3(ceil_div((size_t)W, 64)), dim3(64), 0, s, d, W, H, gap, h->prm.add_corners);
  }
  PLVS_KERNEL_CHECK();
  return PLVS_OK;
}
// ProcessStereoLibelas' disparity -> depth (src/PointCloudKeyFrame.cc:399-420): with subsampling the depth image is zero
// but for rows m = step * m1, where pixels n = step * n1 and n + 1 take bf / d1; without, depth = bf / d everywhere
__global__ __launch_bounds__(256) void depth_from_disparity(const float* __restrict__ d1, int W1, int H1, float bf, int subsampling,
                                                            int step, float* __restrict__ depth, int W, int H) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y;
  if (u >= W || v >= H) return;
  float out = 0.0f;
  if (!subsampling) {
    out = bf / d1[(size_t)v * W1 + u];
  } else if (v % step == 0 && v / step < H1) {
    // pixel u = n or n + 1 of some n = step * n1 (n1 < W1); with step = 2 every pixel of the row is written once, a larger
    // step leaves the pixels in between at zero and lets a later pair overwrite nothing
    const int n1a = u / step, n1b = (u - 1) / step;
    if (u % step == 0 && n1a < W1) out = bf / d1[(size_t)(v / step) * W1 + n1a];
    else if (u >= 1 && (u - 1) % step == 0 && n1b < W1) out = bf / d1[(size_t)(v / step) * W1 + n1b];
  }
  depth[(size_t)v * W + u] = out;
}
}  // namespace

extern "C" {

int plvs_hip_elas_create(const plvs_elas_params* p, plvs_elas** out) {
  PLVS_REQUIRE(p && out, "null argument");
  PLVS_REQUIRE(p->grid_size > 0 && p->beta != 0.0f && p->sigma > 0.0f, "grid_size, beta, sigma");
  plvs_elas* h = new plvs_elas();
  h->prm = *p;
  hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete h;
    plvs::set_error("hipStreamCreateWithFlags failed: %s", hipGetErrorString(e));
    return PLVS_ERR_HIP;
  }
  *out = h;
  return PLVS_OK;
}

int plvs_hip_elas_destroy(plvs_elas* h) {
  if (!h) return PLVS_OK;
  if (h->stream) (void)hipStreamDestroy(h->stream);
  h->desc1.release(); h->desc2.release(); h->support.release(); h->tri.release(); h->grid.release(); h->prior.release();
  h->owner.release(); h->D.release(); h->D_copy.release(); h->D_tmp.release(); h->D_can.release(); h->seg_parent.release(); h->seg_size.release(); h->seg_run.release(); h->img.release(); h->sob_u.release(); h->sob_v.release();
  h->res[0].release(); h->res[1].release();
  delete h;
  return PLVS_OK;
}

int plvs_hip_elas_set_images(plvs_elas* h, const uint8_t* I1, const uint8_t* I2, int width, int height, int stride) {
  PLVS_REQUIRE(h && I1 && I2, "null argument");
  PLVS_REQUIRE(width >= 16 && height >= 16 && stride >= width, "image size");
  hipStream_t s = h->stream;
  const int bpl = width + 15 - (width - 1) % 16;   // elas.cpp:41
  const size_t n = (size_t)bpl * height, desc_bytes = (size_t)16 * width * height, raw = (size_t)stride * height;
  PLVS_REQUIRE(n < 0x7FFFFFFFull, "image size");
  PLVS_HIP_TRY(h->img.reserve(2 * raw));
  PLVS_HIP_TRY(h->sob_u.reserve(n));
  PLVS_HIP_TRY(h->sob_v.reserve(n));
  PLVS_HIP_TRY(h->desc1.reserve(desc_bytes));
  PLVS_HIP_TRY(h->desc2.reserve(desc_bytes));
  h->desc_width = h->desc_height = 0;
  // (one flat copy per image, lines as the caller laid them out: a pitched copy from pageable memory goes line by line)
  const size_t used = (size_t)stride * (height - 1) + (size_t)width;   // (the last line may end with its last pixel)
  PLVS_HIP_TRY(hipMemcpyAsync(h->img.p, I1, used, hipMemcpyHostToDevice, s));
  PLVS_HIP_TRY(hipMemcpyAsync(h->img.p + raw, I2, used, hipMemcpyHostToDevice, s));
  uint8_t* descs[2] = {h->desc1.p, h->desc2.p};
  for (int k = 0; k < 2; ++k) {
    const PaddedImage img{h->img.p + (size_t)k * raw, width, stride, bpl, height};
    hipLaunchKernelGGL(elas_sobel, dim3(ceil_div(n, 256)), dim3(256), 0, s, img, h->sob_u.p, h->sob_v.p);
    hipLaunchKernelGGL(elas_describe, dim3(ceil_div((size_t)width, 256), (unsigned)height), dim3(256), 0, s, h->sob_u.p,
                       h->sob_v.p, bpl, width, height, h->prm.subsampling ? 1 : 0, reinterpret_cast<uint4*>(descs[k]));
    PLVS_KERNEL_CHECK();
  }
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  h->desc_width = width;
  h->desc_height = height;
  return PLVS_OK;
}

int plvs_hip_elas_download_descriptors(plvs_elas* h, uint8_t* I1_desc, uint8_t* I2_desc) {
  PLVS_REQUIRE(h && I1_desc && I2_desc, "null argument");
  PLVS_REQUIRE(h->desc_width > 0, "no descriptor images staged");
  const size_t desc_bytes = (size_t)16 * h->desc_width * h->desc_height;
  PLVS_HIP_TRY(hipMemcpyAsync(I1_desc, h->desc1.p, desc_bytes, hipMemcpyDeviceToHost, h->stream));
  PLVS_HIP_TRY(hipMemcpyAsync(I2_desc, h->desc2.p, desc_bytes, hipMemcpyDeviceToHost, h->stream));
  PLVS_HIP_TRY(hipStreamSynchronize(h->stream));
  return PLVS_OK;
}

int plvs_hip_elas_support_candidates(plvs_elas* h, const uint8_t* I1_desc, const uint8_t* I2_desc, int width, int height,
                                     int16_t* D_can) {
  PLVS_REQUIRE(h && D_can, "null argument");
  PLVS_REQUIRE(width >= 16 && height >= 16, "image size");
  PLVS_REQUIRE((I1_desc == nullptr) == (I2_desc == nullptr), "both descriptor images or neither");
  PLVS_REQUIRE(I1_desc != nullptr || (h->desc_width == width && h->desc_height == height),
               "no descriptor images staged for this size (plvs_hip_elas_set_images, or pass them)");
  PLVS_REQUIRE(h->prm.candidate_stepsize > 0 && h->prm.disp_max >= h->prm.disp_min && h->prm.disp_max < 32767, "support parameters");
  hipStream_t s = h->stream;
  SupportParams sp;
  sp.width = width; sp.height = height;
  sp.step = h->prm.candidate_stepsize + (h->prm.subsampling ? h->prm.candidate_stepsize % 2 : 0);   // elas.cpp:420-422
  sp.can_w = (width + sp.step - 1) / sp.step;
  sp.can_h = (height + sp.step - 1) / sp.step;
  sp.disp_min = h->prm.disp_min; sp.disp_max = h->prm.disp_max; sp.support_texture = h->prm.support_texture;
  sp.lr_threshold = h->prm.lr_threshold; sp.support_threshold = h->prm.support_threshold;
  const size_t desc_bytes = (size_t)16 * width * height, ncan = (size_t)sp.can_w * sp.can_h;
  PLVS_HIP_TRY(h->D_can.reserve(ncan));
  if (I1_desc != nullptr) {
    PLVS_HIP_TRY(h->desc1.reserve(desc_bytes));
    PLVS_HIP_TRY(h->desc2.reserve(desc_bytes));
    h->desc_width = h->desc_height = 0;
    PLVS_HIP_TRY(hipMemcpyAsync(h->desc1.p, I1_desc, desc_bytes, hipMemcpyHostToDevice, s));
    PLVS_HIP_TRY(hipMemcpyAsync(h->desc2.p, I2_desc, desc_bytes, hipMemcpyHostToDevice, s));
  }
  PLVS_HIP_TRY(hipMemsetAsync(h->D_can.p, 0, ncan * sizeof(int16_t), s));   // (row 0 / column 0: calloc's zeros, elas.cpp:429)
  if (sp.can_w > 1 && sp.can_h > 1) {
    hipLaunchKernelGGL(elas_support_candidates, dim3((unsigned)(sp.can_w - 1), (unsigned)(sp.can_h - 1)), dim3(64), 0, s, sp,
                       reinterpret_cast<const uint4*>(h->desc1.p), reinterpret_cast<const uint4*>(h->desc2.p), h->D_can.p);
    PLVS_KERNEL_CHECK();
  }
  PLVS_HIP_TRY(hipMemcpyAsync(D_can, h->D_can.p, ncan * sizeof(int16_t), hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  h->desc_width = width;     // (the pair stays staged for the compute_disparity calls that follow)
  h->desc_height = height;
  return PLVS_OK;
}

int plvs_hip_elas_compute_disparity(plvs_elas* h, const int32_t* support, int n_support, const void* tri, int n_tri,
                                    const int32_t* disparity_grid, const int32_t* grid_dims, const uint8_t* I1_desc,
                                    const uint8_t* I2_desc, int width, int height, int right_image, float* D) {
  PLVS_REQUIRE(h && grid_dims, "null argument");   // (D = NULL: the map stays in HBM for plvs_hip_elas_postprocess)
  PLVS_REQUIRE(width >= 8 && height >= 8, "image size");
  PLVS_REQUIRE(n_support >= 0 && n_tri >= 0 && (n_tri == 0 || (support && tri)), "support points / triangles");
  PLVS_REQUIRE(grid_dims[0] >= 2 && grid_dims[1] > 0 && grid_dims[2] > 0 && disparity_grid, "disparity grid");
  PLVS_REQUIRE((I1_desc == nullptr) == (I2_desc == nullptr), "both descriptor images or neither");
  PLVS_REQUIRE(I1_desc != nullptr || (h->desc_width == width && h->desc_height == height),
               "no descriptor images staged for this size (pass them with the first call of a pair)");
  PLVS_REQUIRE((int64_t)grid_dims[1] * h->prm.grid_size >= width && (int64_t)grid_dims[2] * h->prm.grid_size >= height,
               "the disparity grid does not cover the image");
  const Support* hs = reinterpret_cast<const Support*>(support);
  const Triangle* ht = static_cast<const Triangle*>(tri);
  // (u, v, d per support point: inside the image as computeSupportMatches leaves them and addCornerSupportPoints its four
  // corners — plus the two points that function adds for the right image at u = width - 1 + d, beyond the right edge,
  // elas.cpp:262-293; elas_owner clamps columns and rows to the maps)
  for (int i = 0; i < n_support; ++i)
    PLVS_REQUIRE(support[3 * i] >= 0 && support[3 * i] <= width + h->prm.disp_max + 1 && support[3 * i + 1] >= 0 &&
                     support[3 * i + 1] <= height, "support point outside the image");
  for (int i = 0; i < n_tri; ++i)
    PLVS_REQUIRE(ht[i].c1 >= 0 && ht[i].c1 < n_support && ht[i].c2 >= 0 && ht[i].c2 < n_support && ht[i].c3 >= 0 &&
                     ht[i].c3 < n_support, "triangle corner outside the support points");
  hipStream_t s = h->stream;
  const bool sub = h->prm.subsampling != 0;
  const int ow = sub ? width / 2 : width, oh = sub ? height / 2 : height;
  const size_t npix = (size_t)ow * oh, desc_bytes = (size_t)16 * width * height;
  const size_t ngrid = (size_t)grid_dims[0] * grid_dims[1] * grid_dims[2];
  const int disp_num = grid_dims[0] - 1;
  // the prior table and the plane radius on the host, in the reference's float arithmetic (elas.cpp:861-865: exp / log /
  // ceil of floats under `using namespace std`)
  std::vector<int32_t> P((size_t)disp_num);
  const float two_sigma_squared = 2 * h->prm.sigma * h->prm.sigma;
  for (int32_t delta_d = 0; delta_d < disp_num; ++delta_d)
    P[(size_t)delta_d] = (int32_t)((-std::log(h->prm.gamma + std::exp((float)(-delta_d * delta_d) / two_sigma_squared)) +
                                    std::log(h->prm.gamma)) / h->prm.beta);
  MatchParams mp;
  mp.width = width; mp.height = height; mp.subsampling = sub; mp.right_image = right_image != 0;
  mp.grid_size = h->prm.grid_size; mp.match_texture = h->prm.match_texture;
  mp.plane_radius = (int32_t)std::max(std::ceil(h->prm.sigma * h->prm.sradius), 2.0f);
  mp.disp_num = disp_num; mp.grid_w = grid_dims[1]; mp.grid_stride = grid_dims[0];

  PLVS_HIP_TRY(h->support.reserve((size_t)std::max(n_support, 1)));
  PLVS_HIP_TRY(h->tri.reserve((size_t)std::max(n_tri, 1)));
  PLVS_HIP_TRY(h->grid.reserve(ngrid));
  PLVS_HIP_TRY(h->prior.reserve((size_t)disp_num));
  PLVS_HIP_TRY(h->owner.reserve(npix));
  PLVS_HIP_TRY(h->D.reserve(npix));
  if (I1_desc != nullptr) {
    PLVS_HIP_TRY(h->desc1.reserve(desc_bytes));
    PLVS_HIP_TRY(h->desc2.reserve(desc_bytes));
    h->desc_width = h->desc_height = 0;
    PLVS_HIP_TRY(hipMemcpyAsync(h->desc1.p, I1_desc, desc_bytes, hipMemcpyHostToDevice, s));
    PLVS_HIP_TRY(hipMemcpyAsync(h->desc2.p, I2_desc, desc_bytes, hipMemcpyHostToDevice, s));
  }
  if (n_support) PLVS_HIP_TRY(hipMemcpyAsync(h->support.p, hs, (size_t)n_support * sizeof(Support), hipMemcpyHostToDevice, s));
  if (n_tri) PLVS_HIP_TRY(hipMemcpyAsync(h->tri.p, ht, (size_t)n_tri * sizeof(Triangle), hipMemcpyHostToDevice, s));
  PLVS_HIP_TRY(hipMemcpyAsync(h->grid.p, disparity_grid, ngrid * sizeof(int32_t), hipMemcpyHostToDevice, s));
  PLVS_HIP_TRY(hipMemcpyAsync(h->prior.p, P.data(), (size_t)disp_num * sizeof(int32_t), hipMemcpyHostToDevice, s));
  PLVS_HIP_TRY(hipMemsetAsync(h->owner.p, 0, npix * sizeof(uint32_t), s));
  if (n_tri) {
    hipLaunchKernelGGL(elas_owner, dim3((unsigned)n_tri), dim3(64), 0, s, mp, h->tri.p, n_tri, h->support.p, h->owner.p);
    PLVS_KERNEL_CHECK();
  }
  hipLaunchKernelGGL(elas_match, dim3(ceil_div((size_t)ow, 256), (unsigned)oh), dim3(256), 0, s, mp, h->tri.p, h->support.p,
                     h->owner.p, h->grid.p, h->prior.p, reinterpret_cast<const uint4*>(h->desc1.p),
                     reinterpret_cast<const uint4*>(h->desc2.p), h->D.p);
  PLVS_KERNEL_CHECK();
  {   // the map stays in HBM as well (the post-processing chain reads it there)
    const int side = right_image ? 1 : 0;
    PLVS_HIP_TRY(h->res[side].reserve(npix));
    PLVS_HIP_TRY(hipMemcpyAsync(h->res[side].p, h->D.p, npix * sizeof(float), hipMemcpyDeviceToDevice, s));
    h->res_w[side] = ow;
    h->res_h[side] = oh;
  }
  if (D) PLVS_HIP_TRY(hipMemcpyAsync(D, h->D.p, npix * sizeof(float), hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  if (I1_desc != nullptr) {
    h->desc_width = width;
    h->desc_height = height;
  }
  return PLVS_OK;
}

// (host flavours of the post-processing stages: one upload, the stage, one download)
#define ELAS_STAGE_SIZES()                                                          \
  hipStream_t s = h->stream;                                                        \
  const bool sub = h->prm.subsampling != 0;                                         \
  const int W = sub ? width / 2 : width, H = sub ? height / 2 : height;             \
  const size_t n = (size_t)W * H
int plvs_hip_elas_adaptive_mean(plvs_elas* h, float* D, int width, int height) {
  PLVS_REQUIRE(h && D, "null argument");
  PLVS_REQUIRE(width >= 2 && height >= 2, "image size");
  ELAS_STAGE_SIZES();
  PLVS_HIP_TRY(h->D.reserve(n));
  PLVS_HIP_TRY(hipMemcpyAsync(h->D.p, D, n * sizeof(float), hipMemcpyHostToDevice, s));
  const int rc = mean_core(h, h->D.p, W, H);
  if (rc != PLVS_OK) return rc;
  PLVS_HIP_TRY(hipMemcpyAsync(D, h->D.p, n * sizeof(float), hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  return PLVS_OK;
}

int plvs_hip_elas_left_right_check(plvs_elas* h, float* D1, float* D2, int width, int height) {
  PLVS_REQUIRE(h && D1 && D2, "null argument");
  PLVS_REQUIRE(width >= 2 && height >= 2, "image size");
  ELAS_STAGE_SIZES();
  PLVS_HIP_TRY(h->D.reserve(n));
  PLVS_HIP_TRY(h->D_copy.reserve(n));
  PLVS_HIP_TRY(hipMemcpyAsync(h->D.p, D1, n * sizeof(float), hipMemcpyHostToDevice, s));
  PLVS_HIP_TRY(hipMemcpyAsync(h->D_copy.p, D2, n * sizeof(float), hipMemcpyHostToDevice, s));
  const int rc = lr_core(h, h->D.p, h->D_copy.p, W, H);
  if (rc != PLVS_OK) return rc;
  PLVS_HIP_TRY(hipMemcpyAsync(D1, h->D.p, n * sizeof(float), hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipMemcpyAsync(D2, h->D_copy.p, n * sizeof(float), hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  return PLVS_OK;
}

int plvs_hip_elas_remove_small_segments(plvs_elas* h, float* D, int width, int height) {
  PLVS_REQUIRE(h && D, "null argument");
  PLVS_REQUIRE(width >= 2 && height >= 2, "image size");
  ELAS_STAGE_SIZES();
  PLVS_HIP_TRY(h->D.reserve(n));
  PLVS_HIP_TRY(hipMemcpyAsync(h->D.p, D, n * sizeof(float), hipMemcpyHostToDevice, s));
  const int rc = seg_core(h, h->D.p, W, H);
  if (rc != PLVS_OK) return rc;
  PLVS_HIP_TRY(hipMemcpyAsync(D, h->D.p, n * sizeof(float), hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  return PLVS_OK;
}

int plvs_hip_elas_gap_interpolation(plvs_elas* h, float* D, int width, int height) {
  PLVS_REQUIRE(h && D, "null argument");
  PLVS_REQUIRE(width >= 2 && height >= 2, "image size");
  ELAS_STAGE_SIZES();
  PLVS_HIP_TRY(h->D.reserve(n));
  PLVS_HIP_TRY(hipMemcpyAsync(h->D.p, D, n * sizeof(float), hipMemcpyHostToDevice, s));
  const int rc = gap_core(h, h->D.p, W, H);
  if (rc != PLVS_OK) return rc;
  PLVS_HIP_TRY(hipMemcpyAsync(D, h->D.p, n * sizeof(float), hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  return PLVS_OK;
}

// Elas::process behind its two computeDisparity calls (elas.cpp:100-135) in ONE call, on the maps those calls left in HBM:
// leftRightConsistencyCheck (lr_threshold >= 0), removeSmallSegments (speckle_size > 0), gapInterpolation
// (ipol_gap_width > 0), adaptiveMean (filter_adaptive_mean) — the right map only unless postprocess_only_left.  D1 / D2
// (host, may be NULL): where the finished maps go; they also stay in HBM for plvs_hip_elas_depth_dev.
int plvs_hip_elas_postprocess(plvs_elas* h, int width, int height, int postprocess_only_left, int filter_adaptive_mean, float* D1,
                              float* D2) {
  PLVS_REQUIRE(h, "null argument");
  PLVS_REQUIRE(width >= 2 && height >= 2, "image size");
  ELAS_STAGE_SIZES();
  PLVS_REQUIRE(h->res_w[0] == W && h->res_h[0] == H && h->res_w[1] == W && h->res_h[1] == H,
               "no disparity maps of this size in HBM (plvs_hip_elas_compute_disparity for the left and the right image first)");
  float* d1 = h->res[0].p;
  float* d2 = h->res[1].p;
  int rc = PLVS_OK;
  if (h->prm.lr_threshold >= 0) rc = lr_core(h, d1, d2, W, H);
  if (rc == PLVS_OK && h->prm.speckle_size > 0) {
    rc = seg_core(h, d1, W, H);
    if (rc == PLVS_OK && !postprocess_only_left) rc = seg_core(h, d2, W, H);
  }
  if (rc == PLVS_OK && h->prm.ipol_gap_width > 0) {
    rc = gap_core(h, d1, W, H);
    if (rc == PLVS_OK && !postprocess_only_left) rc = gap_core(h, d2, W, H);
  }
  if (rc == PLVS_OK && filter_adaptive_mean) {
    rc = mean_core(h, d1, W, H);
    if (rc == PLVS_OK && !postprocess_only_left) rc = mean_core(h, d2, W, H);
  }
  if (rc != PLVS_OK) return rc;
  if (D1) PLVS_HIP_TRY(hipMemcpyAsync(D1, d1, n * sizeof(float), hipMemcpyDeviceToHost, s));
  if (D2) PLVS_HIP_TRY(hipMemcpyAsync(D2, d2, n * sizeof(float), hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  return PLVS_OK;
}

// The depth image PointCloudKeyFrame::ProcessStereoLibelas makes of the left map (src/PointCloudKeyFrame.cc:399-420):
// d_depth (device, width x height floats) = bf / d — with subsampling only at rows step * m1, pixels step * n1 and
// step * n1 + 1, zero elsewhere (step = PointCloudMapping::skDownsampleStep).  Reads the left map in HBM; asynchronous on
// `stream` after the handle's own work.
int plvs_hip_elas_depth_dev(plvs_elas* h, float bf, int step, float* d_depth, int width, int height, void* stream) {
  PLVS_REQUIRE(h && d_depth, "null argument");
  PLVS_REQUIRE(width >= 2 && height >= 2 && step >= 1, "image size / step");
  const bool sub = h->prm.subsampling != 0;
  const int W1 = sub ? width / 2 : width, H1 = sub ? height / 2 : height;
  PLVS_REQUIRE(h->res_w[0] == W1 && h->res_h[0] == H1, "no left disparity map of this size in HBM");
  PLVS_HIP_TRY(hipStreamSynchronize(h->stream));
  hipLaunchKernelGGL(depth_from_disparity, dim3(ceil_div((size_t)width, 256), (unsigned)height), dim3(256), 0,
                     static_cast<hipStream_t>(stream), h->res[0].p, W1, H1, bf, sub ? 1 : 0, step, d_depth, width, height);
  PLVS_KERNEL_CHECK();
  return PLVS_OK;
}
#undef ELAS_STAGE_SIZES

}  // extern "C"
