// ORB feature extraction (ORBextractor::operator(), src/ORBextractor.cc:1245-1389)
// for gfx950.
//
// Device stages (all levels of the pyramid are processed by ONE launch per
// stage; a block finds its level / tile / cell through small descriptor tables):
//   resize_level      cv::resize(INTER_LINEAR) 8-bit fixed point, level k from k-1
//                     (:1481-1506; 7 dependent launches, tables built on the host)
//   fast_score_map    FAST-9/16 arc response A of every pixel of the detection
//                     region: image rows are read coalesced into an LDS tile with a
//                     3-px halo; the pixel is a corner at threshold t iff A > t and
//                     its OpenCV score is then A - 1 (stored: A, or 0 if A <= minTh)
//   cell_select       one block per 35-px cell (:919-997): 3x3 non-max suppression
//                     inside the cell at iniThFAST, fallback to minThFAST when the
//                     cell stays empty, ordered (raster) compaction
//   gather_cells      cells -> dense candidate list in (level, cell, raster) order
//   [host]            quadtree distribution per level (orb_octree.hpp)
//   orient_kernel     intensity-centroid moments, one wavefront per keypoint,
//                     cv::fastAtan2
//   blur_levels       7x7 sigma=2 Gaussian, exact fixed point (weights 18 34 48 56 ..),
//                     LDS-tiled separable passes; overlaps the host quadtree
//   [host]            cosf/sinf of the angle with the same libm the reference calls
//   describe_kernel   256-bit rBRIEF, 32 lanes per keypoint
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <functional>
#include <thread>
#include <vector>

#include "common.hpp"
#include "orb_internal.hpp"
#include "orb_octree.hpp"

using namespace plvs;
using namespace plvs::orb;

namespace {

constexpr int kPatchSize = 31;
constexpr int kHalfPatch = 15;
constexpr int kEdgeThreshold = 19;
constexpr int kMinBorder = kEdgeThreshold - 3;  // 16
constexpr int kMaxLevels = 16;

__constant__ int8_t c_pattern[1024];  // 512 (x,y) sample points
__constant__ int c_umax[kHalfPatch + 1];

const int8_t h_pattern[1024] = {
#include "orb_pattern.inc"
};

inline int cv_round_f(float v) { return (int)lrintf(v); }
inline int cv_round_d(double v) { return (int)lrint(v); }
inline int cv_floor_d(double v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil_d(double v) { int i = (int)v; return i + (i < v); }

struct LevelDev {      // per pyramid level, lives in a device table
  int w, h, pitch;     // image size and row pitch (bytes)
  size_t off;          // byte offset of the level inside the pyramid buffers
  int tiles_x, tiles_y, tile0;  // 64x16 tiles of the level; index of its first tile
  int cell0, ncells;   // its cells inside the cell table
};

struct CellDev {
  int level;
  int x0, y0;          // first detection pixel (level coordinates)
  int dw, dh;          // detection size
  int slot0;           // first candidate slot of this cell
};

// ------------------------------------------------------------------ kernels

// cv::resize INTER_LINEAR 8UC1: horizontal taps (xofs, alpha), vertical (yofs, beta)
// precomputed on the host exactly as resize.cpp does.
__global__ __launch_bounds__(256) void resize_level(const uint8_t* __restrict__ src, int sw, int sh,
                                                    int spitch, uint8_t* __restrict__ dst, int dw,
                                                    int dh, int dpitch, const int* __restrict__ xofs,
                                                    const short* __restrict__ alpha,
                                                    const int* __restrict__ yofs,
                                                    const short* __restrict__ beta) {
  const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
  const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (dx >= dw || dy >= dh) return;
  const int sx0 = xofs[2 * dx], sx1 = xofs[2 * dx + 1];
  const int a0 = alpha[2 * dx], a1 = alpha[2 * dx + 1];
  const int sy0 = yofs[2 * dy], sy1 = yofs[2 * dy + 1];
  const int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
  const uint8_t* r0 = src + (size_t)sy0 * spitch;
  const uint8_t* r1 = src + (size_t)sy1 * spitch;
  const int h0 = r0[sx0] * a0 + r0[sx1] * a1;
  const int h1 = r1[sx0] * a0 + r1[sx1] * a1;
  const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
  dst[(size_t)dy * dpitch + dx] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
}

// FAST-9/16 response of one pixel given its 16 circle differences d[k] = v - p[k]:
// max over the 16 arcs of 9 contiguous pixels of min(d) (dark) and of min(-d)
// (bright).  A pixel is a corner at threshold t iff that maximum exceeds t, and
// cornerScore<16> (fast_score.cpp) then equals maximum - 1.
__device__ __forceinline__ int fast_arc_response(const int d[16]) {
  int lo2[16], hi2[16], lo4[16], hi4[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    lo2[k] = min(d[k], d[(k + 1) & 15]);
    hi2[k] = max(d[k], d[(k + 1) & 15]);
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    lo4[k] = min(lo2[k], lo2[(k + 2) & 15]);
    hi4[k] = max(hi2[k], hi2[(k + 2) & 15]);
  }
  int best_dark = -512, best_bright = -512;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int lo9 = min(min(lo4[k], lo4[(k + 4) & 15]), d[(k + 8) & 15]);
    const int hi9 = max(max(hi4[k], hi4[(k + 4) & 15]), d[(k + 8) & 15]);
    best_dark = max(best_dark, lo9);
    best_bright = max(best_bright, -hi9);
  }
  return max(best_dark, best_bright);
}

constexpr int kTileW = 64, kTileH = 16, kHalo = 3;
constexpr int kTilePitch = kTileW + 2 * kHalo + 2;  // 72

__global__ __launch_bounds__(256) void fast_score_map(const LevelDev* __restrict__ levels, int nlevels,
                                                      const uint8_t* __restrict__ pyr,
                                                      uint8_t* __restrict__ score, int min_th) {
  __shared__ uint8_t tile[(kTileH + 2 * kHalo) * kTilePitch];
  // which level / tile is this block?
  int lv = 0;
  for (int l = 1; l < nlevels; ++l)
    if ((int)blockIdx.x >= levels[l].tile0) lv = l;
  const LevelDev L = levels[lv];
  const int t = blockIdx.x - L.tile0;
  const int tx = t % L.tiles_x, ty = t / L.tiles_x;
  const int x0 = tx * kTileW, y0 = ty * kTileH;
  const uint8_t* img = pyr + L.off;
  // coalesced row loads of the (64+6) x (16+6) neighbourhood, clamped at the image edge
  for (int i = threadIdx.x; i < (kTileH + 2 * kHalo) * (kTileW + 2 * kHalo); i += 256) {
    const int r = i / (kTileW + 2 * kHalo), c = i % (kTileW + 2 * kHalo);
    int gx = x0 + c - kHalo, gy = y0 + r - kHalo;
    gx = gx < 0 ? 0 : (gx >= L.w ? L.w - 1 : gx);
    gy = gy < 0 ? 0 : (gy >= L.h ? L.h - 1 : gy);
    tile[r * kTilePitch + c] = img[(size_t)gy * L.pitch + gx];
  }
  __syncthreads();
  const int lx = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int ly = (threadIdx.x >> 6) * 4 + k;
    const int gx = x0 + lx, gy = y0 + ly;
    // detection region of ComputeKeyPointsOctTree: [16+3, size-16-3)
    if (gx < kMinBorder + 3 || gy < kMinBorder + 3 || gx >= L.w - kMinBorder - 3 ||
        gy >= L.h - kMinBorder - 3)
      continue;
    const uint8_t* p = &tile[(ly + kHalo) * kTilePitch + lx + kHalo];
    const int v = p[0];
    // the 16-pixel Bresenham circle of radius 3, clockwise from (0,+3) as in fast.cpp
    const int d[16] = {v - p[3 * kTilePitch],      v - p[3 * kTilePitch + 1],  v - p[2 * kTilePitch + 2],
                       v - p[kTilePitch + 3],      v - p[3],                   v - p[-kTilePitch + 3],
                       v - p[-2 * kTilePitch + 2], v - p[-3 * kTilePitch + 1], v - p[-3 * kTilePitch],
                       v - p[-3 * kTilePitch - 1], v - p[-2 * kTilePitch - 2], v - p[-kTilePitch - 3],
                       v - p[-3],                  v - p[kTilePitch - 3],      v - p[2 * kTilePitch - 2],
                       v - p[3 * kTilePitch - 1]};
    const int a = fast_arc_response(d);
    score[L.off + (size_t)gy * L.pitch + gx] = (uint8_t)(a > min_th ? a : 0);
  }
}

// One block per FAST cell: 3x3 NMS restricted to the cell, two-threshold rule,
// raster-ordered compaction.  Candidate word: x | y << 12 | score << 24, with
// x,y relative to the detection region origin (level coordinate - 16).
constexpr int kCellMax = 72;  // a cell's detection window is < 70 px
__global__ __launch_bounds__(256) void cell_select(const CellDev* __restrict__ cells,
                                                   const LevelDev* __restrict__ levels,
                                                   const uint8_t* __restrict__ score, int ini_th,
                                                   int min_th, uint32_t* __restrict__ slots,
                                                   uint32_t* __restrict__ counts) {
  __shared__ uint8_t s[(kCellMax + 2) * (kCellMax + 2)];
  __shared__ uint32_t wsum[4];
  __shared__ int any_ini;
  const CellDev C = cells[blockIdx.x];
  const LevelDev L = levels[C.level];
  const int pw = C.dw + 2;
  for (int i = threadIdx.x; i < (C.dh + 2) * pw; i += 256) {
    const int r = i / pw - 1, c = i % pw - 1;
    uint8_t v = 0;
    if (r >= 0 && r < C.dh && c >= 0 && c < C.dw)
      v = score[L.off + (size_t)(C.y0 + r) * L.pitch + (C.x0 + c)];
    s[i] = v;
  }
  if (threadIdx.x == 0) any_ini = 0;
  __syncthreads();
  const int npix = C.dw * C.dh;
  const int per = (npix + 255) / 256;
  const int beg = threadIdx.x * per, end = min(beg + per, npix);
  // OpenCV keeps score = A - 1 for corners (A > th) and 0 elsewhere, and a corner
  // survives when its score is strictly greater than its 8 neighbours'.
  auto survives = [&](int i, int th) -> bool {
    const int r = i / C.dw, c = i % C.dw;
    const uint8_t* q = &s[(r + 1) * pw + (c + 1)];
    const int v = q[0];
    if (v <= th) return false;
    const int sc = v - 1;
    auto nb = [&](int o) -> int { const int x = q[o]; return x > th ? x - 1 : 0; };
    return sc > nb(-1) && sc > nb(1) && sc > nb(-pw - 1) && sc > nb(-pw) && sc > nb(-pw + 1) &&
           sc > nb(pw - 1) && sc > nb(pw) && sc > nb(pw + 1);
  };
  // pass 1: iniThFAST
  uint32_t cnt = 0;
  for (int i = beg; i < end; ++i) cnt += survives(i, ini_th) ? 1u : 0u;
  if (cnt) any_ini = 1;  // benign race: every writer stores 1
  __syncthreads();
  const int th = any_ini ? ini_th : min_th;
  if (!any_ini) {
    cnt = 0;
    for (int i = beg; i < end; ++i) cnt += survives(i, th) ? 1u : 0u;
  }
  // exclusive scan of cnt over the block (thread order == raster order)
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t x = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) wsum[wid] = x;
  __syncthreads();
  uint32_t base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (w < wid) base += wsum[w];
    total += wsum[w];
  }
  uint32_t pos = C.slot0 + base + x - cnt;
  for (int i = beg; i < end; ++i) {
    if (survives(i, th)) {
      const int r = i / C.dw, c = i % C.dw;
      const uint32_t px = (uint32_t)(C.x0 + c - kMinBorder), py = (uint32_t)(C.y0 + r - kMinBorder);
      slots[pos++] = px | (py << 12) | ((uint32_t)(s[(r + 1) * pw + (c + 1)] - 1) << 24);
    }
  }
  if (threadIdx.x == 0) counts[blockIdx.x] = total;
}

// Single block: exclusive scan of the per-cell counts -> dense offsets; per-level totals.
__global__ __launch_bounds__(256) void cell_offsets(const uint32_t* __restrict__ counts, int ncells,
                                                    const LevelDev* __restrict__ levels, int nlevels,
                                                    uint32_t* __restrict__ offsets,
                                                    uint32_t* __restrict__ level_counts) {
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int start = 0; start < ncells; start += 256) {
    const int i = start + threadIdx.x;
    const uint32_t v = i < ncells ? counts[i] : 0u;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = __shfl_up(x, d, 64);
      if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wid] = x;
    __syncthreads();
    uint32_t base = carry_s, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w < wid) base += wsum[w];
      total += wsum[w];
    }
    if (i < ncells) offsets[i] = base + x - v;
    __syncthreads();
    if (threadIdx.x == 0) carry_s += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) offsets[ncells] = carry_s;
  __syncthreads();
  if (threadIdx.x < nlevels) {
    const LevelDev L = levels[threadIdx.x];
    const uint32_t b = offsets[L.cell0], e = offsets[L.cell0 + L.ncells];
    level_counts[threadIdx.x] = e - b;
  }
}

__global__ __launch_bounds__(64) void gather_cells(const CellDev* __restrict__ cells,
                                                   const uint32_t* __restrict__ counts,
                                                   const uint32_t* __restrict__ offsets,
                                                   const uint32_t* __restrict__ slots,
                                                   uint32_t* __restrict__ dense) {
  const uint32_t n = counts[blockIdx.x], src = cells[blockIdx.x].slot0, dst = offsets[blockIdx.x];
  for (uint32_t i = threadIdx.x; i < n; i += 64) dense[dst + i] = slots[src + i];
}

// cv::fastAtan2 (degrees), mathfuncs_core.simd.hpp atan_f32; plain IEEE f32 ops.
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float scale = (float)(180 / 3.1415926535897932384626433832795);
  const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale,
              p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)2.2204460492503131e-16);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)2.2204460492503131e-16);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

struct KpDev {   // selected keypoint, level coordinates
  int x, y, level;
  float angle;   // out (orient_kernel)
  float a, b;    // cos / sin of the angle, in (describe_kernel)
};

// IC_Angle (src/ORBextractor.cc:110-137): one wavefront per keypoint, the 709
// pixels of the circular patch are spread over the 64 lanes.
__global__ __launch_bounds__(256) void orient_kernel(KpDev* __restrict__ kps, int n,
                                                     const LevelDev* __restrict__ levels,
                                                     const uint8_t* __restrict__ pyr) {
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= n) return;
  const int lane = threadIdx.x & 63;
  const KpDev kp = kps[k];
  const LevelDev L = levels[kp.level];
  const uint8_t* center = pyr + L.off + (size_t)kp.y * L.pitch + kp.x;
  int m01 = 0, m10 = 0;
  // rows v = -15..15; lanes 0..30 cover u = -15..15 of two rows at a time
  const int u = (lane & 31) - kHalfPatch;
  for (int vv = -kHalfPatch + (lane >> 5); vv <= kHalfPatch; vv += 2) {
    const int av = vv < 0 ? -vv : vv;
    if ((lane & 31) <= 2 * kHalfPatch && u >= -c_umax[av] && u <= c_umax[av]) {
      const int val = center[vv * L.pitch + u];
      m10 += u * val;
      m01 += vv * val;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    m10 += __shfl_xor(m10, d, 64);
    m01 += __shfl_xor(m01, d, 64);
  }
  if (lane == 0) kps[k].angle = fast_atan2_deg((float)m01, (float)m10);
}

// GaussianBlur 7x7 sigma 2, BORDER_REFLECT_101, the exact ufixedpoint path:
// out = (sum_y w_y * sum_x w_x * p + 2^15) >> 16 with w = {18,34,48,56,48,34,18}.
constexpr int kBlurR = 3;
__device__ __forceinline__ int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
  return p;
}
__global__ __launch_bounds__(256) void blur_levels(const LevelDev* __restrict__ levels, int nlevels,
                                                   const uint8_t* __restrict__ pyr,
                                                   uint8_t* __restrict__ blurred) {
  __shared__ uint8_t tile[(kTileH + 2 * kBlurR) * kTilePitch];
  __shared__ uint16_t hrow[(kTileH + 2 * kBlurR) * kTileW];
  int lv = 0;
  for (int l = 1; l < nlevels; ++l)
    if ((int)blockIdx.x >= levels[l].tile0) lv = l;
  const LevelDev L = levels[lv];
  const int t = blockIdx.x - L.tile0;
  const int x0 = (t % L.tiles_x) * kTileW, y0 = (t / L.tiles_x) * kTileH;
  const uint8_t* img = pyr + L.off;
  for (int i = threadIdx.x; i < (kTileH + 2 * kBlurR) * (kTileW + 2 * kBlurR); i += 256) {
    const int r = i / (kTileW + 2 * kBlurR), c = i % (kTileW + 2 * kBlurR);
    const int gx = reflect101(x0 + c - kBlurR, L.w), gy = reflect101(y0 + r - kBlurR, L.h);
    tile[r * kTilePitch + c] = img[(size_t)gy * L.pitch + gx];
  }
  __syncthreads();
  // horizontal pass: 8.8 fixed point, at most 255*256
  for (int i = threadIdx.x; i < (kTileH + 2 * kBlurR) * kTileW; i += 256) {
    const int r = i / kTileW, c = i % kTileW;
    const uint8_t* p = &tile[r * kTilePitch + c];
    hrow[i] = (uint16_t)(18 * (p[0] + p[6]) + 34 * (p[1] + p[5]) + 48 * (p[2] + p[4]) + 56 * p[3]);
  }
  __syncthreads();
  const int lx = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int ly = (threadIdx.x >> 6) * 4 + k;
    const int gx = x0 + lx, gy = y0 + ly;
    if (gx >= L.w || gy >= L.h) continue;
    const uint16_t* q = &hrow[ly * kTileW + lx];
    const uint32_t acc = 18u * (q[0] + q[6 * kTileW]) + 34u * (q[kTileW] + q[5 * kTileW]) +
                         48u * (q[2 * kTileW] + q[4 * kTileW]) + 56u * q[3 * kTileW];
    blurred[L.off + (size_t)gy * L.pitch + gx] = (uint8_t)((acc + (1u << 15)) >> 16);
  }
}

// computeOrbDescriptor (src/ORBextractor.cc:141-182): lane = descriptor byte, 32
// lanes per keypoint, two keypoints per wavefront.
__global__ __launch_bounds__(256) void describe_kernel(const KpDev* __restrict__ kps, int n,
                                                       const LevelDev* __restrict__ levels,
                                                       const uint8_t* __restrict__ blurred,
                                                       uint8_t* __restrict__ desc) {
  const int k = blockIdx.x * 8 + (threadIdx.x >> 5);
  if (k >= n) return;
  const int byte = threadIdx.x & 31;
  const KpDev kp = kps[k];
  const LevelDev L = levels[kp.level];
  const uint8_t* center = blurred + L.off + (size_t)kp.y * L.pitch + kp.x;
  const float a = kp.a, b = kp.b;
  const int8_t* pat = c_pattern + byte * 32;
  int val = 0;
#pragma unroll
  for (int bit = 0; bit < 8; ++bit) {
    const float x0 = (float)pat[4 * bit + 0], y0 = (float)pat[4 * bit + 1];
    const float x1 = (float)pat[4 * bit + 2], y1 = (float)pat[4 * bit + 3];
    const int t0 = center[__float2int_rn(x0 * b + y0 * a) * L.pitch + __float2int_rn(x0 * a - y0 * b)];
    const int t1 = center[__float2int_rn(x1 * b + y1 * a) * L.pitch + __float2int_rn(x1 * a - y1 * b)];
    val |= (t0 < t1) << bit;
  }
  desc[(size_t)k * 32 + byte] = (uint8_t)val;
}

}  // namespace

// ------------------------------------------------------------------ handle
struct plvs_orb {
  // configuration (ORBextractor constructor, src/ORBextractor.cc:446-523)
  int nfeatures, nlevels, ini_th, min_th;
  float scale_factor;
  std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
  std::vector<int> features_per_level;
  std::vector<plvs::orb::QuadTree> tree_scratch;   // per level: node array and key pool of the quadtree, kept across frames
  plvs::HostPool pool;                             // the quadtree's helper threads
  int umax[kHalfPatch + 1];
  // geometry for the current image size
  int img_w = 0, img_h = 0;
  std::vector<LevelDev> levels;
  std::vector<CellDev> cells;
  int total_tiles = 0, total_slots = 0;
  size_t pyr_bytes = 0;
  // device buffers
  uint8_t *d_pyr = nullptr, *d_blur = nullptr, *d_score = nullptr;
  LevelDev* d_levels = nullptr;
  CellDev* d_cells = nullptr;
  std::vector<int*> d_xofs, d_yofs;
  std::vector<short*> d_alpha, d_beta;
  uint32_t *d_slots = nullptr, *d_counts = nullptr, *d_offsets = nullptr, *d_level_counts = nullptr,
           *d_dense = nullptr;
  KpDev* d_kps = nullptr;
  uint8_t* d_desc = nullptr;
  int kp_cap = 0;
  // pinned host staging
  uint8_t* h_img = nullptr;
  uint32_t* h_dense = nullptr;
  uint32_t* h_level_counts = nullptr;
  KpDev* h_kps = nullptr;
  uint8_t* h_desc = nullptr;
  hipStream_t stream = nullptr, stream2 = nullptr;
  hipEvent_t ev_pyr = nullptr;
  std::function<void()> pyramid_hook;   // see orb_internal.hpp
  // stage timing of the last call (ms): gpu segments by events, host by clock
  double last_ms[8] = {};
};

namespace {

void free_geometry(plvs_orb* o) {
  (void)hipFree(o->d_pyr); (void)hipFree(o->d_blur); (void)hipFree(o->d_score);
  (void)hipFree(o->d_levels); (void)hipFree(o->d_cells);
  for (auto p : o->d_xofs) (void)hipFree(p);
  for (auto p : o->d_yofs) (void)hipFree(p);
  for (auto p : o->d_alpha) (void)hipFree(p);
  for (auto p : o->d_beta) (void)hipFree(p);
  o->d_xofs.clear(); o->d_yofs.clear(); o->d_alpha.clear(); o->d_beta.clear();
  (void)hipFree(o->d_slots); (void)hipFree(o->d_counts); (void)hipFree(o->d_offsets);
  (void)hipFree(o->d_level_counts); (void)hipFree(o->d_dense);
  if (o->h_img) (void)hipHostFree(o->h_img);
  if (o->h_dense) (void)hipHostFree(o->h_dense);
  if (o->h_level_counts) (void)hipHostFree(o->h_level_counts);
  o->d_pyr = o->d_blur = o->d_score = nullptr;
  o->d_levels = nullptr; o->d_cells = nullptr;
  o->d_slots = o->d_counts = o->d_offsets = o->d_level_counts = o->d_dense = nullptr;
  o->h_img = nullptr; o->h_dense = nullptr; o->h_level_counts = nullptr;
  o->img_w = o->img_h = 0;
}

// Linear-resize tap tables of cv::resize (resize.cpp), for one axis.
void resize_taps(int ssize, int dsize, std::vector<int>& ofs, std::vector<short>& coef, bool is_x) {
  const double inv_scale = (double)dsize / ssize;
  const double scale = 1. / inv_scale;
  ofs.resize(2 * dsize);
  coef.resize(2 * dsize);
  for (int d = 0; d < dsize; ++d) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = cv_floor_d(f);
    f -= s;
    int s0, s1;
    if (is_x) {
      if (s < 0) { f = 0; s = 0; }
      if (s >= ssize - 1) { f = 0; s = ssize - 1; }
      s0 = s;
      s1 = s + 1 < ssize ? s + 1 : ssize - 1;  // weight is 0 whenever this clamps
    } else {
      s0 = s < 0 ? 0 : (s >= ssize ? ssize - 1 : s);
      s1 = s + 1 < 0 ? 0 : (s + 1 >= ssize ? ssize - 1 : s + 1);
    }
    ofs[2 * d] = s0;
    ofs[2 * d + 1] = s1;
    auto sat = [](int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); };
    coef[2 * d] = sat(cv_round_f((1.f - f) * 2048));
    coef[2 * d + 1] = sat(cv_round_f(f * 2048));
  }
}

int build_geometry(plvs_orb* o, int w, int h) {
  free_geometry(o);
  o->levels.assign(o->nlevels, LevelDev{});
  o->cells.clear();
  size_t off = 0;
  int tile0 = 0, slot0 = 0;
  for (int l = 0; l < o->nlevels; ++l) {
    LevelDev& L = o->levels[l];
    L.w = cv_round_f((float)w * o->inv_scale[l]);   // ComputePyramid :1485
    L.h = cv_round_f((float)h * o->inv_scale[l]);
    if (L.w < 1 || L.h < 1) {
      plvs::set_error("orb: pyramid level %d is empty for a %dx%d image", l, w, h);
      return PLVS_ERR_INVALID_ARG;
    }
    L.pitch = (L.w + 63) & ~63;
    L.off = off;
    off += (size_t)L.pitch * L.h;
    L.tiles_x = (L.w + kTileW - 1) / kTileW;
    L.tiles_y = (L.h + kTileH - 1) / kTileH;
    L.tile0 = tile0;
    tile0 += L.tiles_x * L.tiles_y;
    // cells of ComputeKeyPointsOctTree (:867-997)
    L.cell0 = (int)o->cells.size();
    const int maxBX = L.w - kEdgeThreshold + 3, maxBY = L.h - kEdgeThreshold + 3;
    const float width = (float)(maxBX - kMinBorder), height = (float)(maxBY - kMinBorder);
    if (width > 0 && height > 0) {
      const int nCols = (int)(width / 35.f), nRows = (int)(height / 35.f);
      if (nCols > 0 && nRows > 0) {
        const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
        for (int i = 0; i < nRows; ++i) {
          const float iniY = (float)(kMinBorder + i * hCell);
          float maxY = iniY + hCell + 6;
          if (iniY >= maxBY - 3) continue;
          if (maxY > maxBY) maxY = (float)maxBY;
          for (int j = 0; j < nCols; ++j) {
            const float iniX = (float)(kMinBorder + j * wCell);
            float maxX = iniX + wCell + 6;
            if (iniX >= maxBX - 6) continue;
            if (maxX > maxBX) maxX = (float)maxBX;
            CellDev c;
            c.level = l;
            c.x0 = (int)iniX + 3;
            c.y0 = (int)iniY + 3;
            c.dw = (int)maxX - (int)iniX - 6;   // cv::FAST skips a 3-px frame of the sub-image
            c.dh = (int)maxY - (int)iniY - 6;
            if (c.dw <= 0 || c.dh <= 0) continue;
            if (c.dw > kCellMax - 2 || c.dh > kCellMax - 2) {
              plvs::set_error("orb: FAST cell of %dx%d exceeds the kernel limit", c.dw, c.dh);
              return PLVS_ERR_CAPACITY;
            }
            c.slot0 = slot0;
            slot0 += ((c.dw + 1) / 2) * ((c.dh + 1) / 2);   // NMS survivors are never 8-adjacent
            o->cells.push_back(c);
          }
        }
      }
    }
    L.ncells = (int)o->cells.size() - L.cell0;
  }
  o->pyr_bytes = off;
  o->total_tiles = tile0;
  o->total_slots = slot0 > 0 ? slot0 : 1;
  o->img_w = w;
  o->img_h = h;
  const size_t ncells = o->cells.size();
  PLVS_HIP_TRY(hipMalloc((void**)&o->d_pyr, off));
  PLVS_HIP_TRY(hipMalloc((void**)&o->d_blur, off));
  PLVS_HIP_TRY(hipMalloc((void**)&o->d_score, off));
  PLVS_HIP_TRY(hipMemset(o->d_score, 0, off));
  PLVS_HIP_TRY(hipMalloc((void**)&o->d_levels, sizeof(LevelDev) * o->nlevels));
  PLVS_HIP_TRY(hipMemcpy(o->d_levels, o->levels.data(), sizeof(LevelDev) * o->nlevels, hipMemcpyHostToDevice));
  PLVS_HIP_TRY(hipMalloc((void**)&o->d_cells, sizeof(CellDev) * (ncells + 1)));
  if (ncells)
    PLVS_HIP_TRY(hipMemcpy(o->d_cells, o->cells.data(), sizeof(CellDev) * ncells, hipMemcpyHostToDevice));
  PLVS_HIP_TRY(hipMalloc((void**)&o->d_slots, sizeof(uint32_t) * o->total_slots));
  PLVS_HIP_TRY(hipMalloc((void**)&o->d_dense, sizeof(uint32_t) * o->total_slots));
  PLVS_HIP_TRY(hipMalloc((void**)&o->d_counts, sizeof(uint32_t) * (ncells + 1)));
  PLVS_HIP_TRY(hipMalloc((void**)&o->d_offsets, sizeof(uint32_t) * (ncells + 2)));
  PLVS_HIP_TRY(hipMalloc((void**)&o->d_level_counts, sizeof(uint32_t) * kMaxLevels));
  PLVS_HIP_TRY(hipHostMalloc((void**)&o->h_img, (size_t)o->levels[0].pitch * h));
  PLVS_HIP_TRY(hipHostMalloc((void**)&o->h_dense, sizeof(uint32_t) * o->total_slots));
  PLVS_HIP_TRY(hipHostMalloc((void**)&o->h_level_counts, sizeof(uint32_t) * kMaxLevels));
  // resize tap tables, level l from level l-1
  o->d_xofs.assign(o->nlevels, nullptr); o->d_yofs.assign(o->nlevels, nullptr);
  o->d_alpha.assign(o->nlevels, nullptr); o->d_beta.assign(o->nlevels, nullptr);
  for (int l = 1; l < o->nlevels; ++l) {
    std::vector<int> xo, yo;
    std::vector<short> al, be;
    resize_taps(o->levels[l - 1].w, o->levels[l].w, xo, al, true);
    resize_taps(o->levels[l - 1].h, o->levels[l].h, yo, be, false);
    PLVS_HIP_TRY(hipMalloc((void**)&o->d_xofs[l], xo.size() * sizeof(int)));
    PLVS_HIP_TRY(hipMalloc((void**)&o->d_yofs[l], yo.size() * sizeof(int)));
    PLVS_HIP_TRY(hipMalloc((void**)&o->d_alpha[l], al.size() * sizeof(short)));
    PLVS_HIP_TRY(hipMalloc((void**)&o->d_beta[l], be.size() * sizeof(short)));
    PLVS_HIP_TRY(hipMemcpy(o->d_xofs[l], xo.data(), xo.size() * sizeof(int), hipMemcpyHostToDevice));
    PLVS_HIP_TRY(hipMemcpy(o->d_yofs[l], yo.data(), yo.size() * sizeof(int), hipMemcpyHostToDevice));
    PLVS_HIP_TRY(hipMemcpy(o->d_alpha[l], al.data(), al.size() * sizeof(short), hipMemcpyHostToDevice));
    PLVS_HIP_TRY(hipMemcpy(o->d_beta[l], be.data(), be.size() * sizeof(short), hipMemcpyHostToDevice));
  }
  return PLVS_OK;
}

double now_ms() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

}  // namespace

extern "C" {

int plvs_hip_orb_create(int nfeatures, float scale_factor, int nlevels, int ini_th_fast,
                        int min_th_fast, plvs_orb** out) {
  PLVS_REQUIRE(out, "null output");
  PLVS_REQUIRE(nfeatures > 0 && nlevels > 0 && nlevels <= kMaxLevels && scale_factor > 1.0f,
               "bad extractor parameters");
  plvs_orb* o = new plvs_orb();
  o->nfeatures = nfeatures;
  o->nlevels = nlevels;
  o->ini_th = ini_th_fast;
  o->min_th = min_th_fast;
  o->scale_factor = scale_factor;
  // scale tables (:455-470)
  o->scale.resize(nlevels); o->sigma2.resize(nlevels);
  o->inv_scale.resize(nlevels); o->inv_sigma2.resize(nlevels);
  o->scale[0] = 1.0f; o->sigma2[0] = 1.0f;
  for (int i = 1; i < nlevels; ++i) {
    o->scale[i] = o->scale[i - 1] * scale_factor;
    o->sigma2[i] = o->scale[i] * o->scale[i];
  }
  for (int i = 0; i < nlevels; ++i) {
    o->inv_scale[i] = 1.0f / o->scale[i];
    o->inv_sigma2[i] = 1.0f / o->sigma2[i];
  }
  // features per level (:481-493)
  o->features_per_level.resize(nlevels);
  o->tree_scratch.resize(nlevels);
  const float factor = 1.0f / scale_factor;
  float desired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
  int sum = 0;
  for (int l = 0; l < nlevels - 1; ++l) {
    o->features_per_level[l] = cv_round_f(desired);
    sum += o->features_per_level[l];
    desired *= factor;
  }
  o->features_per_level[nlevels - 1] = std::max(nfeatures - sum, 0);
  // circular patch rows (:501-517)
  int v, v0;
  const int vmax = cv_floor_d(kHalfPatch * sqrtf(2.f) / 2 + 1);
  const int vmin = cv_ceil_d(kHalfPatch * sqrtf(2.f) / 2);
  const double hp2 = kHalfPatch * kHalfPatch;
  for (v = 0; v <= vmax; ++v) o->umax[v] = cv_round_d(std::sqrt(hp2 - v * v));
  for (v = kHalfPatch, v0 = 0; v >= vmin; --v) {
    while (o->umax[v0] == o->umax[v0 + 1]) ++v0;
    o->umax[v] = v0;
    ++v0;
  }
#define ORB_CREATE_TRY(call)                                               \
  do {                                                                     \
    hipError_t _e = (call);                                                \
    if (_e != hipSuccess) {                                                \
      plvs::set_error("%s failed: %s", #call, hipGetErrorString(_e));     \
      plvs_hip_orb_destroy(o);                                             \
      return PLVS_ERR_HIP;                                                 \
    }                                                                      \
  } while (0)
  ORB_CREATE_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_pattern), h_pattern, sizeof(h_pattern)));
  ORB_CREATE_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_umax), o->umax, sizeof(o->umax)));
  ORB_CREATE_TRY(hipStreamCreateWithFlags(&o->stream, hipStreamNonBlocking));
  ORB_CREATE_TRY(hipStreamCreateWithFlags(&o->stream2, hipStreamNonBlocking));
  ORB_CREATE_TRY(hipEventCreateWithFlags(&o->ev_pyr, hipEventDisableTiming));
  o->kp_cap = nfeatures * 2 + 1024;
  ORB_CREATE_TRY(hipMalloc((void**)&o->d_kps, sizeof(KpDev) * o->kp_cap));
  ORB_CREATE_TRY(hipMalloc((void**)&o->d_desc, (size_t)32 * o->kp_cap));
  ORB_CREATE_TRY(hipHostMalloc((void**)&o->h_kps, sizeof(KpDev) * o->kp_cap));
  ORB_CREATE_TRY(hipHostMalloc((void**)&o->h_desc, (size_t)32 * o->kp_cap));
#undef ORB_CREATE_TRY
  *out = o;
  return PLVS_OK;
}

int plvs_hip_orb_destroy(plvs_orb* o) {
  if (!o) return PLVS_OK;
  free_geometry(o);
  (void)hipFree(o->d_kps);
  (void)hipFree(o->d_desc);
  if (o->h_kps) (void)hipHostFree(o->h_kps);
  if (o->h_desc) (void)hipHostFree(o->h_desc);
  if (o->stream) (void)hipStreamDestroy(o->stream);
  if (o->stream2) (void)hipStreamDestroy(o->stream2);
  if (o->ev_pyr) (void)hipEventDestroy(o->ev_pyr);
  delete o;
  return PLVS_OK;
}

int plvs_hip_orb_get_levels(plvs_orb* o) { return o ? o->nlevels : 0; }
float plvs_hip_orb_get_scale_factor(plvs_orb* o) { return o ? o->scale_factor : 0.f; }
int plvs_hip_orb_get_scale_tables(plvs_orb* o, float* scale, float* inv_scale, float* sigma2,
                                  float* inv_sigma2) {
  PLVS_REQUIRE(o, "null handle");
  for (int i = 0; i < o->nlevels; ++i) {
    if (scale) scale[i] = o->scale[i];
    if (inv_scale) inv_scale[i] = o->inv_scale[i];
    if (sigma2) sigma2[i] = o->sigma2[i];
    if (inv_sigma2) inv_sigma2[i] = o->inv_sigma2[i];
  }
  return PLVS_OK;
}
int plvs_hip_orb_features_per_level(plvs_orb* o, int* out) {
  PLVS_REQUIRE(o && out, "null argument");
  for (int i = 0; i < o->nlevels; ++i) out[i] = o->features_per_level[i];
  return PLVS_OK;
}

// Shared body: the level-0 image is already in d_pyr (pitch = levels[0].pitch).
static int orb_extract_body(plvs_orb* o, int lap0, int lap1, plvs_keypoint* kps, uint8_t* desc,
                            int cap, int* n_out, int* mono_out) {
  hipStream_t s = o->stream;
  const int nl = o->nlevels;
  const int ncells = (int)o->cells.size();
  const double t0 = now_ms();
  for (int l = 1; l < nl; ++l) {
    const LevelDev &S = o->levels[l - 1], &D = o->levels[l];
    hipLaunchKernelGGL(resize_level, dim3((D.w + 63) / 64, (D.h + 3) / 4), dim3(256), 0, s,
                       o->d_pyr + S.off, S.w, S.h, S.pitch, o->d_pyr + D.off, D.w, D.h, D.pitch,
                       o->d_xofs[l], o->d_alpha[l], o->d_yofs[l], o->d_beta[l]);
  }
  PLVS_HIP_TRY(hipEventRecord(o->ev_pyr, s));
  if (o->pyramid_hook) {
    std::function<void()> hook = std::move(o->pyramid_hook);
    o->pyramid_hook = nullptr;
    hook();
  }
  hipLaunchKernelGGL(fast_score_map, dim3(o->total_tiles), dim3(256), 0, s, o->d_levels, nl, o->d_pyr,
                     o->d_score, o->min_th);
  uint32_t total = 0;
  if (ncells > 0) {
    hipLaunchKernelGGL(cell_select, dim3(ncells), dim3(256), 0, s, o->d_cells, o->d_levels,
                       o->d_score, o->ini_th, o->min_th, o->d_slots, o->d_counts);
    hipLaunchKernelGGL(cell_offsets, dim3(1), dim3(256), 0, s, o->d_counts, ncells, o->d_levels, nl,
                       o->d_offsets, o->d_level_counts);
    hipLaunchKernelGGL(gather_cells, dim3(ncells), dim3(64), 0, s, o->d_cells, o->d_counts,
                       o->d_offsets, o->d_slots, o->d_dense);
    PLVS_KERNEL_CHECK();
    PLVS_HIP_TRY(hipMemcpyAsync(o->h_level_counts, o->d_level_counts, sizeof(uint32_t) * nl,
                                hipMemcpyDeviceToHost, s));
    PLVS_HIP_TRY(hipStreamSynchronize(s));
    for (int l = 0; l < nl; ++l) total += o->h_level_counts[l];
    if (total)
      PLVS_HIP_TRY(hipMemcpyAsync(o->h_dense, o->d_dense, sizeof(uint32_t) * total,
                                  hipMemcpyDeviceToHost, s));
  } else {
    for (int l = 0; l < nl; ++l) o->h_level_counts[l] = 0;
  }
  // the blur does not depend on the keypoints: run it on the second stream while
  // the host distributes the candidates
  PLVS_HIP_TRY(hipStreamWaitEvent(o->stream2, o->ev_pyr, 0));
  hipLaunchKernelGGL(blur_levels, dim3(o->total_tiles), dim3(256), 0, o->stream2, o->d_levels, nl,
                     o->d_pyr, o->d_blur);
  PLVS_KERNEL_CHECK();
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  const double t1 = now_ms();

  // ---- host: quadtree distribution, the levels in parallel (:1001-1003 runs one thread per level; three
  // threads are as fast here — scripts/experiments/frontend_threads.txt — and start sooner)
  std::vector<std::vector<Cand>> selected(nl);
  {
    std::vector<uint32_t> start(nl + 1, 0);
    for (int l = 0; l < nl; ++l) start[l + 1] = start[l] + o->h_level_counts[l];
    auto work = [&](int l) {
      const LevelDev& L = o->levels[l];
      const uint32_t nb = o->h_level_counts[l];
      if (nb == 0) return;
      std::vector<Cand> cands(nb);
      const uint32_t* src = o->h_dense + start[l];
      for (uint32_t i = 0; i < nb; ++i) {
        const uint32_t wd = src[i];
        cands[i] = Cand{(float)(wd & 0xfffu), (float)((wd >> 12) & 0xfffu), (float)(wd >> 24)};
      }
      selected[l] = distribute_quadtree(cands, kMinBorder, L.w - kEdgeThreshold + 3, kMinBorder,
                                        L.h - kEdgeThreshold + 3, o->features_per_level[l], &o->tree_scratch[l]);
    };
    // the levels are dealt to nt threads (the caller is one of them), largest first to the least loaded
    const int nt = plvs::env_int("PLVS_HIP_ORB_TREE_THREADS", std::min(nl, 3), 1, nl);
    std::vector<int> order(nl);
    for (int l = 0; l < nl; ++l) order[l] = l;
    std::stable_sort(order.begin(), order.end(),
                     [&](int a, int b) { return o->h_level_counts[a] > o->h_level_counts[b]; });
    std::vector<std::vector<int>> share(nt);
    std::vector<uint64_t> load(nt, 0);
    for (int l : order) {
      const int k = (int)(std::min_element(load.begin(), load.end()) - load.begin());
      share[k].push_back(l);
      load[k] += o->h_level_counts[l] + 1u;
    }
    auto run = [&](int k) { for (int l : share[k]) work(l); };
    const std::function<void(int)> job = [&](int j) { run(j + 1); };   // (helper threads kept between frames)
    o->pool.start(nt - 1, &job);
    run(0);
    o->pool.wait();
  }
  int nk = 0;
  for (int l = 0; l < nl; ++l) nk += (int)selected[l].size();
  const double t2 = now_ms();
  *n_out = nk;
  *mono_out = 0;
  if (nk > o->kp_cap) {
    plvs::set_error("orb: %d keypoints exceed the internal capacity %d", nk, o->kp_cap);
    return PLVS_ERR_CAPACITY;
  }
  if (nk == 0) {
    PLVS_HIP_TRY(hipStreamSynchronize(o->stream2));
    return PLVS_OK;
  }
  std::vector<float> response(nk);
  {
    int i = 0;
    for (int l = 0; l < nl; ++l)
      for (const Cand& c : selected[l]) {
        response[i] = c.response;
        KpDev& k = o->h_kps[i++];
        k.x = (int)c.x + kMinBorder;
        k.y = (int)c.y + kMinBorder;
        k.level = l;
        k.angle = 0.f; k.a = 1.f; k.b = 0.f;
      }
  }
  PLVS_HIP_TRY(hipMemcpyAsync(o->d_kps, o->h_kps, sizeof(KpDev) * nk, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(orient_kernel, dim3((nk + 3) / 4), dim3(256), 0, s, o->d_kps, nk, o->d_levels,
                     o->d_pyr);
  PLVS_KERNEL_CHECK();
  PLVS_HIP_TRY(hipMemcpyAsync(o->h_kps, o->d_kps, sizeof(KpDev) * nk, hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  const double t3 = now_ms();
  // cos / sin through the same libm the reference calls (computeOrbDescriptor :147-148)
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
  for (int i = 0; i < nk; ++i) {
    const float ang = o->h_kps[i].angle * factorPI;
    o->h_kps[i].a = cosf(ang);
    o->h_kps[i].b = sinf(ang);
  }
  PLVS_HIP_TRY(hipMemcpyAsync(o->d_kps, o->h_kps, sizeof(KpDev) * nk, hipMemcpyHostToDevice, s));
  PLVS_HIP_TRY(hipStreamSynchronize(o->stream2));  // blurred pyramid complete
  hipLaunchKernelGGL(describe_kernel, dim3((nk + 7) / 8), dim3(256), 0, s, o->d_kps, nk, o->d_levels,
                     o->d_blur, o->d_desc);
  PLVS_KERNEL_CHECK();
  PLVS_HIP_TRY(hipMemcpyAsync(o->h_desc, o->d_desc, (size_t)32 * nk, hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  const double t4 = now_ms();
  // ---- output packing (:1357-1378): overlap-area keypoints go to the back
  if (nk <= cap) {
    int mono = 0, stereo = nk - 1;
    for (int i = 0; i < nk; ++i) {
      const KpDev& k = o->h_kps[i];
      plvs_keypoint kp;
      kp.x = (float)k.x;
      kp.y = (float)k.y;
      if (k.level != 0) {
        kp.x *= o->scale[k.level];
        kp.y *= o->scale[k.level];
      }
      kp.size = (float)(int)(kPatchSize * o->scale[k.level]);
      kp.angle = k.angle;
      kp.response = response[i];
      kp.octave = k.level;
      kp.class_id = -1;
      const int dst = (kp.x >= lap0 && kp.x <= lap1) ? stereo-- : mono++;
      kps[dst] = kp;
      memcpy(desc + (size_t)dst * 32, o->h_desc + (size_t)i * 32, 32);
    }
    *mono_out = mono;
  }
  const double t5 = now_ms();
  o->last_ms[0] = t1 - t0;  // pyramid + FAST + cell selection (+ D2H)
  o->last_ms[1] = t2 - t1;  // host quadtree
  o->last_ms[2] = t3 - t2;  // orientation
  o->last_ms[3] = t4 - t3;  // cos/sin + descriptors
  o->last_ms[4] = t5 - t4;  // packing
  return PLVS_OK;
}

int plvs_hip_orb_extract(plvs_orb* o, const uint8_t* image, int w, int h, int stride, int lap0,
                         int lap1, plvs_keypoint* kps, uint8_t* desc, int cap, int* n,
                         int* mono_index) {
  PLVS_REQUIRE(o && n && mono_index, "null argument");
  *n = 0;
  *mono_index = -1;
  if (!image || w <= 0 || h <= 0) {  // `if(_image.empty()) return -1;`
    plvs::set_error("orb: empty image");
    return PLVS_ERR_EMPTY;
  }
  PLVS_REQUIRE(stride >= w, "stride smaller than width");
  PLVS_REQUIRE(w < 4096 + 32 && h < 4096 + 32, "image larger than 4096 px is not supported");
  if (w != o->img_w || h != o->img_h) {
    int rc = build_geometry(o, w, h);
    if (rc != PLVS_OK) return rc;
  }
  const int pitch = o->levels[0].pitch;
  for (int y = 0; y < h; ++y) memcpy(o->h_img + (size_t)y * pitch, image + (size_t)y * stride, w);
  PLVS_HIP_TRY(hipMemcpyAsync(o->d_pyr, o->h_img, (size_t)pitch * h, hipMemcpyHostToDevice, o->stream));
  int rc = orb_extract_body(o, lap0, lap1, kps, desc, cap, n, mono_index);
  return rc;
}

int plvs_hip_orb_extract_dev(plvs_orb* o, const uint8_t* d_image, int w, int h, int stride, int lap0,
                             int lap1, plvs_keypoint* kps, uint8_t* desc, int cap, int* n,
                             int* mono_index) {
  PLVS_REQUIRE(o && n && mono_index, "null argument");
  *n = 0;
  *mono_index = -1;
  if (!d_image || w <= 0 || h <= 0) {
    plvs::set_error("orb: empty image");
    return PLVS_ERR_EMPTY;
  }
  PLVS_REQUIRE(stride >= w, "stride smaller than width");
  PLVS_REQUIRE(w < 4096 + 32 && h < 4096 + 32, "image larger than 4096 px is not supported");
  if (w != o->img_w || h != o->img_h) {
    int rc = build_geometry(o, w, h);
    if (rc != PLVS_OK) return rc;
  }
  PLVS_HIP_TRY(hipMemcpy2DAsync(o->d_pyr, o->levels[0].pitch, d_image, stride, w, h,
                                hipMemcpyDeviceToDevice, o->stream));
  return orb_extract_body(o, lap0, lap1, kps, desc, cap, n, mono_index);
}

int plvs_hip_orb_last_stage_ms(plvs_orb* o, double* ms, int cap) {
  PLVS_REQUIRE(o && ms, "null argument");
  for (int i = 0; i < 5 && i < cap; ++i) ms[i] = o->last_ms[i];
  return PLVS_OK;
}

// Debug / parity accessors: a pyramid level (blurred or not) and the FAST
// candidates of the last call.
int plvs_hip_orb_level_size(plvs_orb* o, int level, int* w, int* h) {
  PLVS_REQUIRE(o && w && h && level >= 0 && level < o->nlevels && o->img_w > 0, "bad argument");
  *w = o->levels[level].w;
  *h = o->levels[level].h;
  return PLVS_OK;
}
int plvs_hip_orb_download_level(plvs_orb* o, int level, int blurred, uint8_t* out) {
  PLVS_REQUIRE(o && out && level >= 0 && level < o->nlevels && o->img_w > 0, "bad argument");
  const LevelDev& L = o->levels[level];
  PLVS_HIP_TRY(hipMemcpy2D(out, L.w, (blurred ? o->d_blur : o->d_pyr) + L.off, L.pitch, L.w, L.h,
                           hipMemcpyDeviceToHost));
  return PLVS_OK;
}
int plvs_hip_orb_last_candidates(plvs_orb* o, int level, float* xyr, int cap, int* n) {
  PLVS_REQUIRE(o && n && level >= 0 && level < o->nlevels && o->img_w > 0, "bad argument");
  uint32_t start = 0;
  for (int l = 0; l < level; ++l) start += o->h_level_counts[l];
  const int nb = (int)o->h_level_counts[level];
  *n = nb;
  for (int i = 0; i < nb && i < cap; ++i) {
    const uint32_t wd = o->h_dense[start + i];
    xyr[3 * i] = (float)(wd & 0xfffu);
    xyr[3 * i + 1] = (float)((wd >> 12) & 0xfffu);
    xyr[3 * i + 2] = (float)(wd >> 24);
  }
  return PLVS_OK;
}

}  // extern "C"

namespace plvs {

hipEvent_t orb_pyramid_event(const plvs_orb* o) { return o ? o->ev_pyr : nullptr; }
void orb_set_pyramid_hook(plvs_orb* o, std::function<void()> hook) {
  if (o) o->pyramid_hook = std::move(hook);
}

bool orb_pyramid_view(const plvs_orb* o, OrbPyramidView* v) {
  if (o == nullptr || v == nullptr || o->img_w <= 0 || o->d_pyr == nullptr) return false;
  if (o->nlevels > kMaxOrbLevels) return false;
  v->nlevels = o->nlevels;
  for (int l = 0; l < o->nlevels; ++l) {
    const LevelDev& L = o->levels[l];
    v->level[l] = o->d_pyr + L.off;
    v->w[l] = L.w;
    v->h[l] = L.h;
    v->pitch[l] = L.pitch;
    v->scale[l] = o->scale[l];
    v->inv_scale[l] = o->inv_scale[l];
  }
  return true;
}

}  // namespace plvs
