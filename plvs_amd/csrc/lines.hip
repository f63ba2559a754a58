// Line front end: EDLines detection + LBD description as driven by
// LineExtractor::operator() (src/LineExtractor.cc:150, 199-289; Line.LSD.on = 0,
// Line.pyramidPrecomputation = 0).
//
// Split of the work (SURVEY.md §7 "hard parts"):
//   device  lines_blur5        5x5 Gaussian, exact 8.8 fixed point, sigma per octave (:787-815)
//           lines_sobel_grad   3x3 Sobel (s16) + |dx|+|dy| thresholded at 81, /4 with
//                              round-half-even, direction bit (:1642-1654), packed u16
//           lines_resize       cv::resize(fx = fy = 1/scale, INTER_LINEAR) (:840)
//           lbd_kernel         one workgroup per line: a lane per row of the line support
//                              region keeps the reference's sequential coordinates and float
//                              accumulators, all 256 threads gather the gradients of 32
//                              pixels of every row at a time; 9 lanes then fold the
//                              rows into the band statistics in row order; lane 0
//                              normalises and emits the 256-bit descriptor (:1151-1488, :438-449)
//   host    lines_host.hpp     anchor linking, line fitting / validation, octave grouping,
//                              selection — order dependent, one thread per octave
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <thread>
#include <mutex>
#include <functional>
#include <condition_variable>
#include <vector>

#include "common.hpp"
#include "lines_host.hpp"
#include "lsd_host.hpp"
#include "orb_internal.hpp"

using namespace plvs;
using namespace plvs::lines;

namespace {

constexpr int kBands = 9, kBandWidth = 7, kRows = kBands * kBandWidth;  // 63
constexpr int kMaxOctaves = 8;

__device__ __forceinline__ int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
  return p;
}

struct Q5 { int w[5]; };

// exact 2-D fixed-point Gaussian: (sum_y w_y sum_x w_x p + 2^15) >> 16
__global__ __launch_bounds__(256) void lines_blur5(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                   int w, int h, Q5 k) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  uint32_t acc = 0;
#pragma unroll
  for (int j = -2; j <= 2; ++j) {
    const uint8_t* r = src + (size_t)reflect101(y + j, h) * w;
    uint32_t rowv = 0;
#pragma unroll
    for (int i = -2; i <= 2; ++i) rowv += (uint32_t)k.w[i + 2] * r[reflect101(x + i, w)];
    acc += (uint32_t)k.w[j + 2] * rowv;
  }
  dst[(size_t)y * w + x] = (uint8_t)((acc + (1u << 15)) >> 16);
}

__device__ __forceinline__ int round_half_even_quarter(int v) {
  // cvRound(v * 0.25) for a non-negative integer v
  const int q = v >> 2, r = v & 3;
  return q + ((r > 2) || (r == 2 && (q & 1)));
}

__global__ __launch_bounds__(256) void lines_sobel_grad(const uint8_t* __restrict__ img, int w, int h,
                                                        int16_t* __restrict__ dxo, int16_t* __restrict__ dyo,
                                                        uint16_t* __restrict__ gd, int grad_threshold) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  const uint8_t* r0 = img + (size_t)reflect101(y - 1, h) * w;
  const uint8_t* r1 = img + (size_t)y * w;
  const uint8_t* r2 = img + (size_t)reflect101(y + 1, h) * w;
  const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
  const int dx = (r0[xp] - r0[xm]) + 2 * (r1[xp] - r1[xm]) + (r2[xp] - r2[xm]);
  const int dy = (r2[xm] + 2 * r2[x] + r2[xp]) - (r0[xm] + 2 * r0[x] + r0[xp]);
  const int ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy;
  const int sum = ax + ay;
  const int g = round_half_even_quarter(sum > grad_threshold + 1 ? sum : 0);  // THRESH_TOZERO then / 4
  const size_t i = (size_t)y * w + x;
  dxo[i] = (int16_t)dx;
  dyo[i] = (int16_t)dy;
  gd[i] = (uint16_t)(g | (ax < ay ? 0x8000 : 0));
}

// Anchor test of EdgeDrawing (:1665-1691) on the stride-2 scan grid, written in scan
// (column-major) order so that the host consumes it with one linear pass.
__global__ __launch_bounds__(256) void lines_anchor_flags(const uint16_t* __restrict__ gd, int w, int h,
                                                          int rows, int cols, int anchor_threshold,
                                                          uint8_t* __restrict__ flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const int c = i / rows, r = i % rows;
  const int x = 2 * c + 1, y = 2 * r + 1;
  const int idx = y * w + x;
  const int v = gd[idx];
  const int gi = v & 0x1ff;
  bool a;
  if (v & 0x8000)
    a = gi >= (gd[idx - w] & 0x1ff) + anchor_threshold && gi >= (gd[idx + w] & 0x1ff) + anchor_threshold;
  else
    a = gi >= (gd[idx - 1] & 0x1ff) + anchor_threshold && gi >= (gd[idx + 1] & 0x1ff) + anchor_threshold;
  flags[i] = a ? 1 : 0;
}

__global__ __launch_bounds__(256) void lines_resize(const uint8_t* __restrict__ src, int sw,
                                                    uint8_t* __restrict__ dst, int dw, int dh,
                                                    const int* __restrict__ xofs, const short* __restrict__ alpha,
                                                    const int* __restrict__ yofs, const short* __restrict__ beta) {
  const int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (dx >= dw || dy >= dh) return;
  const int sx0 = xofs[2 * dx], sx1 = xofs[2 * dx + 1], a0 = alpha[2 * dx], a1 = alpha[2 * dx + 1];
  const int sy0 = yofs[2 * dy], sy1 = yofs[2 * dy + 1], b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
  const uint8_t* r0 = src + (size_t)sy0 * sw;
  const uint8_t* r1 = src + (size_t)sy1 * sw;
  const int h0 = r0[sx0] * a0 + r0[sx1] * a1, h1 = r1[sx0] * a0 + r1[sx1] * a1;
  const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
  dst[(size_t)dy * dw + dx] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
}

struct LbdLine {   // per selected line, prepared on the host
  int octave, num_pixels;
  float dL0, dL1;          // (float)cos / sin of the line direction (double libm on the host)
  float s0x, s0y;          // sCorX0 / sCorY0 of row 0
};

struct LbdOctave {
  const int16_t* dx;
  const int16_t* dy;
  int w, h;
};

struct LbdTables {
  float g[kRows];          // (float)gaussCoefG_[hID]
  float l[3 * kBandWidth]; // (float)gaussCoefL_[i]
  LbdOctave oct[kMaxOctaves];
};

__constant__ int c_comb[32][2] = {
    {0, 1}, {0, 2}, {0, 3}, {0, 4}, {0, 5}, {0, 6}, {1, 2}, {1, 3}, {1, 4}, {1, 5}, {1, 6},
    {2, 3}, {2, 4}, {2, 5}, {2, 6}, {2, 7}, {2, 8}, {3, 4}, {3, 5}, {3, 6}, {3, 7}, {3, 8},
    {4, 5}, {4, 6}, {4, 7}, {4, 8}, {5, 6}, {5, 7}, {5, 8}, {6, 7}, {6, 8}, {7, 8}};

// One workgroup of 256 per line.  A row of the line support region is a sequential sum over its pixels (float adds in
// pixel order, as the reference takes them) of gradients gathered at rounded, sequentially accumulated coordinates.
// The coordinates and the sums stay with the row's lane; the GATHER — two loads per (row, pixel), what a lane per row
// waited for pixel by pixel: 92 us for 100 lines — is dealt to all 256 threads, kLbdChunk pixels of every row at a time,
// through LDS.
constexpr int kLbdThreads = 256;
constexpr int kLbdChunk = 32;
__global__ __launch_bounds__(kLbdThreads) void lbd_kernel(const LbdLine* __restrict__ lines, LbdTables T,
                                                          uint8_t* __restrict__ desc) {
  __shared__ float rows[kRows][8];   // per row: pgdL, ngdL, pgdL2, ngdL2, pgdO, ngdO, pgdO2, ngdO2
  __shared__ float band[kBands][8];
  __shared__ int s_idx[kRows][kLbdChunk];        // pixel index of (row, pixel of the chunk)
  __shared__ float s_gl[kRows][kLbdChunk + 1], s_go[kRows][kLbdChunk + 1];
  const LbdLine L = lines[blockIdx.x];
  const LbdOctave O = T.oct[L.octave];
  const int hID = threadIdx.x;
  const float dL0 = L.dL0, dL1 = L.dL1, dO0 = -L.dL1, dO1 = L.dL0;
  float sx = 0.f, sy = 0.f, pL = 0, nL = 0, pO = 0, nO = 0;
  const short maxx = (short)(O.w - 1), maxy = (short)(O.h - 1);
  if (hID < kRows) {
    // row start: sCorX0 -= dL[1], sCorY0 += dL[0] once per preceding row (sequential f32)
    float sx0 = L.s0x, sy0 = L.s0y;
    for (int r = 0; r < hID; ++r) { sx0 -= dL1; sy0 += dL0; }
    sx = sx0;
    sy = sy0;
  }
  for (int w0 = 0; w0 < L.num_pixels; w0 += kLbdChunk) {
    const int nw = min(kLbdChunk, L.num_pixels - w0);
    if (hID < kRows) {
      for (int k = 0; k < nw; ++k) {
        short t = (short)roundf(sx);
        const short xc = (t < 0) ? (short)0 : (t > maxx) ? maxx : t;
        t = (short)roundf(sy);
        const short yc = (t < 0) ? (short)0 : (t > maxy) ? maxy : t;
        s_idx[hID][k] = yc * O.w + xc;
        sx += dL0;
        sy += dL1;
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < kRows * nw; e += kLbdThreads) {
      const int r = e / nw, k = e - r * nw;
      const int idx = s_idx[r][k];
      const short dx = O.dx[idx], dy = O.dy[idx];
      s_gl[r][k] = dx * dL0 + dy * dL1;
      s_go[r][k] = dx * dO0 + dy * dO1;
    }
    __syncthreads();
    if (hID < kRows) {
      for (int k = 0; k < nw; ++k) {
        const float gDL = s_gl[hID][k], gDO = s_go[hID][k];
        if (gDL > 0) pL += gDL; else nL -= gDL;
        if (gDO > 0) pO += gDO; else nO -= gDO;
      }
    }
  }
  if (hID < kRows) {
    const float cg = T.g[hID];
    pL = cg * pL; nL = cg * nL; pO = cg * pO; nO = cg * nO;
    rows[hID][0] = pL; rows[hID][1] = nL; rows[hID][2] = pL * pL; rows[hID][3] = nL * nL;
    rows[hID][4] = pO; rows[hID][5] = nO; rows[hID][6] = pO * pO; rows[hID][7] = nO * nO;
  }
  __syncthreads();
  if (threadIdx.x < kBands) {
    // band b receives, in row order: rows of band b-1 (weight l[r%7]), of band b
    // (l[r%7 + 7]) and of band b+1 (l[r%7 + 14])
    const int b = threadIdx.x;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = 0; r < kRows; ++r) {
      const int rb = r / kBandWidth;
      float c;
      if (rb == b - 1) c = T.l[r % kBandWidth];
      else if (rb == b) c = T.l[r % kBandWidth + kBandWidth];
      else if (rb == b + 1) c = T.l[r % kBandWidth + 2 * kBandWidth];
      else continue;
      acc[0] += c * rows[r][0];
      acc[1] += c * rows[r][1];
      acc[2] += c * c * rows[r][2];
      acc[3] += c * c * rows[r][3];
      acc[4] += c * rows[r][4];
      acc[5] += c * rows[r][5];
      acc[6] += c * c * rows[r][6];
      acc[7] += c * c * rows[r][7];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) band[b][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float d[kBands * 8];
    const float invN2 = (float)(1.0 / (kBandWidth * 2.0)), invN3 = (float)(1.0 / (kBandWidth * 3.0));
    for (int b = 0; b < kBands; ++b) {
      const float invN = (b == 0 || b == kBands - 1) ? invN2 : invN3;
      float t = band[b][0] * invN;
      d[b * 8 + 0] = t; d[b * 8 + 4] = sqrtf(band[b][2] * invN - t * t);
      t = band[b][1] * invN;
      d[b * 8 + 1] = t; d[b * 8 + 5] = sqrtf(band[b][3] * invN - t * t);
      t = band[b][4] * invN;
      d[b * 8 + 2] = t; d[b * 8 + 6] = sqrtf(band[b][6] * invN - t * t);
      t = band[b][5] * invN;
      d[b * 8 + 3] = t; d[b * 8 + 7] = sqrtf(band[b][7] * invN - t * t);
    }
    float tM = 0, tS = 0;
    for (int b = 0; b < kBands; ++b) {
      const float* p = &d[b * 8];
      tM += p[0] * p[0]; tM += p[1] * p[1]; tM += p[2] * p[2]; tM += p[3] * p[3];
      tS += p[4] * p[4]; tS += p[5] * p[5]; tS += p[6] * p[6]; tS += p[7] * p[7];
    }
    tM = 1 / sqrtf(tM);
    tS = 1 / sqrtf(tS);
    for (int b = 0; b < kBands; ++b)
      for (int k = 0; k < 8; ++k) d[b * 8 + k] = d[b * 8 + k] * (k < 4 ? tM : tS);
    for (int i = 0; i < kBands * 8; ++i)
      if ((double)d[i] > 0.4) d[i] = (float)0.4;
    float t = 0;
    for (int i = 0; i < kBands * 8; ++i) t += d[i] * d[i];
    t = 1.f / sqrtf(t);
    for (int i = 0; i < kBands * 8; ++i) d[i] = d[i] * t;
    uint8_t* out = desc + (size_t)blockIdx.x * 32;
    for (int c = 0; c < 32; ++c) {
      const float* f1 = &d[8 * c_comb[c][0]];
      const float* f2 = &d[8 * c_comb[c][1]];
      unsigned v = 0;
      for (int i = 0; i < 8; ++i)
        if (f1[i] > f2[i]) v += 1u << i;
      out[c] = (uint8_t)v;
    }
  }
}

inline int cvr(double v) { return (int)lrint(v); }
inline int cvrf(float v) { return (int)lrintf(v); }
inline int cvfloor(double v) { int i = (int)v; return i - (i > v); }

// getGaussianKernel bit-exact fixed point, 8 fractional bits (smooth.dispatch.cpp)
Q5 gaussian_q8_5(double sigma) {
  double k[5], sum = 0;
  const double s2 = -0.5 / (sigma * sigma);
  for (int i = 0; i < 5; ++i) { const double x = i - 2.0; k[i] = std::exp(s2 * x * x); sum += k[i]; }
  for (int i = 0; i < 5; ++i) k[i] /= sum;
  Q5 q;
  double err = 0;
  long long s = 0;
  for (int i = 0; i < 2; ++i) {
    const double adj = k[i] * 256.0 + err;
    const long long v0 = (long long)lrint(adj);
    err = adj - (double)v0;
    q.w[i] = q.w[4 - i] = (int)v0;
    s += v0;
  }
  q.w[2] = (int)(256 - 2 * s);
  return q;
}

// cv::resize tap tables with an explicit inverse scale (fx = fy given, dsize empty)
void resize_taps_factor(int ssize, int dsize, double inv_scale, std::vector<int>& ofs, std::vector<short>& coef, bool is_x) {
  const double scale = 1. / inv_scale;
  ofs.resize(2 * dsize);
  coef.resize(2 * dsize);
  for (int d = 0; d < dsize; ++d) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = cvfloor(f);
    f -= s;
    int s0, s1;
    if (is_x) {
      if (s < 0) { f = 0; s = 0; }
      if (s >= ssize - 1) { f = 0; s = ssize - 1; }
      s0 = s;
      s1 = s + 1 < ssize ? s + 1 : ssize - 1;
    } else {
      s0 = s < 0 ? 0 : (s >= ssize ? ssize - 1 : s);
      s1 = s + 1 < 0 ? 0 : (s + 1 >= ssize ? ssize - 1 : s + 1);
    }
    ofs[2 * d] = s0;
    ofs[2 * d + 1] = s1;
    auto sat = [](int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); };
    coef[2 * d] = sat(cvrf((1.f - f) * 2048));
    coef[2 * d + 1] = sat(cvrf(f * 2048));
  }
}

double now_ms() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

}  // namespace

struct plvs_lines {
  int nfeatures, nlevels;
  float scale;
  int nlevels_cfg = 0;            // the constructor's values; nlevels / scale follow the shared
  float scale_cfg = 0.f;          // pyramid while one is set (BinaryDescriptor::setGaussianPyramid)
  plvs_orb* shared = nullptr;     // ORB extractor whose device pyramid feeds the octaves, or null
  bool geometry_shared = false;   // what the current geometry was built for
  double min_length;
  EdParams ed;
  float gaussG[kRows], gaussL[3 * kBandWidth];
  // geometry of the current image size
  int img_w = 0, img_h = 0;
  std::vector<std::pair<int, int>> sizes;      // (w, h) per octave
  std::vector<Q5> kernels;
  std::vector<uint8_t*> d_img, d_blur;         // octave input / blurred
  std::vector<int16_t*> d_dx, d_dy;
  std::vector<uint16_t*> d_gd;
  std::vector<uint8_t*> d_anchor, h_anchor;    // scan-order anchor flags (device / pinned host)
  std::vector<int*> d_xofs, d_yofs;
  std::vector<short*> d_alpha, d_beta;
  std::vector<int16_t*> h_dx, h_dy;            // pinned
  std::vector<uint16_t*> h_gd;
  uint8_t* h_img = nullptr;
  LbdLine* d_lines = nullptr;
  LbdLine* h_lines = nullptr;
  uint8_t* d_desc = nullptr;
  uint8_t* h_desc = nullptr;
  int line_cap = 0;
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;           // the gradient images go home beside the next octave's kernels
  std::vector<hipEvent_t> ev_route;            // octave i's packed map and anchor flags have landed (what EdgeDrawing reads)
  std::vector<hipEvent_t> ev_maps;             // ... and its dx / dy images (what the line fit reads)
  std::vector<OctaveDetector> det;
  double last_ms[6] = {};
  plvs::HostPool pool;
};

namespace {

void lines_free_geometry(plvs_lines* o) {
  for (auto p : o->d_img) (void)hipFree(p);
  for (auto p : o->d_blur) (void)hipFree(p);
  for (auto p : o->d_dx) (void)hipFree(p);
  for (auto p : o->d_dy) (void)hipFree(p);
  for (auto p : o->d_gd) (void)hipFree(p);
  for (auto p : o->d_anchor) (void)hipFree(p);
  for (auto p : o->h_anchor) if (p) (void)hipHostFree(p);
  o->d_anchor.clear(); o->h_anchor.clear();
  for (auto p : o->d_xofs) (void)hipFree(p);
  for (auto p : o->d_yofs) (void)hipFree(p);
  for (auto p : o->d_alpha) (void)hipFree(p);
  for (auto p : o->d_beta) (void)hipFree(p);
  for (auto p : o->h_dx) if (p) (void)hipHostFree(p);
  for (auto p : o->h_dy) if (p) (void)hipHostFree(p);
  for (auto p : o->h_gd) if (p) (void)hipHostFree(p);
  if (o->h_img) (void)hipHostFree(o->h_img);
  o->d_img.clear(); o->d_blur.clear(); o->d_dx.clear(); o->d_dy.clear(); o->d_gd.clear();
  o->d_xofs.clear(); o->d_yofs.clear(); o->d_alpha.clear(); o->d_beta.clear();
  o->h_dx.clear(); o->h_dy.clear(); o->h_gd.clear();
  o->h_img = nullptr;
  o->img_w = o->img_h = 0;
}

// `view` non-null: octave i is level i of a shared ORB pyramid (OctaveKeyLines :805-808, no
// resize between octaves, :836); otherwise the extractor's own blur -> resize chain.
int lines_build_geometry(plvs_lines* o, int w, int h, const plvs::OrbPyramidView* view = nullptr) {
  lines_free_geometry(o);
  if (view != nullptr) {
    // BinaryDescriptor::setGaussianPyramid :1497-1498
    o->nlevels = std::min(o->nlevels_cfg, view->nlevels);
    o->scale = view->scale[1 < view->nlevels ? 1 : 0];   // ORBextractor::GetScaleFactor(): mvScaleFactor[1]
  } else {
    o->nlevels = o->nlevels_cfg;
    o->scale = o->scale_cfg;
  }
  o->geometry_shared = view != nullptr;
  const int n = o->nlevels;
  o->sizes.assign(n, {0, 0});
  o->kernels.resize(n);
  o->d_img.assign(n, nullptr); o->d_blur.assign(n, nullptr); o->d_dx.assign(n, nullptr);
  o->d_dy.assign(n, nullptr); o->d_gd.assign(n, nullptr);
  o->d_xofs.assign(n, nullptr); o->d_yofs.assign(n, nullptr); o->d_alpha.assign(n, nullptr); o->d_beta.assign(n, nullptr);
  o->h_dx.assign(n, nullptr); o->h_dy.assign(n, nullptr); o->h_gd.assign(n, nullptr);
  o->d_anchor.assign(n, nullptr); o->h_anchor.assign(n, nullptr);
  while ((int)o->ev_maps.size() < n) {
    hipEvent_t e = nullptr;
    PLVS_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    o->ev_maps.push_back(e);
  }
  while ((int)o->ev_route.size() < n) {
    hipEvent_t e = nullptr;
    PLVS_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    o->ev_route.push_back(e);
  }
  // OctaveKeyLines :785-846: sigma schedule and octave sizes
  float preSigma2 = (float)std::pow(0.5, 2);
  float curSigma2 = (float)std::pow(1.0f, 2);
  const double factor = o->scale, factor2 = factor * factor;
  const double inv = (1.f / factor);
  int cw = w, ch = h;
  for (int i = 0; i < n; ++i) {
    if (view != nullptr) {
      cw = view->w[i];
      ch = view->h[i];
    }
    if (cw < 8 || ch < 8) {
      plvs::set_error("lines: octave %d of a %dx%d image is too small", i, w, h);
      return PLVS_ERR_INVALID_ARG;
    }
    o->sizes[i] = {cw, ch};
    const float increaseSigma = std::sqrt(curSigma2 - preSigma2);
    o->kernels[i] = gaussian_q8_5(increaseSigma);
    const size_t px = (size_t)cw * ch;
    PLVS_HIP_TRY(hipMalloc((void**)&o->d_img[i], px));
    PLVS_HIP_TRY(hipMalloc((void**)&o->d_blur[i], px));
    PLVS_HIP_TRY(hipMalloc((void**)&o->d_dx[i], px * 2));
    PLVS_HIP_TRY(hipMalloc((void**)&o->d_dy[i], px * 2));
    PLVS_HIP_TRY(hipMalloc((void**)&o->d_gd[i], px * 2));
    PLVS_HIP_TRY(hipHostMalloc((void**)&o->h_dx[i], px * 2));
    PLVS_HIP_TRY(hipHostMalloc((void**)&o->h_dy[i], px * 2));
    PLVS_HIP_TRY(hipHostMalloc((void**)&o->h_gd[i], px * 2));
    const size_t na = (size_t)((cw - 1) / 2) * ((ch - 1) / 2) + 1;
    PLVS_HIP_TRY(hipMalloc((void**)&o->d_anchor[i], na));
    PLVS_HIP_TRY(hipHostMalloc((void**)&o->h_anchor[i], na));
    const int nw = cvr(cw * inv), nh = cvr(ch * inv);   // dsize = saturate_cast<int>(ssize * f)
    if (i + 1 < n && view == nullptr) {
      std::vector<int> xo, yo;
      std::vector<short> al, be;
      resize_taps_factor(cw, nw, inv, xo, al, true);
      resize_taps_factor(ch, nh, inv, yo, be, false);
      PLVS_HIP_TRY(hipMalloc((void**)&o->d_xofs[i], xo.size() * sizeof(int)));
      PLVS_HIP_TRY(hipMalloc((void**)&o->d_yofs[i], yo.size() * sizeof(int)));
      PLVS_HIP_TRY(hipMalloc((void**)&o->d_alpha[i], al.size() * sizeof(short)));
      PLVS_HIP_TRY(hipMalloc((void**)&o->d_beta[i], be.size() * sizeof(short)));
      PLVS_HIP_TRY(hipMemcpy(o->d_xofs[i], xo.data(), xo.size() * sizeof(int), hipMemcpyHostToDevice));
      PLVS_HIP_TRY(hipMemcpy(o->d_yofs[i], yo.data(), yo.size() * sizeof(int), hipMemcpyHostToDevice));
      PLVS_HIP_TRY(hipMemcpy(o->d_alpha[i], al.data(), al.size() * sizeof(short), hipMemcpyHostToDevice));
      PLVS_HIP_TRY(hipMemcpy(o->d_beta[i], be.data(), be.size() * sizeof(short), hipMemcpyHostToDevice));
    }
    cw = nw; ch = nh;
    preSigma2 = curSigma2;
    curSigma2 = (float)(curSigma2 * factor2);
  }
  PLVS_HIP_TRY(hipHostMalloc((void**)&o->h_img, (size_t)w * h));
  o->img_w = w;
  o->img_h = h;
  return PLVS_OK;
}

int lines_extract_body(plvs_lines* o, plvs_keyline* keylines, uint8_t* desc, int cap, int* n_out) {
  hipStream_t s = o->stream;
  const int n = o->nlevels;
  const double t0 = now_ms();
  for (int i = 0; i < n; ++i) {
    const int w = o->sizes[i].first, h = o->sizes[i].second;
    const dim3 grid((w + 63) / 64, (h + 3) / 4), block(256);
    hipLaunchKernelGGL(lines_blur5, grid, block, 0, s, o->d_img[i], o->d_blur[i], w, h, o->kernels[i]);
    hipLaunchKernelGGL(lines_sobel_grad, grid, block, 0, s, o->d_blur[i], w, h, o->d_dx[i], o->d_dy[i],
                       o->d_gd[i], 80);
    const size_t px = (size_t)w * h;
    const int arows = (h - 1) / 2, acols = (w - 1) / 2;
    if (arows > 0 && acols > 0) {
      hipLaunchKernelGGL(lines_anchor_flags, dim3((arows * acols + 255) / 256), block, 0, s, o->d_gd[i], w, h,
                         arows, acols, o->ed.anchor_threshold, o->d_anchor[i]);
      PLVS_HIP_TRY(hipMemcpyAsync(o->h_anchor[i], o->d_anchor[i], (size_t)arows * acols, hipMemcpyDeviceToHost, s));
    }
    PLVS_HIP_TRY(hipMemcpyAsync(o->h_gd[i], o->d_gd[i], px * 2, hipMemcpyDeviceToHost, s));
    PLVS_HIP_TRY(hipEventRecord(o->ev_route[i], s));
    // the gradient images (two thirds of the bytes) follow on a second stream: EdgeDrawing starts without them, the
    // next octave's kernels do not queue behind them
    PLVS_HIP_TRY(hipStreamWaitEvent(o->copy_stream, o->ev_route[i], 0));
    PLVS_HIP_TRY(hipMemcpyAsync(o->h_dx[i], o->d_dx[i], px * 2, hipMemcpyDeviceToHost, o->copy_stream));
    PLVS_HIP_TRY(hipMemcpyAsync(o->h_dy[i], o->d_dy[i], px * 2, hipMemcpyDeviceToHost, o->copy_stream));
    PLVS_HIP_TRY(hipEventRecord(o->ev_maps[i], o->copy_stream));
    if (i + 1 < n && !o->geometry_shared) {
      const int nw = o->sizes[i + 1].first, nh = o->sizes[i + 1].second;
      hipLaunchKernelGGL(lines_resize, dim3((nw + 63) / 64, (nh + 3) / 4), block, 0, s, o->d_blur[i], w,
                         o->d_img[i + 1], nw, nh, o->d_xofs[i], o->d_alpha[i], o->d_yofs[i], o->d_beta[i]);
    }
  }
  PLVS_KERNEL_CHECK();
  // every octave is routed as soon as ITS maps have landed (the routing of octave 0 — the long pole of the host
  // stage — starts while the smaller octaves are still being computed and copied)
  PLVS_HIP_TRY(hipEventSynchronize(o->ev_route[0]));
  const double t1 = now_ms();
  // ---- host: EdgeDrawing is sequential per octave (one routing thread each); every finished
  // edge chain is fitted independently, so fitting threads consume batches of chains while
  // the routing is still running.  Results are concatenated in chain order (= the
  // reference's line order).
  o->det.resize(n);
  {
    constexpr int kBatch = kEdgeChainBatch;   // chains per fitting task (= the routing thread's publishing step)
    // extra fitting threads beside the routing ones, and whether an idle one yields its core instead of spinning
    const int kFitThreads = plvs::env_int("PLVS_HIP_LINES_FIT_THREADS", 3, 0, 16);
    const bool idle_yield = plvs::env_int("PLVS_HIP_LINES_FIT_YIELD", 0, 0, 1) != 0;
    struct OctaveWork {
      ChainProgress prog;
      std::atomic<int> grad{0};                   // 1: the octave's dx / dy images are known to have landed; -1: failed
      std::atomic<int> next{0};                   // next batch to claim
      std::vector<std::vector<Segment>> out;      // per batch
    };
    std::vector<OctaveWork> work(n);
    for (int i = 0; i < n; ++i) {
      o->det[i].progress = &work[i].prog;
      work[i].out.resize((size_t)o->sizes[i].first * o->sizes[i].second / 100 / kBatch + 2);
    }
    auto route = [&](int i) {
      if (i > 0 && hipEventSynchronize(o->ev_route[i]) != hipSuccess) {
        o->det[i].failed = true;
        work[i].prog.done.store(1, std::memory_order_release);
        return;
      }
      OctaveMaps m;
      m.w = o->sizes[i].first; m.h = o->sizes[i].second;
      m.gd = o->h_gd[i]; m.dx = o->h_dx[i]; m.dy = o->h_dy[i];
      m.anchors = (m.w >= 3 && m.h >= 3) ? o->h_anchor[i] : nullptr;
      // on "failure: lines extraction on octave i" the octave contributes no lines
      (void)o->det[i].prepare(m, o->ed);
    };
    // claims and fits ready batches until every octave is routed and drained
    auto fit = [&]() {
      for (;;) {
        bool all_drained = true, did = false;
        for (int i = 0; i < n; ++i) {
          OctaveWork& wk = work[i];
          const int done = wk.prog.done.load(std::memory_order_acquire);
          const int ready = wk.prog.ready.load(std::memory_order_acquire);
          int b = wk.next.load(std::memory_order_relaxed);
          const int c0 = b * kBatch;
          const bool full = ready >= c0 + kBatch, tail = done && ready > c0;
          if (full || tail) {
            all_drained = false;
            if (wk.next.compare_exchange_strong(b, b + 1)) {
              const int c1 = full ? c0 + kBatch : ready;
              if (wk.grad.load(std::memory_order_acquire) == 0)   // (the fit reads dx / dy: by now they are nearly always there)
                wk.grad.store(hipEventSynchronize(o->ev_maps[i]) == hipSuccess ? 1 : -1, std::memory_order_release);
              if (wk.grad.load(std::memory_order_acquire) > 0) o->det[i].fit_range(c0, c1, wk.out[b]);
              else o->det[i].failed = true;
              did = true;
            }
          } else if (!done) {
            all_drained = false;
          }
        }
        if (all_drained) break;
        if (!did) {   // nothing ready: leave the progress words alone for a moment
          if (idle_yield) std::this_thread::yield();
          else for (int k = 0; k < 64; ++k) __builtin_ia32_pause();
        }
      }
    };
    int device = 0;
    (void)hipGetDevice(&device);   // (the current device is per thread: the pool's threads take the caller's)
    const std::function<void(int)> job = [&](int j) {
      (void)hipSetDevice(device);
      if (j < n - 1) route(j + 1);
      fit();
    };
    o->pool.start(n - 1 + kFitThreads, &job);
    route(0);
    fit();
    o->pool.wait();
    PLVS_HIP_TRY(hipStreamSynchronize(o->copy_stream));   // (long done; the next frame's kernels overwrite what it read)
    for (int i = 0; i < n; ++i) {
      o->det[i].progress = nullptr;
      if (o->det[i].failed) continue;
      for (auto& v : work[i].out) o->det[i].segments.insert(o->det[i].segments.end(), v.begin(), v.end());
    }
  }
  const double t1a = t1 + o->det[0].ms_draw;   // octave 0's routing; fitting overlaps it
  const double t1b = now_ms();
  std::vector<KeyLine> kl = group_and_flatten(o->det, o->sizes, o->scale);
  select_lines(kl, o->nfeatures, o->img_w, o->img_h, o->min_length);
  const double t2 = now_ms();
  o->last_ms[3] = t1a - t1;   // EdgeDrawing
  o->last_ms[4] = t1b - t1a;  // line fitting + validation still running after the routing
  o->last_ms[5] = t2 - t1b;   // octave grouping + selection
  const int nl = (int)kl.size();
  *n_out = nl;
  if (nl == 0) {
    // "LineExtractor::detectLineFeatures() - no lines!" : empty outputs
    return PLVS_OK;
  }
  if (nl > o->line_cap) {
    plvs::set_error("lines: %d lines exceed the internal capacity %d", nl, o->line_cap);
    return PLVS_ERR_CAPACITY;
  }
  // ---- LBD on the device
  const short halfHeight = (kRows - 1) / 2;
  for (int i = 0; i < nl; ++i) {
    const KeyLine& k = kl[i];
    LbdLine& L = o->h_lines[i];
    L.octave = k.octave;
    L.num_pixels = (short)k.numOfPixels;
    const short halfWidth = (short)(((short)k.numOfPixels - 1) / 2);
    const float midx = (float)(0.5 * (k.sPointInOctaveX + k.ePointInOctaveX));
    const float midy = (float)(0.5 * (k.sPointInOctaveY + k.ePointInOctaveY));
    L.dL0 = (float)cos((double)k.angle);
    L.dL1 = (float)sin((double)k.angle);
    L.s0x = -L.dL0 * halfWidth + L.dL1 * halfHeight + midx;
    L.s0y = -L.dL1 * halfWidth - L.dL0 * halfHeight + midy;
  }
  LbdTables T;
  for (int i = 0; i < kRows; ++i) T.g[i] = o->gaussG[i];
  for (int i = 0; i < 3 * kBandWidth; ++i) T.l[i] = o->gaussL[i];
  for (int i = 0; i < kMaxOctaves; ++i) T.oct[i] = LbdOctave{nullptr, nullptr, 0, 0};
  for (int i = 0; i < n; ++i) T.oct[i] = LbdOctave{o->d_dx[i], o->d_dy[i], o->sizes[i].first, o->sizes[i].second};
  PLVS_HIP_TRY(hipMemcpyAsync(o->d_lines, o->h_lines, sizeof(LbdLine) * nl, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(lbd_kernel, dim3(nl), dim3(kLbdThreads), 0, s, o->d_lines, T, o->d_desc);
  PLVS_KERNEL_CHECK();
  PLVS_HIP_TRY(hipMemcpyAsync(o->h_desc, o->d_desc, (size_t)32 * nl, hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  if (nl <= cap) {
    static_assert(sizeof(KeyLine) == sizeof(plvs_keyline), "KeyLine layout");
    memcpy(keylines, kl.data(), sizeof(KeyLine) * nl);
    memcpy(desc, o->h_desc, (size_t)32 * nl);
  }
  const double t3 = now_ms();
  o->last_ms[0] = t1 - t0;   // device maps + D2H
  o->last_ms[1] = t2 - t1;   // host linking / fitting / grouping
  o->last_ms[2] = t3 - t2;   // LBD
  return PLVS_OK;
}

}  // namespace

namespace {

// Octave inputs from the shared ORB pyramid (Frame::PrecomputeGaussianPyramid, src/Frame.cc:841-865,
// USE_UNFILTERED_PYRAMID_FOR_LINES: the unblurred levels).  The copies are ordered after the
// extractor's pyramid kernels through its event; nothing leaves the device.
int lines_load_shared(plvs_lines* o, int w, int h) {
  plvs::OrbPyramidView v;
  if (!plvs::orb_pyramid_view(o->shared, &v)) {
    plvs::set_error("lines: the shared ORB extractor holds no pyramid yet");
    return PLVS_ERR_INVALID_ARG;
  }
  PLVS_REQUIRE(v.w[0] == w && v.h[0] == h, "image size differs from the shared pyramid's level 0");
  bool rebuild = !o->geometry_shared || w != o->img_w || h != o->img_h ||
                 o->nlevels != std::min(o->nlevels_cfg, v.nlevels);
  for (int i = 0; !rebuild && i < o->nlevels; ++i)
    rebuild = o->sizes[i].first != v.w[i] || o->sizes[i].second != v.h[i];
  if (rebuild) {
    const int rc = lines_build_geometry(o, w, h, &v);
    if (rc != PLVS_OK) return rc;
  }
  PLVS_HIP_TRY(hipStreamWaitEvent(o->stream, plvs::orb_pyramid_event(o->shared), 0));
  for (int i = 0; i < o->nlevels; ++i)
    PLVS_HIP_TRY(hipMemcpy2DAsync(o->d_img[i], v.w[i], v.level[i], v.pitch[i], v.w[i], v.h[i],
                                  hipMemcpyDeviceToDevice, o->stream));
  return PLVS_OK;
}

}  // namespace

extern "C" {

int plvs_hip_lines_create(int nfeatures, int nlevels, float scale_factor, double min_line_length,
                          double line_fit_err_threshold, plvs_lines** out) {
  PLVS_REQUIRE(out, "null output");
  PLVS_REQUIRE(nfeatures >= 0 && nlevels >= 1 && nlevels <= kMaxOctaves && scale_factor > 1.0f,
               "bad line extractor parameters");
  plvs_lines* o = new plvs_lines();
  o->nfeatures = nfeatures;
  o->nlevels = o->nlevels_cfg = nlevels;
  o->scale = o->scale_cfg = scale_factor;
  o->min_length = min_line_length;
  o->ed.fit_err_threshold = line_fit_err_threshold;
  // BinaryDescriptor ctor :248-274 (note the integer divisions)
  {
    double u = (kBandWidth * 3 - 1) / 2;
    double sigma = (kBandWidth * 2 + 1) / 2;
    double inv = -1 / (2 * sigma * sigma);
    for (int i = 0; i < kBandWidth * 3; ++i) { const double d = i - u; o->gaussL[i] = (float)exp(d * d * inv); }
    u = (kBands * kBandWidth - 1) / 2;
    sigma = u;
    inv = -1 / (2 * sigma * sigma);
    for (int i = 0; i < kRows; ++i) { const double d = i - u; o->gaussG[i] = (float)exp(d * d * inv); }
  }
  o->line_cap = 4096;
#define LN_TRY(call)                                                       \
  do {                                                                     \
    hipError_t _e = (call);                                                \
    if (_e != hipSuccess) {                                                \
      plvs::set_error("%s failed: %s", #call, hipGetErrorString(_e));     \
      plvs_hip_lines_destroy(o);                                           \
      return PLVS_ERR_HIP;                                                 \
    }                                                                      \
  } while (0)
  {
    // the line stage is the frame's critical path when points and lines are extracted side by side
    // (plvs_hip_frame_extract_dev): a few short kernels, then a long host stage — its stream goes first on the device
    int least = 0, greatest = 0;
    LN_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    LN_TRY(hipStreamCreateWithPriority(&o->stream, hipStreamNonBlocking, greatest));
    LN_TRY(hipStreamCreateWithPriority(&o->copy_stream, hipStreamNonBlocking, greatest));
  }
  LN_TRY(hipMalloc((void**)&o->d_lines, sizeof(LbdLine) * o->line_cap));
  LN_TRY(hipHostMalloc((void**)&o->h_lines, sizeof(LbdLine) * o->line_cap));
  LN_TRY(hipMalloc((void**)&o->d_desc, (size_t)32 * o->line_cap));
  LN_TRY(hipHostMalloc((void**)&o->h_desc, (size_t)32 * o->line_cap));
#undef LN_TRY
  *out = o;
  return PLVS_OK;
}

int plvs_hip_lines_destroy(plvs_lines* o) {
  if (!o) return PLVS_OK;
  lines_free_geometry(o);
  (void)hipFree(o->d_lines);
  (void)hipFree(o->d_desc);
  if (o->h_lines) (void)hipHostFree(o->h_lines);
  if (o->h_desc) (void)hipHostFree(o->h_desc);
  for (hipEvent_t e : o->ev_maps) (void)hipEventDestroy(e);
  for (hipEvent_t e : o->ev_route) (void)hipEventDestroy(e);
  if (o->copy_stream) (void)hipStreamDestroy(o->copy_stream);
  if (o->stream) (void)hipStreamDestroy(o->stream);
  delete o;
  return PLVS_OK;
}

int plvs_hip_lines_extract(plvs_lines* o, const uint8_t* image, int w, int h, int stride,
                           plvs_keyline* keylines, uint8_t* desc, int cap, int* n) {
  PLVS_REQUIRE(o && n, "null argument");
  *n = 0;
  if (!image || w <= 0 || h <= 0) {
    plvs::set_error("lines: empty image");
    return PLVS_ERR_EMPTY;
  }
  PLVS_REQUIRE(stride >= w, "stride smaller than width");
  PLVS_REQUIRE(w < 32768 && h < 32768, "image too large");
  if (o->shared != nullptr) {
    const int rc = lines_load_shared(o, w, h);
    return rc != PLVS_OK ? rc : lines_extract_body(o, keylines, desc, cap, n);
  }
  if (w != o->img_w || h != o->img_h || o->geometry_shared) {
    int rc = lines_build_geometry(o, w, h);
    if (rc != PLVS_OK) return rc;
  }
  for (int y = 0; y < h; ++y) memcpy(o->h_img + (size_t)y * w, image + (size_t)y * stride, w);
  PLVS_HIP_TRY(hipMemcpyAsync(o->d_img[0], o->h_img, (size_t)w * h, hipMemcpyHostToDevice, o->stream));
  return lines_extract_body(o, keylines, desc, cap, n);
}

int plvs_hip_lines_extract_dev(plvs_lines* o, const uint8_t* d_image, int w, int h, int stride,
                               plvs_keyline* keylines, uint8_t* desc, int cap, int* n) {
  PLVS_REQUIRE(o && n, "null argument");
  *n = 0;
  if (!d_image || w <= 0 || h <= 0) {
    plvs::set_error("lines: empty image");
    return PLVS_ERR_EMPTY;
  }
  PLVS_REQUIRE(stride >= w, "stride smaller than width");
  PLVS_REQUIRE(w < 32768 && h < 32768, "image too large");
  if (o->shared != nullptr) {
    const int rc = lines_load_shared(o, w, h);
    return rc != PLVS_OK ? rc : lines_extract_body(o, keylines, desc, cap, n);
  }
  if (w != o->img_w || h != o->img_h || o->geometry_shared) {
    int rc = lines_build_geometry(o, w, h);
    if (rc != PLVS_OK) return rc;
  }
  PLVS_HIP_TRY(hipMemcpy2DAsync(o->d_img[0], w, d_image, stride, w, h, hipMemcpyDeviceToDevice, o->stream));
  return lines_extract_body(o, keylines, desc, cap, n);
}

int plvs_hip_lines_set_gaussian_pyramid(plvs_lines* o, plvs_orb* orb) {
  PLVS_REQUIRE(o, "null handle");
  o->shared = orb;
  return PLVS_OK;
}

int plvs_hip_lines_last_stage_ms(plvs_lines* o, double* ms, int cap) {
  PLVS_REQUIRE(o && ms, "null argument");
  for (int i = 0; i < 6 && i < cap; ++i) ms[i] = o->last_ms[i];
  return PLVS_OK;
}

// Parity accessors: which = 0 blurred octave image (u8), 1 dx, 2 dy (s16), 3 packed
// gradient/direction map (u16), of the last call.
int plvs_hip_lines_octave_size(plvs_lines* o, int octave, int* w, int* h) {
  PLVS_REQUIRE(o && w && h && octave >= 0 && octave < o->nlevels && o->img_w > 0, "bad argument");
  *w = o->sizes[octave].first;
  *h = o->sizes[octave].second;
  return PLVS_OK;
}
int plvs_hip_lines_download_map(plvs_lines* o, int octave, int which, void* out) {
  PLVS_REQUIRE(o && out && octave >= 0 && octave < o->nlevels && o->img_w > 0, "bad argument");
  const size_t px = (size_t)o->sizes[octave].first * o->sizes[octave].second;
  const void* src = which == 0 ? (const void*)o->d_blur[octave]
                    : which == 1 ? (const void*)o->d_dx[octave]
                    : which == 2 ? (const void*)o->d_dy[octave] : (const void*)o->d_gd[octave];
  PLVS_HIP_TRY(hipMemcpy(out, src, which == 0 ? px : px * 2, hipMemcpyDeviceToHost));
  return PLVS_OK;
}
int plvs_hip_lines_num_in_octave(plvs_lines* o, int octave) {
  if (!o || octave < 0 || octave >= (int)o->det.size()) return 0;
  return (int)o->det[octave].segments.size();
}

}  // extern "C"

namespace plvs {
plvs_orb* lines_shared_orb(const plvs_lines* o) { return o ? o->shared : nullptr; }
}  // namespace plvs

#include "lsd_lines.inc"
