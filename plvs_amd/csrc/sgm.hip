// Dense stereo by semi-global matching — the disparity image PLVS obtains from libsgm for its stereo
// configurations (BASELINE config 5).
//
// Replaces sgm::StereoSGM(w, h, 64, 8, 8, ...)::execute as PointCloudKeyFrame::ProcessStereoLibsgm uses it
// (src/PointCloudKeyFrame.cc:435-481; Thirdparty/libsgm/src/stereo_sgm.cpp:133-181): 9 x 7 centre-symmetric
// census (census_transform.cu), the SGM recurrence over 8 paths with P1 / P2 (path_aggregation_common.hpp:45-92
// and the vertical / horizontal / oblique kernels), winner-takes-all on the summed costs with the uniqueness
// test for the left and the right view (winner_takes_all.cu), 3 x 3 median of both (median_filter.cu), left-right
// check on the left one (check_consistency.cu).  Every stage is integer arithmetic, so the result is the
// reference's whatever the thread layout; the layout here is gfx950's:
//
//   * 64 disparities = the 64 lanes of a wavefront.  One wave walks one path; lane d holds L_r(p, d), its
//     neighbours d -+ 1 come through one-lane shuffles, min_d through a wave reduction.  (libsgm spreads a
//     path over 4 or 8 threads of a 32-wide warp with 16 or 8 disparities each in registers.)
//   * the right-image features a pixel needs, right[x - d], are 64 consecutive words: one coalesced load per
//     step, issued eight steps ahead of the recurrence that consumes them.
//   * costs go out as bytes, 64 B per pixel and path, fully coalesced; winner-takes-all reads the eight
//     path volumes once, keeps the 16-bit sums for the right view, and both views take their best two
//     candidates with two wave reductions (the packed (cost, d) values are distinct).
#include <algorithm>

#include "common.hpp"

namespace {

constexpr int kDisp = 64;
constexpr int kPaths = 8;
constexpr int kBatch = 8;   // path steps whose loads are in flight together
constexpr int kSideStreams = 3;   // the path passes are spread over the call's stream and three more (see execute)

__global__ __launch_bounds__(256) void sgm_census(const uint8_t* __restrict__ src, int w, int h,
                                                  uint32_t* __restrict__ dst) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  uint32_t f = 0;
  if (x >= 4 && x < w - 4 && y >= 3 && y < h - 3) {   // census_transform.cu:71; the border keeps 0
    for (int dy = -3; dy < 0; ++dy)
#pragma unroll
      for (int dx = -4; dx <= 4; ++dx) {
        const uint8_t a = src[(size_t)(y + dy) * w + (x + dx)], b = src[(size_t)(y - dy) * w + (x - dx)];
        f = (f << 1) | (uint32_t)(a > b);
      }
#pragma unroll
    for (int dx = -4; dx < 0; ++dx) {
      const uint8_t a = src[(size_t)y * w + (x + dx)], b = src[(size_t)y * w + (x - dx)];
      f = (f << 1) | (uint32_t)(a > b);
    }
  }
  dst[(size_t)y * w + x] = f;
}

// Wave-wide minimum, the same value in every lane: four DPP steps fold each row of 16 lanes (xor 1, xor 2,
// mirror of 8, mirror of 16), the four row results are read as scalars.  No LDS crossbar traffic on the
// recurrence's critical path.
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, true));   // row_half_mirror
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xf, 0xf, true));   // row_mirror
  return min(min((uint32_t)__builtin_amdgcn_readlane((int)v, 0), (uint32_t)__builtin_amdgcn_readlane((int)v, 16)),
             min((uint32_t)__builtin_amdgcn_readlane((int)v, 32), (uint32_t)__builtin_amdgcn_readlane((int)v, 48)));
}
// the value of lane d - 1 / d + 1 (lane 0 / 63 keep their own: the callers do not use it there)
__device__ __forceinline__ uint32_t lane_prev(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xf, 0xf, true);   // wave_shr:1
}
__device__ __forceinline__ uint32_t lane_next(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x130, 0xf, 0xf, true);   // wave_shl:1
}

// One path per wave.  DX, DY: direction; path id -> where it enters the image.
//   horizontal (DY == 0): path = row, `n` = w steps
//   vertical   (DX == 0): path = column, `n` = h steps
//   oblique: path = diagonal x0 = id - (h - 1) (DX > 0) or id (DX < 0), h steps, pixels outside the image skipped
template <int DX, int DY>
__global__ __launch_bounds__(256) void sgm_path(const uint32_t* __restrict__ left, const uint32_t* __restrict__ right,
                                                int w, int h, uint32_t p1, uint32_t p2, uint8_t* __restrict__ dest) {
  const int lane = threadIdx.x & 63;
  const int path = blockIdx.x * 4 + (threadIdx.x >> 6);
  int npaths, nsteps;
  if (DY == 0) { npaths = h; nsteps = w; }
  else if (DX == 0) { npaths = w; nsteps = h; }
  else { npaths = w + h - 1; nsteps = h; }
  if (path >= npaths) return;
  uint32_t dp = 0, last_min = 0;   // DynamicProgramming(): a path starts from zero costs
  // position of step i on this path
  auto pos = [&](int i, int* x, int* y) {
    if (DY == 0) { *y = path; *x = DX > 0 ? i : w - 1 - i; }
    else if (DX == 0) { *x = path; *y = DY > 0 ? i : h - 1 - i; }
    else { *y = DY > 0 ? i : h - 1 - i; *x = (DX > 0 ? path - (h - 1) : path) + i * DX; }
  };
  // features of the kBatch steps from i0 on: the left one (the wave's pixel) and right[x - d] per lane
  auto fetch = [&](int i0, uint32_t (&fl)[kBatch], uint32_t (&fr)[kBatch]) {
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      int x, y;
      pos(i0 + k, &x, &y);
      fl[k] = 0;
      fr[k] = 0;
      if (i0 + k < nsteps && x >= 0 && x < w) {
        fl[k] = left[(size_t)y * w + x];
        if (x - lane >= 0) fr[k] = right[(size_t)y * w + (x - lane)];   // beyond the left border: feature 0
      }
    }
  };
  uint32_t fl[kBatch], fr[kBatch], nl[kBatch], nr[kBatch];
  uint32_t done[kBatch];   // costs of the previous batch, stored at the start of the next one
  bool done_ok[kBatch];
#pragma unroll
  for (int k = 0; k < kBatch; ++k) { done[k] = 0; done_ok[k] = false; }
  fetch(0, fl, fr);
  // Loads and stores share one in-order counter (vmcnt).  Within a batch the order is: stores of the previous
  // batch, loads of the next one, then eight steps of pure register work — so the only wait is for loads that
  // have had a whole batch to arrive, never for a store.  The first batch is waited for here, outside the loop.
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  for (int i0 = 0; i0 < nsteps + kBatch; i0 += kBatch) {
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      if (done_ok[k]) {
        int x, y;
        pos(i0 - kBatch + k, &x, &y);
        dest[((size_t)y * w + x) * kDisp + lane] = (uint8_t)done[k];
      }
    }
    if (i0 >= nsteps) break;
    fetch(i0 + kBatch, nl, nr);   // the next batch is in flight while this one runs through the recurrence
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      int x, y;
      pos(i0 + k, &x, &y);
      const bool valid = i0 + k < nsteps && x >= 0 && x < w;   // uniform: the position is the wave's
      const uint32_t cost = (uint32_t)__popc(fl[k] ^ fr[k]);
      // DynamicProgramming::update
      uint32_t out = min(dp - last_min, p2);
      const uint32_t prev = lane_prev(dp), next = lane_next(dp);
      if (lane != 0) out = min(out, prev - last_min + p1);
      if (lane != kDisp - 1) out = min(out, next - last_min + p1);
      const uint32_t ndp = out + cost;
      const uint32_t nmin = wave_min_u32(ndp);
      dp = valid ? ndp : dp;
      last_min = valid ? nmin : last_min;
      done[k] = ndp;
      done_ok[k] = valid;
    }
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      fl[k] = nl[k];
      fr[k] = nr[k];
    }
  }
}

__device__ __forceinline__ uint32_t compute_disparity(uint32_t v0, uint32_t v1, float uniqueness) {
  const float cost0 = (float)(v0 >> 16), cost1 = (float)(v1 >> 16);
  const int disp0 = (int)(v0 & 0xffffu), disp1 = (int)(v1 & 0xffffu);
  if (cost1 * uniqueness >= cost0) return (uint32_t)disp0;
  if (abs(disp1 - disp0) <= 1) return (uint32_t)disp0;
  return 0u;
}

// the two smallest of 64 distinct packed values, one per lane
__device__ __forceinline__ void wave_top2(uint32_t packed, uint32_t* v0, uint32_t* v1) {
  *v0 = wave_min_u32(packed);
  *v1 = wave_min_u32(packed == *v0 ? 0xffffffffu : packed);
}

// left view: sum of the path volumes, best two disparities; the sums are kept for the right view
__global__ __launch_bounds__(256) void sgm_wta_left(const uint8_t* __restrict__ cost, int npix, float uniqueness,
                                                    uint16_t* __restrict__ sum, uint8_t* __restrict__ left_disp) {
  const int lane = threadIdx.x & 63;
  const size_t step = (size_t)npix * kDisp;
  for (int p = blockIdx.x * 4 + (threadIdx.x >> 6); p < npix; p += gridDim.x * 4) {
    const size_t o = (size_t)p * kDisp + lane;
    uint32_t s = 0;
#pragma unroll
    for (int r = 0; r < kPaths; ++r) s += cost[r * step + o];
    sum[o] = (uint16_t)s;
    uint32_t v0, v1;
    wave_top2((s << 16) | (uint32_t)lane, &v0, &v1);
    if (lane == 0) left_disp[p] = (uint8_t)compute_disparity(v0, v1, uniqueness);
  }
}

// right view: pixel p sees the left pixels p + d at disparity d (winner_takes_all.cu:196-236)
__global__ __launch_bounds__(256) void sgm_wta_right(const uint16_t* __restrict__ sum, int w, int h, float uniqueness,
                                                     uint8_t* __restrict__ right_disp) {
  const int lane = threadIdx.x & 63;
  const int npix = w * h;
  for (int p = blockIdx.x * 4 + (threadIdx.x >> 6); p < npix; p += gridDim.x * 4) {
    const int y = p / w, x = p - y * w;
    uint32_t packed = 0xffffffffu;
    if (x + lane < w) packed = ((uint32_t)sum[((size_t)y * w + (x + lane)) * kDisp + lane] << 16) | (uint32_t)lane;
    uint32_t v0, v1;
    wave_top2(packed, &v0, &v1);
    // with fewer than two candidates the second stays 0xffffffff, as the reference's initial value
    if (lane == 0) right_disp[p] = (uint8_t)compute_disparity(v0, v1, uniqueness);
  }
}

__device__ __forceinline__ void sort2(uint8_t& a, uint8_t& b) {
  const uint8_t lo = a < b ? a : b, hi = a < b ? b : a;
  a = lo;
  b = hi;
}

__global__ __launch_bounds__(256) void sgm_median(const uint8_t* __restrict__ src, int w, int h, uint8_t* __restrict__ dst) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  uint8_t m = 0;   // the border is never written by the reference (buffers cleared at allocation)
  if (x >= 1 && x < w - 1 && y >= 1 && y < h - 1) {
    uint8_t b[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) b[i] = src[(size_t)(y - 1 + i / 3) * w + (x - 1 + i % 3)];
    // median of nine by a full sorting network of the three rows and columns (exact median)
    sort2(b[0], b[1]); sort2(b[3], b[4]); sort2(b[6], b[7]);
    sort2(b[1], b[2]); sort2(b[4], b[5]); sort2(b[7], b[8]);
    sort2(b[0], b[1]); sort2(b[3], b[4]); sort2(b[6], b[7]);
    const uint8_t lo = max(max(b[0], b[3]), b[6]);          // largest of the row minima
    const uint8_t hi = min(min(b[2], b[5]), b[8]);          // smallest of the row maxima
    uint8_t m0 = b[1], m1 = b[4], m2 = b[7];                // median of the row medians
    sort2(m0, m1); sort2(m1, m2); sort2(m0, m1);
    uint8_t a0 = lo, a1 = m1, a2 = hi;
    sort2(a0, a1); sort2(a1, a2); sort2(a0, a1);
    m = a1;
  }
  dst[(size_t)y * w + x] = m;
}

__global__ __launch_bounds__(256) void sgm_check(const uint8_t* __restrict__ left_img, const uint8_t* __restrict__ ml,
                                                 const uint8_t* __restrict__ mr, int w, int h, uint8_t* __restrict__ out) {
  const int j = blockIdx.x * 64 + (threadIdx.x & 63), i = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (j >= w || i >= h) return;
  int d = ml[(size_t)i * w + j];
  // check_consistency.cu launches (w / 16) x (h / 16) blocks of 16 x 16: the remainder columns / rows are not checked
  if (j < (w / 16) * 16 && i < (h / 16) * 16) {
    const int k = j - d;
    if (left_img[(size_t)i * w + j] == 0 || d <= 0 || (k >= 0 && k < w && abs((int)mr[(size_t)i * w + k] - d) > 1)) d = 0;
  }
  out[(size_t)i * w + j] = (uint8_t)d;
}

}  // namespace

struct plvs_sgm {
  int w = 0, h = 0, p1 = 10, p2 = 120;
  float uniqueness = 0.95f;
  hipStream_t stream = nullptr;
  hipStream_t path_stream[kSideStreams] = {};   // the eight path passes are independent: they run side by side
  hipEvent_t ev_census = nullptr, ev_path[kSideStreams] = {};
  uint8_t *d_left = nullptr, *d_right = nullptr, *d_cost = nullptr, *d_dl = nullptr, *d_dr = nullptr, *d_ml = nullptr,
          *d_mr = nullptr, *d_out = nullptr;
  uint32_t *d_cl = nullptr, *d_cr = nullptr;
  uint16_t* d_sum = nullptr;
};

extern "C" {

int plvs_hip_sgm_destroy(plvs_sgm* s) {
  if (s == nullptr) return PLVS_OK;
  void* bufs[] = {s->d_left, s->d_right, s->d_cost, s->d_dl, s->d_dr, s->d_ml, s->d_mr, s->d_out, s->d_cl, s->d_cr, s->d_sum};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  for (int i = 0; i < kSideStreams; ++i) {
    if (s->path_stream[i]) (void)hipStreamDestroy(s->path_stream[i]);
    if (s->ev_path[i]) (void)hipEventDestroy(s->ev_path[i]);
  }
  if (s->ev_census) (void)hipEventDestroy(s->ev_census);
  delete s;
  return PLVS_OK;
}

int plvs_hip_sgm_create(int width, int height, int disparity_size, int p1, int p2, float uniqueness, plvs_sgm** out) {
  PLVS_REQUIRE(out != nullptr, "out is null");
  PLVS_REQUIRE(width >= 16 && height >= 16 && width <= 16384 && height <= 16384, "image size");
  PLVS_REQUIRE(disparity_size == kDisp, "disparity_size must be 64 (what PLVS passes; 128 is not built)");
  PLVS_REQUIRE(p1 >= 0 && p2 >= p1 && p2 + 31 < 256, "penalties: 0 <= P1 <= P2 and P2 + 31 must fit a byte");
  plvs_sgm* s = new plvs_sgm();
  s->w = width;
  s->h = height;
  s->p1 = p1;
  s->p2 = p2;
  s->uniqueness = uniqueness;
  const size_t n = (size_t)width * height;
  hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
  for (int i = 0; i < kSideStreams && e == hipSuccess; ++i) {
    e = hipStreamCreateWithFlags(&s->path_stream[i], hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_path[i], hipEventDisableTiming);
  }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_census, hipEventDisableTiming);
  auto alloc = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
  alloc((void**)&s->d_left, n);
  alloc((void**)&s->d_right, n);
  alloc((void**)&s->d_cl, n * 4);
  alloc((void**)&s->d_cr, n * 4);
  alloc((void**)&s->d_cost, n * kDisp * kPaths);
  alloc((void**)&s->d_sum, n * kDisp * 2);
  alloc((void**)&s->d_dl, n);
  alloc((void**)&s->d_dr, n);
  alloc((void**)&s->d_ml, n);
  alloc((void**)&s->d_mr, n);
  alloc((void**)&s->d_out, n);
  if (e != hipSuccess) {
    plvs::set_error("sgm_create: %s", hipGetErrorString(e));
    plvs_hip_sgm_destroy(s);
    return PLVS_ERR_HIP;
  }
  *out = s;
  return PLVS_OK;
}

// d_left / d_right / d_disparity: device images (w x h u8, tightly packed); asynchronous on `stream`.
int plvs_hip_sgm_execute_dev(plvs_sgm* s, const uint8_t* d_left, const uint8_t* d_right, uint8_t* d_disparity,
                             void* stream) {
  PLVS_REQUIRE(s && d_left && d_right && d_disparity, "null argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int w = s->w, h = s->h;
  const size_t n = (size_t)w * h, step = n * kDisp;
  const dim3 grid2d(plvs::ceil_div((size_t)w, 64), plvs::ceil_div((size_t)h, 4)), block(256);
  sgm_census<<<grid2d, block, 0, st>>>(d_left, w, h, s->d_cl);
  sgm_census<<<grid2d, block, 0, st>>>(d_right, w, h, s->d_cr);
  const uint32_t p1 = (uint32_t)s->p1, p2 = (uint32_t)s->p2;
  const unsigned gv = plvs::ceil_div((size_t)w, 4), gh = plvs::ceil_div((size_t)h, 4), go = plvs::ceil_div((size_t)(w + h - 1), 4);
  // A path pass is a chain of dependent steps with at most a wave or two per SIMD: alone it leaves the machine
  // idle.  The eight passes write disjoint volumes, so they run side by side (libsgm does the same,
  // path_aggregation.cu:56-83) — on FOUR streams, because the runtime feeds the device through four hardware
  // queues and streams beyond that share one (measured with eight: a queue got both 1240-step horizontal passes
  // and a third pass, 605 of the stage's 616 us).  The passes are dealt by their measured lengths: the longest
  // horizontal pass alone on the call's own stream, the other three streams about equal.
  PLVS_HIP_TRY(hipEventRecord(s->ev_census, st));
  for (int i = 0; i < kSideStreams; ++i) PLVS_HIP_TRY(hipStreamWaitEvent(s->path_stream[i], s->ev_census, 0));
  hipStream_t q1 = s->path_stream[0], q2 = s->path_stream[1], q3 = s->path_stream[2];
  sgm_path<-1, 0><<<gh, block, 0, st>>>(s->d_cl, s->d_cr, w, h, p1, p2, s->d_cost + 3 * step);
  sgm_path<1, 0><<<gh, block, 0, q1>>>(s->d_cl, s->d_cr, w, h, p1, p2, s->d_cost + 2 * step);
  sgm_path<-1, 1><<<go, block, 0, q2>>>(s->d_cl, s->d_cr, w, h, p1, p2, s->d_cost + 5 * step);
  sgm_path<0, -1><<<gv, block, 0, q3>>>(s->d_cl, s->d_cr, w, h, p1, p2, s->d_cost + 1 * step);
  sgm_path<1, -1><<<go, block, 0, q1>>>(s->d_cl, s->d_cr, w, h, p1, p2, s->d_cost + 7 * step);
  sgm_path<1, 1><<<go, block, 0, q2>>>(s->d_cl, s->d_cr, w, h, p1, p2, s->d_cost + 4 * step);
  sgm_path<0, 1><<<gv, block, 0, q3>>>(s->d_cl, s->d_cr, w, h, p1, p2, s->d_cost + 0 * step);
  sgm_path<-1, -1><<<go, block, 0, q3>>>(s->d_cl, s->d_cr, w, h, p1, p2, s->d_cost + 6 * step);
  PLVS_KERNEL_CHECK();
  for (int i = 0; i < kSideStreams; ++i) {
    PLVS_HIP_TRY(hipEventRecord(s->ev_path[i], s->path_stream[i]));
    PLVS_HIP_TRY(hipStreamWaitEvent(st, s->ev_path[i], 0));
  }
  const unsigned gw = (unsigned)std::min<size_t>(plvs::ceil_div(n, 4), 65536);
  sgm_wta_left<<<gw, block, 0, st>>>(s->d_cost, (int)n, s->uniqueness, s->d_sum, s->d_dl);
  sgm_wta_right<<<gw, block, 0, st>>>(s->d_sum, w, h, s->uniqueness, s->d_dr);
  sgm_median<<<grid2d, block, 0, st>>>(s->d_dl, w, h, s->d_ml);
  sgm_median<<<grid2d, block, 0, st>>>(s->d_dr, w, h, s->d_mr);
  sgm_check<<<grid2d, block, 0, st>>>(d_left, s->d_ml, s->d_mr, w, h, d_disparity);
  PLVS_KERNEL_CHECK();
  return PLVS_OK;
}

int plvs_hip_sgm_execute(plvs_sgm* s, const uint8_t* left, const uint8_t* right, uint8_t* disparity) {
  PLVS_REQUIRE(s && left && right && disparity, "null argument");
  const size_t n = (size_t)s->w * s->h;
  PLVS_HIP_TRY(hipMemcpyAsync(s->d_left, left, n, hipMemcpyHostToDevice, s->stream));
  PLVS_HIP_TRY(hipMemcpyAsync(s->d_right, right, n, hipMemcpyHostToDevice, s->stream));
  const int rc = plvs_hip_sgm_execute_dev(s, s->d_left, s->d_right, s->d_out, s->stream);
  if (rc != PLVS_OK) return rc;
  PLVS_HIP_TRY(hipMemcpyAsync(disparity, s->d_out, n, hipMemcpyDeviceToHost, s->stream));
  PLVS_HIP_TRY(hipStreamSynchronize(s->stream));
  return PLVS_OK;
}

// Parity accessors of the last call: which = 0 census left, 1 census right (u32), 2 cost sums (u16, w*h*64),
// 3 / 4 raw left / right disparity, 5 / 6 median-filtered left / right (u8).
int plvs_hip_sgm_download(plvs_sgm* s, int which, void* out) {
  PLVS_REQUIRE(s && out && which >= 0 && which <= 6, "bad argument");
  const size_t n = (size_t)s->w * s->h;
  const void* src[] = {s->d_cl, s->d_cr, s->d_sum, s->d_dl, s->d_dr, s->d_ml, s->d_mr};
  const size_t bytes[] = {n * 4, n * 4, n * kDisp * 2, n, n, n, n};
  PLVS_HIP_TRY(hipMemcpy(out, src[which], bytes[which], hipMemcpyDeviceToHost));
  return PLVS_OK;
}

}  // extern "C"
