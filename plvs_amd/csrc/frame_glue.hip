// The per-frame glue between extraction and the searches: Frame::UndistortKeyPoints
// (reference src/Frame.cc:1507-1552), Frame::UndistortKeyLines (:1555-1700, pinhole branch), Frame::ComputeImageBounds
// (:1749-1778) and Frame::AssignFeaturesToGrid (:716-746, the key-point grid) — what Frame::Frame runs between
// ExtractORB / ExtractLSD and the first SearchByProjection.  One launch each; host flavours (the reference's Frame holds
// host vectors), staged through the calling thread's pinned block like the searches.
//
// Arithmetic: cv::undistortPoints is OpenCV's (calib3d/undistort.dispatch.cpp, cvUndistortPointsInternal with the default
// criteria: five fixed-point iterations of the inverse distortion, all in double) — restated from the published
// algorithm like the other OpenCV primitives of this repository (unverified against a real OpenCV: there is none here);
// cv::fastAtan2 as in orb.hip.  The translation unit is built with -ffp-contract=off: every double operation below
// rounds like the host's.
#include <cmath>
#include <cstring>

#include "common.hpp"

namespace {

constexpr int kGridCols = 64, kGridRows = 48;   // FRAME_GRID_COLS / ROWS, include/Frame.h:67-68

struct Calib {
  double fx, fy, cx, cy;   // cameraMatrix (K)
  double k[8];             // k1 k2 p1 p2 k3 k4 k5 k6 (absent ones zero)
  double pfx, pfy, pcx, pcy;   // P (the "linear" K: the same matrix for a pinhole camera)
};

__device__ __forceinline__ float2 undistort_point(const Calib& c, float xin, float yin) {
  const double ifx = 1. / c.fx, ify = 1. / c.fy;
  double x = (double)xin, y = (double)yin;
  x = (x - c.cx) * ifx;
  y = (y - c.cy) * ify;
  const double x0 = x, y0 = y;
  for (int j = 0; j < 5; ++j) {
    const double r2 = x * x + y * y;
    const double icdist = (1 + ((c.k[7] * r2 + c.k[6]) * r2 + c.k[5]) * r2) / (1 + ((c.k[4] * r2 + c.k[1]) * r2 + c.k[0]) * r2);
    if (icdist < 0) {   // (OpenCV: give up, keep the normalised input)
      x = ((double)xin - c.cx) * ifx;
      y = ((double)yin - c.cy) * ify;
      break;
    }
    const double deltaX = 2 * c.k[2] * x * y + c.k[3] * (r2 + 2 * x * x);   // (+ the thin-prism terms: zero coefficients)
    const double deltaY = c.k[2] * (r2 + 2 * y * y) + 2 * c.k[3] * x * y;
    x = (x0 - deltaX) * icdist;
    y = (y0 - deltaY) * icdist;
  }
  // P * (x, y, 1) with P = [pfx 0 pcx; 0 pfy pcy; 0 0 1]: xx = pfx x + 0 y + pcx, ww = 1 / (0 x + 0 y + 1)
  const double xx = c.pfx * x + 0.0 * y + c.pcx;
  const double yy = 0.0 * x + c.pfy * y + c.pcy;
  const double ww = 1. / (0.0 * x + 0.0 * y + 1.0);
  return make_float2((float)(xx * ww), (float)(yy * ww));
}

__global__ __launch_bounds__(256) void undistort_points_kernel(Calib c, const float2* __restrict__ in, int n, float2* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = undistort_point(c, in[i].x, in[i].y);
}

// cv::fastAtan2 (degrees), mathfuncs_core.simd.hpp atan_f32; plain IEEE f32 ops (as orb.hip's).
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float scale = (float)(180 / 3.1415926535897932384626433832795);
  const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale, p5 = 0.1555786518463281f * scale,
              p7 = -0.04432655554792128f * scale;
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

// Frame::UndistortKeyLines, one thread per line: both end points undistorted, the angle from them, the bounds test
// (:1631-1648); kept[i] = 1 when the line stays.
__global__ __launch_bounds__(256) void undistort_keylines_kernel(Calib c, const plvs_keyline* __restrict__ in, int n, float min_x,
                                                                 float max_x, float min_y, float max_y,
                                                                 plvs_keyline* __restrict__ out, uint8_t* __restrict__ kept) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  plvs_keyline kl = in[i];
  const float2 s = undistort_point(c, kl.startPointX, kl.startPointY), e = undistort_point(c, kl.endPointX, kl.endPointY);
  kl.startPointX = s.x; kl.startPointY = s.y;
  kl.endPointX = e.x; kl.endPointY = e.y;
  const float deg2rad = (float)(M_PI / 180.0f);   // constexpr float DEG2RAD = M_PI/180.0f
  kl.angle = fast_atan2_deg(kl.endPointY - kl.startPointY, kl.endPointX - kl.startPointX) * deg2rad;
  const bool out_of_bounds = kl.startPointX < min_x || kl.startPointX >= max_x || kl.startPointY < min_y || kl.startPointY >= max_y ||
                             kl.endPointX < min_x || kl.endPointX >= max_x || kl.endPointY < min_y || kl.endPointY >= max_y;
  out[i] = kl;
  kept[i] = out_of_bounds ? 0 : 1;
}

// Frame::AssignFeaturesToGrid, the key-point grid: cell lists in key-point order (push_back), as one CSR — cell =
// column * 48 + row (mGrid[ix][iy]).  One workgroup: the cells of 4096 key points at a time are counted in LDS; a key
// point's place inside its cell = the key points of the cell before it, which a stable two-pass count gives without a sort:
// pass 1 counts per cell, a scan places the cells, pass 2 walks the key points IN ORDER in chunks of the workgroup's size and
// ranks a chunk's members of a cell by their thread index (ballot-free: a lane's rank = lanes of lower index with its cell,
// found by comparing against the chunk's cells kept in LDS — chunks are 1024 key points, cells rarely hold more than a few).
constexpr int kCells = kGridCols * kGridRows;
__global__ __launch_bounds__(1024) void assign_grid_kernel(const float2* __restrict__ xy, int n, float min_x, float min_y, float inv_w,
                                                           float inv_h, int32_t* __restrict__ cell_start, int32_t* __restrict__ items) {
  __shared__ int cnt[kCells + 1];
  __shared__ int chunk_cell[1024];
  __shared__ int wsum[16];
  const int tid = threadIdx.x;
  auto cell_of = [&](int i) {
    // PosInGrid (:1305-1316): round() of a float product — std::round, half away from zero
    const int px = (int)roundf((xy[i].x - min_x) * inv_w), py = (int)roundf((xy[i].y - min_y) * inv_h);
    return (px < 0 || px >= kGridCols || py < 0 || py >= kGridRows) ? -1 : px * kGridRows + py;
  };
  for (int c = tid; c <= kCells; c += 1024) cnt[c] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += 1024) {
    const int c = cell_of(i);
    if (c >= 0) atomicAdd(&cnt[c], 1);
  }
  __syncthreads();
  // exclusive scan of the 3072 counters: three per thread
  int v[3], sum = 0;
  for (int k = 0; k < 3; ++k) {
    v[k] = cnt[3 * tid + k];
    sum += v[k];
  }
  int inc = sum;
  for (int off = 1; off < 64; off <<= 1) {
    const int up = __shfl_up(inc, off);
    if ((tid & 63) >= off) inc += up;
  }
  if ((tid & 63) == 63) wsum[tid >> 6] = inc;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
  int run = base + inc - sum;
  __syncthreads();
  for (int k = 0; k < 3; ++k) {
    cnt[3 * tid + k] = run;   // now: where the cell's next member goes
    cell_start[3 * tid + k] = run;
    run += v[k];
  }
  if (tid == 1023) cell_start[kCells] = run;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 1024) {
    const int i = i0 + tid;
    const int c = i < n ? cell_of(i) : -1;
    chunk_cell[tid] = c;
    __syncthreads();
    int before = 0, total = 0;
    if (c >= 0) {
      for (int j = 0; j < 1024 && i0 + j < n; ++j) {
        const bool same = chunk_cell[j] == c;
        before += (same && j < tid) ? 1 : 0;
        total += same ? 1 : 0;
      }
      items[cnt[c] + before] = i;
    }
    __syncthreads();   // every lane of the chunk has read cnt before anybody moves it (barriers outside the branch: uniform)
    if (c >= 0 && before == 0) cnt[c] += total;
    __syncthreads();
  }
}

int make_calib(const float* K4, const float* dist, int ndist, Calib* c) {
  PLVS_REQUIRE(K4 && (ndist == 0 || dist), "null calibration");
  PLVS_REQUIRE(ndist == 0 || ndist == 4 || ndist == 5 || ndist == 8, "4, 5 or 8 distortion coefficients (k1 k2 p1 p2 [k3 [k4 k5 k6]])");
  c->fx = K4[0]; c->fy = K4[1]; c->cx = K4[2]; c->cy = K4[3];
  c->pfx = K4[0]; c->pfy = K4[1]; c->pcx = K4[2]; c->pcy = K4[3];
  for (int i = 0; i < 8; ++i) c->k[i] = i < ndist ? (double)dist[i] : 0.0;
  return PLVS_OK;
}

int undistort_xy(const Calib& c, const float* xy, int n, float* xy_out) {
  if (n == 0) return PLVS_OK;
  plvs::HostStage& st = plvs::thread_stage();
  const size_t bytes = sizeof(float) * 2 * (size_t)n, o_out = (bytes + 15) & ~(size_t)15;
  PLVS_HIP_TRY(st.reserve(o_out + bytes));
  memcpy(st.pinned, xy, bytes);
  PLVS_HIP_TRY(hipMemcpyAsync(st.dev, st.pinned, bytes, hipMemcpyHostToDevice, st.stream));
  hipLaunchKernelGGL(undistort_points_kernel, dim3(plvs::ceil_div((size_t)n, 256)), dim3(256), 0, st.stream, c,
                     reinterpret_cast<const float2*>(st.dev), n, reinterpret_cast<float2*>(st.dev + o_out));
  PLVS_KERNEL_CHECK();
  PLVS_HIP_TRY(hipMemcpyAsync(st.pinned + o_out, st.dev + o_out, bytes, hipMemcpyDeviceToHost, st.stream));
  PLVS_HIP_TRY(hipStreamSynchronize(st.stream));
  memcpy(xy_out, st.pinned + o_out, bytes);
  return PLVS_OK;
}

}  // namespace

extern "C" {

int plvs_hip_frame_undistort_points(const float* xy, int n, const float* K4, const float* dist, int ndist, float* xy_out) {
  PLVS_REQUIRE(n >= 0 && (n == 0 || (xy && xy_out)), "bad arguments");
  Calib c;
  const int rc = make_calib(K4, dist, ndist, &c);
  if (rc != PLVS_OK) return rc;
  return undistort_xy(c, xy, n, xy_out);
}

int plvs_hip_frame_undistort_keypoints(const plvs_keypoint* kps, int n, const float* K4, const float* dist, int ndist,
                                       plvs_keypoint* kps_un) {
  PLVS_REQUIRE(n >= 0 && (n == 0 || (kps && kps_un)), "bad arguments");
  if (n == 0) return PLVS_OK;
  if (kps_un != kps) memcpy(kps_un, kps, sizeof(plvs_keypoint) * (size_t)n);
  if (ndist == 0 || dist[0] == 0.0f) return PLVS_OK;   // mDistCoef.at<float>(0) == 0.0: mvKeysUn = mvKeys (:1510-1514)
  Calib c;
  const int rc = make_calib(K4, dist, ndist, &c);
  if (rc != PLVS_OK) return rc;
  std::vector<float> xy(2 * (size_t)n), out(2 * (size_t)n);
  for (int i = 0; i < n; ++i) {
    xy[2 * i] = kps[i].x;
    xy[2 * i + 1] = kps[i].y;
  }
  const int rc2 = undistort_xy(c, xy.data(), n, out.data());
  if (rc2 != PLVS_OK) return rc2;
  for (int i = 0; i < n; ++i) {
    kps_un[i].x = out[2 * i];
    kps_un[i].y = out[2 * i + 1];
  }
  return PLVS_OK;
}

int plvs_hip_frame_compute_image_bounds(int width, int height, const float* K4, const float* dist, int ndist, float* bounds5) {
  PLVS_REQUIRE(bounds5 && width > 0 && height > 0, "bad arguments");
  float mnx = 0.0f, mxx = (float)width, mny = 0.0f, mxy = (float)height;
  if (ndist > 0 && dist[0] != 0.0f) {
    Calib c;
    const int rc = make_calib(K4, dist, ndist, &c);
    if (rc != PLVS_OK) return rc;
    const float corners[8] = {0.0f, 0.0f, (float)width, 0.0f, 0.0f, (float)height, (float)width, (float)height};
    float m[8];
    const int rc2 = undistort_xy(c, corners, 4, m);
    if (rc2 != PLVS_OK) return rc2;
    mnx = std::min(m[0], m[4]);   // min(mat(0,0), mat(2,0))
    mxx = std::max(m[2], m[6]);   // max(mat(1,0), mat(3,0))
    mny = std::min(m[1], m[3]);   // min(mat(0,1), mat(1,1))
    mxy = std::max(m[5], m[7]);   // max(mat(2,1), mat(3,1))
  }
  bounds5[0] = mnx; bounds5[1] = mxx; bounds5[2] = mny; bounds5[3] = mxy;
  bounds5[4] = (float)std::sqrt(std::pow(mxx - mnx, 2) + std::pow(mxy - mny, 2));   // mnMaxDiag: float operands, double pow / sqrt
  return PLVS_OK;
}

int plvs_hip_frame_undistort_keylines(const plvs_keyline* keylines, int n, const float* K4, const float* dist, int ndist,
                                      const float* bounds4, plvs_keyline* keylines_un, int32_t* kept_index, int* n_kept) {
  PLVS_REQUIRE(n >= 0 && n_kept && (n == 0 || (keylines && keylines_un && kept_index)), "bad arguments");
  *n_kept = 0;
  if (n == 0) return PLVS_OK;
  if (ndist == 0 || dist[0] == 0.0f) {   // mvKeyLinesUn = mvKeyLines (:1560-1565)
    memcpy(keylines_un, keylines, sizeof(plvs_keyline) * (size_t)n);
    for (int i = 0; i < n; ++i) kept_index[i] = i;
    *n_kept = n;
    return PLVS_OK;
  }
  PLVS_REQUIRE(bounds4, "null bounds");
  Calib c;
  const int rc = make_calib(K4, dist, ndist, &c);
  if (rc != PLVS_OK) return rc;
  plvs::HostStage& st = plvs::thread_stage();
  const size_t b_in = sizeof(plvs_keyline) * (size_t)n, o_out = (b_in + 15) & ~(size_t)15, o_kept = o_out + ((b_in + 15) & ~(size_t)15);
  PLVS_HIP_TRY(st.reserve(o_kept + (size_t)n + 16));
  memcpy(st.pinned, keylines, b_in);
  PLVS_HIP_TRY(hipMemcpyAsync(st.dev, st.pinned, b_in, hipMemcpyHostToDevice, st.stream));
  hipLaunchKernelGGL(undistort_keylines_kernel, dim3(plvs::ceil_div((size_t)n, 256)), dim3(256), 0, st.stream, c,
                     reinterpret_cast<const plvs_keyline*>(st.dev), n, bounds4[0], bounds4[1], bounds4[2], bounds4[3],
                     reinterpret_cast<plvs_keyline*>(st.dev + o_out), reinterpret_cast<uint8_t*>(st.dev + o_kept));
  PLVS_KERNEL_CHECK();
  PLVS_HIP_TRY(hipMemcpyAsync(st.pinned + o_out, st.dev + o_out, o_kept - o_out + (size_t)n, hipMemcpyDeviceToHost, st.stream));
  PLVS_HIP_TRY(hipStreamSynchronize(st.stream));
  const plvs_keyline* un = reinterpret_cast<const plvs_keyline*>(st.pinned + o_out);
  const uint8_t* kept = reinterpret_cast<const uint8_t*>(st.pinned + o_kept);
  int m = 0;
  for (int i = 0; i < n; ++i)
    if (kept[i]) {   // (the reference's loop: lines out of the undistorted bounds are dropped, the others keep their order)
      keylines_un[m] = un[i];
      kept_index[m] = i;
      ++m;
    }
  *n_kept = m;
  return PLVS_OK;
}

int plvs_hip_frame_assign_features_to_grid(const plvs_keypoint* kps_un, int n, float min_x, float min_y, float grid_w_inv,
                                           float grid_h_inv, int32_t* cell_start, int32_t* cell_items, int* n_items) {
  PLVS_REQUIRE(n >= 0 && cell_start && n_items && (n == 0 || (kps_un && cell_items)), "bad arguments");
  *n_items = 0;
  if (n == 0) {
    for (int c = 0; c <= kCells; ++c) cell_start[c] = 0;
    return PLVS_OK;
  }
  plvs::HostStage& st = plvs::thread_stage();
  const size_t b_xy = sizeof(float) * 2 * (size_t)n, o_start = (b_xy + 15) & ~(size_t)15,
               o_items = o_start + ((sizeof(int32_t) * (kCells + 1) + 15) & ~(size_t)15);
  PLVS_HIP_TRY(st.reserve(o_items + sizeof(int32_t) * (size_t)n + 16));
  float* xy = reinterpret_cast<float*>(st.pinned);
  for (int i = 0; i < n; ++i) {
    xy[2 * i] = kps_un[i].x;
    xy[2 * i + 1] = kps_un[i].y;
  }
  PLVS_HIP_TRY(hipMemcpyAsync(st.dev, st.pinned, b_xy, hipMemcpyHostToDevice, st.stream));
  hipLaunchKernelGGL(assign_grid_kernel, dim3(1), dim3(1024), 0, st.stream, reinterpret_cast<const float2*>(st.dev), n, min_x, min_y,
                     grid_w_inv, grid_h_inv, reinterpret_cast<int32_t*>(st.dev + o_start), reinterpret_cast<int32_t*>(st.dev + o_items));
  PLVS_KERNEL_CHECK();
  PLVS_HIP_TRY(hipMemcpyAsync(st.pinned + o_start, st.dev + o_start, o_items - o_start + sizeof(int32_t) * (size_t)n,
                              hipMemcpyDeviceToHost, st.stream));
  PLVS_HIP_TRY(hipStreamSynchronize(st.stream));
  memcpy(cell_start, st.pinned + o_start, sizeof(int32_t) * (kCells + 1));
  *n_items = cell_start[kCells];
  memcpy(cell_items, st.pinned + o_items, sizeof(int32_t) * (size_t)*n_items);
  return PLVS_OK;
}

}  // extern "C"
