// Depth image -> camera-frame cloud, the step feeding both TSDF back ends.
//
// Replaces PointCloudMapping::GeneratePointCloudInCameraFrameBGRA
// (src/PointCloudMapping.cc:929-1031) for the shipped configuration: PointT =
// pcl::PointSurfelSegment with normals (include/PointDefinitions.h:52-55),
// NeighborhoodT = EigthNeighborhoodIndicesFast (src/PointCloudMapping.cc:100),
// Segmentation.on 0 and filterDepth.on 0 (Examples_old/RGB-D/TUM1.yaml:186,203).
//
// The reference makes three serial passes (push_back valid pixels, then per-point
// neighbour look-ups through idxCloud).  Here the cloud index of a grid pixel is a
// prefix count of valid pixels, so two launches suffice:
//   cloud_count : 256 grid pixels per workgroup -> number of valid depths
//   cloud_emit  : workgroup prefix (<= a few hundred counts, one strided sum), ballot
//                 rank inside the wave, then every valid pixel writes its own record.
// Normals never read the emitted cloud: a neighbour's point is recomputed from its depth
// with the same two float multiplies, so the values are the ones the reference reads back.
// Normal arithmetic is double (Eigen::Vector3d, :1007-1030), no contraction.
#include <vector>

#include "common.hpp"

namespace {

constexpr int kCloudThreads = 256;

struct CloudArgs {
  const float* depth;
  int depth_pitch;  // floats
  const uint8_t* bgr;
  int bgr_pitch;  // bytes
  int width, height, step, gcols, ngrid;
  const float2* grid;
  double min_depth, max_depth;
  uint32_t kfid;
};

__device__ __forceinline__ bool depth_valid(const CloudArgs& a, float d) {
  return ((double)d > a.min_depth) && ((double)d < a.max_depth);  // :967 (double limits)
}

__global__ __launch_bounds__(kCloudThreads) void cloud_count(CloudArgs a, uint32_t* __restrict__ counts) {
  const int ii = blockIdx.x * kCloudThreads + threadIdx.x;
  bool ok = false;
  if (ii < a.ngrid) {
    const int m = (ii / a.gcols) * a.step, n = (ii % a.gcols) * a.step;
    ok = depth_valid(a, a.depth[(size_t)m * a.depth_pitch + n]);
  }
  __shared__ uint32_t wsum[kCloudThreads / 64];
  const unsigned long long b = __ballot(ok);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// Back-projection of grid pixel (gm, gn) if it exists and its depth is valid.
__device__ __forceinline__ bool neighbour_point(const CloudArgs& a, int m, int n, double p[3]) {
  if (m < 0 || m >= a.height || n < 0 || n >= a.width) return false;  // :890
  const float d = a.depth[(size_t)m * a.depth_pitch + n];
  if (!depth_valid(a, d)) return false;  // idxCloud < 0, :1018
  const float2 g = a.grid[(m / a.step) * a.gcols + n / a.step];
  p[0] = (double)(g.x * d);
  p[1] = (double)(g.y * d);
  p[2] = (double)d;
  return true;
}

template <bool kRecords>
__global__ __launch_bounds__(kCloudThreads) void cloud_emit(
    CloudArgs a, const uint32_t* __restrict__ counts, plvs_point_surfel* __restrict__ rec,
    float* __restrict__ xyz, uint8_t* __restrict__ rgb, uint8_t* __restrict__ rgba,
    uint32_t* __restrict__ kfid,
    float* __restrict__ normals, float* __restrict__ depth_out, int32_t* __restrict__ p2p,
    int* __restrict__ total) {
  __shared__ uint32_t red[kCloudThreads / 64];
  __shared__ uint32_t wbase[kCloudThreads / 64 + 1];
  // Workgroup prefix: sum of the counts of the workgroups before this one.
  uint32_t s = 0;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += kCloudThreads) s += counts[b];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;

  const int ii = blockIdx.x * kCloudThreads + threadIdx.x;
  const bool in = ii < a.ngrid;
  int m = 0, n = 0;
  float d = 0.0f;
  bool ok = false;
  if (in) {
    m = (ii / a.gcols) * a.step;
    n = (ii % a.gcols) * a.step;
    d = a.depth[(size_t)m * a.depth_pitch + n];
    ok = depth_valid(a, d);
  }
  const unsigned long long bal = __ballot(ok);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (threadIdx.x == 0) {
    wbase[0] = red[0] + red[1] + red[2] + red[3];
  }
  __syncthreads();
  if (lane == 0) red[wave] = __popcll(bal);
  __syncthreads();
  uint32_t base = wbase[0];
  for (int w = 0; w < wave; ++w) base += red[w];
  if (total != nullptr && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
    *total = (int)(wbase[0] + red[0] + red[1] + red[2] + red[3]);

  const int32_t idx = ok ? (int32_t)(base + __popcll(bal & ((1ull << lane) - 1ull))) : -1;
  if (in && p2p != nullptr) {
    // pixelToPointIndex = -1 everywhere except valid grid pixels (:948, :991): this thread
    // owns the step x step cell of its grid pixel.
    for (int dm = 0; dm < a.step && m + dm < a.height; ++dm)
      for (int dn = 0; dn < a.step && n + dn < a.width; ++dn)
        p2p[(size_t)(m + dm) * a.width + (n + dn)] = (dm == 0 && dn == 0) ? idx : -1;
  }
  if (!ok) return;

  const float2 g = a.grid[ii];
  const float px = g.x * d, py = g.y * d;  // :973-974
  const uint8_t* c = a.bgr + (size_t)m * a.bgr_pitch + 3 * n;
  const uint8_t c0 = c[0], c1 = c[1], c2 = c[2];

  // Area-weighted normal over the up/left/down/right neighbours, pairs (1,3) (3,5) (5,7) (7,1)
  // (Neighborhood.h:57-78, PointCloudMapping.cc:1010-1026).
  const double vc[3] = {(double)px, (double)py, (double)d};
  double q[4][3];
  bool qv[4];
  qv[0] = neighbour_point(a, m - a.step, n, q[0]);  // kk = 1
  qv[1] = neighbour_point(a, m, n - a.step, q[1]);  // kk = 3
  qv[2] = neighbour_point(a, m + a.step, n, q[2]);  // kk = 5
  qv[3] = neighbour_point(a, m, n + a.step, q[3]);  // kk = 7
  double nx = 0.0, ny = 0.0, nz = 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int k2 = (k + 1) & 3;
    if (!qv[k] || !qv[k2]) continue;
    const double ax = q[k][0] - vc[0], ay = q[k][1] - vc[1], az = q[k][2] - vc[2];
    const double bx = q[k2][0] - vc[0], by = q[k2][1] - vc[1], bz = q[k2][2] - vc[2];
    nx += ay * bz - az * by;
    ny += az * bx - ax * bz;
    nz += ax * by - ay * bx;
  }
  const double z2 = nx * nx + (ny * ny + nz * nz);  // Eigen redux of three
  if (z2 > 0.0) {                                    // Eigen 3.3 normalize()
    const double len = sqrt(z2);
    nx /= len;
    ny /= len;
    nz /= len;
  }

  if constexpr (kRecords) {
    plvs_point_surfel p;
    p.x = px;
    p.y = py;
    p.z = d;
    p.kfid = a.kfid;
    p.normal_x = (float)nx;
    p.normal_y = (float)ny;
    p.normal_z = (float)nz;
    p.normal_pad = 0.0f;
    p.b = c2;  // p.b = colour[n3 + 2], :980
    p.g = c1;
    p.r = c0;  // p.r = colour[n3], :978
    p.a = 0;   // rgba = 0 in the constructor and never set
    p.depth = d;
    p.label = 0;
    p.label_confidence = 0;
    rec[idx] = p;
  } else {
    xyz[3 * idx + 0] = px;
    xyz[3 * idx + 1] = py;
    xyz[3 * idx + 2] = d;
    if (rgb != nullptr) {
      rgb[3 * idx + 0] = c0;  // the r, g, b members, in that order
      rgb[3 * idx + 1] = c1;
      rgb[3 * idx + 2] = c2;
    }
    if (rgba != nullptr) reinterpret_cast<uchar4*>(rgba)[idx] = make_uchar4(c0, c1, c2, 0);  // r, g, b, a
    if (kfid != nullptr) kfid[idx] = a.kfid;
    if (normals != nullptr) {
      normals[3 * idx + 0] = (float)nx;
      normals[3 * idx + 1] = (float)ny;
      normals[3 * idx + 2] = (float)nz;
    }
    if (depth_out != nullptr) depth_out[idx] = d;
  }
}

}  // namespace

struct plvs_cloudgen {
  int width = 0, height = 0, step = 0, gcols = 0, grows = 0, ngrid = 0, nblocks = 0;
  float2* d_grid = nullptr;
  uint32_t* d_counts = nullptr;
  int* d_total = nullptr;
  int* h_total = nullptr;  // pinned
  hipStream_t stream = nullptr;
  // staging for the host flavour
  plvs::DevBuf<float> depth;
  plvs::DevBuf<uint8_t> bgr;
  plvs::DevBuf<plvs_point_surfel> rec;
  plvs::DevBuf<int32_t> p2p;
};

extern "C" {

int plvs_hip_cloudgen_grid_points(int width, int height, int step, double fx, double fy, double cx,
                                  double cy, float* grid) {
  PLVS_REQUIRE(width > 0 && height > 0 && step > 0 && grid != nullptr, "cloudgen_grid_points arguments");
  PLVS_REQUIRE(fx != 0.0 && fy != 0.0, "focal lengths must be non-zero");
  int ii = 0;
  for (int m = 0; m < height; m += step)
    for (int n = 0; n < width; n += step, ++ii) {
      // src/PointCloudMapping.cc:877: double arithmetic on the float pixel coordinate,
      // rounded by the Eigen::Vector3f constructor.
      grid[2 * ii + 0] = (float)(((double)(float)n - cx) / fx);
      grid[2 * ii + 1] = (float)(((double)(float)m - cy) / fy);
    }
  return PLVS_OK;
}

int plvs_hip_cloudgen_num_grid_points(int width, int height, int step) {
  if (width <= 0 || height <= 0 || step <= 0) return 0;
  return ((width + step - 1) / step) * ((height + step - 1) / step);
}

int plvs_hip_cloudgen_create(int width, int height, int step, const float* grid_points,
                             plvs_cloudgen** out) {
  PLVS_REQUIRE(out != nullptr, "out is null");
  PLVS_REQUIRE(width > 0 && height > 0 && step > 0, "image size / step");
  PLVS_REQUIRE(grid_points != nullptr, "grid_points is null (see plvs_hip_cloudgen_grid_points)");
  plvs_cloudgen* h = new plvs_cloudgen();
  h->width = width;
  h->height = height;
  h->step = step;
  h->gcols = (width + step - 1) / step;
  h->grows = (height + step - 1) / step;
  h->ngrid = h->gcols * h->grows;
  h->nblocks = (h->ngrid + kCloudThreads - 1) / kCloudThreads;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&h->d_grid), sizeof(float2) * (size_t)h->ngrid);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->d_counts), sizeof(uint32_t) * (size_t)h->nblocks);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->d_total), sizeof(int));
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&h->h_total), sizeof(int));
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e == hipSuccess)
    e = hipMemcpy(h->d_grid, grid_points, sizeof(float2) * (size_t)h->ngrid, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    plvs::set_error("cloudgen_create: %s", hipGetErrorString(e));
    plvs_hip_cloudgen_destroy(h);
    return PLVS_ERR_HIP;
  }
  *out = h;
  return PLVS_OK;
}

int plvs_hip_cloudgen_destroy(plvs_cloudgen* h) {
  if (h == nullptr) return PLVS_OK;
  if (h->d_grid) (void)hipFree(h->d_grid);
  if (h->d_counts) (void)hipFree(h->d_counts);
  if (h->d_total) (void)hipFree(h->d_total);
  if (h->h_total) (void)hipHostFree(h->h_total);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  h->depth.release();
  h->bgr.release();
  h->rec.release();
  h->p2p.release();
  delete h;
  return PLVS_OK;
}

static CloudArgs make_args(const plvs_cloudgen* h, const float* d_depth, int depth_pitch,
                           const uint8_t* d_bgr, int bgr_pitch, double min_depth, double max_depth,
                           uint32_t kfid) {
  CloudArgs a;
  a.depth = d_depth;
  a.depth_pitch = depth_pitch;
  a.bgr = d_bgr;
  a.bgr_pitch = bgr_pitch;
  a.width = h->width;
  a.height = h->height;
  a.step = h->step;
  a.gcols = h->gcols;
  a.ngrid = h->ngrid;
  a.grid = h->d_grid;
  a.min_depth = min_depth;
  a.max_depth = max_depth;
  a.kfid = kfid;
  return a;
}

int plvs_hip_cloudgen_generate_dev(plvs_cloudgen* h, const float* d_depth, int depth_pitch,
                                   const uint8_t* d_bgr, int bgr_pitch, double min_depth,
                                   double max_depth, uint32_t kfid, float* d_xyz, uint8_t* d_rgb,
                                   uint8_t* d_rgba, uint32_t* d_kfid, float* d_normals, float* d_point_depth,
                                   int32_t* d_pixel_to_point, int* d_count, void* stream, int* n) {
  PLVS_REQUIRE(h != nullptr, "handle is null");
  PLVS_REQUIRE(d_depth != nullptr && d_bgr != nullptr, "depth / colour image is null");
  PLVS_REQUIRE(depth_pitch >= h->width && bgr_pitch >= 3 * h->width, "row pitch smaller than a row");
  PLVS_REQUIRE(d_xyz != nullptr, "d_xyz is null");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const CloudArgs a = make_args(h, d_depth, depth_pitch, d_bgr, bgr_pitch, min_depth, max_depth, kfid);
  int* total = d_count != nullptr ? d_count : h->d_total;
  cloud_count<<<h->nblocks, kCloudThreads, 0, st>>>(a, h->d_counts);
  PLVS_KERNEL_CHECK();
  cloud_emit<false><<<h->nblocks, kCloudThreads, 0, st>>>(a, h->d_counts, nullptr, d_xyz, d_rgb, d_rgba, d_kfid,
                                                          d_normals, d_point_depth, d_pixel_to_point,
                                                          total);
  PLVS_KERNEL_CHECK();
  if (n != nullptr) {
    PLVS_HIP_TRY(hipMemcpyAsync(h->h_total, total, sizeof(int), hipMemcpyDeviceToHost, st));
    PLVS_HIP_TRY(hipStreamSynchronize(st));
    *n = *h->h_total;
  }
  return PLVS_OK;
}

int plvs_hip_cloudgen_generate(plvs_cloudgen* h, const float* depth, int depth_pitch,
                               const uint8_t* bgr, int bgr_pitch, double min_depth, double max_depth,
                               uint32_t kfid, plvs_point_surfel* out, int capacity,
                               int32_t* pixel_to_point, int* n) {
  PLVS_REQUIRE(h != nullptr && n != nullptr, "handle / n is null");
  PLVS_REQUIRE(depth != nullptr && bgr != nullptr && out != nullptr, "depth / colour / out is null");
  PLVS_REQUIRE(depth_pitch >= h->width && bgr_pitch >= 3 * h->width, "row pitch smaller than a row");
  PLVS_REQUIRE(capacity >= h->ngrid, "capacity must hold one point per grid pixel");
  const size_t dn = (size_t)depth_pitch * h->height, cn = (size_t)bgr_pitch * h->height;
  PLVS_HIP_TRY(h->depth.reserve(dn));
  PLVS_HIP_TRY(h->bgr.reserve(cn));
  PLVS_HIP_TRY(h->rec.reserve((size_t)h->ngrid));
  if (pixel_to_point != nullptr) PLVS_HIP_TRY(h->p2p.reserve((size_t)h->width * h->height));
  hipStream_t st = h->stream;
  PLVS_HIP_TRY(hipMemcpyAsync(h->depth.p, depth, dn * sizeof(float), hipMemcpyHostToDevice, st));
  PLVS_HIP_TRY(hipMemcpyAsync(h->bgr.p, bgr, cn, hipMemcpyHostToDevice, st));
  const CloudArgs a = make_args(h, h->depth.p, depth_pitch, h->bgr.p, bgr_pitch, min_depth, max_depth, kfid);
  cloud_count<<<h->nblocks, kCloudThreads, 0, st>>>(a, h->d_counts);
  PLVS_KERNEL_CHECK();
  cloud_emit<true><<<h->nblocks, kCloudThreads, 0, st>>>(
      a, h->d_counts, h->rec.p, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
      pixel_to_point != nullptr ? h->p2p.p : nullptr, h->d_total);
  PLVS_KERNEL_CHECK();
  PLVS_HIP_TRY(hipMemcpyAsync(h->h_total, h->d_total, sizeof(int), hipMemcpyDeviceToHost, st));
  PLVS_HIP_TRY(hipStreamSynchronize(st));
  *n = *h->h_total;
  if (*n > 0)
    PLVS_HIP_TRY(hipMemcpyAsync(out, h->rec.p, sizeof(plvs_point_surfel) * (size_t)*n,
                                hipMemcpyDeviceToHost, st));
  if (pixel_to_point != nullptr)
    PLVS_HIP_TRY(hipMemcpyAsync(pixel_to_point, h->p2p.p, sizeof(int32_t) * (size_t)h->width * h->height,
                                hipMemcpyDeviceToHost, st));
  PLVS_HIP_TRY(hipStreamSynchronize(st));
  return PLVS_OK;
}

}  // extern "C"
