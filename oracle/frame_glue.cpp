// TEST INFRASTRUCTURE ONLY — CPU restatement of the per-frame glue between extraction and the searches, the oracle of
// plvs_amd/csrc/frame_glue.hip:
//   Frame::UndistortKeyPoints      src/Frame.cc:1507-1552
//   Frame::UndistortKeyLines       src/Frame.cc:1555-1700 (pinhole branch, single camera)
//   Frame::ComputeImageBounds      src/Frame.cc:1749-1778
//   Frame::AssignFeaturesToGrid    src/Frame.cc:716-746 (the key-point grid), PosInGrid :1305-1316
// cv::undistortPoints is restated from OpenCV 4.10's published algorithm (calib3d/undistort.dispatch.cpp,
// cvUndistortPointsInternal, default criteria = 5 iterations, all in double) like the other OpenCV primitives here
// (cv_primitives.hpp) — unverified against a real OpenCV; everything else is pinned by the reference's own Frame.cc
// compiled into oracle/_ref/libmatchers_ref.so (tests/test_oracle_pinned_matchers.py).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "cv_primitives.hpp"

namespace {
struct KeyPoint { float x, y, size, angle, response; int32_t octave, class_id; };
struct KeyLine {
  float angle; int32_t class_id, octave; float pt_x, pt_y, response, size, startPointX, startPointY, endPointX, endPointY,
      sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY, lineLength; int32_t numOfPixels;
};
static_assert(sizeof(KeyPoint) == 28 && sizeof(KeyLine) == 68, "record layouts");

void undistort(const float* K4, const float* dist, int ndist, const float* xy, int n, float* out) {
  double k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < ndist && i < 8; ++i) k[i] = (double)dist[i];
  const double fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3], ifx = 1. / fx, ify = 1. / fy;
  for (int i = 0; i < n; ++i) {
    double x = xy[2 * i], y = xy[2 * i + 1];
    x = (x - cx) * ifx;
    y = (y - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; ++j) {
      const double r2 = x * x + y * y;
      const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
      if (icdist < 0) {
        x = ((double)xy[2 * i] - cx) * ifx;
        y = ((double)xy[2 * i + 1] - cy) * ify;
        break;
      }
      const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
      const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
      x = (x0 - deltaX) * icdist;
      y = (y0 - deltaY) * icdist;
    }
    const double xx = fx * x + 0.0 * y + cx, yy = 0.0 * x + fy * y + cy, ww = 1. / (0.0 * x + 0.0 * y + 1.0);
    out[2 * i] = (float)(xx * ww);
    out[2 * i + 1] = (float)(yy * ww);
  }
}
}  // namespace

extern "C" {

void oracle_frame_undistort_keypoints(const KeyPoint* kps, int n, const float* K4, const float* dist, int ndist, KeyPoint* un) {
  if (n > 0 && un != kps) std::memcpy(un, kps, sizeof(KeyPoint) * (size_t)n);
  if (n == 0 || ndist == 0 || dist[0] == 0.0f) return;
  std::vector<float> xy(2 * (size_t)n), out(2 * (size_t)n);
  for (int i = 0; i < n; ++i) { xy[2 * i] = kps[i].x; xy[2 * i + 1] = kps[i].y; }
  undistort(K4, dist, ndist, xy.data(), n, out.data());
  for (int i = 0; i < n; ++i) { un[i].x = out[2 * i]; un[i].y = out[2 * i + 1]; }
}

void oracle_frame_compute_image_bounds(int width, int height, const float* K4, const float* dist, int ndist, float* bounds5) {
  float mnx = 0.0f, mxx = (float)width, mny = 0.0f, mxy = (float)height;
  if (ndist > 0 && dist[0] != 0.0f) {
    const float c[8] = {0.0f, 0.0f, (float)width, 0.0f, 0.0f, (float)height, (float)width, (float)height};
    float m[8];
    undistort(K4, dist, ndist, c, 4, m);
    mnx = std::min(m[0], m[4]); mxx = std::max(m[2], m[6]); mny = std::min(m[1], m[3]); mxy = std::max(m[5], m[7]);
  }
  bounds5[0] = mnx; bounds5[1] = mxx; bounds5[2] = mny; bounds5[3] = mxy;
  bounds5[4] = (float)std::sqrt(std::pow(mxx - mnx, 2) + std::pow(mxy - mny, 2));
}

int oracle_frame_undistort_keylines(const KeyLine* kl, int n, const float* K4, const float* dist, int ndist, const float* bounds4,
                                    KeyLine* un, int32_t* kept_index) {
  if (n == 0) return 0;
  if (ndist == 0 || dist[0] == 0.0f) {
    std::memcpy(un, kl, sizeof(KeyLine) * (size_t)n);
    for (int i = 0; i < n; ++i) kept_index[i] = i;
    return n;
  }
  std::vector<float> xy(4 * (size_t)n), out(4 * (size_t)n);
  for (int i = 0; i < n; ++i) {
    xy[4 * i] = kl[i].startPointX; xy[4 * i + 1] = kl[i].startPointY; xy[4 * i + 2] = kl[i].endPointX; xy[4 * i + 3] = kl[i].endPointY;
  }
  undistort(K4, dist, ndist, xy.data(), 2 * n, out.data());
  constexpr float DEG2RAD = M_PI / 180.0f;
  int m = 0;
  for (int i = 0; i < n; ++i) {
    KeyLine k = kl[i];
    k.startPointX = out[4 * i]; k.startPointY = out[4 * i + 1]; k.endPointX = out[4 * i + 2]; k.endPointY = out[4 * i + 3];
    k.angle = ocv::fast_atan2(k.endPointY - k.startPointY, k.endPointX - k.startPointX) * DEG2RAD;
    if (k.startPointX < bounds4[0] || k.startPointX >= bounds4[1] || k.startPointY < bounds4[2] || k.startPointY >= bounds4[3] ||
        k.endPointX < bounds4[0] || k.endPointX >= bounds4[1] || k.endPointY < bounds4[2] || k.endPointY >= bounds4[3])
      continue;
    un[m] = k;
    kept_index[m] = i;
    ++m;
  }
  return m;
}

// -> items written; cell_start[64 * 48 + 1], cell = column * 48 + row, members in key-point order
int oracle_frame_assign_features_to_grid(const KeyPoint* un, int n, float min_x, float min_y, float inv_w, float inv_h,
                                         int32_t* cell_start, int32_t* cell_items) {
  const int cols = 64, rows = 48;
  std::vector<std::vector<int32_t>> grid((size_t)cols * rows);
  for (int i = 0; i < n; ++i) {
    const int px = (int)std::round((un[i].x - min_x) * inv_w), py = (int)std::round((un[i].y - min_y) * inv_h);
    if (px < 0 || px >= cols || py < 0 || py >= rows) continue;
    grid[(size_t)px * rows + py].push_back(i);
  }
  int at = 0;
  for (int c = 0; c < cols * rows; ++c) {
    cell_start[c] = at;
    for (int32_t i : grid[c]) cell_items[at++] = i;
  }
  cell_start[cols * rows] = at;
  return at;
}

}  // extern "C"
