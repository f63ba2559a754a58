/*
 * oracle/cv_primitives.hpp — restatements of the OpenCV 4.10 primitives that
 * the reference's front end leans on (OpenCV itself is NOT in the reference
 * tree and not installed here, so these follow the published algorithms of
 * modules/imgproc/src/{resize,smooth.dispatch,fixedpoint.inl,deriv}.cpp,
 * modules/features2d/src/{fast,fast_score}.cpp and
 * modules/core/src/mathfuncs_core.simd.hpp; "OpenCV-parity unverified").
 *
 * TEST INFRASTRUCTURE ONLY — part of the CPU oracle, never linked into the
 * product library.
 */
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace ocv {

struct Image {
  int w = 0, h = 0;
  std::vector<uint8_t> d;
  Image() {}
  Image(int w_, int h_) : w(w_), h(h_), d((size_t)w_ * h_) {}
  uint8_t* row(int y) { return d.data() + (size_t)y * w; }
  const uint8_t* row(int y) const { return d.data() + (size_t)y * w; }
};

/* cvRound / cvFloor / cvCeil (core/fast_math.hpp): round-half-to-even via lrint. */
inline int cv_round(double v) { return (int)lrint(v); }
inline int cv_round(float v) { return (int)lrintf(v); }
inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }
inline short saturate_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }
inline int border_reflect101(int p, int len) {
  /* borderInterpolate(BORDER_REFLECT_101) */
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
  return p;
}

/* cv::resize(..., INTER_LINEAR) for CV_8UC1 (resize.cpp: resizeGeneric_ with
 * HResizeLinear<uchar,int,short,2048> and VResizeLinear<uchar,int,short,
 * FixedPtCast<int,uchar,22>>).  Two call forms:
 *   dsize given  (ORB pyramid): inv_scale = dsize / ssize
 *   fx, fy given (line pyramid): dsize = saturate_cast<int>(ssize * f), inv_scale = f */
inline void resize_linear_u8_impl(const Image& src, Image& dst, double inv_scale_x, double inv_scale_y) {
  const int sw = src.w, sh = src.h, dw = dst.w, dh = dst.h;
  const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  const int ONE = 2048;
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> ialpha(2 * dw), ibeta(2 * dh);
  int xmin = 0, xmax = dw;
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor(fx);
    fx -= sx;
    if (sx < 0) { xmin = dx + 1; fx = 0, sx = 0; }
    if (sx + 1 >= sw) {
      xmax = xmax < dx ? xmax : dx;
      if (sx >= sw - 1) fx = 0, sx = sw - 1;
    }
    xofs[dx] = sx;
    ialpha[2 * dx] = saturate_short(cv_round((1.f - fx) * ONE));
    ialpha[2 * dx + 1] = saturate_short(cv_round(fx * ONE));
  }
  (void)xmin;
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cv_floor(fy);
    fy -= sy;
    yofs[dy] = sy;
    ibeta[2 * dy] = saturate_short(cv_round((1.f - fy) * ONE));
    ibeta[2 * dy + 1] = saturate_short(cv_round(fy * ONE));
  }
  std::vector<int> r0(dw), r1(dw);
  auto hresize = [&](int sy, std::vector<int>& D) {
    sy = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy); /* clip(sy, 0, ssize.height) */
    const uint8_t* S = src.row(sy);
    int dx = 0;
    for (; dx < xmax; dx++) {
      const int sx = xofs[dx];
      D[dx] = S[sx] * ialpha[2 * dx] + S[sx + 1] * ialpha[2 * dx + 1];
    }
    for (; dx < dw; dx++) D[dx] = S[xofs[dx]] * ONE;
  };
  for (int dy = 0; dy < dh; dy++) {
    hresize(yofs[dy], r0);
    hresize(yofs[dy] + 1, r1);
    const int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
    uint8_t* D = dst.row(dy);
    for (int x = 0; x < dw; x++) {
      const int v = (((b0 * (r0[x] >> 4)) >> 16) + ((b1 * (r1[x] >> 4)) >> 16) + 2) >> 2;
      D[x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
  }
}

inline void resize_linear_u8(const Image& src, Image& dst) {
  resize_linear_u8_impl(src, dst, (double)dst.w / src.w, (double)dst.h / src.h);
}
inline void resize_linear_u8_factor(const Image& src, Image& dst, double fx, double fy) {
  dst = Image(cv_round(src.w * fx), cv_round(src.h * fy));
  resize_linear_u8_impl(src, dst, fx, fy);
}

/* cv::resize(src, dst, Size(), fx, fy, INTER_LINEAR_EXACT) for CV_8UC1 (OpenCV 4.x resize.cpp: resize_bitExact<uint8_t,
 * interpolationLinear<uint8_t>>, the "bit-exact" path the LSD detector rescales with, lsd_custom.cpp:493):
 *   dsize = (cvRound(w fx), cvRound(h fy));  per destination index v: f = (1 / inv_scale) (v + 0.5) - 0.5 in IEEE double
 *   (softdouble), i = floor(f); inside the source: offset i, coefficients c1 = cvRound((f - i) 256), c0 = 256 - c1
 *   (ufixedpoint16, 8 fraction bits); before the first / behind the last sample the edge value.  Horizontal pass into
 *   16-bit fixed point (c0 p0 + c1 p1), vertical pass in 32-bit fixed point, rounded once: (v + 2^15) >> 16.
 * Restated from the published source; like the other primitives here it could not be run against an OpenCV. */
inline void resize_linear_exact_u8_factor(const Image& src, Image& dst, double fx, double fy) {
  const int sw = src.w, sh = src.h;
  const int dw = cv_round(sw * fx), dh = cv_round(sh * fy);
  dst = Image(dw, dh);
  struct Axis { std::vector<int> ofs; std::vector<unsigned> c0, c1; int lo, hi; };
  auto coeffs = [](double inv_scale, int srcsize, int dstsize) {
    Axis a;
    a.ofs.assign(dstsize, 0); a.c0.assign(dstsize, 256u); a.c1.assign(dstsize, 0u);
    a.lo = 0; a.hi = dstsize;
    const double scale = 1.0 / inv_scale;
    for (int v = 0; v < dstsize; ++v) {
      const double f = scale * ((double)v + 0.5) - 0.5;
      const int i = cv_floor(f);
      if (i >= 0 && srcsize > 1) {
        if (i < srcsize - 1) {
          a.ofs[v] = i;
          a.c1[v] = (unsigned)cv_round((f - (double)i) * 256.0);
          a.c0[v] = 256u > a.c1[v] ? 256u - a.c1[v] : 0u;
        } else {
          a.ofs[v] = srcsize - 1;
          a.hi = a.hi < v ? a.hi : v;
        }
      } else {
        a.lo = a.lo > v + 1 ? a.lo : v + 1;
      }
    }
    return a;
  };
  const Axis X = coeffs(fx, sw, dw), Y = coeffs(fy, sh, dh);
  auto hline = [&](int sy, std::vector<unsigned>& D) {   // ufixedpoint16 values
    const uint8_t* S = src.row(sy);
    int i = 0;
    for (; i < X.lo && i < dw; ++i) D[i] = (unsigned)S[0] << 8;
    for (; i < X.hi; ++i) D[i] = X.c0[i] * S[X.ofs[i]] + X.c1[i] * S[X.ofs[i] + 1];
    const unsigned last = (unsigned)S[X.ofs[dw - 1]] << 8;
    for (; i < dw; ++i) D[i] = last;
  };
  std::vector<unsigned> r0(dw), r1(dw);
  for (int dy = 0; dy < dh; ++dy) {
    uint8_t* D = dst.row(dy);
    if (dy < Y.lo || dy >= Y.hi) {   // the first / last source row alone
      hline(dy < Y.lo ? 0 : Y.ofs[dh - 1], r0);
      for (int x = 0; x < dw; ++x) D[x] = (uint8_t)((r0[x] + 128u) >> 8);
      continue;
    }
    hline(Y.ofs[dy], r0);
    hline(Y.ofs[dy] + 1, r1);
    for (int x = 0; x < dw; ++x) {
      const unsigned long long v = (unsigned long long)Y.c0[dy] * r0[x] + (unsigned long long)Y.c1[dy] * r1[x];
      const unsigned long long q = (v + 32768ull) >> 16;
      D[x] = (uint8_t)(q > 255ull ? 255ull : q);
    }
  }
}

/* cv::Sobel(src, dst, CV_16S, dx, dy, 3) with the default BORDER_REFLECT_101: exact integers. */
inline void sobel3_s16(const Image& src, std::vector<short>& dxo, std::vector<short>& dyo) {
  const int w = src.w, h = src.h;
  dxo.assign((size_t)w * h, 0);
  dyo.assign((size_t)w * h, 0);
  for (int y = 0; y < h; y++) {
    const uint8_t* r0 = src.row(border_reflect101(y - 1, h));
    const uint8_t* r1 = src.row(y);
    const uint8_t* r2 = src.row(border_reflect101(y + 1, h));
    for (int x = 0; x < w; x++) {
      const int xm = border_reflect101(x - 1, w), xp = border_reflect101(x + 1, w);
      dxo[(size_t)y * w + x] = (short)((r0[xp] - r0[xm]) + 2 * (r1[xp] - r1[xm]) + (r2[xp] - r2[xm]));
      dyo[(size_t)y * w + x] = (short)((r2[xm] + 2 * r2[x] + r2[xp]) - (r0[xm] + 2 * r0[x] + r0[xp]));
    }
  }
}

/* Bit-exact fixed-point Gaussian kernel, 8 fractional bits, weights sum to 256
 * (smooth.dispatch.cpp: getGaussianKernelBitExact + getGaussianKernelFixedPoint_ED,
 * rounding error diffused from the edge towards the centre tap). */
inline std::vector<int> gaussian_kernel_q8(int n, double sigma) {
  if (sigma <= 0) sigma = ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
  std::vector<double> k(n);
  const double scale2x = -0.5 / (sigma * sigma);
  double sum = 0;
  for (int i = 0; i < n; i++) {
    const double x = i - (n - 1) * 0.5;
    k[i] = std::exp(scale2x * x * x);
    sum += k[i];
  }
  for (int i = 0; i < n; i++) k[i] /= sum;
  std::vector<int> q(n);
  const int n2 = n / 2;
  double err = 0;
  long long s = 0;
  for (int i = 0; i < n2; i++) {
    const double adj = k[i] * 256.0 + err;
    const long long v0 = (long long)lrint(adj);
    err = adj - (double)v0;
    q[i] = q[n - 1 - i] = (int)v0;
    s += v0;
  }
  q[n2] = (int)(256 - 2 * s);
  return q;
}

/* cv::GaussianBlur(src, dst, Size(k,k), sigma, sigma, BORDER_REFLECT_101) for a
 * continuous CV_8UC1 image: the ufixedpoint16 / ufixedpoint32 separable path.
 * The row pass keeps all 8 fractional bits, so the result is the exact 2-D
 * integer convolution rounded once: (sum + 2^15) >> 16. */
inline void gaussian_blur_u8(const Image& src, Image& dst, int ksize, double sigma) {
  const std::vector<int> kq = gaussian_kernel_q8(ksize, sigma);
  const int r = ksize / 2, w = src.w, h = src.h;
  std::vector<uint32_t> tmp((size_t)w * h);
  for (int y = 0; y < h; y++) {
    const uint8_t* S = src.row(y);
    for (int x = 0; x < w; x++) {
      uint32_t acc = 0;
      for (int t = -r; t <= r; t++) acc += (uint32_t)kq[t + r] * S[border_reflect101(x + t, w)];
      tmp[(size_t)y * w + x] = acc; /* 8.8 fixed point, <= 255*256 */
    }
  }
  dst = Image(w, h);
  for (int y = 0; y < h; y++) {
    uint8_t* D = dst.row(y);
    for (int x = 0; x < w; x++) {
      uint32_t acc = 0;
      for (int t = -r; t <= r; t++)
        acc += (uint32_t)kq[t + r] * tmp[(size_t)border_reflect101(y + t, h) * w + x];
      D[x] = (uint8_t)((acc + (1u << 15)) >> 16);
    }
  }
}

/* cv::fastAtan2 (mathfuncs_core.simd.hpp atan_f32): degrees in [0, 360). */
inline float fast_atan2(float y, float x) {
  static const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
  static const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
  static const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
  static const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
  const float ax = std::fabs(x), ay = std::fabs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

struct FastKp {
  float x, y, response;
};

/* cornerScore<16> (fast_score.cpp). `pixel` holds the 25 circle offsets. */
inline int fast_corner_score16(const uint8_t* ptr, const int pixel[25], int threshold) {
  const int N = 25;
  int k, v = ptr[0];
  short d[N];
  for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
  int a0 = threshold;
  for (k = 0; k < 16; k += 2) {
    int a = std::min((int)d[k + 1], (int)d[k + 2]);
    a = std::min(a, (int)d[k + 3]);
    if (a <= a0) continue;
    a = std::min(a, (int)d[k + 4]);
    a = std::min(a, (int)d[k + 5]);
    a = std::min(a, (int)d[k + 6]);
    a = std::min(a, (int)d[k + 7]);
    a = std::min(a, (int)d[k + 8]);
    a0 = std::max(a0, std::min(a, (int)d[k]));
    a0 = std::max(a0, std::min(a, (int)d[k + 9]));
  }
  int b0 = -a0;
  for (k = 0; k < 16; k += 2) {
    int b = std::max((int)d[k + 1], (int)d[k + 2]);
    b = std::max(b, (int)d[k + 3]);
    b = std::max(b, (int)d[k + 4]);
    b = std::max(b, (int)d[k + 5]);
    if (b >= b0) continue;
    b = std::max(b, (int)d[k + 6]);
    b = std::max(b, (int)d[k + 7]);
    b = std::max(b, (int)d[k + 8]);
    b0 = std::min(b0, std::max(b, (int)d[k]));
    b0 = std::min(b0, std::max(b, (int)d[k + 9]));
  }
  return -b0 - 1;
}

/* cv::FAST(img, kps, threshold, nonmaxSuppression, TYPE_9_16) (fast.cpp FAST_t<16>)
 * on the cols x rows sub-image starting at `img` with row stride `stride`. */
inline void fast_9_16(const uint8_t* img, int stride, int cols, int rows, int threshold, bool nonmax,
                      std::vector<FastKp>& out) {
  static const int offsets16[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},  {3, 0},  {3, -1},
                                       {2, -2}, {1, -3},  {0, -3},  {-1, -3}, {-2, -2}, {-3, -1},
                                       {-3, 0}, {-3, 1},  {-2, 2},  {-1, 3}};
  const int K = 8, N = 25;
  int pixel[25];
  for (int k = 0; k < 16; k++) pixel[k] = offsets16[k][0] + offsets16[k][1] * stride;
  for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
  threshold = std::min(std::max(threshold, 0), 255);
  uint8_t tab[512];
  for (int i = -255; i <= 255; i++) tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
  std::vector<uint8_t> bufv((size_t)cols * 3, 0);
  std::vector<int> cpv((size_t)(cols + 1) * 3, 0);
  uint8_t* buf[3] = {bufv.data(), bufv.data() + cols, bufv.data() + 2 * cols};
  int* cpbuf[3] = {cpv.data() + 1, cpv.data() + (cols + 1) + 1, cpv.data() + 2 * (cols + 1) + 1};
  for (int i = 3; i < rows - 2; i++) {
    const uint8_t* ptr = img + (size_t)i * stride + 3;
    uint8_t* curr = buf[(i - 3) % 3];
    int* cornerpos = cpbuf[(i - 3) % 3];
    memset(curr, 0, cols);
    int ncorners = 0;
    if (i < rows - 3) {
      for (int j = 3; j < cols - 3; j++, ptr++) {
        const int v = ptr[0];
        const uint8_t* t = &tab[0] - v + 255;
        int d = t[ptr[pixel[0]]] | t[ptr[pixel[8]]];
        if (d == 0) continue;
        d &= t[ptr[pixel[2]]] | t[ptr[pixel[10]]];
        d &= t[ptr[pixel[4]]] | t[ptr[pixel[12]]];
        d &= t[ptr[pixel[6]]] | t[ptr[pixel[14]]];
        if (d == 0) continue;
        d &= t[ptr[pixel[1]]] | t[ptr[pixel[9]]];
        d &= t[ptr[pixel[3]]] | t[ptr[pixel[11]]];
        d &= t[ptr[pixel[5]]] | t[ptr[pixel[13]]];
        d &= t[ptr[pixel[7]]] | t[ptr[pixel[15]]];
        if (d & 1) {
          const int vt = v - threshold;
          int count = 0;
          for (int k = 0; k < N; k++) {
            const int x = ptr[pixel[k]];
            if (x < vt) {
              if (++count > K) {
                cornerpos[ncorners++] = j;
                if (nonmax) curr[j] = (uint8_t)fast_corner_score16(ptr, pixel, threshold);
                break;
              }
            } else
              count = 0;
          }
        }
        if (d & 2) {
          const int vt = v + threshold;
          int count = 0;
          for (int k = 0; k < N; k++) {
            const int x = ptr[pixel[k]];
            if (x > vt) {
              if (++count > K) {
                cornerpos[ncorners++] = j;
                if (nonmax) curr[j] = (uint8_t)fast_corner_score16(ptr, pixel, threshold);
                break;
              }
            } else
              count = 0;
          }
        }
      }
    }
    cornerpos[-1] = ncorners;
    if (i == 3) continue;
    const uint8_t* prev = buf[(i - 4 + 3) % 3];
    const uint8_t* pprev = buf[(i - 5 + 3) % 3];
    cornerpos = cpbuf[(i - 4 + 3) % 3];
    ncorners = cornerpos[-1];
    for (int k = 0; k < ncorners; k++) {
      const int j = cornerpos[k];
      const int score = prev[j];
      if (!nonmax || (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] &&
                      score > pprev[j] && score > pprev[j + 1] && score > curr[j - 1] &&
                      score > curr[j] && score > curr[j + 1]))
        out.push_back(FastKp{(float)j, (float)(i - 1), (float)score});
    }
  }
}

}  // namespace ocv
