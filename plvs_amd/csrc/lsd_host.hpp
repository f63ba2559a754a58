// The LSD line detector (Line.LSD.on: 1) — host side.
//
// Replaces, for LineExtractor with skUseLsdExtractor (src/LineExtractor.cc:203-206, 275-279):
//   cv::lsd::LineSegmentDetectorImpl::detect      Thirdparty/line_descriptor/src/lsd_custom.cpp:433-1081
//   LSDDetectorC::detect / detectImpl              Thirdparty/line_descriptor/src/LSDDetector_custom.cpp:50-298
//
// Split of the work (lines.hip holds the kernels and the C ABI):
//   device  lsd_blur          the Gaussian in front of the rescaling (sigma = sigma_scale, or sigma_scale / scale below 1),
//                             exact 8.8 fixed point like the other blurs of the front end
//           lsd_resize_exact  cv::resize(..., INTER_LINEAR_EXACT) by `scale` (PLVS passes its PYRAMID scale here,
//                             src/LineExtractor.cc:205: every level is first magnified by 1.2)
//           lsd_ll_angle      the level-line field (lsd_custom.cpp:549-598): 2 x 2 gradient, its norm in double, the angle by
//                             cv::fastAtan2, NOTDEF below the threshold and on the last row / column; the maximum norm
//           lines_blur5 / lines_resize / lines_sobel_grad / lbd_kernel   the detector's pyramid and the descriptors (as EDLines')
//   host    the pseudo-ordering — libstdc++'s std::sort over every pixel's bin, UNSTABLE: the seed order, and with it the
//           result, is that implementation's introsort on that input, so the only way to reproduce it is to run it —, the
//           region growing over the shared `used` map, rectangle fit, density refinement, NFA improvement: one sequential
//           loop over the seeds, as in the reference (this file).
// Every arithmetic step keeps the reference's operand types and order (float / double, which libm function): the parity
// tests compare segments, KeyLines and descriptors bit for bit with the reference's own sources compiled here
// (tests/test_lsd.py) and with digests they made.
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <ctime>
#include <mutex>
#include <thread>
#include <vector>

namespace plvs {
namespace lsd {

struct Options {        // LSDDetectorC::LSDOptions (descriptor_custom.hpp:928-957) as far as the detector reads them
  int refine = 2;       // cv::LSD_REFINE_ADV
  double scale = 0.8, sigma_scale = 0.6, quant = 2.0, ang_th = 22.5, log_eps = 0.0, density_th = 0.7;
  int n_bins = 1024;
};

constexpr double kPi = 3.1415926535897932384626433832795;   // CV_PI
constexpr double kNotDef = -1024.0;                          // NOTDEF
constexpr double kDegToRad = kPi / 180;

// cv::fastAtan2 (degrees in [0, 360)), mathfuncs_core.simd.hpp atan_f32: plain IEEE f32 operations (host and device)
#if defined(__HIPCC__)
#define PLVS_LSD_HD __host__ __device__ inline
#else
#define PLVS_LSD_HD inline
#endif
PLVS_LSD_HD float fast_atan2_deg(float y, float x) {
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

// ---------------------------------------------------------------- image-primitive tables (host side of the kernels)
struct Q8Kernel {       // cv::getGaussianKernel in 8.8 fixed point (the bit-exact CV_8U path), up to 31 taps
  int n = 0;
  int w[31] = {0};
};
inline Q8Kernel gaussian_q8(int n, double sigma) {
  Q8Kernel q;
  q.n = n;
  if (sigma <= 0) sigma = ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
  double k[31], sum = 0;
  const double scale2x = -0.5 / (sigma * sigma);
  for (int i = 0; i < n; ++i) {
    const double x = i - (n - 1) * 0.5;
    k[i] = std::exp(scale2x * x * x);
    sum += k[i];
  }
  for (int i = 0; i < n; ++i) k[i] /= sum;
  const int n2 = n / 2;
  double err = 0;
  long long s = 0;
  for (int i = 0; i < n2; ++i) {   // the rounding error of a tap is carried to the next; the centre takes the rest
    const double adj = k[i] * 256.0 + err;
    const long long v0 = (long long)lrint(adj);
    err = adj - (double)v0;
    q.w[i] = q.w[n - 1 - i] = (int)v0;
    s += v0;
  }
  q.w[n2] = (int)(256 - 2 * s);
  return q;
}

// One axis of cv::resize(..., INTER_LINEAR_EXACT) for CV_8U (resize.cpp: interpolationLinear<uint8_t>): per destination
// index the source offset and the two 8.8 coefficients; [lo, hi) = the indices that interpolate, the others take the edge.
struct ExactAxis {
  std::vector<int> ofs;
  std::vector<uint16_t> c0, c1;
  int lo = 0, hi = 0;
};
inline ExactAxis exact_resize_axis(double inv_scale, int srcsize, int dstsize) {
  ExactAxis a;
  a.ofs.assign((size_t)dstsize, 0);
  a.c0.assign((size_t)dstsize, 256);
  a.c1.assign((size_t)dstsize, 0);
  a.lo = 0;
  a.hi = dstsize;
  const double scale = 1.0 / inv_scale;
  for (int v = 0; v < dstsize; ++v) {
    const double f = scale * ((double)v + 0.5) - 0.5;
    int i = (int)f;
    i -= (i > f);   // cvFloor
    if (i >= 0 && srcsize > 1) {
      if (i < srcsize - 1) {
        a.ofs[(size_t)v] = i;
        const int c1 = (int)lrint((f - (double)i) * 256.0);
        a.c1[(size_t)v] = (uint16_t)c1;
        a.c0[(size_t)v] = (uint16_t)(256 > c1 ? 256 - c1 : 0);
      } else {
        a.ofs[(size_t)v] = srcsize - 1;
        a.hi = std::min(a.hi, v);
      }
    } else {
      a.lo = std::max(a.lo, v + 1);
    }
  }
  return a;
}

// ---------------------------------------------------------------- the detector on one level-line field
struct Segment4 { float x1, y1, x2, y2; };

// std::sort's result — libstdc++'s introsort — with the halves of its partitions on several threads.  The reference's
// ordering is an UNSTABLE sort whose comparator sees part of the element only: the order of equal keys is whatever this
// implementation of std::sort does, and the only way to have it is to run its steps.  They are run here through libstdc++'s
// own building blocks: __introsort_loop is, per range, { partition around the median of three; recurse into the right part;
// continue with the left }, the recursion touches nothing outside its range, so the right part can go to another thread —
// same comparisons, same moves, same array afterwards; __final_insertion_sort then runs over the whole array as std::sort does.
// (tests/test_lsd.py compares with the reference's own std::sort call, compiled by g++ from its sources.)
#if defined(__GLIBCXX__)
template <typename It, typename Comp>
void introsort_loop_mt(It first, It last, long depth, Comp comp, int level, std::vector<std::thread>& pool, std::mutex& mu) {
  while (last - first > int(std::_S_threshold)) {
    if (depth == 0) {
      std::__partial_sort(first, last, last, comp);
      return;
    }
    --depth;
    It cut = std::__unguarded_partition_pivot(first, last, comp);
    bool handed_over = false;
    if (level < 3 && last - cut > 8192) {
      ++level;
      std::lock_guard<std::mutex> g(mu);
      try {
        pool.emplace_back([=, &pool, &mu] { introsort_loop_mt(cut, last, depth, comp, level, pool, mu); });
        handed_over = true;
      } catch (...) {   // (no thread to be had: the right part is sorted here, as std::sort would)
      }
    }
    if (!handed_over) std::__introsort_loop(cut, last, depth, comp);
    last = cut;
  }
}
template <typename It, typename Cmp>
void sort_as_std(It first, It last, Cmp cmp, bool threads) {
  if (!threads || last - first < 65536) {
    std::sort(first, last, cmp);
    return;
  }
  auto comp = __gnu_cxx::__ops::__iter_comp_iter(cmp);
  std::vector<std::thread> pool;
  std::mutex mu;
  pool.reserve(16);   // (at most 7 are made; no reallocation while a worker holds a reference)
  introsort_loop_mt(first, last, std::__lg(last - first) * 2, comp, 0, pool, mu);
  for (size_t i = 0;; ++i) {
    std::thread t;
    {
      std::lock_guard<std::mutex> g(mu);
      if (i >= pool.size()) break;
      t = std::move(pool[i]);
    }
    t.join();
  }
  std::__final_insertion_sort(first, last, comp);
}
#else
template <typename It, typename Cmp>
void sort_as_std(It first, It last, Cmp cmp, bool) { std::sort(first, last, cmp); }
#endif


class Level {
 public:
  // angles / modgrad: w x h doubles as lsd_ll_angle leaves them (angles = kNotDef where undefined; the last row and column
  // of modgrad are never read); max_grad: the largest norm above the threshold, or -1.
  void detect(const double* angles, const double* modgrad, int w, int h, double max_grad, const Options& o,
              std::vector<Segment4>& out) {
    ang_ = angles;
    mod_ = modgrad;
    w_ = w;
    h_ = h;
    out.clear();
    // ---- pseudo-ordering (lsd_custom.cpp:600-611): every pixel of the field but the last row / column, by bin, descending.
    // The reference sorts {x, y, bin} records with std::sort and a comparator that reads the bin only: UNSTABLE, the order of
    // equal bins is whatever libstdc++'s introsort does on this input.  Which elements it compares and moves depends on the
    // comparisons' outcomes alone, not on what else an element carries — so the same call on 4-byte keys (bin above the pixel
    // index) yields the same permutation with a third of the memory traffic.
    // Beside it, on helper threads: cosf / sinf of every defined pixel's angle, which region growing adds up point by point
    // (the same libm calls on the same arguments, made ahead of the loop instead of inside it).
    const double t_a = now_ms_();
    const size_t npx = (size_t)w * h;
    cs_.resize(2 * npx);
    const int helpers = std::thread::hardware_concurrency() >= 4 ? 2 : 0;
    std::thread th[2];
    auto fill_cs = [this, angles, w](int y0, int y1) {
      for (size_t at = (size_t)y0 * w, end = (size_t)y1 * w; at < end; ++at)
        if (angles[at] != kNotDef) {
          cs_[2 * at] = cosf(float(angles[at]));
          cs_[2 * at + 1] = sinf(float(angles[at]));
        }
    };
    bool inline_part[2] = {false, false};
    for (int i = 0; i < helpers; ++i) {
      try {
        th[i] = std::thread(fill_cs, (h * i) / helpers, (h * (i + 1)) / helpers);
      } catch (...) {   // (no thread to be had: this part after the ordering, on this one)
        inline_part[i] = true;
      }
    }
    const double bin_coef = (max_grad > 0) ? double(o.n_bins - 1) / max_grad : 0;
    const bool narrow = npx <= (size_t(1) << 22) && o.n_bins <= 1024;
    const char* force = getenv("PLVS_LSD_SORT_THREADS");   // 0 / 1: never / always (tests); default: where there are cores to spare
    const bool mt = force ? force[0] == '1' : std::thread::hardware_concurrency() >= 16;
    if (narrow) order_seeds<uint32_t, 22>(modgrad, w, h, bin_coef, keys32_, mt);
    else order_seeds<uint64_t, 32>(modgrad, w, h, bin_coef, keys64_, mt);
    const size_t nseeds = narrow ? keys32_.size() : keys64_.size();
    if (helpers == 0) fill_cs(0, h);
    for (int i = 0; i < helpers; ++i) {
      if (inline_part[i]) fill_cs((h * i) / helpers, (h * (i + 1)) / helpers);
      else th[i].join();
    }
    ms_order = now_ms_() - t_a;

    const double prec = kPi * o.ang_th / 180;
    const double p = o.ang_th / 180;
    log_nt_ = 5 * (std::log10(double(w)) + std::log10(double(h))) / 2 + std::log10(11.0);
    const size_t min_reg_size = size_t(-log_nt_ / std::log10(p));
    used_.assign((size_t)w * h, 0);
    std::vector<Pt> reg;
    for (size_t i = 0; i < nseeds; ++i) {
      const size_t at = narrow ? size_t(keys32_[i] & ((1u << 22) - 1)) : size_t(keys64_[i] & 0xffffffffull);
      if (used_[at] != 0 || angles[at] == kNotDef) continue;
      double reg_angle;
      grow(int(at % (size_t)w), int(at / (size_t)w), reg, reg_angle, prec);
      if (reg.size() < min_reg_size) continue;
      Rect rec;
      to_rect(reg, reg_angle, prec, p, rec);
      double log_nfa = -1;
      if (o.refine > 0) {
        if (!refine(reg, reg_angle, prec, p, rec, o.density_th)) continue;
        if (o.refine >= 2) {
          log_nfa = improve(rec, o.log_eps);
          if (log_nfa <= o.log_eps) continue;
        }
      }
      (void)log_nfa;
      rec.x1 += 0.5; rec.y1 += 0.5;
      rec.x2 += 0.5; rec.y2 += 0.5;
      if (o.scale != 1) {
        rec.x1 /= o.scale; rec.y1 /= o.scale;
        rec.x2 /= o.scale; rec.y2 /= o.scale;
        rec.width /= o.scale;
      }
      out.push_back(Segment4{float(rec.x1), float(rec.y1), float(rec.x2), float(rec.y2)});
    }
    ms_regions = now_ms_() - t_a - ms_order;
  }
  double ms_order = 0, ms_regions = 0;   // of the last detect: the ordering; the seed loop

 private:
  // normPoint {p, norm} as a key: the bin above the pixel index y w + x
  template <typename K, int kShift>
  static void order_seeds(const double* modgrad, int w, int h, double bin_coef, std::vector<K>& keys, bool threads) {
    keys.clear();
    keys.reserve((size_t)(w - 1) * (size_t)(h - 1));
    for (int y = 0; y < h - 1; ++y) {
      const double* row = modgrad + (size_t)y * w;
      for (int x = 0; x < w - 1; ++x) keys.push_back((K(int(row[x] * bin_coef)) << kShift) | K((size_t)y * w + x));
    }
    sort_as_std(keys.begin(), keys.end(), [](const K& a, const K& b) { return (a >> kShift) > (b >> kShift); }, threads);
  }
  struct Pt { int x, y; double angle, modgrad; };   // RegionPoint (its `used` pointer is the index y w + x)
  struct Rect {
    double x1, y1, x2, y2, width, x, y, theta, dx, dy, prec, p;
  };
  const double* ang_ = nullptr;
  const double* mod_ = nullptr;
  int w_ = 0, h_ = 0;
  double log_nt_ = 0;
  std::vector<uint32_t> keys32_;
  std::vector<uint64_t> keys64_;
  std::vector<float> cs_;          // cosf, sinf of the angle of every defined pixel
  std::vector<uint8_t> used_;

  static double now_ms_() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
  }
  static double dist2(double x1, double y1, double x2, double y2) { return (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1); }
  static double dist(double x1, double y1, double x2, double y2) { return std::sqrt(dist2(x1, y1, x2, y2)); }
  static double diff_signed(double a, double b) {
    double d = a - b;
    while (d <= -kPi) d += (2 * kPi);
    while (d > kPi) d -= (2 * kPi);
    return d;
  }
  static double diff_abs(double a, double b) { return std::fabs(diff_signed(a, b)); }
  static bool rel_equal(double a, double b) {
    if (a == b) return true;
    const double abs_diff = std::fabs(a - b), aa = std::fabs(a), bb = std::fabs(b);
    double abs_max = (aa > bb) ? aa : bb;
    if (abs_max < DBL_MIN) abs_max = DBL_MIN;
    return (abs_diff / abs_max) <= (100.0 * DBL_EPSILON);
  }

  bool aligned(int x, int y, double theta, double prec) const {   // isAligned :1121-1137
    if (x < 0 || y < 0 || x >= w_ || y >= h_) return false;
    const double a = ang_[(size_t)y * w_ + x];
    if (a == kNotDef) return false;
    double n_theta = theta - a;
    if (n_theta < 0) n_theta = -n_theta;
    if (n_theta > (3 * kPi) / 2) {
      n_theta -= (2 * kPi);
      if (n_theta < 0) n_theta = -n_theta;
    }
    return n_theta <= prec;
  }

  // region_grow :613-668.  The running direction is summed in float; each new point contributes the FLOAT cosine / sine of
  // its angle rounded to float (`cos(float(angle))`: with the reference's includes the unqualified name resolves to the
  // float overload, i.e. libm's cosf / sinf — checked on the object code of the compiled reference).
  void grow(int sx, int sy, std::vector<Pt>& reg, double& reg_angle, double prec) {
    reg.clear();
    const size_t s_at = (size_t)sy * w_ + sx;
    reg_angle = ang_[s_at];
    reg.push_back(Pt{sx, sy, reg_angle, mod_[s_at]});
    float sumdx = float(std::cos(reg_angle));
    float sumdy = float(std::sin(reg_angle));
    used_[s_at] = 1;
    for (size_t i = 0; i < reg.size(); ++i) {
      const int px = reg[i].x, py = reg[i].y;
      const int xx_min = std::max(px - 1, 0), xx_max = std::min(px + 1, w_ - 1);
      const int yy_min = std::max(py - 1, 0), yy_max = std::min(py + 1, h_ - 1);
      for (int yy = yy_min; yy <= yy_max; ++yy) {
        for (int xx = xx_min; xx <= xx_max; ++xx) {
          const size_t at = (size_t)yy * w_ + xx;
          if (used_[at] != 1 && aligned(xx, yy, reg_angle, prec)) {
            const double angle = ang_[at];
            used_[at] = 1;
            reg.push_back(Pt{xx, yy, angle, mod_[at]});
            sumdx += cs_[2 * at];        // cosf(float(angle))
            sumdy += cs_[2 * at + 1];    // sinf(float(angle))
            reg_angle = fast_atan2_deg(sumdy, sumdx) * kDegToRad;
          }
        }
      }
    }
  }

  double theta_of(const std::vector<Pt>& reg, double x, double y, double reg_angle, double prec) const {   // get_theta :723-757
    double Ixx = 0.0, Iyy = 0.0, Ixy = 0.0;
    for (size_t i = 0; i < reg.size(); ++i) {
      const double regx = reg[i].x, regy = reg[i].y, weight = reg[i].modgrad;
      const double dx = regx - x, dy = regy - y;
      Ixx += dy * dy * weight;
      Iyy += dx * dx * weight;
      Ixy -= dx * dy * weight;
    }
    const double lambda = 0.5 * (Ixx + Iyy - std::sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
    double theta = (std::fabs(Ixx) > std::fabs(Iyy)) ? double(fast_atan2_deg(float(lambda - Ixx), float(Ixy)))
                                                     : double(fast_atan2_deg(float(Ixy), float(lambda - Iyy)));
    theta *= kDegToRad;
    if (diff_abs(theta, reg_angle) > prec) theta += kPi;
    return theta;
  }

  void to_rect(const std::vector<Pt>& reg, double reg_angle, double prec, double p, Rect& rec) const {   // region2rect :670-721
    double x = 0, y = 0, sum = 0;
    for (size_t i = 0; i < reg.size(); ++i) {
      const double weight = reg[i].modgrad;
      x += double(reg[i].x) * weight;
      y += double(reg[i].y) * weight;
      sum += weight;
    }
    x /= sum;
    y /= sum;
    const double theta = theta_of(reg, x, y, reg_angle, prec);
    const double dx = std::cos(theta), dy = std::sin(theta);
    double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
    for (size_t i = 0; i < reg.size(); ++i) {
      const double regdx = double(reg[i].x) - x, regdy = double(reg[i].y) - y;
      const double l = regdx * dx + regdy * dy;
      const double w = -regdx * dy + regdy * dx;
      if (l > l_max) l_max = l;
      else if (l < l_min) l_min = l;
      if (w > w_max) w_max = w;
      else if (w < w_min) w_min = w;
    }
    rec.x1 = x + l_min * dx;
    rec.y1 = y + l_min * dy;
    rec.x2 = x + l_max * dx;
    rec.y2 = y + l_max * dy;
    rec.width = w_max - w_min;
    rec.x = x;
    rec.y = y;
    rec.theta = theta;
    rec.dx = dx;
    rec.dy = dy;
    rec.prec = prec;
    rec.p = p;
    if (rec.width < 1.0) rec.width = 1.0;
  }

  bool refine(std::vector<Pt>& reg, double reg_angle, double prec, double p, Rect& rec, double density_th) {   // :759-813
    double density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    if (density >= density_th) return true;
    const double xc = double(reg[0].x), yc = double(reg[0].y), ang_c = reg[0].angle;
    double sum = 0, s_sum = 0;
    int n = 0;
    for (size_t i = 0; i < reg.size(); ++i) {
      used_[(size_t)reg[i].y * w_ + reg[i].x] = 0;
      if (dist(xc, yc, reg[i].x, reg[i].y) < rec.width) {
        const double ang_d = diff_signed(reg[i].angle, ang_c);
        sum += ang_d;
        s_sum += ang_d * ang_d;
        ++n;
      }
    }
    const double mean_angle = sum / double(n);
    const double tau = 2.0 * std::sqrt((s_sum - 2.0 * mean_angle * sum) / double(n) + mean_angle * mean_angle);
    const int rx = reg[0].x, ry = reg[0].y;
    grow(rx, ry, reg, reg_angle, tau);
    if (reg.size() < 2) return false;
    to_rect(reg, reg_angle, prec, p, rec);
    density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    if (density < density_th) return shrink(reg, reg_angle, prec, p, rec, density, density_th);
    return true;
  }

  bool shrink(std::vector<Pt>& reg, double reg_angle, double prec, double p, Rect& rec, double density, double density_th) {   // :815-852
    const double xc = double(reg[0].x), yc = double(reg[0].y);
    const double r1 = dist2(xc, yc, rec.x1, rec.y1), r2 = dist2(xc, yc, rec.x2, rec.y2);
    double rad2 = r1 > r2 ? r1 : r2;
    while (density < density_th) {
      rad2 *= 0.75 * 0.75;
      for (size_t i = 0; i < reg.size(); ++i) {
        if (dist2(xc, yc, double(reg[i].x), double(reg[i].y)) > rad2) {
          used_[(size_t)reg[i].y * w_ + reg[i].x] = 0;
          std::swap(reg[i], reg[reg.size() - 1]);
          reg.pop_back();
          --i;
        }
      }
      if (reg.size() < 2) return false;
      to_rect(reg, reg_angle, prec, p, rec);
      density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    }
    return true;
  }

  double improve(Rect& rec, double log_eps) const {   // rect_improve :854-956
    const double delta = 0.5, delta_2 = delta / 2.0;
    double log_nfa = rect_nfa(rec);
    if (log_nfa > log_eps) return log_nfa;
    Rect r = rec;
    for (int n = 0; n < 5; ++n) {
      r.p /= 2;
      r.prec = r.p * kPi;
      const double v = rect_nfa(r);
      if (v > log_nfa) { log_nfa = v; rec = r; }
    }
    if (log_nfa > log_eps) return log_nfa;
    r = rec;
    for (unsigned n = 0; n < 5; ++n) {
      if ((r.width - delta) >= 0.5) {
        r.width -= delta;
        const double v = rect_nfa(r);
        if (v > log_nfa) { rec = r; log_nfa = v; }
      }
    }
    if (log_nfa > log_eps) return log_nfa;
    r = rec;
    for (unsigned n = 0; n < 5; ++n) {
      if ((r.width - delta) >= 0.5) {
        r.x1 += -r.dy * delta_2;
        r.y1 += r.dx * delta_2;
        r.x2 += -r.dy * delta_2;
        r.y2 += r.dx * delta_2;
        r.width -= delta;
        const double v = rect_nfa(r);
        if (v > log_nfa) { rec = r; log_nfa = v; }
      }
    }
    if (log_nfa > log_eps) return log_nfa;
    r = rec;
    for (unsigned n = 0; n < 5; ++n) {
      if ((r.width - delta) >= 0.5) {
        r.x1 -= -r.dy * delta_2;
        r.y1 -= r.dx * delta_2;
        r.x2 -= -r.dy * delta_2;
        r.y2 -= r.dx * delta_2;
        r.width -= delta;
        const double v = rect_nfa(r);
        if (v > log_nfa) { rec = r; log_nfa = v; }
      }
    }
    if (log_nfa > log_eps) return log_nfa;
    r = rec;
    for (unsigned n = 0; n < 5; ++n) {
      if ((r.width - delta) >= 0.5) {
        r.p /= 2;
        r.prec = r.p * kPi;
        const double v = rect_nfa(r);
        if (v > log_nfa) { rec = r; log_nfa = v; }
      }
    }
    return log_nfa;
  }

  // rect_nfa :958-1081: the rectangle's pixels are walked between two edges that advance by INTEGER-quotient steps (the
  // reference divides ints) and whose second slopes compare a y with an x (`tailp->p.x`): reproduced as written.
  double rect_nfa(const Rect& rec) const {
    int total_pts = 0, alg_pts = 0;
    const double half_width = rec.width / 2.0;
    const double dyhw = rec.dy * half_width, dxhw = rec.dx * half_width;
    struct Corner { int x, y; bool taken; };
    Corner c[4];
    c[0] = Corner{int(rec.x1 - dyhw), int(rec.y1 + dxhw), false};
    c[1] = Corner{int(rec.x2 - dyhw), int(rec.y2 + dxhw), false};
    c[2] = Corner{int(rec.x2 + dyhw), int(rec.y2 - dxhw), false};
    c[3] = Corner{int(rec.x1 + dyhw), int(rec.y1 - dxhw), false};
    std::sort(c, c + 4, [](const Corner& a, const Corner& b) { return a.x == b.x ? a.y < b.y : a.x < b.x; });
    Corner* min_y = &c[0];
    Corner* max_y = &c[0];
    for (unsigned i = 1; i < 4; ++i) {
      if (min_y->y > c[i].y) min_y = &c[i];
      if (max_y->y < c[i].y) max_y = &c[i];
    }
    min_y->taken = true;
    Corner* leftmost = nullptr;
    for (unsigned i = 0; i < 4; ++i)
      if (!c[i].taken) {
        if (!leftmost) leftmost = &c[i];
        else if (leftmost->x > c[i].x) leftmost = &c[i];
      }
    leftmost->taken = true;
    Corner* rightmost = nullptr;
    for (unsigned i = 0; i < 4; ++i)
      if (!c[i].taken) {
        if (!rightmost) rightmost = &c[i];
        else if (rightmost->x < c[i].x) rightmost = &c[i];
      }
    rightmost->taken = true;
    Corner* tailp = nullptr;
    for (unsigned i = 0; i < 4; ++i)
      if (!c[i].taken) {
        if (!tailp) tailp = &c[i];
        else if (tailp->x > c[i].x) tailp = &c[i];
      }
    tailp->taken = true;
    const double flstep = (min_y->y != leftmost->y) ? (min_y->x - leftmost->x) / (min_y->y - leftmost->y) : 0;
    const double slstep = (leftmost->y != tailp->x) ? (leftmost->x - tailp->x) / (leftmost->y - tailp->x) : 0;
    const double frstep = (min_y->y != rightmost->y) ? (min_y->x - rightmost->x) / (min_y->y - rightmost->y) : 0;
    const double srstep = (rightmost->y != tailp->x) ? (rightmost->x - tailp->x) / (rightmost->y - tailp->x) : 0;
    double lstep = flstep, rstep = frstep;
    double left_x = min_y->x, right_x = min_y->x;
    const int min_iter = min_y->y, max_iter = max_y->y;
    for (int y = min_iter; y <= max_iter; ++y) {
      if (y < 0 || y >= h_) continue;
      for (int x = int(left_x); x <= int(right_x); ++x) {
        if (x < 0 || x >= w_) continue;
        ++total_pts;
        if (aligned(x, y, rec.theta, rec.prec)) ++alg_pts;
      }
      if (y >= leftmost->y) lstep = slstep;
      if (y >= rightmost->y) rstep = srstep;
      left_x += lstep;
      right_x += rstep;
    }
    return nfa(total_pts, alg_pts, rec.p);
  }

  static double lgamma_w(double x) {   // log_gamma_windschitl
    return 0.918938533204673 + (x - 0.5) * std::log(x) - x + 0.5 * x * std::log(x * std::sinh(1 / x) + 1 / (810.0 * std::pow(x, 6.0)));
  }
  static double lgamma_l(double x) {   // log_gamma_lanczos
    static const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511};
    double a = (x + 0.5) * std::log(x + 5.5) - (x + 5.5);
    double b = 0;
    for (int n = 0; n < 7; ++n) {
      a -= std::log(x + double(n));
      b += q[n] * std::pow(x, double(n));
    }
    return a + std::log(b);
  }
  static double lgamma_(double x) { return x > 15.0 ? lgamma_w(x) : lgamma_l(x); }

  double nfa(int n, int k, double p) const {   // :1083-1119
    if (n == 0 || k == 0) return -log_nt_;
    if (n == k) return -log_nt_ - double(n) * std::log10(p);
    const double p_term = p / (1 - p);
    const double log1term = (double(n) + 1) - lgamma_(double(k) + 1) - lgamma_(double(n - k) + 1) + double(k) * std::log(p) +
                            double(n - k) * std::log(1.0 - p);
    double term = std::exp(log1term);
    if (rel_equal(term, 0)) {
      if (k > n * p) return -log1term / 2.30258509299404568402 - log_nt_;
      return -log_nt_;
    }
    double bin_tail = term;
    const double tolerance = 0.1;
    for (int i = k + 1; i <= n; ++i) {
      const double bin_term = double(n - i + 1) / double(i);
      const double mult_term = bin_term * p_term;
      term *= mult_term;
      bin_tail += term;
      if (bin_term < 1) {
        const double err = term * ((1 - std::pow(mult_term, double(n - i + 1))) / (1 - mult_term) - 1);
        if (err < tolerance * std::fabs(-std::log10(bin_tail) - log_nt_) * bin_tail) break;
      }
    }
    return -std::log10(bin_tail) - log_nt_;
  }
};

// LSDDetectorC::detectImpl's check of a segment's end points (LSDDetector_custom.cpp:86-133; borders 0: no pyramid was set)
inline void clamp_extremes(float e[4], int width, int height) {
  const int minX = 0, minY = 0;
  const int maxX = (float)width - 1.0f;
  const int maxY = (float)height - 1.0f;
  if (e[0] < minX) e[0] = minX;
  if (e[0] > maxX) e[0] = maxX;
  if (e[2] < minX) e[2] = minX;
  if (e[2] > maxX) e[2] = maxX;
  if (e[1] < minY) e[1] = minY;
  if (e[1] > maxY) e[1] = maxY;
  if (e[3] < minY) e[3] = minY;
  if (e[3] > maxY) e[3] = maxY;
}

}  // namespace lsd
}  // namespace plvs
