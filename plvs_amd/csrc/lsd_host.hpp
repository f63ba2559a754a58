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

// A helper thread that is joined when its owner goes out of scope, whatever way it leaves (an exception between the start
// of a helper and its join would otherwise destroy a joinable std::thread: std::terminate inside the C ABI).
struct JoinedThread {
  std::thread t;
  JoinedThread() = default;
  explicit JoinedThread(std::thread&& th) : t(std::move(th)) {}
  JoinedThread(JoinedThread&&) = default;
  JoinedThread& operator=(JoinedThread&& o) {
    if (t.joinable()) t.join();
    t = std::move(o.t);
    return *this;
  }
  ~JoinedThread() {
    if (t.joinable()) t.join();
  }
};

// std::sort's result — libstdc++'s introsort — with the halves of its partitions on several threads.  The reference's
// ordering is an UNSTABLE sort whose comparator sees part of the element only: the order of equal keys is whatever this
// implementation of std::sort does, and the only way to have it is to run its steps.  They are run here through libstdc++'s
// own building blocks: __introsort_loop is, per range, { partition around the median of three; recurse into the right part;
// continue with the left }, the recursion touches nothing outside its range, so the right part can go to another thread —
// same comparisons, same moves, same array afterwards; __final_insertion_sort then runs over the whole array as std::sort does.
// Those building blocks are private to libstdc++: the first use compares the threaded form with std::sort itself on a probe
// (sort_threads_ok) and a library whose internals have moved on simply keeps the one-thread std::sort.
// (tests/test_lsd.py compares with the reference's own std::sort call, compiled by g++ from its sources.)
#if defined(__GLIBCXX__)
struct SortPool {
  std::vector<JoinedThread> threads;   // (joined by its destructor at the latest)
  std::mutex mu;
};
template <typename It, typename Comp>
void introsort_loop_mt(It first, It last, long depth, Comp comp, int level, SortPool& pool) {
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
      std::lock_guard<std::mutex> g(pool.mu);
      try {
        pool.threads.emplace_back(std::thread([=, &pool] { introsort_loop_mt(cut, last, depth, comp, level, pool); }));
        handed_over = true;
      } catch (...) {   // (no thread to be had: the right part is sorted here, as std::sort would)
      }
    }
    if (!handed_over) std::__introsort_loop(cut, last, depth, comp);
    last = cut;
  }
}
template <typename It, typename Cmp>
void sort_threaded(It first, It last, Cmp cmp) {
  auto comp = __gnu_cxx::__ops::__iter_comp_iter(cmp);
  SortPool pool;
  pool.threads.reserve(16);   // (at most 7 are made; no reallocation while a worker holds a reference)
  introsort_loop_mt(first, last, std::__lg(last - first) * 2, comp, 0, pool);
  for (size_t i = 0;; ++i) {
    std::thread t;
    {
      std::lock_guard<std::mutex> g(pool.mu);
      if (i >= pool.threads.size()) break;
      t = std::move(pool.threads[i].t);
    }
    t.join();
  }
  std::__final_insertion_sort(first, last, comp);
}
// Does the threaded form give std::sort's permutation with THIS libstdc++?  Checked once per process on 2^17 keys of 64
// distinct bins (long runs of equal keys: the case that decides the order of LSD's seeds).
inline bool sort_threads_ok() {
  static const bool ok = [] {
    std::vector<uint32_t> a((size_t)1 << 17), b;
    uint32_t x = 2463534242u;
    for (size_t i = 0; i < a.size(); ++i) {
      x ^= x << 13; x ^= x >> 17; x ^= x << 5;
      a[i] = ((x & 63u) << 22) | (uint32_t)i;
    }
    b = a;
    auto by_bin = [](const uint32_t& p, const uint32_t& q) { return (p >> 22) > (q >> 22); };
    try {
      sort_threaded(a.begin(), a.end(), by_bin);
    } catch (...) {
      return false;
    }
    std::sort(b.begin(), b.end(), by_bin);
    return a == b;
  }();
  return ok;
}
template <typename It, typename Cmp>
void sort_as_std(It first, It last, Cmp cmp, bool threads) {
  if (!threads || last - first < 65536 || !sort_threads_ok()) {
    std::sort(first, last, cmp);
    return;
  }
  sort_threaded(first, last, cmp);
}
#else
template <typename It, typename Cmp>
void sort_as_std(It first, It last, Cmp cmp, bool) { std::sort(first, last, cmp); }
#endif

// One level of the detector.  What the reference does per seed (lsd_custom.cpp:461-547: grow a region of aligned pixels,
// fit a rectangle, tighten it until it is dense, improve its NFA, keep it if meaningful) on this file's own data:
//   * a region is a list of pixel INDICES with their coordinates (angle and gradient norm stay in the field arrays);
//   * the rectangle's pixels are enumerated as one [first, last] column SPAN per image row (built once per rectangle,
//     clipped to the image once per row) and counted over the row of the field — the reference tests every pixel's
//     coordinates against the image inside its double loop;
//   * the NFA search is ONE loop over a table of moves (halve the tolerance, narrow, shift one side in, the other, halve
//     again) instead of five copies of the loop; log-gamma of the integer arguments the binomial tail needs is memoised.
// The arithmetic — operand types, order of the sums (region order), which libm function — is the reference's: the segments
// are compared bit for bit with those of its compiled sources (tests/test_lsd.py).
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
    auto fill_cs = [this, angles, w](int y0, int y1) {
      for (size_t at = (size_t)y0 * w, end = (size_t)y1 * w; at < end; ++at)
        if (angles[at] != kNotDef) {
          cs_[2 * at] = cosf(float(angles[at]));
          cs_[2 * at + 1] = sinf(float(angles[at]));
        }
    };
    const int helpers = std::thread::hardware_concurrency() >= 4 ? 2 : 0;
    bool inline_part[2] = {helpers < 1, helpers < 2};
    {
      JoinedThread th[2];   // (joined when this block ends — also when the ordering throws)
      for (int i = 0; i < helpers; ++i) {
        try {
          th[i] = JoinedThread(std::thread(fill_cs, (h * i) / helpers, (h * (i + 1)) / helpers));
        } catch (...) {   // (no thread to be had: this part after the ordering, on this one)
          inline_part[i] = true;
        }
      }
      const double bin_coef = (max_grad > 0) ? double(o.n_bins - 1) / max_grad : 0;
      narrow_ = npx <= (size_t(1) << 22) && o.n_bins <= 1024;
      const char* force = getenv("PLVS_LSD_SORT_THREADS");   // 0 / 1: never / always (tests); default: where there are cores to spare
      const bool mt = force ? force[0] == '1' : std::thread::hardware_concurrency() >= 16;
      if (narrow_) order_seeds<uint32_t, 22>(modgrad, w, h, bin_coef, keys32_, mt);
      else order_seeds<uint64_t, 32>(modgrad, w, h, bin_coef, keys64_, mt);
    }
    if (helpers == 0) fill_cs(0, h);
    else
      for (int i = 0; i < helpers; ++i)
        if (inline_part[i]) fill_cs((h * i) / helpers, (h * (i + 1)) / helpers);
    const size_t nseeds = narrow_ ? keys32_.size() : keys64_.size();
    ms_order = now_ms_() - t_a;

    const double prec = kPi * o.ang_th / 180;
    const double p = o.ang_th / 180;
    log_nt_ = 5 * (std::log10(double(w)) + std::log10(double(h))) / 2 + std::log10(11.0);
    const size_t min_reg_size = size_t(-log_nt_ / std::log10(p));
    // (a pixel without a level-line angle can neither seed a region nor join one: it starts out as taken, and the growth loop
    // never loads its angle — two pixels in five of a camera image)
    used_.resize(npx);
    for (size_t at = 0; at < npx; ++at) used_[at] = angles[at] == kNotDef ? 1 : 0;
    for (size_t i = 0; i < nseeds; ++i) {
      const size_t at = narrow_ ? size_t(keys32_[i] & ((1u << 22) - 1)) : size_t(keys64_[i] & 0xffffffffull);
      if (used_[at] != 0) continue;
      double run_angle = collect(at, prec);
      if (px_.size() < min_reg_size) continue;
      Box box;
      fit_box(run_angle, prec, p, box);
      if (o.refine > 0) {
        if (!make_dense(run_angle, prec, p, box, o.density_th)) continue;
        if (o.refine >= 2 && best_nfa(box, o.log_eps) <= o.log_eps) continue;
      }
      // (:523-537) half a pixel, then back to the scale of the input
      double e[4] = {box.ax + 0.5, box.ay + 0.5, box.bx + 0.5, box.by + 0.5};
      if (o.scale != 1)
        for (double& v : e) v /= o.scale;
      out.push_back(Segment4{float(e[0]), float(e[1]), float(e[2]), float(e[3])});
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

  struct Px { uint32_t at; int32_t x, y; };   // a pixel of the region in hand: index in the field, coordinates
  struct Box {                                // a region's rectangle, in the numbers the reference's NFA moves act on
    double ax, ay, bx, by;                    //   ends of the axis (rect::x1, y1, x2, y2)
    double width, theta, ux, uy;              //   width, direction and its unit vector (dx, dy)
    double prec, p;                           //   angle tolerance and the probability that goes with it
  };
  const double* ang_ = nullptr;
  const double* mod_ = nullptr;
  int w_ = 0, h_ = 0;
  bool narrow_ = true;
  double log_nt_ = 0;
  std::vector<uint32_t> keys32_;
  std::vector<uint64_t> keys64_;
  std::vector<float> cs_;          // cosf, sinf of the angle of every defined pixel
  std::vector<uint8_t> used_;
  std::vector<Px> px_;             // the region in hand, in the order its pixels joined
  std::vector<double> lgam_;       // log_gamma(m) for integer m, as far as asked for (NaN: not yet)
  struct Span { int y, x0, x1; };
  mutable std::vector<Span> spans_;

  static double now_ms_() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
  }
  static double sq_dist(double x1, double y1, double x2, double y2) { return (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1); }
  static double wrap_pi(double d) {   // angle_diff_signed's result for the difference d (:1090-1097)
    while (d <= -kPi) d += (2 * kPi);
    while (d > kPi) d -= (2 * kPi);
    return d;
  }
  // isAligned (:1121-1137) for a pixel inside the field
  static bool within(double a, double theta, double tol) {
    if (a == kNotDef) return false;
    double d = std::fabs(theta - a);
    if (d > (3 * kPi) / 2) d = std::fabs(d - (2 * kPi));
    return d <= tol;
  }

  // region_grow (:613-668) from the pixel `seed` with tolerance `tol`: px_ = the region, the return value its direction.
  // A pixel joins when it is free and within `tol` of the direction the region has at that moment; the direction is the
  // angle of the FLOAT sum of the members' (cosf, sinf) — `cos(float(angle))` resolves to the float overload in the
  // reference (checked on its object code) — so the order of the 3 x 3 scan (rows, then columns) is part of the result.
  double collect(size_t seed, double tol) {
    px_.clear();
    double dir = ang_[seed];
    px_.push_back(Px{(uint32_t)seed, int32_t(seed % (size_t)w_), int32_t(seed / (size_t)w_)});
    float sx = float(std::cos(dir)), sy = float(std::sin(dir));   // (the seed's own terms: double functions, rounded)
    used_[seed] = 1;
    for (size_t i = 0; i < px_.size(); ++i) {
      const Px c = px_[i];
      const int x_lo = c.x > 0 ? c.x - 1 : 0, x_hi = c.x + 1 < w_ ? c.x + 1 : w_ - 1;
      const int y_lo = c.y > 0 ? c.y - 1 : 0, y_hi = c.y + 1 < h_ ? c.y + 1 : h_ - 1;
      for (int y = y_lo; y <= y_hi; ++y) {
        const size_t row = (size_t)y * w_;
        // (inside a grown region the three neighbours of a row are all taken: one look at the row)
        if (x_hi - x_lo == 2 && (used_[row + x_lo] & used_[row + x_lo + 1] & used_[row + x_hi]) == 1) continue;
        for (int x = x_lo; x <= x_hi; ++x) {
          const size_t at = row + x;
          if (used_[at] == 1 || !within(ang_[at], dir, tol)) continue;
          used_[at] = 1;
          px_.push_back(Px{(uint32_t)at, x, y});
          sx += cs_[2 * at];
          sy += cs_[2 * at + 1];
          dir = fast_atan2_deg(sy, sx) * kDegToRad;
        }
      }
    }
    return dir;
  }

  // region2rect + get_theta (:670-757): gradient-weighted centroid, direction of the smallest inertia, extent along and
  // across it.  Three passes in region order (the sums are doubles: their order is the result).
  void fit_box(double dir, double prec, double p, Box& b) const {
    double mx = 0, my = 0, mass = 0;
    for (const Px& q : px_) {
      const double m = mod_[q.at];
      mx += double(q.x) * m;
      my += double(q.y) * m;
      mass += m;
    }
    mx /= mass;
    my /= mass;
    double ixx = 0.0, iyy = 0.0, ixy = 0.0;
    for (const Px& q : px_) {
      const double m = mod_[q.at], rx = double(q.x) - mx, ry = double(q.y) - my;
      ixx += ry * ry * m;
      iyy += rx * rx * m;
      ixy -= rx * ry * m;
    }
    const double small = 0.5 * (ixx + iyy - std::sqrt((ixx - iyy) * (ixx - iyy) + 4.0 * ixy * ixy));   // smallest eigenvalue
    double th = (std::fabs(ixx) > std::fabs(iyy)) ? double(fast_atan2_deg(float(small - ixx), float(ixy)))
                                                  : double(fast_atan2_deg(float(ixy), float(small - iyy)));
    th *= kDegToRad;
    if (std::fabs(wrap_pi(th - dir)) > prec) th += kPi;
    const double ux = std::cos(th), uy = std::sin(th);
    double along_lo = 0, along_hi = 0, across_lo = 0, across_hi = 0;
    for (const Px& q : px_) {
      const double rx = double(q.x) - mx, ry = double(q.y) - my;
      const double along = rx * ux + ry * uy, across = -rx * uy + ry * ux;
      if (along > along_hi) along_hi = along;
      else if (along < along_lo) along_lo = along;
      if (across > across_hi) across_hi = across;
      else if (across < across_lo) across_lo = across;
    }
    b.ax = mx + along_lo * ux;
    b.ay = my + along_lo * uy;
    b.bx = mx + along_hi * ux;
    b.by = my + along_hi * uy;
    b.width = across_hi - across_lo;
    if (b.width < 1.0) b.width = 1.0;
    b.theta = th;
    b.ux = ux;
    b.uy = uy;
    b.prec = prec;
    b.p = p;
  }
  double density(const Box& b) const { return double(px_.size()) / (std::sqrt(sq_dist(b.ax, b.ay, b.bx, b.by)) * b.width); }

  // refine + reduce_region_radius (:759-852): a region too sparse for its rectangle is grown again from its first pixel
  // with the tolerance its pixels near that one suggest (twice their standard deviation), and if that is not enough it
  // loses the pixels beyond a radius that shrinks by a quarter per round.  false: nothing worth a rectangle is left.
  bool make_dense(double dir, double prec, double p, Box& b, double want) {
    if (density(b) >= want) return true;
    const Px first = px_[0];
    const double fx = double(first.x), fy = double(first.y), fa = ang_[first.at];
    double s1 = 0, s2 = 0;
    int near = 0;
    for (const Px& q : px_) {
      used_[q.at] = 0;
      if (std::sqrt(sq_dist(fx, fy, q.x, q.y)) < b.width) {
        const double d = wrap_pi(ang_[q.at] - fa);
        s1 += d;
        s2 += d * d;
        ++near;
      }
    }
    const double mean = s1 / double(near);
    const double tol = 2.0 * std::sqrt((s2 - 2.0 * mean * s1) / double(near) + mean * mean);
    dir = collect(first.at, tol);
    if (px_.size() < 2) return false;
    fit_box(dir, prec, p, b);
    double have = density(b);
    if (have >= want) return true;
    const double ra = sq_dist(fx, fy, b.ax, b.ay), rb = sq_dist(fx, fy, b.bx, b.by);
    double r2 = ra > rb ? ra : rb;
    while (have < want) {
      r2 *= 0.75 * 0.75;
      for (size_t i = 0; i < px_.size();) {   // (a pixel that goes is replaced by the last one, which is looked at next)
        if (sq_dist(fx, fy, double(px_[i].x), double(px_[i].y)) > r2) {
          used_[px_[i].at] = 0;
          px_[i] = px_.back();
          px_.pop_back();
        } else {
          ++i;
        }
      }
      if (px_.size() < 2) return false;
      fit_box(dir, prec, p, b);
      have = density(b);
    }
    return true;
  }

  // rect_improve (:854-956): five kinds of move, each tried up to five times on the rectangle the previous tries left; a
  // try that raises the NFA replaces the best rectangle; a kind is skipped once the best is meaningful.
  //   0 halve the tolerance      1 narrow by half a pixel      2 / 3 move one long side in by a quarter pixel (and narrow)
  //   4 halve the tolerance, only while the rectangle could still be narrowed
  double best_nfa(Box& best, double log_eps) {
    const double step = 0.5, half_step = step / 2.0;
    double top = box_nfa(best);
    for (int kind = 0; kind < 5 && top <= log_eps; ++kind) {
      Box t = best;
      for (int n = 0; n < 5; ++n) {
        if (kind != 0 && !((t.width - step) >= 0.5)) continue;
        if (kind == 0 || kind == 4) {
          t.p /= 2;
          t.prec = t.p * kPi;
        } else {
          if (kind == 2) {
            t.ax += -t.uy * half_step; t.ay += t.ux * half_step;
            t.bx += -t.uy * half_step; t.by += t.ux * half_step;
          } else if (kind == 3) {
            t.ax -= -t.uy * half_step; t.ay -= t.ux * half_step;
            t.bx -= -t.uy * half_step; t.by -= t.ux * half_step;
          }
          t.width -= step;
        }
        const double v = box_nfa(t);
        if (v > top) {
          top = v;
          best = t;
        }
      }
    }
    return top;
  }

  // rect_nfa (:958-1081): pixels of the rectangle and how many of them are aligned with it.  The reference walks the rows
  // between the lowest and the highest corner with two edges that start at the lowest corner and advance by INTEGER
  // quotients of corner differences (it divides ints), switching to a second slope at the rows of the left and the right
  // corner; the second slopes compare a row with the fourth corner's COLUMN (`tailp->p.x`): as written there.  Rows outside
  // the image advance nothing.
  double box_nfa(const Box& b) const {
    const double hw = b.width / 2.0, ox = b.uy * hw, oy = b.ux * hw;
    struct Corner { int x, y; };
    Corner c[4] = {{int(b.ax - ox), int(b.ay + oy)}, {int(b.bx - ox), int(b.by + oy)},
                   {int(b.bx + ox), int(b.by - oy)}, {int(b.ax + ox), int(b.ay - oy)}};
    for (int i = 1; i < 4; ++i)   // by column, then row (four elements: insertion)
      for (int j = i; j > 0 && (c[j].x < c[j - 1].x || (c[j].x == c[j - 1].x && c[j].y < c[j - 1].y)); --j) std::swap(c[j], c[j - 1]);
    int low = 0, high = 0;        // first of the lowest / highest rows
    for (int i = 1; i < 4; ++i) {
      if (c[low].y > c[i].y) low = i;
      if (c[high].y < c[i].y) high = i;
    }
    // of the other three (still in column order): the first is the left corner, the first of the largest column among the
    // remaining two the right one, the last the fourth
    int rest[3], nr = 0;
    for (int i = 0; i < 4; ++i)
      if (i != low) rest[nr++] = i;
    const int left = rest[0];
    const int right = c[rest[1]].x < c[rest[2]].x ? rest[2] : rest[1];
    const int fourth = right == rest[1] ? rest[2] : rest[1];
    auto slope = [](int dx, int dy) { return dy != 0 ? double(dx / dy) : 0.0; };
    const double l1 = slope(c[low].x - c[left].x, c[low].y - c[left].y), l2 = slope(c[left].x - c[fourth].x, c[left].y - c[fourth].x);
    const double r1 = slope(c[low].x - c[right].x, c[low].y - c[right].y), r2 = slope(c[right].x - c[fourth].x, c[right].y - c[fourth].x);
    spans_.clear();
    double lx = c[low].x, rx = c[low].x, ls = l1, rs = r1;
    for (int y = c[low].y; y <= c[high].y; ++y) {
      if (y < 0 || y >= h_) continue;
      spans_.push_back(Span{y, int(lx), int(rx)});
      if (y >= c[left].y) ls = l2;
      if (y >= c[right].y) rs = r2;
      lx += ls;
      rx += rs;
    }
    int total = 0, good = 0;
    for (const Span& sp : spans_) {
      const int x0 = sp.x0 > 0 ? sp.x0 : 0, x1 = sp.x1 < w_ - 1 ? sp.x1 : w_ - 1;
      if (x1 < x0) continue;
      total += x1 - x0 + 1;
      const double* row = ang_ + (size_t)sp.y * w_;
      for (int x = x0; x <= x1; ++x) good += within(row[x], b.theta, b.prec) ? 1 : 0;
    }
    return tail_nfa(total, good, b.p);
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
  double lgamma_int(int m) const {   // log_gamma(double(m)), memoised (the same call on the same argument)
    std::vector<double>& t = const_cast<std::vector<double>&>(lgam_);
    if ((size_t)m >= t.size()) t.resize((size_t)m + 1024, std::nan(""));
    if (t[(size_t)m] != t[(size_t)m]) t[(size_t)m] = double(m) > 15.0 ? lgamma_w(double(m)) : lgamma_l(double(m));
    return t[(size_t)m];
  }

  // nfa (:1083-1119): -log10 of the number of false alarms of k aligned pixels among n at probability p — the binomial
  // tail, summed until the rest is provably below a tenth of the result.  (Its first term carries `n + 1` where the formula
  // has log_gamma(n + 1): as in the reference.)
  double tail_nfa(int n, int k, double p) const {
    if (n == 0 || k == 0) return -log_nt_;
    if (n == k) return -log_nt_ - double(n) * std::log10(p);
    const double odds = p / (1 - p);
    const double log_first = (double(n) + 1) - lgamma_int(k + 1) - lgamma_int(n - k + 1) + double(k) * std::log(p) +
                             double(n - k) * std::log(1.0 - p);
    double term = std::exp(log_first);
    {   // double_equal(term, 0): relative difference within 100 epsilon
      const double mag = std::fabs(term) < DBL_MIN ? DBL_MIN : std::fabs(term);
      if (term == 0.0 || (std::fabs(term) / mag) <= (100.0 * DBL_EPSILON))
        return (k > n * p) ? -log_first / 2.30258509299404568402 - log_nt_ : -log_nt_;
    }
    double tail = term;
    for (int i = k + 1; i <= n; ++i) {
      const double ratio = double(n - i + 1) / double(i), mult = ratio * odds;
      term *= mult;
      tail += term;
      if (ratio < 1) {
        const double rest = term * ((1 - std::pow(mult, double(n - i + 1))) / (1 - mult) - 1);
        if (rest < 0.1 * std::fabs(-std::log10(tail) - log_nt_) * tail) break;
      }
    }
    return -std::log10(tail) - log_nt_;
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
