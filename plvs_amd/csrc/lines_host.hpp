// Host-side (sequential, order-dependent) stages of the EDLines detector.
//
// EdgeDrawing's smart routing (binary_descriptor_custom.cpp:1698-2349) consumes
// anchors in scan order and marks pixels as it walks, the incremental
// least-squares line fit (:2409-2654, :2656-2815) extends segments pixel by
// pixel with float accumulators, and the Helmholtz validation (:2817-2898, nfa
// descriptor_custom.hpp:763-845) closes each segment: none of it is data
// parallel, all of it is cheap.  The device produces the per-pixel maps
// (blur, Sobel, gradient magnitude / direction); this header turns them into
// line segments, octave by octave (one host thread per octave), then groups
// the segments across octaves (:903-1149) and flattens them to KeyLines (:504-555).
#pragma once
#include <time.h>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <emmintrin.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

namespace plvs {
namespace lines {

// ---------------------------------------------------------------- NFA
inline bool nearly_equal(double a, double b) {
  if (a == b) return true;
  const double diff = fabs(a - b), aa = fabs(a), bb = fabs(b);
  double m = aa > bb ? aa : bb;
  if (m < DBL_MIN) m = DBL_MIN;
  return (diff / m) <= (100.0 * DBL_EPSILON);
}
inline double lgamma_lanczos(double x) {
  static const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705,
                              1168.92649479, 83.8676043424, 2.50662827511};
  double a = (x + 0.5) * log(x + 5.5) - (x + 5.5), b = 0.0;
  for (int n = 0; n < 7; ++n) {
    a -= log(x + (double)n);
    b += q[n] * pow(x, (double)n);
  }
  return a + log(b);
}
inline double lgamma_windschitl(double x) {
  return 0.918938533204673 + (x - 0.5) * log(x) - x +
         0.5 * x * log(x * std::sinh(1 / x) + 1 / (810.0 * pow(x, 6.0)));
}
inline double lgamma_any(double x) { return x > 15.0 ? lgamma_windschitl(x) : lgamma_lanczos(x); }

// -log10(NFA) of k aligned points among n, p = 1/8
inline double nfa(int n, int k, double p, double logNT) {
  if (n == 0 || k == 0) return -logNT;
  if (n == k) return -logNT - (double)n * log10(p);
  const double p_term = p / (1.0 - p);
  const double log1term = lgamma_any((double)n + 1.0) - lgamma_any((double)k + 1.0) -
                          lgamma_any((double)(n - k) + 1.0) + (double)k * log(p) +
                          (double)(n - k) * log(1.0 - p);
  double term = exp(log1term);
  if (nearly_equal(term, 0.0)) {
    if ((double)k > (double)n * p) return -log1term / 2.30258509299404568402 - logNT;
    return -logNT;
  }
  double tail = term;
  for (int i = k + 1; i <= n; ++i) {
    const double bin_term = (double)(n - i + 1) / (double)i;
    const double mult_term = bin_term * p_term;
    term *= mult_term;
    tail += term;
    if (bin_term < 1.0) {
      const double err = term * ((1.0 - pow(mult_term, (double)(n - i + 1))) / (1.0 - mult_term) - 1.0);
      if (err < 0.1 * fabs(-log10(tail) - logNT) * tail) break;
    }
  }
  return -log10(tail) - logNT;
}

// ------------------------------------------------------------ per octave
struct Segment {              // one validated line segment of an octave
  double eq[3];               // w1 x + w2 y + w3 = 0, w1^2 + w2^2 = 1
  float ep[4];                // endpoints (x0, y0, x1, y1)
  float direction;
  int num_pixels;
};

struct OctaveMaps {           // what the device hands over for one octave
  int w = 0, h = 0;
  const uint16_t* gd = nullptr;   // bits 0..8: gradient magnitude / 4 (0 below threshold), bit 15: |dx| < |dy|
  const int16_t* dx = nullptr;    // Sobel derivatives (for the validation)
  const int16_t* dy = nullptr;
  // optional: anchor flags of the stride-2 scan grid in scan (column-major) order,
  // flag[(x-1)/2 * rows + (y-1)/2] for x = 1,3,.. and y = 1,3,.. (rows = number of scanned y)
  const uint8_t* anchors = nullptr;
};

struct EdParams {
  int anchor_threshold = 8, scan_interval = 2, min_line_len = 15;
  double fit_err_threshold = 1.6;
};

inline double clock_ms() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

// Progress of one octave's EdgeDrawing, read by the threads that fit its chains while
// the routing is still running: chains [0, ready) are complete; done is set at the end.
// The routing thread publishes `ready` once per kEdgeChainBatch chains (the size of a fitting task) and the two words live
// on cache lines of their own: with a store per chain next to the routing's own state, the polling fitters kept
// pulling that line away from it (EdgeDrawing of a 640x480 octave: 0.46 ms alone, 0.60 ms beside two fitters,
// 1.0 ms beside eight).
constexpr int kEdgeChainBatch = 24;
struct ChainProgress {
  alignas(64) std::atomic<int> ready{0};
  alignas(64) std::atomic<int> done{0};
  char pad_[64 - sizeof(std::atomic<int>)];
};

class OctaveDetector {
 public:
  std::vector<Segment> segments;
  ChainProgress* progress = nullptr;   // optional; set by the caller before prepare()
  double ms_draw = 0, ms_fit = 0, ms_validate = 0, ms_anchor = 0, ms_route = 0;
  int n_anchor = 0, n_walked = 0;
  bool failed = false;   // wall-clock split of the last run

  // returns false on the reference's failure paths ("lines not found", buffer overrun)
  bool run(const OctaveMaps& m, const EdParams& prm) {
    segments.clear();
    W = m.w; H = m.h; maps = m; P = prm;
    logNT = 2.0 * (log10((double)(unsigned)W) + log10((double)(unsigned)H));
    const double t0 = clock_ms();
    const bool ok1 = draw_edges();
    const double t1 = clock_ms();
    const bool ok2 = ok1 && fit_lines();
    ms_draw = t1 - t0;
    ms_fit = clock_ms() - t1;
    return ok2;
  }

 private:
  int W = 0, H = 0;
  OctaveMaps maps;
  EdParams P;
  std::vector<uint16_t> cx, cy;       // edge chains, pixel by pixel
  std::vector<uint16_t> fx_, fy_, sx_, sy_;   // scratch kept across calls (no reallocation per frame)
  std::vector<uint16_t> wk_;
  int n_chains = 0;
  std::vector<uint32_t> cstart;       // chain c = [cstart[c], cstart[c+1])
  double logNT = 0;
  struct FitState { float ata[4] = {0, 0, 0, 0}, atv[2] = {0, 0}; };

  int g(int idx) const { return maps.gd[idx] & 0x1ff; }
  bool horizontal(int idx) const { return (maps.gd[idx] & 0x8000u) != 0; }

  // ---- EdgeDrawing: anchors (column-major scan) + smart routing
  bool draw_edges() {
    const double t_begin = clock_ms();
    const size_t npix = (size_t)W * H;
    const size_t cap = npix / 5, max_edges = cap / 20;
    std::vector<uint16_t> ax, ay;
    if (maps.anchors != nullptr && P.scan_interval == 2) {
      // the device already evaluated the anchor test; its flag map is laid out in
      // scan order, so this is one linear pass
      const int rows = (H - 1) / 2, cols = (W - 1) / 2;   // y = 1,3,.. < H-1 ; x = 1,3,.. < W-1
      const uint8_t* f = maps.anchors;
      const size_t nflags = (size_t)rows * cols;           // flat index = c * rows + r: already the scan order
      auto take = [&](size_t i) -> bool {
        if (ax.size() >= cap) return false;
        const size_t c = i / (size_t)rows;
        ax.push_back((uint16_t)(2 * c + 1));
        ay.push_back((uint16_t)(2 * (i - c * rows) + 1));
        return true;
      };
      size_t i = 0;
      for (; i + 8 <= nflags; i += 8) {                    // anchors are sparse: eight flags per test
        uint64_t wd;
        memcpy(&wd, f + i, 8);
        if (wd == 0) continue;
        for (size_t k = 0; k < 8; ++k)
          if (f[i + k] && !take(i + k)) return false;
      }
      for (; i < nflags; ++i)
        if (f[i] && !take(i)) return false;
    } else {
      for (int x = 1; x < W - 1; x += P.scan_interval)
        for (int y = 1; y < H - 1; y += P.scan_interval) {
          const int i = y * W + x, gi = g(i);
          const bool a = horizontal(i) ? (gi >= g(i - W) + P.anchor_threshold && gi >= g(i + W) + P.anchor_threshold)
                                       : (gi >= g(i - 1) + P.anchor_threshold && gi >= g(i + 1) + P.anchor_threshold);
          if (a) {
            if (ax.size() >= cap) return false;
            ax.push_back((uint16_t)x);
            ay.push_back((uint16_t)y);
          }
        }
    }
    const double ta = clock_ms();
    // working copy of the packed map: bits 0..8 magnitude, bit 15 orientation, bit 14 = visited
    wk_.resize(npix);
    memcpy(wk_.data(), maps.gd, npix * sizeof(uint16_t));
    uint16_t* wk = wk_.data();
    constexpr uint16_t kUsed = 0x4000, kMag = 0x1ff;
    // the two halves of every edge, as the reference stores them (capacity = its buffer size)
    fx_.resize(cap); fy_.resize(cap); sx_.resize(cap); sy_.resize(cap);
    uint16_t* half_x[2] = {fx_.data(), sx_.data()};
    uint16_t* half_y[2] = {fy_.data(), sy_.data()};
    size_t half_n[2] = {0, 0};
    enum { kUp = 1, kRight = 2, kDown = 3, kLeft = 4 };
    // candidate moves per heading: {diag1, straight, diag3} as (dx, dy) and as index deltas
    static const int mv[5][3][2] = {{{0, 0}, {0, 0}, {0, 0}},
                                    {{1, -1}, {0, -1}, {-1, -1}},   // up
                                    {{1, -1}, {1, 0}, {1, 1}},      // right
                                    {{1, 1}, {0, 1}, {-1, 1}},      // down
                                    {{-1, -1}, {-1, 0}, {-1, 1}}};  // left
    int dl[5][3];
    for (int hdg = 0; hdg < 5; ++hdg)
      for (int k = 0; k < 3; ++k) dl[hdg][k] = mv[hdg][k][1] * W + mv[hdg][k][0];
    const int Wm1 = W - 1, Hm1 = H - 1;
    int lastX = 0, lastY = 0;   // persists across walks, as in the reference
    size_t walked = 0;
    auto walk = [&](int x, int y, int heading, int half) -> bool {
      uint16_t* ox = half_x[half];
      uint16_t* oy = half_y[half];
      size_t n = half_n[half];
      int idx = y * W + x;
      uint16_t v = wk[idx];
      // continue while the magnitude is non-zero and the pixel is unvisited
      while ((v & kMag) != 0 && (v & kUsed) == 0) {
        wk[idx] = v | kUsed;
        if (n >= cap) return false;
        ox[n] = (uint16_t)x;
        oy[n++] = (uint16_t)y;
        int go;
        if (v & 0x8000u) {   // horizontal edge here
          if (heading == kUp || heading == kDown) go = (x > lastX) ? kRight : kLeft;
          else go = heading;
        } else {
          if (heading == kRight || heading == kLeft) go = (y > lastY) ? kDown : kUp;
          else go = heading;
        }
        lastX = x; lastY = y;
        // image border in the direction of travel ends the walk
        bool border;
        switch (go) {
          case kRight: border = x == Wm1 || y == 0 || y == Hm1; break;
          case kLeft: border = x == 0 || y == 0 || y == Hm1; break;
          case kDown: border = x == 0 || x == Wm1 || y == Hm1; break;
          default: border = x == 0 || x == Wm1 || y == 0; break;
        }
        if (border) break;
        // the reference compares the neighbours' magnitudes truncated to 8 bits
        const int g1 = wk[idx + dl[go][0]] & 0xff;
        const int g2 = wk[idx + dl[go][1]] & 0xff;
        const int g3 = wk[idx + dl[go][2]] & 0xff;
        const int pick = (g1 >= g2 && g1 >= g3) ? 0 : ((g3 >= g2 && g3 >= g1) ? 2 : 1);
        x += mv[go][pick][0];
        y += mv[go][pick][1];
        idx += dl[go][pick];
        heading = go;
        v = wk[idx];
      }
      walked += n - half_n[half];
      half_n[half] = n;
      return true;
    };
    // chains are published edge by edge: cx/cy/cstart never reallocate during the routing
    cx.resize(2 * cap); cy.resize(2 * cap);
    cstart.resize(max_edges + 2);
    n_chains = 0;
    size_t w = 0;
    cstart[0] = 0;
    const uint16_t* fx = half_x[0]; const uint16_t* fy = half_y[0];
    const uint16_t* sx = half_x[1]; const uint16_t* sy = half_y[1];
    for (size_t a = 0; a < ax.size(); ++a) {
      const int x = ax[a], y = ay[a], idx = y * W + x;
      if (wk[idx] & kUsed) continue;
      if ((size_t)n_chains >= max_edges) return false;
      const size_t f0 = half_n[0], s0 = half_n[1];
      const bool hor = (wk[idx] & 0x8000u) != 0;
      if (!walk(x, y, hor ? kRight : kDown, 0)) return false;
      wk[idx] &= (uint16_t)~kUsed;   // the anchor is walked again as the head of the second half
      if (!walk(x, y, hor ? kLeft : kUp, 1)) return false;
      if ((int)(half_n[0] - f0) + (int)(half_n[1] - s0) < P.min_line_len + 1) {
        half_n[0] = f0; half_n[1] = s0;   // short edge: dropped (pixels stay marked)
      } else {
        // the edge = first half reversed + second half without its head (the anchor)
        for (size_t i = half_n[0]; i-- > f0;) { cx[w] = fx[i]; cy[w++] = fy[i]; }
        for (size_t i = s0 + 1; i < half_n[1]; ++i) { cx[w] = sx[i]; cy[w++] = sy[i]; }
        cstart[++n_chains] = (uint32_t)w;
        if (progress && n_chains % kEdgeChainBatch == 0) progress->ready.store(n_chains, std::memory_order_release);
      }
    }
    ms_anchor = ta - t_begin;
    ms_route = clock_ms() - ta;
    n_anchor = (int)ax.size();
    n_walked = (int)walked;
    return n_chains > 0;   // otherwise "Edge drawing Error: lines not found"
  }

  static void solve(const FitState& F, double eq[2]) {
    const float* ata = F.ata;
    const float* atv = F.atv;
    const double c = 1.0 / (double(ata[0]) * double(ata[3]) - double(ata[1]) * double(ata[2]));
    eq[0] = c * (double(ata[3]) * double(atv[0]) - double(ata[1]) * double(atv[1]));
    eq[1] = c * (double(ata[0]) * double(atv[1]) - double(ata[2]) * double(atv[0]));
  }
  // accumulate the normal equations over pixels [b, e) of (px, py); `fresh` restarts them.
  // cv::Mat_<float> products: every entry is a double-accumulated dot product stored as float.
  static void accumulate(FitState& F, const uint16_t* px, const uint16_t* py, size_t b, size_t e, bool hor, bool fresh) {
    float* ata = F.ata;
    float* atv = F.atv;
    double suu = 0, su = 0, suv = 0, sv = 0;
    for (size_t i = b; i < e; ++i) {
      const double u = (double)(float)(hor ? px[i] : py[i]), v = (double)(float)(hor ? py[i] : px[i]);
      suu += u * u;
      su += u * 1.0;
      suv += u * v;
      sv += 1.0 * v;
    }
    const float n = (float)(double)(e - b);
    const float t[4] = {(float)suu, (float)su, (float)su, n};
    const float r[2] = {(float)suv, (float)sv};
    for (int i = 0; i < 4; ++i) ata[i] = fresh ? t[i] : ata[i] + t[i];
    for (int i = 0; i < 2; ++i) atv[i] = fresh ? r[i] : atv[i] + r[i];
  }

  bool validate(const uint16_t* px, const uint16_t* py, size_t b, size_t e, const double eq[3], float* dir_out) const {
    const int n = (int)(e - b);
    int mgx = 0, mgy = 0;
    for (int i = 0; i < n; ++i) {
      const int idx = py[b + i] * W + px[b + i];
      mgx += maps.dx[idx];
      mgy += maps.dy[idx];
    }
    const double dx = fabs(eq[1]), dy = fabs(eq[0]);
    if (mgx == 0 && mgy == 0) return false;
    float direction = *dir_out;   // the reference leaves it untouched if no quadrant matches
    if (mgx > 0 && mgy >= 0) direction = (float)atan2(-dy, dx);
    if (mgx <= 0 && mgy > 0) direction = (float)atan2(dy, dx);
    if (mgx < 0 && mgy <= 0) direction = (float)atan2(dy, -dx);
    if (mgx >= 0 && mgy < 0) direction = (float)atan2(-dy, -dx);
    *dir_out = direction;
    if (fabs(direction) < 0.15 || M_PI - fabs(direction) < 0.15)
      if (fabs(eq[2]) < 10 || fabs((unsigned)H - fabs(eq[2])) < 10) return false;
    if (fabs(fabs(direction) - M_PI * 0.5) < 0.15)
      if (fabs(eq[2]) < 10 || fabs((unsigned)W - fabs(eq[2])) < 10) return false;
    // Count the pixels whose level-line angle atan2(-gx, gy) lies within 0.392699 rad of the
    // line direction (circular distance).  The reference evaluates atan2 for every pixel;
    // here the angle between the two directions is bounded through their dot product and
    // libm's atan2 is only consulted when that bound is within 1e-9 of the threshold, so
    // the count is identical.
    const double D = (double)direction, cD = cos(D), sD = sin(D);
    const double ct = cos(0.392699), lo = ct * (1.0 - 1e-9), hi = ct * (1.0 + 1e-9);
    int k = 0;
    for (int i = 0; i < n; ++i) {
      const int idx = py[b + i] * W + px[b + i];
      const double gx = (double)maps.dx[idx], gy = (double)maps.dy[idx];
      const double nrm = sqrt(gx * gx + gy * gy);
      const double c = cD * gy - sD * gx;          // |g| * cos(angle between)
      bool aligned;
      if (nrm > 0 && c > hi * nrm) aligned = true;
      else if (nrm > 0 && c < lo * nrm) aligned = false;
      else {
        const double d = fabs(direction - atan2(-gx, gy));
        aligned = fabs(2 * M_PI - d) < 0.392699 || d < 0.392699;
      }
      if (aligned) ++k;
    }
    return nfa(n, k, 0.125, logNT) > 0;
  }

  bool fit_lines() {
    fit_range(0, num_chains(), segments);
    return true;
  }

 public:
  int num_chains() const { return n_chains; }
  int max_chains() const { return (int)((size_t)W * H / 100) + 1; }
  size_t chain_pixels(int c) const { return cstart[c + 1] - cstart[c]; }
  // EdgeDrawing only (the sequential part); the chains can then be fitted in
  // parallel with fit_range — every chain is fitted independently of the others.
  bool prepare(const OctaveMaps& m, const EdParams& prm) {
    segments.clear();
    W = m.w; H = m.h; maps = m; P = prm;
    logNT = 2.0 * (log10((double)(unsigned)W) + log10((double)(unsigned)H));
    const double t0 = clock_ms();
    const bool ok = draw_edges();
    ms_draw = clock_ms() - t0;
    if (!ok) n_chains = 0;
    failed = !ok;
    if (progress) {
      progress->ready.store(n_chains, std::memory_order_release);   // (the chains since the last full batch)
      progress->done.store(1, std::memory_order_release);
    }
    return ok;
  }

  // Line fitting + validation of chains [c0, c1) (EDline :2442-2650), appended to `out`.
  void fit_range(int c0, int c1, std::vector<Segment>& out) const {
    const int L = P.min_line_len;
    const double thr = P.fit_err_threshold;
    if (c1 <= c0) return;
    const size_t base = cstart[c0];
    std::vector<uint16_t> lx(cstart[c1] - base), ly(cstart[c1] - base);   // pixels of accepted / tentative lines
    size_t lpos = 0;
    float direction = 0;
    double eq2[2] = {0, 0};
    FitState F;
    const uint16_t* ex = cx.data();
    const uint16_t* ey = cy.data();
    for (int c = c0; c < c1; ++c) {
      size_t s = cstart[c];
      const size_t e = cstart[c + 1];
      while (e > s + L) {
        // slide until 15 consecutive pixels fit a line
        // The window's sums are sums of integers below 2^53: exact in the reference's double accumulators in
        // any order, so they are carried along the slide (two pixels out, two in) instead of re-added.
        double err = 0;
        int64_t wx = 0, wy = 0, wxx = 0, wyy = 0, wxy = 0;
        size_t wbeg = s, wend = s;   // the sums cover [wbeg, wend)
        while (e > s + L) {
          const bool hor0 = horizontal(ey[s] * W + ex[s]);
          for (; wbeg < s; ++wbeg) {
            const int64_t x = ex[wbeg], y = ey[wbeg];
            wx -= x; wy -= y; wxx -= x * x; wyy -= y * y; wxy -= x * y;
          }
          for (; wend < s + L; ++wend) {
            const int64_t x = ex[wend], y = ey[wend];
            wx += x; wy += y; wxx += x * x; wyy += y * y; wxy += x * y;
          }
          {
            const double suu = (double)(hor0 ? wxx : wyy), su = (double)(hor0 ? wx : wy), suv = (double)wxy,
                         sv = (double)(hor0 ? wy : wx);
            F.ata[0] = (float)suu; F.ata[1] = (float)su; F.ata[2] = (float)su; F.ata[3] = (float)(double)L;
            F.atv[0] = (float)suv; F.atv[1] = (float)sv;
          }
          solve(F, eq2);
          err = 0;
          for (int i = 0; i < L; ++i) {
            const double r = hor0 ? double(ey[s + i]) - double(ex[s + i]) * eq2[0] - eq2[1]
                                  : double(ex[s + i]) - double(ey[s + i]) * eq2[0] - eq2[1];
            err += r * r;
          }
          err = sqrt(err);
          if (err <= thr) break;
          s += 2;   // SkipEdgePoint
        }
        if (err > thr) break;
        const size_t lbeg = lpos;
        const bool hor = horizontal(ey[s] * W + ex[s]);
        double coef1 = 0;
        size_t new_from = 0;
        int tries = 0;
        bool first = true, extended = true;
        while (extended) {
          ++tries;
          if (first) {
            first = false;
            for (int i = 0; i < L; ++i) { lx[lpos] = ex[s]; ly[lpos++] = ey[s++]; }
          } else if (lpos > new_from) {
            // re-estimate with the pixels added by the previous try; the fit direction is
            // re-read from the FIRST pixel of the line, as the reference does
            const bool horf = horizontal(ly[lbeg] * W + lx[lbeg]);
            accumulate(F, lx.data(), ly.data(), new_from, lpos, horf, false);
            solve(F, eq2);
          }
          coef1 = 1 / sqrt(eq2[0] * eq2[0] + 1);
          int outliers = 0;
          new_from = lpos;
          while (e > s) {
            const double d = hor ? fabs(eq2[0] * ex[s] - ey[s] + eq2[1]) * coef1
                                 : fabs(ex[s] - eq2[0] * ey[s] - eq2[1]) * coef1;
            lx[lpos] = ex[s];
            ly[lpos++] = ey[s++];
            if (d > thr) {
              if (++outliers > 3) break;
            } else {
              outliers = 0;
            }
          }
          lpos -= outliers;
          s -= outliers;
          if (!(lpos > new_from && tries < 6)) extended = false;   // TryTime
        }
        double eq[3];
        if (hor) { eq[0] = eq2[0] * coef1; eq[1] = -1 * coef1; eq[2] = eq2[1] * coef1; }
        else { eq[0] = 1 * coef1; eq[1] = -eq2[0] * coef1; eq[2] = -eq2[1] * coef1; }
        if (validate(lx.data(), ly.data(), lbeg, lpos, eq, &direction)) {
          Segment sg;
          sg.eq[0] = eq[0]; sg.eq[1] = eq[1]; sg.eq[2] = eq[2];
          const double a1 = eq[1] * eq[1], a2 = eq[0] * eq[0], a3 = eq[0] * eq[1], a4 = eq[2] * eq[0], a5 = eq[2] * eq[1];
          unsigned Px = lx[lbeg], Py = ly[lbeg];
          sg.ep[0] = (float)(a1 * Px - a3 * Py - a4);
          sg.ep[1] = (float)(a2 * Py - a3 * Px - a5);
          Px = lx[lpos - 1]; Py = ly[lpos - 1];
          sg.ep[2] = (float)(a1 * Px - a3 * Py - a4);
          sg.ep[3] = (float)(a2 * Py - a3 * Px - a5);
          sg.direction = direction;
          sg.num_pixels = (int)(lpos - lbeg);
          out.push_back(sg);
        } else {
          lpos = lbeg;
        }
      }
    }
  }
};

// ------------------------------------------------------------ across octaves
struct KeyLine {   // cv::line_descriptor_c::KeyLine, 68 bytes (descriptor_custom.hpp:104-172)
  float angle;
  int32_t class_id, octave;
  float pt_x, pt_y, response, size;
  float startPointX, startPointY, endPointX, endPointY;
  float sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY;
  float lineLength;
  int32_t numOfPixels;
};

// OctaveKeyLines' grouping (:860-1149) + detectImpl's flattening (:504-555)
inline std::vector<KeyLine> group_and_flatten(const std::vector<OctaveDetector>& oct,
                                              const std::vector<std::pair<int, int>>& sizes, double factor) {
  struct Ref { int octave, id, group; float length; };
  const int no = (int)oct.size();
  std::vector<float> scale(no);
  scale[0] = 1;
  for (int o = 1; o < no; ++o) scale[o] = (float)(factor * scale[o - 1]);
  std::vector<Ref> all;
  int groups = 0;
  auto seg_len = [](const Segment& s) {
    const float dx = (float)fabs(s.ep[0] - s.ep[2]), dy = (float)fabs(s.ep[1] - s.ep[3]);
    return std::sqrt(dx * dx + dy * dy);
  };
  // per already-seen line, in discovery order: what the comparison below reads of it
  // (direction, scaled distance to the origin, scaled endpoints, scaled length)
  std::vector<float> a_dir, a_rho, a_len, a_ep;
  size_t total = 0;
  for (int o = 0; o < no; ++o) total += oct[o].segments.size();
  all.reserve(total); a_dir.reserve(total); a_rho.reserve(total); a_len.reserve(total); a_ep.reserve(4 * total);
  auto remember = [&](int o, int i, int group, float length) {
    const Segment& sg = oct[o].segments[i];
    all.push_back(Ref{o, i, group, length});
    a_dir.push_back(sg.direction);
    a_rho.push_back((float)(scale[o] * fabs(sg.eq[2])));
    a_len.push_back(length);
    for (int k = 0; k < 4; ++k) a_ep.push_back(scale[o] * sg.ep[k]);
  };
  for (int i = 0; i < (int)oct[0].segments.size(); ++i) remember(0, i, groups++, seg_len(oct[0].segments[i]));
  const double twoPI = 2 * M_PI;
  for (int o = 1; o < no; ++o) {
    const size_t lower = all.size();   // lines of the octaves below
    for (int i = 0; i < (int)oct[o].segments.size(); ++i) {
      const Segment& cur = oct[o].segments[i];
      const float rho1 = (float)(scale[o] * fabs(cur.eq[2]));
      const float tv = (float)(rho1 * 0.0152);
      float near_thr = (tv > 6) ? tv : 6;
      near_thr = (near_thr < 12) ? near_thr : 12;
      const float length = scale[o] * seg_len(cur);
      const float lp[4] = {scale[o] * cur.ep[0], scale[o] * cur.ep[1], scale[o] * cur.ep[2], scale[o] * cur.ep[3]};
      const float cdir = cur.direction;
      float best = 12;
      int best_ref = 0;
      // The distance-to-origin test rejects almost every pair: it is evaluated four lines at a time (same
      // float operations), and the rest of the comparison only runs where it passes.  Both tests are plain
      // rejections, so their order does not matter.
      const __m128 v_rho = _mm_set1_ps(rho1), v_thr = _mm_set1_ps(near_thr);
      const __m128 v_abs = _mm_castsi128_ps(_mm_set1_epi32(0x7fffffff));
      for (size_t r = 0; r < lower; ++r) {
        if ((r & 3) == 0 && r + 4 <= lower) {
          const __m128 d = _mm_and_ps(_mm_sub_ps(v_rho, _mm_loadu_ps(&a_rho[r])), v_abs);
          if (_mm_movemask_ps(_mm_cmpgt_ps(d, v_thr)) == 0xf) { r += 3; continue; }
        }
        if ((float)fabs(rho1 - a_rho[r]) > near_thr) continue;
        const float ddir = (float)fabs(cdir - a_dir[r]);
        if (ddir > 0.1745 && (twoPI - ddir > 0.1745)) continue;
        const float* np = &a_ep[4 * r];
        auto dist = [](float ax, float ay, float bx, float by) {
          const float dx = ax - bx, dy = ay - by;
          return std::sqrt(dx * dx + dy * dy);
        };
        float d = dist(lp[0], lp[1], np[0], np[1]);
        float mn = d, mx = d;
        d = dist(lp[2], lp[3], np[2], np[3]); mn = (d < mn) ? d : mn; mx = (d > mx) ? d : mx;
        d = dist(lp[0], lp[1], np[2], np[3]); mn = (d < mn) ? d : mn; mx = (d > mx) ? d : mx;
        d = dist(lp[2], lp[3], np[0], np[1]); mn = (d < mn) ? d : mn; mx = (d > mx) ? d : mx;
        if ((mx < 0.8 * (length + a_len[r])) && (mn < best)) {
          best = mn;
          best_ref = (int)r;
        }
      }
      remember(o, i, (best < 12) ? all[best_ref].group : groups++, length);
    }
  }
  // ScaleLines: group -> members in discovery order; flattened group by group
  std::vector<int> first(groups + 1, 0), order(all.size());   // stable counting sort by group
  for (const Ref& r : all) ++first[r.group + 1];
  for (int gidx = 0; gidx < groups; ++gidx) first[gidx + 1] += first[gidx];
  {
    std::vector<int> fill(first.begin(), first.end() - 1);
    for (size_t r = 0; r < all.size(); ++r) order[fill[all[r].group]++] = (int)r;
  }
  std::vector<KeyLine> out;
  out.reserve(all.size());
  for (int gidx = 0; gidx < groups; ++gidx)
    for (int m = first[gidx]; m < first[gidx + 1]; ++m) {
      const Ref& ref = all[order[m]];
      const Segment& s = oct[ref.octave].segments[ref.id];
      const float direction = s.direction;
      const float s1 = s.ep[0], s2 = s.ep[1], e1 = s.ep[2], e2 = s.ep[3];
      const float dx = e1 - s1, dy = e2 - s2;
      bool swap = false;
      if (direction >= -0.75 * M_PI && direction < -0.25 * M_PI && dy > 0) swap = true;
      if (direction >= -0.25 * M_PI && direction < 0.25 * M_PI && dx < 0) swap = true;
      if (direction >= 0.25 * M_PI && direction < 0.75 * M_PI && dy < 0) swap = true;
      if (((direction >= 0.75 * M_PI && direction < M_PI) || (direction >= -M_PI && direction < -0.75 * M_PI)) && dx > 0)
        swap = true;
      const float sc = scale[ref.octave];
      KeyLine k;
      k.sPointInOctaveX = swap ? e1 : s1; k.sPointInOctaveY = swap ? e2 : s2;
      k.ePointInOctaveX = swap ? s1 : e1; k.ePointInOctaveY = swap ? s2 : e2;
      k.startPointX = sc * k.sPointInOctaveX; k.startPointY = sc * k.sPointInOctaveY;
      k.endPointX = sc * k.ePointInOctaveX; k.endPointY = sc * k.ePointInOctaveY;
      k.lineLength = ref.length;
      k.numOfPixels = s.num_pixels;
      k.angle = direction;
      k.class_id = gidx;
      k.octave = ref.octave;
      k.size = (k.endPointX - k.startPointX) * (k.endPointY - k.startPointY);
      k.response = k.lineLength / std::max(sizes[ref.octave].first, sizes[ref.octave].second);
      k.pt_x = (k.endPointX + k.startPointX) / 2;
      k.pt_y = (k.endPointY + k.startPointY) / 2;
      out.push_back(k);
    }
  return out;
}

// LineExtractor::detectLineFeatures' selection (src/LineExtractor.cc:229-266)
inline void select_lines(std::vector<KeyLine>& lines, int nfeatures, int img_w, int img_h, double min_length) {
  if ((int)lines.size() > nfeatures && nfeatures != 0) {
    std::sort(lines.begin(), lines.end(), [](const KeyLine& a, const KeyLine& b) { return a.response > b.response; });
    lines.resize(nfeatures);
  }
  const int lo = 5, hx = img_w - 5, hy = img_h - 5;   // kBorderThreshold
  lines.erase(std::remove_if(lines.begin(), lines.end(),
                             [&](const KeyLine& l) {
                               return ((l.startPointX < lo) && (l.endPointX < lo)) || ((l.startPointX > hx) && (l.endPointX > hx)) ||
                                      ((l.startPointY < lo) && (l.endPointY < lo)) || ((l.startPointY > hy) && (l.endPointY > hy));
                             }),
              lines.end());
  size_t keep = lines.size();
  for (size_t i = 0; i < lines.size(); ++i) {
    lines[i].class_id = (int)i;
    if (lines[i].response < min_length) { keep = i; break; }
  }
  lines.resize(keep);
}

}  // namespace lines
}  // namespace plvs
