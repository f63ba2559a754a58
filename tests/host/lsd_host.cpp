// Host compile of the product's LSD host stages (plvs_amd/csrc/lsd_host.hpp: tap tables, pseudo-ordering, region growing,
// rectangle refinement, NFA), fed with the per-pixel maps a plain C++ loop makes with the formulas of the device kernels
// (lsd_blur / lsd_resize_exact / lsd_ll_angle in plvs_amd/csrc/lsd_lines.inc) — a CPU-side agreement check against the
// reference's compiled LSD.  Test infrastructure only.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../plvs_amd/csrc/lsd_host.hpp"

using namespace plvs::lsd;

static int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
  return p;
}

extern "C" int hostlsd_segments(const uint8_t* img, int w, int h, int stride, int refine, double scale, double sigma_scale,
                                double quant, double ang_th, double log_eps, double density_th, int n_bins, float* out, int cap) {
  Options o;
  o.refine = refine; o.scale = scale; o.sigma_scale = sigma_scale; o.quant = quant; o.ang_th = ang_th; o.log_eps = log_eps;
  o.density_th = density_th; o.n_bins = n_bins;
  std::vector<uint8_t> field((size_t)w * h);
  for (int y = 0; y < h; ++y) memcpy(&field[(size_t)y * w], img + (size_t)y * stride, (size_t)w);
  int fw = w, fh = h;
  if (scale != 1) {
    const double sigma = (scale < 1) ? (sigma_scale / scale) : sigma_scale;
    const unsigned int hh = (unsigned int)(ceil(sigma * sqrt(2 * 3.0 * log(10.0))));
    const int ksize = 1 + 2 * (int)hh;
    if (ksize > 31) return -1;
    const Q8Kernel k = gaussian_q8(ksize, sigma);
    std::vector<uint8_t> blur((size_t)w * h);
    const int r = ksize / 2;
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        uint32_t acc = 0;
        for (int j = -r; j <= r; ++j) {
          const uint8_t* row = &field[(size_t)reflect101(y + j, h) * w];
          uint32_t rowv = 0;
          for (int i = -r; i <= r; ++i) rowv += (uint32_t)k.w[i + r] * row[reflect101(x + i, w)];
          acc += (uint32_t)k.w[j + r] * rowv;
        }
        blur[(size_t)y * w + x] = (uint8_t)((acc + (1u << 15)) >> 16);
      }
    fw = (int)lrint(w * scale);
    fh = (int)lrint(h * scale);
    if (fw < 2 || fh < 2) return -1;
    const ExactAxis X = exact_resize_axis(scale, w, fw), Y = exact_resize_axis(scale, h, fh);
    std::vector<uint8_t> scaled((size_t)fw * fh);
    auto hline = [&](int sy, int dx) -> uint32_t {
      const uint8_t* S = &blur[(size_t)sy * w];
      if (dx < X.lo) return (uint32_t)S[0] << 8;
      if (dx < X.hi) return (uint32_t)X.c0[dx] * S[X.ofs[dx]] + (uint32_t)X.c1[dx] * S[X.ofs[dx] + 1];
      return (uint32_t)S[X.ofs[fw - 1]] << 8;
    };
    for (int dy = 0; dy < fh; ++dy)
      for (int dx = 0; dx < fw; ++dx) {
        uint8_t v8;
        if (dy < Y.lo || dy >= Y.hi) {
          v8 = (uint8_t)((hline(dy < Y.lo ? 0 : Y.ofs[fh - 1], dx) + 128u) >> 8);
        } else {
          const unsigned long long v = (unsigned long long)Y.c0[dy] * hline(Y.ofs[dy], dx) + (unsigned long long)Y.c1[dy] * hline(Y.ofs[dy] + 1, dx);
          const unsigned long long q = (v + 32768ull) >> 16;
          v8 = (uint8_t)(q > 255ull ? 255ull : q);
        }
        scaled[(size_t)dy * fw + dx] = v8;
      }
    field.swap(scaled);
  }
  const double prec = kPi * ang_th / 180;
  const double rho = quant / sin(prec);
  std::vector<double> ang((size_t)fw * fh, kNotDef), mod((size_t)fw * fh, 0.0);
  double max_grad = -1;
  for (int y = 0; y < fh - 1; ++y)
    for (int x = 0; x < fw - 1; ++x) {
      const size_t at = (size_t)y * fw + x;
      const int DA = (int)field[at + fw + 1] - (int)field[at];
      const int BC = (int)field[at + 1] - (int)field[at + fw];
      const int gx = DA + BC, gy = DA - BC;
      const double norm = std::sqrt((double)(gx * gx + gy * gy) / 4.0);
      mod[at] = norm;
      if (norm <= rho) continue;
      ang[at] = (double)fast_atan2_deg(float(gx), float(-gy)) * kDegToRad;
      if (norm > max_grad) max_grad = norm;
    }
  Level level;
  std::vector<Segment4> segs;
  level.detect(ang.data(), mod.data(), fw, fh, max_grad, o, segs);
  if (getenv("PLVS_LSD_PROFILE")) fprintf(stderr, "%d x %d: ordering %.2f ms, seed loop %.2f ms, %zu segments\n", fw, fh, level.ms_order, level.ms_regions, segs.size());
  if ((int)segs.size() <= cap && !segs.empty()) memcpy(out, segs.data(), segs.size() * sizeof(Segment4));
  return (int)segs.size();
}
