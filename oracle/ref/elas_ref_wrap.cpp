// C entry point over the reference's OWN dense stereo matcher — Thirdparty/libelas-gpu/CPU/{elas,descriptor,filter,
// matrix,triangle}.cpp (libelas: support points, Delaunay triangulation, plane priors, dense matching, left/right
// check, gap interpolation, median filters), compiled UNMODIFIED where they lie (oracle/ref/Makefile ->
// oracle/_ref/libelas_ref.so).  PLVS calls it from PointCloudKeyFrame::ProcessStereoLibelas (src/PointCloudKeyFrame.cc:335).
// tests/test_oracle_pinned_elas.py runs it over the reference's own input pairs (Thirdparty/libelas-gpu/input/*.pgm,
// committed as fixtures where small) and checks the disparities against the outputs the reference STORES in its tree
// (Thirdparty/libelas-gpu/GPU_test/2016_12_06_cpu/*_disp.pgm) — the one golden vector the reference holds for this path.
// TEST INFRASTRUCTURE ONLY: nothing under plvs_amd/ may call into this file.
#include <chrono>
#include <cstdint>
#include <cstring>

#include "elas.h"
#include "descriptor.h"

// The two methods the reference's own accelerated build overrides (Thirdparty/libelas-gpu/GPU/elas_gpu.h:41-45,
// class ElasGPU : public Elas): computeDisparity and adaptiveMean.  HookedElas lets a test stand in the same place:
// a hook receives the arguments the reference pipeline hands over and either calls the reference's own method
// (ref_elas_base_*: capture of inputs and expected outputs) or fills D itself (the oracle, the HIP path) — the rest of
// the pipeline stays the reference's compiled code.
extern "C" {
struct ref_elas_hooks {
  // support: n_support x {u, v, d}; tri: n_tri x {c1, c2, c3, t1a, t1b, t1c, t2a, t2b, t2c} (36-byte records)
  void (*compute_disparity)(void* user, void* call, const int32_t* support, int n_support, const void* tri, int n_tri,
                            const int32_t* grid, const int32_t* grid_dims, const uint8_t* I1_desc, const uint8_t* I2_desc,
                            int right_image, float* D);
  void (*adaptive_mean)(void* user, void* elas, float* D);
  void* user;
  // (not one of ElasGPU's: the candidate grid inside Elas::computeSupportMatches, elas.cpp:434-456.  D_can arrives
  // zeroed, D_can_width x D_can_height; the reference's own filters and the conversion to support points follow.)
  void (*support_candidates)(void* user, const uint8_t* I1_desc, const uint8_t* I2_desc, int16_t* D_can, int D_can_width,
                             int D_can_height);
  // (the post-processing between computeDisparity and adaptiveMean, in place of the reference's own methods)
  void (*left_right_check)(void* user, float* D1, float* D2);
  void (*remove_small_segments)(void* user, float* D);
  void (*gap_interpolation)(void* user, float* D);
};
}

namespace {

struct HookedElas;
struct DisparityCall {
  HookedElas* self;
  std::vector<libelas::Elas::support_pt>* support;
  std::vector<libelas::Elas::triangle>* tri;
  int32_t *grid, *grid_dims;
  uint8_t *d1, *d2;
  bool right;
};

struct HookedElas : libelas::Elas {
  const ref_elas_hooks* hooks;
  HookedElas(Parameters p, const ref_elas_hooks* h) : libelas::Elas(p), hooks(h) {}
  void computeDisparity(std::vector<support_pt> p_support, std::vector<triangle> tri, int32_t* disparity_grid, int32_t* grid_dims,
                        uint8_t* I1_desc, uint8_t* I2_desc, bool right_image, float* D) override {
    static_assert(sizeof(support_pt) == 12 && sizeof(triangle) == 36, "record layouts handed to the hooks");
    if (hooks == nullptr || hooks->compute_disparity == nullptr) {
      Timed t(this, 4);
      libelas::Elas::computeDisparity(p_support, tri, disparity_grid, grid_dims, I1_desc, I2_desc, right_image, D);
      return;
    }
    Timed t(this, 4);
    DisparityCall call{this, &p_support, &tri, disparity_grid, grid_dims, I1_desc, I2_desc, right_image};
    hooks->compute_disparity(hooks->user, &call, reinterpret_cast<const int32_t*>(p_support.data()), (int)p_support.size(),
                             tri.data(), (int)tri.size(), disparity_grid, grid_dims, I1_desc, I2_desc, right_image ? 1 : 0, D);
  }
  void adaptiveMean(float* D) override {
    Timed t(this, 9);
    if (hooks == nullptr || hooks->adaptive_mean == nullptr) libelas::Elas::adaptiveMean(D);
    else hooks->adaptive_mean(hooks->user, this, D);
  }
  // where Elas::process spends its time: seconds per virtual stage of the last run (ref_elas_stage_seconds)
  double seconds[12] = {};
  struct Timed {
    HookedElas* e;
    int i;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    Timed(HookedElas* e, int i) : e(e), i(i) {}
    ~Timed() { e->seconds[i] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
  };
  std::vector<support_pt> computeSupportMatches(uint8_t* a, uint8_t* b) override {
    Timed t(this, 0);
    if (hooks == nullptr || hooks->support_candidates == nullptr) return libelas::Elas::computeSupportMatches(a, b);
    // elas.cpp:416-489 around the candidate loop (:434-456), which the hook runs
    int32_t step = param.candidate_stepsize;
    if (param.subsampling) step += step % 2;
    int32_t W = 0, H = 0;
    for (int32_t u = 0; u < width; u += step) W++;
    for (int32_t v = 0; v < height; v += step) H++;
    int16_t* D_can = (int16_t*)calloc(W * H, sizeof(int16_t));
    hooks->support_candidates(hooks->user, a, b, D_can, W, H);
    removeInconsistentSupportPoints(D_can, W, H);
    removeRedundantSupportPoints(D_can, W, H, 5, 1, true);
    removeRedundantSupportPoints(D_can, W, H, 5, 1, false);
    std::vector<support_pt> p_support;
    for (int32_t u_can = 1; u_can < W; u_can++)
      for (int32_t v_can = 1; v_can < H; v_can++)
        if (D_can[v_can * W + u_can] >= 0) p_support.push_back(support_pt(u_can * step, v_can * step, D_can[v_can * W + u_can]));
    if (param.add_corners) addCornerSupportPoints(p_support);
    free(D_can);
    return p_support;
  }
  std::vector<triangle> computeDelaunayTriangulation(std::vector<support_pt> p, int32_t r) override { Timed t(this, 1); return libelas::Elas::computeDelaunayTriangulation(p, r); }
  void computeDisparityPlanes(std::vector<support_pt> p, std::vector<triangle>& tr, int32_t r) override { Timed t(this, 2); libelas::Elas::computeDisparityPlanes(p, tr, r); }
  void createGrid(std::vector<support_pt> p, int32_t* g, int32_t* gd, bool r) override { Timed t(this, 3); libelas::Elas::createGrid(p, g, gd, r); }
  void leftRightConsistencyCheck(float* D1, float* D2) override {
    Timed t(this, 5);
    if (hooks && hooks->left_right_check) hooks->left_right_check(hooks->user, D1, D2);
    else libelas::Elas::leftRightConsistencyCheck(D1, D2);
  }
  void removeSmallSegments(float* D) override {
    Timed t(this, 6);
    if (hooks && hooks->remove_small_segments) hooks->remove_small_segments(hooks->user, D);
    else libelas::Elas::removeSmallSegments(D);
  }
  void gapInterpolation(float* D) override {
    Timed t(this, 7);
    if (hooks && hooks->gap_interpolation) hooks->gap_interpolation(hooks->user, D);
    else libelas::Elas::gapInterpolation(D);
  }
  void median(float* D) override { Timed t(this, 8); libelas::Elas::median(D); }
};

}  // namespace

extern "C" {

// Elas::process with the parameters of main_cpu.cpp (the run that produced the stored outputs): Parameters() defaults
// (ROBOTICS setting), postprocess_only_left = false; `plvs` != 0: the parameters PLVS sets instead
// (postprocess_only_left = true, subsampling = `subsampling`; src/PointCloudKeyFrame.cc:348-350).
// D1 / D2: width x height floats (width/2 x height/2 with subsampling), -1 = invalid.
void ref_elas_process(const uint8_t* left, const uint8_t* right, int width, int height, int stride, int plvs, int subsampling,
                      float* D1, float* D2) {
  libelas::Elas::Parameters param;
  param.postprocess_only_left = plvs != 0;
  if (plvs) param.subsampling = subsampling != 0;
  libelas::Elas elas(param);
  const int32_t dims[3] = {width, height, stride};
  elas.process(const_cast<uint8_t*>(left), const_cast<uint8_t*>(right), D1, D2, dims);
}


// The same with hooks in ElasGPU's two places (either may be null = the reference's own method).
// (Elas::adaptiveMean reads its scratch image where the horizontal pass never wrote it, findMatch the descriptor images'
// unwritten borders: elas_zero_malloc.h pins that memory to zero, the content of fresh pages.)
void ref_elas_process_hooked(const uint8_t* left, const uint8_t* right, int width, int height, int stride, int plvs,
                             int subsampling, float* D1, float* D2, const ref_elas_hooks* hooks) {
  // plvs: bit 0 = postprocess_only_left (PLVS), bit 1 = the MIDDLEBURY parameter set (add_corners, wider gaps, no texture floor)
  libelas::Elas::Parameters param((plvs & 2) ? libelas::Elas::MIDDLEBURY : libelas::Elas::ROBOTICS);
  param.postprocess_only_left = (plvs & 1) != 0;
  param.subsampling = subsampling != 0;
  HookedElas elas(param, hooks);
  const int32_t dims[3] = {width, height, stride};
  elas.process(const_cast<uint8_t*>(left), const_cast<uint8_t*>(right), D1, D2, dims);
}

// Where Elas::process spends its time: the pipeline once more, seconds per stage — 0 support matches, 1 triangulation
// (both images), 2 planes, 3 grid, 4 computeDisparity, 5 left/right check, 6 speckles, 7 gaps, 8 median, 9 adaptiveMean,
// 10 the whole of process (the remainder is the two Descriptor constructions).
void ref_elas_stage_seconds(const uint8_t* left, const uint8_t* right, int width, int height, int stride, int plvs,
                            int subsampling, double* seconds11) {
  libelas::Elas::Parameters param;
  param.postprocess_only_left = plvs != 0;
  param.subsampling = subsampling != 0;
  HookedElas elas(param, nullptr);
  std::vector<float> D1((size_t)width * height), D2((size_t)width * height);
  const int32_t dims[3] = {width, height, stride};
  const auto t0 = std::chrono::steady_clock::now();
  elas.process(const_cast<uint8_t*>(left), const_cast<uint8_t*>(right), D1.data(), D2.data(), dims);
  elas.seconds[10] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (int i = 0; i < 11; ++i) seconds11[i] = elas.seconds[i];
}

// From inside a hook: the reference's own computeDisparity / adaptiveMean on the arguments of that call.
void ref_elas_base_compute_disparity(void* call, float* D) {
  DisparityCall* c = static_cast<DisparityCall*>(call);
  c->self->libelas::Elas::computeDisparity(*c->support, *c->tri, c->grid, c->grid_dims, c->d1, c->d2, c->right, D);
}
void ref_elas_base_adaptive_mean(void* elas, float* D) { static_cast<HookedElas*>(elas)->libelas::Elas::adaptiveMean(D); }

// Elas::adaptiveMean alone on a disparity image (width x height is the IMAGE size: with subsampling D is half of it).
void ref_elas_adaptive_mean(float* D, int width, int height, int subsampling) {
  libelas::Elas::Parameters param;
  param.subsampling = subsampling != 0;
  struct Sized : libelas::Elas {
    Sized(Parameters p, int w, int h) : libelas::Elas(p) { width = w; height = h; }
  } e(param, width, height);
  e.adaptiveMean(D);
}

// libelas::Descriptor on an image as Elas::process hands it over (elas.cpp:39-57: rows copied into a zeroed buffer of line
// length bpl): desc = 16 * width * height bytes.
void ref_elas_descriptor(const uint8_t* img, int width, int height, int stride, int half_resolution, uint8_t* desc) {
  const int32_t bpl = width + 15 - (width - 1) % 16;
  uint8_t* I = (uint8_t*)_mm_malloc(bpl * height * sizeof(uint8_t), 16);
  memset(I, 0, bpl * height * sizeof(uint8_t));
  for (int32_t v = 0; v < height; v++) memcpy(I + v * bpl, img + v * stride, width * sizeof(uint8_t));
  {
    libelas::Descriptor d(I, width, height, bpl, half_resolution != 0);
    memcpy(desc, d.I_desc, (size_t)16 * width * height);
  }
  _mm_free(I);
}

}  // extern "C"
