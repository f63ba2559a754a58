// C entry point over the reference's OWN dense stereo matcher — Thirdparty/libelas-gpu/CPU/{elas,descriptor,filter,
// matrix,triangle}.cpp (libelas: support points, Delaunay triangulation, plane priors, dense matching, left/right
// check, gap interpolation, median filters), compiled UNMODIFIED where they lie (oracle/ref/Makefile ->
// oracle/_ref/libelas_ref.so).  PLVS calls it from PointCloudKeyFrame::ProcessStereoLibelas (src/PointCloudKeyFrame.cc:335).
// tests/test_oracle_pinned_elas.py runs it over the reference's own input pairs (Thirdparty/libelas-gpu/input/*.pgm,
// committed as fixtures where small) and checks the disparities against the outputs the reference STORES in its tree
// (Thirdparty/libelas-gpu/GPU_test/2016_12_06_cpu/*_disp.pgm) — the one golden vector the reference holds for this path.
// TEST INFRASTRUCTURE ONLY: nothing under plvs_amd/ may call into this file.
#include <cstdint>
#include <cstring>

#include "elas.h"

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

}  // extern "C"
