// C entry points over the reference's OWN ORB extractor — src/ORBextractor.cc (CPU branch: ComputePyramid,
// ComputeKeyPointsOctTree with its per-cell FAST loop, DistributeOctTree / DivideNode, IC_Angle, computeOrbDescriptor,
// operator() with the lapping-area packing) compiled UNMODIFIED against the OpenCV stand-in of oracle/ref/cv_full,
// whose image primitives (FAST, resize, GaussianBlur, copyMakeBorder, fastAtan2, cvRound) forward to
// oracle/cv_primitives.hpp.  tests/test_oracle_pinned_frontend.py checks oracle/orb.cpp against it field by field:
// the pin is "up to the OpenCV primitives".
// TEST INFRASTRUCTURE ONLY: nothing under plvs_amd/ may call into this file.
#include <cstdint>
#include <cstring>
#include <vector>

#include "ORBextractor.h"

namespace {
struct Extractor : PLVS2::ORBextractor {   // (the constructor tables are protected)
  using PLVS2::ORBextractor::ORBextractor;
  const std::vector<int>& per_level() const { return mnFeaturesPerLevel; }
  const std::vector<int>& u_max() const { return umax; }
};
}  // namespace

extern "C" {

struct ref_kp {
  float x, y, size, angle, response;
  int32_t octave, class_id;
};

void* ref_orb_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST) {
  return static_cast<PLVS2::ORBextractor*>(new Extractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST));
}
void ref_orb_destroy(void* h) { delete static_cast<PLVS2::ORBextractor*>(h); }

// Frame::ExtractORB (src/Frame.cc:806-813): (*extractor)(image, cv::Mat(), keys, descriptors, vLapping).
// Returns monoIndex (-1 on an empty image); *n = number of key points (nothing is written when it exceeds cap).
int ref_orb_extract(void* h, const uint8_t* img, int w, int hh, int stride, int lap0, int lap1, ref_kp* kps, uint8_t* desc,
                    int cap, int* n) {
  PLVS2::ORBextractor* e = static_cast<PLVS2::ORBextractor*>(h);
  cv::Mat image;
  if (w > 0 && hh > 0) image = cv::Mat(hh, w, CV_8UC1, const_cast<uint8_t*>(img), (size_t)stride);
  std::vector<cv::KeyPoint> k;
  cv::Mat d;
  std::vector<int> lap = {lap0, lap1};
  const int mono = (*e)(image, cv::Mat(), k, d, lap);
  if (mono < 0) { *n = 0; return -1; }
  *n = (int)k.size();
  if ((int)k.size() <= cap) {
    for (size_t i = 0; i < k.size(); ++i)
      kps[i] = ref_kp{k[i].pt.x, k[i].pt.y, k[i].size, k[i].angle, k[i].response, k[i].octave, k[i].class_id};
    for (int i = 0; i < d.rows; ++i) std::memcpy(desc + (size_t)i * 32, d.ptr(i), 32);
  }
  return mono;
}

// the pyramid / the blurred pyramid of the last extract (tightly packed copies)
int ref_orb_level_size(void* h, int level, int* w, int* hh) {
  PLVS2::ORBextractor* e = static_cast<PLVS2::ORBextractor*>(h);
  *w = e->mvImagePyramid[level].cols;
  *hh = e->mvImagePyramid[level].rows;
  return 0;
}
void ref_orb_get_level(void* h, int level, int blurred, uint8_t* out) {
  PLVS2::ORBextractor* e = static_cast<PLVS2::ORBextractor*>(h);
  const cv::Mat& m = blurred ? e->mvImagePyramidFiltered[level] : e->mvImagePyramid[level];
  for (int r = 0; r < m.rows; ++r) std::memcpy(out + (size_t)r * m.cols, m.ptr(r), (size_t)m.cols);
}
int ref_orb_features_per_level(void* h, int* out) {
  const Extractor* e = static_cast<Extractor*>(static_cast<PLVS2::ORBextractor*>(h));
  for (size_t i = 0; i < e->per_level().size(); ++i) out[i] = e->per_level()[i];
  return (int)e->per_level().size();
}
void ref_orb_umax(void* h, int* out) {
  const Extractor* e = static_cast<Extractor*>(static_cast<PLVS2::ORBextractor*>(h));
  for (size_t i = 0; i < e->u_max().size(); ++i) out[i] = e->u_max()[i];
}
int ref_orb_scale_tables(void* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2) {
  PLVS2::ORBextractor* e = static_cast<PLVS2::ORBextractor*>(h);
  const int n = e->GetLevels();
  for (int i = 0; i < n; ++i) {
    scale[i] = e->GetScaleFactors()[i];
    inv_scale[i] = e->GetInverseScaleFactors()[i];
    sigma2[i] = e->GetScaleSigmaSquares()[i];
    inv_sigma2[i] = e->GetInverseScaleSigmaSquares()[i];
  }
  return n;
}

}  // extern "C"
