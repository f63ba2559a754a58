// TEST INFRASTRUCTURE ONLY — the reference's LSD line detector compiled from where it lies:
//   Thirdparty/line_descriptor/src/lsd_custom.cpp          cv::lsd::LineSegmentDetectorImpl (ll_angle, region_grow, region2rect,
//                                                          refine, rect_improve, rect_nfa, nfa)
//   Thirdparty/line_descriptor/src/LSDDetector_custom.cpp  LSDDetectorC (pyramid, per-octave detection, KeyLine records)
//   Thirdparty/line_descriptor/src/binary_descriptor_custom.cpp   BinaryDescriptor::compute without detection data (computeSobel + LBD)
//   src/LineExtractor.cc                                   detectLineFeatures with skUseLsdExtractor = true
// all UNMODIFIED, against oracle/ref/cv_full with PLVS_CVFULL_LSD (the output-array proxy, Vec4f, LineIterator's count and the
// INTER_LINEAR_EXACT resize: see cvfull.hpp; the image primitives are the restatements of oracle/cv_primitives.hpp, as for the
// rest of the front end).  This library IS the checker of the HIP path for row L9: there is no second restatement in oracle/ —
// tests compare the device's segments, KeyLines and descriptors with what these sources return, and with digests they made
// (tests/golden/lsd_reference_digests.json) where the library is absent.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "LineExtractor.h"

namespace {
using cv::line_descriptor_c::KeyLine;
using cv::line_descriptor_c::LSDDetectorC;
struct Extractor : PLVS2::LineExtractor {
  Extractor(int n, LSDDetectorC::LSDOptions& o) : PLVS2::LineExtractor(n, o) {}
};
LSDDetectorC::LSDOptions make_opts(int num_octaves, int refine, double scale, double sigma_scale, double quant, double ang_th,
                                   double log_eps, double density_th, int n_bins, double min_length) {
  LSDDetectorC::LSDOptions o;
  o.numOctaves = num_octaves;
  o.refine = refine;
  o.scale = scale;
  o.sigma_scale = sigma_scale;
  o.quant = quant;
  o.ang_th = ang_th;
  o.log_eps = log_eps;
  o.density_th = density_th;
  o.n_bins = n_bins;
  o.min_length = min_length;
  return o;
}
}  // namespace

extern "C" {

// cv::lsd::createLineSegmentDetector(...)->detect(image, lines) on ONE image: segments as x1, y1, x2, y2 floats.
int ref_lsd_segments(const uint8_t* img, int w, int h, int stride, int refine, double scale, double sigma_scale, double quant,
                     double ang_th, double log_eps, double density_th, int n_bins, float* out, int cap) {
  cv::Mat image(h, w, CV_8UC1, const_cast<uint8_t*>(img), (size_t)stride);
  cv::Ptr<cv::lsd::LineSegmentDetector> ls =
      cv::lsd::createLineSegmentDetector(refine, scale, sigma_scale, quant, ang_th, log_eps, density_th, n_bins);
  std::vector<cv::Vec4f> lines;
  ls->detect(image, lines);
  for (size_t i = 0; i < lines.size() && (int)i < cap; ++i)
    for (int k = 0; k < 4; ++k) out[4 * i + k] = lines[i][k];
  return (int)lines.size();
}

// LSDDetectorC::detect(image, keylines, scale, numOctaves, opts) (LSDDetector_custom.cpp:165-298): KeyLine records of 68 bytes;
// `pyramid_scale` is that call's scale argument, `scale` the options' (the detector's own rescaling).
int ref_lsd_detect(const uint8_t* img, int w, int h, int stride, int num_octaves, float pyramid_scale, int refine, double scale, double sigma_scale,
                   double quant, double ang_th, double log_eps, double density_th, int n_bins, double min_length,
                   void* keylines, int cap) {
  cv::Mat image(h, w, CV_8UC1, const_cast<uint8_t*>(img), (size_t)stride);
  LSDDetectorC::LSDOptions o = make_opts(num_octaves, refine, scale, sigma_scale, quant, ang_th, log_eps, density_th, n_bins, min_length);
  cv::Ptr<LSDDetectorC> det = LSDDetectorC::createLSDDetectorC(o);
  std::vector<KeyLine> lines;
  det->detect(image, lines, pyramid_scale, num_octaves, o, cv::Mat());
  static_assert(sizeof(KeyLine) == 68, "KeyLine must be 68 bytes");
  if ((int)lines.size() <= cap && !lines.empty()) std::memcpy(keylines, lines.data(), lines.size() * sizeof(KeyLine));
  return (int)lines.size();
}

// PLVS2::LineExtractor::operator() with LineExtractor::skUseLsdExtractor = true (Line.LSD.on: 1): detection by LSD, selection,
// LBD without detection data.  Options as Tracking's settings parser fills them (src/Tracking.cc:1458-1485).
int ref_lsd_extract(const uint8_t* img, int w, int h, int stride, int nfeatures, int num_octaves, int refine, double scale,
                    double sigma_scale, double quant, double ang_th, double log_eps, double density_th, int n_bins,
                    double min_length, double fit_err, void* keylines, uint8_t* desc, int cap) {
  PLVS2::LineExtractor::skUseLsdExtractor = true;
  LSDDetectorC::LSDOptions o = make_opts(num_octaves, refine, scale, sigma_scale, quant, ang_th, log_eps, density_th, n_bins, min_length);
  o.lineFitErrThreshold = fit_err;
  Extractor ex(nfeatures, o);
  cv::Mat image(h, w, CV_8UC1, const_cast<uint8_t*>(img), (size_t)stride);
  std::vector<KeyLine> lines;
  cv::Mat d;
  ex(image, lines, d);
  if ((int)lines.size() <= cap) {
    if (!lines.empty()) std::memcpy(keylines, lines.data(), lines.size() * sizeof(KeyLine));
    for (int i = 0; i < d.rows; ++i) std::memcpy(desc + (size_t)i * 32, d.ptr(i), 32);
  }
  return (int)lines.size();
}

}  // extern "C"
