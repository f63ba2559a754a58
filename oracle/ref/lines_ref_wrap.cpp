// C entry points over the reference's OWN line front end — src/LineExtractor.cc (constructor tables, detectLineFeatures:
// sort by response, border filter, minimum length cut, compute with the detection data reused) driving
// Thirdparty/line_descriptor/src/binary_descriptor_custom.cpp (OctaveKeyLines with its octave grouping, EDLineDetector:
// EdgeDrawing's anchors and smart routing, the least-squares fits, the Helmholtz validation; detectImpl, computeImpl,
// computeLBD), both compiled UNMODIFIED against the OpenCV stand-in of oracle/ref/cv_full, whose image primitives
// (GaussianBlur, resize, Sobel, the CV_16S element-wise operations, the small float products) forward to
// oracle/cv_primitives.hpp.  tests/test_oracle_pinned_frontend.py checks oracle/lines.cpp against it byte by byte.
// The LSD detector (Line.LSD.on: 1, off in every shipped YAML; lsd_custom.cpp needs far more of OpenCV) is not
// compiled: its three entry points named by LineExtractor.cc abort.
// TEST INFRASTRUCTURE ONLY: nothing under plvs_amd/ may call into this file.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "LineExtractor.h"
#include "ORBextractor.h"

namespace cv {
namespace line_descriptor_c {
Ptr<LSDDetectorC> LSDDetectorC::createLSDDetectorC(const LSDOptions&) { std::abort(); }
void LSDDetectorC::detect(const Mat&, std::vector<KeyLine>&, float, int, const LSDOptions&, const Mat&) { std::abort(); }
void LSDDetectorC::setGaussianPyramid(const std::vector<cv::Mat>&, int, float, int, int) { std::abort(); }
}  // namespace line_descriptor_c
}  // namespace cv

namespace {
using cv::line_descriptor_c::BinaryDescriptor;
using cv::line_descriptor_c::KeyLine;
struct Extractor : PLVS2::LineExtractor {   // (mLbd is protected)
  Extractor(int n, cv::line_descriptor_c::LSDDetectorC::LSDOptions& o) : PLVS2::LineExtractor(n, o) {}
  BinaryDescriptor* lbd() { return mLbd.get(); }
};
struct Handle {
  cv::line_descriptor_c::LSDDetectorC::LSDOptions opts;
  Extractor* ex = nullptr;
  std::vector<cv::Mat> pyramid;
};
}  // namespace

extern "C" {

// Tracking::ParseFeaturesParamFile (src/Tracking.cc:1500-1556): numOctaves = Line.nLevels, scale = Line.scaleFactor,
// min_length = Line.minLineLength, lineFitErrThreshold = Line.lineFitErrThreshold; LineExtractor(Line.nfeatures, opts).
void* ref_lines_create(int nfeatures, int nlevels, float scale, double min_length, double fit_err) {
  Handle* h = new Handle;
  h->opts.numOctaves = nlevels;
  h->opts.scale = scale;
  h->opts.min_length = min_length;
  h->opts.lineFitErrThreshold = fit_err;
  h->ex = new Extractor(nfeatures, h->opts);
  return h;
}
void ref_lines_destroy(void* hv) {
  Handle* h = static_cast<Handle*>(hv);
  delete h->ex;
  delete h;
}

// Frame::PrecomputeGaussianPyramid (src/Frame.cc:841-865): the ORB pyramid handed to the line extractor, border 0
void ref_lines_set_pyramid(void* hv, const uint8_t* const* levels, const int* w, const int* hh, int npyr, int num_octaves,
                           float scale) {
  Handle* h = static_cast<Handle*>(hv);
  h->pyramid.clear();
  for (int i = 0; i < npyr; ++i) {
    cv::Mat m(hh[i], w[i], CV_8UC1);
    for (int r = 0; r < hh[i]; ++r) std::memcpy(m.ptr(r), levels[i] + (size_t)r * w[i], (size_t)w[i]);
    h->pyramid.push_back(m);
  }
  if (npyr > 0) h->ex->SetGaussianPyramid(h->pyramid, num_octaves, scale, 0, 0);
}

// Frame::PrecomputeGaussianPyramid(0, im) itself (src/Frame.cc:841-852, USE_UNFILTERED_PYRAMID_FOR_LINES 1): the ORB
// extractor (a handle of orb_ref_wrap.cpp) builds and blurs its pyramid and hands its UNBLURRED levels — regions of
// interest inside the bordered level buffers, shared storage — to the line extractor.
void ref_frame_precompute_pyramid(void* orb_h, void* lines_h, const uint8_t* img, int w, int hh, int stride) {
  PLVS2::ORBextractor* orb = static_cast<PLVS2::ORBextractor*>(orb_h);
  Handle* h = static_cast<Handle*>(lines_h);
  cv::Mat image(hh, w, CV_8UC1, const_cast<uint8_t*>(img), (size_t)stride);
  orb->PrecomputeGaussianPyramid(image);
  h->ex->SetGaussianPyramid(orb->mvImagePyramid, h->ex->GetLevels(), orb->GetScaleFactor());
}

// Frame::ExtractLSD (src/Frame.cc:815-837): (*mpLineExtractor)(image, keylines, descriptors).
// keylines: cap entries of 68 bytes (KeyLine); desc: cap x 32.  Returns the line count.
int ref_lines_extract(void* hv, const uint8_t* img, int w, int hh, int stride, void* keylines, uint8_t* desc, int cap) {
  Handle* h = static_cast<Handle*>(hv);
  cv::Mat image(hh, w, CV_8UC1, const_cast<uint8_t*>(img), (size_t)stride);
  std::vector<KeyLine> lines;
  cv::Mat d;
  (*h->ex)(image, lines, d);
  static_assert(sizeof(KeyLine) == 68, "KeyLine must be 68 bytes");
  if ((int)lines.size() <= cap) {
    if (!lines.empty()) std::memcpy(keylines, lines.data(), lines.size() * sizeof(KeyLine));
    for (int i = 0; i < d.rows; ++i) std::memcpy(desc + (size_t)i * 32, d.ptr(i), 32);
  }
  return (int)lines.size();
}

// stage accessors for the last extract
int ref_lines_octave_size(void* hv, int octave, int* w, int* hh) {
  BinaryDescriptor* b = static_cast<Handle*>(hv)->ex->lbd();
  *w = b->images_sizes[octave].width;
  *hh = b->images_sizes[octave].height;
  return 0;
}
// which: 1 dx, 2 dy (CV_16S, w * h shorts) — the Sobel images EDLineDetector keeps for computeLBD
void ref_lines_get_map(void* hv, int octave, int which, void* out) {
  BinaryDescriptor* b = static_cast<Handle*>(hv)->ex->lbd();
  const cv::Mat& m = which == 1 ? b->edLineVec_[octave]->dxImg_ : b->edLineVec_[octave]->dyImg_;
  for (int r = 0; r < m.rows; ++r) std::memcpy(static_cast<uint8_t*>(out) + (size_t)r * m.cols * 2, m.ptr(r), (size_t)m.cols * 2);
}
int ref_lines_num_in_octave(void* hv, int octave) {
  return (int)static_cast<Handle*>(hv)->ex->lbd()->edLineVec_[octave]->lines_.numOfLines;
}

}  // extern "C"
