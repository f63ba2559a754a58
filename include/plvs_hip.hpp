// C++ host-side mirror of the PLVS interfaces that the hot path hides behind (header only, C++14).
//
// plvs_hip.h is the boundary (C ABI).  This header is what a PLVS translation unit includes to keep
// its own class names, argument meaning and error behaviour while the arithmetic runs in
// libplvs_hip.so.  The reference classes take cv::Mat / pcl::PointCloud / Sophus::SE3f; OpenCV,
// PCL, Eigen and Sophus are not part of this repository, so the mirror takes the plain views of
// those types defined below (same memory layout as the originals: a PLVS build passes
// mat.data / mat.step, cloud.points.data(), Twc.matrix3x4().data() — see INTEGRATION.md).
//
//   PLVS2hip::ORBextractor              include/ORBextractor.h:76-113, src/ORBextractor.cc:446, :1245
//   PLVS2hip::LineExtractor             include/LineExtractor.h:48-83, src/LineExtractor.cc:150, :170
//   PLVS2hip::BinaryDescriptorMatcher   binary_descriptor_matcher_custom.cpp:258
//   PLVS2hip::ORBmatcher                include/ORBmatcher.h (DescriptorDistance, SearchByProjection x2, SearchByBoW)
//   PLVS2hip::LineMatcher               src/LineMatcher.cc:156, :303, :454
//   PLVS2hip::ComputeStereoMatches      src/Frame.cc:1780
//   PLVS2hip::StereoSGM                 Thirdparty/libsgm/include/libsgm.h:57 (as src/PointCloudKeyFrame.cc:435 uses it)
//   PLVS2hip::PointCloudGenerator       src/PointCloudMapping.cc:796, :929
//   PLVS2hip::PointCloudMapChisel       include/PointCloudMapChisel.h:61, src/PointCloudMapChisel.cc:76-246
//   PLVS2hip::PointCloudMapVoxblox      include/PointCloudMapVoxblox.h:54, src/PointCloudMapVoxblox.cc:81
//
// Errors of the library surface as std::runtime_error carrying plvs_hip_last_error(); there is no
// CPU fallback anywhere.
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "plvs_hip.h"

namespace PLVS2hip {

inline void check(int rc) {
  if (rc != PLVS_OK) throw std::runtime_error(std::string("plvs_hip: ") + plvs_hip_last_error());
}

// ---- views of the reference's argument types
struct Image8U {      // cv::Mat of type CV_8UC1 (or CV_8UC3 for colour): rows, cols, step, data
  int rows = 0, cols = 0;
  size_t step = 0;
  const uint8_t* data = nullptr;
  bool empty() const { return data == nullptr || rows <= 0 || cols <= 0; }
};
struct Image32F {     // cv::Mat of type CV_32FC1 (depth)
  int rows = 0, cols = 0;
  size_t step = 0;    // bytes, like cv::Mat::step
  const float* data = nullptr;
};
using KeyPoint = plvs_keypoint;             // cv::KeyPoint, field for field
using KeyLine = plvs_keyline;               // cv::line_descriptor_c::KeyLine
using PointSurfelSegment = plvs_point_surfel;   // pcl::PointSurfelSegment, 48 bytes
struct DMatch {                             // cv::DMatch
  int queryIdx = -1, trainIdx = -1, imgIdx = 0;
  float distance = 0.f;
};
struct SE3f {                               // Sophus::SE3f as the 3x4 row-major [R|t] the integrators read
  float m[12];
};

// ------------------------------------------------------------------------------------ ORB
class ORBextractor {
 public:
  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST) : nfeatures_(nfeatures) {
    check(plvs_hip_orb_create(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, &h_));
  }
  ~ORBextractor() { plvs_hip_orb_destroy(h_); }
  ORBextractor(const ORBextractor&) = delete;
  ORBextractor& operator=(const ORBextractor&) = delete;

  // int operator()(InputArray image, InputArray mask /*ignored*/, vector<KeyPoint>&, OutputArray descriptors,
  //                vector<int>& vLappingArea): returns monoIndex, -1 on an empty image (ORBextractor.cc:1254).
  int operator()(const Image8U& image, std::vector<KeyPoint>& keypoints, std::vector<uint8_t>& descriptors,
                 const std::vector<int>& vLappingArea = {0, 0}) {
    keypoints.clear();
    descriptors.clear();
    if (image.empty()) return -1;
    int cap = 2 * nfeatures_ + 1024, n = 0, mono = 0;
    for (;;) {
      keypoints.resize((size_t)cap);
      descriptors.resize((size_t)cap * 32);
      const int lap0 = vLappingArea.size() > 0 ? vLappingArea[0] : 0, lap1 = vLappingArea.size() > 1 ? vLappingArea[1] : 0;
      const int rc = plvs_hip_orb_extract(h_, image.data, image.cols, image.rows, (int)image.step, lap0, lap1,
                                          keypoints.data(), descriptors.data(), cap, &n, &mono);
      if (rc == PLVS_ERR_EMPTY) { keypoints.clear(); descriptors.clear(); return -1; }
      check(rc);
      if (n > cap) { cap = n; continue; }   // nothing was written: call again with room for n
      break;
    }
    keypoints.resize((size_t)n);
    descriptors.resize((size_t)n * 32);
    return mono;
  }
  int GetLevels() { return plvs_hip_orb_get_levels(h_); }
  float GetScaleFactor() { return plvs_hip_orb_get_scale_factor(h_); }
  std::vector<float> GetScaleFactors() { return table(0); }
  std::vector<float> GetInverseScaleFactors() { return table(1); }
  std::vector<float> GetScaleSigmaSquares() { return table(2); }
  std::vector<float> GetInverseScaleSigmaSquares() { return table(3); }
  plvs_orb* handle() { return h_; }

 private:
  std::vector<float> table(int which) {
    const int n = GetLevels();
    std::vector<float> t[4];
    for (auto& v : t) v.resize((size_t)n);
    check(plvs_hip_orb_get_scale_tables(h_, t[0].data(), t[1].data(), t[2].data(), t[3].data()));
    return t[which];
  }
  plvs_orb* h_ = nullptr;
  int nfeatures_;
};

// ------------------------------------------------------------------------------------ lines
struct LSDOptions {   // LSDDetectorC::LSDOptions (descriptor_custom.hpp:928-957); the EDLines path reads the first four
  int numOctaves = 3;
  float scale = 1.2f;
  double min_length = 0.02;
  double lineFitErrThreshold = 1.6;
  int refine = 2;       // cv::LSD_REFINE_ADV — Tracking passes Line.LSD.refine (1), .logEps (1.0), .densityTh (0.6), Tracking.cc:1476-1482
  double sigma_scale = 0.6, quant = 2.0, ang_th = 22.5, log_eps = 0.0, density_th = 0.7;
  int n_bins = 1024;
  plvs_lsd_options c() const { return plvs_lsd_options{refine, (double)scale, sigma_scale, quant, ang_th, log_eps, density_th, n_bins}; }
};

class LineExtractor {
 public:
  // LineExtractor::skUseLsdExtractor (Line.LSD.on): read when an extractor is constructed, as the reference's mLsd is made
  static bool& UseLsdExtractor() { static bool on = false; return on; }
  LineExtractor(int numLinefeatures, const LSDOptions& opts = LSDOptions()) : n_(numLinefeatures), opts_(opts) {
    check(plvs_hip_lines_create(numLinefeatures, opts.numOctaves, opts.scale, opts.min_length, opts.lineFitErrThreshold, &h_));
    if (UseLsdExtractor()) check(plvs_hip_lsd_create(&lsd_));
  }
  ~LineExtractor() { plvs_hip_lines_destroy(h_); plvs_hip_lsd_destroy(lsd_); }
  LineExtractor(const LineExtractor&) = delete;
  LineExtractor& operator=(const LineExtractor&) = delete;

  // void operator()(const cv::Mat& image, std::vector<KeyLine>& keylines, cv::Mat& descriptors): empty outputs
  // (after "no lines!") when nothing is found, LineExtractor.cc:269-273.
  void operator()(const Image8U& image, std::vector<KeyLine>& keylines, std::vector<uint8_t>& descriptors) {
    int cap = 4096, n = 0;
    for (;;) {
      keylines.resize((size_t)cap);
      descriptors.resize((size_t)cap * 32);
      if (lsd_) {
        const plvs_lsd_options o = opts_.c();
        check(plvs_hip_lsd_extract(lsd_, image.data, image.cols, image.rows, (int)image.step, n_, opts_.numOctaves, &o,
                                   opts_.min_length, keylines.data(), descriptors.data(), cap, &n));
      } else {
        check(plvs_hip_lines_extract(h_, image.data, image.cols, image.rows, (int)image.step, keylines.data(),
                                     descriptors.data(), cap, &n));
      }
      if (n <= cap) break;
      cap = n;   // (nfeatures = 0 keeps every line)
    }
    keylines.resize((size_t)n);
    descriptors.resize((size_t)n * 32);
  }
  // SetGaussianPyramid(mpORBextractor->mvImagePyramid, levels, scale) of Frame::PrecomputeGaussianPyramid: the
  // pyramid stays on the device, so the extractor itself is handed over (nullptr = own pyramid again).
  void SetGaussianPyramid(ORBextractor* orb) {
    if (lsd_) throw std::invalid_argument("LineExtractor: Line.pyramidPrecomputation with Line.LSD.on is undefined behaviour in the reference (DESIGN.md 6), not reproduced");
    check(plvs_hip_lines_set_gaussian_pyramid(h_, orb ? orb->handle() : nullptr));
  }
  plvs_lines* handle() { return h_; }

 private:
  plvs_lines* h_ = nullptr;
  plvs_lsd* lsd_ = nullptr;
  int n_;
  LSDOptions opts_;
};

// ------------------------------------------------------------------------------------ matching
class BinaryDescriptorMatcher {
 public:
  // knnMatch(query, train, matches, k, mask, compactResult), k = 2 as LineMatcher calls it; query / train are
  // N x 32 byte rows.  Masked-out queries give an empty entry, dropped when compactResult.
  void knnMatch(const uint8_t* query, int nq, const uint8_t* train, int nt, std::vector<std::vector<DMatch>>& matches,
                int k = 2, const uint8_t* mask = nullptr, bool compactResult = false) const {
    matches.clear();
    if (k != 2) throw std::invalid_argument("BinaryDescriptorMatcher::knnMatch: only k = 2 is on the accelerated path");
    if (nq <= 0 || nt <= 0) return;   // "empty query or train descriptors", :263-266
    std::vector<int32_t> idx((size_t)2 * nq), dist((size_t)2 * nq);
    check(plvs_hip_hamming_knn2(query, nq, train, nt, mask, PLVS_TIE_MIH, idx.data(), dist.data()));
    for (int q = 0; q < nq; ++q) {
      std::vector<DMatch> row;
      for (int j = 0; j < 2; ++j)
        if (idx[2 * q + j] >= 0) {
          DMatch m;
          m.queryIdx = q;
          m.trainIdx = idx[2 * q + j];
          m.distance = (float)dist[2 * q + j];
          row.push_back(m);
        }
      if (!(compactResult && (mask != nullptr && mask[q] == 0))) matches.push_back(row);
    }
  }
};

// The search functions take the arrays the reference functions read of their Frame / MapPoint / KeyFrame
// objects (plvs_frame_view, plvs_mappoint_view, plvs_lastframe_view, plvs_featvec_view of plvs_hip.h) and
// return the reference's return value; `assigned[i]` is what the reference stores per keypoint (an index
// into the other side, -1 = none) — the caller maps it back to its MapPoint* / MapLine*.
class ORBmatcher {
 public:
  static constexpr int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 12;
  explicit ORBmatcher(float nnratio = 0.6f, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

  static int DescriptorDistance(const uint8_t* a, const uint8_t* b) {
    const int32_t z = 0;
    int32_t d = 0;
    check(plvs_hip_hamming_pairs(a, 1, b, 1, &z, &z, 1, &d));
    return d;
  }
  // SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, th, bFarPoints, thFarPoints), ORBmatcher.cc:71
  int SearchByProjection(const plvs_frame_view& F, const plvs_mappoint_view& vpMapPoints, float th, bool bFarPoints,
                         float thFarPoints, const uint8_t* occupied, std::vector<int32_t>& assigned) const {
    assigned.assign((size_t)F.n, -1);
    int n = 0;
    check(plvs_hip_orb_search_by_projection(&F, &vpMapPoints, th, bFarPoints ? 1 : 0, thFarPoints, mfNNratio, occupied,
                                            assigned.data(), &n));
    return n;
  }
  // SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono), ORBmatcher.cc:1774
  int SearchByProjection(const plvs_frame_view& CurrentFrame, const float* curAngles, float mnMaxX, float mnMaxY, float mbf,
                         const plvs_lastframe_view& LastFrame, float th, bool bForward, bool bBackward,
                         const uint8_t* occupied, std::vector<int32_t>& assigned) const {
    assigned.assign((size_t)CurrentFrame.n, -1);
    int n = 0;
    check(plvs_hip_orb_search_by_projection_ff(&CurrentFrame, curAngles, mnMaxX, mnMaxY, mbf, &LastFrame, th, bForward ? 1 : 0,
                                               bBackward ? 1 : 0, mbCheckOrientation ? 1 : 0, occupied, assigned.data(), &n));
    return n;
  }
  // SearchByBoW(KeyFramePtr& pKF, Frame& F, vpMapPointMatches), ORBmatcher.cc:300
  int SearchByBoW(const plvs_featvec_view& kfFeatVec, const uint8_t* kfDescriptors, int kfN, const uint8_t* kfValid,
                  const float* kfAngles, const plvs_featvec_view& fFeatVec, const uint8_t* fDescriptors, int fN,
                  const float* fAngles, std::vector<int32_t>& assigned) const {
    assigned.assign((size_t)fN, -1);
    int n = 0;
    check(plvs_hip_orb_search_by_bow(&kfFeatVec, kfDescriptors, kfN, kfValid, kfAngles, &fFeatVec, fDescriptors, fN, fAngles,
                                     mfNNratio, mbCheckOrientation ? 1 : 0, assigned.data(), &n));
    return n;
  }
  float mfNNratio;
  bool mbCheckOrientation;
};

class LineMatcher {
 public:
  static constexpr int TH_HIGH = 110, TH_LOW = 60, TH_LOW_STEREO = 50, HISTO_LENGTH = 12;   // LineMatcher.cc:87-90
  explicit LineMatcher(float nnratio = 0.6f, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}
  // SearchByKnn(Frame& CurrentFrame, const Frame& LastFrame), LineMatcher.cc:303
  int SearchByKnn(const uint8_t* descLast, int nLast, const uint8_t* validLast, const float* angleLast,
                  const uint8_t* descCur, int nCur, const float* angleCur, std::vector<int32_t>& assigned) const {
    assigned.assign((size_t)nCur, -1);
    int n = 0;
    check(plvs_hip_lines_search_by_knn(descLast, nLast, validLast, angleLast, descCur, nCur, angleCur, mfNNratio,
                                       mbCheckOrientation ? 1 : 0, assigned.data(), &n));
    return n;
  }
  // SearchByKnn(KeyFramePtr& pKF, const Frame& F, vpMapLineMatches), LineMatcher.cc:156
  int SearchByKnnKF(const uint8_t* descKF, int nKF, const uint8_t* validKF, const float* angleKF, const uint8_t* descF,
                    int nF, const float* angleF, std::vector<int32_t>& assigned) const {
    assigned.assign((size_t)nF, -1);
    int n = 0;
    check(plvs_hip_lines_search_by_knn_kf(descKF, nKF, validKF, angleKF, descF, nF, angleF, mfNNratio,
                                          mbCheckOrientation ? 1 : 0, assigned.data(), &n));
    return n;
  }
  // SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, bLargerSearch, bMono), LineMatcher.cc:837 — the
  // per-frame line match of Tracking::TrackWithMotionModel.  The caller hands over the projections of the last
  // frame's map lines into the current frame (see plvs_hip_lines_search_by_projection_ff); direction: 0, 1 forward,
  // 2 backward.  assigned[i2] = last-frame line whose map line the current line i2 takes, -1 = none.
  int SearchByProjection(const plvs_line_frame_view& CurrentFrame, const uint8_t* occupied, int nLast, const uint8_t* valid,
                         const float* proj, const int32_t* octave, const float* angle, const uint8_t* desc,
                         const uint8_t* hasObservations, bool bLargerSearch, int direction,
                         std::vector<int32_t>& assigned) const {
    assigned.assign((size_t)CurrentFrame.n, -1);
    int n = 0;
    check(plvs_hip_lines_search_by_projection_ff(&CurrentFrame, occupied, nLast, valid, proj, octave, angle, desc,
                                                 hasObservations, bLargerSearch ? 1 : 0, direction, mfNNratio,
                                                 mbCheckOrientation ? 1 : 0, assigned.data(), &n));
    return n;
  }
  // SearchByProjection(Frame& F, const std::vector<MapLinePtr>& vpMapLines, bLargerSearch), LineMatcher.cc:1286 —
  // Tracking::SearchLocalLines.  assigned[i2] = map line the frame line i2 takes, -1 = none.
  int SearchByProjection(const plvs_line_frame_view& F, const uint8_t* occupied, int nMapLines, const uint8_t* inView,
                         const float* proj, const int32_t* level, const uint8_t* desc, const uint8_t* hasObservations,
                         bool bLargerSearch, std::vector<int32_t>& assigned) const {
    assigned.assign((size_t)F.n, -1);
    int n = 0;
    check(plvs_hip_lines_search_by_projection(&F, occupied, nMapLines, inView, proj, level, desc, hasObservations,
                                              bLargerSearch ? 1 : 0, mfNNratio, assigned.data(), &n));
    return n;
  }
  // SearchStereoMatchesByKnn(frame, vMatches, vValidMatches, descriptorDist), LineMatcher.cc:454
  int SearchStereoMatchesByKnn(const uint8_t* descLeft, int nLeft, const float* angleLeft, const int32_t* octaveLeft,
                               const uint8_t* descRight, int nRight, const float* angleRight, const int32_t* octaveRight,
                               std::vector<DMatch>& vMatches, std::vector<bool>& vValidMatches,
                               int descriptorDist = TH_LOW_STEREO) const {
    const int cap = nRight > 0 ? nRight : 1;
    std::vector<int32_t> q((size_t)cap), t((size_t)cap);
    std::vector<float> d((size_t)cap);
    std::vector<uint8_t> v((size_t)cap);
    int k = 0, n = 0;
    check(plvs_hip_lines_search_stereo_by_knn(descLeft, nLeft, angleLeft, octaveLeft, descRight, nRight, angleRight,
                                              octaveRight, mfNNratio, mbCheckOrientation ? 1 : 0, descriptorDist, q.data(),
                                              t.data(), d.data(), v.data(), cap, &k, &n));
    vMatches.clear();
    vValidMatches.clear();
    for (int i = 0; i < k; ++i) {
      DMatch m;
      m.queryIdx = q[(size_t)i]; m.trainIdx = t[(size_t)i]; m.distance = d[(size_t)i];
      vMatches.push_back(m);
      vValidMatches.push_back(v[(size_t)i] != 0);
    }
    return n;
  }
  float mfNNratio;
  bool mbCheckOrientation;
};

// Frame::ComputeStereoMatches: mvuRight / mvDepth of the left keypoints (-1 = none).
inline void ComputeStereoMatches(ORBextractor& left, ORBextractor& right, const std::vector<KeyPoint>& mvKeys,
                                 const std::vector<uint8_t>& mDescriptors, const std::vector<KeyPoint>& mvKeysRight,
                                 const std::vector<uint8_t>& mDescriptorsRight, float mb, float mbf,
                                 std::vector<float>& mvuRight, std::vector<float>& mvDepth) {
  plvs_stereo* s = nullptr;
  check(plvs_hip_stereo_create(left.handle(), right.handle(), &s));
  mvuRight.assign(mvKeys.size(), -1.0f);
  mvDepth.assign(mvKeys.size(), -1.0f);
  int n = 0;
  const int rc = plvs_hip_stereo_matches(s, mvKeys.data(), mDescriptors.data(), (int)mvKeys.size(), mvKeysRight.data(),
                                         mDescriptorsRight.data(), (int)mvKeysRight.size(), mb, mbf, mvuRight.data(),
                                         mvDepth.data(), &n);
  plvs_hip_stereo_destroy(s);
  check(rc);
}

// ------------------------------------------------------------------------------------ dense stereo
// sgm::StereoSGM (Thirdparty/libsgm/include/libsgm.h:57-110) for 8-bit images and an 8-bit disparity, the way
// PointCloudKeyFrame::ProcessStereoLibsgm constructs it; execute() is EXECUTE_INOUT_HOST2HOST.
class StereoSGM {
 public:
  struct Parameters {
    int P1, P2;
    float uniqueness;
    Parameters(int P1 = 10, int P2 = 120, float uniqueness = 0.95f) : P1(P1), P2(P2), uniqueness(uniqueness) {}
  };
  StereoSGM(int width, int height, int disparity_size, int input_depth_bits = 8, int output_depth_bits = 8,
            const Parameters& param = Parameters()) {
    if (input_depth_bits != 8 || output_depth_bits != 8) throw std::logic_error("depth bits must be 8 on the accelerated path");
    if (disparity_size != 64 && disparity_size != 128) throw std::logic_error("disparity size must be 64 or 128");
    check(plvs_hip_sgm_create(width, height, disparity_size, param.P1, param.P2, param.uniqueness, &h_));
  }
  ~StereoSGM() { plvs_hip_sgm_destroy(h_); }
  StereoSGM(const StereoSGM&) = delete;
  StereoSGM& operator=(const StereoSGM&) = delete;
  void execute(const void* left_pixels, const void* right_pixels, void* dst) {
    check(plvs_hip_sgm_execute(h_, static_cast<const uint8_t*>(left_pixels), static_cast<const uint8_t*>(right_pixels),
                               static_cast<uint8_t*>(dst)));
  }

 private:
  plvs_sgm* h_ = nullptr;
};

// libelas::ElasGPU (Thirdparty/libelas-gpu/GPU/elas_gpu.h:29-47) without the inheritance: the two methods of
// libelas::Elas it overrides.  In a PLVS tree the class is the reference's own —
//     class ElasGPU : public Elas { ... computeDisparity(...) override; adaptiveMean(float* D) override; };
// — with these two bodies (INTEGRATION.md shows it); here the reference's types are spelled out so that the header
// stands alone.  support_pt / triangle have the layout of Elas::support_pt / Elas::triangle (elas.h:178-190).
class ElasGPU {
 public:
  struct support_pt { int32_t u, v, d; };
  struct triangle { int32_t c1, c2, c3; float t1a, t1b, t1c, t2a, t2b, t2c; };
  struct Parameters {   // the fields of Elas::Parameters the two methods read; defaults: ROBOTICS (elas.h:97-121)
    bool subsampling;
    int32_t grid_size, match_texture;
    float beta, gamma, sigma, sradius;
    int32_t disp_min, disp_max, candidate_stepsize, support_texture, lr_threshold;   // supportCandidates (lr_threshold: the check too)
    float support_threshold;
    float speckle_sim_threshold;                                                      // the post-processing methods
    int32_t speckle_size, ipol_gap_width;
    bool add_corners;
    Parameters()
        : subsampling(false), grid_size(20), match_texture(1), beta(0.02f), gamma(3), sigma(1), sradius(2), disp_min(0),
          disp_max(255), candidate_stepsize(5), support_texture(10), lr_threshold(2), support_threshold(0.85f),
          speckle_sim_threshold(1), speckle_size(200), ipol_gap_width(3), add_corners(false) {}
  };
  explicit ElasGPU(const Parameters& param = Parameters()) : param(param) {
    plvs_elas_params p{param.subsampling ? 1 : 0, param.grid_size, param.match_texture, param.beta, param.gamma, param.sigma,
                       param.sradius, param.disp_min, param.disp_max, param.candidate_stepsize, param.support_texture,
                       param.lr_threshold, param.support_threshold, param.speckle_sim_threshold, param.speckle_size,
                       param.ipol_gap_width, param.add_corners ? 1 : 0};
    check(plvs_hip_elas_create(&p, &h_));
  }
  ~ElasGPU() { plvs_hip_elas_destroy(h_); }
  ElasGPU(const ElasGPU&) = delete;
  ElasGPU& operator=(const ElasGPU&) = delete;
  // Elas::process sets width / height before it calls the two methods (elas.cpp:39-40); here the caller does.
  int32_t width = 0, height = 0;
  Parameters param;

  // Elas::computeDisparity's arguments.  The descriptor images are uploaded with the left image's call and reused by
  // the right image's (Elas::process calls the two back to back on the same pair, elas.cpp:113-114).
  void computeDisparity(const std::vector<support_pt>& p_support, const std::vector<triangle>& tri, const int32_t* disparity_grid,
                        const int32_t* grid_dims, const uint8_t* I1_desc, const uint8_t* I2_desc, bool right_image, float* D) {
    static_assert(sizeof(support_pt) == 12 && sizeof(triangle) == 36, "record layouts of the C ABI");
    const bool staged = (right_image || staged_by_support_) && I1_desc == staged_1_ && I2_desc == staged_2_ && staged_w_ == width &&
                        staged_h_ == height;
    check(plvs_hip_elas_compute_disparity(h_, reinterpret_cast<const int32_t*>(p_support.data()), (int)p_support.size(), tri.data(),
                                          (int)tri.size(), disparity_grid, grid_dims, staged ? nullptr : I1_desc,
                                          staged ? nullptr : I2_desc, width, height, right_image ? 1 : 0, D));
    staged_by_support_ = staged_by_support_ && !right_image;   // (the pair supportCandidates staged serves its two calls)
    staged_1_ = right_image ? nullptr : I1_desc;       // (only a left-image call vouches for the staged pair)
    staged_2_ = right_image ? nullptr : I2_desc;
    staged_w_ = width;
    staged_h_ = height;
  }
  void adaptiveMean(float* D) { check(plvs_hip_elas_adaptive_mean(h_, D, width, height)); }
  // the other virtual post-processing methods of Elas (elas.cpp:971-1347), in place
  void leftRightConsistencyCheck(float* D1, float* D2) { check(plvs_hip_elas_left_right_check(h_, D1, D2, width, height)); }
  void removeSmallSegments(float* D) { check(plvs_hip_elas_remove_small_segments(h_, D, width, height)); }
  void gapInterpolation(float* D) { check(plvs_hip_elas_gap_interpolation(h_, D, width, height)); }
  // libelas::Descriptor of both images on the device; supportCandidates / computeDisparity then take nullptr descriptors
  void setImages(const uint8_t* I1, const uint8_t* I2, int32_t stride) {
    check(plvs_hip_elas_set_images(h_, I1, I2, width, height, stride));
    staged_1_ = staged_2_ = nullptr;
    staged_w_ = width;
    staged_h_ = height;
    staged_by_support_ = true;
  }
  // The candidate loop of Elas::computeSupportMatches (elas.cpp:434-456): D_can as that function allocates it
  // (D_can_width x D_can_height int16); the reference's filters follow on the host.  Stages the descriptor pair.
  void supportCandidates(const uint8_t* I1_desc, const uint8_t* I2_desc, int16_t* D_can) {
    check(plvs_hip_elas_support_candidates(h_, I1_desc, I2_desc, width, height, D_can));
    staged_1_ = I1_desc;
    staged_2_ = I2_desc;
    staged_w_ = width;
    staged_h_ = height;
    staged_by_support_ = true;
  }

 private:
  plvs_elas* h_ = nullptr;
  const uint8_t *staged_1_ = nullptr, *staged_2_ = nullptr;
  int32_t staged_w_ = 0, staged_h_ = 0;
  bool staged_by_support_ = false;
};

// ------------------------------------------------------------------------------------ depth -> cloud
class PointCloudGenerator {   // PointCloudMapping::InitCamGridPoints + GeneratePointCloudInCameraFrameBGRA
 public:
  // matCamGridPoints: N x 2 floats after InitCamGridPoints, or empty for an undistorted camera (computed here)
  PointCloudGenerator(int width, int height, int downsampleStep, double fx, double fy, double cx, double cy,
                      double minDepth, double maxDepth, std::vector<float> matCamGridPoints = {})
      : w_(width), h_img_(height), min_(minDepth), max_(maxDepth) {
    ngrid_ = plvs_hip_cloudgen_num_grid_points(width, height, downsampleStep);
    if (matCamGridPoints.empty()) {
      matCamGridPoints.resize((size_t)2 * ngrid_);
      check(plvs_hip_cloudgen_grid_points(width, height, downsampleStep, fx, fy, cx, cy, matCamGridPoints.data()));
    }
    check(plvs_hip_cloudgen_create(width, height, downsampleStep, matCamGridPoints.data(), &h_));
  }
  ~PointCloudGenerator() { plvs_hip_cloudgen_destroy(h_); }
  PointCloudGenerator(const PointCloudGenerator&) = delete;
  PointCloudGenerator& operator=(const PointCloudGenerator&) = delete;

  std::vector<PointSurfelSegment> GeneratePointCloudInCameraFrameBGRA(uint32_t kfid, const Image8U& color,
                                                                      const Image32F& depth,
                                                                      std::vector<int32_t>* pixelToPointIndex = nullptr) {
    std::vector<PointSurfelSegment> cloud((size_t)ngrid_);
    if (pixelToPointIndex) pixelToPointIndex->resize((size_t)w_ * h_img_);
    int n = 0;
    check(plvs_hip_cloudgen_generate(h_, depth.data, (int)(depth.step / sizeof(float)), color.data, (int)color.step, min_,
                                     max_, kfid, cloud.data(), ngrid_, pixelToPointIndex ? pixelToPointIndex->data() : nullptr,
                                     &n));
    cloud.resize((size_t)n);
    return cloud;
  }

 private:
  plvs_cloudgen* h_ = nullptr;
  int w_, h_img_, ngrid_ = 0;
  double min_, max_;
};

// ------------------------------------------------------------------------------------ chisel map
struct Mesh {   // chisel::Mesh after RecomputeMesh
  std::vector<float> vertices, normals, colors;   // n x 3
  std::vector<uint32_t> kfids;
};

class PointCloudMapChisel {
 public:
  using ChunkID = std::tuple<int, int, int>;
  // bCloudDeformationOnSparseMapChange: the map keeps the reference's chunk order from its first cloud on, so that
  // OnMapChange can deform it as Chisel::Deform does (plvs_hip_tsdf_chisel_enable_deform)
  explicit PointCloudMapChisel(float resolution, bool useCarving = false, float carvingDist = 0.05f,
                               float nearPlaneDist = 0.05f, float farPlaneDist = 5.0f, bool bResetOnSparseMapChange = true,
                               bool bCloudDeformationOnSparseMapChange = false, bool queueInsertions = true)
      : useCarving_(useCarving), carvingDist_(carvingDist), near_(nearPlaneDist), far_(farPlaneDist),
        resetOnChange_(bResetOnSparseMapChange), deformOnChange_(bCloudDeformationOnSparseMapChange),
        // queueInsertions: InsertCloud uploads the key frame's cloud, UpdateMap (or whatever reads or changes the map
        // first) integrates the ones waiting — at most five between two UpdateMap calls of PointCloudMapping
        // (src/PointCloudMapping.cc:540-552) — in one batch: the reference only reads the map in UpdateMap, and the batch
        // gives the same map bit for bit in the (default) ordered mode (plvs_hip_tsdf_chisel_queue).  Off with the
        // deformation bookkeeping, which follows the map call by call.
        queue_(queueInsertions && !bCloudDeformationOnSparseMapChange) {
    plvs_tsdf_chisel_params p;
    check(plvs_hip_tsdf_chisel_default_params(resolution, &p));
    check(plvs_hip_tsdf_chisel_create(&p, &h_));
    if (deformOnChange_) check(plvs_hip_tsdf_chisel_enable_deform(h_));
  }
  ~PointCloudMapChisel() { plvs_hip_tsdf_chisel_destroy(h_); }
  PointCloudMapChisel(const PointCloudMapChisel&) = delete;
  PointCloudMapChisel& operator=(const PointCloudMapChisel&) = delete;

  // InsertCloud(cloud_camera, Twc, max_range): SetPointCloud + IntegrateLastPointCloud (PointCloudMapChisel.cc:76-98)
  void InsertCloud(const std::vector<PointSurfelSegment>& cloud_camera, const SE3f& Twc, double /*max_range*/ = 0) {
    const size_t n = cloud_camera.size();
    xyz_.resize(3 * n);
    rgb_.resize(3 * n);
    kfid_.resize(n);
    for (size_t i = 0; i < n; ++i) {   // PclPointCloudToChisel: position, the r, g, b members, kfid
      const PointSurfelSegment& p = cloud_camera[i];
      xyz_[3 * i] = p.x; xyz_[3 * i + 1] = p.y; xyz_[3 * i + 2] = p.z;
      rgb_[3 * i] = p.r; rgb_[3 * i + 1] = p.g; rgb_[3 * i + 2] = p.b;
      kfid_[i] = p.kfid;
    }
    if (queue_) {
      check(plvs_hip_tsdf_chisel_queue(h_, xyz_.data(), rgb_.data(), kfid_.data(), (int)n, Twc.m));
      return;
    }
    check(plvs_hip_tsdf_chisel_integrate(h_, xyz_.data(), rgb_.data(), kfid_.data(), (int)n, Twc.m));
    MarkUpdated();
  }
  // Integrates the queued key frames (one batch) and notes the meshes they invalidate.
  void Flush() {
    int waiting = 0;
    check(plvs_hip_tsdf_chisel_queued(h_, &waiting));
    if (waiting > 0) {
      check(plvs_hip_tsdf_chisel_flush(h_));
      MarkUpdated();
    }
  }
  // InsertCloudWithDepth: carving of the depth image's frustum first when useCarving (Chisel.cpp:394-438)
  void InsertCloudWithDepth(const std::vector<PointSurfelSegment>& cloud_camera, const SE3f& Twc, const Image32F& depthImage,
                            float fx, float fy, float cx, float cy, double max_range = 0) {
    if (useCarving_) {
      Flush();   // (carving reads and changes the map: what is waiting goes in first)
      if (depthImage.step != (size_t)depthImage.cols * sizeof(float))
        throw std::invalid_argument("InsertCloudWithDepth: the depth image must be continuous");
      int carved = 0;
      check(plvs_hip_tsdf_chisel_carve(h_, depthImage.data, depthImage.cols, depthImage.rows, fx, fy, cx, cy, near_, far_,
                                       Twc.m, carvingDist_, &carved));
      if (carved > 0) {   // meshesToUpdate[chunkID] = true for every carved chunk, the chunk alone (Chisel.cpp:432)
        std::vector<int32_t> ids((size_t)3 * carved);
        int n = 0;
        check(plvs_hip_tsdf_chisel_updated_chunk_ids(h_, ids.data(), carved, &n));
        for (int i = 0; i < n && i < carved; ++i) meshesToUpdate_.insert(ChunkID(ids[3 * i], ids[3 * i + 1], ids[3 * i + 2]));
      }
    }
    InsertCloud(cloud_camera, Twc, max_range);
  }
  // UpdateMap: UpdateMesh (meshes of the 27-neighbourhood of every chunk updated since the last call,
  // Chisel.cpp:553-568) + GetPointCloud (ChiselServer.cpp:971-1068).  Returns the cloud size.
  int UpdateMap() {
    Flush();
    std::vector<int32_t> ids;
    for (const ChunkID& c : meshesToUpdate_) { ids.push_back(std::get<0>(c)); ids.push_back(std::get<1>(c)); ids.push_back(std::get<2>(c)); }
    const int nch = (int)meshesToUpdate_.size();
    if (nch > 0) {
      std::vector<int32_t> first((size_t)nch + 1);
      std::vector<float> V, N, C;
      std::vector<uint32_t> K;
      int nv = 0;
      int rc = plvs_hip_tsdf_chisel_mesh_chunks(h_, ids.data(), nch, nullptr, nullptr, nullptr, nullptr, 0, first.data(), &nv);
      if (rc == PLVS_ERR_CAPACITY) {
        V.resize((size_t)3 * nv); N.resize((size_t)3 * nv); C.resize((size_t)3 * nv); K.resize((size_t)nv);
        rc = plvs_hip_tsdf_chisel_mesh_chunks(h_, ids.data(), nch, V.data(), N.data(), C.data(), K.data(), nv, first.data(), &nv);
      }
      check(rc);
      int c = 0;
      for (const ChunkID& id : meshesToUpdate_) {
        const int a = first[(size_t)c], b = first[(size_t)c + 1];
        ++c;
        if (a == b) {
          // RecomputeMesh stores non-empty meshes only (ChunkManager.cpp:165-167) — but it re-uses the mesh object
          // already in allMeshes and GenerateMesh clears it first (:581): an existing mesh becomes empty in place
          auto it = allMeshes_.find(id);
          if (it != allMeshes_.end()) { it->second.vertices.clear(); it->second.normals.clear(); it->second.colors.clear(); it->second.kfids.clear(); }
          continue;
        }
        Mesh& m = allMeshes_[id];
        m.vertices.assign(V.begin() + 3 * a, V.begin() + 3 * b);
        m.normals.assign(N.begin() + 3 * a, N.begin() + 3 * b);
        m.colors.assign(C.begin() + 3 * a, C.begin() + 3 * b);
        m.kfids.assign(K.begin() + a, K.begin() + b);
      }
      meshesToUpdate_.clear();
    }
    pointCloud_.clear();
    for (const auto& kv : allMeshes_) {   // the reference walks an unordered_map: its order is unspecified, this one is by id
      const Mesh& m = kv.second;
      for (size_t i = 0; i < m.kfids.size(); ++i) {
        PointSurfelSegment p;
        std::memset(&p, 0, sizeof p);
        p.x = m.vertices[3 * i]; p.y = m.vertices[3 * i + 1]; p.z = m.vertices[3 * i + 2];
        p.r = (uint8_t)(m.colors[3 * i] * 255);          // point.r = meshCol[0]*255, ChiselServer.cpp:1023-1025
        p.g = (uint8_t)(m.colors[3 * i + 1] * 255);
        p.b = (uint8_t)(m.colors[3 * i + 2] * 255);
        p.normal_x = m.normals[3 * i]; p.normal_y = m.normals[3 * i + 1]; p.normal_z = m.normals[3 * i + 2];
        p.kfid = m.kfids[i];
        pointCloud_.push_back(p);
      }
    }
    return (int)pointCloud_.size();
  }
  // LoadMap once PointCloudMap::LoadMap has read the saved cloud (src/PointCloudMapChisel.cc:527-546):
  // ChiselServer::IntegrateWorldPointCloud(cloud, identity) — every point along its normal
  // (Chisel::IntegrateWorldPointCloudWithNormals) — then UpdateMap.
  int LoadMap(const std::vector<PointSurfelSegment>& cloud) {
    const size_t n = cloud.size();
    xyz_.resize(3 * n); rgb_.resize(3 * n); kfid_.resize(n);
    std::vector<float> nrm(3 * n);
    for (size_t i = 0; i < n; ++i) {
      const PointSurfelSegment& p = cloud[i];
      xyz_[3 * i] = p.x; xyz_[3 * i + 1] = p.y; xyz_[3 * i + 2] = p.z;
      rgb_[3 * i] = p.r; rgb_[3 * i + 1] = p.g; rgb_[3 * i + 2] = p.b;
      nrm[3 * i] = p.normal_x; nrm[3 * i + 1] = p.normal_y; nrm[3 * i + 2] = p.normal_z;
      kfid_[i] = p.kfid;
    }
    const float identity[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    Flush();
    check(plvs_hip_tsdf_chisel_integrate_world_normals(h_, xyz_.data(), rgb_.data(), kfid_.data(), nrm.data(), (int)n, identity));
    MarkUpdated();
    return UpdateMap();
  }
  void Clear() {
    check(plvs_hip_tsdf_chisel_clear(h_));
    meshesToUpdate_.clear();
    allMeshes_.clear();
    pointCloud_.clear();
  }
  // OnMapChange (src/PointCloudMapChisel.cc:262-274, :389-496).  mapKfidToRt: kfid -> Twc_new * Tcw_at_integration
  // (12 floats: R row-major, then t) for every valid key frame, which the caller derives from its key frames as
  // :420-478 does.  Reset, or UpdateMap + Chisel::Deform (volume and stored meshes) + UpdateMap.
  struct Rt12 { float v[12]; };
  int OnMapChange(const std::map<uint32_t, Rt12>& mapKfidToRt = {}) {
    if (resetOnChange_) { check(plvs_hip_tsdf_chisel_clear(h_)); meshesToUpdate_.clear(); allMeshes_.clear(); }
    if (deformOnChange_) {
      UpdateMap();
      std::vector<uint32_t> kf;
      std::vector<float> rt;
      for (const auto& e : mapKfidToRt) { kf.push_back(e.first); rt.insert(rt.end(), e.second.v, e.second.v + 12); }
      plvs_tsdf_deform_stats st;
      check(plvs_hip_tsdf_chisel_deform(h_, kf.data(), rt.data(), (int)kf.size(), &st));
      for (auto& kv : allMeshes_) {   // ChunkManager.cpp:1020-1051
        Mesh& m = kv.second;
        if (!m.kfids.empty())
          check(plvs_hip_tsdf_chisel_deform_mesh(m.vertices.data(), m.normals.data(), m.kfids.data(), (int)m.kfids.size(), kf.data(),
                                                 rt.data(), (int)kf.size()));
      }
    }
    return UpdateMap();
  }
  const std::vector<PointSurfelSegment>& GetPointCloud() const { return pointCloud_; }
  const std::map<ChunkID, Mesh>& GetAllMeshes() const { return allMeshes_; }
  plvs_tsdf_chisel* handle() { return h_; }

 private:
  void MarkUpdated() {
    int n = 0;
    check(plvs_hip_tsdf_chisel_updated_chunk_ids(h_, nullptr, 0, &n));
    std::vector<int32_t> ids((size_t)3 * (n > 0 ? n : 1));
    if (n > 0) check(plvs_hip_tsdf_chisel_updated_chunk_ids(h_, ids.data(), n, &n));
    for (int i = 0; i < n; ++i)
      for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
          for (int dz = -1; dz <= 1; ++dz)
            meshesToUpdate_.insert(ChunkID(ids[3 * i] + dx, ids[3 * i + 1] + dy, ids[3 * i + 2] + dz));
  }
  plvs_tsdf_chisel* h_ = nullptr;
  bool useCarving_;
  float carvingDist_, near_, far_;
  bool resetOnChange_ = true, deformOnChange_ = false, queue_ = true;
  std::vector<float> xyz_;
  std::vector<uint8_t> rgb_;
  std::vector<uint32_t> kfid_;
  std::set<ChunkID> meshesToUpdate_;
  std::map<ChunkID, Mesh> allMeshes_;
  std::vector<PointSurfelSegment> pointCloud_;
};

// ------------------------------------------------------------------------------------ voxblox map
struct VoxbloxMesh {   // voxblox::Mesh after updateMeshForBlock (three consecutive vertices = one triangle)
  std::vector<float> vertices, normals;   // n x 3
  std::vector<uint8_t> colors;            // n x 4: Color r, g, b, a
};

class PointCloudMapVoxblox {
 public:
  using BlockID = std::tuple<int, int, int>;
  // integrationMethod: PointCloudMapping.voxbloxIntegrationMethod (src/PointCloudMapVoxblox.cc:44, :74) — "simple",
  // "merged" and "fast" are all on the accelerated path, each the one-thread schedule of the reference's integrator bit
  // for bit ("fast": with its two approximate hash sets word for word, kept from scan to scan; plvs_hip.h has the
  // details — on the device it is the slowest of the three: INTEGRATION.md §4).  The default is the reference's own
  // (skIntegrationMethod = "fast", :44); "simple" is the opt-in for speed.
  // queueInsertions ("simple" only): InsertCloud uploads, UpdateMap — the one reader of the layer — integrates what waits as
  // one batch (plvs_hip_tsdf_voxblox_queue / _flush): the same layer bit for bit, at a quarter of the per-key-frame cost.
  explicit PointCloudMapVoxblox(float voxelSize, bool useCarving = false, const std::string& integrationMethod = "fast",
                                bool queueInsertions = true)
      : merged_(integrationMethod == "merged"), fast_(integrationMethod == "fast"),
        queue_(queueInsertions && integrationMethod == "simple") {
    if (integrationMethod != "simple" && integrationMethod != "merged" && integrationMethod != "fast")
      throw std::runtime_error("plvs_hip: unknown voxblox integration method '" + integrationMethod + "'");
    plvs_tsdf_voxblox_params p;
    check(plvs_hip_tsdf_voxblox_default_params(voxelSize, useCarving ? 1 : 0, &p));
    check(plvs_hip_tsdf_voxblox_create(&p, &h_));
  }
  ~PointCloudMapVoxblox() { plvs_hip_tsdf_voxblox_destroy(h_); }
  PointCloudMapVoxblox(const PointCloudMapVoxblox&) = delete;
  PointCloudMapVoxblox& operator=(const PointCloudMapVoxblox&) = delete;

  // InsertCloud: pcl cloud -> voxblox Pointcloud / Colors (r, g, b, a members) -> integratePointCloud
  // (PointCloudMapVoxblox.cc:81-99)
  void InsertCloud(const std::vector<PointSurfelSegment>& cloud_camera, const SE3f& Twc, double /*max_range*/ = 0) {
    const size_t n = cloud_camera.size();
    xyz_.resize(3 * n);
    rgba_.resize(4 * n);
    for (size_t i = 0; i < n; ++i) {
      const PointSurfelSegment& p = cloud_camera[i];
      xyz_[3 * i] = p.x; xyz_[3 * i + 1] = p.y; xyz_[3 * i + 2] = p.z;
      rgba_[4 * i] = p.r; rgba_[4 * i + 1] = p.g; rgba_[4 * i + 2] = p.b; rgba_[4 * i + 3] = p.a;
    }
    if (merged_) check(plvs_hip_tsdf_voxblox_integrate_merged(h_, xyz_.data(), rgba_.data(), (int)n, Twc.m));
    else if (fast_) check(plvs_hip_tsdf_voxblox_integrate_fast(h_, xyz_.data(), rgba_.data(), (int)n, Twc.m));
    else if (queue_) {
      check(plvs_hip_tsdf_voxblox_queue(h_, xyz_.data(), rgba_.data(), (int)n, Twc.m));
      pending_ = true;
      return;
    } else check(plvs_hip_tsdf_voxblox_integrate(h_, xyz_.data(), rgba_.data(), (int)n, Twc.m));
    MarkUpdated();
  }
  // Integrates what InsertCloud has queued (UpdateMap and every other reader of the layer call it).
  void Flush() {
    if (!pending_) return;
    check(plvs_hip_tsdf_voxblox_flush(h_));
    pending_ = false;
    MarkUpdated();
  }
  // The reference's LoadMap leaves the blocks of the loaded cloud outside the layer until the next InsertCloud
  // (integrateWorlPointCloud never publishes them: plvs_hip_tsdf_voxblox_set_deferred_world_blocks) — its UpdateMap
  // meshes none of them.  Off by default (the loaded map shows at once); on = what a PLVS build does today.
  void SetReferenceLoadMapVisibility(bool on) { Flush(); check(plvs_hip_tsdf_voxblox_set_deferred_world_blocks(h_, on ? 1 : 0)); }
  // LoadMap of a saved cloud once PointCloudMap::LoadMap has read it (src/PointCloudMapVoxblox.cc:233-258):
  // TsdfServer::insertWorldPointCloud(cloud, identity) — every point along its normal — then UpdateMap.
  int LoadMap(const std::vector<PointSurfelSegment>& cloud) {
    Flush();
    const size_t n = cloud.size();
    xyz_.resize(3 * n);
    rgba_.resize(4 * n);
    std::vector<float> nrm(3 * n);
    for (size_t i = 0; i < n; ++i) {
      const PointSurfelSegment& p = cloud[i];
      xyz_[3 * i] = p.x; xyz_[3 * i + 1] = p.y; xyz_[3 * i + 2] = p.z;
      rgba_[4 * i] = p.r; rgba_[4 * i + 1] = p.g; rgba_[4 * i + 2] = p.b; rgba_[4 * i + 3] = p.a;
      nrm[3 * i] = p.normal_x; nrm[3 * i + 1] = p.normal_y; nrm[3 * i + 2] = p.normal_z;
    }
    const float identity[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    check(plvs_hip_tsdf_voxblox_integrate_world_normals(h_, xyz_.data(), rgba_.data(), nrm.data(), (int)n, identity));
    MarkUpdated();
    return UpdateMap();
  }
  // UpdateMap: TsdfServer::updateMesh (generateMesh(only_mesh_updated_blocks, clear_updated_flag), tsdf_server.cc:775-787)
  // + getMeshAsPointcloud (voxblox_ros/mesh_vis.h:272-318, ColorMode::kColor)  (src/PointCloudMapVoxblox.cc:160-179).
  // Returns the cloud size.
  int UpdateMap() {
    Flush();
    const int nb = (int)updated_.size();
    if (nb > 0) {
      std::vector<int32_t> ids;
      for (const BlockID& b : updated_) { ids.push_back(std::get<0>(b)); ids.push_back(std::get<1>(b)); ids.push_back(std::get<2>(b)); }
      std::vector<int32_t> first((size_t)nb + 1);
      std::vector<float> V, N;
      std::vector<uint8_t> C;
      int nv = 0;
      int rc = plvs_hip_tsdf_voxblox_mesh_blocks(h_, ids.data(), nb, nullptr, nullptr, nullptr, 0, first.data(), &nv);
      if (rc == PLVS_ERR_CAPACITY) {
        V.resize((size_t)3 * nv); N.resize((size_t)3 * nv); C.resize((size_t)4 * nv);
        rc = plvs_hip_tsdf_voxblox_mesh_blocks(h_, ids.data(), nb, V.data(), N.data(), C.data(), nv, first.data(), &nv);
      }
      check(rc);
      int c = 0;
      for (const BlockID& id : updated_) {   // allocateMeshPtrByIndex + mesh->clear(): an updated block always owns a mesh
        const int a = first[(size_t)c], b = first[(size_t)c + 1];
        ++c;
        VoxbloxMesh& m = meshLayer_[id];
        m.vertices.assign(V.begin() + 3 * a, V.begin() + 3 * b);
        m.normals.assign(N.begin() + 3 * a, N.begin() + 3 * b);
        m.colors.assign(C.begin() + 4 * a, C.begin() + 4 * b);
      }
      updated_.clear();
    }
    pointCloud_.clear();
    for (const auto& kv : meshLayer_) {   // the reference walks an unordered_map: its order is unspecified, this one is by id
      const VoxbloxMesh& m = kv.second;
      for (size_t i = 0; i < m.vertices.size() / 3; ++i) {
        PointSurfelSegment p;
        std::memset(&p, 0, sizeof p);
        p.x = m.vertices[3 * i]; p.y = m.vertices[3 * i + 1]; p.z = m.vertices[3 * i + 2];
        p.r = CloudColour(m.colors[4 * i]); p.g = CloudColour(m.colors[4 * i + 1]); p.b = CloudColour(m.colors[4 * i + 2]);
        p.normal_x = m.normals[3 * i]; p.normal_y = m.normals[3 * i + 1]; p.normal_z = m.normals[3 * i + 2];
        pointCloud_.push_back(p);
      }
    }
    return (int)pointCloud_.size();
  }
  void Clear() {
    check(plvs_hip_tsdf_voxblox_clear(h_));   // (drops what was queued)
    pending_ = false;
    updated_.clear();
    meshLayer_.clear();
    pointCloud_.clear();
  }
  int NumBlocks() { Flush(); int n = 0; check(plvs_hip_tsdf_voxblox_num_blocks(h_, &n)); return n; }
  // The device side of TsdfServer::saveMap / loadMap (tsdf_server.cc:859-872): every block's voxel planes out of /
  // into HBM; the `.proto` serialisation of a StoredBlock stays with the reference's protobuf code.  A loaded block
  // replaces / creates its block and is marked updated (BlockMergingStrategy::kReplace, core/layer_inl.h:195-197, :215).
  struct StoredBlock {
    BlockID id;
    std::vector<float> distance, weight;   // 4096 each, index x + 16 * (y + 16 * z)
    std::vector<uint32_t> rgba;
  };
  std::vector<StoredBlock> SaveLayer() {
    const int n = NumBlocks();
    std::vector<int32_t> ids((size_t)3 * (n > 0 ? n : 1));
    int m = 0;
    check(plvs_hip_tsdf_voxblox_block_ids(h_, ids.data(), n, &m));
    std::vector<StoredBlock> out((size_t)n);
    for (int i = 0; i < n; ++i) {
      StoredBlock& b = out[(size_t)i];
      b.id = BlockID(ids[3 * i], ids[3 * i + 1], ids[3 * i + 2]);
      b.distance.resize(4096); b.weight.resize(4096); b.rgba.resize(4096);
      check(plvs_hip_tsdf_voxblox_download_block(h_, ids[3 * i], ids[3 * i + 1], ids[3 * i + 2], b.distance.data(), b.weight.data(),
                                                 b.rgba.data()));
    }
    return out;
  }
  bool LoadLayer(const std::vector<StoredBlock>& blocks) {
    Flush();
    for (const StoredBlock& b : blocks) {
      check(plvs_hip_tsdf_voxblox_upload_block(h_, std::get<0>(b.id), std::get<1>(b.id), std::get<2>(b.id), b.distance.data(),
                                               b.weight.data(), b.rgba.data()));
      updated_.insert(b.id);
    }
    return true;
  }
  const std::vector<PointSurfelSegment>& GetPointCloud() const { return pointCloud_; }
  const std::map<BlockID, VoxbloxMesh>& GetMeshLayer() const { return meshLayer_; }
  plvs_tsdf_voxblox* handle() { return h_; }

 private:
  // every block the integrator touched has updated() set (tsdf_integrator.cc:151) until the next updateMesh
  void MarkUpdated() {
    int nu = 0;
    check(plvs_hip_tsdf_voxblox_updated_block_ids(h_, nullptr, 0, &nu));
    std::vector<int32_t> ids((size_t)3 * (nu > 0 ? nu : 1));
    if (nu > 0) check(plvs_hip_tsdf_voxblox_updated_block_ids(h_, ids.data(), nu, &nu));
    for (int i = 0; i < nu; ++i) updated_.insert(BlockID(ids[3 * i], ids[3 * i + 1], ids[3 * i + 2]));
  }
  // colorVoxbloxToMsg then colorMsgToVoxblox (voxblox_ros/conversions.h:44-60): through a float in [0, 1] and back
  static uint8_t CloudColour(uint8_t c) {
    const float msg = (float)(c / 255.0);
    return (uint8_t)(msg * 255.0);
  }
  plvs_tsdf_voxblox* h_ = nullptr;
  bool merged_ = false;
  bool fast_ = false;
  bool queue_ = false, pending_ = false;
  std::vector<float> xyz_;
  std::vector<uint8_t> rgba_;
  std::set<BlockID> updated_;
  std::map<BlockID, VoxbloxMesh> meshLayer_;
  std::vector<PointSurfelSegment> pointCloud_;
};

}  // namespace PLVS2hip
