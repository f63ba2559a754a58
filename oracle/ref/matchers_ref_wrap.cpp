// C entry points over the reference's OWN matchers and Frame methods — src/ORBmatcher.cc, src/LineMatcher.cc, src/Frame.cc
// compiled unmodified against oracle/ref/slam_shim (see slam_shim.h) — for tests/test_oracle_pinned_matchers.py, which
// holds the restatements in oracle/orb_search.c, line_search.c, line_proj_search.c, stereo.c to them.  TEST
// INFRASTRUCTURE ONLY.  The wrapper builds Frame / KeyFrame / MapPoint / MapLine objects from the same flat views the
// C ABI of the product takes (include/plvs_hip.h: plvs_frame_view, ...) and calls the reference's functions; where the
// product takes a quantity "handed over by the caller" (the projection of a map point into the frame), the wrapper
// takes the poses and points the reference derives it from, and the TEST repeats those two lines of arithmetic.
#define private public      // Frame::AssignFeaturesToGrid / UndistortKeyPoints / UndistortKeyLines / ComputeImageBounds
#define protected public    // ORBmatcher / LineMatcher helpers
#include "Frame.h"
#include "ORBmatcher.h"
#include "LineMatcher.h"
#include "ORBextractor.h"
#undef private
#undef protected

#include <cstring>

#include "plvs_hip.h"

using namespace PLVS2;

// (the LSD detector is not compiled — see lines_ref_wrap.cpp: its entry points named by LineExtractor.cc abort)
namespace cv {
namespace line_descriptor_c {
Ptr<LSDDetectorC> LSDDetectorC::createLSDDetectorC(const LSDOptions&) { std::abort(); }
void LSDDetectorC::detect(const Mat&, std::vector<KeyLine>&, float, int, const LSDOptions&, const Mat&) { std::abort(); }
void LSDDetectorC::setGaussianPyramid(const std::vector<cv::Mat>&, int, float, int, int) { std::abort(); }
}  // namespace line_descriptor_c
}  // namespace cv

// ---- statics the three sources expect from translation units that are not compiled
float PLVS2::Tracking::skLineStereoMaxDist = 20.f;          // src/Tracking.cc:151
float PLVS2::Tracking::skMaxDistFovCenters = 0.5f;          // src/Tracking.cc
bool PLVS2::Tracking::skUsePyramidPrecomputation = false;   // src/Tracking.cc
float PLVS2::KeyFrame::skFovCenterDistance = 1.5f;          // src/KeyFrame.cc
std::mutex PLVS2::MapPoint::mGlobalMutex;
std::mutex PLVS2::MapLine::mGlobalMutex;

namespace {

cv::Mat rows32(const uint8_t* p, int n) { return cv::Mat(n, 32, CV_8U, const_cast<uint8_t*>(p)); }

// What SearchByProjection / SearchByBoW / GetFeaturesInArea read of a single-camera frame.
void fill_frame(Frame& F, const plvs_frame_view* v, const float* angle) {
  F.N = v->n;
  F.Nleft = -1;
  F.mvKeysUn.resize(v->n);
  for (int i = 0; i < v->n; ++i) {
    cv::KeyPoint& k = F.mvKeysUn[i];
    k.pt.x = v->x[i];
    k.pt.y = v->y[i];
    k.octave = v->octave[i];
    k.angle = angle ? angle[i] : 0.f;
  }
  F.mvKeys = F.mvKeysUn;
  F.mvuRight.assign(v->u_right, v->u_right + v->n);
  F.mvDepth.assign(v->n, -1.f);
  F.mDescriptors = rows32(v->desc, v->n);
  F.mvScaleFactors.assign(v->scale_factors, v->scale_factors + 8);
  F.mvpMapPoints.assign(v->n, static_cast<MapPointPtr>(nullptr));
  F.mvbOutlier.assign(v->n, false);
  Frame::mnMinX = v->min_x;
  Frame::mnMinY = v->min_y;
  Frame::mfGridElementWidthInv = v->grid_w_inv;
  Frame::mfGridElementHeightInv = v->grid_h_inv;
  F.AssignFeaturesToGrid();   // (the reference's: Frame.cc:716)
}

}  // namespace

extern "C" {

// ORBmatcher::SearchByProjection(Frame&, const vector<MapPointPtr>&, th, bFarPoints, thFarPoints)  src/ORBmatcher.cc:71
int ref_orb_search_by_projection(const plvs_frame_view* fv, const plvs_mappoint_view* mv, float th, int far_points,
                                 float th_far, float nn_ratio, const uint8_t* occupied, int32_t* assigned) {
  Frame F;
  fill_frame(F, fv, nullptr);
  MapPoint taken;   // what an occupied keypoint holds: a map point with observations
  taken.mnObs = 1;
  if (occupied)
    for (int i = 0; i < fv->n; ++i)
      if (occupied[i]) F.mvpMapPoints[i] = &taken;
  std::vector<MapPoint> store(mv->m);
  std::vector<MapPointPtr> vp(mv->m);
  for (int k = 0; k < mv->m; ++k) {
    MapPoint& p = store[k];
    p.mbTrackInView = mv->track_in_view[k] != 0;
    p.mbTrackInViewR = false;
    p.mbBad = mv->bad[k] != 0;
    p.mTrackProjX = mv->proj_x[k];
    p.mTrackProjY = mv->proj_y[k];
    p.mTrackProjXR = mv->proj_xr[k];
    p.mTrackViewCos = mv->view_cos[k];
    p.mTrackDepth = mv->track_depth[k];
    p.mnTrackScaleLevel = mv->level[k];
    p.mDescriptor = rows32(mv->desc + 32 * (size_t)k, 1);
    p.mnObs = (!mv->has_obs || mv->has_obs[k]) ? 1 : 0;
    vp[k] = &p;
  }
  ORBmatcher matcher(nn_ratio, true);
  const int n = matcher.SearchByProjection(F, vp, th, far_points != 0, th_far);
  for (int i = 0; i < fv->n; ++i) {
    MapPointPtr p = F.mvpMapPoints[i];
    assigned[i] = (p && p != &taken) ? (int32_t)(p - store.data()) : -1;
  }
  return n;
}

// ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono)  src/ORBmatcher.cc:1774.
// Tcw / Tlw: 3 x 4 row-major poses of the two frames; cam = fx, fy, cx, cy of CurrentFrame.mpCamera; xyz_w = world
// position of the map point of last-frame keypoint i (n_last x 3); valid[i] = mvpMapPoints[i] && !mvbOutlier[i].
int ref_orb_search_by_projection_ff(const plvs_frame_view* fv, const float* cur_angle, float max_x, float max_y, float mbf,
                                    float mb, const float* Tcw, const float* Tlw, const float* cam, int n_last,
                                    const uint8_t* valid, const float* xyz_w, const int32_t* octave, const float* angle,
                                    const uint8_t* desc, const uint8_t* has_obs, float th, int mono, float nn_ratio,
                                    int check_orientation, const uint8_t* occupied, int32_t* assigned) {
  Frame C, L;
  fill_frame(C, fv, cur_angle);
  Frame::mnMaxX = max_x;
  Frame::mnMaxY = max_y;
  C.mbf = mbf;
  C.mb = mb;
  Pinhole camera(std::vector<float>(cam, cam + 4));
  C.mpCamera = &camera;
  auto pose_of = [](const float* T) {
    Eigen::Matrix3f R;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) R(r, c) = T[4 * r + c];
    return Sophus::SE3f(R, Eigen::Vector3f(T[3], T[7], T[11]));
  };
  C.SetPose(pose_of(Tcw));
  L.SetPose(pose_of(Tlw));
  MapPoint taken;
  taken.mnObs = 1;
  if (occupied)
    for (int i = 0; i < fv->n; ++i)
      if (occupied[i]) C.mvpMapPoints[i] = &taken;
  L.N = n_last;
  L.Nleft = -1;
  L.mvKeys.resize(n_last);
  for (int i = 0; i < n_last; ++i) {
    L.mvKeys[i].octave = octave[i];
    L.mvKeys[i].angle = angle[i];
  }
  L.mvKeysUn = L.mvKeys;
  L.mvbOutlier.assign(n_last, false);
  std::vector<MapPoint> store(n_last);
  L.mvpMapPoints.assign(n_last, static_cast<MapPointPtr>(nullptr));
  for (int i = 0; i < n_last; ++i) {
    if (!valid[i]) continue;
    MapPoint& p = store[i];
    p.mWorldPos = Eigen::Vector3f(xyz_w[3 * i], xyz_w[3 * i + 1], xyz_w[3 * i + 2]);
    p.mDescriptor = rows32(desc + 32 * (size_t)i, 1);
    p.mnObs = (!has_obs || has_obs[i]) ? 1 : 0;
    L.mvpMapPoints[i] = &p;
  }
  ORBmatcher matcher(nn_ratio, check_orientation != 0);
  const int n = matcher.SearchByProjection(C, L, th, mono != 0);
  for (int i = 0; i < fv->n; ++i) {
    MapPointPtr p = C.mvpMapPoints[i];
    assigned[i] = (p && p != &taken) ? (int32_t)(p - store.data()) : -1;
  }
  return n;
}

// ORBmatcher::SearchByBoW(KeyFramePtr& pKF, Frame& F, vector<MapPointPtr>&)  src/ORBmatcher.cc:300
int ref_orb_search_by_bow(const plvs_featvec_view* kf_vec, const uint8_t* kf_desc, int kf_n, const uint8_t* kf_valid,
                          const float* kf_angle, const plvs_featvec_view* f_vec, const uint8_t* f_desc, int f_n,
                          const float* f_angle, float nn_ratio, int check_orientation, int32_t* assigned) {
  auto fill_vec = [](DBoW2::FeatureVector& fvv, const plvs_featvec_view* v) {
    for (int a = 0; a < v->nnodes; ++a)
      for (int j = v->offset[a]; j < v->offset[a + 1]; ++j) fvv.addFeature(v->node_id[a], v->index[j]);
  };
  KeyFrame K;
  K.N = kf_n;
  K.NLeft = -1;
  K.mDescriptors = rows32(kf_desc, kf_n);
  K.mvKeysUn.resize(kf_n);
  for (int i = 0; i < kf_n; ++i) K.mvKeysUn[i].angle = kf_angle[i];
  K.mvKeys = K.mvKeysUn;
  fill_vec(K.mFeatVec, kf_vec);
  std::vector<MapPoint> store(kf_n);
  K.mvpMapPoints.assign(kf_n, static_cast<MapPointPtr>(nullptr));
  for (int i = 0; i < kf_n; ++i)
    if (kf_valid[i]) K.mvpMapPoints[i] = &store[i];
  Frame F;
  F.N = f_n;
  F.Nleft = -1;
  F.mDescriptors = rows32(f_desc, f_n);
  F.mvKeys.resize(f_n);
  for (int i = 0; i < f_n; ++i) F.mvKeys[i].angle = f_angle[i];
  F.mvKeysUn = F.mvKeys;
  fill_vec(F.mFeatVec, f_vec);
  std::vector<MapPointPtr> matches;
  KeyFramePtr pKF = &K;
  ORBmatcher matcher(nn_ratio, check_orientation != 0);
  const int n = matcher.SearchByBoW(pKF, F, matches);
  for (int i = 0; i < f_n; ++i) assigned[i] = matches[i] ? (int32_t)(matches[i] - store.data()) : -1;
  return n;
}

// ---- LineMatcher::SearchByKnn (three flavours): descriptors, angles (radians), octaves as flat arrays
static void fill_keylines(std::vector<cv::line_descriptor_c::KeyLine>& v, int n, const float* angle, const int32_t* octave) {
  v.resize(n);
  for (int i = 0; i < n; ++i) {
    std::memset(&v[i], 0, sizeof v[i]);
    v[i].angle = angle ? angle[i] : 0.f;
    v[i].octave = octave ? octave[i] : 0;
    v[i].class_id = i;
  }
}

// LineMatcher::SearchByKnn(Frame& CurrentFrame, const Frame& LastFrame)  src/LineMatcher.cc:303
int ref_lines_search_by_knn(const uint8_t* last_desc, int n_last, const uint8_t* valid, const float* ang_last,
                            const uint8_t* cur_desc, int n_cur, const float* ang_cur, float nn_ratio, int check_orientation,
                            int32_t* assigned) {
  Frame C, L;
  L.Nlines = n_last;
  L.mLineDescriptors = rows32(last_desc, n_last);
  fill_keylines(L.mvKeyLinesUn, n_last, ang_last, nullptr);
  L.mvbLineOutlier.assign(n_last, false);
  std::vector<MapLine> store(n_last);
  L.mvpMapLines.assign(n_last, static_cast<MapLinePtr>(nullptr));
  for (int i = 0; i < n_last; ++i)
    if (valid[i]) L.mvpMapLines[i] = &store[i];
  C.Nlines = n_cur;
  C.mLineDescriptors = rows32(cur_desc, n_cur);
  fill_keylines(C.mvKeyLinesUn, n_cur, ang_cur, nullptr);
  C.mvpMapLines.assign(n_cur, static_cast<MapLinePtr>(nullptr));
  LineMatcher matcher(nn_ratio, false, check_orientation != 0);
  const int n = matcher.SearchByKnn(C, L);
  for (int i = 0; i < n_cur; ++i) assigned[i] = C.mvpMapLines[i] ? (int32_t)(C.mvpMapLines[i] - store.data()) : -1;
  return n;
}

// LineMatcher::SearchByKnn(KeyFramePtr& pKF, const Frame& F, vector<MapLinePtr>&)  src/LineMatcher.cc:156
int ref_lines_search_by_knn_kf(const uint8_t* kf_desc, int n_kf, const uint8_t* valid, const float* ang_kf,
                               const uint8_t* f_desc, int n_f, const float* ang_f, float nn_ratio, int check_orientation,
                               int32_t* assigned) {
  KeyFrame K;
  K.Nlines = n_kf;
  K.mLineDescriptors = rows32(kf_desc, n_kf);
  fill_keylines(K.mvKeyLinesUn, n_kf, ang_kf, nullptr);
  std::vector<MapLine> store(n_kf);
  K.mvpMapLines.assign(n_kf, static_cast<MapLinePtr>(nullptr));
  for (int i = 0; i < n_kf; ++i)
    if (valid[i]) K.mvpMapLines[i] = &store[i];
  Frame F;
  F.Nlines = n_f;
  F.mLineDescriptors = rows32(f_desc, n_f);
  fill_keylines(F.mvKeyLinesUn, n_f, ang_f, nullptr);
  std::vector<MapLinePtr> matches;
  KeyFramePtr pKF = &K;
  LineMatcher matcher(nn_ratio, false, check_orientation != 0);
  const int n = matcher.SearchByKnn(pKF, F, matches);
  for (int i = 0; i < n_f; ++i) assigned[i] = ((int)matches.size() > i && matches[i]) ? (int32_t)(matches[i] - store.data()) : -1;
  return n;
}

// LineMatcher::SearchStereoMatchesByKnn(Frame&, vector<DMatch>&, vector<bool>&, descriptorDist)  src/LineMatcher.cc:454
int ref_lines_search_stereo_by_knn(const uint8_t* left_desc, int n_left, const float* ang_left, const int32_t* oct_left,
                                   const uint8_t* right_desc, int n_right, const float* ang_right, const int32_t* oct_right,
                                   float nn_ratio, int check_orientation, int descriptor_dist, int32_t* match_query,
                                   int32_t* match_train, float* match_distance, uint8_t* match_valid, int* n_out) {
  Frame F;
  F.Nlines = n_left;
  F.mLineDescriptors = rows32(left_desc, n_left);
  F.mLineDescriptorsRight = rows32(right_desc, n_right);
  fill_keylines(F.mvKeyLinesUn, n_left, ang_left, oct_left);
  F.mvKeyLines = F.mvKeyLinesUn;
  fill_keylines(F.mvKeyLinesRightUn, n_right, ang_right, oct_right);
  F.mvKeyLinesRight = F.mvKeyLinesRightUn;
  std::vector<cv::DMatch> m;
  std::vector<bool> v;
  LineMatcher matcher(nn_ratio, false, check_orientation != 0);
  const int n = matcher.SearchStereoMatchesByKnn(F, m, v, descriptor_dist);
  *n_out = (int)m.size();
  for (size_t i = 0; i < m.size(); ++i) {
    match_query[i] = m[i].queryIdx;
    match_train[i] = m[i].trainIdx;
    match_distance[i] = m[i].distance;
    match_valid[i] = v[i] ? 1 : 0;
  }
  return n;
}

// ---- LineMatcher::SearchByProjection (two flavours).  What they read of a single-camera frame: the undistorted key
// lines, their descriptors, the right-image abscissae of the end points, mbf, the line scale tables and the (theta, d)
// grid — filled by the reference's own Frame::AssignFeaturesToGrid.
static void fill_line_frame(Frame& F, const plvs_line_frame_view* v) {
  F.N = 0;
  F.Nleft = -1;
  F.Nlines = v->n;
  F.NlinesLeft = -1;
  F.mvKeyLinesUn.resize(v->n);
  static_assert(sizeof(plvs_keyline) == sizeof(cv::line_descriptor_c::KeyLine), "plvs_keyline is KeyLine field for field");
  if (v->n) std::memcpy(F.mvKeyLinesUn.data(), v->keylines_un, sizeof(plvs_keyline) * (size_t)v->n);
  F.mvKeyLines = F.mvKeyLinesUn;
  F.mLineDescriptors = rows32(v->descriptors, v->n);
  if (v->u_right_start) {
    F.mvuRightLineStart.assign(v->u_right_start, v->u_right_start + v->n);
    F.mvuRightLineEnd.assign(v->u_right_end, v->u_right_end + v->n);
  }
  F.mbf = v->bf;
  F.mvLineScaleFactors.assign(v->line_scale_factors, v->line_scale_factors + v->n_levels);
  F.mvLineInvLevelSigma2.assign(v->line_inv_level_sigma2, v->line_inv_level_sigma2 + v->n_levels);
  F.mvpMapLines.assign(v->n, static_cast<MapLinePtr>(nullptr));
  F.mvbLineOutlier.assign(v->n, false);
  // the line grid's geometry as Frame::Frame sets it on the first frame (src/Frame.cc:446-452 with LINE_THETA_SPAN etc.)
  Frame::mnMaxDiag = v->max_diag;
  Frame::mfLineGridElementThetaInv = static_cast<float>(LINE_THETA_GRID_ROWS) / static_cast<float>(LINE_THETA_SPAN);
  Frame::mfLineGridElementDInv = static_cast<float>(LINE_D_GRID_COLS) / (2.0f * Frame::mnMaxDiag);
  // (AssignFeaturesToGrid fills the line grid only for a frame with a line extractor: a non-null pointer it never follows)
  F.mpLineExtractorLeft = std::shared_ptr<LineExtractor>(std::shared_ptr<int>(), reinterpret_cast<LineExtractor*>(16));
  F.AssignFeaturesToGrid();
  F.mpLineExtractorLeft.reset();
}

// LineMatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, bLargerSearch, bMono)  src/LineMatcher.cc:837.
// xyz_w = world end points of the map line of last-frame line i (n_last x 6: start, end); bounds = mnMinX, mnMaxX, mnMinY,
// mnMaxY of the current frame; the other arguments as ref_orb_search_by_projection_ff.
int ref_lines_search_by_projection_ff(const plvs_line_frame_view* fv, const uint8_t* occupied, const float* bounds, float mb,
                                      const float* Tcw, const float* Tlw, const float* cam, int n_last, const uint8_t* valid,
                                      const float* xyz_w, const int32_t* octave, const float* angle, const uint8_t* desc,
                                      const uint8_t* has_obs, int larger_search, int mono, float nn_ratio,
                                      int check_orientation, int32_t* assigned) {
  Frame C, L;
  fill_line_frame(C, fv);
  Frame::mnMinX = bounds[0]; Frame::mnMaxX = bounds[1]; Frame::mnMinY = bounds[2]; Frame::mnMaxY = bounds[3];
  C.mb = mb;
  Pinhole camera(std::vector<float>(cam, cam + 4));
  C.mpCamera = &camera;
  auto pose_of = [](const float* T) {
    Eigen::Matrix3f R;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) R(r, c) = T[4 * r + c];
    return Sophus::SE3f(R, Eigen::Vector3f(T[3], T[7], T[11]));
  };
  C.SetPose(pose_of(Tcw));
  L.SetPose(pose_of(Tlw));
  MapLine taken;
  taken.mnObs = 1;
  if (occupied)
    for (int i = 0; i < fv->n; ++i)
      if (occupied[i]) C.mvpMapLines[i] = &taken;
  L.Nlines = n_last;
  L.NlinesLeft = -1;
  fill_keylines(L.mvKeyLinesUn, n_last, angle, octave);
  L.mvbLineOutlier.assign(n_last, false);
  std::vector<MapLine> store(n_last);
  L.mvpMapLines.assign(n_last, static_cast<MapLinePtr>(nullptr));
  for (int i = 0; i < n_last; ++i) {
    if (!valid[i]) continue;
    MapLine& m = store[i];
    m.mWorldPosStart = Eigen::Vector3f(xyz_w[6 * i], xyz_w[6 * i + 1], xyz_w[6 * i + 2]);
    m.mWorldPosEnd = Eigen::Vector3f(xyz_w[6 * i + 3], xyz_w[6 * i + 4], xyz_w[6 * i + 5]);
    m.mDescriptor = rows32(desc + 32 * (size_t)i, 1);
    m.mnObs = (!has_obs || has_obs[i]) ? 1 : 0;
    L.mvpMapLines[i] = &m;
  }
  LineMatcher matcher(nn_ratio, false, check_orientation != 0);
  const int n = matcher.SearchByProjection(C, L, larger_search != 0, mono != 0);
  for (int i = 0; i < fv->n; ++i) {
    MapLinePtr p = C.mvpMapLines[i];
    assigned[i] = (p && p != &taken) ? (int32_t)(p - store.data()) : -1;
  }
  return n;
}

// LineMatcher::SearchByProjection(Frame& F, const vector<MapLinePtr>&, bLargerSearch)  src/LineMatcher.cc:1286.
// proj[6 m ..] = mTrackProjStartX, StartY, EndX, EndY, mTrackStartDepth, mTrackEndDepth.
int ref_lines_search_by_projection(const plvs_line_frame_view* fv, const uint8_t* occupied, int n_map, const uint8_t* in_view,
                                   const float* proj, const int32_t* level, const uint8_t* desc, const uint8_t* has_obs,
                                   int larger_search, float nn_ratio, int32_t* assigned) {
  Frame F;
  fill_line_frame(F, fv);
  MapLine taken;
  taken.mnObs = 1;
  if (occupied)
    for (int i = 0; i < fv->n; ++i)
      if (occupied[i]) F.mvpMapLines[i] = &taken;
  std::vector<MapLine> store(n_map);
  std::vector<MapLinePtr> vp(n_map);
  for (int m = 0; m < n_map; ++m) {
    MapLine& l = store[m];
    l.mbTrackInView = in_view[m] != 0;
    l.mbTrackInViewR = false;
    l.mTrackProjStartX = proj[6 * m];
    l.mTrackProjStartY = proj[6 * m + 1];
    l.mTrackProjEndX = proj[6 * m + 2];
    l.mTrackProjEndY = proj[6 * m + 3];
    l.mTrackStartDepth = proj[6 * m + 4];
    l.mTrackEndDepth = proj[6 * m + 5];
    l.mnTrackScaleLevel = level[m];
    l.mDescriptor = rows32(desc + 32 * (size_t)m, 1);
    l.mnObs = (!has_obs || has_obs[m]) ? 1 : 0;
    vp[m] = &l;
  }
  LineMatcher matcher(nn_ratio, false, true);
  const int n = matcher.SearchByProjection(F, vp, larger_search != 0);
  for (int i = 0; i < fv->n; ++i) {
    MapLinePtr p = F.mvpMapLines[i];
    assigned[i] = (p && p != &taken) ? (int32_t)(p - store.data()) : -1;
  }
  return n;
}

// Frame::ComputeStereoMatches  src/Frame.cc:1780-1990 (CPU branch) on a rectified pair: the reference's own
// ORBextractor runs on both images (Frame::ExtractORB, :806-813) — its pyramids are what the function reads — and the
// frame takes its key points and descriptors as Frame::Frame does (:331-343).  mb = mbf / fx.  Outputs: mvuRight, mvDepth
// (n_left entries, n_left returned); *n_right = right key points.
int ref_frame_compute_stereo_matches(const uint8_t* left, const uint8_t* right, int w, int h, int stride, int nfeatures,
                                     float scale_factor, int nlevels, int ini_th, int min_th, float mb, float mbf,
                                     float* u_right, float* depth, int cap, int* n_right) {
  ORBextractor exl(nfeatures, scale_factor, nlevels, ini_th, min_th), exr(nfeatures, scale_factor, nlevels, ini_th, min_th);
  Frame F;
  F.mpORBextractorLeft = &exl;
  F.mpORBextractorRight = &exr;
  std::vector<int> lap = {0, 0};
  cv::Mat il(h, w, CV_8UC1, const_cast<uint8_t*>(left), (size_t)stride), ir(h, w, CV_8UC1, const_cast<uint8_t*>(right), (size_t)stride);
  F.monoLeft = exl(il, cv::Mat(), F.mvKeys, F.mDescriptors, lap);
  F.monoRight = exr(ir, cv::Mat(), F.mvKeysRight, F.mDescriptorsRight, lap);
  F.N = (int)F.mvKeys.size();
  F.mvKeysUn = F.mvKeys;   // (rectified: no distortion)
  F.mnScaleLevels = exl.GetLevels();
  F.mvScaleFactors = exl.GetScaleFactors();
  F.mvInvScaleFactors = exl.GetInverseScaleFactors();
  F.mb = mb;
  F.mbf = mbf;
  *n_right = (int)F.mvKeysRight.size();
  if (F.N > cap) return -F.N;
  F.ComputeStereoMatches();
  for (int i = 0; i < F.N; ++i) {
    u_right[i] = F.mvuRight[i];
    depth[i] = F.mvDepth[i];
  }
  return F.N;
}

// ---- the glue between extraction and the searches: Frame::UndistortKeyPoints (:1507), ComputeImageBounds (:1749),
// UndistortKeyLines (:1555), AssignFeaturesToGrid (:716) — the reference's own, on a frame filled from flat arrays.
// K4 = fx, fy, cx, cy; dist = mDistCoef (4, 5 or 8 coefficients; ndist = 0: a zero 4-vector).
static void set_calib(Frame& F, Pinhole& cam, const float* dist, int ndist) {
  F.mpCamera = &cam;
  F.mDistCoef = cv::Mat(ndist > 0 ? ndist : 4, 1, CV_32F);
  for (int i = 0; i < F.mDistCoef.rows; ++i) F.mDistCoef.at<float>(i) = i < ndist ? dist[i] : 0.0f;
}

void ref_frame_undistort_keypoints(const plvs_keypoint* kps, int n, const float* K4, const float* dist, int ndist,
                                   plvs_keypoint* un) {
  Frame F;
  Pinhole cam(std::vector<float>(K4, K4 + 4));
  set_calib(F, cam, dist, ndist);
  F.N = n;
  F.mvKeys.resize(n);
  for (int i = 0; i < n; ++i)
    F.mvKeys[i] = cv::KeyPoint(kps[i].x, kps[i].y, kps[i].size, kps[i].angle, kps[i].response, kps[i].octave, kps[i].class_id);
  F.UndistortKeyPoints();
  for (int i = 0; i < n; ++i) {
    const cv::KeyPoint& k = F.mvKeysUn[i];
    un[i] = plvs_keypoint{k.pt.x, k.pt.y, k.size, k.angle, k.response, k.octave, k.class_id};
  }
}

void ref_frame_compute_image_bounds(int width, int height, const float* K4, const float* dist, int ndist, float* bounds5) {
  Frame F;
  Pinhole cam(std::vector<float>(K4, K4 + 4));
  set_calib(F, cam, dist, ndist);
  cv::Mat im(height, width, CV_8UC1);
  F.ComputeImageBounds(im);
  bounds5[0] = Frame::mnMinX; bounds5[1] = Frame::mnMaxX; bounds5[2] = Frame::mnMinY; bounds5[3] = Frame::mnMaxY;
  bounds5[4] = Frame::mnMaxDiag;
}

int ref_frame_undistort_keylines(const plvs_keyline* kl, int n, const float* K4, const float* dist, int ndist, const float* bounds4,
                                 plvs_keyline* un, int32_t* kept_index) {
  Frame F;
  Pinhole cam(std::vector<float>(K4, K4 + 4));
  set_calib(F, cam, dist, ndist);
  Frame::mnMinX = bounds4[0]; Frame::mnMaxX = bounds4[1]; Frame::mnMinY = bounds4[2]; Frame::mnMaxY = bounds4[3];
  F.Nlines = n;
  F.NlinesLeft = -1;
  F.NlinesRight = -1;
  F.mvKeyLines.resize(n);
  if (n) std::memcpy(F.mvKeyLines.data(), kl, sizeof(plvs_keyline) * (size_t)n);
  for (int i = 0; i < n; ++i) F.mvKeyLines[i].class_id = i;   // (to report which lines stayed)
  std::vector<uint8_t> d(32 * (size_t)std::max(n, 1), 0);
  F.mLineDescriptors = rows32(d.data(), n);
  F.UndistortKeyLines();
  const int m = (int)F.mvKeyLinesUn.size();
  for (int i = 0; i < m; ++i) {
    kept_index[i] = F.mvKeyLinesUn[i].class_id;
    std::memcpy(&un[i], &F.mvKeyLinesUn[i], sizeof(plvs_keyline));
    un[i].class_id = kl[kept_index[i]].class_id;
  }
  return m;
}

int ref_frame_assign_features_to_grid(const plvs_keypoint* un, int n, float min_x, float min_y, float inv_w, float inv_h,
                                      int32_t* cell_start, int32_t* cell_items) {
  Frame F;
  F.N = n;
  F.Nleft = -1;
  F.mvKeysUn.resize(n);
  for (int i = 0; i < n; ++i) F.mvKeysUn[i].pt = cv::Point2f(un[i].x, un[i].y);
  Frame::mnMinX = min_x; Frame::mnMinY = min_y;
  Frame::mfGridElementWidthInv = inv_w; Frame::mfGridElementHeightInv = inv_h;
  F.AssignFeaturesToGrid();
  int at = 0;
  for (int ix = 0; ix < FRAME_GRID_COLS; ++ix)
    for (int iy = 0; iy < FRAME_GRID_ROWS; ++iy) {
      cell_start[ix * FRAME_GRID_ROWS + iy] = at;
      for (size_t v : F.mGrid[ix][iy]) cell_items[at++] = (int32_t)v;
    }
  cell_start[FRAME_GRID_COLS * FRAME_GRID_ROWS] = at;
  return at;
}

}  // extern "C"
