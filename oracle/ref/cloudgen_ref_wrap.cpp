// TEST INFRASTRUCTURE ONLY — the reference's depth image -> cloud step, compiled from where it lies, as the pin of
// oracle/cloudgen.c.
//
//   PointCloudMapping::InitCamGridPoints                      /root/reference/src/PointCloudMapping.cc:796-905
//   PointCloudMapping::GeneratePointCloudInCameraFrameBGRA    /root/reference/src/PointCloudMapping.cc:929-1226
//
// src/PointCloudMapping.cc as a whole is the system's point-cloud thread (constructor from the settings file, Run(), the
// key-frame queues, every map back end, the Pangolin images): it cannot be compiled here.  The two functions on the hot
// path are taken from it VERBATIM at build time — oracle/ref/Makefile cuts the two definitions out of the file where it
// lies (from the line the definition starts on to its closing brace) into oracle/_ref/gen_cloudgen_ref.inc, a build
// product like the .so (git-ignored, never part of the tree) — and compiled inside the stand-in class below, which
// declares exactly the members the two functions touch, with the reference's names and types
// (include/PointCloudMapping.h:79,194-216; include/PointCloudKeyFrame.h:73-91).  The reference's own headers are used
// where they compile: include/PointSurfelSegment.h (the point type, against oracle/ref/pcl_shim/), include/PointDefinitions.h
// is replaced by its `USE_POINTSURFELSEGMENT 1` branch with COMPUTE_SEGMENTS 0 (the segmentation block, :1037-1219, needs
// OpenCV's drawing / morphology / connected components and runs only with Segmentation.on 1: every shipped YAML has 0),
// include/PointCloudMapTypes.h (PointCloudMapParameters), include/Neighborhood.h (the neighbour table).
// Stand-ins: cv::Mat / Mat_ / Size and cv::undistortPoints (oracle/ref/cv_full/, slam_shim/cv_more.hpp: OpenCV's published
// algorithm, as for src/Frame.cc), Eigen (oracle/ref/eigen_full/), pcl::PointCloud (pcl_shim/), KeyFrame (the six members
// read), MSG_ASSERT / TICKCLOUD / TOCKCLOUD (no-ops: diagnostics), PointUtils::setKFid / updateDepth
// (include/PointUtils.h:43-46, 361-364: the overloads a point type WITH those fields selects), TimeUtils::getTimestampfromSec
// (include/TimeUtils.h:44-49), Settings (rectification off: PointCloudKeyFrame<PointT>::skbNeedRectification = false, its
// default, src/PointCloudKeyFrame.cc:50).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include <opencv2/core/core.hpp>
#include "slam_shim/cv_more.hpp"   // cv::undistortPoints
#include <Eigen/Core>

namespace cv {
struct Vec2f {
  float v[2];
  float operator[](int i) const { return v[i]; }
  float& operator[](int i) { return v[i]; }
};
template <class M>
inline void cv2eigen(const Mat& src, M& dst) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) dst(r, c) = src.at<float>(r, c);
}
}  // namespace cv

// ---- the reference's own headers that compile here
#define POINT_TYPE_DEFINITIONS_H                 // include/PointDefinitions.h, the branch the tree selects (:28-29, 50-57) ...
#include "PointSurfelSegment.h"                  // (the reference's file, against pcl_shim/)
#define USE_NORMALS 1
#define USE_POINTSURFELSEGMENT 1
#define POINT_TYPE pcl::PointSurfelSegment
#define COMPUTE_NORMALS (1 && USE_NORMALS)
#define COMPUTE_SEGMENTS 0                       // ... with the segmentation block compiled out (see above)
#include "PointCloudMapTypes.h"
#include "Neighborhood.h"

#define MSG_ASSERT(condition, message) do { } while (0)
#define TICKCLOUD(name) do { } while (0)
#define TOCKCLOUD(name) do { } while (0)

using namespace std;

namespace PLVS2 {

typedef POINT_TYPE PointT;
typedef EigthNeighborhoodIndicesFast NeighborhoodT;   // src/PointCloudMapping.cc:100

struct KeyFrame {            // what the two functions read of a key frame
  long unsigned int mnId = 0;
  float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0;
  cv::Mat mK, mDistCoef;
  double mTimeStamp = 0.0;
};
typedef std::shared_ptr<KeyFrame> KeyFramePtr;

struct PointCloudCamParams {   // include/PointCloudKeyFrame.h:73-91
  double fx, fy, cx, cy, bf;
  int width, height;
  double minDist, maxDist;
  cv::Mat mDistCoef, mK;
};
template <typename P>
struct PointCloudKeyFrame {
  static bool skbNeedRectification;
};
template <typename P>
bool PointCloudKeyFrame<P>::skbNeedRectification = false;   // src/PointCloudKeyFrame.cc:50

struct Settings {
  static const Settings* instance() { static Settings s; return &s; }
  cv::Mat R_r1_u1() const { std::abort(); }                 // (rectification is off)
};

namespace PointUtils {
template <class P> inline void updateDepth(const float& depth, P& point) { point.depth = depth; }   // include/PointUtils.h:43-46
template <class P> inline void setKFid(P& point, const int& kfid) { point.kfid = kfid; }            // :361-364
}  // namespace PointUtils
struct TimeUtils {
  static std::uint64_t getTimestampfromSec(double t) {      // include/TimeUtils.h:44-49
    uint32_t sec = (uint32_t)floor(t);
    uint32_t usec = (uint32_t)round((t - sec) * 1e6);
    return sec * 1000000 + usec;
  }
};

struct Image4Viewer {        // include/PointCloudMapping.h:60-67
  Image4Viewer() : bReady(false) {}
  std::string name;
  bool bReady;
  cv::Mat img;
};

class PointCloudMapping {    // the members of include/PointCloudMapping.h the two functions touch
 public:
  typedef pcl::PointCloud<PointT> PointCloudT;
  static int skDownsampleStep;                                              // :79
  void InitCamGridPoints(const KeyFramePtr& kf, const cv::Size& depthSize);
  PointCloudT::Ptr GeneratePointCloudInCameraFrameBGRA(KeyFramePtr& kf, cv::Mat& color, cv::Mat& depth, cv::Mat& pixelToPointIndex,
                                                       std::vector<unsigned int>& segmentsCardinality);
  void FilterDepthimage(cv::Mat&, int, double, double) { std::abort(); }   // (filterDepth.on 0 in every shipped YAML)
  std::shared_ptr<PointCloudCamParams> pCameraParams_;                      // :194
  std::shared_ptr<PointCloudMapParameters> pPointCloudMapParameters_;      // :199
  bool bInitCamGridPoints_ = false;
  cv::Mat matCamGridPoints_;                                                // :208
  std::vector<std::vector<int> > vCamGridPointsNeighborsIdxs_;              // :210
  std::vector<Image4Viewer> vecImages_;                                     // :216
};
int PointCloudMapping::skDownsampleStep = 2;                                // src/PointCloudMapping.cc:70

// ---- the two definitions, verbatim from /root/reference/src/PointCloudMapping.cc (cut out by oracle/ref/Makefile)
#include "gen_cloudgen_ref.inc"

}  // namespace PLVS2

// ------------------------------------------------------------------ C interface for the tests
struct RefCloudgen {
  PLVS2::PointCloudMapping m;
  PLVS2::KeyFramePtr kf;
};

extern "C" {

// K: 3x3 f32 row-major; dist: ndist (4, 5 or 8) f32 coefficients (dist[0] == 0: no undistortion, as the reference tests it)
void* ref_cloudgen_create(const float* K, const float* dist, int ndist, double bf, int step, double min_depth, double max_depth) {
  RefCloudgen* h = new RefCloudgen();
  PLVS2::PointCloudMapping::skDownsampleStep = step;
  h->m.pCameraParams_ = std::make_shared<PLVS2::PointCloudCamParams>();
  auto& c = *h->m.pCameraParams_;
  c.mK = cv::Mat(3, 3, CV_32F);
  std::memcpy(c.mK.ptr(0), K, 9 * sizeof(float));
  c.mDistCoef = cv::Mat(ndist, 1, CV_32F);
  for (int i = 0; i < ndist; ++i) c.mDistCoef.at<float>(i) = dist[i];
  // src/PointCloudMapping.cc:177-181: the doubles are the FLOAT entries of K
  c.fx = c.mK.at<float>(0, 0);
  c.fy = c.mK.at<float>(1, 1);
  c.cx = c.mK.at<float>(0, 2);
  c.cy = c.mK.at<float>(1, 2);
  c.bf = bf;
  c.minDist = min_depth;
  c.maxDist = max_depth;
  h->m.pPointCloudMapParameters_ = std::make_shared<PLVS2::PointCloudMapParameters>();
  auto& p = *h->m.pPointCloudMapParameters_;
  p.minDepthDistance = min_depth;
  p.maxDepthDistance = max_depth;
  p.bFilterDepthImages = false;
  p.bSegmentationOn = false;
  h->m.vecImages_.resize(8);
  h->kf = std::make_shared<PLVS2::KeyFrame>();
  h->kf->mK = c.mK;
  h->kf->mDistCoef = c.mDistCoef;
  return h;
}
void ref_cloudgen_destroy(void* hv) { delete static_cast<RefCloudgen*>(hv); }

int ref_cloudgen_point_size() { return (int)sizeof(PLVS2::PointT); }

// InitCamGridPoints alone: grid = N x 2 floats (N as the reference sizes it); returns the rows the loops filled.
int ref_cloudgen_grid(void* hv, int width, int height, float* grid, int grid_rows) {
  RefCloudgen* h = static_cast<RefCloudgen*>(hv);
  h->m.bInitCamGridPoints_ = false;
  h->m.InitCamGridPoints(h->kf, cv::Size(width, height));
  const cv::Mat& g = h->m.matCamGridPoints_;
  const int step = PLVS2::PointCloudMapping::skDownsampleStep;
  const int filled = ((width + step - 1) / step) * ((height + step - 1) / step);
  for (int i = 0; i < filled && i < grid_rows && i < g.rows; ++i) {
    grid[2 * i] = g.at<float>(i, 0);
    grid[2 * i + 1] = g.at<float>(i, 1);
  }
  return filled;
}

// One key frame: depth (height x width f32, pitch in floats), colour (height x width x 3 u8, pitch in bytes) -> the cloud's
// points byte for byte (sizeof(PointT) each; capacity in points) and pixelToPointIndex (height x width int32).  Returns the
// number of points.  The function reads the colour image as rows of uchar (color.ptr<uchar>(m)[3 n + k]): a CV_8U matrix
// of 3 width columns carries it.
int ref_cloudgen_generate(void* hv, const float* depth, int depth_pitch, const uint8_t* bgr, int bgr_pitch, int width, int height,
                          unsigned long kfid, double timestamp, void* out_points, int capacity, int32_t* pixel_to_point,
                          uint64_t* stamp) {
  RefCloudgen* h = static_cast<RefCloudgen*>(hv);
  cv::Mat d(height, width, CV_32F, const_cast<float*>(depth), (size_t)depth_pitch * sizeof(float));
  cv::Mat c(height, 3 * width, CV_8U, const_cast<uint8_t*>(bgr), (size_t)bgr_pitch);
  cv::Mat p2p;
  std::vector<unsigned int> card;
  h->kf->mnId = kfid;
  h->kf->mTimeStamp = timestamp;
  auto cloud = h->m.GeneratePointCloudInCameraFrameBGRA(h->kf, c, d, p2p, card);
  const int n = (int)cloud->points.size();
  if (n <= capacity && n > 0) std::memcpy(out_points, cloud->points.data(), (size_t)n * sizeof(PLVS2::PointT));
  if (pixel_to_point)
    for (int r = 0; r < height; ++r) std::memcpy(pixel_to_point + (size_t)r * width, p2p.ptr<int>(r), (size_t)width * sizeof(int32_t));
  if (stamp) *stamp = cloud->header.stamp;
  return n;
}

}  // extern "C"
