// Second half of slam_shim.h: the map-side classes (KeyFrame, MapPoint, MapLine, MapObject, Map, Tracking) as data
// holders.  See slam_shim.h.  TEST INFRASTRUCTURE ONLY.
#pragma once
#ifndef UNUSED_VAR
#define UNUSED_VAR(x) (void)x
#endif
#include "Thirdparty/DBoW2/DBoW2/BowVector.h"
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
#include "LineExtractor.h"   // (the reference's: cv::line_descriptor_c::KeyLine)
#include "Pointers.h"        // (the reference's: MapPointPtr = MapPoint*, ...)

namespace PLVS2 {

class Frame;
class Map {};
class Atlas {};

class Stopwatch {   // Stopwatch.h: timing only
 public:
  void start(const char* = nullptr) {}
  void stop(const char* = nullptr) {}
  void tic() {}
  void toc() {}
  double elapsedMs() const { return 0.0; }
};
#define TICKCLOUD(name)
#define TOCKCLOUD(name)
#define TICKTRACK(name)
#define TOCKTRACK(name)

class Tracking {
 public:
  static float skLineStereoMaxDist;
  static float skMaxDistFovCenters;
  static bool skUsePyramidPrecomputation;
  // (names of the member functions Frame.cc / the matchers mention in log messages only)
};

class MapPoint {
 public:
  // ---- what the test sets
  Eigen::Vector3f mWorldPos;
  cv::Mat mDescriptor;                         // 1 x 32 CV_8U
  bool mbBad = false;
  int mnObs = 1;
  float mfMinDistance = 0.f, mfMaxDistance = 1e9f;
  Eigen::Vector3f mNormalVector;
  long unsigned int mnId = 0;
  std::map<KeyFramePtr, std::tuple<int, int>> mObservations;
  // ---- tracking variables (MapPoint.h:158-176): public members the matchers read and write
  float mTrackProjX = 0, mTrackProjY = 0, mTrackDepth = 0, mTrackDepthR = 0, mTrackProjXR = 0, mTrackProjYR = 0;
  bool mbTrackInView = false, mbTrackInViewR = false;
  int mnTrackScaleLevel = 0, mnTrackScaleLevelR = 0;
  float mTrackViewCos = 1.f, mTrackViewCosR = 1.f;
  long unsigned int mnTrackReferenceForFrame = 0, mnLastFrameSeen = 0;
  long unsigned int mnBALocalForKF = 0, mnFuseCandidateForKF = 0, mnLoopPointForKF = 0, mnCorrectedByKF = 0,
                    mnCorrectedReference = 0;
  static std::mutex mGlobalMutex;

  bool isBad() const { return mbBad; }
  int Observations() const { return mnObs; }
  cv::Mat GetDescriptor() const { return mDescriptor; }
  Eigen::Vector3f GetWorldPos() const { return mWorldPos; }
  Eigen::Vector3f GetNormal() const { return mNormalVector; }
  float GetMinDistanceInvariance() const { return 0.8f * mfMinDistance; }   // MapPoint.cc:  0.8 * min, 1.2 * max
  float GetMaxDistanceInvariance() const { return 1.2f * mfMaxDistance; }
  bool IsInKeyFrame(KeyFramePtr pKF) const { return mObservations.count(pKF) != 0; }
  std::map<KeyFramePtr, std::tuple<int, int>> GetObservations() const { return mObservations; }
  std::tuple<int, int> GetIndexInKeyFrame(KeyFramePtr pKF) const {
    auto it = mObservations.find(pKF);
    return it == mObservations.end() ? std::tuple<int, int>(-1, -1) : it->second;
  }
  int PredictScale(const float&, KeyFramePtr) const { SLAM_SHIM_UNUSED("MapPoint::PredictScale(KF)"); }
  int PredictScale(const float&, Frame*) const { SLAM_SHIM_UNUSED("MapPoint::PredictScale(F)"); }
  void AddObservation(KeyFramePtr, int) { SLAM_SHIM_UNUSED("MapPoint::AddObservation"); }
  void Replace(MapPointPtr) { SLAM_SHIM_UNUSED("MapPoint::Replace"); }
  void IncreaseVisible(int = 1) {}
  void IncreaseFound(int = 1) {}
  Map* GetMap() const { return nullptr; }
};

class MapLine {
 public:
  Eigen::Vector3f mWorldPosStart, mWorldPosEnd, mNormalVector;
  cv::Mat mDescriptor;                         // 1 x 32 CV_8U
  bool mbBad = false;
  int mnObs = 1;
  float mfMinDistance = 0.f, mfMaxDistance = 1e9f;
  long unsigned int mnId = 0;
  std::map<KeyFramePtr, std::tuple<int, int>> mObservations;
  // tracking variables (MapLine.h)
  float mTrackProjStartX = 0, mTrackProjStartY = 0, mTrackStartDepth = 0, mTrackProjStartXR = 0, mTrackProjStartYR = 0,
        mTrackStartDepthR = 0;
  float mTrackProjEndX = 0, mTrackProjEndY = 0, mTrackEndDepth = 0, mTrackProjEndXR = 0, mTrackProjEndYR = 0,
        mTrackEndDepthR = 0;
  float mTrackProjMiddleX = 0, mTrackProjMiddleY = 0, mTrackProjMiddleXR = 0, mTrackProjMiddleYR = 0, mTrackMiddleDepth = 0,
        mTrackMiddleDepthR = 0;
  bool mbTrackInView = false, mbTrackInViewR = false;
  int mnTrackScaleLevel = 0, mnTrackScaleLevelR = 0;
  float mTrackViewCos = 1.f, mTrackViewCosR = 1.f;
  long unsigned int mnTrackReferenceForFrame = 0, mnLastFrameSeen = 0, mnFuseCandidateForKF = 0;
  static std::mutex mGlobalMutex;

  bool isBad() const { return mbBad; }
  int Observations() const { return mnObs; }
  cv::Mat GetDescriptor() const { return mDescriptor; }
  Eigen::Vector3f GetWorldPosStart() const { return mWorldPosStart; }
  Eigen::Vector3f GetWorldPosEnd() const { return mWorldPosEnd; }
  void GetWorldEndPoints(Eigen::Vector3f& s, Eigen::Vector3f& e) const { s = mWorldPosStart; e = mWorldPosEnd; }
  Eigen::Vector3f GetNormal() const { return mNormalVector; }
  float GetMinDistanceInvariance() const { return 0.8f * mfMinDistance; }
  float GetMaxDistanceInvariance() const { return 1.2f * mfMaxDistance; }
  float GetLength() const { return (mWorldPosEnd - mWorldPosStart).norm(); }
  bool IsInKeyFrame(KeyFramePtr pKF) const { return mObservations.count(pKF) != 0; }
  std::map<KeyFramePtr, std::tuple<int, int>> GetObservations() const { return mObservations; }
  std::tuple<int, int> GetIndexInKeyFrame(KeyFramePtr pKF) const {
    auto it = mObservations.find(pKF);
    return it == mObservations.end() ? std::tuple<int, int>(-1, -1) : it->second;
  }
  int PredictScale(const float&, KeyFramePtr) const { SLAM_SHIM_UNUSED("MapLine::PredictScale(KF)"); }
  int PredictScale(const float&, Frame*) const { SLAM_SHIM_UNUSED("MapLine::PredictScale(F)"); }
  void AddObservation(KeyFramePtr, int) { SLAM_SHIM_UNUSED("MapLine::AddObservation"); }
  void Replace(MapLinePtr) { SLAM_SHIM_UNUSED("MapLine::Replace"); }
  void IncreaseVisible(int = 1) {}
  void IncreaseFound(int = 1) {}
  Map* GetMap() const { return nullptr; }
};

class MapObject {
 public:
  bool isBad() const { return false; }
};

class KeyFrame {
 public:
  static const int kMaxInt = 2147483647;
  static float skFovCenterDistance;
  long unsigned int mnId = 0, mnFrameId = 0;
  int N = 0, NLeft = -1, Nlines = 0, NlinesLeft = -1;
  float fx = 0, fy = 0, cx = 0, cy = 0, invfx = 0, invfy = 0, mbf = 0, mb = 0, mThDepth = 0;
  std::vector<cv::KeyPoint> mvKeys, mvKeysUn, mvKeysRight;
  std::vector<float> mvuRight, mvDepth;
  cv::Mat mDescriptors, mLineDescriptors;
  DBoW2::BowVector mBowVec;
  DBoW2::FeatureVector mFeatVec;
  std::vector<cv::line_descriptor_c::KeyLine> mvKeyLines, mvKeyLinesUn, mvKeyLinesRight, mvKeyLinesRightUn;
  std::vector<float> mvuRightLineStart, mvuRightLineEnd, mvDepthLineStart, mvDepthLineEnd;
  int mnScaleLevels = 0;
  float mfScaleFactor = 1.f, mfLogScaleFactor = 0.f;
  std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
  int mnLineScaleLevels = 0;
  float mfLineScaleFactor = 1.f, mfLineLogScaleFactor = 0.f;
  std::vector<float> mvLineScaleFactors, mvLineLevelSigma2, mvLineInvLevelSigma2;
  GeometricCamera *mpCamera = nullptr, *mpCamera2 = nullptr;
  std::vector<MapPointPtr> mvpMapPoints;
  std::vector<MapLinePtr> mvpMapLines;
  Sophus::SE3f mTcw;
  bool mbBad = false;
  float mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;
  Sophus::SE3f GetRelativePoseTrl() const { SLAM_SHIM_UNUSED("KeyFrame::GetRelativePoseTrl"); }
  Sophus::SE3f GetRelativePoseTlr() const { SLAM_SHIM_UNUSED("KeyFrame::GetRelativePoseTlr"); }

  bool isBad() const { return mbBad; }
  std::vector<MapPointPtr> GetMapPointMatches() const { return mvpMapPoints; }
  std::vector<MapLinePtr> GetMapLineMatches() const { return mvpMapLines; }
  MapPointPtr GetMapPoint(const size_t i) const { return mvpMapPoints[i]; }
  MapLinePtr GetMapLine(const size_t i) const { return mvpMapLines[i]; }
  std::set<MapPointPtr> GetMapPoints() const { SLAM_SHIM_UNUSED("KeyFrame::GetMapPoints"); }
  std::unordered_set<MapPointPtr> GetMapPointsUnordered() const { SLAM_SHIM_UNUSED("KeyFrame::GetMapPointsUnordered"); }
  std::unordered_set<MapLinePtr> GetMapLinesUnordered() const { SLAM_SHIM_UNUSED("KeyFrame::GetMapLinesUnordered"); }
  void AddMapPoint(MapPointPtr, const size_t) { SLAM_SHIM_UNUSED("KeyFrame::AddMapPoint"); }
  void AddMapLine(MapLinePtr, const size_t) { SLAM_SHIM_UNUSED("KeyFrame::AddMapLine"); }
  Sophus::SE3f GetPose() const { return mTcw; }
  Sophus::SE3f GetPoseInverse() const { return mTcw.inverse(); }
  Sophus::SE3f GetRightPose() const { SLAM_SHIM_UNUSED("KeyFrame::GetRightPose"); }
  Sophus::SE3f GetRightPoseInverse() const { SLAM_SHIM_UNUSED("KeyFrame::GetRightPoseInverse"); }
  Eigen::Vector3f GetCameraCenter() const { return mTcw.inverse().translation(); }
  Eigen::Vector3f GetRightCameraCenter() const { SLAM_SHIM_UNUSED("KeyFrame::GetRightCameraCenter"); }
  Eigen::Matrix3f GetRotation() const { return mTcw.rotationMatrix(); }
  Eigen::Vector3f GetTranslation() const { return mTcw.translation(); }
  bool IsInImage(const float&, const float&) const { SLAM_SHIM_UNUSED("KeyFrame::IsInImage"); }
  std::vector<size_t> GetFeaturesInArea(const float&, const float&, const float&, const bool = false) const {
    SLAM_SHIM_UNUSED("KeyFrame::GetFeaturesInArea");
  }
  template <class... A>
  std::vector<size_t> GetLineFeaturesInArea(A&&...) const { SLAM_SHIM_UNUSED("KeyFrame::GetLineFeaturesInArea"); }
  Map* GetMap() const { return nullptr; }
};

}  // namespace PLVS2
