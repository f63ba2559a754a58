// What src/Frame.cc, src/ORBmatcher.cc and src/LineMatcher.cc name from OpenCV beyond oracle/ref/cv_full/cvfull.hpp.
// TEST INFRASTRUCTURE ONLY (see slam_shim.h).  Arithmetic that lives inside OpenCV and lies on the compared path:
//   cv::norm(a, b, NORM_L1) on CV_8U   the exact integer sum of absolute differences (returned as double)
//   cv::undistortPoints                 OpenCV 4.10 calib3d/undistort.dispatch.cpp cvUndistortPointsInternal: normalise
//                                       with the camera matrix, five fixed-point iterations of the inverse distortion
//                                       (TermCriteria COUNT 5, EPS 0.01 — the default), project with P; in double
#pragma once
#include <opencv2/core/core.hpp>
namespace cv {
template <class T>
struct Point3_ {
  T x, y, z;
  Point3_() : x(0), y(0), z(0) {}
  Point3_(T a, T b, T c) : x(a), y(b), z(c) {}
};
typedef Point3_<float> Point3f;
typedef Point3_<double> Point3d;
enum { NORM_INF = 1 };
enum { COLOR_GRAY2BGR = 8, LINE_AA = 16 };
struct DescriptorMatcher {};
struct BFMatcher : DescriptorMatcher {
  explicit BFMatcher(int = NORM_L2, bool = false) {}
  void knnMatch(const Mat&, const Mat&, std::vector<std::vector<DMatch>>&, int) const { std::abort(); }
};
inline double norm(const Mat& a, const Mat& b, int type) {
  if (type != NORM_L1 || a.type() != CV_8U || b.type() != CV_8U || a.rows != b.rows || a.cols != b.cols) std::abort();
  long long s = 0;
  for (int r = 0; r < a.rows; ++r) {
    const uchar *pa = a.ptr<uchar>(r), *pb = b.ptr<uchar>(r);
    for (int c = 0; c < a.cols; ++c) s += pa[c] > pb[c] ? pa[c] - pb[c] : pb[c] - pa[c];
  }
  return (double)s;
}
// cv::undistortPoints(src, dst, cameraMatrix, distCoeffs, R = noArray(), P) — OpenCV 4.10 modules/calib3d/src/undistort.dispatch.cpp,
// cvUndistortPointsInternal with the default criteria (MAX_ITER 5): src / dst = N x 2 CV_32F (see Mat::reshape above), camera
// matrices CV_32F 3 x 3, 4 / 5 / 8 distortion coefficients CV_32F (k1 k2 p1 p2 [k3 [k4 k5 k6]]); all arithmetic in double,
// results rounded to float on the store.  No tilt, no thin-prism terms (their coefficients are zero here), R = identity.
inline void undistortPoints(const Mat& src, Mat& dst, const Mat& K, const Mat& D, const Mat& R, const Mat& P) {
  if (!R.empty() || src.type() != CV_32F || src.cols != 2 || K.type() != CV_32F || P.type() != CV_32F || D.type() != CV_32F)
    std::abort();
  double k[14] = {0};
  const int nd = (int)D.total();
  if (nd != 4 && nd != 5 && nd != 8) std::abort();
  for (int i = 0; i < nd; ++i) k[i] = (double)D.at<float>(i);
  const double fx = K.at<float>(0, 0), fy = K.at<float>(1, 1), ifx = 1. / fx, ify = 1. / fy, cx = K.at<float>(0, 2), cy = K.at<float>(1, 2);
  double RR[3][3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) RR[r][c] = (double)P.at<float>(r, c);   // PP * RR with RR = identity
  Mat out(src.rows, 2, CV_32F);
  for (int i = 0; i < src.rows; ++i) {
    double x = src.at<float>(i, 0), y = src.at<float>(i, 1);
    const double u = x, v = y;
    (void)u; (void)v;
    x = (x - cx) * ifx;
    y = (y - cy) * ify;
    const double x0 = x, y0 = y;   // (vecUntilt = identity * (x, y, 1), invProj = 1)
    for (int j = 0; j < 5; ++j) {
      const double r2 = x * x + y * y;
      double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
      if (icdist < 0) {   // test: undistortPoints.regression_14583
        x = (src.at<float>(i, 0) - cx) * ifx;
        y = (src.at<float>(i, 1) - cy) * ify;
        break;
      }
      const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
      const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
      x = (x0 - deltaX) * icdist;
      y = (y0 - deltaY) * icdist;
    }
    const double xx = RR[0][0] * x + RR[0][1] * y + RR[0][2];
    const double yy = RR[1][0] * x + RR[1][1] * y + RR[1][2];
    const double ww = 1. / (RR[2][0] * x + RR[2][1] * y + RR[2][2]);
    out.at<float>(i, 0) = (float)(xx * ww);
    out.at<float>(i, 1) = (float)(yy * ww);
  }
  dst = out;
}
namespace fisheye {
inline void undistortPoints(const Mat&, Mat&, const Mat&, const Mat&, const Mat& = Mat(), const Mat& = Mat()) { std::abort(); }
}
struct LineIterator {   // (ComputeStereoLinesFromRGBD: outside the compiled path)
  LineIterator(const Mat&, Point, Point, int = 8, bool = false) { std::abort(); }
  int count = 0;
  LineIterator& operator++() { return *this; }
  LineIterator operator++(int) { return *this; }
  Point pos() const { return Point(); }
};
inline void line(Mat&, Point, Point, const Scalar&, int = 1, int = 8, int = 0) { std::abort(); }
inline void vconcat(const Mat&, const Mat&, Mat&) { std::abort(); }
inline void undistort(const Mat&, Mat&, const Mat&, const Mat&, const Mat& = Mat()) { std::abort(); }
inline void imshow(const std::string&, const Mat&) { std::abort(); }
inline int waitKey(int = 0) { std::abort(); }
}  // namespace cv
