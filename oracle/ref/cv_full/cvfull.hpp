// Stand-in for the part of OpenCV that the reference's front end (src/ORBextractor.cc, src/LineExtractor.cc,
// Thirdparty/line_descriptor/src/binary_descriptor_custom.cpp) is written against, so that those sources compile
// UNMODIFIED where they lie (oracle/ref/Makefile -> oracle/_ref/libfrontend_ref.so).  TEST INFRASTRUCTURE ONLY.
//
// Two kinds of content:
//  * containers and glue — cv::Mat (reference-counted, typed, with a row step and regions of interest that share
//    the parent's storage: ORBextractor reads the 19-pixel border around a pyramid level through exactly that),
//    Mat_<T> with the few matrix expressions line_descriptor uses, KeyPoint, Ptr, Algorithm, ...: plain C++;
//  * the image primitives whose ARITHMETIC lives inside OpenCV (4.10 is what the reference pins; its source is not in
//    the reference tree): FAST, resize(INTER_LINEAR), GaussianBlur, Sobel, copyMakeBorder, fastAtan2, cvRound and
//    the element-wise CV_16S operations.  They forward to oracle/cv_primitives.hpp, the same restatements
//    oracle/orb.cpp and oracle/lines.cpp use.  So what the compiled reference pins is everything that is PLVS's own
//    — the per-cell FAST loop, DistributeOctTree, IC_Angle, computeOrbDescriptor, the key point packing; EdgeDrawing's
//    routing, the line fits, validation, octave grouping, computeLBD — "up to the OpenCV primitives".
#pragma once
#include <algorithm>
#include <cassert>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <list>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../cv_primitives.hpp"

#define CV_EXPORTS
#define CV_EXPORTS_W
#define CV_EXPORTS_W_SIMPLE
#define CV_WRAP
#define CV_OUT
#define CV_IN_OUT
#define CV_PROP_RW
#define CV_PROP
#ifndef CV_OVERRIDE
#define CV_OVERRIDE override
#endif
#ifndef CV_FINAL
#define CV_FINAL final
#endif
#define CV_8U 0
#define CV_8S 1
#define CV_16U 2
#define CV_16S 3
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_8UC1 CV_8U
#define CV_8SC1 CV_8S
#define CV_16SC1 CV_16S
#define CV_32SC1 CV_32S
#define CV_32FC1 CV_32F
#define CV_64FC1 CV_64F
#define CV_Assert(x) assert(x)
#define CV_DbgAssert(x) assert(x)
#define CV_Error(code, msg) throw std::runtime_error(msg)
#define CV_StsBadArg 0
#define CV_StsBadSize 0
#define CV_PI 3.1415926535897932384626433832795

typedef unsigned char uchar;
typedef signed char schar;
typedef unsigned short ushort;

namespace cv {

namespace Error { enum { StsBadArg = -5, StsBadSize = -201, BadDepth = -17 }; }

typedef std::string String;

// opencv2/core/cvstd.hpp pulls these into namespace cv: inside cv:: (line_descriptor's code) an unqualified
// sqrt / exp / pow / log / abs / min / max on floats therefore takes the std:: float overload, while cos, sin,
// atan2, fabs, round, log10 fall through to the C (double) functions.  Part of the arithmetic contract.
using std::min;
using std::max;
using std::abs;
using std::swap;
using std::sqrt;
using std::exp;
using std::pow;
using std::log;

inline int cvRound(double v) { return ocv::cv_round(v); }
inline int cvRound(float v) { return ocv::cv_round(v); }
inline int cvRound(int v) { return v; }
inline int cvFloor(double v) { return ocv::cv_floor(v); }
inline int cvCeil(double v) { return ocv::cv_ceil(v); }
inline float fastAtan2(float y, float x) { return ocv::fast_atan2(y, x); }

template <class T> inline T saturate_cast(int v) { return (T)v; }
template <> inline uchar saturate_cast<uchar>(int v) { return (uchar)(v < 0 ? 0 : v > 255 ? 255 : v); }
template <> inline short saturate_cast<short>(int v) { return ocv::saturate_short(v); }

template <class T>
struct Ptr : std::shared_ptr<T> {
  Ptr() {}
  Ptr(T* p) : std::shared_ptr<T>(p) {}
  template <class U> Ptr(const std::shared_ptr<U>& o) : std::shared_ptr<T>(o) {}
  void release() { this->reset(); }
  bool empty() const { return this->get() == nullptr; }
};
template <class T, class... A>
Ptr<T> makePtr(A&&... a) { return Ptr<T>(std::make_shared<T>(std::forward<A>(a)...)); }
enum { LSD_REFINE_NONE = 0, LSD_REFINE_STD = 1, LSD_REFINE_ADV = 2 };

template <class T>
struct Point_ {
  T x, y;
  Point_() : x(0), y(0) {}
  Point_(T a, T b) : x(a), y(b) {}
  template <class U> Point_(const Point_<U>& o) : x((T)o.x), y((T)o.y) {}
  Point_& operator*=(float s) { x = (T)(x * s); y = (T)(y * s); return *this; }   // (types.hpp: saturate_cast<T>(x * s); T = float here)
  Point_& operator+=(const Point_& o) { x += o.x; y += o.y; return *this; }
  bool operator==(const Point_& o) const { return x == o.x && y == o.y; }
};
typedef Point_<float> Point2f;
typedef Point_<int> Point2i;
typedef Point2i Point;
struct Size {
  int width = 0, height = 0;
  Size() {}
  Size(int w, int h) : width(w), height(h) {}
  bool operator==(const Size& o) const { return width == o.width && height == o.height; }
  bool operator!=(const Size& o) const { return !(*this == o); }
  bool empty() const { return width <= 0 || height <= 0; }
};
struct Rect {
  int x = 0, y = 0, width = 0, height = 0;
  Rect() {}
  Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {}
};
struct Range { int start, end; Range(int s, int e) : start(s), end(e) {} };
struct Scalar {
  double v[4];
  Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : v{a, b, c, d} {}
  static Scalar all(double a) { return Scalar(a, a, a, a); }
};
struct KeyPoint {
  Point2f pt;
  float size = 0, angle = -1, response = 0;
  int octave = 0, class_id = -1;
  KeyPoint() {}
  KeyPoint(float x, float y, float s, float a = -1, float r = 0, int o = 0, int c = -1)
      : pt(x, y), size(s), angle(a), response(r), octave(o), class_id(c) {}
};
struct DMatch {
  int queryIdx = -1, trainIdx = -1, imgIdx = -1;
  float distance = 0;
};
struct FileNode {   // (parameters are never read from a file on this path)
  FileNode operator[](const char*) const { return FileNode(); }
  bool empty() const { return true; }
  bool isReal() const { return false; }
  bool isInt() const { return false; }
  bool isString() const { return false; }
  operator double() const { return 0.0; }
  operator std::string() const { return std::string(); }
  operator int() const { return 0; }
  operator float() const { return 0.f; }
};
struct FileStorage {
  FileNode operator[](const std::string&) const { return FileNode(); }
  FileNode operator[](const char*) const { return FileNode(); }
};
template <class T> inline FileStorage& operator<<(FileStorage& fs, const T&) { return fs; }
enum { NORM_L1 = 2, NORM_L2 = 4, NORM_HAMMING = 6 };

enum { INTER_NEAREST = 0, INTER_LINEAR = 1, INTER_LINEAR_EXACT = 5 };
enum { BORDER_CONSTANT = 0, BORDER_REPLICATE = 1, BORDER_REFLECT = 2, BORDER_WRAP = 3, BORDER_REFLECT_101 = 4,
       BORDER_DEFAULT = 4, BORDER_ISOLATED = 16 };
enum { THRESH_BINARY = 0, THRESH_TOZERO = 3 };
enum { CMP_EQ = 0, CMP_GT = 1, CMP_GE = 2, CMP_LT = 3, CMP_LE = 4, CMP_NE = 5 };
enum { COLOR_BGR2GRAY = 6 };

inline size_t elem_size_of(int type) {
  static const size_t s[7] = {1, 1, 2, 2, 4, 4, 8};
  return s[type & 7];
}
template <class T> struct DataType;
template <> struct DataType<uchar> { enum { type = CV_8U }; };
template <> struct DataType<schar> { enum { type = CV_8S }; };
template <> struct DataType<short> { enum { type = CV_16S }; };
template <> struct DataType<int> { enum { type = CV_32S }; };
template <> struct DataType<float> { enum { type = CV_32F }; };
template <> struct DataType<double> { enum { type = CV_64F }; };

struct MatStep {
  size_t p = 0;
  operator size_t() const { return p; }
  size_t operator[](int i) const { return i == 0 ? p : 0; }
};

#ifdef PLVS_CVFULL_LSD
template <class T, int N> struct Vec;
struct _OutputArray;
#endif
class Mat {
 public:
  int rows = 0, cols = 0;
  int type_ = CV_8U;
  uchar* data = nullptr;
  MatStep step;
  std::shared_ptr<std::vector<uchar> > buf;   // the storage a region of interest shares with its parent

  Mat() {}
  Mat(int r, int c, int t) { create(r, c, t); }
  Mat(Size s, int t) { create(s.height, s.width, t); }
  Mat(int r, int c, int t, const Scalar&) { std::abort(); }   // (drawing code only: never on a compared path)
  Mat(int r, int c, int t, void* ext, size_t st = 0) : rows(r), cols(c), type_(t), data(static_cast<uchar*>(ext)) {
    step.p = st ? st : (size_t)c * elem_size_of(t);   // (no copy, no ownership)
  }
  void create(int r, int c, int t) {
    if (data && rows == r && cols == c && type_ == t) return;   // Mat::create keeps a fitting allocation: resize / copyMakeBorder into a ROI rely on it
    rows = r; cols = c; type_ = t;
    step.p = (size_t)c * elem_size_of(t);
    buf = std::make_shared<std::vector<uchar> >(step.p * (size_t)r + 64);
    data = buf->data();
  }
  void create(Size s, int t) { create(s.height, s.width, t); }
  bool empty() const { return rows == 0 || cols == 0 || data == nullptr; }
  int type() const { return type_; }
  int depth() const { return type_ & 7; }
  int channels() const { return 1; }
  size_t elemSize() const { return elem_size_of(type_); }
  size_t step1() const { return step.p / elemSize(); }
  size_t total() const { return (size_t)rows * cols; }
  Size size() const { return Size(cols, rows); }
  bool isContinuous() const { return step.p == (size_t)cols * elemSize(); }
  void release() { rows = cols = 0; buf.reset(); data = nullptr; step.p = 0; }
  Mat clone() const {
    Mat m;
    copyTo(m);
    return m;
  }
  void copyTo(Mat& m) const {
    if (empty()) { m.release(); return; }
    m.create(rows, cols, type_);
    for (int r = 0; r < rows; ++r) std::memmove(m.ptr(r), ptr(r), (size_t)cols * elemSize());
  }
  void copyTo(Mat&& m) const { Mat& ref = m; copyTo(ref); }   // (desc.row(i).copyTo(descriptors.row(k)))
  Mat operator()(const Rect& r) const {
    Mat m = *this;
    m.rows = r.height; m.cols = r.width;
    m.data = data + (size_t)r.y * step.p + (size_t)r.x * elemSize();
    return m;
  }
  // (Frame.cc reshapes an N x 2 CV_32F matrix to N x 1 two-channel and back around cv::undistortPoints: the stand-in
  // keeps the N x 2 layout, its undistortPoints reads that)
  Mat reshape(int) const { return *this; }
  void reserve(size_t) {}
  static Mat eye(int, int, int) { std::abort(); }
  Mat rowRange(int a, int b) const { return (*this)(Rect(0, a, cols, b - a)); }
  Mat colRange(int a, int b) const { return (*this)(Rect(a, 0, b - a, rows)); }
  Mat row(int r) const { return rowRange(r, r + 1); }
  uchar* ptr(int r = 0) { return data + (size_t)r * step.p; }
  const uchar* ptr(int r = 0) const { return data + (size_t)r * step.p; }
  template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * step.p); }
  template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * step.p); }
  template <class T> T& at(int r, int c) { return ptr<T>(r)[c]; }
  template <class T> const T& at(int r, int c) const { return ptr<T>(r)[c]; }
  template <class T> const T& at(const Point_<int>& p) const { return ptr<T>(p.y)[p.x]; }
  template <class T> T& at(const Point_<int>& p) { return ptr<T>(p.y)[p.x]; }
  Mat col(int c) const { return colRange(c, c + 1); }
#ifdef PLVS_CVFULL_LSD
  // (lsd_custom.cpp:452-455: Mat(lines).copyTo(_lines) — a matrix over a vector's elements, copied into the caller's vector)
  explicit Mat(const std::vector<Vec<float, 4> >& v);
  explicit Mat(const std::vector<double>& v);
  void copyTo(const _OutputArray& o) const;
  int checkVector(int) const { std::abort(); }                 // (drawSegments / compareSegments: never on a compared path)
  void convertTo(Mat&, int) const { std::abort(); }
#endif
  template <class T> T& at(int i) { return rows == 1 ? ptr<T>(0)[i] : ptr<T>(i / cols)[i % cols]; }
  template <class T> const T& at(int i) const { return rows == 1 ? ptr<T>(0)[i] : ptr<T>(i / cols)[i % cols]; }
  void push_back(const Mat& o) {   // append rows
    if (o.empty()) return;
    Mat m;
    assert(empty() || (cols == o.cols && type_ == o.type_));
    m.create(rows + o.rows, o.cols, o.type_);
    for (int r = 0; r < rows; ++r) std::memcpy(m.ptr(r), ptr(r), (size_t)cols * elemSize());
    for (int r = 0; r < o.rows; ++r) std::memcpy(m.ptr(rows + r), o.ptr(r), (size_t)o.cols * o.elemSize());
    *this = m;
  }
  Mat& operator=(const Scalar& s) {   // setTo
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < cols; ++c) set(r, c, s.v[0]);
    return *this;
  }
  Mat& setTo(const Scalar& s) { return *this = s; }
  double get(int r, int c) const {
    switch (depth()) {
      case CV_8U: return at<uchar>(r, c);
      case CV_8S: return at<schar>(r, c);
      case CV_16U: return at<ushort>(r, c);
      case CV_16S: return at<short>(r, c);
      case CV_32S: return at<int>(r, c);
      case CV_32F: return at<float>(r, c);
      default: return at<double>(r, c);
    }
  }
  void set(int r, int c, double v) {
    switch (depth()) {
      case CV_8U: at<uchar>(r, c) = saturate_cast<uchar>(cvRound(v)); break;
      case CV_8S: at<schar>(r, c) = (schar)cvRound(v); break;
      case CV_16U: at<ushort>(r, c) = (ushort)cvRound(v); break;
      case CV_16S: at<short>(r, c) = saturate_cast<short>(cvRound(v)); break;
      case CV_32S: at<int>(r, c) = cvRound(v); break;
      case CV_32F: at<float>(r, c) = (float)v; break;
      default: at<double>(r, c) = v; break;
    }
  }
  Mat t() const {
    Mat m(cols, rows, type_);
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < cols; ++c) std::memcpy(m.ptr(c) + (size_t)r * elemSize(), ptr(r) + (size_t)c * elemSize(), elemSize());
    return m;
  }
  // the InputArray / OutputArray faces of a matrix
  const Mat& getMat() const { return *this; }
  Mat& getMat() { return *this; }
  static Mat zeros(int r, int c, int t) { Mat m(r, c, t); for (int i = 0; i < r; ++i) std::memset(m.ptr(i), 0, m.step.p); return m; }
  static Mat ones(int r, int c, int t) { Mat m(r, c, t); m = Scalar(1); return m; }
};
typedef const Mat& InputArray;
#ifndef PLVS_CVFULL_LSD
typedef Mat& OutputArray;
typedef Mat& InputOutputArray;
inline Mat& noArray() { static Mat none; return none; }
#else
// The LSD sources (lsd_custom.cpp, LSDDetector_custom.cpp: oracle/_ref/liblsd_ref.so) hand a std::vector<Vec4f> to an
// OutputArray and ask an optional one whether it is needed(): for that build the output face is a small proxy — a matrix, a
// vector of segments or of doubles, or nothing — that still converts to the Mat& the image primitives below take.
template <class T, int N>
struct Vec {
  T val[N];
  Vec() { for (int i = 0; i < N; ++i) val[i] = T(0); }
  Vec(T a, T b) { static_assert(N == 2, ""); val[0] = a; val[1] = b; }
  Vec(T a, T b, T c, T d) { static_assert(N == 4, ""); val[0] = a; val[1] = b; val[2] = c; val[3] = d; }
  T& operator[](int i) { return val[i]; }
  const T& operator[](int i) const { return val[i]; }
};
typedef Vec<float, 4> Vec4f;
typedef Vec<int, 4> Vec4i;
struct _OutputArray {
  Mat* m = nullptr;
  std::vector<Vec4f>* v4 = nullptr;
  std::vector<double>* vd = nullptr;
  _OutputArray() {}
  _OutputArray(Mat& x) : m(&x) {}
  _OutputArray(std::vector<Vec4f>& x) : v4(&x) {}
  _OutputArray(std::vector<double>& x) : vd(&x) {}
  bool needed() const { return m != nullptr || v4 != nullptr || vd != nullptr; }
  bool empty() const { return m == nullptr || m->empty(); }
  operator Mat&() const { if (!m) std::abort(); return *m; }
  Mat& getMatRef() const { if (!m) std::abort(); return *m; }
  Mat getMat() const { return m ? *m : Mat(); }
  int channels() const { return 1; }
  Size size() const { return m ? m->size() : Size(); }
  void create(int r, int c, int t) const { if (!m) std::abort(); m->create(r, c, t); }
  void release() const { if (m) m->release(); }
};
typedef const _OutputArray& OutputArray;
typedef const _OutputArray& InputOutputArray;
inline const _OutputArray& noArray() { static _OutputArray none; return none; }
#endif
typedef const std::vector<Mat>& InputArrayOfArrays;

template <class T> struct MatCommaInitializer_;
template <class T>
struct Mat_ : Mat {
  Mat_() { type_ = DataType<T>::type; }
  Mat_(int r, int c) : Mat(r, c, DataType<T>::type) {}
  explicit Mat_(Size s) : Mat(s.height, s.width, DataType<T>::type) {}
  static Mat_ zeros(Size s) { Mat_ m(s); for (int i = 0; i < s.height; ++i) std::memset(m.ptr(i), 0, m.step.p); return m; }
  Mat_(int r, int c, const T& v) : Mat(r, c, DataType<T>::type) {   // (rows x cols, every element v)
    for (int i = 0; i < r; ++i)
      for (int k = 0; k < c; ++k) this->template at<T>(i, k) = v;
  }
  Mat_(const Mat& m) { *this = m; }
  Mat_& operator=(const Mat& m) {   // shares a matrix of the same type, converts one of another
    if (m.type() == DataType<T>::type || m.empty()) {
      Mat::operator=(m);
      type_ = DataType<T>::type;
    } else {
      Mat c(m.rows, m.cols, DataType<T>::type);
      for (int r = 0; r < m.rows; ++r)
        for (int k = 0; k < m.cols; ++k) c.set(r, k, m.get(r, k));
      Mat::operator=(c);
    }
    return *this;
  }
  template <class U> Mat_& operator=(const Mat_<U>& m) { return *this = static_cast<const Mat&>(m); }
  Mat_& operator=(const Mat_& m) { Mat::operator=(static_cast<const Mat&>(m)); return *this; }
  Mat_(const Mat_& m) : Mat(static_cast<const Mat&>(m)) {}
  Mat_(const MatCommaInitializer_<T>& ci);
  T* operator[](int r) { return this->template ptr<T>(r); }
  const T* operator[](int r) const { return this->template ptr<T>(r); }
  T& operator()(int r, int c) { return this->template at<T>(r, c); }
  const T& operator()(int r, int c) const { return this->template at<T>(r, c); }
};
template <class T>
struct MatCommaInitializer_ {
  Mat_<T> m;
  int i = 0;
  explicit MatCommaInitializer_(const Mat_<T>& m_) : m(m_) {}
  template <class U> MatCommaInitializer_& operator,(U v) { m.template ptr<T>(i / m.cols)[i % m.cols] = (T)v; ++i; return *this; }
  operator Mat() const { return m; }
};
template <class T> Mat_<T>::Mat_(const MatCommaInitializer_<T>& ci) : Mat(static_cast<const Mat&>(ci.m)) {}
template <class T, class U>
MatCommaInitializer_<T> operator<<(const Mat_<T>& m, U v) {
  MatCommaInitializer_<T> ci(m);
  return (ci, v);
}

// ---- the matrix expressions of line_descriptor's line fit (CV_32F): products accumulate in double and store float
// (gemm's small-matrix path, as oracle/lines.cpp's dotf reads it); sums are float + float
inline Mat operator*(const Mat& a, const Mat& b) {
  assert(a.type() == CV_32F && b.type() == CV_32F && a.cols == b.rows);
  Mat m(a.rows, b.cols, CV_32F);
  for (int r = 0; r < a.rows; ++r)
    for (int c = 0; c < b.cols; ++c) {
      double s = 0;
      for (int k = 0; k < a.cols; ++k) s += (double)a.at<float>(r, k) * (double)b.at<float>(k, c);
      m.at<float>(r, c) = (float)s;
    }
  return m;
}
inline Mat operator+(const Mat& a, const Mat& b) {
  assert(a.type() == CV_32F && b.type() == CV_32F && a.rows == b.rows && a.cols == b.cols);
  Mat m(a.rows, a.cols, CV_32F);
  for (int r = 0; r < a.rows; ++r)
    for (int c = 0; c < a.cols; ++c) m.at<float>(r, c) = a.at<float>(r, c) + b.at<float>(r, c);
  return m;
}
// Mat / scalar on an integer matrix = convertTo(alpha = 1 / s): saturate_cast<short>(cvRound(x * alpha)) — round half to even
inline Mat operator/(const Mat& a, double s) {
  assert(a.type() == CV_16S);
  Mat m(a.rows, a.cols, CV_16S);
  const float alpha = (float)(1.0 / s);
  for (int r = 0; r < a.rows; ++r)
    for (int c = 0; c < a.cols; ++c) m.at<short>(r, c) = saturate_cast<short>(cvRound((float)a.at<short>(r, c) * alpha));
  return m;
}

class Algorithm {
 public:
  virtual ~Algorithm() {}
  virtual void clear() {}
  virtual void read(const FileNode&) {}
  virtual void write(FileStorage&) const {}
};

#ifdef PLVS_CVFULL_LSD
inline Mat::Mat(const std::vector<Vec<float, 4> >& v) {
  create((int)v.size(), 4, CV_32F);
  for (size_t i = 0; i < v.size(); ++i) std::memcpy(ptr((int)i), v[i].val, 4 * sizeof(float));
}
inline Mat::Mat(const std::vector<double>& v) {
  create((int)v.size(), 1, CV_64F);
  for (size_t i = 0; i < v.size(); ++i) at<double>((int)i, 0) = v[i];
}
inline void Mat::copyTo(const _OutputArray& o) const {
  if (o.v4) {
    o.v4->resize((size_t)rows);
    for (int i = 0; i < rows; ++i) std::memcpy((*o.v4)[(size_t)i].val, ptr(i), 4 * sizeof(float));
  } else if (o.vd) {
    o.vd->resize((size_t)rows);
    for (int i = 0; i < rows; ++i) (*o.vd)[(size_t)i] = at<double>(i, 0);
  } else if (o.m) {
    copyTo(*o.m);
  }
}
// cv::LineIterator as LSDDetectorC uses it (LSDDetector_custom.cpp:256-257): only `count`, for end points inside the image
// (checkLineExtremes has clamped them).  imgproc/drawing.cpp: the Point2f end points convert to Point by saturate_cast<int>
// (cvRound); 8-connected: count = max(|dx|, |dy|) + 1.
struct LineIterator {
  int count;
  LineIterator(const Mat& img, Point2f p1, Point2f p2) {
    const int x1 = cvRound(p1.x), y1 = cvRound(p1.y), x2 = cvRound(p2.x), y2 = cvRound(p2.y);
    if ((unsigned)x1 >= (unsigned)img.cols || (unsigned)x2 >= (unsigned)img.cols || (unsigned)y1 >= (unsigned)img.rows ||
        (unsigned)y2 >= (unsigned)img.rows)
      std::abort();   // (would need clipLine: cannot happen behind checkLineExtremes)
    const int dx = x2 > x1 ? x2 - x1 : x1 - x2, dy = y2 > y1 ? y2 - y1 : y1 - y2;
    count = (dx > dy ? dx : dy) + 1;
  }
};
enum { COLOR_GRAY2BGR = 8 };
inline void cvtColor(const _OutputArray&, const _OutputArray&, int) { std::abort(); }
template <class P> inline void line(const _OutputArray&, P, P, const Scalar&, int = 1) { std::abort(); }
inline void bitwise_xor(const Mat&, const Mat&, Mat&) { std::abort(); }
inline int countNonZero(const Mat&) { std::abort(); }
#endif

// ---------------------------------------------------------------- image primitives (forwarded to oracle/cv_primitives.hpp)
inline ocv::Image to_image(const Mat& m) {
  assert(m.depth() == CV_8U);
  ocv::Image im(m.cols, m.rows);
  for (int r = 0; r < m.rows; ++r) std::memcpy(im.row(r), m.ptr(r), (size_t)m.cols);
  return im;
}
inline void from_image(const ocv::Image& im, Mat& m) {
  m.create(im.h, im.w, CV_8U);
  for (int r = 0; r < im.h; ++r) std::memcpy(m.ptr(r), im.row(r), (size_t)im.w);
}

inline void FAST(InputArray image, std::vector<KeyPoint>& kps, int threshold, bool nonmax = true) {
  std::vector<ocv::FastKp> out;
  ocv::fast_9_16(image.data, (int)image.step.p, image.cols, image.rows, threshold, nonmax, out);
  kps.clear();
  for (const ocv::FastKp& k : out) kps.push_back(KeyPoint(k.x, k.y, 7.f, -1, k.response));
}
inline void resize(InputArray src, Mat& dst, Size dsize, double fx = 0, double fy = 0, int interpolation = INTER_LINEAR) {
  const ocv::Image s = to_image(src);
  ocv::Image d;
  if (interpolation == INTER_LINEAR_EXACT) {   // (the LSD detector's rescaling, lsd_custom.cpp:493)
    assert(dsize.width == 0 && dsize.height == 0);
    ocv::resize_linear_exact_u8_factor(s, d, fx, fy);
    from_image(d, dst);
    return;
  }
  assert(interpolation == INTER_LINEAR);
  if (dsize.width > 0 && dsize.height > 0) {
    d = ocv::Image(dsize.width, dsize.height);
    ocv::resize_linear_u8(s, d);
  } else {
    ocv::resize_linear_u8_factor(s, d, fx, fy);
  }
  from_image(d, dst);
}
inline void copyMakeBorder(InputArray src, Mat& dst, int top, int bottom, int left, int right, int borderType) {
  assert((borderType & ~BORDER_ISOLATED) == BORDER_REFLECT_101);   // (src may be a region of dst: work from a copy)
  const ocv::Image s = to_image(src);
  ocv::Image d(s.w + left + right, s.h + top + bottom);
  for (int y = 0; y < d.h; ++y) {
    const uint8_t* sr = s.row(ocv::border_reflect101(y - top, s.h));
    uint8_t* dr = d.row(y);
    for (int x = 0; x < d.w; ++x) dr[x] = sr[ocv::border_reflect101(x - left, s.w)];
  }
  from_image(d, dst);
}
inline void GaussianBlur(InputArray src, Mat& dst, Size ksize, double sigmaX, double sigmaY = 0, int borderType = BORDER_DEFAULT) {
  assert(ksize.width == ksize.height && (sigmaY == 0 || sigmaY == sigmaX) && borderType == BORDER_REFLECT_101);
  const ocv::Image s = to_image(src);
  ocv::Image d;
  ocv::gaussian_blur_u8(s, d, ksize.width, sigmaX);
  from_image(d, dst);
}
inline void Sobel(InputArray src, Mat& dst, int ddepth, int dx, int dy, int ksize = 3) {
  assert(ddepth == CV_16S && ksize == 3 && dx + dy == 1);
  std::vector<short> gx, gy;
  ocv::sobel3_s16(to_image(src), gx, gy);
  dst.create(src.rows, src.cols, CV_16S);
  const std::vector<short>& g = dx ? gx : gy;
  for (int r = 0; r < src.rows; ++r) std::memcpy(dst.ptr(r), g.data() + (size_t)r * src.cols, (size_t)src.cols * 2);
}
inline Mat abs(const Mat& a) {
  assert(a.type() == CV_16S);
  Mat m(a.rows, a.cols, CV_16S);
  for (int r = 0; r < a.rows; ++r)
    for (int c = 0; c < a.cols; ++c) { const int v = a.at<short>(r, c); m.at<short>(r, c) = saturate_cast<short>(v < 0 ? -v : v); }
  return m;
}
inline void add(InputArray a, InputArray b, Mat& dst) {
  assert(a.type() == CV_16S && b.type() == CV_16S);
  Mat m(a.rows, a.cols, CV_16S);
  for (int r = 0; r < a.rows; ++r)
    for (int c = 0; c < a.cols; ++c) m.at<short>(r, c) = saturate_cast<short>((int)a.at<short>(r, c) + (int)b.at<short>(r, c));
  dst = m;
}
inline double threshold(InputArray src, Mat& dst, double thresh, double /*maxval*/, int type) {
  assert(src.type() == CV_16S && type == THRESH_TOZERO);
  const int ith = cvFloor(thresh);
  Mat m(src.rows, src.cols, CV_16S);
  for (int r = 0; r < src.rows; ++r)
    for (int c = 0; c < src.cols; ++c) { const short v = src.at<short>(r, c); m.at<short>(r, c) = v > ith ? v : (short)0; }
  dst = m;
  return thresh;
}
inline void compare(InputArray a, InputArray b, Mat& dst, int op) {
  assert(a.type() == CV_16S && b.type() == CV_16S && op == CMP_LT);
  Mat m(a.rows, a.cols, CV_8U);
  for (int r = 0; r < a.rows; ++r)
    for (int c = 0; c < a.cols; ++c) m.at<uchar>(r, c) = a.at<short>(r, c) < b.at<short>(r, c) ? 255 : 0;
  dst = m;
}
// named by paths the hot path never takes (colour input, the pyrDown pyramid, the sharpener)
inline void cvtColor(InputArray, Mat&, int) { std::abort(); }
inline void pyrDown(InputArray, Mat&, Size = Size()) { std::abort(); }
inline void filter2D(InputArray, Mat&, int, InputArray) { std::abort(); }

struct KeyPointsFilter {   // features2d: keep the n strongest (and every tie with the n-th)
  static void retainBest(std::vector<KeyPoint>& kps, int n) {
    if (n < 0 || kps.size() <= (size_t)n) return;
    if (n == 0) { kps.clear(); return; }
    std::nth_element(kps.begin(), kps.begin() + n - 1, kps.end(), [](const KeyPoint& a, const KeyPoint& b) { return a.response > b.response; });
    const float amb = kps[n - 1].response;
    auto e = std::partition(kps.begin() + n, kps.end(), [amb](const KeyPoint& k) { return k.response >= amb; });
    kps.resize(e - kps.begin());
  }
};

}  // namespace cv
using cv::cvRound;
using cv::cvFloor;
using cv::cvCeil;
