// Stand-in for the OpenCV names line_descriptor's header and binary_descriptor_matcher_custom.cpp mention, so that the
// matcher (Mihasher: multi-index hashing k-NN) compiles UNMODIFIED.  TEST INFRASTRUCTURE ONLY.
// cv::Mat here is a reference-counted 2-D byte matrix (rows x cols, one channel): all the matcher needs.
#pragma once
#include <cassert>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <list>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#define CV_EXPORTS
#define CV_EXPORTS_W
#define CV_EXPORTS_W_SIMPLE
#define CV_WRAP
#define CV_OUT
#define CV_IN_OUT
#define CV_PROP_RW
#define CV_PROP
#define CV_8UC1 0
#define CV_8U 0
#define CV_Assert(x) assert(x)
#define CV_Error(code, msg) throw std::runtime_error(msg)
#define CV_StsBadArg 0
#define CV_StsBadSize 0
#define CV_PI 3.1415926535897932384626433832795

typedef unsigned char uchar;
typedef unsigned short ushort;

namespace cv {

typedef std::string String;

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
};
typedef Point_<float> Point2f;
typedef Point_<int> Point;
struct Size { int width = 0, height = 0; Size() {} Size(int w, int h) : width(w), height(h) {} };
struct Scalar { double v[4]; Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : v{a, b, c, d} {} static Scalar all(double a) { return Scalar(a, a, a, a); } };
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };
struct DMatch {
  int queryIdx = -1, trainIdx = -1, imgIdx = -1;
  float distance = 0;
};
struct FileNode {};
struct FileStorage {};

class Mat {
 public:
  int rows = 0, cols = 0;
  std::shared_ptr<std::vector<uchar> > buf;
  uchar* data = nullptr;
  Mat() {}
  Mat(int r, int c, int /*type*/) { create(r, c); }
  Mat(int r, int c, int /*type*/, void* ext) : rows(r), cols(c), data(static_cast<uchar*>(ext)) {}   // (no copy, no ownership)
  void create(int r, int c) {
    rows = r; cols = c;
    buf = std::make_shared<std::vector<uchar> >((size_t)r * c);
    data = buf->data();
  }
  bool empty() const { return rows == 0 || cols == 0 || data == nullptr; }
  Mat clone() const {
    Mat m;
    m.create(rows, cols);
    if (!empty()) std::memcpy(m.data, data, (size_t)rows * cols);
    return m;
  }
  void release() { rows = cols = 0; buf.reset(); data = nullptr; }
  uchar* ptr(int r = 0) { return data + (size_t)r * cols; }
  const uchar* ptr(int r = 0) const { return data + (size_t)r * cols; }
  template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * cols); }
  template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * cols); }
  template <class T> T& at(int r, int c) { return reinterpret_cast<T*>(data + (size_t)r * cols)[c]; }
  template <class T> const T& at(int r, int c) const { return reinterpret_cast<const T*>(data + (size_t)r * cols)[c]; }
  template <class T> T& at(int i) { return reinterpret_cast<T*>(data)[i]; }
  template <class T> const T& at(int i) const { return reinterpret_cast<const T*>(data)[i]; }
  Mat row(int r) const { Mat m; m.rows = 1; m.cols = cols; m.buf = buf; m.data = data + (size_t)r * cols; return m; }
  void push_back(const Mat& o) {   // append rows
    if (o.empty()) return;
    Mat m;
    assert(empty() || cols == o.cols);
    m.create(rows + o.rows, o.cols);
    if (!empty()) std::memcpy(m.data, data, (size_t)rows * cols);
    std::memcpy(m.data + (size_t)rows * o.cols, o.data, (size_t)o.rows * o.cols);
    *this = m;
  }
  int type() const { return CV_8UC1; }
  Size size() const { return Size(cols, rows); }
  static Mat zeros(int r, int c, int t) { return Mat(r, c, t); }
  static Mat ones(int r, int c, int t) { Mat m(r, c, t); std::memset(m.data, 1, (size_t)r * c); return m; }
};
template <class T> struct Mat_ : Mat {};
typedef const Mat& InputArray;
typedef Mat& OutputArray;
typedef Mat& InputOutputArray;
typedef const std::vector<Mat>& InputArrayOfArrays;

inline Mat& noArray() { static Mat none; return none; }

class Algorithm {
 public:
  virtual ~Algorithm() {}
  virtual void clear() {}
  virtual void read(const FileNode&) {}
  virtual void write(FileStorage&) const {}
};

inline int cvRound(double v) { return (int)std::lrint(v); }

}  // namespace cv
using cv::cvRound;
