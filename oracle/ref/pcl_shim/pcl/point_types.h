// TEST INFRASTRUCTURE ONLY — a stand-in for the few PCL names the reference's include/PointSurfelSegment.h uses, so that the
// reference's OWN point type compiles here unmodified (oracle/ref/cloudgen_ref_wrap.cpp; PCL is in neither tree nor image).
// The macros restate PCL 1.x's pcl/impl/point_types.hpp member layouts (the Eigen map accessors they also add are left out:
// nothing on the compared path calls them):
//   PCL_ADD_NORMAL4D   union { float data_n[4]; float normal[3]; struct { float normal_x, normal_y, normal_z; }; }
//   PCL_ADD_UNION_RGB  union { union { struct { uint8_t b, g, r, a; }; float rgb; }; uint32_t rgba; }
#pragma once
#include <cstdint>
#include <ostream>
#ifndef EIGEN_ALIGN16
#define EIGEN_ALIGN16 alignas(16)
#endif
#ifndef EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#endif
#define PCL_ADD_EIGEN_MAPS_POINT4D
#define PCL_ADD_EIGEN_MAPS_NORMAL4D
#define PCL_ADD_EIGEN_MAPS_RGB
#define PCL_ADD_UNION_NORMAL4D \
  union EIGEN_ALIGN16 {        \
    float data_n[4];           \
    float normal[3];           \
    struct {                   \
      float normal_x;          \
      float normal_y;          \
      float normal_z;          \
    };                         \
  };
#define PCL_ADD_NORMAL4D PCL_ADD_UNION_NORMAL4D PCL_ADD_EIGEN_MAPS_NORMAL4D
#define PCL_ADD_UNION_RGB \
  union {                 \
    union {               \
      struct {            \
        std::uint8_t b;   \
        std::uint8_t g;   \
        std::uint8_t r;   \
        std::uint8_t a;   \
      };                  \
      float rgb;          \
    };                    \
    std::uint32_t rgba;   \
  };
// (the field list of a point type: PCL's reflection, not used here)
#define POINT_CLOUD_REGISTER_POINT_STRUCT(name, fseq)
