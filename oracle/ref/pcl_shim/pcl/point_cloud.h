// TEST INFRASTRUCTURE ONLY — pcl::PointCloud as far as PointCloudMapping::GeneratePointCloudInCameraFrameBGRA touches it
// (src/PointCloudMapping.cc:929-1226): points, reserve, is_dense, header.stamp, Ptr.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>
namespace pcl {
struct PCLHeader {
  std::uint32_t seq = 0;
  std::uint64_t stamp = 0;
};
template <class PointT>
struct PointCloud {
  typedef std::shared_ptr<PointCloud<PointT> > Ptr;
  PCLHeader header;
  std::vector<PointT> points;
  std::uint32_t width = 0, height = 0;
  bool is_dense = true;
  void reserve(size_t n) { points.reserve(n); }
  size_t size() const { return points.size(); }
};
}  // namespace pcl
