// Stand-in for the protobuf-generated voxblox::BlockProto: the (de)serialisation members of Block / Layer name it, the
// integrator sources compiled into oracle/_ref never call them.  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <cstdint>
#include <vector>

#include <google/protobuf/message.h>
namespace voxblox {
class BlockProto : public google::protobuf::Message {
 public:
  int voxels_per_side() const { return 0; }
  float voxel_size() const { return 0.f; }
  float origin_x() const { return 0.f; }
  float origin_y() const { return 0.f; }
  float origin_z() const { return 0.f; }
  bool has_data() const { return false; }
  int voxel_data_size() const { return 0; }
  const std::vector<uint32_t>& voxel_data() const { return d_; }
  void set_voxels_per_side(int) {}
  void set_voxel_size(float) {}
  void set_origin_x(float) {}
  void set_origin_y(float) {}
  void set_origin_z(float) {}
  void set_has_data(bool) {}
  void add_voxel_data(uint32_t) {}
 private:
  std::vector<uint32_t> d_;
};
}  // namespace voxblox
