// Stand-in for the protobuf-generated voxblox::LayerProto (see Block.pb.h).  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <string>

#include <google/protobuf/message.h>
namespace voxblox {
class LayerProto : public google::protobuf::Message {
 public:
  float voxel_size() const { return 0.f; }
  int voxels_per_side() const { return 0; }
  std::string type() const { return std::string(); }
  void set_voxel_size(float) {}
  void set_voxels_per_side(int) {}
  void set_type(const std::string&) {}
};
}  // namespace voxblox
