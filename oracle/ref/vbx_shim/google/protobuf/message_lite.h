// Empty stand-in: voxblox's layer_inl.h / protobuf_utils.h name google::protobuf types in the (de)serialisation
// members the integrator sources compiled into oracle/_ref never call.  TEST INFRASTRUCTURE ONLY.
#pragma once
namespace google {
namespace protobuf {
#ifndef VBX_SHIM_PROTOBUF_TYPES
#define VBX_SHIM_PROTOBUF_TYPES
class MessageLite {};
class Message : public MessageLite {};
#endif
}  // namespace protobuf
}  // namespace google
