// No-op stand-in for glog: the checks of the sources compiled into oracle/_ref/libvoxblox_ref.so evaluate nothing and
// swallow their streamed messages.  TEST INFRASTRUCTURE ONLY.
#pragma once
struct VbxShimNullStream {
  template <class T>
  VbxShimNullStream& operator<<(const T&) { return *this; }
};
#define VBX_SHIM_SINK(cond) if (true) {} else VbxShimNullStream()
#define CHECK(c) VBX_SHIM_SINK(c)
#define DCHECK(c) VBX_SHIM_SINK(c)
#define CHECK_NOTNULL(p) (p)
#define CHECK_EQ(a, b) VBX_SHIM_SINK(0)
#define CHECK_NE(a, b) VBX_SHIM_SINK(0)
#define CHECK_LT(a, b) VBX_SHIM_SINK(0)
#define CHECK_LE(a, b) VBX_SHIM_SINK(0)
#define CHECK_GT(a, b) VBX_SHIM_SINK(0)
#define CHECK_GE(a, b) VBX_SHIM_SINK(0)
#define DCHECK_EQ(a, b) VBX_SHIM_SINK(0)
#define DCHECK_NE(a, b) VBX_SHIM_SINK(0)
#define DCHECK_LT(a, b) VBX_SHIM_SINK(0)
#define DCHECK_LE(a, b) VBX_SHIM_SINK(0)
#define DCHECK_GT(a, b) VBX_SHIM_SINK(0)
#define DCHECK_GE(a, b) VBX_SHIM_SINK(0)
#define LOG(severity) VbxShimNullStream()
#define LOG_FIRST_N(severity, n) VbxShimNullStream()
#define LOG_EVERY_N(severity, n) VbxShimNullStream()
#define VLOG(level) VbxShimNullStream()
