// Shared host-side plumbing for the HIP translation units of libplvs_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/plvs_hip.h"

namespace plvs {

// Thread-local last-error string behind plvs_hip_last_error().
char* last_error_buf();
void set_error(const char* fmt, ...);

struct HipFailure {
  hipError_t err;
};

}  // namespace plvs

// Evaluate a HIP call; on failure record it and return PLVS_ERR_HIP from the
// enclosing extern "C" function.
#define PLVS_HIP_TRY(call)                                                              \
  do {                                                                                  \
    hipError_t _e = (call);                                                             \
    if (_e != hipSuccess) {                                                             \
      plvs::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, \
                      __LINE__);                                                        \
      return PLVS_ERR_HIP;                                                              \
    }                                                                                   \
  } while (0)

#define PLVS_KERNEL_CHECK() PLVS_HIP_TRY(hipGetLastError())

#define PLVS_REQUIRE(cond, msg)                 \
  do {                                          \
    if (!(cond)) {                              \
      plvs::set_error("invalid argument: %s", msg); \
      return PLVS_ERR_INVALID_ARG;              \
    }                                           \
  } while (0)

namespace plvs {

// Growable device buffer owned by a handle (never shrinks).
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    size_t want = n + n / 4 + 64;
    if (p) {
      hipError_t e = hipFree(p);
      p = nullptr;
      cap = 0;
      if (e != hipSuccess) return e;
    }
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(T));
    if (e != hipSuccess) return e;
    cap = want;
    return hipSuccess;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// Per-thread staging for the host flavours of small calls (k-NN / candidate distances of a few hundred
// descriptors): one pinned host block, one device block and a stream, kept between calls — a call is
// one host-to-device copy, the kernel, one device-to-host copy and one stream wait instead of a
// hipMalloc / hipFree / blocking pageable copy per array.  Released by plvs_hip_release_thread_buffers()
// (or never: a thread that exits leaves a few hundred KB behind).
struct HostStage {
  char* pinned = nullptr;
  char* dev = nullptr;
  size_t cap = 0;
  hipStream_t stream = nullptr;
  hipError_t reserve(size_t bytes);
  void release();
};
HostStage& thread_stage();

static inline unsigned ceil_div(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// Integer tuning knob from the environment (host thread counts), clamped to [lo, hi].
static inline int env_int(const char* name, int dflt, int lo, int hi) {
  const char* v = getenv(name);
  if (!v || !*v) return dflt;
  const long x = strtol(v, nullptr, 10);
  return (int)(x < lo ? lo : (x > hi ? hi : x));
}

}  // namespace plvs
