// Shared host-side plumbing for the HIP translation units of libplvs_hip.so.
#pragma once
#include <algorithm>
#include <vector>
#include <thread>
#include <mutex>
#include <functional>
#include <condition_variable>
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
    // (geometric growth: a buffer sized by a call's input would otherwise be re-allocated by every call a little larger
    // than any before — a hipFree + hipMalloc of hundreds of MB costs milliseconds, on a fresh process far more)
    // Doubling only while the buffer is small: beyond 256 MB (the masks / records of a long batch run to gigabytes) the
    // next size is a quarter above the need or the old capacity — callers that know better size their request themselves.
    const bool large = cap * sizeof(T) > ((size_t)256 << 20);
    size_t want = std::max(n + n / 4 + 64, large ? cap + cap / 4 : 2 * cap);
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

// Host helper threads kept between calls (the line stage's routing / fitting threads, the ORB quadtree's level threads):
// creating them per frame cost the calling thread 15-25 us EACH before it could start its own share.
class HostPool {
 public:
  ~HostPool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      quit_ = true;
    }
    go_.notify_all();
    for (auto& t : th_) t.join();
  }
  // job(i) for i in [0, n) on n pool threads; the callable must stay alive until wait() returns
  void start(int n, const std::function<void(int)>* job) {
    std::unique_lock<std::mutex> lk(m_);
    while ((int)th_.size() < n) {
      const int idx = (int)th_.size();
      const uint64_t seen = gen_;
      th_.emplace_back([this, idx, seen]() { worker(idx, seen); });
    }
    job_ = job;
    njobs_ = n;
    pending_ = n;
    ++gen_;
    lk.unlock();
    go_.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [&] { return pending_ == 0; });
  }

 private:
  void worker(int idx, uint64_t seen) {
    for (;;) {
      const std::function<void(int)>* job;
      {
        std::unique_lock<std::mutex> lk(m_);
        go_.wait(lk, [&] { return quit_ || gen_ != seen; });
        if (quit_) return;
        seen = gen_;
        if (idx >= njobs_) continue;
        job = job_;
      }
      (*job)(idx);
      std::lock_guard<std::mutex> lk(m_);
      if (--pending_ == 0) done_.notify_all();
    }
  }
  std::mutex m_;
  std::condition_variable go_, done_;
  std::vector<std::thread> th_;
  const std::function<void(int)>* job_ = nullptr;
  uint64_t gen_ = 0;
  int njobs_ = 0, pending_ = 0;
  bool quit_ = false;
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
