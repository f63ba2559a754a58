// Error reporting, device selection and raw device-memory helpers of the C ABI.
#include "common.hpp"

namespace plvs {

char* last_error_buf() {
  static thread_local char buf[512] = "";
  return buf;
}

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), 512, fmt, ap);
  va_end(ap);
}

hipError_t HostStage::reserve(size_t bytes) {
  if (stream == nullptr) {
    hipError_t e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
    if (e != hipSuccess) return e;
  }
  if (bytes <= cap) return hipSuccess;
  const size_t want = bytes + bytes / 2 + 4096;
  if (pinned) (void)hipHostFree(pinned);
  if (dev) (void)hipFree(dev);
  pinned = dev = nullptr;
  cap = 0;
  hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&pinned), want);
  if (e != hipSuccess) return e;
  e = hipMalloc(reinterpret_cast<void**>(&dev), want);
  if (e != hipSuccess) return e;
  cap = want;
  return hipSuccess;
}

void HostStage::release() {
  if (pinned) (void)hipHostFree(pinned);
  if (dev) (void)hipFree(dev);
  if (stream) (void)hipStreamDestroy(stream);
  pinned = dev = nullptr;
  stream = nullptr;
  cap = 0;
}

HostStage& thread_stage() {
  static thread_local HostStage stage;   // no destructor on purpose: thread exit may come after the runtime's
  return stage;
}

}  // namespace plvs

extern "C" {

int plvs_hip_release_thread_buffers(void) {
  plvs::thread_stage().release();
  return PLVS_OK;
}

const char* plvs_hip_last_error(void) { return plvs::last_error_buf(); }

int plvs_hip_abi_version(void) { return 1; }

int plvs_hip_device_count(int* count) {
  PLVS_REQUIRE(count, "count is null");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    plvs::set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return PLVS_ERR_NO_DEVICE;
  }
  *count = n;
  return PLVS_OK;
}

int plvs_hip_set_device(int device) {
  PLVS_HIP_TRY(hipSetDevice(device));
  return PLVS_OK;
}

int plvs_hip_synchronize(void) {
  PLVS_HIP_TRY(hipDeviceSynchronize());
  return PLVS_OK;
}

int plvs_hip_malloc(void** dptr, size_t bytes) {
  PLVS_REQUIRE(dptr, "dptr is null");
  PLVS_HIP_TRY(hipMalloc(dptr, bytes ? bytes : 1));
  return PLVS_OK;
}

int plvs_hip_free(void* dptr) {
  if (dptr) PLVS_HIP_TRY(hipFree(dptr));
  return PLVS_OK;
}

int plvs_hip_memcpy_h2d(void* dst, const void* src, size_t bytes) {
  if (bytes) PLVS_HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return PLVS_OK;
}

int plvs_hip_memcpy_d2h(void* dst, const void* src, size_t bytes) {
  if (bytes) PLVS_HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return PLVS_OK;
}

int plvs_hip_memset(void* dst, int value, size_t bytes) {
  if (bytes) PLVS_HIP_TRY(hipMemset(dst, value, bytes));
  return PLVS_OK;
}

}  // extern "C"
