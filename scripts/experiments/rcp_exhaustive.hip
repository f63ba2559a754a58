// Exhaustive check on the device: for every binary32 significand, is
//   y1 = fma(fma(-b, y0, 1), y0, y0),  y0 = v_rcp_f32(b)
// the correctly rounded reciprocal RN(1/b)?  Prints the number of significands for which
// it is not, and the first few.  (Experiment backing the chain kernel's reciprocal.)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void check(int exponent, unsigned* count, unsigned* first, int cap) {
  const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= (1u << 23)) return;
  const float b = __uint_as_float(((uint32_t)(exponent + 127) << 23) | m);
  const float y0 = __builtin_amdgcn_rcpf(b);
  const float e = fmaf(-b, y0, 1.0f);
  const float y1 = fmaf(e, y0, y0);
  const float ref = 1.0f / b;
  if (__float_as_uint(y1) != __float_as_uint(ref)) {
    const unsigned k = atomicAdd(count, 1u);
    if ((int)k < cap) first[k] = m;
  }
}

int main() {
  unsigned *d_count, *d_first, h_count, h_first[16];
  hipMalloc(&d_count, 4);
  hipMalloc(&d_first, 64);
  for (int e : {0, 1, -20, 35}) {
    hipMemset(d_count, 0, 4);
    hipLaunchKernelGGL(check, dim3((1u << 23) / 256), dim3(256), 0, 0, e, d_count, d_first, 16);
    hipMemcpy(&h_count, d_count, 4, hipMemcpyDeviceToHost);
    hipMemcpy(h_first, d_first, 64, hipMemcpyDeviceToHost);
    printf("exponent %d: %u significands not correctly rounded;", e, h_count);
    for (unsigned i = 0; i < h_count && i < 16; ++i) printf(" 0x%06x", h_first[i]);
    printf("\n");
  }
  return 0;
}
