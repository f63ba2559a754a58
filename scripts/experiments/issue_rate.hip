// Single-wave issue-rate probe for gfx950: cycles per instruction of dependent and
// independent VALU chains, with and without interleaved SALU work, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr int N = 2048;

__global__ void dep_fma(float* out, long long* cyc, float a, float b) {
  float x = out[threadIdx.x];
  const long long t0 = clock64();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) x = fmaf(x, a, b);
  __builtin_amdgcn_sched_barrier(0);
  const long long t1 = clock64();
  __builtin_amdgcn_sched_barrier(0);
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void indep4_fma(float* out, long long* cyc, float a, float b) {
  float x0 = out[threadIdx.x], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  const long long t0 = clock64();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N / 4; ++i) {
    x0 = fmaf(x0, a, b);
    x1 = fmaf(x1, a, b);
    x2 = fmaf(x2, a, b);
    x3 = fmaf(x3, a, b);
  }
  __builtin_amdgcn_sched_barrier(0);
  const long long t1 = clock64();
  __builtin_amdgcn_sched_barrier(0);
  out[threadIdx.x] = x0 + x1 + x2 + x3;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// the sdf recurrence of the chain kernel: mul, add, mul, fma, fma per step
__global__ void dep_step5(float* out, long long* cyc, float w, float x, float y, float wn) {
  float s = out[threadIdx.x];
  const long long t0 = clock64();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N / 4; ++i) {
    const float a = w * s + x;
    const float q = a * y;
    const float r = fmaf(-q, wn, a);
    s = fmaf(r, y, q);
  }
  __builtin_amdgcn_sched_barrier(0);
  const long long t1 = clock64();
  __builtin_amdgcn_sched_barrier(0);
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// same with a compare + scalar AND per step (the range guard as first written)
__global__ void dep_step5_guard(float* out, long long* cyc, float w, float x, float y, float wn) {
  float s = out[threadIdx.x];
  bool ok = true;
  const long long t0 = clock64();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N / 4; ++i) {
    const float a = w * s + x;
    const float q = a * y;
    const float r = fmaf(-q, wn, a);
    s = fmaf(r, y, q);
    ok = ok & (fabsf(a) >= 0x1p-60f) & (fabsf(a) <= 0x1p60f);
  }
  __builtin_amdgcn_sched_barrier(0);
  const long long t1 = clock64();
  __builtin_amdgcn_sched_barrier(0);
  out[threadIdx.x] = ok ? s : 0.f;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void dep_rcp(float* out, long long* cyc) {
  float x = out[threadIdx.x];
  const long long t0 = clock64();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N / 4; ++i) x = __builtin_amdgcn_rcpf(x);
  __builtin_amdgcn_sched_barrier(0);
  const long long t1 = clock64();
  __builtin_amdgcn_sched_barrier(0);
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void dep_div(float* out, long long* cyc, float d) {
  float x = out[threadIdx.x];
  const long long t0 = clock64();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N / 16; ++i) x = x / d + d;
  __builtin_amdgcn_sched_barrier(0);
  const long long t1 = clock64();
  __builtin_amdgcn_sched_barrier(0);
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// row_shr:1 ripple step: dpp mov + 5 ops
__global__ void dep_dpp(float* out, long long* cyc, float w, float x, float y, float wn) {
  float s = out[threadIdx.x];
  const float carry = s;
  const long long t0 = clock64();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N / 8; ++i) {
    const float sp = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(carry), __float_as_int(s), 0x111, 0xF, 0xF, false));
    const float a = w * sp + x;
    const float q = a * y;
    const float r = fmaf(-q, wn, a);
    s = fmaf(r, y, q);
  }
  __builtin_amdgcn_sched_barrier(0);
  const long long t1 = clock64();
  __builtin_amdgcn_sched_barrier(0);
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  float* out;
  long long* cyc;
  hipMalloc(&out, 4096 * 4);
  hipHostMalloc(&cyc, 8);
  hipMemset(out, 0, 4096 * 4);
  for (int threads : {64, 256, 512, 1024}) {
    printf("block of %d threads (%d wave(s) per SIMD):\n", threads, (threads + 255) / 256);
#define RUN(name, n, ...)                                                        \
  for (int rep = 0; rep < 2; ++rep) {                                            \
    hipLaunchKernelGGL(name, dim3(1), dim3(threads), 0, 0, out, cyc, ##__VA_ARGS__); \
    hipDeviceSynchronize();                                                      \
  }                                                                              \
  printf("  %-18s %8lld cycles / %5d instr = %6.2f cyc/instr\n", #name, cyc[0], n, (double)cyc[0] / n);
    RUN(dep_fma, N, 1.0001f, 0.5f)
    RUN(indep4_fma, N, 1.0001f, 0.5f)
    RUN(dep_step5, N / 4 * 5, 3.0f, 0.1f, 0.25f, 4.0f)
    RUN(dep_step5_guard, N / 4 * 5, 3.0f, 0.1f, 0.25f, 4.0f)
    RUN(dep_rcp, N / 4)
    RUN(dep_div, N / 16, 3.0f)
    RUN(dep_dpp, N / 8 * 6, 3.0f, 0.1f, 0.25f, 4.0f)
  }
  return 0;
}
