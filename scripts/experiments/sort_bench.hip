// The library's radix sort against rocPRIM's on the sizes of the TSDF chains (developer probe, not part of the product).
// Round 4, MI355X: 4 M pairs / 24 bits 0.136 ms (rocPRIM 0.138), 11 M pairs with 64-bit payload 0.43 (0.34), 16 M / 30 bits 0.53
// (0.55); outputs identical (both stable).  Two restructurings of the passes were tried against this probe and dropped:
// one kernel per digit with decoupled look-back between tiles (digit totals counted up front): 0.177 / 0.51 / 0.68 ms — with
// ~900 tiles resident a new tile's predecessors mostly hold only their own counts, and 256 threads walking back hundreds of
// words one dependent load at a time cost more than the two launches saved; two launches per digit with per-64-tile group
// sums read by the scatter kernel itself (no scan kernels): 0.130 / 0.53 / 0.65 ms — the reads grow with n^1.5.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I plvs_amd/csrc scripts/experiments/sort_bench.hip -o /tmp/sort_bench
#include "../../plvs_amd/csrc/device_utils.hip"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <random>

template <class TV>
static void bench(size_t n, int bits, const char* what) {
  std::vector<uint32_t> hk(n);
  std::mt19937 rng(7);
  for (auto& k : hk) k = rng() & ((bits >= 32) ? 0xFFFFFFFFu : ((1u << bits) - 1u));
  uint32_t *k0, *k1, *scratch;
  TV *v0, *v1;
  hipMalloc(&k0, n * 4); hipMalloc(&k1, n * 4); hipMalloc(&v0, n * sizeof(TV)); hipMalloc(&v1, n * sizeof(TV));
  hipMalloc(&scratch, plvs::radix_scratch_words(n) * 4);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms_own = 0, ms_rp = 0;
  const int reps = 10;
  for (int r = -2; r < reps; ++r) {
    hipMemcpyAsync(k0, hk.data(), n * 4, hipMemcpyHostToDevice, s);
    hipEventRecord(e0, s);
    bool second;
    if constexpr (sizeof(TV) == 4) plvs::radix_sort_pairs(k0, (uint32_t*)v0, k1, (uint32_t*)v1, n, 0, bits, scratch, s, &second);
    else plvs::radix_sort_pairs_u64(k0, (unsigned long long*)v0, k1, (unsigned long long*)v1, n, 0, bits, scratch, s, &second);
    hipEventRecord(e1, s); hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r >= 0) ms_own += ms;
  }
  // the last own run's output, to be compared with rocPRIM's (both sorts are stable: the outputs must be identical)
  std::vector<uint32_t> own_k(n);
  std::vector<TV> own_v(n);
  {
    std::vector<TV> hv(n);
    for (size_t i = 0; i < n; ++i) hv[i] = (TV)i;
    hipMemcpy(k0, hk.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(v0, hv.data(), n * sizeof(TV), hipMemcpyHostToDevice);
    bool second;
    if constexpr (sizeof(TV) == 4) plvs::radix_sort_pairs(k0, (uint32_t*)v0, k1, (uint32_t*)v1, n, 0, bits, scratch, s, &second);
    else plvs::radix_sort_pairs_u64(k0, (unsigned long long*)v0, k1, (unsigned long long*)v1, n, 0, bits, scratch, s, &second);
    hipStreamSynchronize(s);
    hipMemcpy(own_k.data(), second ? k1 : k0, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(own_v.data(), second ? v1 : v0, n * sizeof(TV), hipMemcpyDeviceToHost);
    hipMemcpy(v0, hv.data(), n * sizeof(TV), hipMemcpyHostToDevice);
  }
  size_t tmp_bytes = 0;
  rocprim::double_buffer<uint32_t> dk(k0, k1);
  rocprim::double_buffer<TV> dv(v0, v1);
  rocprim::radix_sort_pairs(nullptr, tmp_bytes, dk, dv, n, 0, bits, s);
  void* tmp; hipMalloc(&tmp, tmp_bytes);
  for (int r = -2; r < reps; ++r) {
    hipMemcpyAsync(k0, hk.data(), n * 4, hipMemcpyHostToDevice, s);
    rocprim::double_buffer<uint32_t> dk2(k0, k1);
    rocprim::double_buffer<TV> dv2(v0, v1);
    hipEventRecord(e0, s);
    rocprim::radix_sort_pairs(tmp, tmp_bytes, dk2, dv2, n, 0, bits, s);
    hipEventRecord(e1, s); hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r >= 0) ms_rp += ms;
  }
  bool same = true;
  {
    std::vector<TV> hv(n);
    for (size_t i = 0; i < n; ++i) hv[i] = (TV)i;
    hipMemcpy(k0, hk.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(v0, hv.data(), n * sizeof(TV), hipMemcpyHostToDevice);
    rocprim::double_buffer<uint32_t> dk3(k0, k1);
    rocprim::double_buffer<TV> dv3(v0, v1);
    rocprim::radix_sort_pairs(tmp, tmp_bytes, dk3, dv3, n, 0, bits, s);
    hipStreamSynchronize(s);
    std::vector<uint32_t> rk(n);
    std::vector<TV> rv(n);
    hipMemcpy(rk.data(), dk3.current(), n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(rv.data(), dv3.current(), n * sizeof(TV), hipMemcpyDeviceToHost);
    for (size_t i = 0; i < n && same; ++i) same = rk[i] == own_k[i] && rv[i] == own_v[i];
  }
  printf("%-40s n %9zu bits %2d value %zu B: own %.3f ms   rocprim %.3f ms (temp %zu KB)  outputs %s\n", what, n, bits, sizeof(TV),
         ms_own / reps, ms_rp / reps, tmp_bytes >> 10, same ? "identical" : "DIFFER");
  hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(scratch); hipFree(tmp);
}

int main() {
  bench<uint32_t>(4000000, 24, "chisel order-free runs (stream)");
  bench<uint32_t>(2000000, 24, "chisel order-free runs");
  bench<uint32_t>(300000, 24, "a 5-key-frame call");
  bench<unsigned long long>(11000000, 24, "chisel ordered (voxel key, u64 payload)");
  bench<uint32_t>(16000000, 30, "voxblox expand");
  return 0;
}
