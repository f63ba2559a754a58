// The library's radix sort against rocPRIM's on the sizes of the TSDF chains (developer probe, not part of the product):
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
  printf("%-40s n %9zu bits %2d value %zu B: own %.3f ms   rocprim %.3f ms (temp %zu KB)\n", what, n, bits, sizeof(TV), ms_own / reps,
         ms_rp / reps, tmp_bytes >> 10);
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
