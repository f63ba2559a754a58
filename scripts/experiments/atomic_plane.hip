// Feasibility of an accumulator plane fed by device atomics instead of per-tile records: 15 000 tiles of
// 512 threads, two entries per thread, three atomics per entry (i64 add, u64 add, u32 max) onto 105 k voxels with the
// ~100-fold reuse the bench scene has (a tile's entries are neighbours: consecutive voxel ids from a per-tile base).
// Build: hipcc --offload-arch=gfx950 -O3 atomic_plane.hip -o atomic_plane ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
struct Acc { long long wuu; unsigned long long w; uint32_t last, pad; };
__global__ __launch_bounds__(512) void flush(Acc* plane, uint32_t nvox, int per) {
  const uint32_t tile = blockIdx.x;
  uint32_t base = (tile * 2654435761u) % (nvox - 2048u);   // tiles land on overlapping windows of the surface
  for (int k = 0; k < per; ++k) {
    const uint32_t v = base + ((threadIdx.x * 3u + k * 769u) & 1023u);
    atomicAdd(reinterpret_cast<unsigned long long*>(&plane[v].wuu), (unsigned long long)(threadIdx.x + 7));
    atomicAdd(&plane[v].w, (unsigned long long)((1ull << 32) | 5u));
    atomicMax(&plane[v].last, tile * 512u + threadIdx.x);
  }
}
int main() {
  const uint32_t nvox = 105648 * 4;   // the touched voxels sit in 96 chunks = 393 k slots
  Acc* plane;
  hipMalloc(&plane, sizeof(Acc) * nvox);
  hipMemset(plane, 0, sizeof(Acc) * nvox);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int per = 1; per <= 2; ++per) {
    flush<<<15000, 512>>>(plane, nvox, per);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) flush<<<15000, 512>>>(plane, nvox, per);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("entries per thread %d: %.3f ms per launch for %.1f M entries (3 atomics each)\n", per, ms / 10, 15000.0 * 512 * per / 1e6);
  }
  return 0;
}
