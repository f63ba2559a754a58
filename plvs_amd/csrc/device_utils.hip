// Device-wide exclusive scan and stable LSD radix sort (internal primitives).
//
// Both are written for 64-lane wavefronts: wave-level prefix sums use
// __shfl_up over 64 lanes, and the radix scatter ranks equal digits inside a
// wavefront with eight __ballot masks (a "match-any" over the 8-bit digit),
// which is what keeps the sort stable without any per-item atomics.
#include "device_utils.hpp"

#include "common.hpp"

namespace plvs {
namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;

// ------------------------------------------------------------------ scan
constexpr int kScanItems = 8;
constexpr int kScanTile = kThreads * kScanItems;  // 2048

// Exclusive prefix of v over the 256 threads of the block; *total = block sum.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* total,
                                                         uint32_t* lds /* >= kWaves words */) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) lds[wid] = x;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) {
    const uint32_t s = lds[w];
    if (w < wid) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + x - v;
}

__global__ __launch_bounds__(kThreads) void scan_tile_sums(const uint32_t* __restrict__ in, size_t n,
                                                           uint32_t* __restrict__ sums) {
  __shared__ uint32_t lds[kWaves];
  const size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanItems;
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i)
    if (base + i < n) s += in[base + i];
  uint32_t tot;
  (void)block_exclusive_scan(s, &tot, lds);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// Single block: exclusive scan of sums[0..nb) in place, grand total to *total.
__global__ __launch_bounds__(kThreads) void scan_sums(uint32_t* __restrict__ sums, size_t nb,
                                                      uint32_t* __restrict__ total) {
  __shared__ uint32_t lds[kWaves];
  uint32_t carry = 0;
  for (size_t start = 0; start < nb; start += kScanTile) {
    const size_t base = start + (size_t)threadIdx.x * kScanItems;
    uint32_t v[kScanItems];
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
      v[i] = (base + i < nb) ? sums[base + i] : 0u;
      s += v[i];
    }
    uint32_t tot;
    uint32_t ex = block_exclusive_scan(s, &tot, lds) + carry;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
      if (base + i < nb) sums[base + i] = ex;
      ex += v[i];
    }
    carry += tot;
  }
  if (threadIdx.x == 0 && total != nullptr) *total = carry;
}

__global__ __launch_bounds__(kThreads) void scan_tile_apply(const uint32_t* __restrict__ in,
                                                            uint32_t* __restrict__ out, size_t n,
                                                            const uint32_t* __restrict__ sums) {
  __shared__ uint32_t lds[kWaves];
  const size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanItems;
  uint32_t v[kScanItems];
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    v[i] = (base + i < n) ? in[base + i] : 0u;
    s += v[i];
  }
  uint32_t tot;
  uint32_t ex = block_exclusive_scan(s, &tot, lds) + sums[blockIdx.x];
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    if (base + i < n) out[base + i] = ex;
    ex += v[i];
  }
}

// Short inputs: one workgroup of 1024 threads walks the array in spans of 8192 with a running carry — one launch
// instead of three (a dependent launch costs more than such a span takes).
constexpr int kSingleThreads = 1024;
constexpr int kSingleSpan = kSingleThreads * kScanItems;   // 8192
constexpr size_t kSingleMax = 6 * (size_t)kSingleSpan;      // 49152 elements

__global__ __launch_bounds__(kSingleThreads) void scan_single(const uint32_t* in, uint32_t* out,   // (in == out is allowed)
                                                              size_t n, uint32_t* __restrict__ total) {
  // Every span's loads are issued before anything is summed, and one barrier serves all spans: span after span — load,
  // scan, barrier, store — a six-span scan took 11-17 us, the latency of six dependent global loads and twelve barriers.
  constexpr int kSpans = (int)(kSingleMax / kSingleSpan);
  __shared__ uint32_t wsum[kSpans][kSingleThreads / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int spans = (int)((n + kSingleSpan - 1) / kSingleSpan);
  uint32_t v[kSpans][kScanItems], x[kSpans], s[kSpans];
  // (a thread's eight words as two 16-byte loads where they are whole and aligned: eight 4-byte loads at a 32-byte stride
  // across the lanes were what the kernel spent its time on)
  static_assert(kScanItems == 8, "two uint4 per thread and span");
  const bool vec = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0;
#pragma unroll
  for (int k = 0; k < kSpans; ++k) {
    const size_t base = (size_t)k * kSingleSpan + (size_t)threadIdx.x * kScanItems;
    if (vec && k < spans && base + kScanItems <= n) {
      const uint4 a = reinterpret_cast<const uint4*>(in + base)[0], c = reinterpret_cast<const uint4*>(in + base)[1];
      v[k][0] = a.x; v[k][1] = a.y; v[k][2] = a.z; v[k][3] = a.w;
      v[k][4] = c.x; v[k][5] = c.y; v[k][6] = c.z; v[k][7] = c.w;
    } else {
#pragma unroll
      for (int i = 0; i < kScanItems; ++i) v[k][i] = (k < spans && base + i < n) ? in[base + i] : 0u;
    }
  }
#pragma unroll
  for (int k = 0; k < kSpans; ++k) {
    uint32_t t = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) t += v[k][i];
    s[k] = t;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = __shfl_up(t, d, 64);
      if (lane >= d) t += y;
    }
    x[k] = t;
    if (lane == 63) wsum[k][wid] = t;
  }
  __syncthreads();
  uint32_t carry = 0;
#pragma unroll
  for (int k = 0; k < kSpans; ++k) {
    if (k >= spans) break;
    uint32_t wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kSingleThreads / 64; ++w) {
      const uint32_t t = wsum[k][w];
      if (w < wid) wbase += t;
      tot += t;
    }
    const size_t base = (size_t)k * kSingleSpan + (size_t)threadIdx.x * kScanItems;
    uint32_t ex = carry + wbase + x[k] - s[k];
    if (vec && base + kScanItems <= n) {
      uint32_t e[kScanItems];
#pragma unroll
      for (int i = 0; i < kScanItems; ++i) {
        e[i] = ex;
        ex += v[k][i];
      }
      reinterpret_cast<uint4*>(out + base)[0] = make_uint4(e[0], e[1], e[2], e[3]);
      reinterpret_cast<uint4*>(out + base)[1] = make_uint4(e[4], e[5], e[6], e[7]);
    } else {
#pragma unroll
      for (int i = 0; i < kScanItems; ++i) {
        if (base + i < n) out[base + i] = ex;
        ex += v[k][i];
      }
    }
    carry += tot;
  }
  if (threadIdx.x == 0 && total != nullptr) *total = carry;
}

// ------------------------------------------------------------ radix sort
constexpr int kMaxRadixBits = 11;
constexpr int kSortItems = 16;                       // per thread
constexpr int kSortTile = kThreads * kSortItems;     // 4096 keys per block
constexpr int kWaveSpan = 64 * kSortItems;           // 1024 consecutive keys per wave
#ifndef PLVS_LDS_SCATTER_MIN
#define PLVS_LDS_SCATTER_MIN (1u << 20)
#endif
constexpr size_t kLdsScatterMin = PLVS_LDS_SCATTER_MIN;   // shorter arrays: fewer, wider passes (launch-bound)

// hist[d * nb + b] = number of keys of tile b whose digit is d.
template <int kBits>
__global__ __launch_bounds__(kThreads) void radix_hist(const uint32_t* __restrict__ keys, size_t n,
                                                       int shift, size_t nb,
                                                       uint32_t* __restrict__ hist) {
  constexpr int kRadix = 1 << kBits;
  constexpr uint32_t kMask = kRadix - 1;
  __shared__ uint32_t h[kRadix];
  for (int d = threadIdx.x; d < kRadix; d += kThreads) h[d] = 0;
  __syncthreads();
  const size_t tile = (size_t)blockIdx.x * kSortTile;
#pragma unroll 4
  for (int it = 0; it < kSortItems; ++it) {
    const size_t i = tile + (size_t)it * kThreads + threadIdx.x;
    if (i < n) atomicAdd(&h[(keys[i] >> shift) & kMask], 1u);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < kRadix; d += kThreads) hist[(size_t)d * nb + blockIdx.x] = h[d];
}

template <int kBits, typename TV>
__global__ __launch_bounds__(kThreads) void radix_scatter(
    const uint32_t* __restrict__ keys_in, const TV* __restrict__ vals_in,
    uint32_t* __restrict__ keys_out, TV* __restrict__ vals_out, size_t n, int shift,
    size_t nb, const uint32_t* __restrict__ hist_scanned) {
  constexpr int kRadix = 1 << kBits;
  constexpr uint32_t kMask = kRadix - 1;
  // wave_hist[w][d]: running count of digit d inside wave w's 1024-key span,
  // later turned into the wave's exclusive base inside the tile.
  __shared__ uint32_t wave_hist[kWaves][kRadix];
  __shared__ uint32_t gbase[kRadix];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int d = threadIdx.x; d < kRadix; d += kThreads) {
#pragma unroll
    for (int w = 0; w < kWaves; ++w) wave_hist[w][d] = 0;
    gbase[d] = hist_scanned[(size_t)d * nb + blockIdx.x];
  }
  __syncthreads();

  const size_t span = (size_t)blockIdx.x * kSortTile + (size_t)wid * kWaveSpan;
  uint32_t k[kSortItems], rank[kSortItems];
  TV v[kSortItems];
  volatile uint32_t* my_hist = wave_hist[wid];
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int it = 0; it < kSortItems; ++it) {
    const size_t i = span + (size_t)it * 64 + lane;
    const bool valid = i < n;
    k[it] = valid ? keys_in[i] : 0u;
    v[it] = valid ? vals_in[i] : TV(0);
    const uint32_t d = (k[it] >> shift) & kMask;
    // lanes holding the same digit (and a valid key)
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < kBits; ++b) {
      const unsigned long long bal = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? bal : ~bal;
    }
    const uint32_t before = (uint32_t)__popcll(peers & lt_mask);
    const uint32_t cnt = (uint32_t)__popcll(peers);
    uint32_t base = 0;
    if (valid) base = my_hist[d];
    rank[it] = base + before;
    // the lowest peer publishes the new running count; LDS ops of one wave
    // execute in order, so every peer has read `base` before this store lands.
    if (valid && before == 0) my_hist[d] = base + cnt;
  }
  __syncthreads();
  // exclusive prefix over the waves, per digit
  for (int d = threadIdx.x; d < kRadix; d += kThreads) {
    uint32_t run = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      const uint32_t c = wave_hist[w][d];
      wave_hist[w][d] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kSortItems; ++it) {
    const size_t i = span + (size_t)it * 64 + lane;
    if (i < n) {
      const uint32_t d = (k[it] >> shift) & kMask;
      const size_t pos = (size_t)gbase[d] + wave_hist[wid][d] + rank[it];
      keys_out[pos] = k[it];
      vals_out[pos] = v[it];
    }
  }
}

template <int kBits, typename TV>
hipError_t radix_pass(const uint32_t* ki, const TV* vi, uint32_t* ko, TV* vo, size_t n,
                      int shift, size_t nb, uint32_t* hist, uint32_t* scan_scratch,
                      hipStream_t stream) {
  constexpr size_t kRadix = (size_t)1 << kBits;
  hipLaunchKernelGGL(radix_hist<kBits>, dim3((unsigned)nb), dim3(kThreads), 0, stream, ki, n, shift,
                     nb, hist);
  hipError_t e = exclusive_scan_u32(hist, hist, kRadix * nb, nullptr, scan_scratch, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((radix_scatter<kBits, TV>), dim3((unsigned)nb), dim3(kThreads), 0, stream, ki, vi, ko,
                     vo, n, shift, nb, hist);
  return hipGetLastError();
}

// The scatter of a pass over 8-bit digits with the tile put in digit order in LDS first: the keys of a digit leave the
// tile as ONE contiguous piece (16 keys on average for a 4096-key tile), written by consecutive lanes — the
// lane-per-key scatter above sends every 4-byte store to a line of its own (0.2 ms per pass for 11 M pairs).
// Stability as above: a key's place inside its digit's piece = keys of the digit in earlier waves + in earlier trips of
// this wave + in lower lanes of this trip.
template <typename TV>
__global__ __launch_bounds__(kThreads) void radix_scatter_lds(
    const uint32_t* __restrict__ keys_in, const TV* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
    TV* __restrict__ vals_out, size_t n, int shift, size_t nb, const uint32_t* __restrict__ hist_scanned) {
  constexpr int kRadix = 256;
  __shared__ uint32_t wave_hist[kWaves][kRadix];
  __shared__ uint32_t gbase[kRadix], lbase[kRadix];
  __shared__ uint32_t s_key[kSortTile];
  __shared__ TV s_val[kSortTile];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  {
    const int d = threadIdx.x;   // kThreads == kRadix
#pragma unroll
    for (int w = 0; w < kWaves; ++w) wave_hist[w][d] = 0;
    gbase[d] = hist_scanned[(size_t)d * nb + blockIdx.x];
  }
  __syncthreads();
  const size_t tile0 = (size_t)blockIdx.x * kSortTile;
  const size_t span = tile0 + (size_t)wid * kWaveSpan;
  const uint32_t ntile = (uint32_t)(n - tile0 < (size_t)kSortTile ? n - tile0 : (size_t)kSortTile);
  uint32_t k[kSortItems], rank[kSortItems];
  TV v[kSortItems];
  volatile uint32_t* my_hist = wave_hist[wid];
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int it = 0; it < kSortItems; ++it) {
    const size_t i = span + (size_t)it * 64 + lane;
    const bool valid = i < n;
    k[it] = valid ? keys_in[i] : 0u;
    v[it] = valid ? vals_in[i] : TV(0);
    const uint32_t d = (k[it] >> shift) & 255u;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long bal = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? bal : ~bal;
    }
    const uint32_t before = (uint32_t)__popcll(peers & lt_mask);
    const uint32_t cnt = (uint32_t)__popcll(peers);
    uint32_t base = 0;
    if (valid) base = my_hist[d];
    rank[it] = base + before;
    if (valid && before == 0) my_hist[d] = base + cnt;
  }
  __syncthreads();
  uint32_t tot;
  {   // per digit: exclusive prefix over the waves, the tile's count; then the digits' places inside the tile
    const int d = threadIdx.x;
    uint32_t run = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      const uint32_t c = wave_hist[w][d];
      wave_hist[w][d] = run;
      run += c;
    }
    uint32_t* scratch = s_key;   // (not yet in use)
    const uint32_t ex = block_exclusive_scan(run, &tot, scratch);
    lbase[d] = ex;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kSortItems; ++it) {
    const size_t i = span + (size_t)it * 64 + lane;
    if (i < n) {
      const uint32_t d = (k[it] >> shift) & 255u;
      const uint32_t p = lbase[d] + wave_hist[wid][d] + rank[it];
      s_key[p] = k[it];
      s_val[p] = v[it];
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kSortItems; ++it) {
    const uint32_t i = (uint32_t)(it * kThreads + threadIdx.x);
    if (i < ntile) {
      const uint32_t key = s_key[i];
      const uint32_t d = (key >> shift) & 255u;
      const size_t pos = (size_t)gbase[d] + (i - lbase[d]);
      keys_out[pos] = key;
      vals_out[pos] = s_val[i];
    }
  }
}

template <typename TV>
hipError_t radix_pass_lds(const uint32_t* ki, const TV* vi, uint32_t* ko, TV* vo, size_t n, int shift, size_t nb,
                          uint32_t* hist, uint32_t* scan_scratch, hipStream_t stream) {
  hipLaunchKernelGGL(radix_hist<8>, dim3((unsigned)nb), dim3(kThreads), 0, stream, ki, n, shift, nb, hist);
  hipError_t e = exclusive_scan_u32(hist, hist, (size_t)256 * nb, nullptr, scan_scratch, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((radix_scatter_lds<TV>), dim3((unsigned)nb), dim3(kThreads), 0, stream, ki, vi, ko, vo, n, shift, nb, hist);
  return hipGetLastError();
}

// ---- one launch per pass (round 5, late): the chained scan inside the scatter
// A pass of radix_pass_lds is five launches — tile histograms, a three-kernel scan of 256 x tiles counters, the scatter — and on
// the arrays the TSDF pipelines sort (1-20 M pairs) the first four cost as much as the scatter itself (13 + 17 of 58 us per pass
// at 2.5 M pairs, plus four launch gaps on a chain of dependent kernels).  Here the digit totals of EVERY pass come from one
// read of the keys up front (radix_digit_totals: a permutation does not change them), and a pass is one kernel: a tile takes
// a ticket, ranks its keys as radix_scatter_lds does, publishes its 256 digit counts as "aggregate" words, looks back over
// the tiles before it — thread d for digit d, adding aggregates until it meets a tile that has published its inclusive
// prefix — publishes its own prefix, and scatters.  A status word carries its flag and its count together (2 + 30 bits:
// one atomic word, no fence between a flag and a payload); tickets are taken in launch order, so every tile a block waits for
// is already running.  Same stable order as the three-kernel pass (tests: plvs_hip_selftest_radix_sort against a host sort).
constexpr int kOsMaxPasses = 4;
constexpr uint32_t kOsAggregate = 1u << 30, kOsPrefix = 2u << 30, kOsCount = (1u << 30) - 1u;

// (n_dev, here and in radix_onesweep: the array holds min(n, *n_dev) pairs — a sort launched on a BOUND of their number, before
// the host knows it; the tiles behind them leave at once)
__global__ __launch_bounds__(kThreads) void radix_digit_totals(const uint32_t* __restrict__ keys, size_t n, int bit_lo, int passes,
                                                               size_t nb, uint32_t* __restrict__ totals /* [passes][256], zeroed */,
                                                               const uint32_t* __restrict__ n_dev) {
  if (n_dev != nullptr) {
    n = n < (size_t)*n_dev ? n : (size_t)*n_dev;
    nb = (n + kSortTile - 1) / kSortTile;
  }
  __shared__ uint32_t h[kOsMaxPasses][256];
  for (int p = 0; p < passes; ++p) h[p][threadIdx.x] = 0;
  __syncthreads();
  const bool vec = (reinterpret_cast<uintptr_t>(keys) & 15u) == 0;
  for (size_t b = blockIdx.x; b < nb; b += gridDim.x) {
    const size_t tile = b * kSortTile;
    if (vec && tile + kSortTile <= n) {   // (a whole tile: four 16-byte loads per thread, all in flight before the first atomic)
      uint4 q[kSortItems / 4];
#pragma unroll
      for (int it = 0; it < kSortItems / 4; ++it)
        q[it] = reinterpret_cast<const uint4*>(keys + tile)[it * kThreads + threadIdx.x];
#pragma unroll
      for (int it = 0; it < kSortItems / 4; ++it) {
        const uint32_t k4[4] = {q[it].x >> bit_lo, q[it].y >> bit_lo, q[it].z >> bit_lo, q[it].w >> bit_lo};
#pragma unroll
        for (int c = 0; c < 4; ++c)
          for (int p = 0; p < passes; ++p) atomicAdd(&h[p][(k4[c] >> (8 * p)) & 255u], 1u);
      }
      continue;
    }
#pragma unroll 4
    for (int it = 0; it < kSortItems; ++it) {
      const size_t i = tile + (size_t)it * kThreads + threadIdx.x;
      if (i < n) {
        const uint32_t k = keys[i] >> bit_lo;
        for (int p = 0; p < passes; ++p) atomicAdd(&h[p][(k >> (8 * p)) & 255u], 1u);
      }
    }
  }
  __syncthreads();
  for (int p = 0; p < passes; ++p) {
    const uint32_t c = h[p][threadIdx.x];
    if (c) atomicAdd(&totals[p * 256 + threadIdx.x], c);
  }
}

template <typename TV>
__global__ __launch_bounds__(kThreads) void radix_onesweep(
    const uint32_t* __restrict__ keys_in, const TV* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
    TV* __restrict__ vals_out, size_t n, int shift, const uint32_t* __restrict__ totals /* [256] of this pass */,
    uint32_t* __restrict__ status /* [tiles][256], zeroed */, uint32_t* __restrict__ ticket /* zeroed */,
    const uint32_t* __restrict__ n_dev) {
  constexpr int kRadix = 256;
  if (n_dev != nullptr) n = n < (size_t)*n_dev ? n : (size_t)*n_dev;
  __shared__ uint32_t wave_hist[kWaves][kRadix];
  __shared__ uint32_t gbase[kRadix], lbase[kRadix];
  __shared__ uint32_t s_key[kSortTile];
  __shared__ TV s_val[kSortTile];
  __shared__ uint32_t s_tile;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int d = threadIdx.x;   // kThreads == kRadix
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
#pragma unroll
  for (int w = 0; w < kWaves; ++w) wave_hist[w][d] = 0;
  __syncthreads();
  const uint32_t b = s_tile;
  const size_t tile0 = (size_t)b * kSortTile;
  if (tile0 >= n) return;   // (a launch on a bound: tickets are taken in order, so every tile that holds pairs has a smaller one)
  const size_t span = tile0 + (size_t)wid * kWaveSpan;
  const uint32_t ntile = (uint32_t)(n - tile0 < (size_t)kSortTile ? n - tile0 : (size_t)kSortTile);
  uint32_t k[kSortItems], rank[kSortItems];
  TV v[kSortItems];
  volatile uint32_t* my_hist = wave_hist[wid];
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int it = 0; it < kSortItems; ++it) {
    const size_t i = span + (size_t)it * 64 + lane;
    const bool valid = i < n;
    k[it] = valid ? keys_in[i] : 0u;
    v[it] = valid ? vals_in[i] : TV(0);
    const uint32_t dg = (k[it] >> shift) & 255u;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const unsigned long long bal = __ballot((dg >> bit) & 1u);
      peers &= ((dg >> bit) & 1u) ? bal : ~bal;
    }
    const uint32_t before = (uint32_t)__popcll(peers & lt_mask);
    const uint32_t cnt = (uint32_t)__popcll(peers);
    uint32_t base = 0;
    if (valid) base = my_hist[dg];
    rank[it] = base + before;
    if (valid && before == 0) my_hist[dg] = base + cnt;
  }
  __syncthreads();
  {   // digit d: the tile's count -> aggregate out, look back, prefix out; the digits' places inside the tile
    uint32_t run = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      const uint32_t c = wave_hist[w][d];
      wave_hist[w][d] = run;
      run += c;
    }
    uint32_t* const mine = status + (size_t)b * kRadix + d;
    __hip_atomic_store(mine, kOsAggregate | run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t* scratch = s_key;   // (not yet in use)
    uint32_t tot, all;
    const uint32_t ex = block_exclusive_scan(run, &tot, scratch);
    lbase[d] = ex;
    const uint32_t dbase = block_exclusive_scan(totals[d], &all, scratch);   // keys with a smaller digit, whole array
    uint32_t sum = 0;
    for (uint32_t p = b; p-- > 0;) {
      const uint32_t* at = status + (size_t)p * kRadix + d;
      uint32_t st;
      do {
        st = __hip_atomic_load(at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } while ((st >> 30) == 0u);
      sum += st & kOsCount;
      if (st & kOsPrefix) break;
    }
    __hip_atomic_store(mine, kOsPrefix | (sum + run), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    gbase[d] = dbase + sum;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kSortItems; ++it) {
    const size_t i = span + (size_t)it * 64 + lane;
    if (i < n) {
      const uint32_t dg = (k[it] >> shift) & 255u;
      const uint32_t p = lbase[dg] + wave_hist[wid][dg] + rank[it];
      s_key[p] = k[it];
      s_val[p] = v[it];
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kSortItems; ++it) {
    const uint32_t i = (uint32_t)(it * kThreads + threadIdx.x);
    if (i < ntile) {
      const uint32_t key = s_key[i];
      const uint32_t dg = (key >> shift) & 255u;
      const size_t pos = (size_t)gbase[dg] + (i - lbase[dg]);
      keys_out[pos] = key;
      vals_out[pos] = s_val[i];
    }
  }
}

}  // namespace

size_t scan_scratch_words(size_t n) { return (n + kScanTile - 1) / kScanTile + 1; }

hipError_t exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, uint32_t* total,
                              uint32_t* scratch, hipStream_t stream) {
  if (n == 0) {
    if (total) return hipMemsetAsync(total, 0, sizeof(uint32_t), stream);
    return hipSuccess;
  }
  if (n <= kSingleMax) {
    hipLaunchKernelGGL(scan_single, dim3(1), dim3(kSingleThreads), 0, stream, in, out, n, total);
    return hipGetLastError();
  }
  const size_t nb = (n + kScanTile - 1) / kScanTile;
  hipLaunchKernelGGL(scan_tile_sums, dim3((unsigned)nb), dim3(kThreads), 0, stream, in, n, scratch);
  hipLaunchKernelGGL(scan_sums, dim3(1), dim3(kThreads), 0, stream, scratch, nb, total);
  hipLaunchKernelGGL(scan_tile_apply, dim3((unsigned)nb), dim3(kThreads), 0, stream, in, out, n,
                     scratch);
  return hipGetLastError();
}

size_t radix_scratch_words(size_t n) {
  const size_t nb = (n + kSortTile - 1) / kSortTile;
  const size_t radix = (size_t)1 << kMaxRadixBits;
  // (the one-launch passes: at most 4 x (256 totals + 256 x tiles status words) + tickets — less than this)
  return radix * nb + scan_scratch_words(radix * nb) + 2048;
}

// The developer switches of the sort, read once (PLVS_SORT_WIDE_MAX = n up to which 2 passes of <= 11 bits are taken instead of
// 3 of 8; PLVS_SORT_ONESWEEP_MIN / _MAX = the lengths for which a pass is ONE launch, the scan chained inside the scatter).
struct SortSwitches {
  size_t wide_max, onesweep_min, onesweep_max;
};
static const SortSwitches& sort_switches() {
  static const SortSwitches sw{(size_t)env_int("PLVS_SORT_WIDE_MAX", 0, 0, 1 << 30),
                               (size_t)env_int("PLVS_SORT_ONESWEEP_MIN", 1 << 18, 0, 1 << 30),
                               (size_t)env_int("PLVS_SORT_ONESWEEP_MAX", 1 << 23, 0, 1 << 30)};
  return sw;
}
static bool sort_is_wide(size_t n, int total) { return n < sort_switches().wide_max && total > 16 && total <= 2 * kMaxRadixBits; }
// Does a sort of n pairs over `total` key bits take the one-launch passes?  (`bound` = the caller gives the number of pairs on
// the device, radix_sort_pairs_bound: that form exists as one-launch passes only.)  ONE predicate for radix_sort_impl and for
// radix_sort_zero_words, whose callers zero the status words on the side.
static bool sort_takes_onesweep(size_t n, int total, bool bound) {
  const SortSwitches& sw = sort_switches();
  const int os_passes = (total + 7) / 8;
  if (n == 0 || total <= 0 || os_passes > kOsMaxPasses) return false;
  if (bound) return true;
  return sw.onesweep_min != 0 && n >= sw.onesweep_min && n < sw.onesweep_max && !sort_is_wide(n, total);
}

template <typename TV>
static hipError_t radix_sort_impl(uint32_t* keys0, TV* vals0, uint32_t* keys1, TV* vals1,
                            size_t n, int bit_lo, int bit_hi, uint32_t* scratch,
                            hipStream_t stream, bool* result_in_second, const uint32_t* n_dev = nullptr,
                            bool scratch_zeroed = false) {
  *result_in_second = false;
  if (n == 0 || bit_hi <= bit_lo) return hipSuccess;
  const size_t nb = (n + kSortTile - 1) / kSortTile;
  uint32_t* hist = scratch;
  uint32_t* scan_scratch = scratch + ((size_t)1 << kMaxRadixBits) * nb;
  // fewest passes with digits of at most kMaxRadixBits bits, bits spread evenly
  // (a stray high bit of the key inside the last digit is harmless: callers
  // guarantee keys < 2^bit_hi)
  const int total = bit_hi - bit_lo;
  uint32_t *ki = keys0, *ko = keys1;
  TV *vi = vals0, *vo = vals1;
  const bool wide = sort_is_wide(n, total);
  // Measured (MI355X): 2.5 M pairs, three passes: 144 us against 175 us + twelve launch gaps (the chisel colour chain: the step
  // 1.05 -> 1.04 ms); 18 M pairs (voxblox's visits): the look-back costs what the scan kernels did and the totals come on top
  // (1.67 -> 1.70 ms per step) — hence the upper bound.
  const int os_passes = (total + 7) / 8;
  if (n_dev != nullptr && os_passes > kOsMaxPasses) return hipErrorInvalidValue;
  if (sort_takes_onesweep(n, total, n_dev != nullptr)) {
    // (a status word carries a flag in its top two bits and a count below: kOsCount — a tile's inclusive prefix must fit)
    if (n > (size_t)kOsCount) return hipErrorInvalidValue;
    // scratch: [passes][256] digit totals | [passes] tickets | [passes][tiles][256] status words, zeroed together
    uint32_t* totals = scratch;
    uint32_t* tickets = totals + (size_t)os_passes * 256;
    uint32_t* status = tickets + kOsMaxPasses;
    const size_t words = (size_t)os_passes * 256 + kOsMaxPasses + (size_t)os_passes * nb * 256;
    if (!scratch_zeroed) {
      hipError_t e = hipMemsetAsync(scratch, 0, words * sizeof(uint32_t), stream);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(radix_digit_totals, dim3((unsigned)std::min<size_t>(nb, 1024)), dim3(kThreads), 0, stream, ki, n, bit_lo,
                       os_passes, nb, totals, n_dev);
    for (int p = 0; p < os_passes; ++p) {
      hipLaunchKernelGGL((radix_onesweep<TV>), dim3((unsigned)nb), dim3(kThreads), 0, stream, ki, vi, ko, vo, n, bit_lo + 8 * p,
                         totals + (size_t)p * 256, status + (size_t)p * nb * 256, tickets + p, n_dev);
      uint32_t* t = ki; ki = ko; ko = t;
      TV* tv = vi; vi = vo; vo = tv;
      *result_in_second = !*result_in_second;
    }
    return hipGetLastError();
  }
  if (n >= kLdsScatterMin && !wide) {   // long arrays: passes over 8-bit digits, the tiles reordered in LDS
    for (int shift = bit_lo; shift < bit_hi; shift += 8) {
      hipError_t e = radix_pass_lds<TV>(ki, vi, ko, vo, n, shift, nb, hist, scan_scratch, stream);
      if (e != hipSuccess) return e;
      uint32_t* t = ki; ki = ko; ko = t;
      TV* tv = vi; vi = vo; vo = tv;
      *result_in_second = !*result_in_second;
    }
    return hipSuccess;
  }
  const int passes = (total + kMaxRadixBits - 1) / kMaxRadixBits;
  int bits = (total + passes - 1) / passes;
  if (bits < 8) bits = 8;
  for (int p = 0, shift = bit_lo; p < passes; ++p, shift += bits) {
    hipError_t e;
    switch (bits) {
      case 8: e = radix_pass<8, TV>(ki, vi, ko, vo, n, shift, nb, hist, scan_scratch, stream); break;
      case 9: e = radix_pass<9, TV>(ki, vi, ko, vo, n, shift, nb, hist, scan_scratch, stream); break;
      case 10: e = radix_pass<10, TV>(ki, vi, ko, vo, n, shift, nb, hist, scan_scratch, stream); break;
      default: e = radix_pass<11, TV>(ki, vi, ko, vo, n, shift, nb, hist, scan_scratch, stream); break;
    }
    if (e != hipSuccess) return e;
    uint32_t* t = ki; ki = ko; ko = t;
    TV* tv = vi; vi = vo; vo = tv;
    *result_in_second = !*result_in_second;
  }
  return hipSuccess;
}

hipError_t radix_sort_pairs(uint32_t* keys0, uint32_t* vals0, uint32_t* keys1, uint32_t* vals1,
                            size_t n, int bit_lo, int bit_hi, uint32_t* scratch,
                            hipStream_t stream, bool* result_in_second) {
  return radix_sort_impl<uint32_t>(keys0, vals0, keys1, vals1, n, bit_lo, bit_hi, scratch, stream,
                                   result_in_second);
}

size_t radix_sort_zero_words(size_t n, int bit_lo, int bit_hi) {
  const int total = bit_hi - bit_lo, os_passes = (total + 7) / 8;
  if (!sort_takes_onesweep(n, total, false)) return 0;
  const size_t nb = (n + kSortTile - 1) / kSortTile;
  return (size_t)os_passes * 256 + kOsMaxPasses + (size_t)os_passes * nb * 256;
}

hipError_t radix_sort_pairs_zeroed(uint32_t* keys0, uint32_t* vals0, uint32_t* keys1, uint32_t* vals1, size_t n, int bit_lo,
                                   int bit_hi, uint32_t* scratch, hipStream_t stream, bool* result_in_second) {
  return radix_sort_impl<uint32_t>(keys0, vals0, keys1, vals1, n, bit_lo, bit_hi, scratch, stream, result_in_second, nullptr,
                                   radix_sort_zero_words(n, bit_lo, bit_hi) != 0);
}

hipError_t radix_sort_pairs_bound(uint32_t* keys0, uint32_t* vals0, uint32_t* keys1, uint32_t* vals1, size_t n_bound,
                                  const uint32_t* n_dev, int bit_lo, int bit_hi, uint32_t* scratch, hipStream_t stream,
                                  bool* result_in_second) {
  return radix_sort_impl<uint32_t>(keys0, vals0, keys1, vals1, n_bound, bit_lo, bit_hi, scratch, stream, result_in_second, n_dev);
}

hipError_t radix_sort_pairs_u64(uint32_t* keys0, unsigned long long* vals0, uint32_t* keys1,
                                unsigned long long* vals1, size_t n, int bit_lo, int bit_hi,
                                uint32_t* scratch, hipStream_t stream, bool* result_in_second) {
  return radix_sort_impl<unsigned long long>(keys0, vals0, keys1, vals1, n, bit_lo, bit_hi, scratch,
                                             stream, result_in_second);
}

}  // namespace plvs
