// Exact k=2 nearest neighbours in Hamming space over 256-bit descriptors.
//
// Replaces BinaryDescriptorMatcher::knnMatch (LBD lines,
// Thirdparty/line_descriptor/src/binary_descriptor_matcher_custom.cpp:258-336)
// and cv::BFMatcher(NORM_HAMMING).knnMatch(k=2) (src/Frame.cc:2977); the
// distance is ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:2198-2225).
//
// Layout: one 64-lane wavefront per query.  The query's eight dwords sit in
// SGPR-uniform registers; lane l walks train rows l, l+64, ... with two
// 16-byte loads per row (the whole train set is L2-resident: 2000 x 32 B =
// 64 KB), keeps its own (best, second) as 64-bit composite keys, and the wave
// merges the 64 sorted pairs with a 6-step xor-shuffle butterfly.
//
// Composite key (smaller is better):
//   bits 48..56  Hamming distance
//   bits 44..47  s  = min over the 32 bytes of popcount(q_byte ^ t_byte)   (MIH rule only)
//   bits 39..43  k  = first byte index reaching s                          (MIH rule only)
//   bits 31..38  the xor pattern of byte k                                  (MIH rule only)
//   bits  0..30  train index
// With the three middle fields zero this is "lowest train index wins"
// (cv::BFMatcher).  With them filled it reproduces the discovery order of the
// reference's multi-index hash: Mihasher(256,32) splits the code into 32 one-byte
// substrings, grows the per-substring radius s = 0..8, scans substrings
// k = 0..31, enumerates the s-bit flip patterns in increasing numeric order and
// reads each bucket in insertion (= train index) order
// (binary_descriptor_matcher_custom.cpp:633-752, 916-941); an item is first
// seen at the lexicographically smallest (s, k) with popcount(q_k ^ t_k) == s.
#include "common.hpp"

namespace {

constexpr int kWavesPerBlock = 4;
constexpr unsigned long long kNoKey = ~0ull;

__device__ __forceinline__ void insert_key(unsigned long long key, unsigned long long& b1,
                                           unsigned long long& b2) {
  if (key < b1) {
    b2 = b1;
    b1 = key;
  } else if (key < b2) {
    b2 = key;
  }
}

template <bool kMih>
__global__ __launch_bounds__(64 * kWavesPerBlock) void hamming_knn2_kernel(
    const uint4* __restrict__ query, int nq, const uint4* __restrict__ train, int nt,
    const uint8_t* __restrict__ qmask, int32_t* __restrict__ idx, int32_t* __restrict__ dist) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (q >= nq) return;
  if (qmask != nullptr && qmask[q] == 0) {
    if (lane == 0) {
      idx[2 * q] = idx[2 * q + 1] = -1;
      dist[2 * q] = dist[2 * q + 1] = -1;
    }
    return;
  }
  // Wave-uniform query words.
  const uint4 qa = query[2 * q], qb = query[2 * q + 1];
  const uint32_t qw[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};

  unsigned long long b1 = kNoKey, b2 = kNoKey;
  for (int t = lane; t < nt; t += 64) {
    const uint4 ta = train[2 * t], tb = train[2 * t + 1];
    const uint32_t x[8] = {ta.x ^ qw[0], ta.y ^ qw[1], ta.z ^ qw[2], ta.w ^ qw[3],
                           tb.x ^ qw[4], tb.y ^ qw[5], tb.z ^ qw[6], tb.w ^ qw[7]};
    unsigned long long key;
    if constexpr (kMih) {
      uint32_t d = 0, s = 9, k = 0, pat = 0;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        // per-byte popcounts of x[w]
        uint32_t c = x[w] - ((x[w] >> 1) & 0x55555555u);
        c = (c & 0x33333333u) + ((c >> 2) & 0x33333333u);
        c = (c + (c >> 4)) & 0x0f0f0f0fu;
        d += (c * 0x01010101u) >> 24;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t cb = (c >> (8 * j)) & 0xffu;
          if (cb < s) {  // strict: the first byte reaching the minimum wins
            s = cb;
            k = (uint32_t)(w * 4 + j);
            pat = (x[w] >> (8 * j)) & 0xffu;
          }
        }
      }
      key = ((unsigned long long)d << 48) | ((unsigned long long)s << 44) |
            ((unsigned long long)k << 39) | ((unsigned long long)pat << 31) |
            (unsigned long long)(uint32_t)t;
    } else {
      uint32_t d = 0;
#pragma unroll
      for (int w = 0; w < 8; ++w) d += (uint32_t)__popc(x[w]);
      key = ((unsigned long long)d << 48) | (unsigned long long)(uint32_t)t;
    }
    insert_key(key, b1, b2);
  }
  // Butterfly merge of the 64 sorted pairs.
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const unsigned long long p1 = __shfl_xor(b1, m, 64);
    const unsigned long long p2 = __shfl_xor(b2, m, 64);
    const unsigned long long lo = b1 < p1 ? b1 : p1;
    const unsigned long long hi = b1 < p1 ? p1 : b1;
    const unsigned long long s2 = b2 < p2 ? b2 : p2;
    b1 = lo;
    b2 = hi < s2 ? hi : s2;
  }
  if (lane == 0) {
    idx[2 * q] = b1 == kNoKey ? -1 : (int32_t)(b1 & 0x7fffffffull);
    dist[2 * q] = b1 == kNoKey ? -1 : (int32_t)(b1 >> 48);
    idx[2 * q + 1] = b2 == kNoKey ? -1 : (int32_t)(b2 & 0x7fffffffull);
    dist[2 * q + 1] = b2 == kNoKey ? -1 : (int32_t)(b2 >> 48);
  }
}

}  // namespace

extern "C" {

int plvs_hip_hamming_knn2_dev(const uint8_t* d_query, int nq, const uint8_t* d_train, int nt,
                              const uint8_t* d_qmask, int tie_rule, int32_t* d_idx,
                              int32_t* d_dist, void* stream) {
  if (nq <= 0 || nt <= 0) {
    plvs::set_error("hamming_knn2: descriptors matrices cannot be void (nq=%d nt=%d)", nq, nt);
    return PLVS_ERR_EMPTY;
  }
  PLVS_REQUIRE(d_query && d_train && d_idx && d_dist, "null descriptor or output pointer");
  PLVS_REQUIRE(tie_rule == PLVS_TIE_LOWEST_INDEX || tie_rule == PLVS_TIE_MIH, "unknown tie_rule");
  PLVS_REQUIRE((reinterpret_cast<uintptr_t>(d_query) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(d_train) & 15) == 0,
               "descriptor pointers must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(plvs::ceil_div((size_t)nq, kWavesPerBlock)), block(64 * kWavesPerBlock);
  if (tie_rule == PLVS_TIE_MIH) {
    hipLaunchKernelGGL(hamming_knn2_kernel<true>, grid, block, 0, s,
                       reinterpret_cast<const uint4*>(d_query), nq,
                       reinterpret_cast<const uint4*>(d_train), nt, d_qmask, d_idx, d_dist);
  } else {
    hipLaunchKernelGGL(hamming_knn2_kernel<false>, grid, block, 0, s,
                       reinterpret_cast<const uint4*>(d_query), nq,
                       reinterpret_cast<const uint4*>(d_train), nt, d_qmask, d_idx, d_dist);
  }
  PLVS_KERNEL_CHECK();
  return PLVS_OK;
}

int plvs_hip_hamming_knn2(const uint8_t* query, int nq, const uint8_t* train, int nt,
                          const uint8_t* qmask, int tie_rule, int32_t* idx, int32_t* dist) {
  if (nq <= 0 || nt <= 0) {
    plvs::set_error("hamming_knn2: descriptors matrices cannot be void (nq=%d nt=%d)", nq, nt);
    return PLVS_ERR_EMPTY;
  }
  PLVS_REQUIRE(query && train && idx && dist, "null descriptor or output pointer");
  // one staged block: [query | train | mask] in, [idx | dist] out (16-byte aligned pieces)
  auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
  const size_t o_q = 0, o_t = o_q + up16((size_t)nq * 32), o_m = o_t + up16((size_t)nt * 32),
               o_i = o_m + up16(qmask ? (size_t)nq : 0), o_d = o_i + up16((size_t)nq * 2 * sizeof(int32_t)),
               total = o_d + up16((size_t)nq * 2 * sizeof(int32_t));
  plvs::HostStage& st = plvs::thread_stage();
  PLVS_HIP_TRY(st.reserve(total));
  memcpy(st.pinned + o_q, query, (size_t)nq * 32);
  memcpy(st.pinned + o_t, train, (size_t)nt * 32);
  if (qmask) memcpy(st.pinned + o_m, qmask, (size_t)nq);
  PLVS_HIP_TRY(hipMemcpyAsync(st.dev, st.pinned, o_i, hipMemcpyHostToDevice, st.stream));
  const int rc = plvs_hip_hamming_knn2_dev(reinterpret_cast<uint8_t*>(st.dev + o_q), nq, reinterpret_cast<uint8_t*>(st.dev + o_t),
                                           nt, qmask ? reinterpret_cast<uint8_t*>(st.dev + o_m) : nullptr, tie_rule,
                                           reinterpret_cast<int32_t*>(st.dev + o_i), reinterpret_cast<int32_t*>(st.dev + o_d),
                                           st.stream);
  if (rc != PLVS_OK) return rc;
  PLVS_HIP_TRY(hipMemcpyAsync(st.pinned + o_i, st.dev + o_i, total - o_i, hipMemcpyDeviceToHost, st.stream));
  PLVS_HIP_TRY(hipStreamSynchronize(st.stream));
  memcpy(idx, st.pinned + o_i, (size_t)nq * 2 * sizeof(int32_t));
  memcpy(dist, st.pinned + o_d, (size_t)nq * 2 * sizeof(int32_t));
  return PLVS_OK;
}

}  // extern "C"
