// Device-wide primitives used by the hot-path kernels: exclusive scan and a
// stable LSD radix sort of (key, value) pairs.  Internal C++ API (not part of
// the C ABI).  All calls are asynchronous on `stream`.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace plvs {

// Scratch requirement (in uint32 words) of exclusive_scan_u32 for n items.
size_t scan_scratch_words(size_t n);
// out[i] = sum(in[0..i)); *total (device word, may be null) = sum(in[0..n)).
// in == out is allowed.
hipError_t exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, uint32_t* total,
                              uint32_t* scratch, hipStream_t stream);

// Scratch requirement (uint32 words) of radix_sort_pairs for n items.
size_t radix_scratch_words(size_t n);
// Stable sort of n (key,val) pairs by key bits [bit_lo, bit_hi).  Ping-pongs
// between (keys0,vals0) and (keys1,vals1); *result_in_second tells where the
// sorted data ended up.  n is a host value.
hipError_t radix_sort_pairs(uint32_t* keys0, uint32_t* vals0, uint32_t* keys1, uint32_t* vals1,
                            size_t n, int bit_lo, int bit_hi, uint32_t* scratch,
                            hipStream_t stream, bool* result_in_second);

// A sort of 2^18..2^23 pairs starts by zeroing part of its scratch (a launch of its own, 25 us in front of a chain of dependent
// kernels).  A caller whose previous kernel can do that on the side asks how many words (0: this sort zeroes nothing), has them
// zeroed, and calls the _zeroed form.
size_t radix_sort_zero_words(size_t n, int bit_lo, int bit_hi);
hipError_t radix_sort_pairs_zeroed(uint32_t* keys0, uint32_t* vals0, uint32_t* keys1, uint32_t* vals1, size_t n, int bit_lo,
                                   int bit_hi, uint32_t* scratch, hipStream_t stream, bool* result_in_second);

// The same launched on a BOUND: the arrays hold min(n_bound, *n_dev) pairs, n_dev a device word written before the sort runs
// (the host does not know it yet); every pass is one launch sized by the bound whose surplus tiles leave at once.
// Scratch: radix_scratch_words(n_bound).  At most 32 key bits.
hipError_t radix_sort_pairs_bound(uint32_t* keys0, uint32_t* vals0, uint32_t* keys1, uint32_t* vals1, size_t n_bound,
                                  const uint32_t* n_dev, int bit_lo, int bit_hi, uint32_t* scratch, hipStream_t stream,
                                  bool* result_in_second);

// Same with 64-bit values.
hipError_t radix_sort_pairs_u64(uint32_t* keys0, unsigned long long* vals0, uint32_t* keys1,
                                unsigned long long* vals1, size_t n, int bit_lo, int bit_hi,
                                uint32_t* scratch, hipStream_t stream, bool* result_in_second);

}  // namespace plvs
