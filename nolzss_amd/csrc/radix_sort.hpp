// radix_sort.hpp -- LSD radix sort of (u64 key, u32 value) pairs, 8-bit digits.
#pragma once
#include "common.hpp"

namespace nolzss {

constexpr int kRadixBits = 8;

// Sorts n pairs by the key digits at the given bit offsets (least significant first;
// each digit is kRadixBits wide).  keys[0]/vals[0] hold the input; the two buffers
// ping-pong.  Returns the index (0 or 1) of the buffer pair that holds the sorted output.
// Stable.  Temporaries come from the arena and are released on return.
// prof (optional) receives per-kernel HIP-event timings: "rs_hist", "rs_scan", "rs_scatter".
int radix_sort_pairs(uint64_t *keys[2], uint32_t *vals[2], size_t n, const int *shifts,
                     int npasses, Arena &arena, hipStream_t stream, Profiler *prof = nullptr);

}  // namespace nolzss
