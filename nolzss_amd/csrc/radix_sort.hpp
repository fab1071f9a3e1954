// radix_sort.hpp -- LSD radix sort of (u64 or u32 key, u32 value) pairs, 8-bit digits.
#pragma once
#include "common.hpp"
#include "text.hpp"

namespace nolzss {

constexpr int kRadixBits = 8;

// Sorts n pairs by the key digits at the given bit offsets (least significant first;
// each digit is kRadixBits wide).  keys[0]/vals[0] hold the input; the two buffers
// ping-pong.  Returns the index (0 or 1) of the buffer pair that holds the sorted output.
// Stable.  Temporaries come from the arena and are released on return.
// prof (optional) receives per-kernel HIP-event timings: "rs_hist", "rs_scan", "rs_scatter".
int radix_sort_pairs(uint64_t *keys[2], uint32_t *vals[2], size_t n, const int *shifts,
                     int npasses, Arena &arena, hipStream_t stream, Profiler *prof = nullptr);
int radix_sort_pairs(uint32_t *keys[2], uint32_t *vals[2], size_t n, const int *shifts,
                     int npasses, Arena &arena, hipStream_t stream, Profiler *prof = nullptr);

// The round-0 sort of the suffix array: pairs (initial_key(i), i) for i < text.n, sorted by the given
// digits.  The first pass computes the keys from the packed text on the fly (histogram and scatter
// kernels read 2 bits per base instead of 12 bytes per suffix, and no kernel writes the unsorted
// pairs); keys[0] / vals[0] are only used as ping-pong space from the second pass on.
int radix_sort_initial_keys(const PackedText &text, uint64_t *keys[2], uint32_t *vals[2], const int *shifts,
                            int npasses, Arena &arena, hipStream_t stream, Profiler *prof = nullptr);

// out[idx[k]] = val[k] for k < count, idx[k] < n_out (entries with idx >= n_out are dropped).
// A random 4-byte scatter over an array much larger than the caches costs a read-modify-write
// of a whole line per element at HBM.  For large targets the pairs are therefore first
// partitioned by the top 8 bits of idx (one radix pass, coalesced), then written bucket by
// bucket so that all writes in flight fall into a window of n_out/256 entries that L2 /
// Infinity Cache can merge into full lines (a permutation of the whole target is assembled
// window by window in LDS instead).  idx[0]/val[0] hold the input, idx[1]/val[1] are scratch of
// the same size.  With keep_input the input arrays survive (a third pair of buffers is taken
// from the arena); otherwise they are used as scratch too.  The pointer arrays may be updated.
void bucketed_scatter(uint32_t *idx[2], uint32_t *val[2], size_t count, uint32_t *out, uint32_t n_out,
                      Arena &arena, hipStream_t stream, Profiler *prof, bool keep_input);

}  // namespace nolzss
