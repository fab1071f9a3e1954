// radix_sort.hpp -- LSD radix sort of (u64 or u32 key, u32 value) pairs, 8-bit digits.
#pragma once
#include "common.hpp"
#include "text.hpp"

#include <vector>

namespace nolzss {

constexpr int kRadixBits = 8;
// elements per tile of the radix kernels: 16 per thread, 256 or 512 threads.  8192 doubles the length of the 256
// bin runs a tile writes (32 elements = one full 128-byte line per array on average instead of half a line).
#ifndef NOLZSS_SORT_TILE
#define NOLZSS_SORT_TILE 4096
#endif
constexpr int kSortTile = NOLZSS_SORT_TILE;
static_assert(kSortTile == 4096 || kSortTile == 8192, "16 keys per thread on 256 or 512 threads");

// A sorted-by-bucket view of an array for SEGMENTED passes: 256 buckets (the values of a leading
// digit that an earlier pass partitioned by), each cut into tiles that never straddle a bucket.
// desc holds kSegDescWords words per tile (device memory): first element, element count, bucket,
// index of (bin 0, tile) in the bin-major histogram table, distance between the tile's bins there,
// first / end element of the bucket, nearest non-empty bucket below / above.
constexpr int kSegDescWords = 12;
struct SegView {
    const uint32_t *desc = nullptr;
    uint32_t num_tiles = 0;
};

struct TileExtent {
    size_t first;      // first element
    uint32_t count;    // elements (<= kSortTile)
    uint32_t bucket;   // 0 without segmentation
    size_t hist0;      // index of (bin 0, this tile) in the bin-major histogram table
    uint32_t hstride;  // distance between the bins of this tile in that table
    uint32_t bkt_first, bkt_end, prev_ne, next_ne;  // (segmented only)
    uint32_t aux;      // (segmented only) one word per bucket for the source of the pass
};

// tile -> elements; without a SegView tiles are the consecutive kSortTile-element blocks of [0, n)
__device__ __forceinline__ TileExtent tile_extent(uint32_t tile, size_t n, uint32_t num_tiles, const SegView &seg) {
    TileExtent e;
    if (seg.desc == nullptr) {
        e.first = (size_t)tile * kSortTile;
        e.count = (uint32_t)((n - e.first < (size_t)kSortTile) ? (n - e.first) : (size_t)kSortTile);
        e.bucket = 0;
        e.hist0 = tile;
        e.hstride = num_tiles;
        e.bkt_first = 0;
        e.bkt_end = (uint32_t)n;
        e.prev_ne = e.next_ne = 0;
        e.aux = 0;
        return e;
    }
    const uint4 *d = reinterpret_cast<const uint4 *>(seg.desc + (size_t)tile * kSegDescWords);  // uniform: scalar loads
    const uint4 a = d[0], b = d[1], c = d[2];
    e.first = a.x;
    e.count = a.y;
    e.bucket = a.z;
    e.hist0 = a.w;
    e.hstride = b.x;
    e.bkt_first = b.y;
    e.bkt_end = b.z;
    e.prev_ne = b.w;
    e.next_ne = c.x;
    e.aux = c.y;
    return e;
}

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

// Plain 2-bit DNA (40-bit keys): the same sort with 8-byte records.  One most-significant-digit pass
// computes the keys from the text, partitions the suffixes by their first four bases (the top 8 key
// bits) and keeps only the LOW 32 key bits; four segmented passes then sort every bucket by those.
// The top byte of an element is implied by the bucket it lies in (seg_out, whose tile descriptors live
// in seg_mem: device memory for kSegDescWords * (n / kSortTile + 257) words that must outlive the call).  The sorted low halves end
// in keys32[1], the suffixes in vals[1].  12 + 4 * 24 bytes of traffic per suffix instead of
// 12 + 4 * 32 (hist + scatter, keys + values).
void radix_sort_dna_keys(const PackedText &text, uint32_t *keys32[2], uint32_t *vals[2], uint32_t *seg_mem,
                         SegView &seg_out, Arena &arena, hipStream_t stream, Profiler *prof = nullptr);

// Plain one-segment 2-bit DNA, 16-base key (text.hpp, kP16Syms): the most-significant-digit pass from the text
// and THREE segmented passes over the 24 key bits above the tag byte of the stored word [24 key bits][8-bit tag].
// The sorted words end in keys32[0], the suffixes in vals[0].  8 + 3 * 16 bytes of scatter traffic per suffix.
// Texts of 2^28 .. 1.13 * 10^9 suffixes: ONE segmented pass, then the 65 536 sub-buckets sorted in LDS (local_sort.hpp:
// local_sort_kernel) -- 8 + 16 + 16 bytes per suffix; from 3 * 2^28 suffixes up that kernel can do the regroup of round 0
// on the way (Round0Regroup below: the keys are then not written at all).
// (also segmented texts with a terminator table of at most kTermFew entries and segments of at least 16 symbols -- a
// prepared reverse-complement string --: key16_applicable says whether a text takes this sort)
bool key16_applicable(const PackedText &text);
// What the regroup kernel of round 0 produces from the sorted keys (suffix_array.hip: regroup_kernel<true, 3>), asked of the
// sort itself: where the sub-buckets are finished in LDS the sorted keys are at hand -- LCP of every boundary the keys
// decide (0xffffffff = pending elsewhere), the elements that stay tied (slot and slot of their group's head, in slot
// order) and their number.  `done` says whether the sort did it (the keys are then NOT written).
struct Round0Regroup {
    uint32_t *lcp = nullptr;
    uint32_t *new_slot = nullptr, *new_grp = nullptr;
    uint32_t *d_total = nullptr;  // device: [0] survivors, [1] look-back error flag
    bool done = false;
};
void radix_sort_dna_keys16(const PackedText &text, uint32_t *keys32[2], uint32_t *vals[2], uint32_t *seg_mem,
                           SegView &seg_out, Arena &arena, hipStream_t stream, Profiler *prof = nullptr,
                           Round0Regroup *regroup = nullptr);

// The same sort on FUSED records [stored key word : 32 | suffix : 32] (round 4, A/B: NOLZSS_FUSED_SORT): rec[0] and rec[1]
// hold n 64-bit words each; the last pass writes the suffixes to sa_out and the key words to rec[0] (as 32-bit words).
void radix_sort_dna_keys16_fused(const PackedText &text, uint64_t *rec[2], uint32_t *sa_out, uint32_t *seg_mem,
                                 SegView &seg_out, Arena &arena, hipStream_t stream, Profiler *prof = nullptr);

// Independent records of 2-bit DNA (text.terms.seq_shift != 0), at most n / 2^16 of them: the records are
// the buckets -- the text is "partitioned" as it lies -- and every bucket is sorted on 8-byte records by the
// 32-bit key [kRecSyms bases][4-bit length tag], least significant digit first, the first pass making its
// pairs from the packed text: FOUR passes (8 + 3 * 16 bytes per suffix) where the general sort of the merged
// batch takes five on 12-byte records.  h_terms = host copy of the terminator table.  seg_mem: kSegDescWords *
// (n / kSortTile + records + 1) words.  The sorted keys end in keys32[0], the suffixes in vals[0].
void radix_sort_record_keys(const PackedText &text, const std::vector<uint32_t> &h_terms, uint32_t *keys32[2],
                            uint32_t *vals[2], uint32_t *seg_mem, SegView &seg_out, Arena &arena, hipStream_t stream,
                            Profiler *prof = nullptr);

// (u32 key, u32 value) pairs that lie in SEGMENTS [h_start[k], h_start[k + 1]) (h_start[0] = 0, the last entry = n), each
// segment sorted for itself by the low 8 * npasses key bits: the segments are the buckets of bucket-segmented passes on
// 8-byte records -- what the doubling rounds need for the members of their LARGE groups, whose group is known from where
// they lie (round 4: they used to sort 12-byte records by (group, key), eight passes instead of four).  Returns the
// index of the buffer pair that holds the result.
int radix_sort_segments_u32(uint32_t *keys[2], uint32_t *vals[2], size_t n, const std::vector<uint32_t> &h_start, int npasses,
                            Arena &arena, hipStream_t stream, Profiler *prof = nullptr);

// out[idx[k]] = val[k] for k < count, idx[k] < n_out (entries with idx >= n_out are dropped).
// A random 4-byte scatter over an array much larger than the caches costs a read-modify-write
// of a whole line per element at HBM.  For large targets the pairs are therefore first
// partitioned by the top 8 bits of idx (one radix pass, coalesced), then written bucket by
// bucket so that all writes in flight fall into a window of n_out/256 entries that L2 /
// Infinity Cache can merge into full lines (a permutation of the whole target is assembled
// window by window in LDS instead).  idx[0]/val[0] hold the input, idx[1]/val[1] are scratch of
// the same size.  With keep_input the input arrays survive (a third pair of buffers is taken
// from the arena); otherwise they are used as scratch too.  The pointer arrays may be updated.
// keep_val = false (with keep_input): only idx[0] survives, val[0] is used as a ping-pong buffer too.
//
// plan (optional): the target is a text of independent RECORDS and the pairs are (position, value) in suffix-
// array order of such a text -- a block-diagonal permutation: the ranks of a record hold the positions of that
// record (record_scatter_plan).  Then ONE segmented radix pass (the record is the bucket, the digit the window
// inside the record) and the window scatter do it, 28 instead of 48 bytes per pair; idx[0] / val[0] survive,
// idx[1] / val[1] are the only scratch.
struct RecordScatterPlan {
    SegView seg;                    // the base positions of every record as one bucket (ranks = positions)
    const uint32_t *win = nullptr;  // per window: first list element, first target element, elements
    uint32_t num_windows = 0;
    const uint32_t *sep = nullptr;  // per separator: its rank (the first of its record) and its position
    uint32_t num_seps = 0;
    uint32_t n = 0;
    int window_bits = 0;
};
// out2 (optional): a second target, out2[idx[k]] = k + 1 -- for the pairs (sa[r], code[r]) of the pipeline that is
// the inverse suffix array in its 1-based form, delivered by the same permutation.  val[1] must then hold
// 2 * count words (the pairs travel with 64-bit values).
void bucketed_scatter(uint32_t *idx[2], uint32_t *val[2], size_t count, uint32_t *out, uint32_t n_out,
                      Arena &arena, hipStream_t stream, Profiler *prof, bool keep_input, bool keep_val = true,
                      const RecordScatterPlan *plan = nullptr, uint32_t *out2 = nullptr);
// The same permutation for pairs that already carry both values in one 64-bit word (low half -> out, high half ->
// out2) and are a permutation of [0, count): out[idx[k]] = (uint32_t)packed[k], out2[idx[k]] = packed[k] >> 32.
// Both inputs are overwritten (they serve as buffers of the later passes).
void permute_packed(uint32_t *idx, uint64_t *packed, size_t count, uint32_t *out, uint32_t *out2, Arena &arena,
                    hipStream_t stream, Profiler *prof);
// The plan for a text of n symbols whose records end at h_terms[k] (separator positions, the last entry = n);
// false when the shape does not allow it (a record longer than 2^22 bases, or too many short ones).  The
// tables live in the arena (not released here).
bool record_scatter_plan(const std::vector<uint32_t> &h_terms, uint32_t n, Arena &arena, hipStream_t stream,
                         RecordScatterPlan &plan);

}  // namespace nolzss
