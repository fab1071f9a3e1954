// radix_sort.hip -- stable LSD radix sort of (key, u32 value) pairs for gfx950; keys u64 or u32.
//
// One pass = three launches:
//   rs_hist_kernel     per 4096-key tile, a 256-bin digit histogram (LDS atomics) written
//                      bin-major, so one linear exclusive scan yields every (bin, tile) base;
//   scan               exclusive add-scan of the 256 x num_tiles table (scan.hip);
//   rs_scatter_kernel  loads the whole tile (all loads in flight before anything else), ranks
//                      every key inside its wavefront -- 8 ballots give the lanes with the same
//                      digit, the first lane of each digit group does one returning LDS atomic on
//                      the wave's counter, the atomics of all 16 rows are issued back to back --
//                      sorts the tile by digit through LDS, and writes each bin's run with
//                      consecutive lanes on consecutive addresses.
// The pairs of a pass come from arrays or (first pass of the suffix sort) are computed from the
// packed text; passes can be SEGMENTED: tiles that never straddle one of 256 buckets made by an
// earlier most-significant-digit pass (radix_sort.hpp, SegView), which is how plain DNA is sorted
// on 8-byte records.
// HBM-bound: algorithmic traffic of the scatter kernel = 2 * (sizeof(key) + 4) bytes per pair;
// the histogram kernel reads sizeof(key) bytes per pair.  4.0-4.8 TB/s on MI355X (u32 keys).
//
// Occupancy is what the scatter kernel lives on: keys and values take turns in ONE LDS staging
// buffer (37 KiB per workgroup).  Tiles are dealt to XCDs in contiguous ranges (blockIdx % 8
// shares an XCD) so that the bin runs of neighbouring tiles, adjacent in the output, meet in one
// L2 and their partial cache lines merge there: without it the scatter runs at HALF the speed.
#include "radix_sort.hpp"

#include "lookback.hpp"
#include "scan.hpp"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace nolzss {
namespace {

constexpr int kKeysPerThread = 16;  // 12 and 8 measured within 3 % of this on MI355X
constexpr int kTile = kSortTile;    // part of the SegView contract
constexpr int kThreads = kTile / kKeysPerThread;  // 256 (4096-pair tiles) or 512 (8192)
constexpr int kWaves = kThreads / 64;
constexpr int kBins = 1 << kRadixBits;
constexpr int kWaveSpan = 64 * kKeysPerThread;  // 1024 keys per wavefront, 16 rows of 64
// blocks per CU the scatter kernel is compiled for: 3 x 256 or 2 x 512 threads (128 VGPRs at most for the latter)
#ifndef NOLZSS_SCATTER_BLOCKS
#define NOLZSS_SCATTER_BLOCKS 1
#endif
constexpr int kScatterWavesPerSimd = kThreads == 256 ? NOLZSS_SCATTER_BLOCKS : 4;  // (1 = no register cap: the 256-thread form as it always was)

static_assert(kThreads % kBins == 0, "the first kBins threads own one bin each in the offset phase");

// 8-bit digit of a key at a bit offset that is a multiple of 8: the digit never straddles the two
// halves of a 64-bit key, so one v_bfe_u32 on the right half does it (a variable 64-bit shift costs
// several instructions, three times per key)
__device__ __forceinline__ uint32_t digit_of(uint64_t k, int shift) {
    const uint32_t half = shift >= 32 ? (uint32_t)(k >> 32) : (uint32_t)k;
    return (half >> (shift & 31)) & (uint32_t)(kBins - 1);
}
__device__ __forceinline__ uint32_t digit_of(uint32_t k, int shift) { return (k >> shift) & (uint32_t)(kBins - 1); }

// block -> tile.  Blocks b, b + 8, b + 16, .. share an XCD (round-robin dispatch); an XCD takes CHUNKS of
// kXcdChunk consecutive tiles, chunk c going to XCD c % 8: neighbouring tiles of a chunk meet in one L2 (their
// bin runs are adjacent in the output and merge there into full lines), and the eight write fronts of a bin --
// one per XCD -- stay within a few chunks of each other instead of an eighth of the array apart
// (8 / 32 / 64 / 256 / 1024 tiles per chunk: 36.1 / 33.9 / 34.1 / 34.3 / 34.1 ms for the eight large u32 passes,
// 35.2 with one contiguous range per XCD).
#ifndef NOLZSS_XCD_CHUNK
#define NOLZSS_XCD_CHUNK 64
#endif
constexpr uint32_t kXcdChunk = NOLZSS_XCD_CHUNK;
__device__ __forceinline__ uint32_t xcd_tile(uint32_t b, uint32_t num_tiles) {
    const uint32_t x = b % 8, k = b / 8;           // k-th block of XCD x
    const uint32_t chunk = (k / kXcdChunk) * 8 + x;  // chunks of this XCD: x, x + 8, x + 16, ..
    const uint32_t tile = chunk * kXcdChunk + k % kXcdChunk;
    return tile < num_tiles ? tile : 0xffffffffu;
}
// blocks to launch so that every tile is covered by the mapping above
inline uint32_t xcd_grid(uint32_t num_tiles) {
    const uint32_t chunks = (uint32_t)div_up(num_tiles, kXcdChunk);
    return (uint32_t)div_up(chunks, 8) * 8 * kXcdChunk;
}

// where a pass reads its pairs from: arrays, or (first pass of the suffix sort) the packed text.
// Every source splits a key into load() -- nothing but the loads -- and key_of() / hist_digit_of() -- the
// arithmetic: the kernels issue the loads of a whole tile first.  (With the arithmetic inside the load loop
// the compiler waited for every load on its own, `s_waitcnt vmcnt(0)` sixteen times per thread: the passes
// that compute their keys ran 1.4 x slower than the passes that only read them.)
template <typename KeyT> struct ArraySrc {
    using Raw = KeyT;
    const KeyT *__restrict__ keys;
    const uint32_t *__restrict__ vals;
    __device__ __forceinline__ Raw load(size_t idx, const TileExtent &) const { return keys[idx]; }
    __device__ __forceinline__ KeyT key_of(Raw raw, size_t, const TileExtent &) const { return raw; }
    __device__ __forceinline__ uint32_t hist_digit_of(Raw raw, size_t, int shift, const TileExtent &) const { return digit_of(raw, shift); }
    __device__ __forceinline__ uint32_t val(size_t idx) const { return vals[idx]; }
    __device__ __forceinline__ bool digits_from_window(int) const { return false; }
    __device__ __forceinline__ uint64_t window(size_t) const { return 0; }
};
// (target position, value) pairs in list order that carry a SECOND value: the list position itself + 1 (the rank
// of the suffix in the pipeline's 1-based convention).  The two values travel as ONE 64-bit value (first value in
// the low half): two output streams per pass, the value stream in runs of a full 128-byte line, where three
// streams of 4-byte values ran at 2.8 TB/s.  The permutation that brings the factor-length codes into text order
// delivers the inverse suffix array on the way (bucketed_scatter with out2).
struct RankSrc {
    using Raw = uint32_t;
    const uint32_t *__restrict__ keys;
    const uint32_t *__restrict__ vals;
    __device__ __forceinline__ Raw load(size_t idx, const TileExtent &) const { return keys[idx]; }
    __device__ __forceinline__ uint32_t key_of(Raw raw, size_t, const TileExtent &) const { return raw; }
    __device__ __forceinline__ uint32_t hist_digit_of(Raw raw, size_t, int shift, const TileExtent &) const { return digit_of(raw, shift); }
    __device__ __forceinline__ uint64_t val(size_t idx) const { return (uint64_t)vals[idx] | ((uint64_t)((uint32_t)idx + 1u) << 32); }
    __device__ __forceinline__ bool digits_from_window(int) const { return false; }
    __device__ __forceinline__ uint64_t window(size_t) const { return 0; }
};
struct PairSrc {
    using Raw = uint32_t;
    const uint32_t *__restrict__ keys;
    const uint64_t *__restrict__ vals;
    __device__ __forceinline__ Raw load(size_t idx, const TileExtent &) const { return keys[idx]; }
    __device__ __forceinline__ uint32_t key_of(Raw raw, size_t, const TileExtent &) const { return raw; }
    __device__ __forceinline__ uint32_t hist_digit_of(Raw raw, size_t, int shift, const TileExtent &) const { return digit_of(raw, shift); }
    __device__ __forceinline__ uint64_t val(size_t idx) const { return vals[idx]; }
    __device__ __forceinline__ bool digits_from_window(int) const { return false; }
    __device__ __forceinline__ uint64_t window(size_t) const { return 0; }
};
template <int BITS> struct TextSrc {
    using Raw = SymWords;
    const uint64_t *__restrict__ words;
    TermTable terms;
    bool segmented;
    bool digit_from_text = true;  // MSD histogram digit straight from the packed text (no terminators in the text)
    __device__ __forceinline__ Raw load(size_t idx, const TileExtent &) const { return sym_words<BITS>(words, idx); }
    __device__ __forceinline__ uint64_t key_of(const Raw &raw, size_t idx, const TileExtent &) const {
        return initial_key_of<BITS>(sym_word_of<BITS>(raw, idx), terms, segmented, (uint32_t)idx);
    }
    __device__ __forceinline__ uint32_t val(size_t idx) const { return (uint32_t)idx; }
    // histogram passes only need the digit; the top 8 bits of a plain key are the first 8 / BITS
    // symbols, straight from the packed text (no length tag, no terminator search)
    __device__ __forceinline__ uint32_t hist_digit_of(const Raw &raw, size_t idx, int shift, const TileExtent &ext) const {
        constexpr int kKeyBits = KeyLayout<BITS>::kSyms * BITS + KeyLayout<BITS>::kTagBits;
        if (!segmented && digit_from_text && shift == kKeyBits - kRadixBits)
            return digit_of(sym_word_of<BITS>(raw, idx) >> (64 - kKeyBits), shift);
        return digit_of(key_of(raw, idx, ext), shift);
    }
    // the most significant digits of 16 CONSECUTIVE suffixes are 8-bit windows of one 64-bit piece of a
    // 2-bit text: a histogram thread can take 16 neighbours with two loads instead of 16 strided
    // elements with 32
    __device__ __forceinline__ bool digits_from_window(int shift) const {
        constexpr int kKeyBits = KeyLayout<BITS>::kSyms * BITS + KeyLayout<BITS>::kTagBits;
        return BITS == 2 && !segmented && digit_from_text && shift == kKeyBits - kRadixBits;
    }
    __device__ __forceinline__ uint64_t window(size_t idx) const { return sym_word<BITS>(words, idx); }
    // digit of the j-th of the 16 suffixes that start in the window fetched for element idx0
    __device__ __forceinline__ uint32_t window_digit(uint64_t w, int j, size_t) const {
        return (uint32_t)((w >> (64 - kRadixBits - 2 * j)) & (uint64_t)(kBins - 1));
    }
};
// Plain one-segment 2-bit DNA with the 16-base key of text.hpp (kP16Syms): the most-significant-digit pass.  The
// 40-bit value it hands to the kernels is [32 key bits][8-bit tag]: the digit of this pass (shift 32) is the first
// four bases, and the low 32 bits -- what the pass stores -- are [24 key bits][tag].  Element e of the pass is
// suffix n - 1 - e for e < 16 and suffix e - 16 behind them: the suffixes that end inside the key window come
// first, shortest first, and the stable passes leave them in front of the longer suffixes that tie with their
// zero-padded keys (the packed text is zero behind its end: no masking).
struct Text16Src {
    using Raw = SymWords;
    const uint64_t *__restrict__ words;
    uint32_t n;  // >= 32
    __device__ __forceinline__ uint32_t suffix_of(size_t idx) const {
        return idx < 16 ? n - 1u - (uint32_t)idx : (uint32_t)idx - 16u;
    }
    __device__ __forceinline__ Raw load(size_t idx, const TileExtent &) const { return sym_words<2>(words, suffix_of(idx)); }
    __device__ __forceinline__ uint64_t key_of(const Raw &raw, size_t idx, const TileExtent &) const {
        const uint32_t s = suffix_of(idx);
        const uint32_t lim = n - s;
        const uint32_t tag = lim < (uint32_t)kP16Syms ? lim : (uint32_t)kP16Syms;
        return ((sym_word_of<2>(raw, s) >> 32) << kP16TagBits) | tag;
    }
    __device__ __forceinline__ uint32_t val(size_t idx) const { return suffix_of(idx); }
    __device__ __forceinline__ uint32_t hist_digit_of(const Raw &raw, size_t idx, int, const TileExtent &) const {
        return (uint32_t)(sym_word_of<2>(raw, suffix_of(idx)) >> (64 - kRadixBits));
    }
    // (tiles start at multiples of 16 elements: only the very first window of the list holds the rotated suffixes;
    // it is the window of suffix n - 16, read backwards)
    __device__ __forceinline__ bool digits_from_window(int) const { return true; }
    __device__ __forceinline__ uint64_t window(size_t idx0) const { return sym_word<2>(words, idx0 == 0 ? (uint64_t)n - 16u : idx0 - 16u); }
    __device__ __forceinline__ uint32_t window_digit(uint64_t w, int j, size_t idx0) const {
        const int jj = idx0 == 0 ? 15 - j : j;
        return (uint32_t)((w >> (64 - kRadixBits - 2 * jj)) & (uint64_t)(kBins - 1));
    }
};
// The same pass for a SEGMENTED text with a short terminator table (at most kTermFew entries: a prepared reverse-
// complement string T $ rc(T) $ has three).  Every terminator in front of the end has 16 suffixes that end inside the key
// window (0 .. 15 symbols: the terminator's own suffix is the one of 0 symbols), the end of the text 15 (1 .. 15) unless the
// text ends with a terminator.  They come first, ordered by (symbols, terminator) -- the order text.hpp gives suffixes that
// agree up to the nearer terminator -- and the others follow in text order, the removed stretches skipped.  Symbols behind
// the terminator of a suffix belong to the next segment and are masked out of its key.  (Segments of at least 16 symbols:
// key16_applicable.)
struct Text16SegSrc {
    using Raw = SymWords;
    const uint64_t *__restrict__ words;
    uint32_t n;
    uint32_t treal;       // terminators in front of the end of the text
    uint32_t end_shorts;  // 15 or 0
    uint32_t nshort;      // 16 * treal + end_shorts
    uint32_t pos[kTermFew];  // pos[treal] = n
    __device__ __forceinline__ uint32_t suffix_of(size_t idx) const {
        if (idx < nshort) {
            const uint32_t c0 = treal, c1 = treal + (end_shorts ? 1u : 0u);
            uint32_t L = 0, j = (uint32_t)idx;
            if (idx >= c0) {
                L = 1u + ((uint32_t)idx - c0) / c1;
                j = ((uint32_t)idx - c0) % c1;
            }
            const uint32_t pj = j == 0 ? pos[0] : (j == 1 ? pos[1] : (j == 2 ? pos[2] : pos[3]));
            return pj - L;
        }
        uint32_t p = (uint32_t)idx - nshort;
#pragma unroll
        for (uint32_t k = 0; k + 1 < kTermFew; ++k)
            if (k < treal && p + 15u >= pos[k]) p += 16u;
        return p;
    }
    // symbols in front of the next terminator of suffix s
    __device__ __forceinline__ uint32_t limit_of(uint32_t s) const {
        uint32_t next = n;
#pragma unroll
        for (int k = (int)kTermFew - 2; k >= 0; --k)
            if ((uint32_t)k < treal && pos[k] >= s) next = pos[k];
        return next - s;
    }
    __device__ __forceinline__ uint32_t masked(const Raw &raw, uint32_t s, uint32_t &tag) const {
        const uint32_t lim = limit_of(s);
        tag = lim < (uint32_t)kP16Syms ? lim : (uint32_t)kP16Syms;
        uint32_t sym = (uint32_t)(sym_word_of<2>(raw, s) >> 32);
        if (tag < (uint32_t)kP16Syms) sym = tag == 0 ? 0u : (sym & ~((1u << (2 * ((uint32_t)kP16Syms - tag))) - 1u));
        return sym;
    }
    __device__ __forceinline__ Raw load(size_t idx, const TileExtent &) const { return sym_words<2>(words, suffix_of(idx)); }
    __device__ __forceinline__ uint64_t key_of(const Raw &raw, size_t idx, const TileExtent &) const {
        uint32_t tag;
        const uint32_t sym = masked(raw, suffix_of(idx), tag);
        return ((uint64_t)sym << kP16TagBits) | tag;
    }
    __device__ __forceinline__ uint32_t val(size_t idx) const { return suffix_of(idx); }
    __device__ __forceinline__ uint32_t hist_digit_of(const Raw &raw, size_t idx, int, const TileExtent &) const {
        uint32_t tag;
        return masked(raw, suffix_of(idx), tag) >> (32 - kRadixBits);
    }
    __device__ __forceinline__ bool digits_from_window(int) const { return false; }
    __device__ __forceinline__ uint64_t window(size_t) const { return 0; }
};
// Independent records, one BUCKET per record (radix_sort_record_keys): the pairs of a tile are the suffixes at
// the tile's own text positions, the key [kRecSyms bases][4-bit length tag] of a suffix needs the end of its
// record -- the terminator of the tile's bucket, one scalar load per tile instead of a table search per suffix.
struct RecordTextSrc {
    using Raw = SymWords;
    const uint64_t *__restrict__ words;
    const uint32_t *__restrict__ term_pos;  // terminator of record k (the separator behind it; n for the last one)
    __device__ __forceinline__ Raw load(size_t idx, const TileExtent &) const { return sym_words<2>(words, idx); }
    __device__ __forceinline__ uint32_t key_of(const Raw &raw, size_t idx, const TileExtent &ext) const {
        const uint64_t w = sym_word_of<2>(raw, idx);
        const uint32_t lim = term_pos[ext.bucket] - (uint32_t)idx;
        const uint32_t tag = lim < (uint32_t)kRecSyms ? lim : (uint32_t)kRecSyms;
        uint32_t sym = (uint32_t)(w >> (64 - kRecSyms * 2));
        if (tag < (uint32_t)kRecSyms) sym &= ~((1u << (2 * (kRecSyms - (int)tag))) - 1u);
        return (sym << kRecTagBits) | tag;
    }
    __device__ __forceinline__ uint32_t val(size_t idx) const { return (uint32_t)idx; }
    __device__ __forceinline__ uint32_t hist_digit_of(const Raw &raw, size_t idx, int shift, const TileExtent &ext) const {
        return digit_of(key_of(raw, idx, ext), shift);
    }
    __device__ __forceinline__ bool digits_from_window(int) const { return false; }
    __device__ __forceinline__ uint64_t window(size_t) const { return 0; }
};
// (position, value) pairs of a block-diagonal permutation (RecordScatterPlan): the key is the position
// inside the record, ext.aux = first position of the tile's record
struct LocalIdxSrc {
    using Raw = uint32_t;
    const uint32_t *__restrict__ idx;
    const uint32_t *__restrict__ vals;
    __device__ __forceinline__ Raw load(size_t i, const TileExtent &) const { return idx[i]; }
    __device__ __forceinline__ uint32_t key_of(Raw raw, size_t, const TileExtent &ext) const { return raw - ext.aux; }
    __device__ __forceinline__ uint32_t hist_digit_of(Raw raw, size_t, int shift, const TileExtent &ext) const { return digit_of(raw - ext.aux, shift); }
    __device__ __forceinline__ uint32_t val(size_t i) const { return vals[i]; }
    __device__ __forceinline__ bool digits_from_window(int) const { return false; }
    __device__ __forceinline__ uint64_t window(size_t) const { return 0; }
};

// the same with the list position + 1 as a second value (RankSrc): the block-diagonal permutation of a merged batch
// delivers the inverse suffix array on the way, too
struct LocalRankSrc {
    using Raw = uint32_t;
    const uint32_t *__restrict__ idx;
    const uint32_t *__restrict__ vals;
    __device__ __forceinline__ Raw load(size_t i, const TileExtent &) const { return idx[i]; }
    __device__ __forceinline__ uint32_t key_of(Raw raw, size_t, const TileExtent &ext) const { return raw - ext.aux; }
    __device__ __forceinline__ uint32_t hist_digit_of(Raw raw, size_t, int shift, const TileExtent &ext) const { return digit_of(raw - ext.aux, shift); }
    __device__ __forceinline__ uint64_t val(size_t i) const { return (uint64_t)vals[i] | ((uint64_t)((uint32_t)i + 1u) << 32); }
    __device__ __forceinline__ bool digits_from_window(int) const { return false; }
    __device__ __forceinline__ uint64_t window(size_t) const { return 0; }
};

template <typename S, typename = void> struct HasWindowDigits : std::false_type {};
template <typename S> struct HasWindowDigits<S, std::void_t<decltype(&S::window_digit)>> : std::true_type {};

template <typename KeyT, typename Src>
__global__ __launch_bounds__(kThreads) void rs_hist_kernel(Src src, size_t n, int shift,
                                                           uint32_t *__restrict__ tile_hist,
                                                           uint32_t num_tiles, SegView seg) {
    // four interleaved copies of the histogram, one per lane & 3: a pass whose digit takes only a few
    // values (the lowest key byte is mostly the length tag) would otherwise send all 64 lanes of an
    // LDS atomic to the same few addresses, which the LDS executes one after the other
    constexpr int kCopies = 4;
    __shared__ __align__(16) uint32_t hist[kBins * kCopies];
    for (int i = threadIdx.x; i < kBins * kCopies; i += kThreads) hist[i] = 0;
    // XCD-contiguous tile ranges, as in the scatter kernel: the table is bin-major, so the 256 counts of a
    // tile go to 256 different lines, each shared with the 15 neighbouring tiles -- written from one XCD
    // those 4-byte writes merge in its L2; dealt round-robin over the XCDs every one of them reached HBM
    // as a partial line (67 M of them per pass at 2^30 keys): 11.7 -> 8.6 ms per step for all histograms.
    const uint32_t tile = xcd_tile(blockIdx.x, num_tiles);
    if (tile == 0xffffffffu) return;
    __syncthreads();
    const TileExtent ext = tile_extent(tile, n, num_tiles, seg);
    const uint32_t copy = threadIdx.x & (kCopies - 1);
    bool windowed = false;
    if constexpr (HasWindowDigits<Src>::value) {
        if (src.digits_from_window(shift)) {
            static_assert(kKeysPerThread == 16, "16 two-bit symbols and an 8-bit digit fit one 64-bit window");
            windowed = true;
            const uint32_t local0 = threadIdx.x * (uint32_t)kKeysPerThread;
            const uint64_t w = local0 < ext.count ? src.window(ext.first + local0) : 0ull;
#pragma unroll
            for (int j = 0; j < kKeysPerThread; ++j)
                if (local0 + (uint32_t)j < ext.count)
                    atomicAdd(&hist[src.window_digit(w, j, ext.first + local0) * kCopies + copy], 1u);
        }
    }
    if (!windowed) {
        // all loads first: the compiler does not move loads across the LDS atomics (elements past the end of
        // the tile load its first element again: no branch around a load, nothing waits in between)
        typename Src::Raw k[kKeysPerThread];
#pragma unroll
        for (int j = 0; j < kKeysPerThread; ++j) {
            const uint32_t local = (uint32_t)j * kThreads + threadIdx.x;
            k[j] = src.load(ext.first + (local < ext.count ? local : 0u), ext);
        }
#pragma unroll
        for (int j = 0; j < kKeysPerThread; ++j) {
            const uint32_t local = (uint32_t)j * kThreads + threadIdx.x;
            if (local < ext.count)
                atomicAdd(&hist[src.hist_digit_of(k[j], ext.first + local, shift, ext) * kCopies + copy], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < kBins) {
        const uint4 c4 = reinterpret_cast<const uint4 *>(hist)[threadIdx.x];
        tile_hist[ext.hist0 + (size_t)threadIdx.x * ext.hstride] = c4.x + c4.y + c4.z + c4.w;
    }
}

// (kTimed, NOLZSS_SCATTER_PHASES: cycles per phase of a workgroup, summed over every 64th workgroup by its first thread;
// the timed instantiation waits for its loads before it ranks so that the two can be told apart.  Round 4, a segmented
// u32 pass at 2^30 pairs, 28.8 k cycles = 12 us per workgroup of which: start-up, descriptor, counters zeroed 2.0 k; the
// 32 loads ISSUED 5.6 k (the memory pipe takes them at its own pace; they have arrived when the last one is out);
// ranking 6.9 k; offsets 3.2 k (half of it the gather of the tile's 256 base offsets); keys staged 1.2 k, stored 3.8 k;
// values staged 0.7 k, stored 2.4 k, drained 2.9 k.  Half memory phases throttled by back-pressure, half compute: nothing
// to shave off one without the other growing -- asking for the base offsets first made the first key wait behind a
// gather of 256 lines (+2.7 ms per step), barriers that wait for the LDS only between the two stagings let key and
// value stores overlap and cost 1 ms, a software-pipelined form with the next tile's loads in flight needs 211 VGPRs
// (two workgroups per CU: 33.6 instead of 23.4 ms): profiles/r04_ab/scatter_phases_and_variants.txt.)
// (One returning LDS atomic per key instead of the ballots -- what local_sort_kernel does -- loses here: the u32 passes
// 5.2 -> 6.2 ms each at 2^30 pairs, three workgroups per CU already hide the ballots' VALU work behind each other's
// memory phases while the conflicting atomics queue up in the one LDS; only the pass that makes its keys from the text
// gained, 4.8 -> 4.45 ms.  gpurun_out/r4_satom, profiles/r04_ab/local_sort.txt.)
template <typename KeyT, typename OutT, typename Src, typename ValT = uint32_t, bool kTimed = false>
__global__ __launch_bounds__(kThreads, kScatterWavesPerSimd) void rs_scatter_kernel(
    Src src, OutT *__restrict__ keys_out, ValT *__restrict__ vals_out, size_t n, int shift,
    const uint32_t *__restrict__ tile_base, uint32_t num_tiles, SegView seg, unsigned long long *__restrict__ phases = nullptr) {
    const bool timed = kTimed && phases != nullptr && (blockIdx.x & 63) == 0 && threadIdx.x == 0;
    unsigned long long ck[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (kTimed && timed) ck[0] = __builtin_readcyclecounter();
    const uint32_t tile = xcd_tile(blockIdx.x, num_tiles);
    if (tile == 0xffffffffu) return;
    const TileExtent ext = tile_extent(tile, n, num_tiles, seg);
    // keys, then values, take turns here
    __shared__ __align__(16) unsigned char s_stage[(size_t)kTile * (sizeof(KeyT) > sizeof(ValT) ? sizeof(KeyT) : sizeof(ValT))];
    KeyT *s_keys = reinterpret_cast<KeyT *>(s_stage);
    ValT *s_vals = reinterpret_cast<ValT *>(s_stage);
    __shared__ uint32_t s_whist[kWaves * kBins];
    __shared__ uint32_t s_glob[kBins];
    __shared__ uint32_t s_scan[kWaves];

    const int tid = threadIdx.x;
    const int w = tid >> 6;
    const int lane = tid & 63;

    for (int i = tid; i < kWaves * kBins; i += kThreads) s_whist[i] = 0;
    __syncthreads();
    if (kTimed && timed) ck[1] = __builtin_readcyclecounter();  // (includes the descriptor load: ext is used above)

    const size_t base = ext.first;
    KeyT key[kKeysPerThread];
    ValT val[kKeysPerThread];
    uint32_t lrank[kKeysPerThread];

    // all loads of the tile go out before anything is ranked (the ranking below goes through
    // volatile LDS counters, which the compiler will not move loads across: interleaved, every row
    // would wait for its own round trip to HBM)
    // (a source that computes its keys keeps eight raw elements in flight at a time: sixteen would not fit
    // the registers next to the keys)
    constexpr int kBatch = sizeof(typename Src::Raw) > sizeof(KeyT) ? 8 : kKeysPerThread;
#pragma unroll
    for (int r0 = 0; r0 < kKeysPerThread; r0 += kBatch) {
        typename Src::Raw raw[kBatch];
#pragma unroll
        for (int r = 0; r < kBatch; ++r) {
            const uint32_t local = (uint32_t)w * kWaveSpan + (uint32_t)(r0 + r) * 64 + lane;
            raw[r] = src.load(base + (local < ext.count ? local : 0u), ext);  // (past the end: the first element again)
        }
#pragma unroll
        for (int r = 0; r < kBatch; ++r) {
            const uint32_t local = (uint32_t)w * kWaveSpan + (uint32_t)(r0 + r) * 64 + lane;
            const bool valid = local < ext.count;
            key[r0 + r] = valid ? (KeyT)src.key_of(raw[r], base + local, ext) : KeyT(0);
        }
    }
#pragma unroll
    for (int row = 0; row < kKeysPerThread; ++row) {
        const uint32_t local = (uint32_t)w * kWaveSpan + (uint32_t)row * 64 + lane;
        val[row] = local < ext.count ? src.val(base + local) : 0;
    }
    if constexpr (kTimed) {
        if (timed) ck[2] = __builtin_readcyclecounter();  // loads issued
        __builtin_amdgcn_s_waitcnt(0x0f70);                // vmcnt(0): the whole tile has arrived
        if (timed) ck[3] = __builtin_readcyclecounter();
    }
    // rank inside the wavefront: rows of 64 keys in input order (keeps the sort stable).  The lowest
    // lane of every digit group adds the group's size to the wave's counter with ONE returning LDS
    // atomic per row; the atomics of all rows are issued back to back (LDS executes a wave's
    // operations in order, so row r sees rows < r) and the results are handed to the other lanes
    // of the groups afterwards -- no row waits for the LDS round trip of the row in front.
    // lrank[row] packs, until the second loop: counter value seen by the group's first lane (11 bits,
    // <= 1024 keys per wave) | lanes of my group below me << 11 | lane of the first member << 17
    uint32_t *wcount = s_whist + w * kBins;
#pragma unroll
    for (int row = 0; row < kKeysPerThread; ++row) {
        const bool valid = (uint32_t)w * kWaveSpan + (uint32_t)row * 64 + lane < ext.count;
        const uint32_t d = digit_of(key[row], shift);
        // lanes with the same digit: the complement of the lanes that differ in some bit.  Per bit,
        // m = 0 / ~0 (bit clear / set, one v_bfe_i32), and (ballot ^ m) is the set of lanes whose bit
        // differs from mine -- six VALU instructions per bit instead of nine for the select form.
        uint32_t diff_lo = 0, diff_hi = 0;
#pragma unroll
        for (int b = 0; b < kRadixBits; ++b) {
            const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)d, (unsigned)b, 1u);
            const uint64_t bal = __ballot((int)m < 0);
            // diff |= bal ^ m in one v_bitop3 (truth table 0xde = b | (c ^ a)): four VALU per bit and row
            diff_lo = __builtin_amdgcn_bitop3_b32(m, diff_lo, (uint32_t)bal, 0xde);
            diff_hi = __builtin_amdgcn_bitop3_b32(m, diff_hi, (uint32_t)(bal >> 32), 0xde);
        }
        const uint64_t peers = ~(((uint64_t)diff_hi << 32) | diff_lo) & __ballot(valid);
        const uint64_t below = peers & lanemask_lt();
        uint32_t seen = 0;
        if (valid && below == 0) seen = atomicAdd(&wcount[d], (uint32_t)__popcll(peers));
        lrank[row] = seen | ((uint32_t)__popcll(below) << 11) | ((uint32_t)(peers ? __builtin_ctzll(peers) : 0) << 17);
    }
#pragma unroll
    for (int row = 0; row < kKeysPerThread; ++row) {
        const uint32_t packed = lrank[row];
        lrank[row] = ((uint32_t)__shfl((int)packed, (int)(packed >> 17), 64) & 0x7ffu) + ((packed >> 11) & 63u);
    }
    if (kTimed && timed) ck[4] = __builtin_readcyclecounter();  // ranked
    __syncthreads();

    // thread = bin (the first kBins threads): turn per-wave counts into tile-local start positions
    {
        const int d = tid;
        const bool owner = tid < kBins;
        uint32_t c[kWaves], total = 0;
#pragma unroll
        for (int k = 0; k < kWaves; ++k) {
            c[k] = owner ? s_whist[k * kBins + d] : 0u;
            total += c[k];
        }
        uint32_t tile_total;
        const uint32_t bin_start = block_scan_exclusive<kWaves>(total, OpAdd<uint32_t>(), s_scan, tile_total);
        if (owner) {
            uint32_t run = bin_start;
#pragma unroll
            for (int k = 0; k < kWaves; ++k) {
                s_whist[k * kBins + d] = run;
                run += c[k];
            }
            s_glob[d] = tile_base[ext.hist0 + (size_t)d * ext.hstride] - bin_start;
        }
    }
    __syncthreads();
    if (kTimed && timed) ck[5] = __builtin_readcyclecounter();  // tile-local offsets (and the tile's bases from the table)

    // tile-local sorted position of every element (reuses lrank)
#pragma unroll
    for (int row = 0; row < kKeysPerThread; ++row) {
        const uint32_t d = digit_of(key[row], shift);
        lrank[row] += s_whist[w * kBins + d];
    }
#pragma unroll
    for (int row = 0; row < kKeysPerThread; ++row) {
        if ((uint32_t)w * kWaveSpan + (uint32_t)row * 64 + lane < ext.count) s_keys[lrank[row]] = key[row];
    }
    __syncthreads();
    if (kTimed && timed) ck[6] = __builtin_readcyclecounter();  // keys staged

    const uint32_t count = ext.count;
    uint32_t gpos[kKeysPerThread];
#pragma unroll
    for (int j = 0; j < kKeysPerThread; ++j) {
        const uint32_t p = (uint32_t)j * kThreads + tid;
        if (p < count) {
            const KeyT k = s_keys[p];
            const uint32_t d = digit_of(k, shift);
            gpos[j] = s_glob[d] + p;
            keys_out[gpos[j]] = (OutT)k;
        }
    }
    __syncthreads();
    if (kTimed && timed) ck[7] = __builtin_readcyclecounter();  // key stores issued (and, through the barrier, drained)
#pragma unroll
    for (int row = 0; row < kKeysPerThread; ++row) {
        if ((uint32_t)w * kWaveSpan + (uint32_t)row * 64 + lane < ext.count) s_vals[lrank[row]] = val[row];
    }
    __syncthreads();
    if (kTimed && timed) ck[8] = __builtin_readcyclecounter();  // values staged
#pragma unroll
    for (int j = 0; j < kKeysPerThread; ++j) {
        const uint32_t p = (uint32_t)j * kThreads + tid;
        if (p < count) vals_out[gpos[j]] = s_vals[p];
    }
    if constexpr (kTimed) {
        if (timed) {
            ck[9] = __builtin_readcyclecounter();  // value stores issued
            for (int k = 0; k < 9; ++k) atomicAdd(phases + k, ck[k + 1] - ck[k]);
            __builtin_amdgcn_s_waitcnt(0x0f70);
            atomicAdd(phases + 9, (unsigned long long)__builtin_readcyclecounter() - ck[9]);  // value stores drained
            atomicAdd(phases + 10, 1ull);
        }
    }
}

// ---- fused records (round 4, A/B) ------------------------------------------------------------------------
// The same pass on pairs that travel as ONE 64-bit word [key : 32 | value : 32]: one staging buffer of 8-byte
// records, one store loop, bin runs of 16 x 8 = 128 bytes on average where the split form writes two streams of
// 64-byte runs.  The price is paid by the histogram kernel of the pass, which reads the records (8 B per pair)
// where the split form reads the keys alone (4 B).  kSplitOut: the last pass of a sort hands the halves to
// separate arrays (the suffix array and the key words the regroup kernel reads).
struct RecArraySrc {
    using Raw = uint64_t;
    static constexpr bool kDigitInRecord = true;
    const uint64_t *__restrict__ recs;
    __device__ __forceinline__ Raw load(size_t idx, const TileExtent &) const { return recs[idx]; }
    __device__ __forceinline__ uint64_t rec_of(Raw raw, size_t, const TileExtent &, uint32_t &) const { return raw; }
};
// the most-significant-digit pass of the 16-base key sort (Text16Src), writing records: [stored key word | suffix];
// its digit -- the first four bases -- is not part of the record
struct TextRec16Src {
    using Raw = SymWords;
    static constexpr bool kDigitInRecord = false;
    Text16Src t;
    __device__ __forceinline__ Raw load(size_t idx, const TileExtent &ext) const { return t.load(idx, ext); }
    __device__ __forceinline__ uint64_t rec_of(const Raw &raw, size_t idx, const TileExtent &ext, uint32_t &digit) const {
        const uint64_t k = t.key_of(raw, idx, ext);  // [32 key bits][8-bit tag]
        digit = (uint32_t)(k >> 32);
        return (k << 32) | t.suffix_of(idx);
    }
};

template <typename Src, bool kSplitOut>
__global__ __launch_bounds__(kThreads) void rs_scatter_rec_kernel(Src src, uint64_t *__restrict__ rec_out,
                                                                  uint32_t *__restrict__ keys_out,
                                                                  uint32_t *__restrict__ vals_out, size_t n, int shift,
                                                                  const uint32_t *__restrict__ tile_base,
                                                                  uint32_t num_tiles, SegView seg) {
    static_assert(kKeysPerThread == 16, "digits of a thread pack into four registers");
    const uint32_t tile = xcd_tile(blockIdx.x, num_tiles);
    if (tile == 0xffffffffu) return;
    const TileExtent ext = tile_extent(tile, n, num_tiles, seg);
    __shared__ __align__(16) uint64_t s_rec[kTile];
    __shared__ uint32_t s_whist[kWaves * kBins];
    __shared__ uint32_t s_glob[kBins];
    __shared__ uint32_t s_scan[kWaves];
    __shared__ uint8_t s_dig[Src::kDigitInRecord ? 4 : kTile];

    const int tid = threadIdx.x;
    const int w = tid >> 6;
    const int lane = tid & 63;
    for (int i = tid; i < kWaves * kBins; i += kThreads) s_whist[i] = 0;
    __syncthreads();

    const size_t base = ext.first;
    uint64_t rec[kKeysPerThread];
    uint32_t lrank[kKeysPerThread];
    uint32_t dpk[kKeysPerThread / 4] = {0, 0, 0, 0};  // (digits that are not part of the record, four per register)
    auto digit_at = [&](int row) -> uint32_t {
        if constexpr (Src::kDigitInRecord)
            return digit_of(rec[row], 32 + shift);
        else
            return (dpk[row >> 2] >> (8 * (row & 3))) & 255u;
    };
    constexpr int kBatch = sizeof(typename Src::Raw) > sizeof(uint64_t) ? 8 : kKeysPerThread;
#pragma unroll
    for (int r0 = 0; r0 < kKeysPerThread; r0 += kBatch) {
        typename Src::Raw raw[kBatch];
#pragma unroll
        for (int r = 0; r < kBatch; ++r) {
            const uint32_t local = (uint32_t)w * kWaveSpan + (uint32_t)(r0 + r) * 64 + lane;
            raw[r] = src.load(base + (local < ext.count ? local : 0u), ext);  // (past the end: the first element again)
        }
#pragma unroll
        for (int r = 0; r < kBatch; ++r) {
            const uint32_t local = (uint32_t)w * kWaveSpan + (uint32_t)(r0 + r) * 64 + lane;
            const bool valid = local < ext.count;
            uint32_t d = 0;
            const uint64_t x = src.rec_of(raw[r], base + (valid ? local : 0u), ext, d);
            rec[r0 + r] = valid ? x : 0ull;
            if constexpr (!Src::kDigitInRecord) dpk[(r0 + r) >> 2] |= (valid ? d : 0u) << (8 * ((r0 + r) & 3));
        }
    }
    // ranking inside the wavefront, exactly as in rs_scatter_kernel
    uint32_t *wcount = s_whist + w * kBins;
#pragma unroll
    for (int row = 0; row < kKeysPerThread; ++row) {
        const bool valid = (uint32_t)w * kWaveSpan + (uint32_t)row * 64 + lane < ext.count;
        const uint32_t d = digit_at(row);
        uint32_t diff_lo = 0, diff_hi = 0;
#pragma unroll
        for (int b = 0; b < kRadixBits; ++b) {
            const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)d, (unsigned)b, 1u);
            const uint64_t bal = __ballot((int)m < 0);
            diff_lo = __builtin_amdgcn_bitop3_b32(m, diff_lo, (uint32_t)bal, 0xde);
            diff_hi = __builtin_amdgcn_bitop3_b32(m, diff_hi, (uint32_t)(bal >> 32), 0xde);
        }
        const uint64_t peers = ~(((uint64_t)diff_hi << 32) | diff_lo) & __ballot(valid);
        const uint64_t below = peers & lanemask_lt();
        uint32_t seen = 0;
        if (valid && below == 0) seen = atomicAdd(&wcount[d], (uint32_t)__popcll(peers));
        lrank[row] = seen | ((uint32_t)__popcll(below) << 11) | ((uint32_t)(peers ? __builtin_ctzll(peers) : 0) << 17);
    }
#pragma unroll
    for (int row = 0; row < kKeysPerThread; ++row) {
        const uint32_t packed = lrank[row];
        lrank[row] = ((uint32_t)__shfl((int)packed, (int)(packed >> 17), 64) & 0x7ffu) + ((packed >> 11) & 63u);
    }
    __syncthreads();
    {
        const int d = tid;
        const bool owner = tid < kBins;
        uint32_t c[kWaves], total = 0;
#pragma unroll
        for (int k = 0; k < kWaves; ++k) {
            c[k] = owner ? s_whist[k * kBins + d] : 0u;
            total += c[k];
        }
        uint32_t tile_total;
        const uint32_t bin_start = block_scan_exclusive<kWaves>(total, OpAdd<uint32_t>(), s_scan, tile_total);
        if (owner) {
            uint32_t run = bin_start;
#pragma unroll
            for (int k = 0; k < kWaves; ++k) {
                s_whist[k * kBins + d] = run;
                run += c[k];
            }
            s_glob[d] = tile_base[ext.hist0 + (size_t)d * ext.hstride] - bin_start;
        }
    }
    __syncthreads();
#pragma unroll
    for (int row = 0; row < kKeysPerThread; ++row) {
        const uint32_t d = digit_at(row);
        lrank[row] += s_whist[w * kBins + d];
    }
#pragma unroll
    for (int row = 0; row < kKeysPerThread; ++row) {
        if ((uint32_t)w * kWaveSpan + (uint32_t)row * 64 + lane < ext.count) {
            s_rec[lrank[row]] = rec[row];
            if constexpr (!Src::kDigitInRecord) s_dig[lrank[row]] = (uint8_t)digit_at(row);
        }
    }
    __syncthreads();
    const uint32_t count = ext.count;
#pragma unroll
    for (int j = 0; j < kKeysPerThread; ++j) {
        const uint32_t p = (uint32_t)j * kThreads + tid;
        if (p < count) {
            const uint64_t x = s_rec[p];
            uint32_t d;
            if constexpr (Src::kDigitInRecord)
                d = digit_of(x, 32 + shift);
            else
                d = s_dig[p];
            const uint32_t g = s_glob[d] + p;
            if constexpr (kSplitOut) {
                keys_out[g] = (uint32_t)(x >> 32);
                vals_out[g] = (uint32_t)x;
            } else {
                rec_out[g] = x;
            }
        }
    }
}

template <typename Src, bool kSplitOut, typename HistSrc>
void radix_pass_rec(Src src, HistSrc hsrc, int hist_shift, uint64_t *rec_out, uint32_t *keys_out, uint32_t *vals_out,
                    size_t n, int shift, uint32_t *hist, uint32_t num_tiles, double hist_bytes, double scatter_bytes,
                    Arena &arena, hipStream_t stream, Profiler *prof, const SegView &seg = SegView{}) {
    {
        ProfScope ps(prof, "rs_hist", stream, hist_bytes);
        rs_hist_kernel<uint64_t, HistSrc><<<xcd_grid(num_tiles), kThreads, 0, stream>>>(hsrc, n, hist_shift, hist, num_tiles, seg);
        KERNEL_CHECK();
    }
    {
        ProfScope ps(prof, "rs_scan", stream, 8.0 * (double)kBins * num_tiles);
        scan_exclusive_add_u32(hist, hist, (size_t)kBins * num_tiles, nullptr, arena, stream);
    }
    {
        ProfScope ps(prof, Src::kDigitInRecord ? "rs_scatter.rec" : "rs_scatter.text", stream, scatter_bytes);
        rs_scatter_rec_kernel<Src, kSplitOut><<<xcd_grid(num_tiles), kThreads, 0, stream>>>(src, rec_out, keys_out, vals_out, n,
                                                                                     shift, hist, num_tiles, seg);
        KERNEL_CHECK();
    }
}

// one pass: histogram, scan, scatter
template <typename KeyT, typename OutT, typename Src, typename ValT = uint32_t>
void radix_pass(Src src, OutT *keys_out, ValT *vals_out, size_t n, int shift, uint32_t *hist, uint32_t num_tiles,
                double hist_bytes, double scatter_bytes, Arena &arena, hipStream_t stream, Profiler *prof,
                const SegView &seg = SegView{}) {
    {
        ProfScope ps(prof, "rs_hist", stream, hist_bytes);
        rs_hist_kernel<KeyT, Src><<<xcd_grid(num_tiles), kThreads, 0, stream>>>(src, n, shift, hist, num_tiles, seg);
        KERNEL_CHECK();
    }
    {
        ProfScope ps(prof, "rs_scan", stream, 8.0 * (double)kBins * num_tiles);
        scan_exclusive_add_u32(hist, hist, (size_t)kBins * num_tiles, nullptr, arena, stream);
    }
    {
        // classes of launches, so that the bandwidth of the large passes can be told from the many
        // small sorts of the doubling rounds: rs_scatter.{text|u64|u32}[.small]
        const bool small = n < (size_t(1) << 24);
        const char *cls = (std::is_same<Src, ArraySrc<KeyT>>::value || std::is_same<Src, LocalIdxSrc>::value ||
                           std::is_same<Src, LocalRankSrc>::value || std::is_same<Src, RankSrc>::value ||
                           std::is_same<Src, PairSrc>::value)
                              ? (sizeof(KeyT) == 8 ? (small ? "rs_scatter.u64.small" : "rs_scatter.u64")
                                                   : (small ? "rs_scatter.u32.small" : "rs_scatter.u32"))
                              : "rs_scatter.text";
        ProfScope ps(prof, cls, stream, scatter_bytes);
        // (Round 3, NOLZSS_SORT_TILE=8192: tiles of 8192 pairs on 512 threads -- bin runs of a full 128-byte line.  The
        // kernel needs 134 VGPRs and two such workgroups per CU allow 128: 14 registers spilled (72 in the text
        // pass); u32 passes 3857 -> 3017 GB/s, text pass 1811 -> 1314, histograms + scans 10.8 -> 9.0 ms per step,
        // step 118.7 -> 129.3 ms, profiles/r03_ab_tile8k.txt.  4096 stays.)
        // (Round 2 tried 512 threads with 8 keys each -- 75 instead of 139 VGPRs, 24 instead of 12 wavefronts
        // per CU -- and separate LDS buffers for keys and values: the u32 passes stayed at 3.9 TB/s at 2^30
        // pairs either way.  The pass is bound by its scattered 64-byte write runs, not by latency hiding.)
        const uint32_t grid = xcd_grid(num_tiles);
        static const bool want_phases = getenv("NOLZSS_SCATTER_PHASES") != nullptr;
        if (want_phases && n >= (size_t(1) << 24) && std::is_same<Src, ArraySrc<KeyT>>::value) {
            unsigned long long *d_ph = arena.alloc<unsigned long long>(12);
            HIP_CHECK(hipMemsetAsync(d_ph, 0, 12 * sizeof(unsigned long long), stream));
            rs_scatter_kernel<KeyT, OutT, Src, ValT, true><<<grid, kThreads, 0, stream>>>(src, keys_out, vals_out, n, shift, hist,
                                                                                          num_tiles, seg, d_ph);
            KERNEL_CHECK();
            unsigned long long h[12];
            HIP_CHECK(hipMemcpyAsync(h, d_ph, sizeof(h), hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
            const double wn = h[10] ? (double)h[10] : 1.0;
            fprintf(stderr, "[nolzss] rs_scatter phases (cycles per workgroup, %llu sampled, shift %d, %s): start-up + descriptor + zero %.0f  "
                            "loads issued %.0f  loads arrive %.0f  ranking %.0f  offsets %.0f  stage keys %.0f  store keys %.0f  stage values %.0f  "
                            "store values %.0f  drain %.0f\n",
                    h[10], shift, seg.desc ? "segmented" : "whole array", h[0] / wn, h[1] / wn, h[2] / wn, h[3] / wn, h[4] / wn, h[5] / wn,
                    h[6] / wn, h[7] / wn, h[8] / wn, h[9] / wn);
        } else {
            rs_scatter_kernel<KeyT, OutT, Src, ValT><<<grid, kThreads, 0, stream>>>(src, keys_out, vals_out, n, shift, hist,
                                                                                    num_tiles, seg);
            KERNEL_CHECK();
        }
    }
}

template <typename KeyT>
int radix_sort_impl(KeyT *keys[2], uint32_t *vals[2], size_t n, const int *shifts, int npasses, Arena &arena,
                    hipStream_t stream, Profiler *prof, int first_pass = 0) {
    if (n == 0 || npasses == 0) return 0;
    if (sizeof(KeyT) == 8)
        for (int p = 0; p < npasses; ++p)
            if ((shifts[p] & 31) + kRadixBits > 32) throw HipError("radix sort: a digit may not straddle the key halves");
    const size_t m = arena.mark();
    const uint32_t num_tiles = (uint32_t)div_up(n, kTile);
    uint32_t *hist = arena.alloc<uint32_t>((size_t)kBins * num_tiles);
    int cur = first_pass & 1;
    for (int p = first_pass; p < npasses; ++p) {
        // algorithmic bytes of one scatter launch: every (key, value) pair read once and
        // written once = 2 * (sizeof(key) + 4) bytes per pair
        radix_pass<KeyT, KeyT>(ArraySrc<KeyT>{keys[cur], vals[cur]}, keys[cur ^ 1], vals[cur ^ 1], n, shifts[p], hist,
                         num_tiles, (double)sizeof(KeyT) * (double)n, 2.0 * (sizeof(KeyT) + 4.0) * (double)n, arena,
                         stream, prof);
        cur ^= 1;
    }
    arena.rewind(m);
    return cur;
}

__global__ __launch_bounds__(kThreads) void plain_scatter_kernel(const uint32_t *__restrict__ idx,
                                                                 const uint32_t *__restrict__ val, size_t count,
                                                                 uint32_t *__restrict__ out, uint32_t n_out,
                                                                 uint32_t num_tiles) {
    const uint32_t tile = xcd_tile(blockIdx.x, num_tiles);
    if (tile == 0xffffffffu) return;
    const size_t base = (size_t)tile * kTile;
#pragma unroll
    for (int j = 0; j < kKeysPerThread; ++j) {
        const size_t k = base + (size_t)j * kThreads + threadIdx.x;
        if (k < count) {
            const uint32_t i = idx[k];
            if (i < n_out) out[i] = val[k];
        }
    }
}

// out[idx[k]] = low half, out2[idx[k]] = high half of packed[k] (small inputs of permute_packed)
__global__ __launch_bounds__(kThreads) void plain_packed_scatter_kernel(const uint32_t *__restrict__ idx,
                                                                        const uint64_t *__restrict__ packed, size_t count,
                                                                        uint32_t *__restrict__ out, uint32_t *__restrict__ out2) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += stride) {
        const uint32_t i = idx[k];
        const uint64_t v = packed[k];
        if (i < count) {
            out[i] = (uint32_t)v;
            out2[i] = (uint32_t)(v >> 32);
        }
    }
}

// out2[idx[k]] = k + 1 (small inputs: the second value of bucketed_scatter's out2 form, written directly)
__global__ __launch_bounds__(kThreads) void plain_rank_scatter_kernel(const uint32_t *__restrict__ idx, size_t count,
                                                                      uint32_t *__restrict__ out2, uint32_t n_out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += stride) {
        const uint32_t i = idx[k];
        if (i < n_out) out2[i] = (uint32_t)k + 1u;
    }
}

constexpr int kWindowBitsMax = 14;  // 2^14 entries = 64 KiB of LDS

// permutation scatter, final step: the pairs of window w sit at list positions [w*W, (w+1)*W)
// (IdxT = uint16_t: the last partition pass kept only the low 16 bits of every index -- what lies above the
// window bits is implied by the position in the list)
template <typename IdxT>
__global__ __launch_bounds__(kThreads) void window_scatter_kernel(const IdxT *__restrict__ idx,
                                                                  const uint32_t *__restrict__ val,
                                                                  uint32_t *__restrict__ out, uint32_t n_out,
                                                                  int window_bits) {
    __shared__ uint32_t s_out[1 << kWindowBitsMax];
    const uint32_t W = 1u << window_bits;
    const size_t base = (size_t)blockIdx.x << window_bits;
    const uint32_t len = (uint32_t)((n_out - base < (size_t)W) ? (n_out - base) : (size_t)W);
    // eight (index, value) pairs per thread in flight at a time
    constexpr int kBatch = 8;
    for (uint32_t t0 = 0; t0 < len; t0 += kBatch * kThreads) {
        uint32_t ii[kBatch], vv[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const uint32_t t = t0 + (uint32_t)j * kThreads + threadIdx.x;
            const size_t at = base + (t < len ? t : 0u);  // (no branch around the loads)
            ii[j] = (uint32_t)idx[at];
            vv[j] = val[at];
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const uint32_t t = t0 + (uint32_t)j * kThreads + threadIdx.x;
            if (t < len) s_out[ii[j] & (W - 1u)] = vv[j];  // (the window starts at a multiple of W)
        }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < len; t += kThreads) out[base + t] = s_out[t];
}

// the same for pairs that carry two values in one 64-bit word (low half -> out, high half -> out2): both windows
// are assembled side by side in 128 KiB of LDS by one workgroup of 1024 threads per CU
constexpr int kWindow2Threads = 1024;
__global__ __launch_bounds__(kWindow2Threads) void window_scatter2_kernel(const uint16_t *__restrict__ idx,
                                                                          const uint64_t *__restrict__ val,
                                                                          uint32_t *__restrict__ out,
                                                                          uint32_t *__restrict__ out2, uint32_t n_out,
                                                                          int window_bits) {
    __shared__ uint32_t s_out[2 << kWindowBitsMax];
    const uint32_t W = 1u << window_bits;
    uint32_t *s_a = s_out, *s_b = s_out + W;
    const size_t base = (size_t)blockIdx.x << window_bits;
    const uint32_t len = (uint32_t)((n_out - base < (size_t)W) ? (n_out - base) : (size_t)W);
    constexpr int kBatch = 4;
    for (uint32_t t0 = 0; t0 < len; t0 += kBatch * kWindow2Threads) {
        uint32_t ii[kBatch];
        uint64_t vv[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const uint32_t t = t0 + (uint32_t)j * kWindow2Threads + threadIdx.x;
            const size_t at = base + (t < len ? t : 0u);  // (no branch around the loads)
            ii[j] = (uint32_t)idx[at];
            vv[j] = val[at];
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const uint32_t t = t0 + (uint32_t)j * kWindow2Threads + threadIdx.x;
            if (t < len) {
                s_a[ii[j] & (W - 1u)] = (uint32_t)vv[j];
                s_b[ii[j] & (W - 1u)] = (uint32_t)(vv[j] >> 32);
            }
        }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < len; t += kWindow2Threads) {
        out[base + t] = s_a[t];
        out2[base + t] = s_b[t];
    }
}

// the same for the windows of a RecordScatterPlan: window b = list elements [win[3b], +win[3b+2]) -> target
// elements [win[3b+1], +win[3b+2]); the low window_bits of an index are its place in the window
template <typename IdxT>
__global__ __launch_bounds__(kThreads) void record_window_scatter_kernel(const IdxT *__restrict__ idx,
                                                                         const uint32_t *__restrict__ val,
                                                                         uint32_t *__restrict__ out,
                                                                         const uint32_t *__restrict__ win,
                                                                         int window_bits) {
    __shared__ uint32_t s_out[1 << kWindowBitsMax];
    const uint32_t W = 1u << window_bits;
    const size_t base = win[3 * (size_t)blockIdx.x];
    const size_t obase = win[3 * (size_t)blockIdx.x + 1];
    const uint32_t len = win[3 * (size_t)blockIdx.x + 2];
    constexpr int kBatch = 8;
    for (uint32_t t0 = 0; t0 < len; t0 += kBatch * kThreads) {
        uint32_t ii[kBatch], vv[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const uint32_t t = t0 + (uint32_t)j * kThreads + threadIdx.x;
            const size_t at = base + (t < len ? t : 0u);  // (no branch around the loads)
            ii[j] = (uint32_t)idx[at];
            vv[j] = val[at];
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const uint32_t t = t0 + (uint32_t)j * kThreads + threadIdx.x;
            if (t < len) s_out[ii[j] & (W - 1u)] = vv[j];
        }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < len; t += kThreads) out[obase + t] = s_out[t];
}

// two values per pair in one 64-bit word (low half -> out, high half -> out2), as window_scatter2_kernel
__global__ __launch_bounds__(kWindow2Threads) void record_window_scatter2_kernel(const uint16_t *__restrict__ idx,
                                                                                 const uint64_t *__restrict__ val,
                                                                                 uint32_t *__restrict__ out,
                                                                                 uint32_t *__restrict__ out2,
                                                                                 const uint32_t *__restrict__ win,
                                                                                 int window_bits) {
    __shared__ uint32_t s_out[2 << kWindowBitsMax];
    const uint32_t W = 1u << window_bits;
    uint32_t *s_a = s_out, *s_b = s_out + W;
    const size_t base = win[3 * (size_t)blockIdx.x];
    const size_t obase = win[3 * (size_t)blockIdx.x + 1];
    const uint32_t len = win[3 * (size_t)blockIdx.x + 2];
    constexpr int kBatch = 4;
    for (uint32_t t0 = 0; t0 < len; t0 += kBatch * kWindow2Threads) {
        uint32_t ii[kBatch];
        uint64_t vv[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const uint32_t t = t0 + (uint32_t)j * kWindow2Threads + threadIdx.x;
            const size_t at = base + (t < len ? t : 0u);  // (no branch around the loads)
            ii[j] = (uint32_t)idx[at];
            vv[j] = val[at];
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const uint32_t t = t0 + (uint32_t)j * kWindow2Threads + threadIdx.x;
            if (t < len) {
                s_a[ii[j] & (W - 1u)] = (uint32_t)vv[j];
                s_b[ii[j] & (W - 1u)] = (uint32_t)(vv[j] >> 32);
            }
        }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < len; t += kWindow2Threads) {
        out[obase + t] = s_a[t];
        out2[obase + t] = s_b[t];
    }
}

__global__ void separator_scatter_kernel(const uint32_t *__restrict__ sep, uint32_t count,
                                         const uint32_t *__restrict__ idx, const uint32_t *__restrict__ val,
                                         uint32_t *__restrict__ out, uint32_t *__restrict__ err,
                                         uint32_t *__restrict__ out2) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const uint32_t r = sep[2 * k], p = sep[2 * k + 1];
    if (idx[r] != p) atomicOr(err, 1u);  // the separator suffix is the first of its record
    out[p] = val[r];
    if (out2) out2[p] = r + 1u;
}

}  // namespace

bool record_scatter_plan(const std::vector<uint32_t> &h_terms, uint32_t n, Arena &arena, hipStream_t stream,
                         RecordScatterPlan &plan) {
    plan = RecordScatterPlan{};
    const uint32_t nb = (uint32_t)h_terms.size();
    // (partial tiles and windows cost 4096 / the average record: NOLZSS_REC_BUCKET_MIN, as for the key sort)
    static const uint64_t rec_min =
        getenv("NOLZSS_REC_BUCKET_MIN") ? (uint64_t)atoll(getenv("NOLZSS_REC_BUCKET_MIN")) : (uint64_t(1) << 16);
    if (nb < 2 || h_terms.back() != n || rec_min == 0 || (uint64_t)nb * rec_min > (uint64_t)n) return false;
    constexpr int wb = kWindowBitsMax;
    // bucket k = the BASES of record k: ranks [first_k, end_k) hold positions [start_k, start_k + len_k)
    // (the separator behind a record is the smallest suffix of the record: its first rank)
    std::vector<uint32_t> tab(5 * ((size_t)nb + 1)), sep(2 * ((size_t)nb - 1));
    uint32_t *h_first = tab.data(), *h_tile0 = h_first + nb + 1, *h_prev = h_tile0 + nb + 1, *h_next = h_prev + nb + 1,
             *h_aux = h_next + nb + 1;
    std::vector<uint32_t> win;
    uint32_t start = 0, dense = 0;
    h_tile0[0] = 0;
    for (uint32_t k = 0; k < nb; ++k) {
        const uint32_t end = k + 1 < nb ? h_terms[k] + 1 : n;     // end of the record's ranks / positions
        const uint32_t len = (k + 1 < nb ? h_terms[k] : n) - start;  // bases
        if (len == 0 || len > (1u << (wb + kRadixBits))) return false;
        h_first[k] = k + 1 < nb ? start + 1 : start;  // first base rank
        h_aux[k] = start;
        {
            // tiles behind the first one start at multiples of 64 elements: a wavefront's 64-lane loads are
            // then aligned to their 256 bytes (a bucket starts wherever its record does; unaligned, every row
            // of a tile touches three lines instead of two and the pass ran 1.4 x slower)
            const uint32_t f0 = k + 1 < nb ? start + 1 : start;
            const uint32_t c0 = (uint32_t)kTile - f0 % 64u;
            h_tile0[k + 1] = h_tile0[k] + (len <= c0 ? 1u : 1u + (uint32_t)div_up((size_t)(len - c0), kTile));
        }
        h_prev[k] = k ? k - 1 : 0xffffffffu;
        h_next[k] = k + 1 < nb ? k + 1 : 0xffffffffu;
        if (k + 1 < nb) {
            sep[2 * (size_t)k] = start;             // rank of the separator suffix
            sep[2 * (size_t)k + 1] = h_terms[k];    // its position
        }
        for (uint32_t w0 = 0; w0 < len; w0 += 1u << wb) {
            win.push_back(dense + w0);
            win.push_back(start + w0);
            win.push_back(len - w0 < (1u << wb) ? len - w0 : (1u << wb));
        }
        dense += len;
        start = end;
    }
    h_first[nb] = n;
    h_prev[nb] = h_next[nb] = h_aux[nb] = 0;
    // (the separator ranks lie between the buckets and belong to none: the tile descriptors are written here,
    // on the host, instead of by seg_desc_kernel, whose buckets follow each other without gaps)
    const uint32_t num_tiles = h_tile0[nb];
    std::vector<uint32_t> desc((size_t)num_tiles * kSegDescWords);
    {
        uint32_t s0 = 0;
        for (uint32_t k = 0; k < nb; ++k) {
            const uint32_t len = (k + 1 < nb ? h_terms[k] : n) - s0;
            const uint32_t first = h_first[k], t0 = h_tile0[k], nt = h_tile0[k + 1] - t0;
            const uint32_t c0 = (uint32_t)kTile - first % 64u;  // elements of the first tile (see above)
            for (uint32_t local = 0; local < nt; ++local) {
                uint32_t *d = desc.data() + (size_t)(t0 + local) * kSegDescWords;
                const uint32_t f = local == 0 ? first : first + c0 + (local - 1) * (uint32_t)kTile, e = first + len;
                d[0] = f;
                const uint32_t room = local == 0 ? c0 : (uint32_t)kTile;
                d[1] = e - f < room ? e - f : room;
                d[2] = k;
                d[3] = t0 * (uint32_t)kBins + local;
                d[4] = nt;
                d[5] = first;
                d[6] = e;
                d[7] = h_prev[k];
                d[8] = h_next[k];
                d[9] = h_aux[k];
                d[10] = d[11] = 0;
            }
            s0 = k + 1 < nb ? h_terms[k] + 1 : n;
        }
    }
    uint32_t *d_desc = arena.alloc<uint32_t>(desc.size() + 4);
    uint32_t *d_win = arena.alloc<uint32_t>(win.size());
    uint32_t *d_sep = arena.alloc<uint32_t>(sep.size() + 2);
    HIP_CHECK(hipMemcpyAsync(d_desc, desc.data(), desc.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(d_win, win.data(), win.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(d_sep, sep.data(), sep.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipStreamSynchronize(stream));  // local vectors
    plan.seg.desc = d_desc;
    plan.seg.num_tiles = num_tiles;
    plan.win = d_win;
    plan.num_windows = (uint32_t)(win.size() / 3);
    plan.sep = d_sep;
    plan.num_seps = nb - 1;
    plan.n = n;
    plan.window_bits = wb;
    return true;
}

void bucketed_scatter(uint32_t *idx[2], uint32_t *val[2], size_t count, uint32_t *out, uint32_t n_out,
                      Arena &arena, hipStream_t stream, Profiler *prof, bool keep_input, bool keep_val,
                      const RecordScatterPlan *plan, uint32_t *out2) {
    if (count == 0) return;
    const size_t amark = arena.mark();
    {
        int nb = 1;
        while (nb < 32 && (1ull << nb) < (uint64_t)n_out) ++nb;
        const bool big_perm = (size_t)n_out * 4 > (size_t(64) << 20) && count > (size_t(1) << 22) && count == n_out &&
                              nb <= 2 * kRadixBits + kWindowBitsMax && !(plan && plan->seg.desc);
        if (out2 && big_perm) {
            // The permutation with TWO values per pair: out[idx[k]] = val[k] and out2[idx[k]] = k + 1.  The list
            // position is generated by the first pass and travels along as a second value: 20 + 22 + 18 bytes
            // per pair where two permutations of their own take 2 * (16 + 14 + 10) -- and, above all, the second
            // one no longer has to exist before the first (suffix_array.hip: rank[] is not written at all when
            // the direct rounds finish the suffix array).  The two values travel as one 64-bit word (RankSrc).
            const int wb = nb > 2 * kRadixBits + 10 ? nb - 2 * kRadixBits : 10;
            const uint32_t num_tiles = (uint32_t)div_up(count, kTile);
            uint32_t *hist = arena.alloc<uint32_t>((size_t)kBins * num_tiles);
            uint32_t *idx_b = keep_input ? arena.alloc<uint32_t>(count) : idx[0];
            uint64_t *packed1 = reinterpret_cast<uint64_t *>(val[1]);  // (val[1] holds 2 * count words in this form)
            uint64_t *packed2 = arena.alloc<uint64_t>(count);
            radix_pass<uint32_t, uint32_t, RankSrc, uint64_t>(RankSrc{idx[0], val[0]}, idx[1], packed1, count, wb, hist,
                                                              num_tiles, 4.0 * (double)count, 20.0 * (double)count, arena,
                                                              stream, prof);
            uint16_t *idx16 = reinterpret_cast<uint16_t *>(idx_b);
            radix_pass<uint32_t, uint16_t, PairSrc, uint64_t>(PairSrc{idx[1], packed1}, idx16, packed2, count,
                                                              wb + kRadixBits, hist, num_tiles, 4.0 * (double)count,
                                                              22.0 * (double)count, arena, stream, prof);
            {
                ProfScope ps(prof, "window_scatter", stream, 18.0 * (double)count);
                const uint32_t W = 1u << wb;
                window_scatter2_kernel<<<(unsigned)div_up(n_out, W), kWindow2Threads, 0, stream>>>(idx16, packed2, out, out2,
                                                                                                 n_out, wb);
                KERNEL_CHECK();
            }
            arena.rewind(amark);
            return;
        }
        const bool plan_path = plan && plan->seg.desc && count == n_out && n_out == plan->n;
        if (out2 && !plan_path) {  // small inputs and the other shapes: the second value by a scatter of its own
            ProfScope ps(prof, "bucket_scatter", stream, 8.0 * (double)count);
            const unsigned g = (unsigned)std::min<size_t>(div_up(count, kThreads), 256u * 16u);
            plain_rank_scatter_kernel<<<g, kThreads, 0, stream>>>(idx[0], count, out2, n_out);
            KERNEL_CHECK();
        }
    }
    if (plan && plan->seg.desc && count == n_out && n_out == plan->n) {
        // block-diagonal permutation: one pass by the window inside the record, then the windows
        uint16_t *idx16 = reinterpret_cast<uint16_t *>(idx[1]);
        uint32_t *hist = arena.alloc<uint32_t>((size_t)kBins * plan->seg.num_tiles);
        uint32_t *err = arena.alloc<uint32_t>(1);
        HIP_CHECK(hipMemsetAsync(err, 0, sizeof(uint32_t), stream));
        // (32-bit indices out of this pass measured 10 % slower end to end than the low 16 bits)
        if (out2) {  // two values per pair, as one 64-bit word (val[1] holds 2 * count words in this form)
            uint64_t *packed = reinterpret_cast<uint64_t *>(val[1]);
            radix_pass<uint32_t, uint16_t, LocalRankSrc, uint64_t>(LocalRankSrc{idx[0], val[0]}, idx16, packed, count,
                                                                   plan->window_bits, hist, plan->seg.num_tiles,
                                                                   4.0 * (double)count, 18.0 * (double)count, arena, stream,
                                                                   prof, plan->seg);
            ProfScope ps(prof, "window_scatter", stream, 18.0 * (double)count);
            record_window_scatter2_kernel<<<plan->num_windows, kWindow2Threads, 0, stream>>>(idx16, packed, out, out2, plan->win,
                                                                                            plan->window_bits);
            KERNEL_CHECK();
        } else {
            radix_pass<uint32_t, uint16_t>(LocalIdxSrc{idx[0], val[0]}, idx16, val[1], count, plan->window_bits, hist,
                                           plan->seg.num_tiles, 4.0 * (double)count, 14.0 * (double)count, arena, stream,
                                           prof, plan->seg);
            ProfScope ps(prof, "window_scatter", stream, 10.0 * (double)count);
            record_window_scatter_kernel<uint16_t><<<plan->num_windows, kThreads, 0, stream>>>(idx16, val[1], out, plan->win,
                                                                                              plan->window_bits);
            KERNEL_CHECK();
        }
        if (plan->num_seps) {
            separator_scatter_kernel<<<(unsigned)div_up(plan->num_seps, kThreads), kThreads, 0, stream>>>(
                plan->sep, plan->num_seps, idx[0], val[0], out, err, out2);
            KERNEL_CHECK();
        }
        uint32_t h_err = 0;
        HIP_CHECK(hipMemcpyAsync(&h_err, err, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        arena.rewind(amark);
        if (h_err) throw HipError("record scatter: a separator suffix is not the first of its record");
        return;
    }
    int nbits = 1;
    while (nbits < 32 && (1ull << nbits) < (uint64_t)n_out) ++nbits;
    const bool big = (size_t)n_out * 4 > (size_t(64) << 20) && count > (size_t(1) << 22);
    if (big && count == n_out) {
        // idx is a permutation of [0, n_out): radix passes by the digits above the window bits leave window
        // w = [w*W, (w+1)*W) exactly at list positions [w*W, (w+1)*W); each window is assembled in LDS and
        // written out as full lines.  Two passes reach 2^30 targets; above that a third pass takes the top
        // bits (few bins, long runs): 17 ms per 2^30 pairs where the windowed partial scatter below needed 29.
        const bool three = nbits > 2 * kRadixBits + kWindowBitsMax;
        const int wb = three ? kWindowBitsMax : (nbits > 2 * kRadixBits + 10 ? nbits - 2 * kRadixBits : 10);
        const int shifts[3] = {wb, wb + kRadixBits, wb + 2 * kRadixBits};
        // pass 1: buffer 0 -> 1; pass 2: 1 -> 0, or 1 -> a third buffer if the input must survive;
        // (pass 3: that buffer -> 1)
        radix_sort_pairs(idx, val, count, shifts, 1, arena, stream, prof);
        uint32_t *idx2[2] = {idx[1], keep_input ? arena.alloc<uint32_t>(count) : idx[0]};
        uint32_t *val2[2] = {val[1], (keep_input && keep_val) ? arena.alloc<uint32_t>(count) : val[0]};
        if (three) {
            radix_sort_pairs(idx2, val2, count, shifts + 1, 1, arena, stream, prof);
            std::swap(idx2[0], idx2[1]);
            std::swap(val2[0], val2[1]);
        }
        // last pass (top digit): only the low 16 bits of an index travel on -- the window scatter needs the
        // bits below the window size, and everything above them is the position in the list (6 instead of 8
        // bytes per pair written here and read there)
        uint16_t *idx16 = reinterpret_cast<uint16_t *>(idx2[1]);
        {
            const uint32_t num_tiles = (uint32_t)div_up(count, kTile);
            uint32_t *hist = arena.alloc<uint32_t>((size_t)kBins * num_tiles);
            radix_pass<uint32_t, uint16_t>(ArraySrc<uint32_t>{idx2[0], val2[0]}, idx16, val2[1], count,
                                           shifts[three ? 2 : 1], hist, num_tiles, 4.0 * (double)count,
                                           14.0 * (double)count, arena, stream, prof);
        }
        {
            ProfScope ps(prof, "window_scatter", stream, 10.0 * (double)count);
            const uint32_t W = 1u << wb;
            window_scatter_kernel<uint16_t><<<(unsigned)div_up(n_out, W), kThreads, 0, stream>>>(idx16, val2[1], out, n_out, wb);
            KERNEL_CHECK();
        }
        arena.rewind(amark);
        return;
    }
    int cur = 0;
    if (big) {
        // partial scatter: partition so that all writes in flight fall into 2 MiB windows of the
        // target, which one XCD's L2 can merge
        const int window_bits = 19;
        if (nbits > window_bits) {
            int shift = window_bits;
            radix_sort_pairs(idx, val, count, &shift, 1, arena, stream, prof);
            cur = 1;
            if (nbits > window_bits + kRadixBits) {
                uint32_t *idx2[2] = {idx[1], keep_input ? arena.alloc<uint32_t>(count) : idx[0]};
                uint32_t *val2[2] = {val[1], (keep_input && keep_val) ? arena.alloc<uint32_t>(count) : val[0]};
                shift = window_bits + kRadixBits;
                radix_sort_pairs(idx2, val2, count, &shift, 1, arena, stream, prof);
                idx[1] = idx2[1];  // (local copies of the caller's pointers)
                val[1] = val2[1];
            }
        }
    }
    {
        ProfScope ps(prof, "bucket_scatter", stream, 12.0 * (double)count);
        const uint32_t num_tiles = (uint32_t)div_up(count, kTile);
        const uint32_t grid = xcd_grid(num_tiles);
        plain_scatter_kernel<<<grid, kThreads, 0, stream>>>(idx[cur], val[cur], count, out, n_out, num_tiles);
        KERNEL_CHECK();
    }
    arena.rewind(amark);
}

void permute_packed(uint32_t *idx, uint64_t *packed, size_t count, uint32_t *out, uint32_t *out2, Arena &arena,
                    hipStream_t stream, Profiler *prof) {
    if (count == 0) return;
    int nb = 1;
    while (nb < 32 && (1ull << nb) < (uint64_t)count) ++nb;
    if (count <= (size_t(1) << 22) || nb > 2 * kRadixBits + kWindowBitsMax) {
        ProfScope ps(prof, "bucket_scatter", stream, 20.0 * (double)count);
        const unsigned g = (unsigned)std::min<size_t>(div_up(count, kThreads), 256u * 16u);
        plain_packed_scatter_kernel<<<g, kThreads, 0, stream>>>(idx, packed, count, out, out2);
        KERNEL_CHECK();
        return;
    }
    // two partition passes by the digits above the window bits, then the windows (bucketed_scatter, the two-value form)
    const size_t amark = arena.mark();
    const int wb = nb > 2 * kRadixBits + 10 ? nb - 2 * kRadixBits : 10;
    const uint32_t num_tiles = (uint32_t)div_up(count, kTile);
    uint32_t *hist = arena.alloc<uint32_t>((size_t)kBins * num_tiles);
    uint32_t *idx_b = arena.alloc<uint32_t>(count);
    uint64_t *packed_b = arena.alloc<uint64_t>(count);
    radix_pass<uint32_t, uint32_t, PairSrc, uint64_t>(PairSrc{idx, packed}, idx_b, packed_b, count, wb, hist, num_tiles,
                                                      4.0 * (double)count, 24.0 * (double)count, arena, stream, prof);
    uint16_t *idx16 = reinterpret_cast<uint16_t *>(idx);  // (the inputs are free now)
    radix_pass<uint32_t, uint16_t, PairSrc, uint64_t>(PairSrc{idx_b, packed_b}, idx16, packed, count, wb + kRadixBits, hist,
                                                      num_tiles, 4.0 * (double)count, 22.0 * (double)count, arena, stream, prof);
    {
        ProfScope ps(prof, "window_scatter", stream, 18.0 * (double)count);
        const uint32_t W = 1u << wb;
        window_scatter2_kernel<<<(unsigned)div_up(count, W), kWindow2Threads, 0, stream>>>(idx16, packed, out, out2,
                                                                                          (uint32_t)count, wb);
        KERNEL_CHECK();
    }
    arena.rewind(amark);
}

int radix_sort_pairs(uint64_t *keys[2], uint32_t *vals[2], size_t n, const int *shifts, int npasses,
                     Arena &arena, hipStream_t stream, Profiler *prof) {
    return radix_sort_impl<uint64_t>(keys, vals, n, shifts, npasses, arena, stream, prof);
}

int radix_sort_pairs(uint32_t *keys[2], uint32_t *vals[2], size_t n, const int *shifts, int npasses,
                     Arena &arena, hipStream_t stream, Profiler *prof) {
    return radix_sort_impl<uint32_t>(keys, vals, n, shifts, npasses, arena, stream, prof);
}

namespace {
// bucket starts of the partition the MSD pass has just made: the scanned table holds, for bin d
// and tile 0, the first output position of the bin
__global__ void bucket_starts_kernel(const uint32_t *__restrict__ scanned, uint32_t num_tiles, uint32_t n,
                                     uint32_t *__restrict__ bstart) {
    const uint32_t d = threadIdx.x;
    bstart[d] = scanned[(size_t)d * num_tiles];
    if (d == 0) bstart[kBins] = n;
}

// one descriptor per tile of the bucketed view (radix_sort.hpp)
__global__ __launch_bounds__(kThreads) void seg_desc_kernel(const uint32_t *__restrict__ bstart,
                                                            const uint32_t *__restrict__ tile0,
                                                            const uint32_t *__restrict__ prev_ne,
                                                            const uint32_t *__restrict__ next_ne, uint32_t num_tiles,
                                                            uint32_t *__restrict__ desc, uint32_t num_buckets) {
    const uint32_t tile = blockIdx.x * blockDim.x + threadIdx.x;
    if (tile >= num_tiles) return;
    uint32_t lo = 0, hi = num_buckets;  // largest b with tile0[b] <= tile (the non-empty one among equals)
    while (lo + 1 < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (tile0[mid] <= tile)
            lo = mid;
        else
            hi = mid;
    }
    const uint32_t t0 = tile0[lo], local = tile - t0;
    const uint32_t first = bstart[lo] + local * (uint32_t)kTile, end = bstart[lo + 1];
    uint32_t *d = desc + (size_t)tile * kSegDescWords;
    d[0] = first;
    d[1] = end - first < (uint32_t)kTile ? end - first : (uint32_t)kTile;
    d[2] = lo;
    d[3] = t0 * (uint32_t)kBins + local;
    d[4] = tile0[lo + 1] - t0;
    d[5] = bstart[lo];
    d[6] = end;
    d[7] = prev_ne[lo];
    d[8] = next_ne[lo];
    d[9] = d[10] = d[11] = 0;
}

// the same for segments given by a first and an end element each (the large sub-buckets of local_sort_kernel)
__global__ __launch_bounds__(kThreads) void seg_desc_list_kernel(const uint32_t *__restrict__ first_of,
                                                                 const uint32_t *__restrict__ end_of,
                                                                 const uint32_t *__restrict__ tile0, uint32_t num_tiles,
                                                                 uint32_t *__restrict__ desc, uint32_t num_segs) {
    const uint32_t tile = blockIdx.x * blockDim.x + threadIdx.x;
    if (tile >= num_tiles) return;
    uint32_t lo = 0, hi = num_segs;  // the segment of the tile (segments of the list are never empty)
    while (lo + 1 < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (tile0[mid] <= tile)
            lo = mid;
        else
            hi = mid;
    }
    const uint32_t t0 = tile0[lo], local = tile - t0;
    const uint32_t first = first_of[lo] + local * (uint32_t)kTile, end = end_of[lo];
    uint32_t *d = desc + (size_t)tile * kSegDescWords;
    d[0] = first;
    d[1] = end - first < (uint32_t)kTile ? end - first : (uint32_t)kTile;
    d[2] = lo;
    d[3] = t0 * (uint32_t)kBins + local;
    d[4] = tile0[lo + 1] - t0;
    d[5] = first_of[lo];
    d[6] = end;
    d[7] = lo ? lo - 1 : 0xffffffffu;
    d[8] = lo + 1 < num_segs ? lo + 1 : 0xffffffffu;
    d[9] = d[10] = d[11] = 0;
}

// The scanned table of a pass over such a list counts from the first segment of the LIST: every entry of segment k
// is moved by shift[k] = (first element of the segment) - (elements of the list in front of it).  The entries of
// segment k are [tile0[k] * 256, tile0[k + 1] * 256): one workgroup per 256 of them, all in one segment.
// (one workgroup per SEGMENT walked 16.7 M entries alone when a text is one run of A's: +19 ms)
__global__ __launch_bounds__(kBins) void seg_table_shift_kernel(uint32_t *__restrict__ table, const uint32_t *__restrict__ tile0,
                                                               const uint32_t *__restrict__ shift, uint32_t num_segs) {
    const uint32_t q = blockIdx.x;  // < tile0[num_segs]
    uint32_t lo = 0, hi = num_segs;  // the segment with tile0[lo] <= q < tile0[lo + 1] (segments of the list are never empty)
    while (lo + 1 < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (tile0[mid] <= q)
            lo = mid;
        else
            hi = mid;
    }
    table[(size_t)q * kBins + threadIdx.x] += shift[lo];
}

// the elements of the tiles of a SegView copied from one pair of arrays to another
__global__ __launch_bounds__(kThreads) void seg_copy_kernel(const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                            uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                                                            SegView seg) {
    const TileExtent ext = tile_extent(blockIdx.x, 0, seg.num_tiles, seg);
    for (uint32_t p = threadIdx.x; p < ext.count; p += kThreads) {
        keys_out[ext.first + p] = keys_in[ext.first + p];
        vals_out[ext.first + p] = vals_in[ext.first + p];
    }
}

}  // namespace

// the sub-buckets of two most-significant-digit passes sorted in LDS: local_sort_kernel, local_sort_sub_buckets
#define NOLZSS_RADIX_SORT_HIP
#include "local_sort.hpp"

void radix_sort_dna_keys(const PackedText &text, uint32_t *keys32[2], uint32_t *vals[2], uint32_t *seg_mem,
                         SegView &seg_out, Arena &arena, hipStream_t stream, Profiler *prof) {
    const size_t n = text.n;
    if (text.bits != 2) throw HipError("radix_sort_dna_keys: 2-bit texts only");
    const size_t m = arena.mark();
    const uint32_t tiles0 = (uint32_t)div_up(n, kTile);
    uint32_t *hist = arena.alloc<uint32_t>((size_t)kBins * ((size_t)tiles0 + kBins));
    uint32_t *tabs = arena.alloc<uint32_t>(4 * 257);
    uint32_t *bstart = tabs, *tile0 = tabs + 257, *prev_ne = tabs + 2 * 257, *next_ne = tabs + 3 * 257;
    const double text_bytes = (double)n * 2 / 8.0;
    // most significant digit first: key bits 32..39 = the first four bases
    // (the plain key layout for segmented texts too; their histogram digits need the masked key)
    radix_pass<uint64_t, uint32_t>(TextSrc<2>{text.words, text.terms, false, !text.segmented}, keys32[1], vals[1], n, 32, hist, tiles0,
                                   text_bytes, text_bytes + 8.0 * (double)n, arena, stream, prof);
    bucket_starts_kernel<<<1, kBins, 0, stream>>>(hist, tiles0, (uint32_t)n, bstart);
    KERNEL_CHECK();
    uint32_t h_start[kBins + 1], h_tab[3 * 257];
    HIP_CHECK(hipMemcpyAsync(h_start, bstart, sizeof(h_start), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    uint32_t *h_tile0 = h_tab, *h_prev = h_tab + 257, *h_next = h_tab + 2 * 257;
    h_tile0[0] = 0;
    for (int b = 0; b < kBins; ++b) h_tile0[b + 1] = h_tile0[b] + (uint32_t)div_up((size_t)(h_start[b + 1] - h_start[b]), kTile);
    uint32_t last = 0xffffffffu;
    for (int b = 0; b < kBins; ++b) {
        h_prev[b] = last;
        if (h_start[b + 1] > h_start[b]) last = (uint32_t)b;
    }
    last = 0xffffffffu;
    for (int b = kBins - 1; b >= 0; --b) {
        h_next[b] = last;
        if (h_start[b + 1] > h_start[b]) last = (uint32_t)b;
    }
    h_prev[256] = h_next[256] = 0;
    HIP_CHECK(hipMemcpyAsync(tile0, h_tab, sizeof(h_tab), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipStreamSynchronize(stream));  // h_tab is a local array
    seg_out.num_tiles = h_tile0[kBins];
    seg_desc_kernel<<<(unsigned)div_up(seg_out.num_tiles, kThreads), kThreads, 0, stream>>>(bstart, tile0, prev_ne, next_ne,
                                                                                       seg_out.num_tiles, seg_mem, (uint32_t)kBins);
    KERNEL_CHECK();
    seg_out.desc = seg_mem;
    // every bucket by the low 32 key bits, least significant digit first
    int cur = 1;
    for (int p = 0; p < 4; ++p) {
        radix_pass<uint32_t, uint32_t>(ArraySrc<uint32_t>{keys32[cur], vals[cur]}, keys32[cur ^ 1], vals[cur ^ 1], n,
                                       8 * p, hist, seg_out.num_tiles, 4.0 * (double)n, 16.0 * (double)n, arena,
                                       stream, prof, seg_out);
        cur ^= 1;
    }
    arena.rewind(m);  // (cur == 1 again)
}



bool key16_applicable(const PackedText &text) {
    if (text.bits != 2 || text.terms.seq_shift != 0) return false;
    if (!text.segmented) return text.terms.count == 1 && text.n >= 32;
    // a short terminator table whose segments all hold at least 16 symbols (the text may end with a terminator)
    const uint32_t cnt = text.terms.nfew;
    if (cnt < 2 || cnt > kTermFew || text.n < 64) return false;
    uint32_t start = 0;
    for (uint32_t k = 0; k < cnt; ++k) {
        const uint32_t p = text.terms.few[k];
        const bool last = k + 1 == cnt;
        if (p < start) return false;
        const uint32_t len = p - start;
        if (!(len >= 16 || (last && len == 0))) return false;
        start = p + 1;
    }
    return text.terms.few[cnt - 1] == text.n;
}

namespace {
Text16SegSrc make_text16_seg(const PackedText &text) {
    Text16SegSrc src{};
    src.words = text.words;
    src.n = text.n;
    const uint32_t cnt = text.terms.nfew;
    src.treal = cnt - 1;
    for (uint32_t k = 0; k < kTermFew; ++k) src.pos[k] = k < cnt ? text.terms.few[k] : text.n;
    // (the text ends with a terminator: the last segment is empty and the end of the text has no suffixes of its own)
    src.end_shorts = text.terms.few[cnt - 2] + 1 == text.n ? 0u : 15u;
    src.nshort = 16u * src.treal + src.end_shorts;
    return src;
}
}  // namespace

void radix_sort_dna_keys16(const PackedText &text, uint32_t *keys32[2], uint32_t *vals[2], uint32_t *seg_mem,
                           SegView &seg_out, Arena &arena, hipStream_t stream, Profiler *prof, Round0Regroup *regroup) {
    if (regroup) regroup->done = false;
    const size_t n = text.n;
    if (!key16_applicable(text)) throw HipError("radix_sort_dna_keys16: plain 2-bit texts, or segmented ones with a short terminator table");
    const size_t m = arena.mark();
    const uint32_t tiles0 = (uint32_t)div_up(n, kTile);
    uint32_t *hist = arena.alloc<uint32_t>((size_t)kBins * ((size_t)tiles0 + kBins));
    uint32_t *tabs = arena.alloc<uint32_t>(4 * 257);
    uint32_t *bstart = tabs, *tile0 = tabs + 257, *prev_ne = tabs + 2 * 257, *next_ne = tabs + 3 * 257;
    const double text_bytes = (double)n * 2 / 8.0;
    // Two ways from here (both end in keys32[0] / vals[0]): three bucket-segmented passes, or ONE and the sub-buckets it
    // makes sorted in LDS (local_sort_kernel) -- for texts whose 65 536 sub-buckets are large enough to pay for a
    // workgroup each and small enough to fit one (NOLZSS_NO_LOCAL_SORT, NOLZSS_LOCAL_SORT_MIN = smallest such text).
    static const bool no_local = getenv("NOLZSS_NO_LOCAL_SORT") != nullptr;
    static const size_t local_min = getenv("NOLZSS_LOCAL_SORT_MIN") ? (size_t)atoll(getenv("NOLZSS_LOCAL_SORT_MIN")) : (size_t(1) << 28);
    const bool local = !no_local && !local_sort_off.load() && n >= local_min && n <= (size_t)kBins * kBins * kLocalCap / 16 * 15;
    const int msd_to = local ? 0 : 1;
    // most significant digit first: the first four bases (bits 32..39 of [32 key bits][8-bit tag])
    if (text.segmented)
        radix_pass<uint64_t, uint32_t>(make_text16_seg(text), keys32[msd_to], vals[msd_to], n, 32, hist, tiles0, text_bytes,
                                       text_bytes + 8.0 * (double)n, arena, stream, prof);
    else
        radix_pass<uint64_t, uint32_t>(Text16Src{text.words, (uint32_t)n}, keys32[msd_to], vals[msd_to], n, 32, hist, tiles0, text_bytes,
                                       text_bytes + 8.0 * (double)n, arena, stream, prof);
    bucket_starts_kernel<<<1, kBins, 0, stream>>>(hist, tiles0, (uint32_t)n, bstart);
    KERNEL_CHECK();
    uint32_t h_start[kBins + 1], h_tab[3 * 257];
    HIP_CHECK(hipMemcpyAsync(h_start, bstart, sizeof(h_start), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    uint32_t *h_tile0 = h_tab, *h_prev = h_tab + 257, *h_next = h_tab + 2 * 257;
    h_tile0[0] = 0;
    for (int b = 0; b < kBins; ++b) h_tile0[b + 1] = h_tile0[b] + (uint32_t)div_up((size_t)(h_start[b + 1] - h_start[b]), kTile);
    uint32_t last = 0xffffffffu;
    for (int b = 0; b < kBins; ++b) {
        h_prev[b] = last;
        if (h_start[b + 1] > h_start[b]) last = (uint32_t)b;
    }
    last = 0xffffffffu;
    for (int b = kBins - 1; b >= 0; --b) {
        h_next[b] = last;
        if (h_start[b + 1] > h_start[b]) last = (uint32_t)b;
    }
    h_prev[256] = h_next[256] = 0;
    HIP_CHECK(hipMemcpyAsync(tile0, h_tab, sizeof(h_tab), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipStreamSynchronize(stream));  // h_tab is a local array
    seg_out.num_tiles = h_tile0[kBins];
    seg_desc_kernel<<<(unsigned)div_up(seg_out.num_tiles, kThreads), kThreads, 0, stream>>>(bstart, tile0, prev_ne, next_ne,
                                                                                       seg_out.num_tiles, seg_mem, (uint32_t)kBins);
    KERNEL_CHECK();
    seg_out.desc = seg_mem;
    if (local) {
        // the digit below the bucket's (four more bases) first, then every sub-bucket by the 16 bits between it and the tag
        radix_pass<uint32_t, uint32_t>(ArraySrc<uint32_t>{keys32[0], vals[0]}, keys32[1], vals[1], n, kP16TagBits + 16, hist,
                                       seg_out.num_tiles, 4.0 * (double)n, 16.0 * (double)n, arena, stream, prof, seg_out);
        local_sort_sub_buckets(keys32[1], vals[1], keys32[0], vals[0], hist, tile0, bstart, (uint32_t)kBins, kP16TagBits, 2, n, arena, stream, prof, regroup);
        arena.rewind(m);
        return;
    }
    // every bucket by the 24 key bits above the tag byte, least significant digit first: THREE passes
    int cur = 1;
    for (int p = 0; p < 3; ++p) {
        radix_pass<uint32_t, uint32_t>(ArraySrc<uint32_t>{keys32[cur], vals[cur]}, keys32[cur ^ 1], vals[cur ^ 1], n,
                                       kP16TagBits + 8 * p, hist, seg_out.num_tiles, 4.0 * (double)n, 16.0 * (double)n, arena,
                                       stream, prof, seg_out);
        cur ^= 1;
    }
    arena.rewind(m);  // (cur == 0: the sorted pairs are in keys32[0] / vals[0])
}

void radix_sort_dna_keys16_fused(const PackedText &text, uint64_t *rec[2], uint32_t *sa_out, uint32_t *seg_mem,
                                 SegView &seg_out, Arena &arena, hipStream_t stream, Profiler *prof) {
    const size_t n = text.n;
    if (text.bits != 2 || text.segmented || text.terms.count != 1 || n < 32) throw HipError("radix_sort_dna_keys16_fused: plain 2-bit texts only");
    const size_t m = arena.mark();
    const uint32_t tiles0 = (uint32_t)div_up(n, kTile);
    uint32_t *hist = arena.alloc<uint32_t>((size_t)kBins * ((size_t)tiles0 + kBins));
    uint32_t *tabs = arena.alloc<uint32_t>(4 * 257);
    uint32_t *bstart = tabs, *tile0 = tabs + 257, *prev_ne = tabs + 2 * 257, *next_ne = tabs + 3 * 257;
    const double text_bytes = (double)n * 2 / 8.0;
    const Text16Src tsrc{text.words, (uint32_t)n};
    radix_pass_rec<TextRec16Src, false>(TextRec16Src{tsrc}, tsrc, 32, rec[1], nullptr, nullptr, n, 0, hist, tiles0, text_bytes,
                                        text_bytes + 8.0 * (double)n, arena, stream, prof);
    bucket_starts_kernel<<<1, kBins, 0, stream>>>(hist, tiles0, (uint32_t)n, bstart);
    KERNEL_CHECK();
    uint32_t h_start[kBins + 1], h_tab[3 * 257];
    HIP_CHECK(hipMemcpyAsync(h_start, bstart, sizeof(h_start), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    uint32_t *h_tile0 = h_tab, *h_prev = h_tab + 257, *h_next = h_tab + 2 * 257;
    h_tile0[0] = 0;
    for (int b = 0; b < kBins; ++b) h_tile0[b + 1] = h_tile0[b] + (uint32_t)div_up((size_t)(h_start[b + 1] - h_start[b]), kTile);
    uint32_t last = 0xffffffffu;
    for (int b = 0; b < kBins; ++b) {
        h_prev[b] = last;
        if (h_start[b + 1] > h_start[b]) last = (uint32_t)b;
    }
    last = 0xffffffffu;
    for (int b = kBins - 1; b >= 0; --b) {
        h_next[b] = last;
        if (h_start[b + 1] > h_start[b]) last = (uint32_t)b;
    }
    h_prev[256] = h_next[256] = 0;
    HIP_CHECK(hipMemcpyAsync(tile0, h_tab, sizeof(h_tab), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipStreamSynchronize(stream));  // h_tab is a local array
    seg_out.num_tiles = h_tile0[kBins];
    seg_desc_kernel<<<(unsigned)div_up(seg_out.num_tiles, kThreads), kThreads, 0, stream>>>(bstart, tile0, prev_ne, next_ne,
                                                                                       seg_out.num_tiles, seg_mem, (uint32_t)kBins);
    KERNEL_CHECK();
    seg_out.desc = seg_mem;
    // three segmented passes over the 24 key bits above the tag byte; the last one splits the records into the key
    // words (left in the buffer the pass does not read: rec[1], as 32-bit words) and the suffix array
    int cur = 1;
    for (int p = 0; p < 3; ++p) {
        const ArraySrc<uint64_t> hsrc{rec[cur], nullptr};
        const int shift = kP16TagBits + 8 * p;
        if (p < 2)
            radix_pass_rec<RecArraySrc, false>(RecArraySrc{rec[cur]}, hsrc, 32 + shift, rec[cur ^ 1], nullptr, nullptr, n, shift,
                                               hist, seg_out.num_tiles, 8.0 * (double)n, 16.0 * (double)n, arena, stream, prof, seg_out);
        else
            radix_pass_rec<RecArraySrc, true>(RecArraySrc{rec[cur]}, hsrc, 32 + shift, nullptr, reinterpret_cast<uint32_t *>(rec[cur ^ 1]),
                                              sa_out, n, shift, hist, seg_out.num_tiles, 8.0 * (double)n, 16.0 * (double)n, arena,
                                              stream, prof, seg_out);
        cur ^= 1;
    }
    arena.rewind(m);  // (the key words are in rec[0], as 32-bit words; the suffixes in sa_out)
}

void radix_sort_record_keys(const PackedText &text, const std::vector<uint32_t> &h_terms, uint32_t *keys32[2],
                            uint32_t *vals[2], uint32_t *seg_mem, SegView &seg_out, Arena &arena, hipStream_t stream,
                            Profiler *prof) {
    const size_t n = text.n;
    const uint32_t nb = (uint32_t)h_terms.size();  // records = buckets; h_terms[k] = terminator of record k
    if (text.bits != 2 || nb == 0 || h_terms.back() != n) throw HipError("radix_sort_record_keys: bad record table");
    const size_t m = arena.mark();
    // bucket k = the text positions of record k and of the separator behind it: already "partitioned"
    std::vector<uint32_t> tab(4 * ((size_t)nb + 1));
    uint32_t *h_start = tab.data(), *h_tile0 = h_start + nb + 1, *h_prev = h_tile0 + nb + 1, *h_next = h_prev + nb + 1;
    h_start[0] = 0;
    for (uint32_t k = 0; k + 1 < nb; ++k) h_start[k + 1] = h_terms[k] + 1;
    h_start[nb] = (uint32_t)n;
    h_tile0[0] = 0;
    for (uint32_t k = 0; k < nb; ++k) h_tile0[k + 1] = h_tile0[k] + (uint32_t)div_up((size_t)(h_start[k + 1] - h_start[k]), kTile);
    uint32_t last = 0xffffffffu;
    for (uint32_t k = 0; k < nb; ++k) {
        h_prev[k] = last;
        if (h_start[k + 1] > h_start[k]) last = k;
    }
    last = 0xffffffffu;
    for (uint32_t k = nb; k-- > 0;) {
        h_next[k] = last;
        if (h_start[k + 1] > h_start[k]) last = k;
    }
    h_prev[nb] = h_next[nb] = 0;
    uint32_t *d_tab = arena.alloc<uint32_t>(tab.size());
    HIP_CHECK(hipMemcpyAsync(d_tab, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipStreamSynchronize(stream));  // tab is a local vector
    seg_out.num_tiles = h_tile0[nb];
    seg_desc_kernel<<<(unsigned)div_up(seg_out.num_tiles, kThreads), kThreads, 0, stream>>>(
        d_tab, d_tab + (nb + 1), d_tab + 2 * ((size_t)nb + 1), d_tab + 3 * ((size_t)nb + 1), seg_out.num_tiles, seg_mem, nb);
    KERNEL_CHECK();
    seg_out.desc = seg_mem;
    uint32_t *hist = arena.alloc<uint32_t>((size_t)kBins * seg_out.num_tiles);
    const double text_bytes = (double)n * 2 / 8.0;
    // Long records (a megabase and more on average): the pass from the text takes the MOST significant digit, and the
    // 256 sub-buckets it makes of every record -- 16 Ki pairs of a 4-megabase record -- are sorted by the other three in
    // LDS (local_sort_kernel): one pass over HBM and one read + write instead of four passes.
    static const bool no_local = getenv("NOLZSS_NO_LOCAL_SORT") != nullptr;
    static const size_t local_min = getenv("NOLZSS_LOCAL_SORT_MIN") ? (size_t)atoll(getenv("NOLZSS_LOCAL_SORT_MIN")) : (size_t(1) << 20);
    if (!no_local && !local_sort_off.load() && n / nb >= local_min) {
        radix_pass<uint32_t, uint32_t>(RecordTextSrc{text.words, text.terms.pos}, keys32[1], vals[1], n, 24, hist,
                                       seg_out.num_tiles, text_bytes, text_bytes + 8.0 * (double)n, arena, stream, prof, seg_out);
        local_sort_sub_buckets(keys32[1], vals[1], keys32[0], vals[0], hist, d_tab + (nb + 1), d_tab, nb, 0, 3, n, arena, stream, prof);
        arena.rewind(m);
        return;
    }
    // least significant digit first inside every record; the first pass makes its pairs from the text
    radix_pass<uint32_t, uint32_t>(RecordTextSrc{text.words, text.terms.pos}, keys32[1], vals[1], n, 0, hist,
                                   seg_out.num_tiles, text_bytes, text_bytes + 8.0 * (double)n, arena, stream, prof, seg_out);
    int cur = 1;
    for (int p = 1; p < 4; ++p) {
        radix_pass<uint32_t, uint32_t>(ArraySrc<uint32_t>{keys32[cur], vals[cur]}, keys32[cur ^ 1], vals[cur ^ 1], n,
                                       8 * p, hist, seg_out.num_tiles, 4.0 * (double)n, 16.0 * (double)n, arena,
                                       stream, prof, seg_out);
        cur ^= 1;
    }
    arena.rewind(m);  // (cur == 0: the sorted pairs are in keys32[0] / vals[0])
}

int radix_sort_segments_u32(uint32_t *keys[2], uint32_t *vals[2], size_t n, const std::vector<uint32_t> &h_start, int npasses,
                            Arena &arena, hipStream_t stream, Profiler *prof) {
    const uint32_t nb = (uint32_t)h_start.size() - 1;  // segments [h_start[k], h_start[k + 1])
    if (n == 0 || npasses == 0) return 0;
    if (nb == 0 || h_start[0] != 0 || h_start[nb] != n) throw HipError("radix_sort_segments_u32: bad segment table");
    const size_t m = arena.mark();
    std::vector<uint32_t> tab(4 * ((size_t)nb + 1));
    uint32_t *t_start = tab.data(), *t_tile0 = t_start + nb + 1, *t_prev = t_tile0 + nb + 1, *t_next = t_prev + nb + 1;
    for (uint32_t k = 0; k <= nb; ++k) t_start[k] = h_start[k];
    t_tile0[0] = 0;
    for (uint32_t k = 0; k < nb; ++k) t_tile0[k + 1] = t_tile0[k] + (uint32_t)div_up((size_t)(t_start[k + 1] - t_start[k]), kTile);
    uint32_t last = 0xffffffffu;
    for (uint32_t k = 0; k < nb; ++k) {
        t_prev[k] = last;
        if (t_start[k + 1] > t_start[k]) last = k;
    }
    last = 0xffffffffu;
    for (uint32_t k = nb; k-- > 0;) {
        t_next[k] = last;
        if (t_start[k + 1] > t_start[k]) last = k;
    }
    t_prev[nb] = t_next[nb] = 0;
    uint32_t *d_tab = arena.alloc<uint32_t>(tab.size());
    HIP_CHECK(hipMemcpyAsync(d_tab, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipStreamSynchronize(stream));  // tab is a local vector
    SegView seg;
    seg.num_tiles = t_tile0[nb];
    uint32_t *seg_mem = arena.alloc<uint32_t>((size_t)kSegDescWords * seg.num_tiles + 4);
    seg_desc_kernel<<<(unsigned)div_up(seg.num_tiles, kThreads), kThreads, 0, stream>>>(
        d_tab, d_tab + (nb + 1), d_tab + 2 * ((size_t)nb + 1), d_tab + 3 * ((size_t)nb + 1), seg.num_tiles, seg_mem, nb);
    KERNEL_CHECK();
    seg.desc = seg_mem;
    uint32_t *hist = arena.alloc<uint32_t>((size_t)kBins * seg.num_tiles);
    int cur = 0;
    for (int p = 0; p < npasses; ++p) {
        radix_pass<uint32_t, uint32_t>(ArraySrc<uint32_t>{keys[cur], vals[cur]}, keys[cur ^ 1], vals[cur ^ 1], n, 8 * p, hist,
                                       seg.num_tiles, 4.0 * (double)n, 16.0 * (double)n, arena, stream, prof, seg);
        cur ^= 1;
    }
    arena.rewind(m);
    return cur;
}

int radix_sort_initial_keys(const PackedText &text, uint64_t *keys[2], uint32_t *vals[2], const int *shifts,
                            int npasses, Arena &arena, hipStream_t stream, Profiler *prof) {
    const size_t n = text.n;
    if (n == 0 || npasses == 0) return 0;
    {
        // first pass: pairs computed from the packed text, written to buffer 1.  Algorithmic bytes:
        // the text window once per kernel (bits / 8 per symbol) and, for the scatter, the sorted
        // pairs written once.
        const size_t m = arena.mark();
        const uint32_t num_tiles = (uint32_t)div_up(n, kTile);
        uint32_t *hist = arena.alloc<uint32_t>((size_t)kBins * num_tiles);
        const double text_bytes = (double)n * text.bits / 8.0;
        const double out_bytes = text_bytes + 12.0 * (double)n;
        switch (text.bits) {
        case 2:
            radix_pass<uint64_t, uint64_t>(TextSrc<2>{text.words, text.terms, text.segmented}, keys[1], vals[1], n, shifts[0], hist,
                                 num_tiles, text_bytes, out_bytes, arena, stream, prof);
            break;
        case 4:
            radix_pass<uint64_t, uint64_t>(TextSrc<4>{text.words, text.terms, text.segmented}, keys[1], vals[1], n, shifts[0], hist,
                                 num_tiles, text_bytes, out_bytes, arena, stream, prof);
            break;
        default:
            radix_pass<uint64_t, uint64_t>(TextSrc<8>{text.words, text.terms, text.segmented}, keys[1], vals[1], n, shifts[0], hist,
                                 num_tiles, text_bytes, out_bytes, arena, stream, prof);
            break;
        }
        arena.rewind(m);
    }
    return radix_sort_impl<uint64_t>(keys, vals, n, shifts, npasses, arena, stream, prof, 1);
}

}  // namespace nolzss
