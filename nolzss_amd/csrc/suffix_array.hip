// suffix_array.hip -- text packing and suffix-array construction on gfx950.
//
// Replaces the SA/CSA part of sdsl::construct_im(cst, text, 1) that the reference calls at
// /root/reference/src/cpp/factorizer.cpp:340,381 and factorizer_core.hpp:208.
//
// Method: a key sort, ONE direct-comparison round, then prefix doubling for what is left.
//   round 0   key(i) = first K symbols of suffix i (bit-packed, K = 17 for 2-bit DNA) plus a
//             length tag; the keys are computed inside the first radix pass (radix_sort.hip: one
//             most-significant-digit pass + four bucket-segmented passes on 8-byte records for
//             plain DNA, 5-8 passes on 12-byte records otherwise); the last pass lands in SA.
//   regroup   one single-pass kernel per round (decoupled look-back): group heads, ranks, LCP
//             of every boundary that appeared, compacted list of the suffixes still tied.
//   direct    every group of <= 64 suffixes is finished by comparing the packed suffixes
//             themselves, 512 bits per step, pair by pair through LDS (group_refine_kernel).
//   round h   only suffixes whose group is not yet a singleton stay active; key = (group head
//             rank, rank[i + h]); small groups sorted by counting, large ones by radix sort.
// All arrays are 32-bit; rank[i] holds (index of the first slot of i's group) + 1, and 0 means
// "past the end of the text", which sorts before every real suffix exactly as the reference's
// appended terminator does.
#include "lookback.hpp"
#include "pipeline.hpp"
#include "pyramid.hpp"
#include "queues.hpp"
#include "radix_sort.hpp"
#include "scan.hpp"

#include <algorithm>
#include <cstdlib>

namespace nolzss {

void Context::read_back(const uint32_t *d_src, uint32_t *dst, int count) {
    HIP_CHECK(hipMemcpyAsync(h_pinned, d_src, sizeof(uint32_t) * count, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    for (int k = 0; k < count; ++k) dst[k] = h_pinned[k];
}

namespace {

constexpr int kThreads = 256;

struct OpMinU32x {
    __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a < b ? a : b; }
};

inline unsigned grid_for(size_t work_items, int per_block, unsigned cap = 256u * 16u) {
    size_t g = div_up(work_items, (size_t)per_block);
    if (g < 1) g = 1;
    return (unsigned)(g > cap ? cap : g);
}

// ---------------------------------------------------------------------------------------
// alphabet presence: which byte values occur (256-bit mask, OR-reduced per wavefront)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void mark_byte(uint64_t (&m)[4], uint32_t b) {
    const uint64_t bit = 1ull << (b & 63);
    const uint32_t q = b >> 6;
    m[0] |= (q == 0) ? bit : 0;
    m[1] |= (q == 1) ? bit : 0;
    m[2] |= (q == 2) ? bit : 0;
    m[3] |= (q == 3) ? bit : 0;
}

__global__ __launch_bounds__(kThreads) void presence_kernel(const uint8_t *__restrict__ text, size_t n,
                                                            unsigned long long *presence) {
    uint64_t m[4] = {0, 0, 0, 0};
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    if (((uintptr_t)text & 15) == 0) {
        const uint4 *v = reinterpret_cast<const uint4 *>(text);
        const size_t nv = n / 16;
        // four loads in flight per thread (a piece past the end reads the last piece again: marking a byte
        // twice changes nothing)
        for (size_t i = tid; i < nv; i += 4 * stride) {
            uint4 x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t q = i + (size_t)u * stride;
                x[u] = v[q < nv ? q : nv - 1];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t wds[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) mark_byte(m, (wds[k] >> (8 * e)) & 255u);
            }
        }
        for (size_t i = nv * 16 + tid; i < n; i += stride) mark_byte(m, text[i]);
    } else {
        for (size_t i = tid; i < n; i += stride) mark_byte(m, text[i]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint64_t v = m[k];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v |= __shfl_xor(v, d, 64);
        if (lane_id() == 0 && v) atomicOr(&presence[k], (unsigned long long)v);
    }
}

// positions of everything that is not an upper-case nucleotide (at most kMaxTermScan are
// recorded; the count keeps running)
constexpr uint32_t kMaxTermScan = 512;

// kCountOnly: no positions, and ONE atomic per wavefront at the end.  pack_text asks for the count first: a text over
// another alphabet that happens to contain A, C, G and T -- a protein -- has 10^8 bytes that are "not a nucleotide", and
// one returning atomic per such byte on a single counter took 47 ms of the 80 ms of a 2^28-symbol protein text (round 4,
// tools/alphabet_probe.py); the positions are recorded by a second launch only when there are at most 250 of them.
template <bool kCountOnly>
__global__ __launch_bounds__(kThreads) void find_terminators_kernel(const uint8_t *__restrict__ text, uint32_t n,
                                                                    uint32_t *__restrict__ count,
                                                                    uint32_t *__restrict__ pos_out) {
    uint32_t local = 0;
    auto check = [&](uint8_t c, size_t i) {
        if (c != 'A' && c != 'C' && c != 'G' && c != 'T') {
            if (kCountOnly) {
                ++local;
            } else {
                const uint32_t k = atomicAdd(count, 1u);
                if (k < kMaxTermScan) pos_out[k] = (uint32_t)i;
            }
        }
    };
    // 16 bytes per load from the first 16-byte boundary on; a 32-bit word is tested against the four
    // nucleotides at once with exact per-byte equality masks, and only a word that holds something else is
    // looked at byte by byte (1.35 -> 0.35 ms per 2^30-base run of the merged batch)
    const size_t head = (size_t)((16 - (reinterpret_cast<uintptr_t>(text) & 15)) & 15);
    const size_t h = head < n ? head : n;
    const size_t vecs = (n - h) / 16;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < h) check(text[tid], tid);
    const uint4 *v = reinterpret_cast<const uint4 *>(text + h);
    auto all_nucleotides = [](uint32_t w) -> bool {
        // per byte: zero iff the byte equals the pattern; a byte of (x ^ p) is zero <=> haszero
        auto eq = [](uint32_t x, uint32_t p) -> uint32_t {
            const uint32_t y = x ^ p;
            return ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y | 0x7f7f7f7fu);  // 0x80 in every byte that matched
        };
        const uint32_t m = eq(w, 0x41414141u) | eq(w, 0x43434343u) | eq(w, 0x47474747u) | eq(w, 0x54545454u);
        return m == 0x80808080u;
    };
    for (size_t k = tid; k < vecs; k += stride) {
        const uint4 q = v[k];
        if (all_nucleotides(q.x) && all_nucleotides(q.y) && all_nucleotides(q.z) && all_nucleotides(q.w)) continue;
        const size_t base = h + k * 16;
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 16; ++j) check((uint8_t)(w[j >> 2] >> (8 * (j & 3))), base + (size_t)j);
    }
    const size_t tail0 = h + vecs * 16;
    if (tail0 + tid < n) check(text[tail0 + tid], tail0 + tid);
    if (kCountOnly) {
        // (saturating: the caller only asks whether the count is one of at most 250, and 2^15 wavefronts x 1024 fits 32 bits)
        local = local < 1024u ? local : 1024u;
        const uint32_t total = wave_reduce(local, OpAdd<uint32_t>());
        if (lane_id() == 0 && total) atomicAdd(count, total < 1024u ? total : 1024u);
    }
}

// ---------------------------------------------------------------------------------------
// packing: one thread per 64-bit output word
// ---------------------------------------------------------------------------------------
template <int BITS>
__global__ __launch_bounds__(kThreads) void pack_kernel(const uint8_t *__restrict__ text, size_t n,
                                                        const unsigned long long *__restrict__ presence,
                                                        uint64_t *__restrict__ words, size_t nwords) {
    constexpr int kSyms = 64 / BITS;
    __shared__ uint8_t lut[256];
    {
        const int b = threadIdx.x;  // kThreads == 256
        int c = 0;
        for (int k = 0; k < (b >> 6); ++k) c += __popcll(presence[k]);
        c += __popcll(presence[b >> 6] & ((1ull << (b & 63)) - 1ull));
        // bytes outside the alphabet (the unique terminators of a segmented text, which may lie above
        // 'T') pack as code 0: their rank would not fit the symbol width and spill into the
        // neighbouring base
        lut[b] = ((presence[b >> 6] >> (b & 63)) & 1ull) ? (uint8_t)c : (uint8_t)0;
    }
    __syncthreads();
    const bool aligned = ((uintptr_t)text & 15) == 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t wi = (size_t)blockIdx.x * blockDim.x + threadIdx.x; wi < nwords; wi += stride) {
        const size_t base = wi * kSyms;
        uint64_t acc = 0;
        if (aligned && base + kSyms <= n) {
            if constexpr (kSyms == 8) {
                const uint2 x = *reinterpret_cast<const uint2 *>(text + base);
                const uint32_t wds[2] = {x.x, x.y};
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = (acc << BITS) | lut[(wds[k] >> (8 * e)) & 255u];
            } else {
#pragma unroll
                for (int c = 0; c < kSyms / 16; ++c) {
                    const uint4 x = *reinterpret_cast<const uint4 *>(text + base + 16 * c);
                    const uint32_t wds[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc = (acc << BITS) | lut[(wds[k] >> (8 * e)) & 255u];
                }
            }
        } else {
#pragma unroll 4
            for (int e = 0; e < kSyms; ++e) {
                const size_t p = base + e;
                acc = (acc << BITS) | (p < n ? (uint64_t)lut[text[p]] : 0ull);
            }
        }
        words[wi] = acc;
    }
}

// ---------------------------------------------------------------------------------------
// regrouping after a sort
// ---------------------------------------------------------------------------------------
// The sorted view of the m active elements is either the 64-bit round-0 keys (kRound0) or, in
// the doubling rounds, the pair (grp[a], lo[a]) = (slot of the element's current group head,
// rank of the suffix h symbols further on).  Element a starts a new group iff its view differs
// from element a-1.
template <bool kRound0>
__device__ __forceinline__ bool is_head(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ grp,
                                        const uint32_t *__restrict__ lo, size_t a) {
    if (a == 0) return true;
    if (kRound0) return keys[a] != keys[a - 1];
    return grp[a] != grp[a - 1] || lo[a] != lo[a - 1];
}

// LCP code while the suffix array is being built: the boundary has not appeared yet (values
// >= kLcpPendingMin act as +infinity in range minima).
constexpr uint32_t kLcpPending = 0xffffffffu;
constexpr uint32_t kLcpPendingMin = kLcpPending - 64u;
constexpr uint32_t kLcpPendingCompared = kLcpPending - 1u;  // pending, inside a class the direct round has compared

// ---- single-pass regroup --------------------------------------------------------------------
// One kernel does what used to be five passes (mark heads, max-scan, commit, add-scan, compact):
// every workgroup takes the next tile of the sorted view (ticket order), finds the group heads,
// and obtains the two running values it needs from the tiles in front of it -- the slot of the
// last group head (a max-scan) and the number of elements that stay active (an add-scan) -- by
// decoupled look-back over per-tile descriptors in HBM: [status : value] in one 64-bit word,
// status 1 = the tile's own aggregate, 2 = inclusive prefix.  Tickets are handed out in start
// order, so a workgroup only ever waits for workgroups that are already running.
// It then writes the new order (sa), the rank of every element (slot of its group head + 1), the
// LCP of every boundary that became known, and the compacted active list for the next round.
// HBM traffic at round 0: 12 B read + 12 B written per suffix plus 8 B per surviving element,
// where the five passes moved ~88 B.
constexpr int kFuseThreads = 512;  // 8 items per thread keep the registers low: 4 workgroups = 32 waves per CU
constexpr int kFuseItems = 8;
constexpr int kFuseTile = kFuseThreads * kFuseItems;
// Both running values in ONE descriptor, [status:2 | last head slot:31 | kept:31], for lists shorter
// than 2^31: one walk over the tiles in front instead of two.  (The walk is what a regroup tile
// waits for -- with ~1800 small tiles in flight, most of them published but not yet finished, it
// goes back through dozens of 64-descriptor windows, each a device-scope round trip.)
struct MaxSum {
    uint32_t mx, sum;
};
__device__ __forceinline__ uint64_t pack_desc(uint32_t status, MaxSum v) {
    return ((uint64_t)status << 62) | ((uint64_t)v.mx << 31) | (uint64_t)v.sum;
}
#ifndef NOLZSS_LOOKBACK_WINDOWS
#define NOLZSS_LOOKBACK_WINDOWS 1
#endif
__device__ __forceinline__ MaxSum lookback_exclusive_packed(uint64_t *desc, uint32_t tile, MaxSum aggregate,
                                                            uint32_t *err) {
    // kWin windows of 64 descriptors are loaded per round trip and evaluated nearest first.  (Measured with
    // NOLZSS_REGROUP_PHASES at 2^30: a tile spends 27 k cycles on loads and heads, 15 k in this walk, 6 k on
    // its output; four windows per round trip did not shorten the walk -- it waits for the slowest of the
    // tiles in front to publish, not for the number of descriptors -- so one window stays the default.)
    constexpr int kWin = NOLZSS_LOOKBACK_WINDOWS;
    const int lane = lane_id();
    MaxSum excl{0u, 0u};
    if (tile == 0) {
        if (lane == 0) desc_store(desc, pack_desc(2u, aggregate));
        return excl;
    }
    if (lane == 0) desc_store(desc + tile, pack_desc(1u, aggregate));
    int64_t look = (int64_t)tile - 1;
    uint32_t spins = 0;
    for (;;) {
        uint64_t d[kWin];
#pragma unroll
        for (int j = 0; j < kWin; ++j) {
            const int64_t idx = look - 64 * j - lane;
            d[j] = idx >= 0 ? desc_load(desc + idx) : (2ull << 62);  // in front of tile 0: inclusive identity
        }
        bool done = false, stalled = false;
#pragma unroll
        for (int j = 0; j < kWin; ++j) {
            if (done || stalled) continue;  // (wave-uniform)
            const uint32_t st = (uint32_t)(d[j] >> 62);
            const uint64_t inc = __ballot(st == 2);
            // every lane up to and including the first inclusive one must have been published
            const uint64_t need = inc ? (((inc & (~inc + 1ull)) << 1) - 1ull) : ~0ull;
            const uint64_t missing = __ballot(st == 0) & need;
            if (missing) {  // not published yet: wait and read again from this window on
                stalled = true;
                continue;
            }
            const bool use = (need >> lane) & 1ull;
            const uint32_t vm = use ? (uint32_t)(d[j] >> 31) & 0x7fffffffu : 0u;
            const uint32_t vs = use ? (uint32_t)d[j] & 0x7fffffffu : 0u;
            const uint32_t wm = wave_reduce(vm, OpMax<uint32_t>());
            excl.mx = wm > excl.mx ? wm : excl.mx;
            excl.sum += wave_reduce(vs, OpAdd<uint32_t>());
            look -= 64;
            if (inc) done = true;  // an inclusive prefix was reached
        }
        if (done) break;
        if (stalled) {
            if (++spins > kSpinLimit) {  // cannot happen with ticket order; never hang the GPU
                if (lane == 0) atomicExch(err, 1u);
                return excl;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    if (lane == 0) {
        MaxSum incl{excl.mx > aggregate.mx ? excl.mx : aggregate.mx, excl.sum + aggregate.sum};
        desc_store(desc + tile, pack_desc(2u, incl));
    }
    return excl;
}

struct RegroupArgs {
    const uint64_t *keys;     // round 0: sorted keys ...
    const uint32_t *keys32;   // ... or their low halves, the top byte implied by the bucket (seg)
    SegView seg;
    uint32_t num_tiles;
    uint32_t short_tag;       // round 0: elements whose length tag is below this are groups of their own
    uint32_t seq_shift;       // round 0, independent sequences: key bits from here up = number of the sequence
    int sa_is_current;        // the producer has already written the new order into sa (direct round)
    const uint32_t *grp;      // later rounds: (group head slot, secondary key) per list element
    const uint32_t *lo;
    const uint32_t *vals;     // suffix start per element
    const uint32_t *act_slot; // later rounds: slot per list element
    uint32_t m;
    uint32_t *sa;
    uint32_t *rank_val;       // (when rank_by_slot == nullptr) new rank of the elements whose rank changes ...
    uint32_t *chg_idx;        // ... and their suffix starts, appended in any order; chg_count counts them
    uint32_t *chg_count;
    uint32_t *rank_by_slot;   // non-null: the rounds before rank[] exists (no list of changed ranks is kept)
    int store_ranks;          // ... and the rank of every slot is stored there
    uint32_t *lcp;
    int sym_bits, tag_bits, bits, low_bits;  // round 0 key layout
    int bits_shift;                          // log2(bits): a division by a run-time value costs ~20 instructions per item
    const uint32_t *lcp_list; // later rounds: LCP decided by the direct comparison round
    uint32_t dbl_h;
    Pyramid Plcp;
    uint32_t *new_slot, *new_grp;  // compacted active list of the next round
    uint64_t *desc_max, *desc_sum;
    int packed;               // both scans share the descriptors in desc_max (n < 2^31)
    uint32_t *ticket;         // [0] tile tickets, [1] error flag
    uint32_t *d_total;        // number of elements that stay active
    unsigned long long *phases;  // (diagnostics, NOLZSS_REGROUP_PHASES) cycles per phase, summed over sampled tiles
};

// value of the previous / next lane of the wavefront (lane 0 / lane 63 keep `edge`)
__device__ __forceinline__ uint32_t lane_prev(uint32_t v, uint32_t edge) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x138, 0xf, 0xf, false);  // wave_shr:1
}
__device__ __forceinline__ uint32_t lane_next(uint32_t v, uint32_t edge) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x130, 0xf, 0xf, false);  // wave_shl:1
}

// kLayout (round 0 of the bucketed 2-bit key sorts): the key layout is known at compile time, which folds the
// shifts and masks of every item (the kernel is bound by VALU issue: ~1300 instructions per wavefront and 512
// suffixes).  1 = plain DNA: 34 symbol bits, 6-bit tag, no low bits, no sequence numbers, short suffixes flagged;
// 2 = long independent records, bucket = record: 28 symbol bits, 4-bit tag, the record number above bit 32;
// 3 = plain DNA with the 16-base key: 32 symbol bits (bucket + 24 stored bits), the tag in the low byte of the stored word.
template <bool kRound0, int kLayout>
__global__ __launch_bounds__(kFuseThreads) void regroup_kernel(RegroupArgs A) {
    constexpr bool kDnaFast = kLayout != 0;  // (bucketed, compile-time layout)
    const int low_bits = kDnaFast ? 0 : A.low_bits;
    const int tag_bits = kLayout == 1 ? KeyLayout<2>::kTagBits : (kLayout == 2 ? kRecTagBits : (kLayout == 3 ? kP16TagBits : A.tag_bits));
    const int sym_bits = kLayout == 1 ? 2 * KeyLayout<2>::kSyms : (kLayout == 2 ? 2 * kRecSyms : (kLayout == 3 ? 2 * kP16Syms : A.sym_bits));
    const int bits_shift = kDnaFast ? 1 : A.bits_shift;
    const uint32_t short_tag = kLayout == 1 ? (uint32_t)KeyLayout<2>::kSyms : (kLayout == 2 ? 0u : (kLayout == 3 ? (uint32_t)kP16Syms : A.short_tag));
    const uint32_t seq_shift = (kLayout == 1 || kLayout == 3) ? 0u : (kLayout == 2 ? 32u : A.seq_shift);
    constexpr int kWaves = kFuseThreads / 64;
    constexpr int kSegs = kFuseItems * kWaves;  // 64-element segments of the tile, in element order
    __shared__ uint32_t s_tile;
    __shared__ uint32_t s_seg_max[kSegs], s_seg_sum[kSegs];  // per segment: last head slot, kept; then prefixes
    __shared__ uint32_t s_excl[2];
    const bool timed = A.phases != nullptr && (blockIdx.x & 15) == 0 && threadIdx.x == 0;
    unsigned long long clk[5] = {0, 0, 0, 0, 0};
    if (timed) clk[0] = __builtin_readcyclecounter();
    if (threadIdx.x == 0) s_tile = atomicAdd(A.ticket, 1u);  // (blockIdx order measured 5 % faster, not guaranteed)
    __syncthreads();
    if (timed) clk[1] = __builtin_readcyclecounter();
    const uint32_t tile = s_tile;
    const uint32_t m = A.m;
    // the regroup tiles are the tiles of the segmented sort, or kSubTiles equal pieces of each (a piece behind
    // the end of a partial sort tile is empty)
    static_assert(kSortTile % kFuseTile == 0, "a sort tile is a whole number of regroup tiles");
    constexpr uint32_t kSubTiles = kSortTile / kFuseTile;
    TileExtent ext;
    if (A.seg.desc == nullptr) {
        ext = tile_extent(0, m, 1, A.seg);
        ext.first = (size_t)tile * kFuseTile;
        ext.count = (uint32_t)((m - ext.first < (size_t)kFuseTile) ? (m - ext.first) : (size_t)kFuseTile);
    } else {
        ext = tile_extent(tile / kSubTiles, m, A.num_tiles / kSubTiles, A.seg);
        const uint32_t off = (tile % kSubTiles) * (uint32_t)kFuseTile;
        const uint32_t skip = off < ext.count ? off : ext.count;
        ext.first += skip;
        ext.count -= skip;
        ext.count = ext.count < (uint32_t)kFuseTile ? ext.count : (uint32_t)kFuseTile;
    }
    const size_t tile_base = ext.first;
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const uint64_t lt = lanemask_lt();
    const bool bucketed = kDnaFast || (kRound0 && A.keys32 != nullptr);

    auto load_view = [&](size_t a) -> uint64_t {
        if (kRound0) {
            if (!bucketed) return A.keys[a];
            // the element's bucket: the tile's own, unless a is a neighbour across the bucket's end
            const uint32_t b = a < ext.bkt_first ? ext.prev_ne : (a >= ext.bkt_end ? ext.next_ne : ext.bucket);
            return ((uint64_t)b << 32) | A.keys32[a];
        }
        return ((uint64_t)A.grp[a] << 32) | A.lo[a];
    };
    // striped: item k of thread t is element tile_base + k * kFuseThreads + t (coalesced rows).
    // Every load of the tile goes out first (the LCP stores further down may alias the inputs as
    // far as the compiler knows; interleaved, each of the 16 rows would wait for its own round
    // trips to HBM): the view of my elements, their slots, and per row ONE neighbour -- the
    // element in front of the wavefront for lane 0, the element behind it for lane 63.
    uint32_t slot[kFuseItems];
    uint64_t view[kFuseItems], edge[kFuseItems];
#pragma unroll
    for (int k = 0; k < kFuseItems; ++k) {
        const size_t a = tile_base + (size_t)k * kFuseThreads + threadIdx.x;
        const bool in = (uint32_t)k * kFuseThreads + threadIdx.x < ext.count;
        view[k] = in ? load_view(a) : 0ull;
        slot[k] = kRound0 ? (uint32_t)a : (in ? A.act_slot[a] : 0u);
        edge[k] = 0;
        if (lane == 0 && in && a > 0) edge[k] = load_view(a - 1);
        if (lane == 63 && in && a + 1 < m) edge[k] = load_view(a + 1);
    }
    uint64_t hmask[kFuseItems], kmask[kFuseItems];  // wave-uniform: heads / kept elements of my segment
    uint32_t old_head[kFuseItems];                  // later rounds: head slot of the group I come from
#pragma unroll
    for (int k = 0; k < kFuseItems; ++k) {
        const size_t a = tile_base + (size_t)k * kFuseThreads + threadIdx.x;
        const bool in = (uint32_t)k * kFuseThreads + threadIdx.x < ext.count;
        const uint64_t v = view[k];
        old_head[k] = (uint32_t)(v >> 32);
        // the element in front: the previous lane's, except for lane 0
        const uint64_t pv = ((uint64_t)lane_prev((uint32_t)(v >> 32), (uint32_t)(edge[k] >> 32)) << 32) |
                            lane_prev((uint32_t)v, (uint32_t)edge[k]);
        bool head = !in || a == 0 || v != pv;  // "past the end" counts as a head
        // a suffix that meets a terminator inside the key window ties only with copies of itself at
        // other terminators, and the stable sort has left those in their final order
        if (kRound0 && short_tag) head = head || ((uint32_t)(v >> low_bits) & ((1u << tag_bits) - 1u)) < short_tag;
        // is the element behind me a head?
        const uint32_t edge_next = (lane == 63 && in && a + 1 < m) ? (edge[k] != v ? 1u : 0u) : 1u;
        const bool next_head = lane_next(head ? 1u : 0u, edge_next) != 0;
        const bool keep = in && !(head && next_head);
        hmask[k] = __ballot(in && head);
        kmask[k] = __ballot(keep);
        // the LCP of a boundary that has just appeared needs nothing from the other tiles
        if (in && !kRound0) {
            // a new boundary inside an old group
            if (head && a > 0 && (uint32_t)(v >> 32) == (uint32_t)(pv >> 32)) {
                uint32_t l = A.lcp_list ? A.lcp_list[a] : kLcpPending;
                if (l >= kLcpPendingMin) {
                    // created by a doubling step with offset h: the two suffixes agree on h symbols
                    // and continue with suffixes of DIFFERENT h-groups, whose LCP is the minimum of
                    // the boundaries already decided between those groups (undecided entries hold
                    // pending codes, i.e. +infinity):  lcp = h + min LCP(head1 .. head2]
                    const uint32_t p = (uint32_t)pv, q = (uint32_t)v;  // rank codes: head slot + 1
                    l = A.dbl_h;
                    if (p != 0) l += pyr_range<false>(A.Plcp, p, q - 1);
                }
                A.lcp[slot[k]] = l;
                // the range-minimum pyramid over the LCP array is kept up to date instead of being
                // rebuilt every round: a decided value only ever replaces a pending code (+infinity)
                for (int lev = 1; lev < A.Plcp.nlev; ++lev) {
                    uint32_t *up = const_cast<uint32_t *>(A.Plcp.lvl[lev]) + (slot[k] >> (kPyrShift * lev));
                    if (atomicMin(up, l) <= l) break;
                }
            }
        } else if (in) {
            // LCP of neighbours that round 0 already separates can be read off the two keys
            // (symbol prefix, capped by both length tags); the rest is marked pending.
            uint32_t l = kLcpPending;
            if (a == 0) {
                l = 0;
            } else if (head) {
                const uint64_t ka = v >> low_bits, kb = pv >> low_bits;
                const uint64_t tmask = (1ull << tag_bits) - 1ull;
                const uint32_t ta = (uint32_t)(ka & tmask), tb = (uint32_t)(kb & tmask);
                const uint64_t x = (ka ^ kb) >> tag_bits << (64 - sym_bits);  // symbols, left-aligned
                uint32_t ls = x ? (uint32_t)__clzll((long long)x) >> bits_shift : 0xffffffffu;  // (bits per symbol is 2, 4 or 8)
                ls = ls < ta ? ls : ta;
                l = ls < tb ? ls : tb;
                if (seq_shift && (v >> seq_shift) != (pv >> seq_shift)) l = 0;  // different sequences
            }
            A.lcp[a] = l;
        }
        // segment aggregate: slot of its last head (slots grow along the list), elements kept
        uint32_t last = 0;
        if (hmask[k]) last = (uint32_t)__builtin_amdgcn_readlane((int)slot[k], 63 - __builtin_clzll(hmask[k]));
        if (lane == 0) {
            s_seg_max[k * kWaves + w] = last;
            s_seg_sum[k * kWaves + w] = (uint32_t)__popcll(kmask[k]);
        }
    }
    __syncthreads();
    if (timed) clk[2] = __builtin_readcyclecounter();
    if (w == 0) {  // prefixes over the segments, then over the tiles in front
        static_assert(kSegs <= 64, "one lane per segment");
        const uint32_t vmax = lane < kSegs ? s_seg_max[lane] : 0u;
        const uint32_t vsum = lane < kSegs ? s_seg_sum[lane] : 0u;
        const uint32_t imax = wave_scan_inclusive_dpp(vmax, 0u, OpMax<uint32_t>());
        const uint32_t isum = wave_scan_inclusive_dpp(vsum, 0u, OpAdd<uint32_t>());
        const uint32_t agg_max = (uint32_t)__builtin_amdgcn_readlane((int)imax, 63);
        const uint32_t agg_sum = (uint32_t)__builtin_amdgcn_readlane((int)isum, 63);
        const uint32_t emax = lane_prev(imax, 0u);
        if (lane < kSegs) {
            s_seg_max[lane] = emax;
            s_seg_sum[lane] = isum - vsum;
        }
        uint32_t xm, xs;
        if (A.packed) {  // lists shorter than 2^31: one walk for both values
            const MaxSum x = lookback_exclusive_packed(A.desc_max, tile, MaxSum{agg_max, agg_sum}, A.ticket + 1);
            xm = x.mx;
            xs = x.sum;
        } else {
            xm = lookback_exclusive(A.desc_max, tile, agg_max, OpMax<uint32_t>(), A.ticket + 1);
            xs = lookback_exclusive(A.desc_sum, tile, agg_sum, OpAdd<uint32_t>(), A.ticket + 1);
        }
        if (lane == 0) {
            s_excl[0] = xm;
            s_excl[1] = xs;
            if (tile + 1 == A.num_tiles) *A.d_total = xs + agg_sum;  // the last tile
        }
    }
    __syncthreads();
    if (timed) clk[3] = __builtin_readcyclecounter();
    const uint32_t xmax = s_excl[0], xsum = s_excl[1];

    // slot of my group head: the last head at or in front of me
    auto head_slot = [&](int k, size_t a) -> uint32_t {
        const uint64_t mine = hmask[k] & ((2ull << lane) - 1ull);
        const int hl = mine ? 63 - __builtin_clzll(mine) : lane;
        uint32_t head_of = kRound0 ? (uint32_t)(a - (size_t)(lane - hl)) : (uint32_t)__shfl((int)slot[k], hl, 64);
        if (!mine) {
            const uint32_t pm = s_seg_max[k * kWaves + w];
            head_of = pm > xmax ? pm : xmax;
        }
        return head_of;
    };
    // Doubling rounds: rank[i] changes only for the members of groups that split off their old group
    // (on long exact repeats a round moves a few hundred of 10^8 tied suffixes).  Those go, in any
    // order, to the list that bucketed_scatter writes into rank[]: the tile counts them, takes its part
    // of the list with ONE atomic, and every wavefront appends its own.
    const bool list_changes = !kRound0 && !A.rank_by_slot;  // (uniform)
    uint64_t cmask[kFuseItems];
    uint32_t chg_base = 0;
    if (list_changes) {
        __shared__ uint32_t s_chg[kWaves + 1];
        uint32_t mine_total = 0;
#pragma unroll
        for (int k = 0; k < kFuseItems; ++k) {
            const size_t a = tile_base + (size_t)k * kFuseThreads + threadIdx.x;
            const bool in = (uint32_t)k * kFuseThreads + threadIdx.x < ext.count;
            cmask[k] = __ballot(in && head_slot(k, a) != old_head[k]);
            mine_total += (uint32_t)__popcll(cmask[k]);
        }
        if (lane == 0) s_chg[w] = mine_total;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t all = 0;
#pragma unroll
            for (int k = 0; k < kWaves; ++k) {
                const uint32_t c = s_chg[k];
                s_chg[k] = all;
                all += c;
            }
            s_chg[kWaves] = all ? atomicAdd(A.chg_count, all) : 0u;
        }
        __syncthreads();
        chg_base = s_chg[kWaves] + s_chg[w];
    }

#pragma unroll
    for (int k = 0; k < kFuseItems; ++k) {
        const size_t a = tile_base + (size_t)k * kFuseThreads + threadIdx.x;
        const bool in = (uint32_t)k * kFuseThreads + threadIdx.x < ext.count;
        const uint32_t head_of = head_slot(k, a);
        if (list_changes) {
            if ((cmask[k] >> lane) & 1ull) {
                const uint32_t q = chg_base + (uint32_t)__popcll(cmask[k] & lt);
                A.chg_idx[q] = A.vals[a];
                A.rank_val[q] = head_of + 1u;
            }
            chg_base += (uint32_t)__popcll(cmask[k]);
        }
        if (!in) continue;
        // (round 0: the key sort left the suffixes in sa itself; direct round: group_refine_kernel did)
        if (!kRound0 && !A.sa_is_current) A.sa[slot[k]] = A.vals[a];
        // (A.rank_by_slot != nullptr: rank[] is written later, in one pass -- or never: build_suffix_array, which then
        // asks for no store here; should ranks be needed after all, they are recovered from the LCP array)
        if (A.rank_by_slot && A.store_ranks) A.rank_by_slot[slot[k]] = head_of + 1u;
        if ((kmask[k] >> lane) & 1ull) {  // surviving elements keep their slot, learn their group head
            const uint32_t kk = xsum + s_seg_sum[k * kWaves + w] + (uint32_t)__popcll(kmask[k] & lt);
            A.new_slot[kk] = slot[k];
            A.new_grp[kk] = head_of;
        }
    }
    if (timed) {
        __builtin_amdgcn_s_waitcnt(0);
        clk[4] = __builtin_readcyclecounter();
        for (int k = 0; k < 4; ++k) atomicAdd(A.phases + k, clk[k + 1] - clk[k]);
        atomicAdd(A.phases + 4, 1ull);
    }
}

// ---------------------------------------------------------------------------------------
// Long exact repeats.  After the direct round a text with long repeats (similar genomes, a duplicated
// region) is left with millions of small groups of suffixes -- i and i + d for two copies -- that agree
// on more than the cap.  Doubling would need log2(repeat length) rounds over all of them although the
// answer is arithmetic: LCP(i, j) = 1 + LCP(i + 1, j + 1), and the order of (i, j) is the order of
// (i + 1, j + 1).  Along a RUN of text positions i, i + 1, ... whose groups keep the same shape (the same
// distances between the members), everything follows from the group behind the end of the run, and that
// one is already separated (its members carry different rank codes): the order is the order of the codes,
// the LCP of neighbours 1 + the range minimum of the LCP values decided so far -- the rule a doubling step
// applies, with h = 1.  Runs are contiguous in TEXT order, so "where does my run end" is one prefix scan,
// not pointer jumping.  Groups of up to kRunGroupMax members are handled; runs whose end group is only
// partly separated are left to the doubling rounds.
// ---------------------------------------------------------------------------------------
constexpr uint32_t kRunGroupMax = 16;

// members of the undecided group with head slot g: k = its size (0: decided or too large)
__device__ __forceinline__ uint32_t run_group_size(const uint32_t *__restrict__ lcp, uint32_t n, uint32_t g) {
    uint32_t k = 1;
    while (k <= kRunGroupMax && g + k < n && lcp[g + k] >= kLcpPendingMin) ++k;
    return (k >= 2 && k <= kRunGroupMax) ? k : 0u;
}

// link[i] = (next member of my group in text order, cyclically) - i, gsz[i] = size of my group; 0 / 0 if
// suffix i is decided or its group is too large
__global__ __launch_bounds__(kThreads) void group_link_kernel(const uint32_t *__restrict__ rank,
                                                              const uint32_t *__restrict__ sa,
                                                              const uint32_t *__restrict__ lcp, uint32_t n,
                                                              uint32_t *__restrict__ link, uint32_t *__restrict__ gsz) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t g = rank[i] - 1u;  // head slot of my group
        const uint32_t k = run_group_size(lcp, n, g);
        uint32_t d = 0;
        if (k) {
            uint32_t above = 0xffffffffu, lowest = 0xffffffffu;
            for (uint32_t x = 0; x < k; ++x) {
                const uint32_t m = sa[g + x];
                lowest = m < lowest ? m : lowest;
                if (m > (uint32_t)i && m < above) above = m;
            }
            d = (above != 0xffffffffu ? above : lowest) - (uint32_t)i;
        }
        link[i] = d;
        gsz[i] = d ? k : 0u;
    }
}

// does the chain of position t go on at t + 1?
__device__ __forceinline__ bool run_goes_on(const uint32_t *__restrict__ link, const uint32_t *__restrict__ gsz,
                                            uint32_t n, size_t t) {
    const uint32_t d = link[t];
    return d != 0 && t + 1 < n && link[t + 1] == d && gsz[t + 1] == gsz[t];
}

// rev[n - 1 - t] = (n - 1 - t) + 1 where the chain of t ends at t (or t is in no group), else 0: an
// inclusive max-scan over rev then names, for every t, the nearest such end at or behind it
__global__ __launch_bounds__(kThreads) void run_breaks_kernel(const uint32_t *__restrict__ link,
                                                              const uint32_t *__restrict__ gsz, uint32_t n,
                                                              uint32_t *__restrict__ rev) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride)
        rev[n - 1 - t] = run_goes_on(link, gsz, n, t) ? 0u : (uint32_t)(n - 1 - t) + 1u;
}

// togo[i] = steps until the run of my GROUP ends: the shortest chain of its members (the group one
// step further on is my group shifted by one only while every member's chain goes on)
__global__ __launch_bounds__(kThreads) void group_run_kernel(const uint32_t *__restrict__ gsz,
                                                             const uint32_t *__restrict__ rank,
                                                             const uint32_t *__restrict__ sa,
                                                             const uint32_t *__restrict__ end_of, uint32_t n,
                                                             uint32_t *__restrict__ togo) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t k = gsz[i];
        if (k <= 2) continue;  // (a pair's two chains are equally long: run_steps takes its own)
        const uint32_t g = rank[i] - 1u;
        uint32_t best = 0xffffffffu;
        for (uint32_t x = 0; x < k; ++x) {
            const uint32_t m = sa[g + x];
            const uint32_t e = (uint32_t)(n - 1) - (end_of[n - 1 - m] - 1u);  // where the chain of m ends (>= m)
            best = e - m < best ? e - m : best;
        }
        togo[i] = best;
    }
}

// steps from position i to the end of its group's run
__device__ __forceinline__ uint32_t run_steps(uint32_t k, size_t i, uint32_t n, const uint32_t *__restrict__ end_of,
                                              const uint32_t *__restrict__ togo) {
    if (k > 2) return togo[i];
    return ((uint32_t)(n - 1) - (end_of[n - 1 - i] - 1u)) - (uint32_t)i;
}

constexpr uint32_t kRunDeferred = 0xffffffffu;

// Members of a group at the end of its run.  One symbol further on the members carry rank codes; equal
// codes mean "still tied".  The group splits into classes of equal code, in code order: my slot inside
// the group, the first slot of my class (my new group head), and -- if I am the first of a class that is
// not the first -- the LCP to the class in front (decided now).  A class of one is a finished suffix.
__global__ __launch_bounds__(kThreads) void group_end_kernel(const uint32_t *__restrict__ gsz,
                                                             const uint32_t *__restrict__ togo,
                                                             const uint32_t *__restrict__ end_of,
                                                             const uint32_t *__restrict__ rank,
                                                             const uint32_t *__restrict__ sa, uint32_t n, Pyramid Plcp,
                                                             uint32_t *__restrict__ end_place,
                                                             uint32_t *__restrict__ end_head,
                                                             uint32_t *__restrict__ end_lcp) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
        const uint32_t k = gsz[t];
        if (!k || run_steps(k, t, n, end_of, togo) != 0) continue;
        const uint32_t g = rank[t] - 1u;
        uint32_t below = 0, same_before = 0, pred = 0, l = kRunDeferred;
        bool off_end = t + 1 >= n;
        const uint32_t mine = off_end ? 0u : rank[t + 1];  // rank code (head slot + 1) one symbol further on
        for (uint32_t x = 0; x < k; ++x) {
            const uint32_t m = sa[g + x];
            if (m == (uint32_t)t) continue;
            if ((size_t)m + 1 >= n) {  // (every member sees this: the group is left alone as a whole)
                off_end = true;
                continue;
            }
            const uint32_t c = rank[m + 1];
            if (c < mine) {
                ++below;
                pred = c > pred ? c : pred;
            } else if (c == mine && m < (uint32_t)t) {
                ++same_before;
            }
        }
        if (!off_end && same_before == 0 && below > 0) l = 1u + pyr_range<false>(Plcp, pred, mine - 1u);
        end_place[t] = off_end ? kRunDeferred : below + same_before;
        end_head[t] = below;
        end_lcp[t] = l;
    }
}

// every member of every group of a run does what its counterpart in the end group does
__global__ __launch_bounds__(kThreads) void group_members_kernel(const uint32_t *__restrict__ gsz,
                                                                 const uint32_t *__restrict__ togo,
                                                                 const uint32_t *__restrict__ end_of, uint32_t n,
                                                                 const uint32_t *__restrict__ end_place,
                                                                 const uint32_t *__restrict__ end_head,
                                                                 const uint32_t *__restrict__ end_lcp,
                                                                 uint32_t *__restrict__ rank, uint32_t *__restrict__ sa,
                                                                 uint32_t *__restrict__ lcp) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t k = gsz[i];
        if (!k) continue;
        const uint32_t steps = run_steps(k, i, n, end_of, togo);
        const size_t e = i + steps;  // my position in the end group of the run
        const uint32_t place = end_place[e];
        if (place == kRunDeferred) continue;
        const uint32_t g = rank[i] - 1u;
        sa[g + place] = (uint32_t)i;
        const uint32_t le = end_lcp[e];
        if (le != kRunDeferred) lcp[g + place] = le + steps;
        rank[i] = g + end_head[e] + 1u;
    }
}

// the active list after a pass: head slot of every element's (new) group, 1 if that group is still undecided
// (the head is the nearest slot at or in front of mine whose boundary is decided; only groups of up to
// kRunGroupMax members were touched, so the walk back is that short -- the list is in slot order, the
// LCP entries it reads are neighbours in memory)
__global__ __launch_bounds__(kThreads) void still_tied_kernel(const uint32_t *__restrict__ act_slot,
                                                              const uint32_t *__restrict__ act_grp, uint32_t m,
                                                              const uint32_t *__restrict__ lcp, uint32_t n,
                                                              uint32_t *__restrict__ head, uint32_t *__restrict__ keep) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x; a < m; a += stride) {
        const uint32_t s = act_slot[a], g0 = act_grp[a];
        uint32_t h = g0;
        if (s - g0 < kRunGroupMax) {
            h = s;
            while (h > g0 && lcp[h] >= kLcpPendingMin) --h;
        }
        head[a] = h;
        keep[a] = (h + 1u < n && lcp[h + 1] >= kLcpPendingMin) ? 1u : 0u;
    }
}

__global__ __launch_bounds__(kThreads) void compact_active_kernel(const uint32_t *__restrict__ act_slot,
                                                                  const uint32_t *__restrict__ head,
                                                                  const uint32_t *__restrict__ keep,
                                                                  const uint32_t *__restrict__ pos, uint32_t m,
                                                                  uint32_t *__restrict__ new_slot,
                                                                  uint32_t *__restrict__ new_grp) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x; a < m; a += stride)
        if (keep[a]) {
            new_slot[pos[a]] = act_slot[a];
            new_grp[pos[a]] = head[a];
        }
}

// out[q] = q + 1 where a group starts at slot q (its LCP entry is decided), else 0
__global__ __launch_bounds__(kThreads) void head_flags_kernel(const uint32_t *__restrict__ lcp, uint32_t n,
                                                              uint32_t *__restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += stride)
        out[q] = lcp[q] < kLcpPendingMin ? (uint32_t)q + 1u : 0u;
}

// secondary key of a doubling round: rank of the suffix h symbols further on (0 past the end)
__global__ __launch_bounds__(kThreads) void round_keys_kernel(const uint32_t *__restrict__ act_slot,
                                                              uint32_t m, const uint32_t *__restrict__ sa,
                                                              const uint32_t *__restrict__ rank, uint32_t n,
                                                              uint32_t h, uint32_t *__restrict__ lo,
                                                              uint32_t *__restrict__ vals) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x; a < m; a += stride) {
        const uint32_t i = sa[act_slot[a]];
        lo[a] = (n - i > h) ? rank[i + h] : 0u;  // i + h < n without overflow
        vals[a] = i;
    }
}

// *p = min(*p, v) for a value that millions of wavefronts report and that soon stops changing: look
// first, the atomic only if it would lower the value (5 M atomics on one address cost 20 ms)
__device__ __forceinline__ void lower_min(uint32_t *p, uint32_t v) {
    if (*reinterpret_cast<volatile uint32_t *>(p) > v) atomicMin(p, v);
}

// ---------------------------------------------------------------------------------------
// Periodic runs.  On a text with long runs of a short period (a poly-A tract, a tandem repeat, a period-1000
// text) the suffixes of a run tie on whatever depth h has been compared, in groups far larger than the
// pair-run pass takes, and prefix doubling peels only h of them off per round: log2(run length) rounds over
// everything.  The order inside such a group is arithmetic.  Let q be the smallest distance between two
// members of a group in the text, q <= h: the h symbols every member starts with then have period q (two
// members q apart agree on h symbols, so h + q symbols have that period, and every other member starts with
// the same h symbols), a prefix of u^inf for one word u that those h >= q symbols determine.  For a member x let rho(x) = q + lcp(x, x + q): the text keeps that period
// for exactly rho(x) symbols from x; at x + rho(x) it breaks -- with a symbol smaller than the periodic
// continuation ("down", also when the text ends there) or larger ("up").  Two members with different rho
// agree on min(rho) symbols and the one that breaks first goes down below / up above the other; so the group
// in suffix order is: the down members by ascending rho, then the up members by descending rho, the LCP of
// neighbours with different keys being the smaller rho.  Members with the same key stay tied (a smaller
// group for the next pass or the doubling rounds).
// lcp(x, x + q) needs no text: along a run of text positions t, t + 1, .. whose suffixes all have their next
// group member q behind them, lcp(t, t + q) = 1 + lcp(t + 1, t + 1 + q), so it is the distance to the end E
// of that run of positions plus lcp(E, E + q), and suffixes E and E + q are in DIFFERENT groups: their order
// is the order of their rank codes and their LCP the range minimum of the decided LCP entries between them.
// A group with a member for which that fails (E and E + q tied with each other) is left alone as a whole.
// A group whose q exceeds the depth h compared so far (after the 17-base key sort a large group has only
// been compared to depth 17: a 171-base satellite monomer, a period-1000 text) is taken if the TEXT shows
// that its members agree on q symbols -- every member is compared with the next member of its group in text
// order, q symbols deep (per_verify_kernel; q <= kPerVerifyMax) -- and left to the doubling rounds otherwise.
// ---------------------------------------------------------------------------------------
constexpr uint32_t kPerNone = 0xffffffffu;   // gq: no distance seen yet
constexpr uint32_t kPerBad = 0x80000000u;    // gq: flag "leave this group alone" (positions are below 2^31 here)

// count[0] += members beyond the first `limit` of their group, count[1] += groups with more than `limit` members,
// count[2] += groups (the list is in slot order: a member's index inside its group is slot - head), count[3] +=
// members whose successor in the list belongs to the same group and starts at most `near` symbols away in the text
// (tied members keep the order of their text positions through every stable step of the construction, so these are
// -- as an estimate, used to decide whether a pass is worth its sorts -- the members of periodic runs): one atomic
// per counter and workgroup
__global__ __launch_bounds__(kThreads) void per_count_large_kernel(const uint32_t *__restrict__ act_slot,
                                                                   const uint32_t *__restrict__ act_grp, uint32_t m,
                                                                   const uint32_t *__restrict__ sa, uint32_t limit,
                                                                   uint32_t near, uint32_t *__restrict__ count) {
    uint32_t c[4] = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x; a < m; a += stride) {
        const uint32_t slot = act_slot[a], g = act_grp[a];
        const uint32_t j = slot - g;
        c[0] += j >= limit ? 1u : 0u;
        c[1] += j == limit ? 1u : 0u;
        c[2] += j == 0 ? 1u : 0u;
        if (a + 1 < m && act_grp[a + 1] == g) {
            const uint32_t p = sa[slot], q = sa[act_slot[a + 1]];
            const uint32_t d = p < q ? q - p : p - q;
            c[3] += d <= near ? 1u : 0u;
        }
    }
    __shared__ uint32_t s_part[4][kThreads / 64];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t r = wave_reduce(c[k], OpAdd<uint32_t>());
        if (lane_id() == 0) s_part[k][threadIdx.x >> 6] = r;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        uint32_t t = 0;
        for (int i = 0; i < kThreads / 64; ++i) t += s_part[threadIdx.x][i];
        if (t) atomicAdd(count + threadIdx.x, t);
    }
}

// members of groups whose smallest distance between neighbours is at most `limit` (the candidates of the periodic
// pass): one atomic per workgroup
__global__ __launch_bounds__(kThreads) void per_candidates_kernel(const uint64_t *__restrict__ keys, uint32_t m,
                                                                  const uint32_t *__restrict__ gq, uint32_t limit,
                                                                  uint32_t *__restrict__ count) {
    uint32_t mine = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride)
        mine += gq[(uint32_t)(keys[j] >> 32)] <= limit ? 1u : 0u;
    mine = wave_reduce(mine, OpAdd<uint32_t>());
    __shared__ uint32_t s_part[kThreads / 64];
    if (lane_id() == 0) s_part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int i = 0; i < kThreads / 64; ++i) t += s_part[i];
        if (t) atomicAdd(count, t);
    }
}

__global__ __launch_bounds__(kThreads) void per_keys_kernel(const uint32_t *__restrict__ act_slot,
                                                            const uint32_t *__restrict__ act_grp, uint32_t m,
                                                            const uint32_t *__restrict__ sa,
                                                            uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x; a < m; a += stride) {
        const uint32_t pos = sa[act_slot[a]];
        keys[a] = ((uint64_t)act_grp[a] << 32) | pos;
        vals[a] = pos;
    }
}

// list sorted by (group, position): gq[group] = smallest distance between neighbours.  One atomic per
// workgroup / wavefront where it holds one group only (a giant group would otherwise send every lane to
// one address, 13 ns each).  The grid covers the list exactly once (no stride loop: barriers inside).
__global__ __launch_bounds__(kThreads) void per_link_kernel(const uint64_t *__restrict__ keys, uint32_t m,
                                                            uint32_t *__restrict__ gq) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = j < m;
    const uint64_t k = in ? keys[j] : 0;
    const uint32_t g = (uint32_t)(k >> 32);
    uint32_t link = kPerNone;
    if (in && j + 1 < m) {
        const uint64_t k2 = keys[j + 1];
        if ((uint32_t)(k2 >> 32) == g) link = (uint32_t)k2 - (uint32_t)k;
    }
    __shared__ uint32_t s_g0, s_min[kThreads / 64];
    if (threadIdx.x == 0) s_g0 = g;
    __syncthreads();
    const int uniform = __syncthreads_and(in && g == s_g0);
    if (uniform) {
        const uint32_t w = wave_reduce(link, OpMinU32x());
        if (lane_id() == 0) s_min[threadIdx.x >> 6] = w;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t b = s_min[0];
            for (int i = 1; i < kThreads / 64; ++i) b = s_min[i] < b ? s_min[i] : b;
            if (b != kPerNone) atomicMin(&gq[g], b);
        }
        return;
    }
    const uint32_t g_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)g);
    if (__ballot(!in || g != g_first) == 0) {  // the wavefront holds one group
        const uint32_t w = wave_reduce(link, OpMinU32x());
        if (lane_id() == 0 && w != kPerNone) atomicMin(&gq[g], w);
    } else if (in && link != kPerNone) {
        atomicMin(&gq[g], link);
    }
}

// PQ[pos] = q for a member whose next group member is exactly q behind it, q = the group's distance (0 for
// everything else; the array was cleared).  Groups whose q exceeds half the depth compared so far are
// flagged; the smallest such q is reported (hint: try again when the depth has passed twice that).
constexpr uint32_t kPerVerifyMax = 4096;  // longest period whose groups are checked against the text

// members of groups with depth < q <= kPerVerifyMax: do I agree with the next member of my group (in text
// order) on q symbols?  If every such pair does, all members agree on q symbols.  Pairs exactly q apart need
// no text: lcp(x, x + q) = (E - x) + lcp(E, E + q) is known from the run of positions (per_rho_kernel).
template <int BITS>
__global__ __launch_bounds__(kThreads) void per_verify_kernel(const uint64_t *__restrict__ keys, uint32_t m,
                                                              uint32_t *__restrict__ gq, uint32_t depth, uint32_t n,
                                                              const uint64_t *__restrict__ words, TermTable terms) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j + 1 < m; j += stride) {
        const uint64_t k = keys[j], k2 = keys[j + 1];
        const uint32_t g = (uint32_t)(k >> 32);
        if ((uint32_t)(k2 >> 32) != g) continue;
        const uint32_t gv = *reinterpret_cast<volatile uint32_t *>(&gq[g]);
        if (gv & kPerBad) continue;
        const uint32_t q = gv;
        if (q <= depth || q > kPerVerifyMax) continue;
        // (plain texts only: the later member is the shorter suffix, so "agrees on min(q, what the later one has
        // left)" carries from pair to pair -- every member starts with the group's period word as far as it goes)
        const uint32_t b = (uint32_t)k2, left = n - b, need = q < left ? q : left;
        // (a pair exactly q apart is checked without the text, from the length of its run of positions:
        // per_rho_kernel; what is compared here are the few pairs that join two runs)
        if (b - (uint32_t)k == q) continue;
        if (suffix_lcp<BITS>(words, terms, (uint32_t)k, b, 0u, q) < need) atomicOr(&gq[g], kPerBad);
    }
}

__global__ __launch_bounds__(kThreads) void per_flags_kernel(const uint64_t *__restrict__ keys, uint32_t m,
                                                             uint32_t *__restrict__ gq, uint32_t half_depth,
                                                             uint32_t *__restrict__ PQ, uint32_t *__restrict__ hint) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
        const uint64_t k = keys[j];
        const uint32_t g = (uint32_t)(k >> 32), pos = (uint32_t)k;
        const uint32_t q = *reinterpret_cast<volatile uint32_t *>(&gq[g]) & ~kPerBad;
        const bool head = j == 0 || (uint32_t)(keys[j - 1] >> 32) != g;
        if (q > half_depth) {  // (also kPerNone & ~kPerBad)
            if (head) {
                atomicOr(&gq[g], kPerBad);
                if (q != (kPerNone & ~kPerBad)) lower_min(hint, q);
            }
            continue;
        }
        uint32_t link = 0;
        if (j + 1 < m) {
            const uint64_t k2 = keys[j + 1];
            if ((uint32_t)(k2 >> 32) == g) link = (uint32_t)k2 - pos;
        }
        if (link == q) PQ[pos] = q;
    }
}

// rev[n - 1 - t] = (n - 1 - t) + 1 unless the run of positions goes on from t to t + 1 (both carry the same
// distance): the inclusive max-scan of rev names, for every t, the last position of its run
__global__ __launch_bounds__(kThreads) void per_breaks_kernel(const uint32_t *__restrict__ PQ, uint32_t n,
                                                              uint32_t *__restrict__ rev) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
        const uint32_t q = PQ[t];
        const bool on = q != 0 && t + 1 < n && PQ[t + 1] == q;
        rev[n - 1 - t] = on ? 0u : (uint32_t)(n - 1 - t) + 1u;
    }
}

// sort key of every member: rho for the down members, ~rho for the up members (0 is never a key)
__global__ __launch_bounds__(kThreads) void per_rho_kernel(const uint64_t *__restrict__ keys, uint32_t m,
                                                           uint32_t *__restrict__ gq, const uint32_t *__restrict__ PQ,
                                                           const uint32_t *__restrict__ end_of,
                                                           const uint32_t *__restrict__ rank, uint32_t n, Pyramid Plcp,
                                                           uint32_t depth, uint32_t *__restrict__ kraw) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
        const uint64_t k = keys[j];
        const uint32_t g = (uint32_t)(k >> 32), pos = (uint32_t)k;
        const uint32_t gv = *reinterpret_cast<volatile uint32_t *>(&gq[g]);
        kraw[j] = 0;
        if (gv & kPerBad) continue;
        const uint32_t q = gv;
        // E: the first position at or behind pos whose suffix does NOT have its next group member q behind it
        uint32_t E = pos;
        if (PQ[pos] == q) E = ((n - 1) - (end_of[n - 1 - pos] - 1u)) + 1u;
        bool ok = E < n;
        uint32_t lam = 0, c1 = 0, c2 = 0;
        if (ok && (uint64_t)E + q >= n) {
            // the text ends before suffix E has seen a whole period (only a member without a next member:
            // E = pos): periodic as far as it goes, and the end sorts first
            const uint32_t full = (E - pos) + q, left = n - pos;
            kraw[j] = full < left ? full : left;
            continue;
        }
        if (ok) {
            c1 = rank[E];
            const uint64_t e2 = (uint64_t)E + q;
            c2 = e2 < n ? rank[e2] : 0u;  // (e2 < n here)
            if (c1 == c2) {
                ok = false;  // tied with each other: nothing is known about them yet
            } else if (c2 != 0) {
                const uint32_t a = c1 < c2 ? c1 : c2, b = c1 < c2 ? c2 : c1;
                lam = pyr_range<false>(Plcp, a, b - 1u);  // decided entries between the two groups
                if (lam >= kLcpPendingMin) ok = false;
            }
        }
        if (!ok) {
            atomicOr(&gq[g], kPerBad);
            continue;
        }
        // a group taken on a period longer than the depth compared so far: this member and the next one
        // (q behind it) must agree on q symbols, or as far as the later one goes
        if (q > depth && E != pos) {
            const uint32_t left = n - (pos + q), need = q < left ? q : left;
            if ((E - pos) + lam < need) {
                atomicOr(&gq[g], kPerBad);
                continue;
            }
        }
        const uint32_t rho = (E - pos) + lam + q;  // <= n - pos
        kraw[j] = c2 < c1 ? rho : ~rho;            // down (suffix E + q is the smaller one) : up
    }
}

// second sort key (group, K): K = 0 for every member of a group that is left alone
__global__ __launch_bounds__(kThreads) void per_keys2_kernel(const uint64_t *__restrict__ keys, uint32_t m,
                                                             const uint32_t *__restrict__ gq,
                                                             const uint32_t *__restrict__ kraw,
                                                             uint64_t *__restrict__ keys2, uint32_t *__restrict__ vals2) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
        const uint64_t k = keys[j];
        const uint32_t g = (uint32_t)(k >> 32);
        const uint32_t K = (gq[g] & kPerBad) ? 0u : kraw[j];
        keys2[j] = ((uint64_t)g << 32) | K;
        vals2[j] = (uint32_t)k;
    }
}

// the sorted view the regroup kernel takes, and the LCP of every boundary that appears inside an old group
__global__ __launch_bounds__(kThreads) void per_view_kernel(const uint64_t *__restrict__ keys2,
                                                            const uint32_t *__restrict__ vals2, uint32_t m,
                                                            uint32_t *__restrict__ grp, uint32_t *__restrict__ lo,
                                                            uint32_t *__restrict__ vals, uint32_t *__restrict__ lcp_list) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
        const uint64_t k = keys2[j];
        const uint32_t g = (uint32_t)(k >> 32), K = (uint32_t)k;
        grp[j] = g;
        lo[j] = K;
        vals[j] = vals2[j];
        uint32_t l = kLcpPending;
        if (j > 0) {
            const uint64_t kp = keys2[j - 1];
            const uint32_t Kp = (uint32_t)kp;
            if ((uint32_t)(kp >> 32) == g && Kp != K) {
                const uint32_t ra = (Kp & 0x80000000u) ? ~Kp : Kp, rb = (K & 0x80000000u) ? ~K : K;
                l = ra < rb ? ra : rb;
            }
        }
        lcp_list[j] = l;
    }
}

// Segmented sort of the active list by lo inside each group.  Groups are contiguous in the
// list and (for real sequence data) almost all tiny, so each element finds its place by
// counting the smaller members of its own group -- one pass, no radix passes.  Members of
// groups larger than kSmallGroup are flagged for the radix fallback instead.
constexpr uint32_t kSmallGroup = 64;

__global__ __launch_bounds__(kThreads) void small_sort_kernel(const uint32_t *__restrict__ act_slot,
                                                              const uint32_t *__restrict__ act_grp,
                                                              const uint32_t *__restrict__ lo,
                                                              const uint32_t *__restrict__ vals, uint32_t m,
                                                              uint32_t *__restrict__ out_lo,
                                                              uint32_t *__restrict__ out_vals,
                                                              uint32_t *__restrict__ large_flag) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x; a < m; a += stride) {
        const uint32_t g = act_grp[a];
        const size_t g0 = a - (act_slot[a] - g);  // list index of the group's first member
        const uint32_t mine = lo[a];
        uint32_t below = 0, cnt = 0;
        // a member of a large group sees that in one look (the group's members are contiguous in the list):
        // on periodic texts every suffix sits in one of a few huge groups for twenty rounds, and counting
        // to kSmallGroup + 1 from the group's head in every round cost as much as the radix sort beside it
        bool large = g0 + kSmallGroup < m && act_grp[g0 + kSmallGroup] == g;
        for (size_t b = g0; !large && b < m; ++b) {
            if (act_grp[b] != g) break;
            if (++cnt > kSmallGroup) {
                large = true;
                break;
            }
            const uint32_t l = lo[b];
            below += (l < mine || (l == mine && b < a)) ? 1u : 0u;
        }
        large_flag[a] = large ? 1u : 0u;
        if (!large) {
            out_lo[g0 + below] = mine;
            out_vals[g0 + below] = vals[a];
        }
    }
}

// The same for groups of kSmallGroup + 1 .. kMidGroup members (collections of many similar genomes tie in groups
// as large as the collection): one workgroup per tile of kMidGroup list positions takes the groups that START in
// its tile; the keys of the tile and of the kMidGroup positions behind it sit in LDS, and every member counts the
// smaller members of its group there (the members of a group read the same entries: broadcasts).  Those groups went
// through the global radix sort beside the truly large ones: 30 ms per doubling round on 96 genomes of 2^28 bases
// in all, six times what the small groups of 24 genomes cost.  handled += members taken (one atomic per workgroup).
constexpr uint32_t kMidGroup = 1024;
constexpr int kMidThreads = 256;
__global__ __launch_bounds__(kMidThreads) void mid_sort_kernel(const uint32_t *__restrict__ act_slot,
                                                               const uint32_t *__restrict__ act_grp,
                                                               const uint32_t *__restrict__ lo,
                                                               const uint32_t *__restrict__ vals, uint32_t m,
                                                               uint32_t *__restrict__ out_lo,
                                                               uint32_t *__restrict__ out_vals,
                                                               uint32_t *__restrict__ large_flag,
                                                               uint32_t *__restrict__ handled) {
    __shared__ uint32_t s_lo[2 * kMidGroup];
    __shared__ uint32_t s_size[kMidGroup];  // members of the group that starts at this tile position (0: none)
    __shared__ uint32_t s_taken;
    const uint32_t base = blockIdx.x * kMidGroup;
    const uint32_t span = m - base < 2 * kMidGroup ? m - base : 2 * kMidGroup;
    for (uint32_t e = threadIdx.x; e < kMidGroup; e += kMidThreads) s_size[e] = 0;
    if (threadIdx.x == 0) s_taken = 0;
    __syncthreads();
    // group sizes first: a tile in which no group of the kernel's size class starts has nothing to do and leaves before it
    // reads a key (a text whose suffixes all sit in a few huge groups -- a Fibonacci word -- ran this kernel for twenty
    // rounds at 4.8 ms each)
    bool any = false;
    for (uint32_t e = threadIdx.x; e < span; e += kMidThreads) {
        const uint32_t a = base + e, g = act_grp[a], j = act_slot[a] - g;
        if ((a + 1 == m || act_grp[a + 1] != g) && j <= e && e - j < kMidGroup) {  // the last member
            s_size[e - j] = j + 1;
            any |= j + 1 > kSmallGroup && j + 1 <= kMidGroup;
        }
    }
    if (__syncthreads_or(any) == 0) return;
    for (uint32_t e = threadIdx.x; e < span; e += kMidThreads) s_lo[e] = lo[base + e];
    __syncthreads();
    uint32_t taken = 0;
    for (uint32_t e = threadIdx.x; e < span; e += kMidThreads) {
        const uint32_t a = base + e, j = act_slot[a] - act_grp[a];
        if (j > e || e - j >= kMidGroup) continue;  // my group starts in a tile in front of this one
        const uint32_t g0 = e - j, gs = s_size[g0];
        if (gs <= kSmallGroup || gs > kMidGroup) continue;  // (0: the group ends beyond the span, i.e. is larger)
        const uint32_t mine = s_lo[e];
        uint32_t below = 0;
        for (uint32_t b = g0; b < g0 + gs; ++b) {
            const uint32_t l = s_lo[b];
            below += (l < mine || (l == mine && b < e)) ? 1u : 0u;
        }
        out_lo[base + g0 + below] = mine;
        out_vals[base + g0 + below] = vals[a];
        large_flag[a] = 0;
        ++taken;
    }
    taken = wave_reduce(taken, OpAdd<uint32_t>());
    if (lane_id() == 0 && taken) atomicAdd(&s_taken, taken);
    __syncthreads();
    if (threadIdx.x == 0 && s_taken) atomicAdd(handled, s_taken);
}

// First round after the key sort: members of a group are ordered by comparing their suffixes
// DIRECTLY in the packed text (they agree on the first h0 symbols; at most `cap` symbols are
// inspected).  For sequence data nearly every group is small and its members differ within a few
// hundred symbols, so this one round finishes them -- order, new group boundaries and the LCP to
// the predecessor -- where prefix doubling would need log2(LCP / h0) gather + sort + scatter
// rounds.  Members that still agree after `cap` symbols stay grouped (out_lo = number of strictly
// smaller members is equal for them) and go on to the doubling rounds; groups larger than
// kSmallGroup are flagged for the radix path (one ordinary doubling step).

// One workgroup refines all groups that START inside its kRefineTile list positions; every member
// is a thread, all state lives in LDS.
//   * Per round each still-tied member fetches the next kRefineWords * 64 bits (128 bases) of its suffix
//     (one random window per member per round -- never a pairwise re-read; an MI355X sustains
//     ~40 G such windows/s, tools/gatherbench.hip) and parks it in LDS.
//   * The comparisons are organised by PAIR, not by member: the unordered pairs of every group are
//     listed in LDS once, each wavefront owns a stretch of that list, compares the two windows of
//     64 pairs at a time and credits the loser (one more smaller member; the longest common prefix
//     with a smaller member) with LDS atomics.  A decided pair never comes back; tied pairs are
//     compacted to the front of the stretch for the next round.  Lanes therefore stay busy whatever
//     the group sizes are -- a member-per-lane loop runs every wavefront as long as its largest
//     group (group sizes on repeat-rich DNA: mean 3, size-weighted mean 6, tail to the cap), and
//     the kernel is bound by instruction issue, not by the fetches (rocprofv3 SQ_INSTS_*).
//   * cls = number of strictly smaller members; members still tied at the end keep list order.
// A group whose pairs do not fit the list any more is left as it is (out_lo = 0): the doubling
// rounds handle it like any other unfinished group.
// 192 list positions + the 64-member span = 256 threads: four wavefronts, one per SIMD, 8 workgroups per CU.
// (Round 1 ran 256 + 64 = 320 threads: five wavefronts load the SIMDs unevenly and every workgroup stayed for
// 6.25 comparison rounds on average -- as long as its slowest member; phase clocks, NOLZSS_REFINE_PHASES:
// set-up 6.7 k, fetch 12.4 k, compare 13.7 k cycles.  128 / 192 / 256 / 320 positions: 25.5 / 24.5 / 30.8 / 30.6 ms;
// 192 with the pair list cut to 20 KiB of LDS per workgroup (8 instead of 7 per CU): 22.1 ms.)
constexpr int kRefineTile = 192;
constexpr int kRefineThreads = kRefineTile + (int)kSmallGroup;  // one thread per possible member
constexpr int kRefineWaves = kRefineThreads / 64;
constexpr int kRefineWords = 4;
constexpr int kPairCap = 1664;  // pairs per workgroup (192 members in groups of up to ~18 fit); 20 KiB of LDS: 8 workgroups = 32 waves per CU

template <int BITS, bool kTimed>
__global__ __launch_bounds__(kRefineThreads) __attribute__((amdgpu_waves_per_eu(kTimed ? 4 : 8, 8))) void group_refine_kernel(
    const uint32_t *__restrict__ act_slot, const uint32_t *__restrict__ act_grp, uint32_t *sa,
    const uint64_t *__restrict__ words, TermTable terms, uint32_t m, uint32_t h0, uint32_t cap,
    uint32_t *__restrict__ lcp, uint32_t *__restrict__ rank_by_slot, uint32_t *__restrict__ surv_slot,
    uint32_t *__restrict__ surv_head, uint32_t *__restrict__ surv_count, uint32_t *__restrict__ min_depth,
    unsigned long long *__restrict__ phases, bool no_stragglers) {
    const bool timed = kTimed && phases != nullptr && (blockIdx.x & 31) == 0 && threadIdx.x == 0;
    unsigned long long ck0 = 0, ck_fetch = 0, ck_cmp = 0, ck_rounds = 0, ck1 = 0, ck2 = 0;
    if (timed) ck0 = __builtin_readcyclecounter();
    constexpr int kW32 = 2 * kRefineWords;  // window in 32-bit words, text order
    constexpr int kChunks = kW32 / 4;
    constexpr uint32_t kPer32 = 32 / BITS;
    constexpr uint32_t kPerRound = kW32 * kPer32;
    __shared__ uint4 s_w[kChunks][kRefineThreads];  // chunk-major: a wave reads whole 16-byte rows
    __shared__ uint32_t s_pair[kPairCap];           // (higher member) | (lower member) << 16
    __shared__ uint32_t s_lim[kRefineThreads];      // symbols before the member's next terminator
    __shared__ uint32_t s_cls[kRefineThreads];      // strictly smaller members found so far
    __shared__ uint32_t s_best[kRefineThreads];     // longest common prefix with a smaller member
    __shared__ uint32_t s_goff[kRefineThreads];     // [first member of a group] first pair of the group
    __shared__ uint16_t s_term[kRefineThreads];     // index of the member's next terminator
    __shared__ uint8_t s_tied[2][kRefineThreads];   // member takes part in a tied pair (ping-pong)
    __shared__ uint32_t s_wtot[kRefineWaves];
    __shared__ uint32_t s_npairs;
    const size_t a0 = (size_t)blockIdx.x * kRefineTile;
    const size_t a1 = (a0 + kRefineTile < m) ? a0 + kRefineTile : m;
    const int t = threadIdx.x;
    const int lane = lane_id();
    const int w = t >> 6;
    const size_t a = a0 + t;

    // ---- who is here: members of groups that start in this tile -------------------------------
    // The size of a group is found in LDS: its last member (the next list element belongs to another
    // group) sits inside the span whenever the group has at most kSmallGroup members.
    uint32_t my_pos = 0, my_lim = 0, my_term = 0;
    int my_gl = 0, my_gs = 0, my_j = 0;  // first member (local), group size (0: not mine), my index
    bool starts_here = false;
    uint32_t j = 0, my_head = 0;  // my index in the group, slot of the group's first member
    bool last = false;
    bool stays = false;  // a member of a group this round leaves as it is, reported by this workgroup
    s_goff[t] = 0;  // doubles as the group size table until the pair offsets are written
    if (a < m) {
        const uint32_t g = act_grp[a];
        const uint32_t slot = act_slot[a];
        my_head = g;
        last = a + 1 == m || act_grp[a + 1] != g;
        my_pos = sa[slot];
        j = slot - g;  // my index inside the group
        const size_t g0 = a - j;
        starts_here = g0 >= a0 && g0 < a1;
        my_gl = starts_here ? (int)(g0 - a0) : 0;
    }
    __syncthreads();
    if (starts_here && last) s_goff[my_gl] = j + 1;
    __syncthreads();
    if (a < m) {
        const uint32_t sz = starts_here ? s_goff[my_gl] : 0u;  // 0: the group ends beyond the span
        const bool large = starts_here ? (sz == 0 || sz > kSmallGroup) : false;
        // too large for this round: stays one group, in place.  Its first kSmallGroup members are
        // written by the tile it starts in, the others by the tile that owns their list position.
        stays = (large && j < kSmallGroup) || (a < a1 && j >= kSmallGroup);  // (sa keeps its order)
        // the symbols every group that stays tied is known to agree on: h0 for the groups this round does
        // not touch, the depth reached for the others (the doubling rounds start from the minimum)
        if (large && j == 0) lower_min(min_depth + 1, h0);  // ([1]: groups this round does not touch)
        if (starts_here && !large) {
            my_gs = (int)sz;
            my_j = (int)j;
            // (one segment: no table look-up, and above all no load between my position and my first window)
            my_lim = term_limit(terms, my_pos, my_term);
        }
    }
    __syncthreads();  // everybody has read the sizes
    // member j of a group lists its pairs with members 0 .. j-1: the list position is an exclusive
    // scan of j over the tile
    uint32_t inc = wave_scan_inclusive_dpp((uint32_t)my_j, 0u, OpAdd<uint32_t>());
    if (lane == 63) s_wtot[w] = inc;
    if (t == 0) s_npairs = 0;
    s_lim[t] = my_lim;
    s_term[t] = (uint16_t)my_term;
    s_cls[t] = 0;
    s_best[t] = 0;
    __syncthreads();
    uint32_t my_off = inc - (uint32_t)my_j, all_pairs = 0;
#pragma unroll
    for (int k = 0; k < kRefineWaves; ++k) {
        if (k < w) my_off += s_wtot[k];
        all_pairs += s_wtot[k];
    }
    bool handled = my_gs != 0;
    uint32_t npairs = all_pairs;
    if (all_pairs > (uint32_t)kPairCap) {  // (rare, workgroup-uniform) not every group fits the list:
        // the handled groups are a prefix of the tile's groups
        if (my_gs && my_j == 0) s_goff[t] = my_off;
        __syncthreads();
        uint32_t gend = 0;
        if (my_gs) {
            gend = s_goff[my_gl] + (uint32_t)(my_gs * (my_gs - 1) / 2);
            handled = gend <= (uint32_t)kPairCap;
            if (handled && my_j == my_gs - 1) atomicMax(&s_npairs, gend);
            if (!handled) stays = true;  // no room for its pairs: the group stays as it is
            if (!handled && my_j == 0) lower_min(min_depth + 1, h0);
        }
        __syncthreads();
        npairs = s_npairs;
    }
    if (handled)
        for (int y = 0; y < my_j; ++y) s_pair[my_off + y] = (uint32_t)t | ((uint32_t)(my_gl + y) << 16);
    s_tied[0][t] = handled ? 1 : 0;
    __syncthreads();
    if (timed) ck1 = __builtin_readcyclecounter();
    // each wavefront owns a stretch of the pair list
    const uint32_t seg = ((npairs + kRefineWaves - 1) / kRefineWaves + 63u) & ~63u;
    const uint32_t seg0 = (uint32_t)w * seg;
    uint32_t cnt = seg0 < npairs ? (npairs - seg0 < seg ? npairs - seg0 : seg) : 0u;
    const uint64_t lt = lanemask_lt();

    int cur = 0;
    uint32_t depth = h0;
    constexpr int kStragglers = 64, kStragWindows = kRefineThreads / kStragglers;
    uint32_t strag_from = 0xffffffffu;
    if (npairs > 0) {
        for (uint32_t h = h0; h < cap; h += kPerRound) {
            if (s_tied[cur][t]) {  // the next kRefineWords words of my suffix, from symbol h
                const uint64_t bit = ((uint64_t)my_pos + h) * BITS;
                const uint64_t *src = words + (bit >> 6);
                uint32_t r[kW32 + 2];  // text order: high half of each 64-bit word first
#pragma unroll
                for (int k = 0; k <= kRefineWords; ++k) {
                    const uint64_t v = src[k];
                    r[2 * k] = (uint32_t)(v >> 32);
                    r[2 * k + 1] = (uint32_t)v;
                }
                // bit-select instead of ?: -- the compiler turns the conditional form into a
                // scratch array with a dynamic offset
                const uint32_t skip = (bit & 32) ? 0xffffffffu : 0u;
                const uint32_t o = (uint32_t)bit & 31;
                uint32_t q[kW32 + 1], win[kW32];
#pragma unroll
                for (int k = 0; k <= kW32; ++k) q[k] = (r[k + 1] & skip) | (r[k] & ~skip);
#pragma unroll
                for (int k = 0; k < kW32; ++k) win[k] = o ? __builtin_amdgcn_alignbit(q[k], q[k + 1], 32 - o) : q[k];
#pragma unroll
                for (int c = 0; c < kChunks; ++c)
                    s_w[c][t] = make_uint4(win[4 * c], win[4 * c + 1], win[4 * c + 2], win[4 * c + 3]);
            }
            s_tied[cur ^ 1][t] = 0;
            unsigned long long ca = timed ? __builtin_readcyclecounter() : 0;
            __syncthreads();
            unsigned long long cb = timed ? __builtin_readcyclecounter() : 0;

            uint32_t kept = 0;
            bool any_tie = false;
            for (uint32_t c0 = 0; c0 < cnt; c0 += 64) {
                const bool have = c0 + lane < cnt;
                const uint32_t item = have ? s_pair[seg0 + c0 + lane] : 0u;
                const int x = (int)(item & 0xffffu), u = (int)(item >> 16);  // x > u in list order
                bool tie = false;
                if (have) {
                    const uint32_t rem_x = s_lim[x] - h, rem_u = s_lim[u] - h;  // symbols before the terminators
                    uint32_t valid = rem_x < rem_u ? rem_x : rem_u;
                    valid = valid < kPerRound ? valid : kPerRound;
                    // both windows in one go: the compare is bound by LDS round trips, not LDS bytes
                    uint4 p[kChunks], y[kChunks];
#pragma unroll
                    for (int c = 0; c < kChunks; ++c) {
                        p[c] = s_w[c][x];
                        y[c] = s_w[c][u];
                    }
                    uint32_t xd = 0, yd = 0, wi = (uint32_t)kW32;  // the first differing word and its index
#pragma unroll
                    for (int c = kChunks - 1; c >= 0; --c) {
                        const uint32_t px[4] = {p[c].x, p[c].y, p[c].z, p[c].w};
                        const uint32_t yx[4] = {y[c].x, y[c].y, y[c].z, y[c].w};
#pragma unroll
                        for (int i = 3; i >= 0; --i) {
                            const bool diff = px[i] != yx[i];
                            xd = diff ? px[i] : xd;
                            yd = diff ? yx[i] : yd;
                            wi = diff ? (uint32_t)(4 * c + i) : wi;
                        }
                    }
                    uint32_t d = wi == (uint32_t)kW32 ? kPerRound : wi * kPer32 + (uint32_t)__clz((int)(xd ^ yd)) / BITS;
                    bool u_smaller = yd < xd;
                    if (d >= valid) {
                        if (valid == kPerRound) {  // equal windows, both suffixes go on
                            tie = true;
                        } else {  // a terminator is reached: nearer one first, then lower index
                            d = valid;
                            u_smaller = rem_u != rem_x ? rem_u < rem_x : s_term[u] < s_term[x];
                        }
                    }
                    if (!tie) {
                        const int loser = u_smaller ? x : u;  // the greater suffix
                        atomicAdd(&s_cls[loser], 1u);
                        atomicMax(&s_best[loser], h + d);  // deeper rounds only find longer prefixes
                    } else {
                        s_tied[cur ^ 1][x] = 1;
                        s_tied[cur ^ 1][u] = 1;
                    }
                }
                const uint64_t bal = __ballot(tie);  // tied pairs move to the front of the stretch
                if (tie) s_pair[seg0 + kept + (uint32_t)__popcll(bal & lt)] = item;
                kept += (uint32_t)__popcll(bal);
                any_tie |= tie;
            }
            cnt = kept;
            cur ^= 1;
            depth = h + kPerRound;  // pairs that are still tied agree on a whole window more
            // A workgroup that is still mostly tied after two windows sits on a long exact repeat:
            // comparing on to the cap would cost a window fetch per member per round for nothing.
            // Leave those ties to the doubling rounds, which need only log2(LCP) steps.
            const int busy = __syncthreads_count(any_tie);
            if (timed) { const unsigned long long cc = __builtin_readcyclecounter(); ck_fetch += cb - (ck2 ? ck2 : ck1); ck_cmp += cc - cb; ck2 = cc; ck_rounds += 1; (void)ca; }
            if (busy == 0 || (h >= h0 + kPerRound && busy > kRefineThreads / 4)) break;
            if (!no_stragglers && busy <= kStragglers / 2) {  // few tied pairs left: the rounds below
                strag_from = h + kPerRound;
                break;
            }
        }
    }


    // STRAGGLERS.  A workgroup stays as long as its deepest tie: after the first rounds a handful of members
    // is left, and every further round costs them a round trip to the text plus the barriers (phase clocks:
    // ~4 k cycles per round whatever the number of pairs; six rounds on average).  Once at most kStragglers
    // members are tied they are numbered, and the whole workgroup fetches for them: wavefront q takes window q
    // of every straggler, so ONE round trip brings kStragWindows windows each, compared in LDS one after the
    // other.  (s_tied[cur] holds the straggler's number + 1, s_goff its text position: no LDS is added.  The
    // loop is kept apart from the one above: woven into it, the common rounds ran 12-20 % slower.)
    static_assert(kStragWindows * kStragglers == kRefineThreads, "one fetching thread per straggler and window");
    if (strag_from < cap) {  // (workgroup-uniform)
        uint4 *s_flat = &s_w[0][0];
        for (uint32_t h = strag_from; h < cap;) {
            uint32_t nq = (cap - h + kPerRound - 1) / kPerRound;
            nq = nq < (uint32_t)kStragWindows ? nq : (uint32_t)kStragWindows;
            const bool tied = s_tied[cur][t] != 0;
            const uint64_t tb = __ballot(tied);
            if (lane == 0) s_wtot[w] = (uint32_t)__popcll(tb);
            __syncthreads();
            uint32_t sidx = (uint32_t)__popcll(tb & lt), ntied = 0;
#pragma unroll
            for (int k = 0; k < kRefineWaves; ++k) {
                if (k < w) sidx += s_wtot[k];
                ntied += s_wtot[k];
            }
            // (more members than fit -- a wavefront held several tied pairs per lane: their ties stay for the
            // doubling rounds, like ties at the cap)
            if (ntied > (uint32_t)kStragglers) break;
            if (tied) {
                s_tied[cur][t] = (uint8_t)(sidx + 1);
                s_goff[sidx] = my_pos;
            }
            __syncthreads();
            {
                const uint32_t q = (uint32_t)t / kStragglers, i = (uint32_t)t % kStragglers;
                // (a window behind the end of the text is never compared: its pair is decided where the
                // shorter suffix ends; the packed text is padded for windows that START inside it)
                if (i < ntied && q < nq && (uint64_t)s_goff[i] + h + (uint64_t)q * kPerRound <= (uint64_t)terms.end) {
                    const uint64_t bit = ((uint64_t)s_goff[i] + h + (uint64_t)q * kPerRound) * BITS;
                    const uint64_t *src = words + (bit >> 6);
                    uint32_t r[kW32 + 2];
#pragma unroll
                    for (int k = 0; k <= kRefineWords; ++k) {
                        const uint64_t v = src[k];
                        r[2 * k] = (uint32_t)(v >> 32);
                        r[2 * k + 1] = (uint32_t)v;
                    }
                    const uint32_t skip = (bit & 32) ? 0xffffffffu : 0u;
                    const uint32_t o = (uint32_t)bit & 31;
                    uint32_t qq[kW32 + 1], win[kW32];
#pragma unroll
                    for (int k = 0; k <= kW32; ++k) qq[k] = (r[k + 1] & skip) | (r[k] & ~skip);
#pragma unroll
                    for (int k = 0; k < kW32; ++k) win[k] = o ? __builtin_amdgcn_alignbit(qq[k], qq[k + 1], 32 - o) : qq[k];
#pragma unroll
                    for (int c = 0; c < kChunks; ++c)
                        s_flat[((size_t)q * kChunks + c) * kStragglers + i] = make_uint4(win[4 * c], win[4 * c + 1], win[4 * c + 2], win[4 * c + 3]);
                }
            }
            s_tied[cur ^ 1][t] = 0;
            __syncthreads();
            uint32_t kept = 0;
            bool any_tie = false;
            for (uint32_t c0 = 0; c0 < cnt; c0 += 64) {
                const bool have = c0 + lane < cnt;
                const uint32_t item = have ? s_pair[seg0 + c0 + lane] : 0u;
                const int x = (int)(item & 0xffffu), u = (int)(item >> 16);
                bool tie = have;
                if (have) {
                    const uint32_t sx = (uint32_t)s_tied[cur][x] - 1u, su = (uint32_t)s_tied[cur][u] - 1u;
                    uint32_t hq = h;
#pragma unroll 1
                    for (uint32_t q = 0; q < nq; ++q, hq += kPerRound) {
                        const uint32_t rem_x = s_lim[x] - hq, rem_u = s_lim[u] - hq;
                        uint32_t valid = rem_x < rem_u ? rem_x : rem_u;
                        valid = valid < kPerRound ? valid : kPerRound;
                        uint4 p[kChunks], y[kChunks];
#pragma unroll
                        for (int c = 0; c < kChunks; ++c) {
                            p[c] = s_flat[((size_t)q * kChunks + c) * kStragglers + sx];
                            y[c] = s_flat[((size_t)q * kChunks + c) * kStragglers + su];
                        }
                        uint32_t xd = 0, yd = 0, wi = (uint32_t)kW32;
#pragma unroll
                        for (int c = kChunks - 1; c >= 0; --c) {
                            const uint32_t px[4] = {p[c].x, p[c].y, p[c].z, p[c].w};
                            const uint32_t yx[4] = {y[c].x, y[c].y, y[c].z, y[c].w};
#pragma unroll
                            for (int i = 3; i >= 0; --i) {
                                const bool diff = px[i] != yx[i];
                                xd = diff ? px[i] : xd;
                                yd = diff ? yx[i] : yd;
                                wi = diff ? (uint32_t)(4 * c + i) : wi;
                            }
                        }
                        uint32_t d = wi == (uint32_t)kW32 ? kPerRound : wi * kPer32 + (uint32_t)__clz((int)(xd ^ yd)) / BITS;
                        bool u_smaller = yd < xd;
                        if (d >= valid) {
                            if (valid == kPerRound) continue;  // equal windows: on to the next one
                            d = valid;  // a terminator is reached: nearer one first, then lower index
                            u_smaller = rem_u != rem_x ? rem_u < rem_x : s_term[u] < s_term[x];
                        }
                        const int loser = u_smaller ? x : u;
                        atomicAdd(&s_cls[loser], 1u);
                        atomicMax(&s_best[loser], hq + d);
                        tie = false;
                        break;
                    }
                    if (tie) {
                        s_tied[cur ^ 1][x] = 1;
                        s_tied[cur ^ 1][u] = 1;
                    }
                }
                const uint64_t bal = __ballot(tie);
                if (tie) s_pair[seg0 + kept + (uint32_t)__popcll(bal & lt)] = item;
                kept += (uint32_t)__popcll(bal);
                any_tie |= tie;
            }
            cnt = kept;
            cur ^= 1;
            depth = h + nq * kPerRound;
            if (timed) ck_rounds += 1;
            if (__syncthreads_count(any_tie) == 0) break;
            h += nq * kPerRound;
        }
    }

    const unsigned long long ck3 = timed ? __builtin_readcyclecounter() : 0;
    if (cnt > 0 && lane == 0) lower_min(min_depth, depth);
    // ---- members still tied keep their list order: count the tied partners in front of me --------
    // (s_goff is free now; s_tied[0] marks the members of pairs that are still tied)
    s_goff[t] = 0;
    s_tied[0][t] = 0;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < cnt; c0 += 64)
        if (c0 + lane < cnt) {
            const uint32_t item = s_pair[seg0 + c0 + lane];
            atomicAdd(&s_goff[item & 0xffffu], 1u);
            s_tied[0][item & 0xffffu] = 1;
            s_tied[0][item >> 16] = 1;
        }
    // What used to be a pass of its own over the whole list (regroup_kernel: 4.9 ms at 2^30 bases) happens here:
    // the LCP of every boundary that appeared goes straight to its slot, and the members that stay tied --
    // a percent of the list on sequence data -- are collected IN SLOT ORDER: parked at their new position inside
    // the workgroup's 256 list positions, compacted, and written to the workgroup's own region; a small kernel
    // concatenates the regions (compact_survivors_kernel).  (The window buffer is free: it holds the parking lot.)
    uint32_t *s_sv_slot = reinterpret_cast<uint32_t *>(&s_w[0][0]);
    uint32_t *s_sv_head = s_sv_slot + kRefineThreads;
    static_assert(sizeof(s_w) >= 2 * kRefineThreads * sizeof(uint32_t), "the parking lot fits the window buffer");
    s_sv_slot[t] = 0xffffffffu;
    __syncthreads();
    if (handled) {
        const uint32_t cls = s_cls[t], ties_before = s_goff[t];
        const uint32_t head = my_head + cls, slot = head + ties_before;
        // the new order goes straight into the suffix array: my group occupies the slots from my_head
        // on, in list order (only members of the group, all threads of this workgroup, ever read or
        // write those slots, and every read happened before the barriers above)
        sa[slot] = my_pos;
        // LCP to the predecessor in the new order: the closest smaller member shares the longest prefix
        // (the first member of the group keeps the entry it has; a tied predecessor: no boundary, stays pending)
        if (ties_before == 0 && cls > 0) lcp[slot] = s_best[t];
        // a boundary that stays undecided INSIDE a class this round compared: its own pending code, so that the
        // groups the round did not touch (code kLcpPending) can be told from it (the equalising round, below)
        if (ties_before > 0) lcp[slot] = kLcpPendingCompared;
        if (rank_by_slot) rank_by_slot[slot] = head + 1u;
        if (s_tied[0][t]) {
            const uint32_t nl = (uint32_t)my_gl + cls + ties_before;
            s_sv_slot[nl] = slot;
            s_sv_head[nl] = head;
        }
    } else if (stays) {
        const uint32_t slot = my_head + j;
        if (rank_by_slot) rank_by_slot[slot] = my_head + 1u;
        s_sv_slot[t] = slot;
        s_sv_head[t] = my_head;
    }
    __syncthreads();
    {
        const uint32_t sv = s_sv_slot[t], hd = s_sv_head[t];
        const bool keep = sv != 0xffffffffu;
        const uint64_t kb = __ballot(keep);
        if (lane == 0) s_wtot[w] = (uint32_t)__popcll(kb);
        __syncthreads();
        uint32_t at = (uint32_t)__popcll(kb & lt), total = 0;
#pragma unroll
        for (int k = 0; k < kRefineWaves; ++k) {
            if (k < w) at += s_wtot[k];
            total += s_wtot[k];
        }
        if (keep) {
            surv_slot[(size_t)blockIdx.x * kRefineThreads + at] = sv;
            surv_head[(size_t)blockIdx.x * kRefineThreads + at] = hd;
        }
        if (t == 0) surv_count[blockIdx.x] = total;
    }
    // how much of what stays tied was not compared at all: counted in every 64th workgroup (an estimate for the host's
    // choice of what runs next; one atomic per counting workgroup)
    if ((blockIdx.x & 63u) == 0) {
        const int untouched = __syncthreads_count(stays);
        if (t == 0 && untouched) atomicAdd(min_depth + 2, (uint32_t)untouched);
    }
    if (timed) {
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long ck4 = __builtin_readcyclecounter();
        atomicAdd(phases + 0, ck1 - ck0);
        atomicAdd(phases + 1, ck_fetch);
        atomicAdd(phases + 2, ck_cmp);
        atomicAdd(phases + 3, ck4 - ck3);
        atomicAdd(phases + 4, ck_rounds);
        atomicAdd(phases + 5, 1ull);
        atomicAdd(phases + 6, ck4 - ck0);
    }
}

// the survivors of the direct round, region by region (one region of kRefineThreads entries per workgroup of
// group_refine_kernel, `count` of them used), to the active list: four threads per region
__global__ __launch_bounds__(kThreads) void compact_survivors_kernel(const uint32_t *__restrict__ surv_slot,
                                                                     const uint32_t *__restrict__ surv_head,
                                                                     const uint32_t *__restrict__ count,
                                                                     const uint32_t *__restrict__ offset,
                                                                     uint32_t regions, uint32_t *__restrict__ new_slot,
                                                                     uint32_t *__restrict__ new_grp) {
    const size_t stride = (size_t)gridDim.x * blockDim.x / 4;
    for (size_t r = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 4; r < regions; r += stride) {
        const uint32_t c = count[r], o = offset[r];
        for (uint32_t k = threadIdx.x & 3u; k < c; k += 4) {
            new_slot[o + k] = surv_slot[r * kRefineThreads + k];
            new_grp[o + k] = surv_head[r * kRefineThreads + k];
        }
    }
}

#include "group_sort.hpp"

__global__ __launch_bounds__(kThreads) void gather_large_kernel(const uint32_t *__restrict__ large_flag,
                                                                const uint32_t *__restrict__ idx,
                                                                const uint32_t *__restrict__ act_grp,
                                                                const uint32_t *__restrict__ lo,
                                                                const uint32_t *__restrict__ vals, uint32_t m,
                                                                uint64_t *__restrict__ lkeys,
                                                                uint32_t *__restrict__ lvals,
                                                                uint32_t *__restrict__ lidx) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x; a < m; a += stride)
        if (large_flag[a]) {
            const uint32_t k = idx[a];
            lkeys[k] = ((uint64_t)act_grp[a] << 32) | lo[a];
            lvals[k] = vals[a];
            lidx[k] = (uint32_t)a;
        }
}

// The same for the segmented sort of the large groups (radix_sort_segments_u32): 32-bit keys, the group of every
// gathered element beside them; then the first gathered element of every group (heads -> scan -> starts).
__global__ __launch_bounds__(kThreads) void gather_large32_kernel(const uint32_t *__restrict__ large_flag,
                                                                  const uint32_t *__restrict__ idx,
                                                                  const uint32_t *__restrict__ act_grp,
                                                                  const uint32_t *__restrict__ lo,
                                                                  const uint32_t *__restrict__ vals, uint32_t m,
                                                                  uint32_t *__restrict__ lkeys, uint32_t *__restrict__ lvals,
                                                                  uint32_t *__restrict__ lidx, uint32_t *__restrict__ lgrp) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x; a < m; a += stride)
        if (large_flag[a]) {
            const uint32_t k = idx[a];
            lkeys[k] = lo[a];
            lvals[k] = vals[a];
            lidx[k] = (uint32_t)a;
            lgrp[k] = act_grp[a];
        }
}
__global__ __launch_bounds__(kThreads) void large_heads_kernel(const uint32_t *__restrict__ lgrp, uint32_t count,
                                                               uint32_t *__restrict__ head) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += stride)
        head[k] = (k == 0 || lgrp[k] != lgrp[k - 1]) ? 1u : 0u;
}
__global__ __launch_bounds__(kThreads) void large_starts_kernel(const uint32_t *__restrict__ head,
                                                                const uint32_t *__restrict__ pos, uint32_t count,
                                                                uint32_t *__restrict__ starts) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += stride)
        if (head[k]) starts[pos[k]] = (uint32_t)k;
}
__global__ __launch_bounds__(kThreads) void scatter_large32_kernel(const uint32_t *__restrict__ lkeys,
                                                                   const uint32_t *__restrict__ lvals,
                                                                   const uint32_t *__restrict__ lidx, uint32_t count,
                                                                   uint32_t *__restrict__ out_lo,
                                                                   uint32_t *__restrict__ out_vals) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += stride) {
        const uint32_t a = lidx[k];
        out_lo[a] = lkeys[k];
        out_vals[a] = lvals[k];
    }
}

// the k-th smallest large element goes to the k-th list position owned by a large group
__global__ __launch_bounds__(kThreads) void scatter_large_kernel(const uint64_t *__restrict__ lkeys,
                                                                 const uint32_t *__restrict__ lvals,
                                                                 const uint32_t *__restrict__ lidx, uint32_t count,
                                                                 uint32_t *__restrict__ out_lo,
                                                                 uint32_t *__restrict__ out_vals,
                                                                 uint32_t *__restrict__ lcp_list, uint32_t lcp_code) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += stride) {
        const uint32_t a = lidx[k];
        out_lo[a] = (uint32_t)lkeys[k];
        out_vals[a] = lvals[k];
        if (lcp_list) lcp_list[a] = lcp_code;
    }
}

// shared tail of every round: sorted view of m active elements -> sa / rank / next active list
template <bool kRound0>
uint32_t regroup(Context &ctx, const uint64_t *keys, const uint32_t *grp, const uint32_t *lo, uint32_t *vals,
                 const uint32_t *act_slot, uint32_t m, uint32_t n, uint32_t *sa, uint32_t *rank, uint32_t *new_slot,
                 uint32_t *new_grp, uint32_t *scratch_idx, uint32_t *scratch_val, uint32_t *rank_val,
                 uint32_t *d_total, uint32_t *lcp = nullptr, int sym_bits = 0, int tag_bits = 0, int bits = 0,
                 const uint32_t *lcp_list = nullptr, int low_bits = 0, uint32_t dbl_h = 0,
                 uint32_t *rank_by_slot = nullptr, const Pyramid *plcp = nullptr, const uint32_t *keys32 = nullptr,
                 const SegView *seg = nullptr, uint32_t short_tag = 0, bool sa_is_current = false,
                 uint32_t seq_shift = 0, bool store_ranks = true) {
    hipStream_t s = ctx.stream;
    const size_t pmark = ctx.arena.mark();
    // doubling boundaries read range minima of the LCP values decided so far
    const Pyramid Plcp = plcp ? *plcp : Pyramid{};
    const size_t tiles = seg ? (size_t)seg->num_tiles * (kSortTile / kFuseTile) : div_up(m, kFuseTile);
    uint32_t total[2] = {0, 0};
    {
        const double bytes = kRound0 ? 20.0 * m : 28.0 * m;  // view (+ vals) in, (sa +) rank + lcp out (+ survivors)
        ProfScope ps(ctx.profiler(), "sa_regroup", s, bytes);
        // descriptors of both scans, then [ticket, error flag]
        uint64_t *desc = ctx.arena.alloc<uint64_t>(2 * tiles + 1);
        HIP_CHECK(hipMemsetAsync(desc, 0, (2 * tiles + 1) * sizeof(uint64_t), s));
        RegroupArgs A{};
        A.short_tag = short_tag;
        A.seq_shift = seq_shift;
        A.sa_is_current = sa_is_current ? 1 : 0;
        A.keys = keys; A.keys32 = keys32; A.seg = seg ? *seg : SegView{}; A.num_tiles = (uint32_t)tiles; A.grp = grp; A.lo = lo; A.vals = vals; A.act_slot = act_slot; A.m = m;
        A.sa = sa; A.rank_val = rank_val; A.rank_by_slot = rank_by_slot; A.store_ranks = store_ranks ? 1 : 0; A.lcp = lcp;
        A.chg_idx = scratch_idx; A.chg_count = d_total + 2;
        HIP_CHECK(hipMemsetAsync(d_total + 2, 0, sizeof(uint32_t), s));
        A.sym_bits = sym_bits; A.tag_bits = tag_bits; A.bits = bits; A.low_bits = low_bits;
        A.bits_shift = bits == 2 ? 1 : (bits == 4 ? 2 : 3);
        A.lcp_list = lcp_list; A.dbl_h = dbl_h; A.Plcp = Plcp;
        A.new_slot = new_slot; A.new_grp = new_grp;
        A.desc_max = desc; A.desc_sum = desc + tiles;
        A.packed = n < 0x80000000u ? 1 : 0;  // slots and counts fit 31 bits
        A.ticket = reinterpret_cast<uint32_t *>(desc + 2 * tiles);
        A.d_total = d_total;
        static const bool want_phases = getenv("NOLZSS_REGROUP_PHASES") != nullptr;
        if (want_phases) {
            A.phases = ctx.arena.alloc<unsigned long long>(8);
            HIP_CHECK(hipMemsetAsync(A.phases, 0, 64, s));
        }
        const bool fast_layout = kRound0 && keys32 && seg && low_bits == 0 && tag_bits == KeyLayout<2>::kTagBits &&
                                 sym_bits == 2 * KeyLayout<2>::kSyms && bits == 2 && seq_shift == 0 &&
                                 short_tag == (uint32_t)KeyLayout<2>::kSyms;
        const bool rec_layout = kRound0 && keys32 && seg && low_bits == 0 && tag_bits == kRecTagBits &&
                                sym_bits == 2 * kRecSyms && bits == 2 && seq_shift == 32 && short_tag == 0;
        const bool p16_layout = kRound0 && keys32 && seg && low_bits == 0 && tag_bits == kP16TagBits &&
                                sym_bits == 2 * kP16Syms && bits == 2 && seq_shift == 0 && short_tag == (uint32_t)kP16Syms;
        if (fast_layout)
            regroup_kernel<kRound0, kRound0 ? 1 : 0><<<(unsigned)tiles, kFuseThreads, 0, s>>>(A);  // (layouts only exist for round 0)
        else if (p16_layout)
            regroup_kernel<kRound0, kRound0 ? 3 : 0><<<(unsigned)tiles, kFuseThreads, 0, s>>>(A);
        else if (rec_layout)
            regroup_kernel<kRound0, kRound0 ? 2 : 0><<<(unsigned)tiles, kFuseThreads, 0, s>>>(A);
        else
            regroup_kernel<kRound0, 0><<<(unsigned)tiles, kFuseThreads, 0, s>>>(A);
        KERNEL_CHECK();
        if (want_phases) {
            unsigned long long h[8];
            HIP_CHECK(hipMemcpyAsync(h, A.phases, 64, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
            const double wn = h[4] ? (double)h[4] : 1.0;
            fprintf(stderr, "[nolzss] regroup<%d> m=%u tiles=%zu phases (cycles per tile, %llu sampled): ticket %.0f  loads+heads %.0f  look-back %.0f  output %.0f\n",
                    (int)kRound0, m, tiles, h[4], h[0] / wn, h[1] / wn, h[2] / wn, h[3] / wn);
        }
        HIP_CHECK(hipMemcpyAsync(d_total + 1, A.ticket + 1, sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    }
    ctx.arena.rewind(pmark);
    uint32_t total3[3] = {0, 0, 0};
    ctx.read_back(d_total, total3, 3);
    total[0] = total3[0];
    total[1] = total3[1];
    if (total[1]) throw HipError("suffix array: look-back scan timed out");
    if (!rank_by_slot && total3[2] > 0) {
        // rank[start] = new rank for the elements whose rank changed: the one truly random write of the round
        ProfScope ps(ctx.profiler(), "sa_rank_scatter", s);
        uint32_t *idx[2] = {scratch_idx, vals};
        uint32_t *val[2] = {rank_val, scratch_val};
        bucketed_scatter(idx, val, total3[2], rank, n, ctx.arena, s, ctx.profiler(), false);
    }
    return total[0];
}

}  // namespace

void finish_packing(Context &ctx, PackedText &t, const uint8_t *d_text, size_t n,
                    const std::vector<uint32_t> &terminators, const unsigned long long *presence);

PackedText pack_text(Context &ctx, const uint8_t *d_text, size_t n) {
    hipStream_t s = ctx.stream;
    PackedText t;
    t.n = (uint32_t)n;
    unsigned long long *presence = ctx.arena.alloc<unsigned long long>(4);
    HIP_CHECK(hipMemsetAsync(presence, 0, 32, s));
    {
        ProfScope ps(ctx.profiler(), "text_presence", s);
        presence_kernel<<<grid_for(div_up(n, 16), kThreads, 2048), kThreads, 0, s>>>(d_text, n, presence);
        KERNEL_CHECK();
    }
    uint32_t bitsw[8];
    ctx.read_back(reinterpret_cast<const uint32_t *>(presence), bitsw, 8);
    int sigma = 0;
    for (int k = 0; k < 8; ++k) sigma += __builtin_popcount(bitsw[k]);
    t.sigma = sigma;
    t.bits = sigma <= 4 ? 2 : (sigma <= 16 ? 4 : 8);

    // Segmented text?  Upper-case nucleotides plus at most 250 other byte values that occur
    // exactly ONCE each (the shape of the reference's prepared multi-sequence / reverse-
    // complement strings, and of reference + '\\x01' + target): a byte that occurs once can match
    // nothing, so it only terminates matches and the text packs at 2 bits per base.
    std::vector<uint32_t> terminators;
    {
        uint32_t acgt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (unsigned char c : {'A', 'C', 'G', 'T'}) acgt[c >> 5] |= 1u << (c & 31);
        int others = 0, nucleotides = 0;
        for (int k = 0; k < 8; ++k) {
            others += __builtin_popcount(bitsw[k] & ~acgt[k]);
            nucleotides += __builtin_popcount(bitsw[k] & acgt[k]);
        }
        if (others >= 1 && others <= 250 && nucleotides >= 1) {
            uint32_t *count = ctx.arena.alloc<uint32_t>(1);
            uint32_t *pos = ctx.arena.alloc<uint32_t>(kMaxTermScan);
            HIP_CHECK(hipMemsetAsync(count, 0, sizeof(uint32_t), s));
            size_t g = div_up(n, kThreads);
            if (g > 8192) g = 8192;
            find_terminators_kernel<true><<<(unsigned)g, kThreads, 0, s>>>(d_text, (uint32_t)n, count, pos);
            KERNEL_CHECK();
            uint32_t h_count = 0;
            ctx.read_back(count, &h_count, 1);
            if (h_count == (uint32_t)others) {  // every non-nucleotide byte value occurs exactly once
                HIP_CHECK(hipMemsetAsync(count, 0, sizeof(uint32_t), s));
                find_terminators_kernel<false><<<(unsigned)g, kThreads, 0, s>>>(d_text, (uint32_t)n, count, pos);
                KERNEL_CHECK();
                HIP_CHECK(hipStreamSynchronize(s));
                terminators.resize(h_count);
                HIP_CHECK(hipMemcpy(terminators.data(), pos, h_count * sizeof(uint32_t), hipMemcpyDeviceToHost));
                std::sort(terminators.begin(), terminators.end());
                unsigned long long h_presence[4] = {0, 0, 0, 0};
                for (unsigned char c : {'A', 'C', 'G', 'T'}) h_presence[c >> 6] |= 1ull << (c & 63);
                HIP_CHECK(hipMemcpy(presence, h_presence, 32, hipMemcpyHostToDevice));
                t.sigma = 4;
                t.bits = 2;
                t.segmented = true;
            }
        }
    }
    finish_packing(ctx, t, d_text, n, terminators, presence);
    return t;
}

namespace {
// coarse index of a long terminator table (text.hpp): one binary search per 4096-symbol block
__global__ __launch_bounds__(kThreads) void term_coarse_kernel(const uint32_t *__restrict__ pos, uint32_t count,
                                                               uint32_t blocks, uint32_t *__restrict__ coarse) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= blocks) return;
    const uint64_t p = (uint64_t)b << kTermBlockShift;
    uint32_t lo = 0, hi = count - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((uint64_t)pos[mid] >= p)
            hi = mid;
        else
            lo = mid + 1;
    }
    coarse[b] = lo;
}
}  // namespace

// terminator table (the given sorted positions and always the end of the text), then the packed words
void finish_packing(Context &ctx, PackedText &t, const uint8_t *d_text, size_t n,
                    const std::vector<uint32_t> &terminators, const unsigned long long *presence) {
    hipStream_t s = ctx.stream;
    std::vector<uint32_t> table = terminators;
    table.push_back((uint32_t)n);
    uint32_t *d_terms = ctx.arena.alloc<uint32_t>(table.size());
    HIP_CHECK(hipMemcpyAsync(d_terms, table.data(), table.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));  // table is a local vector
    t.terms.pos = d_terms;
    t.terms.count = (uint32_t)table.size();
    t.terms.end = (uint32_t)n;
    if (table.size() <= kTermFew) {  // (a short table travels in the kernel arguments too, text.hpp)
        t.terms.nfew = (uint32_t)table.size();
        for (size_t k = 0; k < table.size(); ++k) t.terms.few[k] = table[k];
    }
    if (table.size() > 256) {
        const uint32_t blocks = (uint32_t)(n >> kTermBlockShift) + 3;
        uint32_t *coarse = ctx.arena.alloc<uint32_t>(blocks);
        term_coarse_kernel<<<(unsigned)div_up(blocks, kThreads), kThreads, 0, s>>>(d_terms, t.terms.count, blocks, coarse);
        KERNEL_CHECK();
        t.terms.coarse = coarse;
    }

    const size_t nwords = div_up(n * (size_t)t.bits, 64) + kRefineWords + 4;  // zero pad: windows read past the end
    uint64_t *words = ctx.arena.alloc<uint64_t>(nwords);
    {
        ProfScope ps(ctx.profiler(), "text_pack", s);
        const unsigned g = grid_for(nwords, kThreads);
        switch (t.bits) {
        case 2: pack_kernel<2><<<g, kThreads, 0, s>>>(d_text, n, presence, words, nwords); break;
        case 4: pack_kernel<4><<<g, kThreads, 0, s>>>(d_text, n, presence, words, nwords); break;
        default: pack_kernel<8><<<g, kThreads, 0, s>>>(d_text, n, presence, words, nwords); break;
        }
        KERNEL_CHECK();
    }
    t.words = words;
}

// The merged per-sequence batch: d_text holds upper-case nucleotide records with ONE separator byte
// (any byte that is not a nucleotide) at each of the given sorted positions.  Returns false -- and
// packs nothing -- if the text holds anything else (the caller then takes the records one by one).
bool pack_independent_text(Context &ctx, const uint8_t *d_text, size_t n, const std::vector<uint32_t> &separators,
                           PackedText &t, bool mirror) {
    hipStream_t s = ctx.stream;
    uint32_t *count = ctx.arena.alloc<uint32_t>(1);
    uint32_t *pos = ctx.arena.alloc<uint32_t>(kMaxTermScan);
    HIP_CHECK(hipMemsetAsync(count, 0, sizeof(uint32_t), s));
    size_t g = div_up(n, kThreads);
    if (g > 8192) g = 8192;
    find_terminators_kernel<false><<<(unsigned)g, kThreads, 0, s>>>(d_text, (uint32_t)n, count, pos);
    KERNEL_CHECK();
    uint32_t h_count = 0;
    ctx.read_back(count, &h_count, 1);
    // the separators are not nucleotides, so an equal count means: nothing else is there
    if (h_count != (uint32_t)separators.size()) return false;
    unsigned long long h_presence[4] = {0, 0, 0, 0};
    for (unsigned char c : {'A', 'C', 'G', 'T'}) h_presence[c >> 6] |= 1ull << (c & 63);
    unsigned long long *presence = ctx.arena.alloc<unsigned long long>(4);
    HIP_CHECK(hipMemcpy(presence, h_presence, 32, hipMemcpyHostToDevice));
    t = PackedText{};
    t.n = (uint32_t)n;
    t.sigma = 4;
    t.bits = 2;
    t.segmented = true;
    finish_packing(ctx, t, d_text, n, separators, presence);
    if (!separators.empty()) t.terms.seq_shift = kIndKeyBits;
    t.terms.mirror = mirror ? 1u : 0u;
    return true;
}

// finishes the LCP entries that round 0 could not decide: both suffixes share their first
// `skip` symbols, so the packed-word comparison starts there
template <int BITS>
__global__ __launch_bounds__(kThreads) void lcp_finish_kernel(const uint64_t *__restrict__ words, uint32_t n,
                                                              TermTable terms, const uint32_t *__restrict__ sa,
                                                              uint32_t skip, uint32_t *__restrict__ lcp) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r <= n; r += stride) {
        if (r == n) {
            lcp[r] = 0;
        } else {
            // safety net: every boundary is decided by the keys, the direct round or a doubling
            // step; anything still pending is compared in the packed text
            if (lcp[r] >= kLcpPendingMin) lcp[r] = suffix_lcp<BITS>(words, terms, sa[r - 1], sa[r], skip);
        }
    }
}

int build_suffix_array(Context &ctx, const PackedText &text, uint32_t *sa, uint32_t *isa, uint32_t *lcp,
                       bool *isa_deferred) {
    if (isa_deferred) *isa_deferred = false;
    const uint32_t n = text.n;
    hipStream_t s = ctx.stream;
    Arena &arena = ctx.arena;
    const size_t mark = arena.mark();

    // Buffers live in scopes (the arena is a stack): what the whole construction needs first -- the two
    // active lists and the rank of every slot -- then the buffers of one phase at a time, released when the
    // phase ends.  Everything behind the direct round is sized by the number of suffixes that are still
    // tied, not by n: 32 n bytes stay for the whole construction (sa, rank, lcp from the caller, the
    // lists, rank_by_slot), the key sort adds 12.5 n (2-bit texts) and the rank scatter 12 n on top of
    // them, the doubling rounds 60 bytes per TIED suffix (96 n only when every suffix is tied, as on a
    // periodic text; a text with 5 % of its suffixes tied after the direct round peaks at 45 n).
    // (NOLZSS_DNA_FAST_MIN: smallest text that takes the bucketed sort; the tests lower it)
    static const uint32_t dna_fast_min =
        getenv("NOLZSS_DNA_FAST_MIN") ? (uint32_t)atoll(getenv("NOLZSS_DNA_FAST_MIN")) : (1u << 20);
    // 2-bit texts, segmented or not, sort on the plain 40-bit key [17 bases][6-bit tag]: suffixes that
    // meet a terminator inside the key window and tie on the key are already in their final order
    // after the (stable) sort -- ascending start = ascending terminator -- and the regroup kernel
    // makes each of them a group of its own.
    // independent sequences (merged batch): the number of the sequence sits above the plain 40-bit key
    const bool independent = text.terms.seq_shift != 0;
    int seq_bits = 0;
    if (independent) {
        if (text.bits != 2 || text.terms.seq_shift != (uint32_t)kIndKeyBits) throw HipError("suffix array: independent sequences need the 2-bit key layout");
        while (seq_bits < 24 && (1u << seq_bits) < text.terms.count) ++seq_bits;
        if ((1u << seq_bits) < text.terms.count) throw HipError("suffix array: too many independent sequences");
    }
    const bool dna_fast = text.bits == 2 && n >= dna_fast_min && !independent;
    // plain one-segment DNA: the 16-base key whose tag is not sorted (text.hpp, kP16Syms) -- one radix pass less
    // (NOLZSS_NO_KEY16: A/B switch back to the 40-bit key [17 bases][6-bit tag])
    static const bool no_key16 = getenv("NOLZSS_NO_KEY16") != nullptr;
    // (segmented texts with a short terminator table take it too: the reverse-complement string of one sequence)
    const bool key16 = dna_fast && !no_key16 && key16_applicable(text);
    // independent LONG records (text.hpp, kRecSyms): the records are the buckets of the segmented sort
    // (NOLZSS_REC_BUCKET_MIN: smallest average record that takes it; partial tiles cost 4096 / that)
    static const uint64_t rec_bucket_min =
        getenv("NOLZSS_REC_BUCKET_MIN") ? (uint64_t)atoll(getenv("NOLZSS_REC_BUCKET_MIN")) : (uint64_t(1) << 16);
    const bool rec_fast = independent && !text.terms.mirror && n >= dna_fast_min && rec_bucket_min > 0 &&
                          (uint64_t)text.terms.count * rec_bucket_min <= (uint64_t)n;
    int key_passes = 0;
    {
        int kb = key16 ? kP16Syms * 2
                 : dna_fast ? KeyLayout<2>::kSyms * 2 + KeyLayout<2>::kTagBits
                 : rec_fast ? kRecSyms * 2 + kRecTagBits
                 : independent ? kIndKeyBits + seq_bits
                 : text.segmented ? kSegSyms * 2 + kSegTagBits + kSegTermBits
                 : text.bits == 2 ? KeyLayout<2>::kSyms * 2 + KeyLayout<2>::kTagBits
                 : text.bits == 4 ? KeyLayout<4>::kSyms * 4 + KeyLayout<4>::kTagBits
                                  : KeyLayout<8>::kSyms * 8 + KeyLayout<8>::kTagBits;
        key_passes = (kb + kRadixBits - 1) / kRadixBits;
        if (key_passes > 8) key_passes = 8;
    }
    // Can the caller be left without rank[] if the direct rounds finish the suffix array (see below)?  The regroup
    // kernels then do not store the rank of every slot either (nobody would read it; if doubling rounds turn out
    // to be needed, write_all_ranks recovers it from the LCP array).
    // (the two-value permutation takes 8 n bytes more than the one it replaces: 57 n at the peak of the candidate
    // stage; an arena that settled for less keeps the rank scatter.  Texts of more than 2^30 symbols keep it too:
    // the two-value form has two partition passes.  NOLZSS_NO_DEFER_ISA: A/B switch.)
    static const bool no_defer = getenv("NOLZSS_NO_DEFER_ISA") != nullptr;
    // (a merged batch of long records: its block-diagonal permutation has the two-value form too, RecordScatterPlan)
    const bool plan_two = ctx.rec_plan && ctx.rec_plan->seg.desc && ctx.rec_plan->n == n && !text.terms.mirror;
    const bool can_defer = isa_deferred && !no_defer && (plan_two || (text.terms.seq_shift == 0 && n <= (1u << 30))) &&
                           arena.capacity() >= 60 * (size_t)n + (size_t(64) << 20);
    if (getenv("NOLZSS_TRACE") && isa_deferred && !can_defer)
        fprintf(stderr, "[nolzss]   rank[] is scattered after the direct rounds (arena %.1f of %.1f GiB, %s)\n",
                (double)arena.capacity() / 1073741824.0, (60.0 * (double)n + 67108864.0) / 1073741824.0,
                text.terms.seq_shift ? (plan_two ? "records with a plan" : "independent records without a plan") : "one text");
    const bool store_ranks = !can_defer;
    uint32_t *act_slot[2] = {arena.alloc<uint32_t>(n), arena.alloc<uint32_t>(n)};
    uint32_t *act_grp[2] = {arena.alloc<uint32_t>(n), arena.alloc<uint32_t>(n)};
    uint32_t *rank_by_slot = arena.alloc<uint32_t>(n);
    uint32_t *d_total = arena.alloc<uint32_t>(4);  // survivors, look-back error flag, elements whose rank changed
    uint32_t *seg_mem = arena.alloc<uint32_t>((size_t)kSegDescWords *
                                              (div_up(n, kSortTile) + (rec_fast ? (size_t)text.terms.count + 1 : (size_t)257)));  // 16-byte aligned
    uint32_t *rank = isa;

    // ---- phase: key sort + first regroup -----------------------------------------------------
    const size_t sort_mark = arena.mark();
    // the bucketed sort of plain DNA works on 8-byte (u32 key, u32 suffix) records: two 4n-byte key buffers;
    // the general sort on 12-byte records: two 8n-byte key buffers
    uint64_t *keys[2];
    // (NOLZSS_FUSED_SORT: the 16-base key sort on fused 64-bit records, radix_sort.hip -- A/B switch)
    static const bool fused_sort = getenv("NOLZSS_FUSED_SORT") != nullptr && atoi(getenv("NOLZSS_FUSED_SORT")) != 0;
    const bool fused = key16 && fused_sort && !text.segmented;
    if (fused) {
        keys[0] = arena.alloc<uint64_t>(n);
        keys[1] = arena.alloc<uint64_t>(n);
    } else if (dna_fast || rec_fast) {
        uint32_t *k32 = arena.alloc<uint32_t>(2 * (size_t)n + 4);
        keys[0] = reinterpret_cast<uint64_t *>(k32);
        keys[1] = reinterpret_cast<uint64_t *>(k32 + (((size_t)n + 1) & ~size_t(1)));
    } else {
        keys[0] = arena.alloc<uint64_t>(n);
        keys[1] = arena.alloc<uint64_t>(n);
    }
    // The value buffers of the key sort: the one the last pass lands in IS sa (no copy afterwards).
    uint32_t *vals_other = fused ? nullptr : arena.alloc<uint32_t>(n);
    uint32_t *vals[2] = {vals_other, vals_other};
    vals[key_passes & 1] = sa;

    // ---- round 0: order by the first K symbols -------------------------------------------
    // (the keys are never materialised in text order: the first radix pass computes them from the
    // packed text, radix_sort_initial_keys)
    int k_syms = 0;
    switch (text.bits) {
    case 2: k_syms = KeyLayout<2>::kSyms; break;
    case 4: k_syms = KeyLayout<4>::kSyms; break;
    default: k_syms = KeyLayout<8>::kSyms; break;
    }
    int cur;
    SegView seg;
    Round0Regroup round0;
    {
        // only the low key_bits of the key are populated
        int key_bits = k_syms * text.bits;
        switch (text.bits) {
        case 2: key_bits += KeyLayout<2>::kTagBits; break;
        case 4: key_bits += KeyLayout<4>::kTagBits; break;
        default: key_bits += KeyLayout<8>::kTagBits; break;
        }
        if (text.segmented) key_bits = kSegSyms * 2 + kSegTagBits + kSegTermBits;
        if (independent) key_bits = kIndKeyBits + seq_bits;
        int shifts0[8], np0 = 0;
        for (int b = 0; b < key_bits && np0 < 8; b += kRadixBits) shifts0[np0++] = b;
        ProfScope ps(ctx.profiler(), "sa_sort_initial", s);
        if (fused) {
            radix_sort_dna_keys16_fused(text, keys, sa, seg_mem, seg, arena, s, ctx.profiler());
            cur = 0;
        } else if (key16) {
            uint32_t *keys32[2] = {reinterpret_cast<uint32_t *>(keys[0]), reinterpret_cast<uint32_t *>(keys[1])};
            // (where the sort finishes its sub-buckets in LDS it does the regroup of round 0 on the way, if nobody needs
            // the ranks it would store: the sorted keys are then never written)
            round0.lcp = lcp;
            round0.new_slot = act_slot[0];
            round0.new_grp = act_grp[0];
            round0.d_total = d_total;
            radix_sort_dna_keys16(text, keys32, vals, seg_mem, seg, arena, s, ctx.profiler(), store_ranks ? nullptr : &round0);
            cur = 0;
            if (vals[cur] != sa) throw HipError("suffix array: key sort did not end in sa");
        } else if (dna_fast) {
            // plain DNA: partition by the first four bases, then sort the buckets on 8-byte records
            uint32_t *keys32[2] = {reinterpret_cast<uint32_t *>(keys[0]), reinterpret_cast<uint32_t *>(keys[1])};
            radix_sort_dna_keys(text, keys32, vals, seg_mem, seg, arena, s, ctx.profiler());
            cur = 1;
            if (vals[cur] != sa) throw HipError("suffix array: key sort did not end in sa");
        } else if (rec_fast) {
            std::vector<uint32_t> h_terms(text.terms.count);
            HIP_CHECK(hipMemcpyAsync(h_terms.data(), text.terms.pos, sizeof(uint32_t) * h_terms.size(), hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
            uint32_t *keys32[2] = {reinterpret_cast<uint32_t *>(keys[0]), reinterpret_cast<uint32_t *>(keys[1])};
            radix_sort_record_keys(text, h_terms, keys32, vals, seg_mem, seg, arena, s, ctx.profiler());
            cur = 0;
            if (vals[cur] != sa) throw HipError("suffix array: key sort did not end in sa");
        } else {
            cur = radix_sort_initial_keys(text, keys, vals, shifts0, np0, arena, s, ctx.profiler());
            if (np0 != key_passes || vals[cur] != sa) throw HipError("suffix array: key sort did not end in sa");
        }
    }
    int tag_bits = 0, low_bits = 0;
    switch (text.bits) {
    case 2: tag_bits = KeyLayout<2>::kTagBits; break;
    case 4: tag_bits = KeyLayout<4>::kTagBits; break;
    default: tag_bits = KeyLayout<8>::kTagBits; break;
    }
    if (independent) {  // [record][kIndSyms bases][4-bit tag]
        k_syms = kIndSyms;
        tag_bits = kIndTagBits;
    }
    if (rec_fast) {  // bucket = record, [kRecSyms bases][4-bit tag]
        k_syms = kRecSyms;
        tag_bits = kRecTagBits;
    }
    if (key16) {  // bucket = first four bases, stored word [24 key bits][8-bit tag]
        k_syms = kP16Syms;
        tag_bits = kP16TagBits;
    }
    const bool bucketed = dna_fast || rec_fast;
    if (text.segmented && !dna_fast && !independent) {  // [kSegSyms symbols][5-bit tag][8-bit terminator index]
        k_syms = kSegSyms;
        tag_bits = kSegTagBits;
        low_bits = kSegTermBits;
    }
    uint32_t m = 0;
    if (round0.done) {
        uint32_t total2[2] = {0, 0};
        ctx.read_back(d_total, total2, 2);
        m = total2[0];
    } else
    m = regroup<true>(ctx, keys[cur], nullptr, nullptr, vals[cur], nullptr, n, n, sa, rank, act_slot[0],
                               act_grp[0], nullptr, nullptr, nullptr, d_total, lcp,
                               k_syms * text.bits, tag_bits, text.bits, nullptr, low_bits, 0, rank_by_slot, nullptr,
                               bucketed ? reinterpret_cast<const uint32_t *>(keys[cur]) : nullptr,
                               bucketed ? &seg : nullptr,
                               // (mirrored independent sequences have two terminators each: a short suffix
                               // can tie with its copy at the other one)
                               (dna_fast || (independent && text.terms.mirror)) ? (uint32_t)k_syms : 0u, false,
                               rec_fast ? 32u : text.terms.seq_shift, store_ranks);

    arena.rewind(sort_mark);  // keys and the second value buffer are done

    // ---- doubling rounds: set-up ------------------------------------------------------------
    int nbits = 1;
    while (nbits < 32 && (1ull << nbits) <= (uint64_t)n) ++nbits;  // ranks and slots are <= n
    const int half_passes = (nbits + kRadixBits - 1) / kRadixBits;
    int shifts[8], npasses = 0;
    for (int p = 0; p < half_passes; ++p) shifts[npasses++] = p * kRadixBits;
    for (int p = 0; p < half_passes; ++p) shifts[npasses++] = 32 + p * kRadixBits;

    int rounds = 0, a_cur = 0;
    uint64_t h = (uint64_t)k_syms;
    static const bool trace = getenv("NOLZSS_TRACE") != nullptr;  // active-list sizes to stderr
    if (trace) fprintf(stderr, "[nolzss] n=%u: %u suffixes tied after the %d-symbol key sort\n", n, m, k_syms);

    // one pass writes rank[] for everybody: rank[sa[slot]] = rank_by_slot[slot] (rank_by_slot is not needed
    // afterwards: it serves as one of the ping-pong buffers)
    auto write_all_ranks = [&] {
        ProfScope ps(ctx.profiler(), "sa_rank_scatter", s);
        const size_t smark = arena.mark();
        // rank of a slot = slot of its group head + 1, and the heads are the slots whose LCP entry is decided: an
        // inclusive max-scan (the regroup kernels no longer write this array: when the direct rounds finish the
        // suffix array nobody reads it)
        if (!store_ranks) {
            head_flags_kernel<<<grid_for(n, kThreads), kThreads, 0, s>>>(lcp, n, rank_by_slot);
            KERNEL_CHECK();
            scan_inclusive_max_u32(rank_by_slot, rank_by_slot, n, arena, s);
        }
        uint32_t *idx[2] = {sa, arena.alloc<uint32_t>(n)};
        uint32_t *val[2] = {rank_by_slot, arena.alloc<uint32_t>(n)};
        bucketed_scatter(idx, val, n, rank, n, arena, s, ctx.profiler(), true, /*keep_val=*/false, ctx.rec_plan);
        arena.rewind(smark);
    };

    // ---- direct round: small groups are finished by comparing packed suffixes ---------------
    // (groups larger than kSmallGroup stay as they are; rank[] is not needed before the doubling
    // rounds, so it is written once, after this round, instead of after each of the two)
    uint32_t depth_compared = 0xffffffffu, depth_untouched = 0xffffffffu;  // what the direct round reports (0xffffffff: none)
    uint64_t untouched_members = 0;                                         // (an estimate)
    if (m > 0 && h < n) {
        const uint32_t *slot = act_slot[a_cur], *grp = act_grp[a_cur];
        const size_t direct_mark = arena.mark();
        // the members that stay tied come back in one region per workgroup (group_refine_kernel's epilogue)
        const unsigned g = (unsigned)div_up(m, kRefineTile);
        uint32_t *surv_slot = arena.alloc<uint32_t>((size_t)g * kRefineThreads);
        uint32_t *surv_head = arena.alloc<uint32_t>((size_t)g * kRefineThreads);
        uint32_t *surv_count = arena.alloc<uint32_t>(g);
        uint32_t *surv_off = arena.alloc<uint32_t>(g);
        uint32_t *rbs = store_ranks ? rank_by_slot : nullptr;
        // at most 32 words (1024 bases of DNA) deep; longer ties are cheaper in the doubling rounds
        static const uint32_t cap_words = getenv("NOLZSS_REFINE_WORDS") ? (uint32_t)atoi(getenv("NOLZSS_REFINE_WORDS")) : 32u;
        const uint32_t cap = (uint32_t)k_syms + cap_words * (64u / (uint32_t)text.bits);
        // [0] classes the round compared and left tied, [1] groups it did not touch, [2] members of such groups in every 64th workgroup
        uint32_t *d_min_depth = arena.alloc<uint32_t>(3);
        HIP_CHECK(hipMemsetAsync(d_min_depth, 0xff, 2 * sizeof(uint32_t), s));
        HIP_CHECK(hipMemsetAsync(d_min_depth + 2, 0, sizeof(uint32_t), s));
        {
            ProfScope ps(ctx.profiler(), "sa_direct_sort", s);
            static const bool want_rphases = getenv("NOLZSS_REFINE_PHASES") != nullptr;
            static const bool no_strag = getenv("NOLZSS_NO_STRAGGLERS") != nullptr;  // (A/B switch)
            unsigned long long *rphases = nullptr;
            if (want_rphases) {
                rphases = arena.alloc<unsigned long long>(8);
                HIP_CHECK(hipMemsetAsync(rphases, 0, 64, s));
            }
            switch (text.bits) {
            case 2:
                if (rphases) group_refine_kernel<2, true><<<g, kRefineThreads, 0, s>>>(slot, grp, sa, text.words, text.terms, m, (uint32_t)h, cap, lcp, rbs, surv_slot, surv_head, surv_count, d_min_depth, rphases, no_strag);
                else group_refine_kernel<2, false><<<g, kRefineThreads, 0, s>>>(slot, grp, sa, text.words, text.terms, m, (uint32_t)h, cap, lcp, rbs, surv_slot, surv_head, surv_count, d_min_depth, nullptr, no_strag);
                break;
            case 4:
                if (rphases) group_refine_kernel<4, true><<<g, kRefineThreads, 0, s>>>(slot, grp, sa, text.words, text.terms, m, (uint32_t)h, cap, lcp, rbs, surv_slot, surv_head, surv_count, d_min_depth, rphases, no_strag);
                else group_refine_kernel<4, false><<<g, kRefineThreads, 0, s>>>(slot, grp, sa, text.words, text.terms, m, (uint32_t)h, cap, lcp, rbs, surv_slot, surv_head, surv_count, d_min_depth, nullptr, no_strag);
                break;
            default:
                if (rphases) group_refine_kernel<8, true><<<g, kRefineThreads, 0, s>>>(slot, grp, sa, text.words, text.terms, m, (uint32_t)h, cap, lcp, rbs, surv_slot, surv_head, surv_count, d_min_depth, rphases, no_strag);
                else group_refine_kernel<8, false><<<g, kRefineThreads, 0, s>>>(slot, grp, sa, text.words, text.terms, m, (uint32_t)h, cap, lcp, rbs, surv_slot, surv_head, surv_count, d_min_depth, nullptr, no_strag);
                break;
            }
            KERNEL_CHECK();
            if (rphases) {
                unsigned long long hp[8];
                HIP_CHECK(hipMemcpyAsync(hp, rphases, 64, hipMemcpyDeviceToHost, s));
                HIP_CHECK(hipStreamSynchronize(s));
                const double wn = hp[5] ? (double)hp[5] : 1.0;
                fprintf(stderr, "[nolzss] group_refine phases (cycles per workgroup, %llu sampled): set-up %.0f  fetch+wait %.0f  compare %.0f  output %.0f  total %.0f  rounds %.2f\n",
                        hp[5], hp[0] / wn, hp[1] / wn, hp[2] / wn, hp[3] / wn, hp[6] / wn, hp[4] / wn);
            }
        }
        {
            // the next active list: the regions one after the other (they are in slot order already)
            ProfScope ps(ctx.profiler(), "sa_regroup", s, 16.0 * (double)g);
            scan_exclusive_add_u32(surv_count, surv_off, g, d_total, arena, s);
            compact_survivors_kernel<<<grid_for((size_t)g * 4, kThreads), kThreads, 0, s>>>(
                surv_slot, surv_head, surv_count, surv_off, g, act_slot[a_cur ^ 1], act_grp[a_cur ^ 1]);
            KERNEL_CHECK();
        }
        ctx.read_back(d_total, &m, 1);
        a_cur ^= 1;
        arena.rewind(direct_mark);
        // every group that is still tied agrees on at least min_depth symbols (K if a group was too large
        // for the round, more if the round left all its ties at the cap or at a bail-out depth): the
        // doubling rounds start there instead of repeating the steps K, 2K, 4K, ...
        if (m > 0) {
            uint32_t depth2[3] = {0, 0, 0};
            ctx.read_back(d_min_depth, depth2, 3);
            depth_compared = depth2[0];
            depth_untouched = depth2[1];
            untouched_members = (uint64_t)depth2[2] * 64u;
            const uint32_t depth = depth2[0] < depth2[1] ? depth2[0] : depth2[1];
            if (depth != 0xffffffffu && depth > h) h = depth;
        }
        if (trace) fprintf(stderr, "[nolzss]   direct round (cap %u symbols): %u still tied, on at least %llu symbols\n", cap, m, (unsigned long long)h);
    }

    // ---- second direct round: what little is left is sorted group by group, by the text (group_sort.hpp) ----
    // (NOLZSS_NO_DIRECT2: A/B switch; NOLZSS_DIRECT2_MAX: largest share of the text, 1 / this, that takes it --
    // a text with more ties than that is repetitive, and the passes below are made for those)
    static const bool no_direct2 = getenv("NOLZSS_NO_DIRECT2") != nullptr;
    static const uint32_t direct2_div = getenv("NOLZSS_DIRECT2_MAX") ? (uint32_t)atoi(getenv("NOLZSS_DIRECT2_MAX")) : 64u;
    // The EQUALISING round (collections of similar genomes): where much is tied and the first direct round left groups
    // untouched -- more than 64 members, or pairs that did not fit its list -- the doubling rounds would start at the
    // key depth for everything, four rounds over the whole list below the depth the compared classes already have.
    // The same kernels take only the untouched groups (told by their pending code) for as many rounds as reach that
    // depth: every round adds a window of 64 symbols to what a tied segment is known to agree on.  Only where the
    // untouched groups hold a minor part of what is tied (estimated by the first round): a wavefront per group of 65 and
    // more members that ALL stay tied takes 170 us per group and round -- 96 genomes of 2^28 bases in all, every suffix
    // in such a group, spent 480 ms here -- and the tiles of small groups 85 ms on 48 genomes, what four doubling rounds cost.
    static const bool no_equalise = getenv("NOLZSS_NO_EQUALISE") != nullptr;  // (A/B switch)
    // (NOLZSS_PIVOT_MIN: the tests and the fuzzer send every text with that many tied suffixes through the pivot rounds)
    static const long long pivot_min = getenv("NOLZSS_PIVOT_MIN") ? atoll(getenv("NOLZSS_PIVOT_MIN")) : -1;
    const bool force_pivot = pivot_min >= 0 && (long long)m >= pivot_min;
    const bool full_direct2 = !force_pivot && !no_direct2 && m > 0 && h < n && !independent && direct2_div > 0 && m <= n / direct2_div + 1024u;
    const bool equalise = !full_direct2 && !no_direct2 && !no_equalise && m > 0 && h < n && !independent && text.bits == 2 &&
                          depth_untouched != 0xffffffffu && depth_untouched == h &&
                          depth_compared != 0xffffffffu && depth_compared >= 2 * h && untouched_members <= m / 3;
    if (trace && m > 0 && depth_untouched != 0xffffffffu)
        fprintf(stderr, "[nolzss]   about %llu of the tied suffixes sit in groups the direct round did not compare (depth %u; compared classes: %u)\n",
                (unsigned long long)untouched_members, depth_untouched, depth_compared);
    // PIVOT rounds (group_sort.hpp, kPivot): a repetitive text whose tied suffixes sit in groups of more than two or three
    // -- a collection of similar sequences -- has every tied group of up to kGroupSortMax members sorted against pivots,
    // kPivotDepth symbols deep: what stays tied agrees that far, and the doubling rounds start there instead of at the
    // key depth.  Texts whose ties are pairs (two copies: the pair-run pass) or runs of a short period (groups as large as
    // the runs: the periodic pass) are told by a count over the list and skip it.  NOLZSS_NO_PIVOT: A/B switch.
    static const bool no_pivot = getenv("NOLZSS_NO_PIVOT") != nullptr;
    static const uint32_t pivot_depth = getenv("NOLZSS_PIVOT_DEPTH") ? (uint32_t)atoi(getenv("NOLZSS_PIVOT_DEPTH")) : 2048u;
    bool pivot = false;
    if (!no_pivot && !full_direct2 && m > 0 && h < n && !independent) {
        uint32_t *d_cnt = arena.alloc<uint32_t>(4);
        HIP_CHECK(hipMemsetAsync(d_cnt, 0, 4 * sizeof(uint32_t), s));
        per_count_large_kernel<<<grid_for(m, kThreads), kThreads, 0, s>>>(act_slot[a_cur], act_grp[a_cur], m, sa, kGroupSortMax,
                                                                        kPerVerifyMax, d_cnt);
        KERNEL_CHECK();
        uint32_t c4[4] = {0, 0, 0, 0};
        ctx.read_back(d_cnt, c4, 4);
        const uint64_t huge = (uint64_t)c4[0] + (uint64_t)kGroupSortMax * c4[1];  // members of groups beyond the kernels' reach
        pivot = force_pivot || ((uint64_t)m * 4 > (uint64_t)c4[2] * 10 && huge <= m / 2 && c4[3] <= m / 2);
        if (trace) fprintf(stderr, "[nolzss]   %u tied suffixes in %u groups, %llu in groups of more than %u, %u next to a member at most %u symbols away: %s\n",
                           m, c4[2], (unsigned long long)huge, kGroupSortMax, c4[3], kPerVerifyMax, pivot ? "pivot rounds" : "no pivot rounds");
    }
    // (pivot rounds come in PASSES: what a pass leaves tied agrees on its cap, and while a pass finishes at least half of what
    // it was given the next one goes four times as deep over what is left -- 96 genomes 0.1 % apart: 2.65e8 -> 5.9e7 -> 1e6
    // tied suffixes after caps of 2048 and 8192 symbols, where the doubling rounds would take four rounds and the rank scatter
    // in front of them.  A pass that finishes less than half -- exact copies -- hands over to the doubling rounds.)
    // (the depth is given in symbols of 2-bit DNA; wider symbols get proportionally fewer, so that a member reads the same
    // number of text words whatever the alphabet: 2048 bases = 512 bytes)
    uint32_t pass_depth = std::max<uint32_t>(64u, pivot_depth * 2u / (uint32_t)text.bits);
    for (int pivot_pass = 0; (pivot_pass == 0 && (full_direct2 || equalise || pivot)) || (pivot_pass > 0 && pivot); ++pivot_pass) {
        // (group_sort.hpp carries the terminator index of a suffix that ends inside a comparison in 16 bits)
        if (text.terms.count > 0x10000u) throw HipError("suffix array: the group-sort rounds take texts of at most 65536 segments");
        const uint32_t max_rounds = pivot ? kGroupSortRounds : equalise ? std::min<uint32_t>(kGroupSortRounds, (depth_compared - (uint32_t)h + 63u) / 64u) : kGroupSortRounds;
        const uint32_t *lcp_mark = (equalise && !pivot) ? lcp : nullptr;
        const uint32_t depth_cap = (uint32_t)std::min<uint64_t>((uint64_t)h + pass_depth, 0xfffffff0u);
        const uint32_t *slot = act_slot[a_cur], *grp = act_grp[a_cur];
        const size_t d2_mark = arena.mark();
        uint32_t *out_lo = arena.alloc<uint32_t>(m);
        uint32_t *lcp_list = arena.alloc<uint32_t>(m);
        uint32_t *d_min_depth = arena.alloc<uint32_t>(1);
        HIP_CHECK(hipMemsetAsync(d_min_depth, 0xff, sizeof(uint32_t), s));
        const unsigned dir_blocks = (unsigned)div_up(m, kThreads);
        ShardQueue q_mid0, q_mid, q_big;
        q_mid0.cap = q_mid.cap = q_big.cap = (uint32_t)shard_queue_cap(dir_blocks, kThreads);
        uint32_t *qcounts = arena.alloc<uint32_t>(3 * kQShards * kQPad);
        ShardQueue *queues[3] = {&q_mid0, &q_mid, &q_big};
        for (int k = 0; k < 3; ++k) {
            queues[k]->items = arena.alloc<uint32_t>((size_t)kQShards * q_big.cap);
            queues[k]->items2 = arena.alloc<uint32_t>((size_t)kQShards * q_big.cap);
            queues[k]->counts = qcounts + (size_t)k * kQShards * kQPad;
        }
        HIP_CHECK(hipMemsetAsync(qcounts, 0, 3 * kQShards * kQPad * sizeof(uint32_t), s));
        {
            ProfScope ps(ctx.profiler(), "sa_direct_sort2", s);
            group_dir_kernel<<<dir_blocks, kThreads, 0, s>>>(slot, grp, m, (uint32_t)h, out_lo, lcp_list, q_mid0, q_mid, q_big, d_min_depth, lcp_mark);
            KERNEL_CHECK();
            // small groups by tiles of the list; the larger ones from the queues (the consumers read the shard
            // counts on the device: no read-back in between)
            constexpr int kSmallN = 2 * (int)kGroupSortSmall, kMid0N = (int)kGroupSortMid0, kMidN = (int)kGroupSortMid,
                          kBigN = (int)kGroupSortMax;
            const unsigned tiles = (unsigned)div_up(m, kGroupSortSmall);
            const unsigned ym = (unsigned)std::min<size_t>(64, std::max<size_t>(1, div_up(m, (size_t)kQShards * 64)));
            const unsigned yb = (unsigned)std::min<size_t>(16, std::max<size_t>(1, div_up(m, (size_t)kQShards * 256)));
            const uint32_t h32 = (uint32_t)h;
#define NOLZSS_GROUP_SORT(B)                                                                                                \
    group_sort_kernel<B, 64, kSmallN, true><<<tiles, 64, 0, s>>>(q_mid, slot, grp, m, sa, text.words, text.terms, h32,    \
                                                                  out_lo, lcp_list, d_min_depth, lcp_mark, max_rounds);     \
    group_sort_kernel<B, 64, kMid0N, false><<<dim3(kQShards, ym), 64, 0, s>>>(q_mid0, slot, grp, m, sa, text.words,        \
                                                                              text.terms, h32, out_lo, lcp_list,           \
                                                                              d_min_depth, lcp_mark, max_rounds);          \
    group_sort_kernel<B, 64, kMidN, false><<<dim3(kQShards, ym), 64, 0, s>>>(q_mid, slot, grp, m, sa, text.words,          \
                                                                             text.terms, h32, out_lo, lcp_list,            \
                                                                             d_min_depth, lcp_mark, max_rounds);           \
    group_sort_kernel<B, 256, kBigN, false><<<dim3(kQShards, yb), 256, 0, s>>>(q_big, slot, grp, m, sa, text.words,        \
                                                                               text.terms, h32, out_lo, lcp_list,          \
                                                                               d_min_depth, lcp_mark, max_rounds)
#define NOLZSS_GROUP_PIVOT(B)                                                                                               \
    group_sort_kernel<B, 64, kSmallN, true, true><<<tiles, 64, 0, s>>>(q_mid, slot, grp, m, sa, text.words, text.terms,    \
                                                                        h32, out_lo, lcp_list, d_min_depth, lcp_mark,       \
                                                                        max_rounds, depth_cap);                             \
    group_sort_kernel<B, 128, kMid0N, false, true><<<dim3(kQShards, ym), 128, 0, s>>>(                                      \
        q_mid0, slot, grp, m, sa, text.words, text.terms, h32, out_lo, lcp_list, d_min_depth, lcp_mark, max_rounds,         \
        depth_cap);                                                                                                         \
    group_sort_kernel<B, 256, kMidN, false, true><<<dim3(kQShards, ym), 256, 0, s>>>(                                       \
        q_mid, slot, grp, m, sa, text.words, text.terms, h32, out_lo, lcp_list, d_min_depth, lcp_mark, max_rounds,          \
        depth_cap);                                                                                                         \
    group_sort_kernel<B, 256, kBigN, false, true><<<dim3(kQShards, yb), 256, 0, s>>>(                                       \
        q_big, slot, grp, m, sa, text.words, text.terms, h32, out_lo, lcp_list, d_min_depth, lcp_mark, max_rounds,          \
        depth_cap)
            if (pivot) {
                switch (text.bits) {
                case 2: NOLZSS_GROUP_PIVOT(2); break;
                case 4: NOLZSS_GROUP_PIVOT(4); break;
                default: NOLZSS_GROUP_PIVOT(8); break;
                }
            } else {
                switch (text.bits) {
                case 2: NOLZSS_GROUP_SORT(2); break;
                case 4: NOLZSS_GROUP_SORT(4); break;
                default: NOLZSS_GROUP_SORT(8); break;
                }
            }
#undef NOLZSS_GROUP_PIVOT
#undef NOLZSS_GROUP_SORT
            KERNEL_CHECK();
        }
        const uint32_t before = m;
        m = regroup<false>(ctx, nullptr, grp, out_lo, nullptr, slot, m, n, sa, rank, act_slot[a_cur ^ 1],
                           act_grp[a_cur ^ 1], nullptr, nullptr, nullptr, d_total, lcp,
                           0, 0, 0, lcp_list, 0, (uint32_t)h, rank_by_slot, nullptr, nullptr, nullptr, 0u,
                           /*sa_is_current=*/true, 0u, store_ranks);
        a_cur ^= 1;
        if (m > 0) {
            uint32_t depth = 0;
            ctx.read_back(d_min_depth, &depth, 1);
            // (equalising: the classes the first round compared were not looked at; they keep their depth)
            if (equalise && !pivot && depth_compared < depth) depth = depth_compared;
            if (depth != 0xffffffffu && depth > h) h = depth;
        }
        arena.rewind(d2_mark);
        if (trace) fprintf(stderr, "[nolzss]   %s: %u of %u finished, %u still tied, on at least %llu symbols\n",
                           pivot ? "pivot rounds" : equalise ? "equalising round (untouched groups only)" : "second direct round", before - m, before, m, (unsigned long long)h);
        // another pass?  only pivot passes repeat: while they make progress, something is left, and the depth has room
        static const int pivot_passes = getenv("NOLZSS_PIVOT_PASSES") ? atoi(getenv("NOLZSS_PIVOT_PASSES")) : 3;
        if (!pivot || m == 0 || h >= n || pivot_pass + 1 >= pivot_passes || (uint64_t)(before - m) * 2 < before) break;
        pass_depth = pass_depth < (1u << 28) ? pass_depth * 4 : pass_depth;
    }
    // Nothing is tied any more: no round below needs rank[].  A caller that can wait gets it from the permutation
    // that brings the factor-length codes into text order (pipeline.hpp) -- one full random permutation per
    // factorization instead of two.
    if (m == 0 && can_defer) {
        *isa_deferred = true;
        if (trace) fprintf(stderr, "[nolzss]   suffix array finished by the direct rounds: rank[] is left to the permutation of the codes\n");
        HIP_CHECK(hipMemsetAsync(lcp + n, 0, sizeof(uint32_t), s));
        arena.rewind(mark);
        return 0;
    }
    write_all_ranks();

    // Work arrays of the rounds that follow, one entry per tied suffix -- or per text position when the
    // pair-run pass will run (it works in text order; the rounds behind it then use the same arrays).
    static const long long pair_runs_min = getenv("NOLZSS_PAIR_RUNS_MIN") ? atoll(getenv("NOLZSS_PAIR_RUNS_MIN")) : -1;
    const bool pair_runs = m > 0 && (pair_runs_min >= 0 ? (long long)m >= pair_runs_min : m >= n / 16);
    const size_t wlen = m == 0 ? 0 : (pair_runs ? (size_t)n : (size_t)m);
    uint32_t *tmp_a = nullptr, *tmp_b = nullptr, *tmp_c = nullptr, *rank_val = nullptr, *scratch_idx = nullptr;
    uint32_t *scratch_val = nullptr, *lo = nullptr, *out_lo = nullptr, *out_vals = nullptr;
    if (m > 0) {
        try {
            tmp_a = arena.alloc<uint32_t>(wlen);
            tmp_b = arena.alloc<uint32_t>(wlen);
            tmp_c = arena.alloc<uint32_t>(wlen);
            rank_val = arena.alloc<uint32_t>(wlen);
            scratch_idx = arena.alloc<uint32_t>(wlen);
            scratch_val = arena.alloc<uint32_t>(wlen);
            lo = arena.alloc<uint32_t>(wlen);
            out_lo = arena.alloc<uint32_t>(wlen);
            out_vals = arena.alloc<uint32_t>(wlen);
        } catch (const HipError &) {
            char buf[256];
            snprintf(buf, sizeof buf,
                     "suffix array: %u of %u suffixes are still tied after the direct round (a highly repetitive text); "
                     "the rounds that resolve them need about %.1f GiB more device memory than this device has left",
                     m, n, 60.0 * (double)wlen / 1073741824.0);
            throw HipError(buf);
        }
    }
    uint32_t *rvals = rank_by_slot;  // free now that rank[] has been written

    // how the tied suffixes are grouped decides which of the two passes can do anything: in_large = members beyond
    // the first kRunGroupMax of their group, large_members = members of groups with more than kRunGroupMax members
    uint32_t in_large = 0, large_members = 0, tied_groups = 0, near_members = 0;
    uint32_t *d_large = arena.alloc<uint32_t>(4);
    if (pair_runs) {
        HIP_CHECK(hipMemsetAsync(d_large, 0, 4 * sizeof(uint32_t), s));
        per_count_large_kernel<<<grid_for(m, kThreads), kThreads, 0, s>>>(act_slot[a_cur], act_grp[a_cur], m, sa, kRunGroupMax,
                                                                        kPerVerifyMax, d_large);
        KERNEL_CHECK();
        uint32_t h4[4] = {0, 0, 0, 0};
        ctx.read_back(d_large, h4, 4);
        in_large = h4[0];
        large_members = h4[0] + kRunGroupMax * h4[1];
        tied_groups = h4[2];
        near_members = h4[3];
        if (trace) fprintf(stderr, "[nolzss]   %u tied suffixes in %u groups, %u of them in groups of more than %u, %u next to a member at most %u symbols away\n",
                           m, tied_groups, large_members, kRunGroupMax, near_members, kPerVerifyMax);
    }

    // one range-minimum pyramid over the LCP values known so far; the regroup kernel keeps it current
    Pyramid Plcp{};
    const size_t pyr_mark = arena.mark();
    if (m > 0) {
        ProfScope ps(ctx.profiler(), "sa_lcp_pyramid", s);
        Plcp = build_pyramid(lcp, n + 1, false, arena, s);
    }

    // ---- periodic runs: groups whose members lie one short period apart are ordered arithmetically ----
    // (kernels and the argument above, "Periodic runs".  Tried when a large part of the text is still tied:
    // once behind the direct round, and again in the doubling rounds when the depth has reached the
    // shortest distance that was too long for it.  NOLZSS_NO_PERIODIC switches the pass off.)
    static const bool periodic_off = getenv("NOLZSS_NO_PERIODIC") != nullptr;
    uint32_t per_hint = 0;
    int per_attempts = 0;
    auto periodic_pass = [&]() -> bool {
        if (periodic_off || m == 0 || n >= 0x80000000u || wlen < n || per_attempts >= 3) return false;
        ProfScope ps(ctx.profiler(), "sa_periodic", s);
        const uint32_t *slot = act_slot[a_cur], *grp = act_grp[a_cur];
        if (per_attempts == 0 && (in_large < m / 8 || near_members < m / 32)) {
            // worth its two sorts only where large groups hold a good part of what is tied: copies of long
            // regions tie in groups of a few members (the pair-run pass takes those), runs of a short period
            // in groups as large as the runs are long -- and only where tied suffixes lie close to each other in
            // the text: two dozen copies of a genome tie in groups of two dozen members a genome apart (the
            // first sort of the pass, 28 of 212 ms on 24 genomes of 2^28 bases in all, found that out before)
            per_attempts = 3;  // (never again for this text)
            return false;
        }
        ++per_attempts;
        uint32_t *PQ = tmp_a, *rev = tmp_b, *end_of = tmp_c, *gq = lo, *kraw = rank_val;
        uint32_t *d_hint = d_total + 3;
        const size_t lmark = arena.mark();
        uint64_t *pk[2] = {arena.alloc<uint64_t>(m), arena.alloc<uint64_t>(m)};
        uint32_t *pv[2] = {arena.alloc<uint32_t>(m), arena.alloc<uint32_t>(m)};
        HIP_CHECK(hipMemsetAsync(gq, 0xff, (size_t)n * sizeof(uint32_t), s));
        HIP_CHECK(hipMemsetAsync(PQ, 0, (size_t)n * sizeof(uint32_t), s));
        HIP_CHECK(hipMemsetAsync(d_hint, 0xff, sizeof(uint32_t), s));
        per_keys_kernel<<<grid_for(m, kThreads), kThreads, 0, s>>>(slot, grp, m, sa, pk[0], pv[0]);
        KERNEL_CHECK();
        const int c = radix_sort_pairs(pk, pv, m, shifts, npasses, arena, s, ctx.profiler());
        per_link_kernel<<<(unsigned)div_up(m, kThreads), kThreads, 0, s>>>(pk[c], m, gq);
        KERNEL_CHECK();
        const uint32_t depth = (uint32_t)std::min<uint64_t>(h, 0x7ffffffeu);
        const bool can_verify = text.terms.count == 1;  // (one segment: see per_verify_kernel)
        {
            // Large groups are not always periodic runs: two dozen copies of a genome tie in groups of two dozen
            // members that lie a genome apart.  When next to nothing can be taken, the pass stops here, before
            // its scans and its second sort (49 of 300 ms on 24 genomes of 2^28 bases in all).
            const uint32_t qmax = std::max<uint32_t>(depth, (can_verify && depth < kPerVerifyMax) ? kPerVerifyMax : depth);
            HIP_CHECK(hipMemsetAsync(d_large, 0, sizeof(uint32_t), s));
            per_candidates_kernel<<<grid_for(m, kThreads), kThreads, 0, s>>>(pk[c], m, gq, qmax, d_large);
            KERNEL_CHECK();
            uint32_t cand = 0;
            ctx.read_back(d_large, &cand, 1);
            if (cand < m / 16) {
                arena.rewind(lmark);
                per_attempts = 3;
                if (trace) fprintf(stderr, "[nolzss]   periodic runs (depth %llu): %u of %u tied suffixes in groups that could be runs -- skipped\n",
                                   (unsigned long long)h, cand, m);
                return false;
            }
        }
        if (can_verify && depth < kPerVerifyMax) {  // longer periods than the depth: taken if the text confirms them
            const unsigned gv = grid_for(m, kThreads, 256u * 64u);
            switch (text.bits) {
            case 2: per_verify_kernel<2><<<gv, kThreads, 0, s>>>(pk[c], m, gq, depth, n, text.words, text.terms); break;
            case 4: per_verify_kernel<4><<<gv, kThreads, 0, s>>>(pk[c], m, gq, depth, n, text.words, text.terms); break;
            default: per_verify_kernel<8><<<gv, kThreads, 0, s>>>(pk[c], m, gq, depth, n, text.words, text.terms); break;
            }
            KERNEL_CHECK();
        }
        // (groups the text check has flagged are out; the others pass up to kPerVerifyMax, beyond it up to the depth)
        per_flags_kernel<<<grid_for(m, kThreads), kThreads, 0, s>>>(pk[c], m, gq, (can_verify && depth < kPerVerifyMax) ? kPerVerifyMax : depth, PQ, d_hint);
        KERNEL_CHECK();
        per_breaks_kernel<<<grid_for(n, kThreads, 256u * 64u), kThreads, 0, s>>>(PQ, n, rev);
        KERNEL_CHECK();
        scan_inclusive_max_u32(rev, end_of, n, arena, s);
        per_rho_kernel<<<grid_for(m, kThreads), kThreads, 0, s>>>(pk[c], m, gq, PQ, end_of, rank, n, Plcp, depth, kraw);
        KERNEL_CHECK();
        per_keys2_kernel<<<grid_for(m, kThreads), kThreads, 0, s>>>(pk[c], m, gq, kraw, pk[c ^ 1], pv[c ^ 1]);
        KERNEL_CHECK();
        uint64_t *pk2[2] = {pk[c ^ 1], pk[c]};
        uint32_t *pv2[2] = {pv[c ^ 1], pv[c]};
        const int c2 = radix_sort_pairs(pk2, pv2, m, shifts, npasses, arena, s, ctx.profiler());
        uint32_t *grp_sorted = tmp_a, *lcp_list = tmp_b;  // (PQ and rev are done)
        per_view_kernel<<<grid_for(m, kThreads), kThreads, 0, s>>>(pk2[c2], pv2[c2], m, grp_sorted, out_lo, out_vals, lcp_list);
        KERNEL_CHECK();
        arena.rewind(lmark);
        uint32_t hint = 0;
        ctx.read_back(d_hint, &hint, 1);
        per_hint = hint == 0xffffffffu ? 0u : hint;
        const uint32_t before = m;
        m = regroup<false>(ctx, nullptr, grp_sorted, out_lo, out_vals, slot, m, n, sa, rank, act_slot[a_cur ^ 1],
                           act_grp[a_cur ^ 1], scratch_idx, scratch_val, rank_val, d_total, lcp,
                           0, 0, 0, lcp_list, 0, 0u, nullptr, &Plcp);
        a_cur ^= 1;
        if (trace) fprintf(stderr, "[nolzss]   periodic runs (depth %llu): %u of %u tied suffixes finished%s\n",
                           (unsigned long long)h, before - m, before, per_hint ? " (a longer period waits for more depth)" : "");
        return m < before - before / 8;
    };
    if (pair_runs) periodic_pass();

    // ---- long exact repeats: small groups along runs of text positions are finished arithmetically ---
    // (worth its passes over the text only when a large part of it is still tied; a pass that splits
    // groups without finishing them -- three copies, one of which differs behind the run -- is followed by
    // another one over the smaller groups.  NOLZSS_PAIR_RUNS_MIN: smallest number of tied suffixes for
    // which it runs, the tests set 1)
    // (the pass takes groups of up to kRunGroupMax members: where most of what is tied sits in larger groups -- 17
    // and more copies of a genome -- its four sweeps over the text finish next to nothing: 120 of 300 ms on 24 genomes)
    // (and a group of k copies is finished by about k - 1 passes, each a sweep over the whole text that costs as much as
    // two doubling rounds: 168 ms for the four passes of five genomes, 120 ms for one pass of twelve that finished 5 %
    // of what was tied, where the doubling rounds from the depth the direct round reached take 85 ms; three genomes:
    // 115 ms with two passes, 92 ms with the doubling rounds; two genomes: 72 against 82 ms, two exact copies 79 against
    // 279 ms -- the passes run where the tied suffixes sit in pairs: mean group size at most 2.5)
    static const uint32_t runs_avg4 = getenv("NOLZSS_PAIR_RUNS_AVG4") ? (uint32_t)atoi(getenv("NOLZSS_PAIR_RUNS_AVG4")) : 10u;  // 4 x mean group size
    const bool runs_can_help = pair_runs_min >= 0 || (large_members <= m / 2 && (uint64_t)m * 4 <= (uint64_t)tied_groups * runs_avg4);
    for (int pass = 0; pair_runs && runs_can_help && pass < 10 && m > 0 && (pair_runs_min >= 0 ? (long long)m >= pair_runs_min : m >= n / 16); ++pass) {
        ProfScope ps(ctx.profiler(), "sa_pair_runs", s);
        uint32_t *link = tmp_a, *gsz = rank_val, *rev = tmp_b, *end_of = tmp_c, *togo = scratch_idx;
        uint32_t *end_place = scratch_val, *end_lcp = lo;
        uint32_t *end_head = out_lo;
        const unsigned g = grid_for(n, kThreads, 256u * 64u);
        {
            ProfScope p1(ctx.profiler(), "runs_link", s);
            group_link_kernel<<<g, kThreads, 0, s>>>(rank, sa, lcp, n, link, gsz);
            KERNEL_CHECK();
        }
        {
            ProfScope p2(ctx.profiler(), "runs_scan", s);
            run_breaks_kernel<<<g, kThreads, 0, s>>>(link, gsz, n, rev);
            KERNEL_CHECK();
            scan_inclusive_max_u32(rev, end_of, n, arena, s);
            group_run_kernel<<<g, kThreads, 0, s>>>(gsz, rank, sa, end_of, n, togo);
            KERNEL_CHECK();
        }
        {
            ProfScope p3(ctx.profiler(), "runs_end", s);
            group_end_kernel<<<g, kThreads, 0, s>>>(gsz, togo, end_of, rank, sa, n, Plcp, end_place, end_head, end_lcp);
            KERNEL_CHECK();
        }
        {
            ProfScope p4(ctx.profiler(), "runs_members", s);
            group_members_kernel<<<g, kThreads, 0, s>>>(gsz, togo, end_of, n, end_place, end_head, end_lcp, rank, sa, lcp);
            KERNEL_CHECK();
        }
        ProfScope p5(ctx.profiler(), "runs_compact", s);
        // the active list without the suffixes that are done, with the new group heads of the others
        uint32_t *keep = tmp_a, *pos = tmp_b, *head = tmp_c;
        still_tied_kernel<<<grid_for(m, kThreads), kThreads, 0, s>>>(act_slot[a_cur], act_grp[a_cur], m, lcp, n, head, keep);
        KERNEL_CHECK();
        scan_exclusive_add_u32(keep, pos, m, d_total, arena, s);
        compact_active_kernel<<<grid_for(m, kThreads), kThreads, 0, s>>>(act_slot[a_cur], head, keep, pos, m,
                                                                       act_slot[a_cur ^ 1], act_grp[a_cur ^ 1]);
        KERNEL_CHECK();
        uint32_t left = 0;
        ctx.read_back(d_total, &left, 1);
        if (trace) fprintf(stderr, "[nolzss]   pair runs: %u of %u tied suffixes finished\n", m - left, m);
        const bool progress = left < m - m / 8;
        m = left;
        a_cur ^= 1;
        if (m > 0) {
            arena.rewind(pyr_mark);
            Plcp = build_pyramid(lcp, n + 1, false, arena, s);
        }
        if (!progress) break;
    }

    static const bool no_mid_sort = getenv("NOLZSS_NO_MID_SORT") != nullptr;  // (A/B switch)
    bool mid_groups = !no_mid_sort;
    while (m > 0) {
        if (h >= n || rounds > 40) throw HipError("suffix array: prefix doubling failed to converge");
        if (pair_runs && per_hint != 0 && h >= per_hint && m >= n / 16) {
            periodic_pass();
            if (m == 0) break;
        }
        const uint32_t *slot = act_slot[a_cur], *grp = act_grp[a_cur];
        {
            ProfScope ps(ctx.profiler(), "sa_round_keys", s);
            round_keys_kernel<<<grid_for(m, kThreads), kThreads, 0, s>>>(slot, m, sa, rank, n, (uint32_t)h, lo, rvals);
            KERNEL_CHECK();
        }
        {
            ProfScope ps(ctx.profiler(), "sa_small_sort", s);
            small_sort_kernel<<<grid_for(m, kThreads), kThreads, 0, s>>>(slot, grp, lo, rvals, m, out_lo, out_vals,
                                                                         tmp_a);
            KERNEL_CHECK();
            if (mid_groups) {  // groups of 65 .. 1024 members, in LDS (until a round finds none: groups only shrink)
                HIP_CHECK(hipMemsetAsync(d_total + 1, 0, sizeof(uint32_t), s));
                mid_sort_kernel<<<(unsigned)div_up(m, kMidGroup), kMidThreads, 0, s>>>(slot, grp, lo, rvals, m, out_lo, out_vals,
                                                                                      tmp_a, d_total + 1);
                KERNEL_CHECK();
            }
            scan_exclusive_add_u32(tmp_a, tmp_b, m, d_total, arena, s);
        }
        uint32_t n_large = 0;
        {
            uint32_t h2[2] = {0, 0};
            ctx.read_back(d_total, h2, 2);
            n_large = h2[0];
            if (mid_groups && h2[1] == 0 && n_large == 0) mid_groups = false;
        }
        // (NOLZSS_NO_SEG_LARGE: A/B switch back to the global sort of 12-byte (group, key) records)
        static const bool no_seg_large = getenv("NOLZSS_NO_SEG_LARGE") != nullptr;
        bool large_done = false;
        if (n_large > 0 && !no_seg_large) {
            // Members of the large groups: the group of an element is known from where it lies (a group's members are
            // consecutive in the list, so also in the gathered array), so the groups are the BUCKETS of a segmented
            // sort by the key alone -- four passes on 8-byte records instead of eight on 12-byte ones.
            ProfScope ps(ctx.profiler(), "sa_sort_large", s);
            const size_t lmark = arena.mark();
            uint32_t *lk[2] = {arena.alloc<uint32_t>(n_large), arena.alloc<uint32_t>(n_large)};
            uint32_t *lv[2] = {arena.alloc<uint32_t>(n_large), arena.alloc<uint32_t>(n_large)};
            uint32_t *lgrp = arena.alloc<uint32_t>(n_large);
            uint32_t *lidx = tmp_c;
            gather_large32_kernel<<<grid_for(m, kThreads), kThreads, 0, s>>>(tmp_a, tmp_b, grp, lo, rvals, m, lk[0], lv[0], lidx, lgrp);
            KERNEL_CHECK();
            // first element of every group -> segment table on the host
            uint32_t *head = lk[1], *pos = lv[1];  // (free until the first pass)
            large_heads_kernel<<<grid_for(n_large, kThreads), kThreads, 0, s>>>(lgrp, n_large, head);
            KERNEL_CHECK();
            scan_exclusive_add_u32(head, pos, n_large, d_total + 3, arena, s);
            uint32_t nb = 0;
            ctx.read_back(d_total + 3, &nb, 1);
            // (a tile of the segmented passes never straddles a group: groups of a few hundred members would leave the
            // 4096-pair tiles mostly empty -- those keep the global sort)
            if ((uint64_t)nb * 2048u <= (uint64_t)n_large) {
            uint32_t *d_starts = arena.alloc<uint32_t>((size_t)nb + 1);
            large_starts_kernel<<<grid_for(n_large, kThreads), kThreads, 0, s>>>(head, pos, n_large, d_starts);
            KERNEL_CHECK();
            std::vector<uint32_t> h_start((size_t)nb + 1);
            HIP_CHECK(hipMemcpyAsync(h_start.data(), d_starts, (size_t)nb * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
            h_start[nb] = n_large;
            const int c = radix_sort_segments_u32(lk, lv, n_large, h_start, half_passes, arena, s, ctx.profiler());
            scatter_large32_kernel<<<grid_for(n_large, kThreads), kThreads, 0, s>>>(lk[c], lv[c], lidx, n_large, out_lo, out_vals);
            KERNEL_CHECK();
            large_done = true;
            }
            arena.rewind(lmark);
        }
        if (n_large > 0 && !large_done) {  // members of groups larger than kSmallGroup: global radix sort
            ProfScope ps(ctx.profiler(), "sa_sort_large", s);
            const size_t lmark = arena.mark();
            uint64_t *lk[2] = {arena.alloc<uint64_t>(n_large), arena.alloc<uint64_t>(n_large)};
            uint32_t *lv[2] = {arena.alloc<uint32_t>(n_large), arena.alloc<uint32_t>(n_large)};
            uint32_t *lidx = tmp_c;
            gather_large_kernel<<<grid_for(m, kThreads), kThreads, 0, s>>>(tmp_a, tmp_b, grp, lo, rvals, m, lk[0],
                                                                           lv[0], lidx);
            KERNEL_CHECK();
            const int c = radix_sort_pairs(lk, lv, n_large, shifts, npasses, arena, s, ctx.profiler());
            scatter_large_kernel<<<grid_for(n_large, kThreads), kThreads, 0, s>>>(lk[c], lv[c], lidx, n_large,
                                                                                  out_lo, out_vals, nullptr, 0u);
            KERNEL_CHECK();
            arena.rewind(lmark);
        }
        m = regroup<false>(ctx, nullptr, grp, out_lo, out_vals, slot, m, n, sa, rank, act_slot[a_cur ^ 1],
                           act_grp[a_cur ^ 1], scratch_idx, scratch_val, rank_val, d_total, lcp,
                           0, 0, 0, nullptr, 0, (uint32_t)h, nullptr, &Plcp);
        a_cur ^= 1;
        if (trace)
            fprintf(stderr, "[nolzss]   doubling round h=%llu: %u in large groups, %u still tied\n",
                    (unsigned long long)h, n_large, m);
        h *= 2;
        ++rounds;
    }
    // (rank[] stays 1-based: every group is a singleton now, so rank[i] = ISA[i] + 1; the consumers
    // subtract the one instead of a pass over the array doing it)
    // every boundary received its LCP when it appeared; the caller checks that no pending code is left
    // while it builds the LCP pyramid (build_lcp_pyramid) instead of a pass of its own
    HIP_CHECK(hipMemsetAsync(lcp + n, 0, sizeof(uint32_t), s));
    arena.rewind(mark);
    return rounds;
}

uint32_t pending_threshold() { return kLcpPendingMin; }

void inject_pending_for_test(Context &ctx, uint32_t *lcp, uint32_t n) {
    // (test hook: make one entry pending so that the safety net runs)
    static const bool inject = getenv("NOLZSS_TEST_INJECT_PENDING") != nullptr;
    if (inject && n > 2) HIP_CHECK(hipMemsetAsync(lcp + n / 2, 0xff, sizeof(uint32_t), ctx.stream));
}

// safety net: compare the suffixes in the packed text wherever an LCP entry is still undecided
void finish_pending_lcp(Context &ctx, const PackedText &text, const uint32_t *sa, uint32_t *lcp) {
    const uint32_t n = text.n;
    hipStream_t s = ctx.stream;
    ProfScope ps(ctx.profiler(), "lcp_finish", s);
    const unsigned g = grid_for((size_t)n + 1, kThreads, 256u * 32u);
    const uint32_t skip = 1;  // (all that is known for sure: the suffixes differ somewhere)
    switch (text.bits) {
    case 2: lcp_finish_kernel<2><<<g, kThreads, 0, s>>>(text.words, n, text.terms, sa, skip - 1, lcp); break;
    case 4: lcp_finish_kernel<4><<<g, kThreads, 0, s>>>(text.words, n, text.terms, sa, skip - 1, lcp); break;
    default: lcp_finish_kernel<8><<<g, kThreads, 0, s>>>(text.words, n, text.terms, sa, skip - 1, lcp); break;
    }
    KERNEL_CHECK();
}

Pyramid build_lcp_pyramid(Context &ctx, const PackedText &text, const uint32_t *sa, uint32_t *lcp) {
    const uint32_t n = text.n;
    hipStream_t s = ctx.stream;
    uint32_t *flag = ctx.arena.alloc<uint32_t>(1);
    HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(uint32_t), s));
    inject_pending_for_test(ctx, lcp, n);
    const size_t mark = ctx.arena.mark();
    Pyramid P = build_pyramid(lcp, n + 1, false, ctx.arena, s, kLcpPendingMin, flag);
    uint32_t pending = 0;
    ctx.read_back(flag, &pending, 1);
    if (pending) {  // safety net: compare the suffixes in the packed text, then build again
        ctx.arena.rewind(mark);
        finish_pending_lcp(ctx, text, sa, lcp);
        P = build_pyramid(lcp, n + 1, false, ctx.arena, s);
    }
    return P;
}

}  // namespace nolzss
