// pyramid.hpp -- 16-ary min / max pyramids over u32 arrays with threshold-search and
// range queries.  These replace the reference's succinct RMQs and tree navigation
// (rmq_succinct_sct over the SA, factorizer_core.hpp:53,72,231-232; lb/rb/parent of cst_sada).
//
// Level k+1 holds the min (max) of each aligned group of 16 entries of level k, so the whole
// structure adds 1/15 of the base array.  A query touches at most 15 entries per level on the
// way up and on the way down; nearly all queries finish inside level 0/1 (one or two 64-byte
// lines) because LCP intervals around a rank are short.
#pragma once
#include "common.hpp"

namespace nolzss {

constexpr int kPyrShift = 4;
constexpr uint32_t kPyrFan = 1u << kPyrShift;
constexpr int kPyrMaxLevels = 9;  // 16^8 = 2^32 base entries

struct Pyramid {
    const uint32_t *lvl[kPyrMaxLevels];
    uint32_t len[kPyrMaxLevels];
    int nlev;
};

// Builds the upper levels over `base` (len entries) in arena memory.  With `flag` (min pyramids
// only), *flag is OR-ed with 1 if some base entry is >= flag_min.
Pyramid build_pyramid(const uint32_t *base, uint32_t len, bool is_max, Arena &arena, hipStream_t stream,
                      uint32_t flag_min = 0, uint32_t *flag = nullptr);
// The same in two steps, for a caller that computes the first level(s) itself (the candidate kernel has every
// aligned block of 16 ranks in LDS anyway): alloc_pyramid reserves the level arrays and computes nothing;
// fill_pyramid computes the levels from `first_level` up (level 0 = base).  fill_pyramid_tail computes the entries
// [first_entry, len[1]) of level 1 only (the blocks at the end of the array that such a caller leaves out).
Pyramid alloc_pyramid(const uint32_t *base, uint32_t len, Arena &arena);
void fill_pyramid(const Pyramid &P, int first_level, bool is_max, hipStream_t stream, uint32_t flag_min = 0,
                  uint32_t *flag = nullptr);
void fill_pyramid_tail(const Pyramid &P, uint32_t first_entry, bool is_max, hipStream_t stream);

template <bool kMax> __device__ __forceinline__ bool pyr_hit(uint32_t v, uint32_t x) {
    return kMax ? (v > x) : (v < x);
}
template <bool kMax> __device__ __forceinline__ uint32_t pyr_op(uint32_t a, uint32_t b) {
    return kMax ? (a > b ? a : b) : (a < b ? a : b);
}

// The queries read whole aligned 16-entry blocks (one 64-byte line, four 16-byte loads in flight
// together) and decide in registers: walking a block entry by entry makes every step a dependent
// load, and the walks of the far / exact / factor kernels are nothing but such chains.  Every level
// array (and the base arrays, which come from the arena) is 256-byte aligned and padded, so a block
// that starts below len can be read in full; entries at or beyond len are masked out.
struct PyrBlock {
    uint32_t v[kPyrFan];
};
__device__ __forceinline__ PyrBlock pyr_load_block(const uint32_t *__restrict__ A, uint32_t block_start) {
    PyrBlock B;
    const uint4 *p = reinterpret_cast<const uint4 *>(A + block_start);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint4 x = p[k];
        B.v[4 * k] = x.x;
        B.v[4 * k + 1] = x.y;
        B.v[4 * k + 2] = x.z;
        B.v[4 * k + 3] = x.w;
    }
    return B;
}
// bit j set iff entry block_start + j exists (< len) and its value is < x (kMax: > x)
template <bool kMax>
__device__ __forceinline__ uint32_t pyr_hit_mask(const PyrBlock &B, uint32_t block_start, uint32_t len, uint32_t x) {
    uint32_t m = 0;
#pragma unroll
    for (int j = 0; j < (int)kPyrFan; ++j) m |= (pyr_hit<kMax>(B.v[j], x) ? 1u : 0u) << j;
    const uint32_t left = len - block_start;
    return left >= kPyrFan ? m : (m & ((1u << left) - 1u));
}
// min (kMax: max) of the entries j0..j1 (inclusive, inside the block)
template <bool kMax>
__device__ __forceinline__ uint32_t pyr_block_reduce(const PyrBlock &B, uint32_t j0, uint32_t j1) {
    uint32_t res = kMax ? 0u : 0xffffffffu;
#pragma unroll
    for (int j = 0; j < (int)kPyrFan; ++j) {
        const bool in = (uint32_t)j >= j0 && (uint32_t)j <= j1;
        const uint32_t v = in ? B.v[j] : (kMax ? 0u : 0xffffffffu);
        res = pyr_op<kMax>(res, v);
    }
    return res;
}

// Largest q in [0, r] whose base value is < x (kMax: > x); -1 if none.
template <bool kMax>
__device__ inline int64_t pyr_nearest_left(const Pyramid &P, uint32_t r, uint32_t x) {
    int l = 0;
    uint32_t idx = r, q;
    for (;;) {
        const uint32_t bstart = idx & ~(kPyrFan - 1);
        const PyrBlock B = pyr_load_block(P.lvl[l], bstart);
        const uint32_t m = pyr_hit_mask<kMax>(B, bstart, P.len[l], x) & ((2u << (idx - bstart)) - 1u);  // entries <= idx
        if (m) {
            q = bstart + (31u - (uint32_t)__clz((int)m));
            break;
        }
        if ((idx >> kPyrShift) == 0) return -1;
        idx = (idx >> kPyrShift) - 1;
        ++l;
    }
    while (l > 0) {
        --l;
        const uint32_t base = q << kPyrShift;
        const PyrBlock B = pyr_load_block(P.lvl[l], base);
        const uint32_t m = pyr_hit_mask<kMax>(B, base, P.len[l], x);
        q = m ? base + (31u - (uint32_t)__clz((int)m)) : base;
    }
    return (int64_t)q;
}

// Smallest q in [r, len0) whose base value is < x (kMax: > x); len0 if none.
template <bool kMax>
__device__ inline uint32_t pyr_nearest_right(const Pyramid &P, uint32_t r, uint32_t x) {
    const uint32_t none = P.len[0];
    if (r >= none) return none;
    int l = 0;
    uint32_t idx = r, q;
    for (;;) {
        const uint32_t bstart = idx & ~(kPyrFan - 1);
        const PyrBlock B = pyr_load_block(P.lvl[l], bstart);
        const uint32_t m = pyr_hit_mask<kMax>(B, bstart, P.len[l], x) & ~((1u << (idx - bstart)) - 1u);  // entries >= idx
        if (m) {
            q = bstart + (uint32_t)__builtin_ctz(m);
            break;
        }
        const uint32_t nxt = (idx >> kPyrShift) + 1;
        if (l + 1 >= P.nlev || nxt >= P.len[l + 1]) return none;
        idx = nxt;
        ++l;
    }
    while (l > 0) {
        --l;
        const uint32_t c = q << kPyrShift;
        uint32_t end = c + kPyrFan - 1;
        if (end >= P.len[l]) end = P.len[l] - 1;
        const PyrBlock B = pyr_load_block(P.lvl[l], c);
        const uint32_t m = pyr_hit_mask<kMax>(B, c, P.len[l], x);
        q = m ? c + (uint32_t)__builtin_ctz(m) : end;
    }
    return q;
}

// min (kMax: max) of base[a..b], a <= b.
template <bool kMax> __device__ inline uint32_t pyr_range(const Pyramid &P, uint32_t a, uint32_t b) {
    uint32_t res = kMax ? 0u : 0xffffffffu;
    int l = 0;
    for (;;) {
        const uint32_t *A = P.lvl[l];
        if (b - a < 2 * kPyrFan) {  // at most three blocks
            for (uint32_t bs = a & ~(kPyrFan - 1); bs <= b; bs += kPyrFan) {
                const PyrBlock B = pyr_load_block(A, bs);
                const uint32_t j0 = a > bs ? a - bs : 0u;
                const uint32_t j1 = b - bs < kPyrFan - 1 ? b - bs : kPyrFan - 1;
                res = pyr_op<kMax>(res, pyr_block_reduce<kMax>(B, j0, j1));
            }
            return res;
        }
        const uint32_t a_up = (a + kPyrFan - 1) >> kPyrShift;
        const uint32_t b_up = (b + 1) >> kPyrShift;
        if (a & (kPyrFan - 1)) {  // entries a .. end of a's block
            const uint32_t bs = a & ~(kPyrFan - 1);
            res = pyr_op<kMax>(res, pyr_block_reduce<kMax>(pyr_load_block(A, bs), a - bs, kPyrFan - 1));
        }
        if ((b + 1) & (kPyrFan - 1)) {  // start of b's block .. b
            const uint32_t bs = b & ~(kPyrFan - 1);
            res = pyr_op<kMax>(res, pyr_block_reduce<kMax>(pyr_load_block(A, bs), 0u, b - bs));
        }
        a = a_up;
        b = b_up - 1;
        ++l;
    }
}

}  // namespace nolzss
