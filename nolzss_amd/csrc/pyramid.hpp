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

template <bool kMax> __device__ __forceinline__ bool pyr_hit(uint32_t v, uint32_t x) {
    return kMax ? (v > x) : (v < x);
}
template <bool kMax> __device__ __forceinline__ uint32_t pyr_op(uint32_t a, uint32_t b) {
    return kMax ? (a > b ? a : b) : (a < b ? a : b);
}

// Largest q in [0, r] whose base value is < x (kMax: > x); -1 if none.
template <bool kMax>
__device__ inline int64_t pyr_nearest_left(const Pyramid &P, uint32_t r, uint32_t x) {
    int l = 0;
    uint32_t idx = r, q;
    for (;;) {
        const uint32_t bstart = idx & ~(kPyrFan - 1);
        const uint32_t *A = P.lvl[l];
        bool found = false;
        for (q = idx;; --q) {
            if (pyr_hit<kMax>(A[q], x)) {
                found = true;
                break;
            }
            if (q == bstart) break;
        }
        if (found) break;
        if ((idx >> kPyrShift) == 0) return -1;
        idx = (idx >> kPyrShift) - 1;
        ++l;
    }
    while (l > 0) {
        --l;
        const uint32_t *A = P.lvl[l];
        const uint32_t base = q << kPyrShift;
        uint32_t c = base + kPyrFan - 1;
        if (c >= P.len[l]) c = P.len[l] - 1;
        while (!pyr_hit<kMax>(A[c], x) && c > base) --c;
        q = c;
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
        uint32_t bend = idx | (kPyrFan - 1);
        if (bend >= P.len[l]) bend = P.len[l] - 1;
        const uint32_t *A = P.lvl[l];
        bool found = false;
        for (q = idx; q <= bend; ++q) {
            if (pyr_hit<kMax>(A[q], x)) {
                found = true;
                break;
            }
        }
        if (found) break;
        const uint32_t nxt = (idx >> kPyrShift) + 1;
        if (l + 1 >= P.nlev || nxt >= P.len[l + 1]) return none;
        idx = nxt;
        ++l;
    }
    while (l > 0) {
        --l;
        const uint32_t *A = P.lvl[l];
        uint32_t c = q << kPyrShift;
        uint32_t end = c + kPyrFan - 1;
        if (end >= P.len[l]) end = P.len[l] - 1;
        while (!pyr_hit<kMax>(A[c], x) && c < end) ++c;
        q = c;
    }
    return q;
}

// min (kMax: max) of base[a..b], a <= b.
template <bool kMax> __device__ inline uint32_t pyr_range(const Pyramid &P, uint32_t a, uint32_t b) {
    uint32_t res = kMax ? 0u : 0xffffffffu;
    int l = 0;
    for (;;) {
        const uint32_t *A = P.lvl[l];
        if (b - a < 2 * kPyrFan) {
            for (uint32_t q = a; q <= b; ++q) res = pyr_op<kMax>(res, A[q]);
            return res;
        }
        const uint32_t a_up = (a + kPyrFan - 1) >> kPyrShift;
        const uint32_t b_up = (b + 1) >> kPyrShift;
        for (uint32_t q = a; q < (a_up << kPyrShift); ++q) res = pyr_op<kMax>(res, A[q]);
        for (uint32_t q = (b_up << kPyrShift); q <= b; ++q) res = pyr_op<kMax>(res, A[q]);
        a = a_up;
        b = b_up - 1;
        ++l;
    }
}

}  // namespace nolzss
