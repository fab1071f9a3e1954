// nearest.hpp -- "nearest qualifying suffix in rank order" searches shared by the plain and the
// reverse-complement candidate kernels.
//
// From rank r, walk towards smaller (up) or larger (down) ranks until a suffix start SA[q]
// qualifies (kGreater ? SA[q] > x : SA[q] < x), carrying the running minimum of the LCP values
// crossed, i.e. the LCP between suffix SA[r] and suffix SA[q].  The first kNeighbourSteps (4) ranks
// are read directly (neighbouring lanes read neighbouring entries: L1/L2-served); if nothing
// qualifies there the search continues in the min/max pyramid over SA and the LCP minimum of the
// skipped stretch comes from the LCP pyramid.
#pragma once
#include "pyramid.hpp"

namespace nolzss {

constexpr uint32_t kNeighbourSteps = 4;
constexpr uint32_t kNoPos = 0xffffffffu;

template <bool kGreater> __device__ __forceinline__ bool qualifies(uint32_t v, uint32_t x) {
    return kGreater ? (v > x) : (v < x);
}

// towards rank 0.  len = 0 / pos = kNoPos if no qualifying suffix shares a symbol with SA[r].
// `floor`: give up as soon as the running LCP minimum drops below it (the other direction
// already found something at least that long).
template <bool kGreater>
__device__ inline void nearest_up(const uint32_t *__restrict__ sa, const uint32_t *__restrict__ lcp,
                                  const Pyramid &Pv, const Pyramid &Plcp, uint32_t r, uint32_t x,
                                  uint32_t floor, uint32_t &len, uint32_t &pos) {
    len = 0;
    pos = kNoPos;
    uint32_t m = 0xffffffffu;
    for (uint32_t s = 1; s <= kNeighbourSteps; ++s) {
        if (r < s) return;
        const uint32_t q = r - s;
        const uint32_t c = lcp[q + 1];
        m = c < m ? c : m;
        if (m == 0 || m < floor) return;
        const uint32_t v = sa[q];
        if (qualifies<kGreater>(v, x)) {
            len = m;
            pos = v;
            return;
        }
    }
    if (r <= kNeighbourSteps) return;
    const int64_t q = pyr_nearest_left<kGreater>(Pv, r - kNeighbourSteps - 1, x);
    if (q < 0) return;
    const uint32_t mm = pyr_range<false>(Plcp, (uint32_t)q + 1, r - kNeighbourSteps);
    m = mm < m ? mm : m;
    if (m == 0 || m < floor) return;
    len = m;
    pos = sa[q];
}

// towards rank n-1
template <bool kGreater>
__device__ inline void nearest_down(const uint32_t *__restrict__ sa, const uint32_t *__restrict__ lcp, uint32_t n,
                                    const Pyramid &Pv, const Pyramid &Plcp, uint32_t r, uint32_t x,
                                    uint32_t floor, uint32_t &len, uint32_t &pos) {
    len = 0;
    pos = kNoPos;
    uint32_t m = 0xffffffffu;
    for (uint32_t s = 1; s <= kNeighbourSteps; ++s) {
        const uint64_t q = (uint64_t)r + s;
        if (q >= n) return;
        const uint32_t c = lcp[q];
        m = c < m ? c : m;
        if (m == 0 || m < floor) return;
        const uint32_t v = sa[q];
        if (qualifies<kGreater>(v, x)) {
            len = m;
            pos = v;
            return;
        }
    }
    if ((uint64_t)r + kNeighbourSteps + 1 >= n) return;
    const uint32_t q = pyr_nearest_right<kGreater>(Pv, r + kNeighbourSteps + 1, x);
    if (q >= n) return;
    const uint32_t mm = pyr_range<false>(Plcp, r + kNeighbourSteps + 1, q);
    m = mm < m ? mm : m;
    if (m == 0 || m < floor) return;
    len = m;
    pos = sa[q];
}

// The same two searches for a rank whose search on an LDS tile (nearest_lds.hpp) left the tile's
// reach: nothing qualifies within kFarCleared ranks, so the pyramids take over right behind them
// (no neighbour-by-neighbour prologue).
constexpr uint32_t kFarCleared = 252;  // ranks a search has passed before it is marked far

template <bool kGreater>
__device__ __forceinline__ void far_up(const uint32_t *__restrict__ sa, const Pyramid &Pv, const Pyramid &Plcp,
                                       uint32_t r, uint32_t x, uint32_t floor, uint32_t &len, uint32_t &pos) {
    len = 0;
    pos = kNoPos;
    if (r <= kFarCleared) return;
    const int64_t q = pyr_nearest_left<kGreater>(Pv, r - kFarCleared - 1, x);
    if (q < 0) return;
    const uint32_t m = pyr_range<false>(Plcp, (uint32_t)q + 1, r);
    if (m == 0 || m < floor) return;
    len = m;
    pos = sa[q];
}
template <bool kGreater>
__device__ __forceinline__ void far_down(const uint32_t *__restrict__ sa, uint32_t n, const Pyramid &Pv,
                                         const Pyramid &Plcp, uint32_t r, uint32_t x, uint32_t floor, uint32_t &len,
                                         uint32_t &pos) {
    len = 0;
    pos = kNoPos;
    if ((uint64_t)r + kFarCleared + 1 >= n) return;
    const uint32_t q = pyr_nearest_right<kGreater>(Pv, r + kFarCleared + 1, x);
    if (q >= n) return;
    const uint32_t m = pyr_range<false>(Plcp, r + 1, q);
    if (m == 0 || m < floor) return;
    len = m;
    pos = sa[q];
}

// I(d) around rank r: [lo, hi] with lcp[lo] < d, lcp[lo+1..hi] >= d, lcp[hi+1] < d.
// Relies on lcp[0] = 0 and lcp[n] = 0 (d >= 1).
__device__ inline void lcp_interval(const Pyramid &Plcp, uint32_t r, uint32_t d, uint32_t &lo, uint32_t &hi) {
    lo = (uint32_t)pyr_nearest_left<false>(Plcp, r, d);
    hi = pyr_nearest_right<false>(Plcp, r + 1, d) - 1;
}

// g(d) = min SA[I(d)]: the leftmost occurrence of the d symbols at position i (r = its rank); g never
// decreases as d grows.  P(d): g(d) + d <= i   (monotone: true for d implies true for d - 1)
__device__ __forceinline__ uint32_t lpnf_leftmost(const Pyramid &Psa, const Pyramid &Plcp, uint32_t r, uint32_t d) {
    uint32_t lo, hi;
    lcp_interval(Plcp, r, d, lo, hi);
    return pyr_range<false>(Psa, lo, hi);
}
__device__ __forceinline__ bool lpnf_pred(const Pyramid &Psa, const Pyramid &Plcp, uint32_t r, uint32_t i,
                                          uint32_t d) {
    return (uint64_t)lpnf_leftmost(Psa, Plcp, r, d) + d <= i;
}

// max{ d in [lo, cap] : P(d) } given that P(lo) holds.
// Every evaluation with P(d) true also bounds the answer from above: L* >= d implies g(L*) >= g(d), hence
// L* <= i - g(d).  Where the search is needed at all -- the best earlier match overlaps position i, i.e. the
// text is periodic around i -- the leftmost occurrence is the same for a long range of d (the start of the
// periodic run), so the bound i - g(lo) is usually the answer itself and the SECOND evaluation confirms it;
// galloping up from lo and bisecting took 2 * log2(L*) evaluations (50 on a long run, each two descents of
// the LCP pyramid and a range minimum of the SA pyramid from global memory).
__device__ inline uint32_t lpnf_search(const Pyramid &Psa, const Pyramid &Plcp, uint32_t r, uint32_t i,
                                       uint32_t lo, uint32_t cap) {
    uint32_t hi = cap;
    if (lo >= hi) return lo;
    {
        const uint32_t g = lpnf_leftmost(Psa, Plcp, r, lo < 1u ? 1u : lo);  // (I(0) is everything; lo = 0 only bounds from below)
        const uint32_t bound = i - g;  // g <= i: P(lo) holds (for lo = 0: the occurrence at i itself is in I(1))
        hi = bound < hi ? bound : hi;
    }
    while (lo < hi) {
        // the upper end first: true there ends the search
        const uint32_t gh = lpnf_leftmost(Psa, Plcp, r, hi);
        if ((uint64_t)gh + hi <= i) return hi;
        --hi;
        if (lo >= hi) break;
        const uint32_t mid = lo + (hi - lo + 1) / 2;
        const uint32_t gm = lpnf_leftmost(Psa, Plcp, r, mid);
        if ((uint64_t)gm + mid <= i) {
            lo = mid;
            const uint32_t bound = i - gm;
            hi = bound < hi ? bound : hi;
        } else {
            hi = mid - 1;
        }
    }
    return lo;
}

}  // namespace nolzss
