// lpnf.hip -- longest previous NON-overlapping factor length L*[i] for every text position.
//
// Semantics restated from detail::nolzss (/root/reference/src/cpp/factorizer_core.hpp:66-109):
// the reference walks the ancestors v of leaf(i) top-down while min SA[v] + depth(v) - 1 < i
// (:75) and resolves the first failing node by cases A/B/C (:82-107).  In array terms
// (DESIGN.md section 3):  with r = ISA[i] and I(d) the maximal rank interval around r whose
// LCP values are >= d,   L*[i] = max{ d : min SA[I(d)] + d <= i },   0 meaning "literal".
//
// Data-parallel evaluation, one thread per RANK r (coalesced SA / LCP reads):
//   1. nearest smaller SA value above and below r (the classic LPF candidates) with the
//      running LCP minimum -> M = max lcp over all earlier suffixes;
//   2. if a candidate j with lcp M also satisfies i - j >= M it does not overlap, so L* = M;
//   3. otherwise (periodic regions: the best match runs into position i) the position is queued
//      and lpnf_fallback_kernel finds max d by galloping + binary search on the monotone
//      predicate, using the LCP pyramid for I(d) and the SA pyramid for the range minimum.
#include "pipeline.hpp"
#include "pyramid.hpp"

namespace nolzss {
namespace {

constexpr int kThreads = 256;
constexpr uint32_t kNeighbourSteps = 32;  // direct steps before switching to the pyramid
constexpr uint32_t kNone = 0xffffffffu;

__global__ __launch_bounds__(kThreads) void lpf_kernel(const uint32_t *__restrict__ sa,
                                                       const uint32_t *__restrict__ lcp, uint32_t n,
                                                       Pyramid Psa, Pyramid Plcp,
                                                       uint32_t *__restrict__ lstar,
                                                       uint32_t *__restrict__ queue,
                                                       uint32_t *__restrict__ queue_count) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t rr = (size_t)blockIdx.x * blockDim.x + threadIdx.x; rr < n; rr += stride) {
        const uint32_t r = (uint32_t)rr;
        const uint32_t i = sa[r];

        // ---- nearest rank above r holding an earlier suffix --------------------------------
        uint32_t lp = 0, jp = kNone;
        {
            uint32_t m = kNone;
            bool done = false;
            for (uint32_t s = 1; s <= kNeighbourSteps; ++s) {
                if (r < s) { done = true; break; }
                const uint32_t q = r - s;
                const uint32_t c = lcp[q + 1];
                m = c < m ? c : m;
                if (m == 0) { done = true; break; }
                const uint32_t v = sa[q];
                if (v < i) { lp = m; jp = v; done = true; break; }
            }
            if (!done && r > kNeighbourSteps) {
                const int64_t q = pyr_nearest_left<false>(Psa, r - kNeighbourSteps - 1, i);
                if (q >= 0) {
                    const uint32_t mm = pyr_range<false>(Plcp, (uint32_t)q + 1, r - kNeighbourSteps);
                    m = mm < m ? mm : m;
                    if (m > 0) { lp = m; jp = sa[q]; }
                }
            }
        }
        // ---- nearest rank below r holding an earlier suffix --------------------------------
        uint32_t ls = 0, js = kNone;
        {
            uint32_t m = kNone;
            bool done = false;
            for (uint32_t s = 1; s <= kNeighbourSteps; ++s) {
                const uint32_t q = r + s;
                if (q >= n) { done = true; break; }
                const uint32_t c = lcp[q];
                m = c < m ? c : m;
                if (m == 0 || m < lp) { done = true; break; }  // cannot beat the upper candidate
                const uint32_t v = sa[q];
                if (v < i) { ls = m; js = v; done = true; break; }
            }
            if (!done && (uint64_t)r + kNeighbourSteps + 1 < n) {
                const uint32_t q = pyr_nearest_right<false>(Psa, r + kNeighbourSteps + 1, i);
                if (q < n) {
                    const uint32_t mm = pyr_range<false>(Plcp, r + kNeighbourSteps + 1, q);
                    m = mm < m ? mm : m;
                    if (m > 0) { ls = m; js = sa[q]; }
                }
            }
        }
        const uint32_t M = lp > ls ? lp : ls;
        if (M == 0) {
            lstar[i] = 0;
            continue;
        }
        const bool ok = (lp == M && i - jp >= M) || (ls == M && i - js >= M);
        if (ok) {
            lstar[i] = M;
            continue;
        }
        // best earlier match overlaps position i: exact search needed; record a true lower bound
        uint32_t lo = 0;
        if (lp > 0) { const uint32_t c = lp < i - jp ? lp : i - jp; lo = c > lo ? c : lo; }
        if (ls > 0) { const uint32_t c = ls < i - js ? ls : i - js; lo = c > lo ? c : lo; }
        lstar[i] = lo;
        queue[atomicAdd(queue_count, 1u)] = i;
    }
}

// P(d): min SA[I(d)] + d <= i   (monotone: true for d implies true for d-1)
__device__ __forceinline__ bool lpnf_pred(const Pyramid &Psa, const Pyramid &Plcp, uint32_t r, uint32_t i,
                                          uint32_t d) {
    const uint32_t lo = (uint32_t)pyr_nearest_left<false>(Plcp, r, d);        // lcp[0] = 0 < d
    const uint32_t hi = pyr_nearest_right<false>(Plcp, r + 1, d) - 1;         // lcp[n] = 0 < d
    const uint32_t mn = pyr_range<false>(Psa, lo, hi);
    return (uint64_t)mn + d <= i;
}

__global__ __launch_bounds__(kThreads) void lpnf_fallback_kernel(const uint32_t *__restrict__ queue,
                                                                 uint32_t count, uint32_t n,
                                                                 const uint32_t *__restrict__ isa,
                                                                 Pyramid Psa, Pyramid Plcp,
                                                                 uint32_t *__restrict__ lstar) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += stride) {
        const uint32_t i = queue[k];
        const uint32_t r = isa[i];
        uint32_t lo = lstar[i];                       // P(lo) holds
        const uint32_t cap = (n - i) < i ? (n - i) : i;  // L* <= i - j <= i and L* <= n - i
        uint32_t hi = cap;
        // gallop up from the known-good lower bound
        uint32_t step = 1;
        while (lo < hi) {
            uint32_t d = lo + step;
            if (d > hi || d < lo) d = hi;
            if (lpnf_pred(Psa, Plcp, r, i, d)) {
                lo = d;
                step <<= 1;
            } else {
                hi = d - 1;
                break;
            }
        }
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo + 1) / 2;
            if (lpnf_pred(Psa, Plcp, r, i, mid))
                lo = mid;
            else
                hi = mid - 1;
        }
        lstar[i] = lo;
    }
}

}  // namespace

uint32_t build_lstar(Context &ctx, uint32_t n, const uint32_t *sa, const uint32_t *isa, const uint32_t *lcp,
                     const Pyramid &Psa, const Pyramid &Plcp, uint32_t *lstar) {
    hipStream_t s = ctx.stream;
    const size_t mark = ctx.arena.mark();
    uint32_t *queue = ctx.arena.alloc<uint32_t>(n);
    uint32_t *count = ctx.arena.alloc<uint32_t>(1);
    HIP_CHECK(hipMemsetAsync(count, 0, sizeof(uint32_t), s));
    {
        ProfScope ps(ctx.profiler(), "lpf", s);
        size_t g = div_up(n, kThreads);
        if (g > 256u * 32u) g = 256u * 32u;
        lpf_kernel<<<(unsigned)g, kThreads, 0, s>>>(sa, lcp, n, Psa, Plcp, lstar, queue, count);
        KERNEL_CHECK();
    }
    uint32_t h_count = 0;
    ctx.read_back(count, &h_count, 1);
    if (h_count > 0) {
        ProfScope ps(ctx.profiler(), "lpnf_fallback", s);
        size_t g = div_up(h_count, kThreads);
        if (g > 256u * 32u) g = 256u * 32u;
        lpnf_fallback_kernel<<<(unsigned)g, kThreads, 0, s>>>(queue, h_count, n, isa, Psa, Plcp, lstar);
        KERNEL_CHECK();
    }
    ctx.arena.rewind(mark);
    return h_count;
}

}  // namespace nolzss
