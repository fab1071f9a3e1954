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
#include "nearest.hpp"

namespace nolzss {
namespace {

constexpr int kThreads = 256;

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
        uint32_t lp, jp, ls, js;
        nearest_up<false>(sa, lcp, Psa, Plcp, r, i, 0u, lp, jp);
        nearest_down<false>(sa, lcp, n, Psa, Plcp, r, i, lp, ls, js);  // cannot beat lp below lp
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

__global__ __launch_bounds__(kThreads) void lpnf_fallback_kernel(const uint32_t *__restrict__ queue,
                                                                 uint32_t count, uint32_t n,
                                                                 const uint32_t *__restrict__ isa,
                                                                 Pyramid Psa, Pyramid Plcp,
                                                                 uint32_t *__restrict__ lstar) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += stride) {
        const uint32_t i = queue[k];
        const uint32_t cap = (n - i) < i ? (n - i) : i;  // L* <= i - j <= i and L* <= n - i
        lstar[i] = lpnf_search(Psa, Plcp, isa[i], i, lstar[i], cap);  // P(lstar[i]) holds on entry
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
