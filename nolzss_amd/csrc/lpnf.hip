// lpnf.hip -- longest previous NON-overlapping factor length L*[i] for every text position.
//
// Semantics restated from detail::nolzss (/root/reference/src/cpp/factorizer_core.hpp:66-109):
// the reference walks the ancestors v of leaf(i) top-down while min SA[v] + depth(v) - 1 < i
// (:75) and resolves the first failing node by cases A/B/C (:82-107).  In array terms
// (DESIGN.md section 3):  with r = ISA[i] and I(d) the maximal rank interval around r whose
// LCP values are >= d,   L*[i] = max{ d : min SA[I(d)] + d <= i },   0 meaning "literal".
//
// Data-parallel evaluation, one thread per RANK r; SA / LCP tiles are staged in LDS
// (nearest_lds.hpp) and ranks whose search leaves the tile's reach go to a compacted second
// kernel that walks the pyramids from global memory (nearest.hpp):
//   1. nearest smaller SA value above and below r (the classic LPF candidates) with the
//      running LCP minimum -> M = max lcp over all earlier suffixes;
//   2. if a candidate j with lcp M also satisfies i - j >= M it does not overlap, so L* = M;
//   3. otherwise (periodic regions: the best match runs into position i) the position is queued
//      and lpnf_fallback_kernel finds max d by galloping + binary search on the monotone
//      predicate, using the LCP pyramid for I(d) and the SA pyramid for the range minimum.
#include "pipeline.hpp"

#include <cstdlib>
#include "nearest_lds.hpp"
#include "radix_sort.hpp"

namespace nolzss {
namespace {

constexpr int kThreads = 256;
constexpr uint32_t kFarBothUnknown = 0xfffffffeu;  // far_aux: neither direction ended inside the tile

// turn the two neighbour candidates into L*[i], or queue i for the exact search
// (*dst receives L*[i] or, for queued positions, a lower bound with P(bound) true)
__device__ __forceinline__ void lpf_decide(uint32_t i, uint32_t lp, uint32_t jp, uint32_t ls, uint32_t js,
                                           uint32_t *__restrict__ dst, uint32_t *__restrict__ queue,
                                           uint32_t *__restrict__ queue_count) {
    const uint32_t M = lp > ls ? lp : ls;
    if (M == 0) {
        *dst = 0;
        return;
    }
    const bool ok = (lp == M && i - jp >= M) || (ls == M && i - js >= M);
    if (ok) {
        *dst = M;
        return;
    }
    // best earlier match overlaps position i: exact search needed; record a true lower bound
    uint32_t lo = 0;
    if (lp > 0) { const uint32_t c = lp < i - jp ? lp : i - jp; lo = c > lo ? c : lo; }
    if (ls > 0) { const uint32_t c = ls < i - js ? ls : i - js; lo = c > lo ? c : lo; }
    *dst = lo;
    queue[atomicAdd(queue_count, 1u)] = i;
}

__global__ __launch_bounds__(kLdsThreads) void lpf_tile_kernel(const uint32_t *__restrict__ sa,
                                                               const uint32_t *__restrict__ lcp, uint32_t n,
                                                               uint32_t *__restrict__ lstar_by_rank,
                                                               uint32_t *__restrict__ queue,
                                                               uint32_t *__restrict__ queue_count,
                                                               uint32_t *__restrict__ far_queue,
                                                               uint32_t *__restrict__ far_count,
                                                               uint32_t *__restrict__ far_aux) {
    constexpr int NS = 2;
    __shared__ uint32_t s_sa[kLdsSpan];
    __shared__ uint32_t s_lcp[kLdsSpan + 1];
    __shared__ uint32_t s_len[NS * kLdsTile];
    __shared__ uint32_t s_pos[NS * kLdsTile];
    __shared__ uint16_t s_list[kLdsWaves][2][NS * kLdsPerWave];
    const uint32_t base = blockIdx.x * (uint32_t)kLdsTile;
    stage_tile(sa, lcp, n, base, s_sa, s_lcp);
    __syncthreads();
    const int w = threadIdx.x >> 6;
    const uint32_t far_bit = n <= 0x80000000u ? 0x80000000u : 0u;
    lds_search_wave<NS, NS, 1>(s_sa, s_lcp, n, base, s_len, s_pos, s_list[w][0], s_list[w][1],
                            [](uint32_t) { return true; }, [](uint32_t) { return 0u; }, far_bit);
    // ranks with a search beyond the reach go to the far queue: one atomic per wavefront
    uint64_t far_mask[kLdsPerWave / 64];
    uint32_t far_total = 0;
#pragma unroll
    for (int row = 0; row < kLdsPerWave / 64; ++row) {
        const int t = w * kLdsPerWave + row * 64 + lane_id();
        // a search that left the reach matters only if its bound can beat the other direction
        uint32_t up = s_len[t], down = s_len[kLdsTile + t];
        const bool far = far_resolve(up, down, far_bit, 0u) && (uint64_t)base + t < n;
        s_len[t] = up;
        s_len[kLdsTile + t] = down;
        far_mask[row] = __ballot(far);
        far_total += (uint32_t)__popcll(far_mask[row]);
    }
    uint32_t far_base = 0;
    if (far_total) {
        if (lane_id() == 0) far_base = atomicAdd(far_count, far_total);
        far_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)far_base);
    }
#pragma unroll
    for (int row = 0; row < kLdsPerWave / 64; ++row) {
        const int t = w * kLdsPerWave + row * 64 + lane_id();
        const uint64_t rr = (uint64_t)base + t;
        const bool far = (far_mask[row] >> lane_id()) & 1;
        if (far) far_queue[far_base + (uint32_t)__popcll(far_mask[row] & lanemask_lt())] = (uint32_t)rr;
        far_base += (uint32_t)__popcll(far_mask[row]);
        if (rr >= n) continue;
        if (far) {
            // finished from global memory by lpf_far_kernel; what the direction that did end inside
            // the tile found travels along (length in the by-rank slot, position + side in far_aux)
            const uint32_t up = s_len[t], down = s_len[kLdsTile + t];
            const bool fu = far_is(up, far_bit), fd = far_is(down, far_bit);
            uint32_t known_len = 0, aux = kFarBothUnknown;
            if (far_bit && fu != fd) {
                known_len = fu ? down : up;
                const uint32_t p = fu ? s_pos[kLdsTile + t] : s_pos[t];
                aux = (known_len ? (p & 0x7fffffffu) : 0x7fffffffu) | (fu ? 0x80000000u : 0u);
            }
            lstar_by_rank[rr] = known_len;
            far_aux[rr] = aux;
            continue;
        }
        lpf_decide(s_sa[t + kLdsReach], s_len[t], s_pos[t], s_len[kLdsTile + t], s_pos[kLdsTile + t],
                   lstar_by_rank + rr, queue, queue_count);
    }
}

// ranks whose nearest earlier suffix lies outside the LDS reach: pyramid search
__global__ __launch_bounds__(kThreads) void lpf_far_kernel(const uint32_t *__restrict__ far_queue, uint32_t count,
                                                           const uint32_t *__restrict__ sa,
                                                           const uint32_t *__restrict__ lcp, uint32_t n,
                                                           Pyramid Psa, Pyramid Plcp,
                                                           const uint32_t *__restrict__ by_rank,
                                                           const uint32_t *__restrict__ far_aux, bool bounded,
                                                           uint32_t *__restrict__ lstar,
                                                           uint32_t *__restrict__ queue,
                                                           uint32_t *__restrict__ queue_count) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += stride) {
        const uint32_t r = far_queue[k];
        const uint32_t i = sa[r];
        const uint32_t aux = far_aux[r];
        uint32_t lp, jp, ls, js;
        if (!bounded) {  // (texts beyond 2^31 symbols: no bounds were kept)
            nearest_up<false>(sa, lcp, Psa, Plcp, r, i, 0u, lp, jp);
            nearest_down<false>(sa, lcp, n, Psa, Plcp, r, i, lp, ls, js);  // cannot beat lp below lp
        } else if (aux == kFarBothUnknown) {
            far_up<false>(sa, Psa, Plcp, r, i, 0u, lp, jp);
            far_down<false>(sa, n, Psa, Plcp, r, i, lp, ls, js);
        } else if (aux >> 31) {  // the search towards larger ranks ended inside the tile
            ls = by_rank[r];
            js = (aux & 0x7fffffffu) == 0x7fffffffu ? kNoPos : (aux & 0x7fffffffu);
            far_up<false>(sa, Psa, Plcp, r, i, ls, lp, jp);
        } else {
            lp = by_rank[r];
            jp = (aux & 0x7fffffffu) == 0x7fffffffu ? kNoPos : (aux & 0x7fffffffu);
            far_down<false>(sa, n, Psa, Plcp, r, i, lp, ls, js);
        }
        lpf_decide(i, lp, jp, ls, js, lstar + i, queue, queue_count);
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
        lstar[i] = lpnf_search(Psa, Plcp, isa[i] - 1u, i, lstar[i], cap);  // P(lstar[i]) holds on entry
    }
}

}  // namespace

uint32_t build_lstar(Context &ctx, uint32_t n, const uint32_t *sa, const uint32_t *isa, const uint32_t *lcp,
                     const Pyramid &Psa, const Pyramid &Plcp, uint32_t *lstar) {
    hipStream_t s = ctx.stream;
    const size_t mark = ctx.arena.mark();
    uint32_t *queue = ctx.arena.alloc<uint32_t>(n);
    uint32_t *far_queue = ctx.arena.alloc<uint32_t>(n);
    uint32_t *counts = ctx.arena.alloc<uint32_t>(2);  // [0] exact-search queue, [1] far queue
    uint32_t *by_rank = ctx.arena.alloc<uint32_t>(n);
    uint32_t *far_aux = ctx.arena.alloc<uint32_t>(n);
    uint32_t *scratch_idx = ctx.arena.alloc<uint32_t>(n);
    uint32_t *scratch_val = ctx.arena.alloc<uint32_t>(n);
    HIP_CHECK(hipMemsetAsync(counts, 0, 2 * sizeof(uint32_t), s));
    {
        ProfScope ps(ctx.profiler(), "lpf", s, 12.0 * (double)n);
        lpf_tile_kernel<<<(unsigned)div_up(n, kLdsTile), kLdsThreads, 0, s>>>(sa, lcp, n, by_rank, queue, counts,
                                                                          far_queue, counts + 1, far_aux);
        KERNEL_CHECK();
    }
    {
        // lstar[sa[r]] = by_rank[r]: rank order -> text order (a permutation scatter)
        ProfScope ps(ctx.profiler(), "lpf_to_text_order", s);
        uint32_t *idx[2] = {const_cast<uint32_t *>(sa), scratch_idx};
        uint32_t *val[2] = {by_rank, scratch_val};
        bucketed_scatter(idx, val, n, lstar, n, ctx.arena, s, ctx.profiler(), true);
    }
    uint32_t h[2] = {0, 0};
    ctx.read_back(counts, h, 2);
    static const bool trace = getenv("NOLZSS_TRACE") != nullptr;
    if (trace) fprintf(stderr, "[nolzss] lpf: %u ranks to the far queue, %u positions to the exact search so far\n", h[1], h[0]);
    if (h[1] > 0) {
        ProfScope ps(ctx.profiler(), "lpf_far", s);
        size_t g = div_up(h[1], kThreads);
        if (g > 256u * 32u) g = 256u * 32u;
        lpf_far_kernel<<<(unsigned)g, kThreads, 0, s>>>(far_queue, h[1], sa, lcp, n, Psa, Plcp, by_rank, far_aux,
                                                        n <= 0x80000000u, lstar, queue, counts);
        KERNEL_CHECK();
        ctx.read_back(counts, h, 1);
    }
    if (h[0] > 0) {
        ProfScope ps(ctx.profiler(), "lpnf_fallback", s);
        size_t g = div_up(h[0], kThreads);
        if (g > 256u * 32u) g = 256u * 32u;
        lpnf_fallback_kernel<<<(unsigned)g, kThreads, 0, s>>>(queue, h[0], n, isa, Psa, Plcp, lstar);
        KERNEL_CHECK();
    }
    ctx.arena.rewind(mark);
    return h[0];
}

}  // namespace nolzss
