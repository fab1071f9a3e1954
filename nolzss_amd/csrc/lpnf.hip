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
#include "queues.hpp"
#include "radix_sort.hpp"
#include "scan.hpp"

namespace nolzss {

__global__ void shard_totals_kernel(const uint32_t *const *counts, int nq, uint32_t *totals) {
    const int q = blockIdx.x;
    if (q >= nq) return;
    uint32_t v = threadIdx.x < kQShards ? counts[q][threadIdx.x * kQPad] : 0u;
    v = wave_reduce(v, OpAdd<uint32_t>());
    __shared__ uint32_t s_part[kQShards / 64];
    if (lane_id() == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (uint32_t k = 0; k < kQShards / 64; ++k) t += s_part[k];
        totals[q] = t;
    }
}

namespace {

constexpr int kThreads = 256;
constexpr uint32_t kFarBothUnknown = 0xfffffffeu;  // far_aux: neither direction ended inside the tile

// turn the two neighbour candidates into L*[i]; returns true if i needs the exact search
// (*dst then receives a lower bound with P(bound) true)
__device__ __forceinline__ bool lpf_decide(uint32_t i, uint32_t lp, uint32_t jp, uint32_t ls, uint32_t js,
                                           uint32_t *__restrict__ dst) {
    const uint32_t M = lp > ls ? lp : ls;
    if (M == 0) {
        *dst = 0;
        return false;
    }
    const bool ok = (lp == M && i - jp >= M) || (ls == M && i - js >= M);
    if (ok) {
        *dst = M;
        return false;
    }
    // best earlier match overlaps position i: exact search needed; record a true lower bound
    uint32_t lo = 0;
    if (lp > 0) { const uint32_t c = lp < i - jp ? lp : i - jp; lo = c > lo ? c : lo; }
    if (ls > 0) { const uint32_t c = ls < i - js ? ls : i - js; lo = c > lo ? c : lo; }
    *dst = lo;
    return true;
}

template <bool kTimed>
__global__ __launch_bounds__(kLdsThreads) void lpf_tile_kernel(const uint32_t *__restrict__ sa,
                                                               const uint32_t *__restrict__ lcp, uint32_t n,
                                                               uint32_t *__restrict__ lstar_by_rank,
                                                               ShardQueue exact_q, uint32_t *__restrict__ far_items,
                                                               uint32_t *__restrict__ far_cnt,
                                                               uint32_t *__restrict__ far_aux,
                                                               unsigned long long *__restrict__ phases,
                                                               uint32_t *__restrict__ psa1, uint32_t *__restrict__ plcp1,
                                                               uint32_t pending_min, uint32_t *__restrict__ pending_flag) {
    constexpr int NS = 2;
    // (diagnostics, NOLZSS_LPF_PHASES: cycles per phase summed over the wavefronts of every 16th workgroup)
    const bool timed = kTimed && phases != nullptr && (blockIdx.x & 15) == 0;
    unsigned long long clk[6] = {0, 0, 0, 0, 0, 0};
    if (timed) clk[0] = __builtin_readcyclecounter();
    __shared__ __align__(16) uint32_t s_sa[kLdsSpan];
    __shared__ __align__(16) uint32_t s_lcp[kLdsSpan + 4];
    __shared__ uint32_t s_len[NS * kLdsTile];
    __shared__ uint16_t s_pos[NS * kLdsTile];  // (local index of the match, nearest_lds.hpp: match_pos)
    // lists of the searches still going after 4 steps (one per rank at most) and after 16 (one in 17 is: 64 entries,
    // emptied by rounds B + C whenever the next step of round A might not fit): 30.9 KiB of LDS with the two-byte
    // match indices, five workgroups per CU instead of four
    constexpr int kListCapB = 64;
    __shared__ uint16_t s_list0[kLdsWaves][NS * kLdsPerWave];
    __shared__ uint16_t s_list1[kLdsWaves][NS * kListCapB];
    __shared__ uint32_t s_blk[3 * kBlkTableLen];
    const uint32_t base = blockIdx.x * (uint32_t)kLdsTile;
    const uint32_t shard = blockIdx.x % kQShards;
    stage_tile(sa, lcp, n, base, s_sa, s_lcp, pending_min, pending_flag);
    __syncthreads();
    const BlockTables T = block_tables<false>(s_blk);
    build_block_tables<false>(s_sa, s_lcp, T);
    __syncthreads();
    // The block tables of the tile's own ranks ARE the first level of the two pyramids the later stages query (min
    // suffix start / min LCP of every aligned block of 16 ranks): written from here, the two streaming passes over SA
    // and LCP that used to build that level are gone (pyramid.hpp, fill_pyramid from level 2; blocks that reach
    // beyond rank n - 1 are left to fill_pyramid_tail).
    if (psa1 != nullptr && threadIdx.x < kLdsTile / kBlk) {
        const uint32_t B = (uint32_t)kLdsReach / kBlk + threadIdx.x;
        const uint64_t first = (uint64_t)base + (uint64_t)threadIdx.x * kBlk;
        if (first + kBlk <= (uint64_t)n) {
            psa1[first >> 4] = T.mn[B];
            plcp1[first >> 4] = T.ldn[B];
        }
    }
    if (timed) clk[1] = __builtin_readcyclecounter();
    const int w = threadIdx.x >> 6;
    const uint32_t far_bit = n <= 0x80000000u ? 0x80000000u : 0u;
    lds_search_wave_blocks<NS, NS, false, kLdsPerWave, kListCapB>(s_sa, s_lcp, T, n, base, s_len, s_pos, s_list0[w], s_list1[w],
                                   [](uint32_t) { return true; }, [](uint32_t) { return 0u; }, far_bit,
                                   timed ? clk + 2 : nullptr);
    if (timed) clk[4] = __builtin_readcyclecounter();
    // epilogue: every rank is decided, or goes to the far queue (a search beyond the reach whose bound can
    // still beat the other direction) or to the exact-search queue -- ONE atomic per wavefront and queue
    constexpr int kRows = kLdsPerWave / 64;
    bool far[kRows], exact[kRows];
    uint32_t aux[kRows], known[kRows], fslot[kRows], eslot[kRows];
#pragma unroll
    for (int row = 0; row < kRows; ++row) {
        const int t = w * kLdsPerWave + row * 64 + lane_id();
        const uint64_t rr = (uint64_t)base + t;
        const bool in = rr < n;
        uint32_t up = s_len[t], down = s_len[kLdsTile + t];
        const bool fu = far_is(up, far_bit), fd = far_is(down, far_bit);
        far[row] = far_resolve(up, down, far_bit, 0u) && in;
        exact[row] = false;
        aux[row] = kFarBothUnknown;
        known[row] = 0;
        if (far[row]) {
            // finished from global memory by lpf_far_kernel; what the direction that did end inside
            // the tile found travels along (length in the by-rank slot, position + side in far_aux)
            if (far_bit && fu != fd) {
                known[row] = fu ? down : up;
                const uint32_t p = match_pos(s_sa, fu ? s_pos[kLdsTile + t] : s_pos[t]);
                aux[row] = (known[row] ? (p & 0x7fffffffu) : 0x7fffffffu) | (fu ? 0x80000000u : 0u);
            }
        } else if (in) {
            exact[row] = lpf_decide(s_sa[t + kLdsReach], up, match_pos(s_sa, s_pos[t]), down, match_pos(s_sa, s_pos[kLdsTile + t]),
                                    lstar_by_rank + rr);
        }
    }
    // Far ranks: every wavefront has a region of its own -- the 256 entries at its own ranks -- and reports how
    // many it used; a scan and a small kernel make one list of them (compact_far_kernel).  The returning atomic
    // on a queue counter that almost every second wavefront needed here kept the wavefront alive for a round trip
    // to memory: the epilogue took 10.8 k of a wavefront's 32 k cycles (NOLZSS_LPF_PHASES).
    {
        const uint32_t region = (blockIdx.x * (uint32_t)kLdsWaves + (uint32_t)w);
        uint32_t used = 0;
#pragma unroll
        for (int row = 0; row < kRows; ++row) {
            const uint64_t bal = __ballot(far[row]);
            fslot[row] = region * (uint32_t)kLdsPerWave + used + (uint32_t)__popcll(bal & lanemask_lt());
            used += (uint32_t)__popcll(bal);
        }
        if (lane_id() == 0) far_cnt[region] = used;
    }
    shard_slots<kRows>(exact_q, shard, exact, eslot);
#pragma unroll
    for (int row = 0; row < kRows; ++row) {
        const int t = w * kLdsPerWave + row * 64 + lane_id();
        const uint64_t rr = (uint64_t)base + t;
        if (far[row]) {
            far_items[fslot[row]] = (uint32_t)rr;
            lstar_by_rank[rr] = known[row];
            far_aux[rr] = aux[row];
        }
        if (exact[row]) exact_q.items[eslot[row]] = s_sa[t + kLdsReach];
    }
    if (timed) {
        __builtin_amdgcn_s_waitcnt(0);
        clk[5] = __builtin_readcyclecounter();
        if (lane_id() == 0) {
            for (int k = 0; k < 5; ++k) atomicAdd(phases + k, clk[k + 1] - clk[k]);
            atomicAdd(phases + 5, 1ull);
        }
    }
}

// ranks whose nearest earlier suffix lies outside the LDS reach: pyramid search.  Grid (kQShards, Y).
// the far ranks of all wavefront regions, one after the other: four threads per region
__global__ __launch_bounds__(kThreads) void compact_far_kernel(const uint32_t *__restrict__ far_items,
                                                               const uint32_t *__restrict__ far_cnt,
                                                               const uint32_t *__restrict__ far_off, uint32_t regions,
                                                               uint32_t *__restrict__ far_list) {
    const size_t stride = (size_t)gridDim.x * blockDim.x / 4;
    for (size_t r = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 4; r < regions; r += stride) {
        const uint32_t c = far_cnt[r], o = far_off[r];
        for (uint32_t k = threadIdx.x & 3u; k < c; k += 4) far_list[o + k] = far_items[r * (size_t)kLdsPerWave + k];
    }
}

// far_exact[k] = position i if the k-th far rank needs the exact search, else kNoPos.  (Not the sharded exact queue:
// a shard's region is sized for what the tiles of that shard can append, and the far list deals ranks to workgroups
// in list order, not by tile.)
__global__ __launch_bounds__(kThreads) void lpf_far_kernel(const uint32_t *__restrict__ far_list, uint32_t far_count,
                                                           const uint32_t *__restrict__ sa,
                                                           const uint32_t *__restrict__ lcp, uint32_t n,
                                                           Pyramid Psa, Pyramid Plcp,
                                                           const uint32_t *__restrict__ by_rank,
                                                           const uint32_t *__restrict__ far_aux, bool bounded,
                                                           uint32_t *__restrict__ lstar, uint32_t *__restrict__ far_exact) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < far_count; k += stride) {
        const uint32_t r = far_list[k];
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
        far_exact[k] = lpf_decide(i, lp, jp, ls, js, lstar + i) ? i : kNoPos;
    }
}

// the exact search for the far ranks that need it (far_exact, above)
__global__ __launch_bounds__(kThreads) void lpnf_fallback_list_kernel(const uint32_t *__restrict__ far_exact,
                                                                      uint32_t far_count, uint32_t n,
                                                                      const uint32_t *__restrict__ isa, Pyramid Psa,
                                                                      Pyramid Plcp, uint32_t *__restrict__ lstar) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < far_count; k += stride) {
        const uint32_t i = far_exact[k];
        if (i == kNoPos) continue;
        const uint32_t cap = (n - i) < i ? (n - i) : i;  // L* <= i - j <= i and L* <= n - i
        lstar[i] = lpnf_search(Psa, Plcp, isa[i] - 1u, i, lstar[i], cap);  // P(lstar[i]) holds on entry
    }
}

__global__ __launch_bounds__(kThreads) void lpnf_fallback_kernel(ShardQueue exact_q, uint32_t n,
                                                                 const uint32_t *__restrict__ isa,
                                                                 Pyramid Psa, Pyramid Plcp,
                                                                 uint32_t *__restrict__ lstar) {
    const uint32_t shard = blockIdx.x;
    const uint32_t count = exact_q.counts[shard * kQPad];
    const uint32_t *items = exact_q.items + (size_t)shard * exact_q.cap;
    for (uint32_t k = blockIdx.y * blockDim.x + threadIdx.x; k < count; k += gridDim.y * blockDim.x) {
        const uint32_t i = items[k];
        const uint32_t cap = (n - i) < i ? (n - i) : i;  // L* <= i - j <= i and L* <= n - i
        lstar[i] = lpnf_search(Psa, Plcp, isa[i] - 1u, i, lstar[i], cap);  // P(lstar[i]) holds on entry
    }
}

}  // namespace

uint32_t build_lstar(Context &ctx, uint32_t n, const uint32_t *sa, const uint32_t *isa, const uint32_t *lcp,
                     const Pyramid &Psa, const Pyramid &Plcp, uint32_t *lstar, uint32_t *isa_fill,
                     const PackedText *fill_pyramids) {
    hipStream_t s = ctx.stream;
    const size_t mark = ctx.arena.mark();
    // fill_pyramids: Psa / Plcp are allocated but empty (alloc_pyramid): the tile kernel writes their first level,
    // the upper levels are filled behind it, and the check for undecided LCP entries happens on the way
    uint32_t *pending_flag = nullptr;
    if (fill_pyramids) {
        pending_flag = ctx.arena.alloc<uint32_t>(1);
        HIP_CHECK(hipMemsetAsync(pending_flag, 0, sizeof(uint32_t), s));
        inject_pending_for_test(ctx, const_cast<uint32_t *>(lcp), n);
    }
    uint32_t *psa1 = fill_pyramids && Psa.nlev > 1 ? const_cast<uint32_t *>(Psa.lvl[1]) : nullptr;
    uint32_t *plcp1 = fill_pyramids && Plcp.nlev > 1 ? const_cast<uint32_t *>(Plcp.lvl[1]) : nullptr;
    if (!psa1 || !plcp1) psa1 = plcp1 = nullptr;
    const unsigned tiles = (unsigned)div_up(n, kLdsTile);
    ShardQueue exact_q;
    exact_q.cap = (uint32_t)shard_queue_cap(tiles, kLdsTile);
    exact_q.items = ctx.arena.alloc<uint32_t>((size_t)kQShards * exact_q.cap);
    // far ranks: one region per wavefront of the tile kernel (its own 256 ranks), then one list
    const uint32_t far_regions = tiles * (uint32_t)kLdsWaves;
    uint32_t *far_items = ctx.arena.alloc<uint32_t>((size_t)far_regions * kLdsPerWave);
    uint32_t *far_cnt = ctx.arena.alloc<uint32_t>(far_regions);
    uint32_t *far_off = ctx.arena.alloc<uint32_t>(far_regions);
    uint32_t *qcounts = ctx.arena.alloc<uint32_t>(kQShards * kQPad);
    exact_q.counts = qcounts;
    const uint32_t **count_ptrs = ctx.arena.alloc<const uint32_t *>(1);
    uint32_t *totals = ctx.arena.alloc<uint32_t>(2);  // [0] exact-search queue, [1] far ranks
    uint32_t *by_rank = ctx.arena.alloc<uint32_t>(n);
    uint32_t *far_aux = ctx.arena.alloc<uint32_t>(n);
    uint32_t *scratch_idx = ctx.arena.alloc<uint32_t>(n);
    uint32_t *scratch_val = ctx.arena.alloc<uint32_t>(isa_fill ? 2 * (size_t)n : (size_t)n);  // (two values per pair: radix_sort.hpp)
    HIP_CHECK(hipMemsetAsync(qcounts, 0, kQShards * kQPad * sizeof(uint32_t), s));
    const uint32_t *h_ptrs[1] = {exact_q.counts};
    HIP_CHECK(hipMemcpyAsync(count_ptrs, h_ptrs, sizeof h_ptrs, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));  // h_ptrs is a local array
    auto read_totals = [&](uint32_t h[2]) {  // (totals[1] is written once, by the scan of the far counts)
        shard_totals_kernel<<<1, kQShards, 0, s>>>(count_ptrs, 1, totals);
        KERNEL_CHECK();
        ctx.read_back(totals, h, 2);
    };
    {
        ProfScope ps(ctx.profiler(), "lpf", s, 12.0 * (double)n);
        static const bool want_phases = getenv("NOLZSS_LPF_PHASES") != nullptr;
        unsigned long long *phases = nullptr;
        if (want_phases) {
            phases = ctx.arena.alloc<unsigned long long>(8);
            HIP_CHECK(hipMemsetAsync(phases, 0, 64, s));
        }
        if (phases)
            lpf_tile_kernel<true><<<tiles, kLdsThreads, 0, s>>>(sa, lcp, n, by_rank, exact_q, far_items, far_cnt, far_aux, phases, psa1, plcp1,
                                                                pending_threshold(), pending_flag);
        else
            lpf_tile_kernel<false><<<tiles, kLdsThreads, 0, s>>>(sa, lcp, n, by_rank, exact_q, far_items, far_cnt, far_aux, nullptr, psa1,
                                                                 plcp1, pending_threshold(), pending_flag);
        KERNEL_CHECK();
        if (phases) {
            unsigned long long h[8];
            HIP_CHECK(hipMemcpyAsync(h, phases, 64, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
            const double w = h[5] ? (double)h[5] : 1.0;
            fprintf(stderr, "[nolzss] lpf_tile phases (cycles per wavefront, %llu sampled): stage %.0f  round0 %.0f  %s %.0f  %s %.0f  epilogue %.0f\n",
                    h[5], h[0] / w, h[1] / w, "roundA", h[2] / w, "roundsBC", h[3] / w, h[4] / w);
        }
    }
    scan_exclusive_add_u32(far_cnt, far_off, far_regions, totals + 1, ctx.arena, s);
    if (fill_pyramids) {
        ProfScope ps(ctx.profiler(), "pyramids", s);
        if (psa1) {  // level 1 came from the tile kernel, but for the blocks that reach beyond the last rank
            fill_pyramid_tail(Psa, n >> kPyrShift, false, s);
            fill_pyramid_tail(Plcp, n >> kPyrShift, false, s);
            fill_pyramid(Psa, 2, false, s);
            fill_pyramid(Plcp, 2, false, s);
        } else {  // (a text of fewer than 16 symbols)
            fill_pyramid(Psa, 1, false, s);
            fill_pyramid(Plcp, 1, false, s);
        }
        uint32_t pending = 0;
        ctx.read_back(pending_flag, &pending, 1);
        if (pending) {
            // safety net (build_lcp_pyramid): an LCP entry nobody decided is compared in the text; the candidates are
            // then computed again, the ordinary way
            ctx.arena.rewind(mark);
            finish_pending_lcp(ctx, *fill_pyramids, sa, const_cast<uint32_t *>(lcp));
            fill_pyramid(Psa, 1, false, s);
            fill_pyramid(Plcp, 1, false, s);
            return build_lstar(ctx, n, sa, isa, lcp, Psa, Plcp, lstar, isa_fill, nullptr);
        }
    }
    {
        // lstar[sa[r]] = by_rank[r]: rank order -> text order (a permutation scatter); with isa_fill the same
        // permutation writes isa[sa[r]] = r + 1
        ProfScope ps(ctx.profiler(), "lpf_to_text_order", s);
        uint32_t *idx[2] = {const_cast<uint32_t *>(sa), scratch_idx};
        uint32_t *val[2] = {by_rank, scratch_val};
        bucketed_scatter(idx, val, n, lstar, n, ctx.arena, s, ctx.profiler(), true, true, ctx.rec_plan, isa_fill);
    }
    uint32_t h[2] = {0, 0};
    read_totals(h);
    static const bool trace = getenv("NOLZSS_TRACE") != nullptr;
    if (trace) fprintf(stderr, "[nolzss] lpf: %u ranks to the far queue, %u positions to the exact search so far\n", h[1], h[0]);
    if (h[1] > 0) {
        ProfScope ps(ctx.profiler(), "lpf_far", s);
        uint32_t *far_list = ctx.arena.alloc<uint32_t>(h[1]);
        compact_far_kernel<<<(unsigned)std::min<size_t>(div_up((size_t)far_regions * 4, kThreads), 256u * 16u), kThreads, 0, s>>>(
            far_items, far_cnt, far_off, far_regions, far_list);
        KERNEL_CHECK();
        uint32_t *far_exact = ctx.arena.alloc<uint32_t>(h[1]);
        const unsigned g = (unsigned)std::min<size_t>(256u * 64u, std::max<size_t>(1, div_up(h[1], kThreads)));
        lpf_far_kernel<<<g, kThreads, 0, s>>>(far_list, h[1], sa, lcp, n, Psa, Plcp, by_rank, far_aux, n <= 0x80000000u,
                                              lstar, far_exact);
        KERNEL_CHECK();
        lpnf_fallback_list_kernel<<<g, kThreads, 0, s>>>(far_exact, h[1], n, isa, Psa, Plcp, lstar);
        KERNEL_CHECK();
    }
    if (h[0] > 0) {
        ProfScope ps(ctx.profiler(), "lpnf_fallback", s);
        const unsigned gy = (unsigned)std::min<size_t>(64, std::max<size_t>(1, div_up(h[0], (size_t)kQShards * kThreads)));
        lpnf_fallback_kernel<<<dim3(kQShards, gy), kThreads, 0, s>>>(exact_q, n, isa, Psa, Plcp, lstar);
        KERNEL_CHECK();
    }
    ctx.arena.rewind(mark);
    return h[0];
}

}  // namespace nolzss
