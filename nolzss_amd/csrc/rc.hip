// rc.hip -- reverse-complement DNA mode on the device.
//
// Restates detail::nolzss_multiple_dna_w_rc (/root/reference/src/cpp/factorizer_core.hpp:177-383)
// over the prepared string  S = T1 s0 ... Tk s(k-1) rc(Tk) sk ... rc(T1) s(2k-1)
// (prepare_multiple_dna_sequences_w_rc, factorizer.cpp:54-172);  N = |S|/2 - 1  (:195).
//
// Array form of the reference's ancestor walk, r = ISA[i], i < N:
//   forward candidate  (fwd_starts / rmqF, :211-231, 264-266, 280-287, 322-326)
//       L_f  = max{ d : min SA[I(d)] + d <= i }                 -- as in plain mode, over S
//       d_u  = deepest EXPLICIT node depth <= L_f = max(LCP[lo], LCP[hi+1]) for [lo,hi] = I(L_f+1)
//       j    = min SA[I(d_u)],   fwd = min(lcp(i, j), i - j)     -- may be shorter than L_f
//   reverse-complement candidate  (rc_ends / rmqRcEnd, :224-232, 269-271, 290-299, 328-330)
//       rc_ends[k] = 2N - SA[k] for rc-strand suffixes, so  rc_ends < i  <=>  SA[k] > 2N - i:
//       rc   = longest LCP between suffix i and a suffix starting after 2N - i
//            = nearest GREATER SA value above / below r (max pyramid), running LCP minimum
//   selection  (:338-352)  forward wins ties; RC must beat the forward match, or 1 if none.
// When the best forward neighbour does not overlap position i (the common case) L_f is an LCA
// depth, hence explicit, and fwd = L_f directly; only overlapping positions take the exact path.
#include <algorithm>

#include "nearest_lds.hpp"
#include "pipeline.hpp"
#include "queues.hpp"
#include "radix_sort.hpp"
#include "scan.hpp"

namespace nolzss {
namespace {

constexpr int kThreads = 256;
constexpr uint32_t kRcFlag = 0x80000000u;

__device__ __forceinline__ uint32_t rc_select(uint32_t fwd, uint32_t rc) {
    if (fwd >= 1) return (rc > fwd) ? (rc | kRcFlag) : fwd;  // :338-344
    return (rc > 1) ? (rc | kRcFlag) : 0u;                   // :345-352 (0 = literal)
}

// combine the forward neighbours (lp/jp above, ls/js below) and the reverse-complement length
// into the factor code of position i; returns true if i needs the exact forward search
// (*dst then receives a forward lower bound with P(bound) true)
__device__ __forceinline__ bool rc_decide(uint32_t i, uint32_t lp, uint32_t jp, uint32_t ls, uint32_t js,
                                          uint32_t rc, uint32_t *__restrict__ dst) {
    const uint32_t M = lp > ls ? lp : ls;
    const bool fwd_final = (M == 0) || (lp == M && i - jp >= M) || (ls == M && i - js >= M);
    if (fwd_final) {
        *dst = rc_select(M, rc);
        return false;
    }
    uint32_t lo = 0;
    if (lp > 0) { const uint32_t c = lp < i - jp ? lp : i - jp; lo = c > lo ? c : lo; }
    if (ls > 0) { const uint32_t c = ls < i - js ? ls : i - js; lo = c > lo ? c : lo; }
    *dst = lo;  // provisional: P(lo) holds
    return true;
}

// LDS-tiled candidate search (nearest_lds.hpp): four searches per rank of the original strand
__global__ __launch_bounds__(kLdsThreads) void rc_tile_kernel(const uint32_t *__restrict__ sa,
                                                              const uint32_t *__restrict__ lcp, uint32_t m,
                                                              uint32_t N, uint32_t *__restrict__ code_by_rank,
                                                              ShardQueue exact_q, ShardQueue far_q,
                                                              uint32_t *__restrict__ pmin1, uint32_t *__restrict__ pmax1,
                                                              uint32_t *__restrict__ plcp1, uint32_t pending_min,
                                                              uint32_t *__restrict__ pending_flag,
                                                              const uint32_t *__restrict__ tile_off,
                                                              uint32_t *__restrict__ cidx, uint64_t *__restrict__ cpacked) {
    constexpr int NS = 4, NP = 2;
    __shared__ __align__(16) uint32_t s_sa[kLdsSpan];
    __shared__ __align__(16) uint32_t s_lcp[kLdsSpan + 4];
    __shared__ uint32_t s_len[NS * kLdsTile];
    __shared__ uint16_t s_pos[NP * kLdsTile];  // (local index of the match, nearest_lds.hpp: match_pos)
    // 128 instead of 256 entries per list: 46.6 instead of 54.8 KiB of LDS, THREE workgroups per CU instead of two --
    // rc_candidates 14.2 -> 10.6 ms at 2^28 bases (240 entries, 53.8 KiB, still ran two per CU).  A list overflows when
    // more than 128 of a wavefront's ranks are still searching in one direction after four steps (one in five is, of
    // the 128 ranks of the original strand a wavefront holds on average): those ranks go to the searches from global memory.
    constexpr int kListCap = 128;
    constexpr int kListCapB = 64;  // (searches still going after 16 steps: one in 17)
    __shared__ uint16_t s_list0[kLdsWaves][NS * kListCap];
    __shared__ uint16_t s_list1[kLdsWaves][NS * kListCapB];
    __shared__ uint32_t s_blk[4 * kBlkTableLen];
    const uint32_t base = blockIdx.x * (uint32_t)kLdsTile;
    const uint32_t shard = blockIdx.x % kQShards;
    stage_tile(sa, lcp, m, base, s_sa, s_lcp, pending_min, pending_flag);
    __syncthreads();
    const BlockTables T = block_tables<true>(s_blk);
    build_block_tables<true>(s_sa, s_lcp, T);
    __syncthreads();
    // the block tables of the tile's own ranks are the first level of the three pyramids (lpnf.hip, lpf_tile_kernel)
    if (pmin1 != nullptr && threadIdx.x < kLdsTile / kBlk) {
        const uint32_t B = (uint32_t)kLdsReach / kBlk + threadIdx.x;
        const uint64_t first = (uint64_t)base + (uint64_t)threadIdx.x * kBlk;
        if (first + kBlk <= (uint64_t)m) {
            pmin1[first >> 4] = T.mn[B];
            pmax1[first >> 4] = T.mx[B];
            plcp1[first >> 4] = T.ldn[B];
        }
    }
    const int w = threadIdx.x >> 6;
    const uint32_t far_bit = m <= 0x80000000u ? 0x80000000u : 0u;
    lds_search_wave_blocks<NS, NP, true, kListCap, kListCapB>(s_sa, s_lcp, T, m, base, s_len, s_pos, s_list0[w], s_list1[w],
                                   [N](uint32_t i) { return i < N; }, [N](uint32_t i) { return 2u * N - i; }, far_bit);
    constexpr int kRows = kLdsPerWave / 64;
    bool far[kRows], exact[kRows];
    uint32_t mask[kRows], rcl[kRows], fslot[kRows], eslot[kRows];
    // COMPACT output (cidx != nullptr): only the ranks of the original strand leave the kernel, as pairs (position,
    // code | rank + 1 << 32) in rank order at tile_off[tile] -- half of the pairs of S never need a place in text order
    // (the codes and the inverse suffix array are read at positions < N only), so the permutation behind this kernel
    // moves N pairs instead of 2 N + 2.  Ranks of the original strand in the wavefronts in front of mine: counted from
    // the staged tile.
    const bool compact_out = cidx != nullptr;
    uint32_t cbase = 0;
    if (compact_out) {
        uint32_t before = 0;
        for (int t2 = lane_id(); t2 < w * kLdsPerWave; t2 += 64)
            before += ((uint64_t)base + t2 < m && s_sa[t2 + kLdsReach] < N) ? 1u : 0u;
        cbase = tile_off[blockIdx.x] + wave_reduce(before, OpAdd<uint32_t>());
    }
    uint32_t cv[kRows], cat[kRows];
#pragma unroll
    for (int row = 0; row < kRows; ++row) {
        const int t = w * kLdsPerWave + row * 64 + lane_id();
        const uint64_t rr = (uint64_t)base + t;
        const uint32_t i = s_sa[t + kLdsReach];
        far[row] = exact[row] = false;
        mask[row] = rcl[row] = 0;
        cv[row] = 0;
        const bool orig = rr < m && i < N;
        {
            const uint64_t ob = __ballot(orig);
            cat[row] = orig ? cbase + (uint32_t)__popcll(ob & lanemask_lt()) : 0xffffffffu;
            cbase += (uint32_t)__popcll(ob);
        }
        if (rr >= m) continue;
        if (i >= N) {  // only positions of the original strand are factorized (:241)
            if (!compact_out) code_by_rank[rr] = 0;
            continue;
        }
        uint32_t lp = s_len[t], ls = s_len[kLdsTile + t];
        uint32_t ru = s_len[2 * kLdsTile + t], rd = s_len[3 * kLdsTile + t];
        // which of the four searches left the reach (they restart behind the ranks already cleared)
        // (a search that found its work list full knows nothing: it restarts at the rank itself, like a search whose bit is clear)
        if (far_bit)
            mask[row] = ((far_is(lp, far_bit) && lp != kListOverflow) ? 1u : 0u) | ((far_is(ls, far_bit) && ls != kListOverflow) ? 2u : 0u) |
                        ((far_is(ru, far_bit) && ru != kListOverflow) ? 4u : 0u) | ((far_is(rd, far_bit) && rd != kListOverflow) ? 8u : 0u);
        // a search that left the reach matters only if its bound can beat the other direction
        // (reverse-complement lengths <= 1 can never be chosen)
        const bool far_f = far_resolve(lp, ls, far_bit, 0u);
        const bool far_r = far_resolve(ru, rd, far_bit, 1u);
        far[row] = far_f || far_r;  // finished from global memory
        if (!far[row]) {
            rcl[row] = ru > rd ? ru : rd;
            exact[row] = rc_decide(i, lp, match_pos(s_sa, s_pos[t]), ls, match_pos(s_sa, s_pos[kLdsTile + t]), rcl[row], &cv[row]);
            if (!compact_out) code_by_rank[rr] = cv[row];
        }
    }
    shard_slots<kRows>(far_q, shard, far, fslot);
    shard_slots<kRows>(exact_q, shard, exact, eslot);
#pragma unroll
    for (int row = 0; row < kRows; ++row) {
        const int t = w * kLdsPerWave + row * 64 + lane_id();
        const uint64_t rr = (uint64_t)base + t;
        if (far[row]) {
            far_q.items[fslot[row]] = (uint32_t)rr;
            code_by_rank[rr] = mask[row];  // (read back by rc_far_kernel, by rank; the code of a far rank comes from there)
        }
        if (exact[row]) {
            exact_q.items[eslot[row]] = s_sa[t + kLdsReach];
            exact_q.items2[eslot[row]] = rcl[row];
        }
        if (compact_out && cat[row] != 0xffffffffu) {
            cidx[cat[row]] = s_sa[t + kLdsReach];
            cpacked[cat[row]] = (uint64_t)cv[row] | ((uint64_t)((uint32_t)rr + 1u) << 32);
        }
    }
}

// ranks of the original strand per tile of rc_tile_kernel (the offsets of its compact output, after a scan)
__global__ __launch_bounds__(kLdsThreads) void rc_count_original_kernel(const uint32_t *__restrict__ sa, uint32_t m, uint32_t N,
                                                                        uint32_t *__restrict__ counts) {
    const size_t base = (size_t)blockIdx.x * kLdsTile;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < kLdsTile / kLdsThreads; ++k) {
        const size_t r = base + (size_t)k * kLdsThreads + threadIdx.x;
        c += (r < m && sa[r < m ? r : m - 1] < N) ? 1u : 0u;
    }
    c = wave_reduce(c, OpAdd<uint32_t>());
    __shared__ uint32_t s_part[kLdsWaves];
    if (lane_id() == 0) s_part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int k = 0; k < kLdsWaves; ++k) t += s_part[k];
        counts[blockIdx.x] = t;
    }
}

// ranks whose searches leave the LDS reach: pyramid searches from global memory.  Grid (kQShards, Y).
__global__ __launch_bounds__(kThreads) void rc_far_kernel(
    ShardQueue far_q, const uint32_t *__restrict__ sa, const uint32_t *__restrict__ lcp, uint32_t m, uint32_t N,
    Pyramid Pmin, Pyramid Pmax, Pyramid Plcp, const uint32_t *__restrict__ by_rank, uint32_t *__restrict__ code,
    ShardQueue exact_q) {
    const uint32_t shard = blockIdx.x;
    const uint32_t count = far_q.counts[shard * kQPad];
    const uint32_t *items = far_q.items + (size_t)shard * far_q.cap;
    for (uint32_t k = blockIdx.y * blockDim.x + threadIdx.x; k < count; k += gridDim.y * blockDim.x) {
        const uint32_t r = items[k];
        const uint32_t i = sa[r];
        const uint32_t mask = by_rank[r];  // searches that left the tile's reach (0: unknown, search all from r)
        // forward: earlier suffixes, SA[q] < i
        uint32_t lp, jp, ls, js;
        if (mask & 1u)
            far_up<false>(sa, Pmin, Plcp, r, i, 0u, lp, jp);
        else
            nearest_up<false>(sa, lcp, Pmin, Plcp, r, i, 0u, lp, jp);
        if (mask & 2u)
            far_down<false>(sa, m, Pmin, Plcp, r, i, lp, ls, js);
        else
            nearest_down<false>(sa, lcp, m, Pmin, Plcp, r, i, lp, ls, js);
        // reverse complement: suffixes starting after 2N - i; lengths <= 1 can never be chosen
        const uint32_t thr = 2u * N - i;
        uint32_t ru, rd, unused;
        if (mask & 4u)
            far_up<true>(sa, Pmax, Plcp, r, thr, 2u, ru, unused);
        else
            nearest_up<true>(sa, lcp, Pmax, Plcp, r, thr, 2u, ru, unused);
        if (mask & 8u)
            far_down<true>(sa, m, Pmax, Plcp, r, thr, ru > 2u ? ru : 2u, rd, unused);
        else
            nearest_down<true>(sa, lcp, m, Pmax, Plcp, r, thr, ru > 2u ? ru : 2u, rd, unused);
        const uint32_t rc = ru > rd ? ru : rd;
        const bool exact = rc_decide(i, lp, jp, ls, js, rc, code + i);
        const uint32_t eslot = shard_slot(exact_q, shard, exact);  // (a rank reaches the exact queue at most once)
        if (exact) {
            exact_q.items[eslot] = i;
            exact_q.items2[eslot] = rc;
        }
    }
}

__global__ __launch_bounds__(kThreads) void rc_fallback_kernel(ShardQueue exact_q, uint32_t m,
                                                               const uint32_t *__restrict__ isa,
                                                               const uint32_t *__restrict__ lcp, Pyramid Pmin,
                                                               Pyramid Plcp, uint32_t *__restrict__ code) {
    const uint32_t shard = blockIdx.x;
    const uint32_t count = exact_q.counts[shard * kQPad];
    const uint32_t *queue = exact_q.items + (size_t)shard * exact_q.cap;
    const uint32_t *queue_rc = exact_q.items2 + (size_t)shard * exact_q.cap;
    for (uint32_t k = blockIdx.y * blockDim.x + threadIdx.x; k < count; k += gridDim.y * blockDim.x) {
        const uint32_t i = queue[k];
        const uint32_t r = isa[i] - 1u;  // (1-based, pipeline.hpp)
        const uint32_t cap = (m - i) < i ? (m - i) : i;
        const uint32_t Lf = lpnf_search(Pmin, Plcp, r, i, code[i], cap);  // >= 1 for queued positions
        // deepest explicit ancestor of leaf(i) with depth <= L_f
        uint32_t a, b;
        lcp_interval(Plcp, r, Lf + 1, a, b);
        const uint32_t da = lcp[a], db = lcp[b + 1];
        const uint32_t d_u = da > db ? da : db;
        if (d_u == 0) {  // cannot happen for L_f >= 1; keep the search well-defined
            code[i] = rc_select(0u, queue_rc[k]);
            continue;
        }
        lcp_interval(Plcp, r, d_u, a, b);
        const uint32_t j = pyr_range<false>(Pmin, a, b);  // best_fwd_start (:284)
        const uint32_t rj = isa[j] - 1u;
        const uint32_t x = r < rj ? r : rj, y = r < rj ? rj : r;
        const uint32_t l = pyr_range<false>(Plcp, x + 1, y);  // lcp(cst, i, best_fwd_start) (:324)
        const uint32_t fwd = l < i - j ? l : i - j;           // :323-325
        code[i] = rc_select(fwd, queue_rc[k]);
    }
}

// S = upper(T) s0 revcomp(upper(T)) s1 for ONE sequence, built on the device
// (prepare_multiple_dna_sequences_w_rc with k = 1, /root/reference/src/cpp/factorizer.cpp:54-172:
// sentinels 1 and 2, :110-125).  first_bad = smallest index holding a non-ACGT byte (n if none).
__global__ __launch_bounds__(kThreads) void rc_prepare_kernel(const uint8_t *__restrict__ T, uint32_t n,
                                                              uint8_t *__restrict__ S,
                                                              uint32_t *__restrict__ first_bad) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint8_t c = T[i];
        if (c >= 'a' && c <= 'z') c = (uint8_t)(c - 'a' + 'A');
        uint8_t comp;
        switch (c) {
        case 'A': comp = 'T'; break;
        case 'C': comp = 'G'; break;
        case 'G': comp = 'C'; break;
        case 'T': comp = 'A'; break;
        default:
            comp = 0;
            atomicMin(first_bad, (uint32_t)i);
            break;
        }
        S[i] = c;
        S[2 * (size_t)n - i] = comp;  // position n + 1 + (n - 1 - i)
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        S[n] = 1;
        S[2 * (size_t)n + 1] = 2;
    }
}

// the same for a batch: T = records with separator bytes between them; whatever is not a nucleotide
// keeps its value on both strands (pack_independent_text checks afterwards that only separators did)
__global__ __launch_bounds__(kThreads) void rc_prepare_batch_kernel(const uint8_t *__restrict__ T, uint32_t n,
                                                                    uint8_t separator, uint8_t *__restrict__ S) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint8_t c = T[i];
        if (c >= 'a' && c <= 'z') c = (uint8_t)(c - 'a' + 'A');
        uint8_t comp = c;
        switch (c) {
        case 'A': comp = 'T'; break;
        case 'C': comp = 'G'; break;
        case 'G': comp = 'C'; break;
        case 'T': comp = 'A'; break;
        default: break;
        }
        S[i] = c;
        S[2 * (size_t)n - i] = comp;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        S[n] = separator;
        S[2 * (size_t)n + 1] = separator;
    }
}

}  // namespace

void prepare_batch_rc_on_device(Context &ctx, const uint8_t *d_T, uint32_t n, uint8_t separator, uint8_t *d_S) {
    size_t g = div_up((size_t)n + 1, kThreads);
    if (g > 8192) g = 8192;
    rc_prepare_batch_kernel<<<(unsigned)g, kThreads, 0, ctx.stream>>>(d_T, n, separator, d_S);
    KERNEL_CHECK();
}

uint32_t prepare_single_rc_on_device(Context &ctx, const uint8_t *d_T, uint32_t n, uint8_t *d_S) {
    uint32_t *first_bad = ctx.arena.alloc<uint32_t>(1);
    HIP_CHECK(hipMemsetAsync(first_bad, 0xff, sizeof(uint32_t), ctx.stream));
    {
        ProfScope ps(ctx.profiler(), "rc_prepare", ctx.stream);
        size_t g = div_up(n, kThreads);
        if (g > 8192) g = 8192;
        rc_prepare_kernel<<<(unsigned)g, kThreads, 0, ctx.stream>>>(d_T, n, d_S, first_bad);
        KERNEL_CHECK();
    }
    uint32_t bad = 0;
    ctx.read_back(first_bad, &bad, 1);
    return bad;  // 0xffffffff when every byte is a nucleotide
}

uint32_t run_rc_pipeline(Context &ctx, const uint8_t *d_S, size_t m_sz, size_t start_pos, void **d_factors_out) {
    const PackedText text = pack_text(ctx, d_S, m_sz);  // segmented 2-bit packing is detected there
    return run_rc_pipeline_packed(ctx, text, start_pos, d_factors_out);
}

uint32_t run_rc_pipeline_packed(Context &ctx, const PackedText &text, size_t start_pos, void **d_factors_out) {
    const uint32_t m = text.n;
    const uint32_t N = m / 2 - 1;
    hipStream_t s = ctx.stream;
    Arena &arena = ctx.arena;

    uint32_t *sa = arena.alloc<uint32_t>(m);
    uint32_t *isa = arena.alloc<uint32_t>(m);
    uint32_t *lcp = arena.alloc<uint32_t>((size_t)m + 1);
    bool isa_deferred = false;
    build_suffix_array(ctx, text, sa, isa, lcp, &isa_deferred);
    // (allocated here, filled behind the tile kernel, which writes their first level: lpnf.hip)
    const Pyramid Pmin = alloc_pyramid(sa, m, arena), Pmax = alloc_pyramid(sa, m, arena),
                  Plcp = alloc_pyramid(lcp, m + 1, arena);
    const bool fused_level1 = Pmin.nlev > 1 && Plcp.nlev > 1;
    uint32_t *pending_flag = arena.alloc<uint32_t>(1);
    HIP_CHECK(hipMemsetAsync(pending_flag, 0, sizeof(uint32_t), s));
    inject_pending_for_test(ctx, lcp, m);
    // code[] spans all of S (entries >= N are unused) so that rank order -> text order is a
    // permutation scatter
    uint32_t *code = arena.alloc<uint32_t>(m);
    {
        const size_t mark = arena.mark();
        const unsigned tiles = (unsigned)div_up(m, kLdsTile);
        ShardQueue exact_q, far_q;
        exact_q.cap = far_q.cap = (uint32_t)shard_queue_cap(tiles, kLdsTile);
        exact_q.items = arena.alloc<uint32_t>((size_t)kQShards * exact_q.cap);
        exact_q.items2 = arena.alloc<uint32_t>((size_t)kQShards * exact_q.cap);
        far_q.items = arena.alloc<uint32_t>((size_t)kQShards * far_q.cap);
        uint32_t *qcounts = arena.alloc<uint32_t>(2 * kQShards * kQPad);
        exact_q.counts = qcounts;
        far_q.counts = qcounts + kQShards * kQPad;
        const uint32_t **count_ptrs = arena.alloc<const uint32_t *>(2);
        uint32_t *totals = arena.alloc<uint32_t>(2);  // [0] exact-search queue, [1] far queue
        uint32_t *by_rank = arena.alloc<uint32_t>(m);
        // compact output of the tile kernel (above): when the permutation also delivers the inverse suffix array, and
        // the N pairs fit its two partition passes
        static const bool no_compact = getenv("NOLZSS_RC_NO_COMPACT") != nullptr;  // (A/B switch)
        static const uint32_t compact_min = getenv("NOLZSS_RC_COMPACT_MIN") ? (uint32_t)atoll(getenv("NOLZSS_RC_COMPACT_MIN")) : (1u << 22);  // (tests: 1)
        const bool compact = isa_deferred && !no_compact && N >= compact_min && N <= (1u << 30);
        uint32_t *scratch_idx = nullptr, *scratch_val = nullptr, *tile_off = nullptr, *cidx = nullptr;
        uint64_t *cpacked = nullptr;
        if (compact) {
            uint32_t *tile_cnt = arena.alloc<uint32_t>(tiles);
            tile_off = arena.alloc<uint32_t>(tiles);
            cidx = arena.alloc<uint32_t>(N);
            cpacked = arena.alloc<uint64_t>(N);
            ProfScope ps(ctx.profiler(), "rc_candidates", s);
            rc_count_original_kernel<<<tiles, kLdsThreads, 0, s>>>(sa, m, N, tile_cnt);
            KERNEL_CHECK();
            scan_exclusive_add_u32(tile_cnt, tile_off, tiles, nullptr, arena, s);
        } else {
            scratch_idx = arena.alloc<uint32_t>(m);
            scratch_val = arena.alloc<uint32_t>(isa_deferred ? 2 * (size_t)m : (size_t)m);  // (two values per pair: radix_sort.hpp)
        }
        HIP_CHECK(hipMemsetAsync(qcounts, 0, 2 * kQShards * kQPad * sizeof(uint32_t), s));
        const uint32_t *h_ptrs[2] = {exact_q.counts, far_q.counts};
        HIP_CHECK(hipMemcpyAsync(count_ptrs, h_ptrs, sizeof h_ptrs, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipStreamSynchronize(s));  // h_ptrs is a local array
        auto read_totals = [&](uint32_t h[2]) {
            shard_totals_kernel<<<2, kQShards, 0, s>>>(count_ptrs, 2, totals);
            KERNEL_CHECK();
            ctx.read_back(totals, h, 2);
        };
        {
            ProfScope ps(ctx.profiler(), "rc_candidates", s);
            rc_tile_kernel<<<tiles, kLdsThreads, 0, s>>>(
                sa, lcp, m, N, by_rank, exact_q, far_q, fused_level1 ? const_cast<uint32_t *>(Pmin.lvl[1]) : nullptr,
                fused_level1 ? const_cast<uint32_t *>(Pmax.lvl[1]) : nullptr,
                fused_level1 ? const_cast<uint32_t *>(Plcp.lvl[1]) : nullptr, pending_threshold(), pending_flag, tile_off, cidx,
                cpacked);
            KERNEL_CHECK();
        }
        {
            ProfScope ps(ctx.profiler(), "pyramids", s);
            const int from = fused_level1 ? 2 : 1;
            if (fused_level1) {
                fill_pyramid_tail(Pmin, m >> kPyrShift, false, s);
                fill_pyramid_tail(Pmax, m >> kPyrShift, true, s);
                fill_pyramid_tail(Plcp, m >> kPyrShift, false, s);
            }
            fill_pyramid(Pmin, from, false, s);
            fill_pyramid(Pmax, from, true, s);
            fill_pyramid(Plcp, from, false, s);
            uint32_t pending = 0;
            ctx.read_back(pending_flag, &pending, 1);
            if (pending) {
                // safety net (build_lcp_pyramid): an undecided LCP entry is compared in the text, the pyramids are
                // built again and the candidates computed once more from the repaired array
                finish_pending_lcp(ctx, text, sa, lcp);
                fill_pyramid(Pmin, 1, false, s);
                fill_pyramid(Pmax, 1, true, s);
                fill_pyramid(Plcp, 1, false, s);
                HIP_CHECK(hipMemsetAsync(qcounts, 0, 2 * kQShards * kQPad * sizeof(uint32_t), s));
                rc_tile_kernel<<<tiles, kLdsThreads, 0, s>>>(sa, lcp, m, N, by_rank, exact_q, far_q, nullptr, nullptr, nullptr,
                                                             0u, nullptr, tile_off, cidx, cpacked);
                KERNEL_CHECK();
            }
        }
        {
            ProfScope ps(ctx.profiler(), "rc_to_text_order", s);
            if (compact) {  // code[i] and isa[i] for the N positions of the original strand
                permute_packed(cidx, cpacked, N, code, isa, arena, s, ctx.profiler());
            } else {
                uint32_t *idx[2] = {sa, scratch_idx};
                uint32_t *val[2] = {by_rank, scratch_val};
                // (isa_deferred: the same permutation writes isa[sa[r]] = r + 1)
                bucketed_scatter(idx, val, m, code, m, arena, s, ctx.profiler(), true, true, nullptr, isa_deferred ? isa : nullptr);
            }
        }
        uint32_t h[2] = {0, 0};
        read_totals(h);
        if (h[1] > 0) {
            ProfScope ps(ctx.profiler(), "rc_far", s);
            const unsigned gy = (unsigned)std::min<size_t>(64, std::max<size_t>(1, div_up(h[1], (size_t)kQShards * kThreads)));
            rc_far_kernel<<<dim3(kQShards, gy), kThreads, 0, s>>>(far_q, sa, lcp, m, N, Pmin, Pmax, Plcp, by_rank, code, exact_q);
            KERNEL_CHECK();
            read_totals(h);
        }
        if (h[0] > 0) {
            ProfScope ps(ctx.profiler(), "rc_fallback", s);
            const unsigned gy = (unsigned)std::min<size_t>(64, std::max<size_t>(1, div_up(h[0], (size_t)kQShards * kThreads)));
            rc_fallback_kernel<<<dim3(kQShards, gy), kThreads, 0, s>>>(exact_q, m, isa, lcp, Pmin, Plcp, code);
            KERNEL_CHECK();
        }
        arena.rewind(mark);
    }
    return resolve_chain(ctx, N, (uint32_t)start_pos, code, sa, isa, lcp, Pmin, Plcp, d_factors_out, N, &Pmax);
}

}  // namespace nolzss
