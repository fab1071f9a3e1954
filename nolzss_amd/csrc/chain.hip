// chain.hip -- resolves the reference's sequential greedy cursor on the device and emits factors.
//
// The reference advances one factor at a time: lambda = next_leaf(lambda, l)
// (/root/reference/src/cpp/factorizer_core.hpp:113-115).  With L*[i] known for every position,
// the cursor is the chain  p -> next(p) = p + max(1, L*[p])  from start_pos, and the factors are
// exactly the chain positions < n.  next(p) > p, which makes the chain resolvable in O(n) work:
//
//   1. chain_exit_kernel   per 4096-position tile: in-LDS pointer jumping gives exit0[p] = first
//                          chain position at or beyond the tile end; every distinct exit target
//                          is recorded in a bitmap.
//   2. node graph          the marked targets (plus start_pos) are the only positions at which
//                          the chain can ENTER a tile.  They are compacted (bitmap rank) into a
//                          small node list with edges node -> node(exit0[node]).
//   3. wyllie_kernel       pointer doubling with a reached flag over the node list only
//                          (ceil(log2 K) rounds over K << n nodes): reached nodes = tile entries.
//   4. chain_mark_kernel   per entered tile: one lane walks the tile's chain from the entry in LDS
//                          point; the marks go out as one ballot word per wavefront row.
//   5. scan + emit         per-tile counts are scanned and the marked positions written in order;
//                          factor_kernel then attaches length and leftmost-occurrence reference.
#include "pipeline.hpp"
#include "pyramid.hpp"
#include "scan.hpp"

namespace nolzss {
namespace {

constexpr int kThreads = 256;
constexpr int kTile = 4096;
constexpr int kPerThread = kTile / kThreads;  // 16
constexpr uint32_t kNone = 0xffffffffu;

// L* codes: bit 31 flags a reverse-complement factor (RC mode only), bits 0..30 the length,
// 0 = literal.
constexpr uint32_t kLenMask = 0x7fffffffu;

__device__ __forceinline__ uint32_t next_pos(uint32_t p, uint32_t code, uint32_t n) {
    const uint32_t L = code & kLenMask;
    const uint64_t nx = (uint64_t)p + (L ? L : 1u);
    return nx > n ? n : (uint32_t)nx;
}

__global__ __launch_bounds__(kThreads) void chain_exit_kernel(const uint32_t *__restrict__ lstar, uint32_t n,
                                                              uint32_t start_pos,
                                                              uint32_t *__restrict__ exit0,
                                                              uint32_t *__restrict__ target_bits) {
    __shared__ uint32_t jump[kTile];
    const uint32_t base = blockIdx.x * (uint32_t)kTile;
    const uint32_t tile_end = (n - base < (uint32_t)kTile) ? n : base + kTile;
    const int tid = threadIdx.x;
    uint32_t ls[kPerThread];  // all loads of the tile in flight together
#pragma unroll
    for (int j = 0; j < kPerThread; ++j) {
        const uint32_t p = base + j * kThreads + tid;
        ls[j] = lstar[p < n ? p : n - 1u];  // (no branch around the load: guarded, each one was waited for on its own)
    }
#pragma unroll
    for (int j = 0; j < kPerThread; ++j) {
        const uint32_t lp = j * kThreads + tid;
        const uint32_t p = base + lp;
        jump[lp] = (p < n) ? next_pos(p, ls[j], n) : n;
    }
    __syncthreads();
    for (int round = 0; round < 13; ++round) {
        int changed = 0;
#pragma unroll
        for (int j = 0; j < kPerThread; ++j) {
            const uint32_t lp = j * kThreads + tid;
            const uint32_t t = jump[lp];
            if (t < tile_end) {  // still inside the tile: hop through
                jump[lp] = jump[t - base];
                changed = 1;
            }
        }
        if (!__syncthreads_or(changed)) break;
    }
#pragma unroll
    for (int j = 0; j < kPerThread; ++j) {
        const uint32_t lp = j * kThreads + tid;
        const uint32_t p = base + lp;
        if (p < n) {
            const uint32_t e = jump[lp];
            exit0[p] = e;
            const bool last = (lp + 1 == (uint32_t)kTile) || (p + 1 >= n);
            if (e < n && (last || jump[lp + 1] != e)) atomicOr(&target_bits[e >> 5], 1u << (e & 31));
        }
    }
    if (blockIdx.x == 0 && tid == 0 && start_pos < n) atomicOr(&target_bits[start_pos >> 5], 1u << (start_pos & 31));
}

__global__ __launch_bounds__(kThreads) void popcount_kernel(const uint32_t *__restrict__ bits, uint32_t nwords,
                                                            uint32_t *__restrict__ counts) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += stride)
        counts[w] = (uint32_t)__popc(bits[w]);
}

__global__ __launch_bounds__(kThreads) void node_fill_kernel(const uint32_t *__restrict__ bits,
                                                             const uint32_t *__restrict__ prefix, uint32_t nwords,
                                                             uint32_t *__restrict__ node_pos) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += stride) {
        uint32_t b = bits[w];
        uint32_t id = prefix[w];
        while (b) {
            const int bit = __ffs((int)b) - 1;
            node_pos[id++] = (uint32_t)(w * 32 + bit);
            b &= b - 1;
        }
    }
}

__device__ __forceinline__ uint32_t node_id(const uint32_t *bits, const uint32_t *prefix, uint32_t p) {
    return prefix[p >> 5] + (uint32_t)__popc(bits[p >> 5] & ((1u << (p & 31)) - 1u));
}

// nxt[k] = node reached by one exit0 hop; K = terminal.  reach[] initialised to "start node only".
__global__ __launch_bounds__(kThreads) void node_edges_kernel(const uint32_t *__restrict__ node_pos, uint32_t K,
                                                              const uint32_t *__restrict__ exit0, uint32_t n,
                                                              const uint32_t *__restrict__ bits,
                                                              const uint32_t *__restrict__ prefix,
                                                              uint32_t start_pos, uint32_t *__restrict__ nxt,
                                                              uint32_t *__restrict__ reach) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k <= K; k += stride) {
        if (k == K) {
            nxt[k] = K;
            reach[k] = 0;
            continue;
        }
        const uint32_t p = node_pos[k];
        const uint32_t e = exit0[p];
        nxt[k] = (e >= n) ? K : node_id(bits, prefix, e);
        reach[k] = (p == start_pos) ? 1u : 0u;
    }
}

__global__ __launch_bounds__(kThreads) void wyllie_kernel(const uint32_t *__restrict__ nxt_in,
                                                          uint32_t *__restrict__ nxt_out, uint32_t K,
                                                          uint32_t *reach) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < K; k += stride) {
        const uint32_t w = nxt_in[k];
        if (w < K) {
            if (reach[k]) reach[w] = 1u;  // monotone flag: races only ever add true chain nodes
            nxt_out[k] = nxt_in[w];
        } else {
            nxt_out[k] = K;
        }
    }
}

__global__ __launch_bounds__(kThreads) void tile_entries_kernel(const uint32_t *__restrict__ node_pos,
                                                                const uint32_t *__restrict__ reach, uint32_t K,
                                                                uint32_t *__restrict__ entry) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < K; k += stride)
        if (reach[k]) {
            const uint32_t p = node_pos[k];
            entry[p / kTile] = p;
        }
}

// One wavefront per entered tile: the tile's local jump table goes to LDS (all loads in flight
// together), then ONE lane walks the chain from the entry position -- a tile holds about
// kTile / (mean factor length) factor starts, i.e. a few hundred dependent LDS reads, far cheaper
// than the log2(kTile) rounds of pointer doubling over all 4096 positions this replaced (3.9 ->
// 2.6 ms at 2^30 bases); many such single-wave workgroups share a CU.
__global__ __launch_bounds__(64) void chain_mark_kernel(const uint32_t *__restrict__ lstar, uint32_t n,
                                                        const uint32_t *__restrict__ entry,
                                                        unsigned long long *__restrict__ chain_bits,
                                                        uint32_t *__restrict__ tile_count) {
    __shared__ uint16_t jmp[kTile];
    __shared__ unsigned long long bits[kTile / 64];
    const uint32_t base = blockIdx.x * (uint32_t)kTile;
    const uint32_t tile_end = (n - base < (uint32_t)kTile) ? n : base + kTile;
    const int lane = threadIdx.x;
    const uint32_t e = entry[blockIdx.x];
    const size_t word0 = (size_t)blockIdx.x * (kTile / 64);
    if (e == kNone) {  // the chain jumps over this tile
        chain_bits[word0 + lane] = 0ull;
        if (lane == 0) tile_count[blockIdx.x] = 0;
        return;
    }
    constexpr int kBatch = 16;
    for (int j0 = 0; j0 < kTile / 64; j0 += kBatch) {
        uint32_t ls[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const uint32_t p = base + (uint32_t)(j0 + j) * 64u + lane;
            ls[j] = lstar[p < n ? p : n - 1u];
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const uint32_t lp = (uint32_t)(j0 + j) * 64u + lane;
            const uint32_t p = base + lp;
            uint32_t nl = kTile;
            if (p < n) {
                const uint32_t nx = next_pos(p, ls[j], n);
                if (nx < tile_end) nl = nx - base;
            }
            jmp[lp] = (uint16_t)nl;
        }
    }
    bits[lane] = 0ull;
    __syncthreads();  // (one wavefront: a fence)
    if (lane == 0) {
        uint32_t p = e - base, cnt = 0;
        while (p < (uint32_t)kTile) {
            bits[p >> 6] |= 1ull << (p & 63);
            ++cnt;
            p = jmp[p];
        }
        tile_count[blockIdx.x] = cnt;
    }
    __syncthreads();
    chain_bits[word0 + lane] = bits[lane];
}

// one wavefront per tile: lane l owns bitmap word l of the tile
__global__ __launch_bounds__(64) void emit_positions_kernel(const unsigned long long *__restrict__ chain_bits,
                                                            const uint32_t *__restrict__ tile_off,
                                                            uint32_t *__restrict__ fpos) {
    const int lane = threadIdx.x;
    const size_t word = (size_t)blockIdx.x * (kTile / 64) + lane;
    unsigned long long b = chain_bits[word];
    const uint32_t c = (uint32_t)__popcll(b);
    uint32_t off = wave_scan_inclusive(c, OpAdd<uint32_t>()) - c + tile_off[blockIdx.x];
    const uint32_t p0 = blockIdx.x * (uint32_t)kTile + lane * 64u;
    while (b) {
        const int bit = __ffsll((long long)b) - 1;
        fpos[off++] = p0 + (uint32_t)bit;
        b &= b - 1;
    }
}

// ranks the factor kernel walks one by one around ISA[i] before it asks the pyramids: the interval
// of a short factor holds all occurrences of a 12- to 14-mer, dozens to hundreds of ranks, and every
// step of the walk is a dependent load (48 -> 4: 14.5 -> 11 ms per 52 M factors)
constexpr uint32_t kEmitScan = 4;

struct FactorRec {
    uint64_t start, length, ref;
};

// (start, length, ref) per factor.
//   forward factor: ref = leftmost occurrence of the factor string = min SA over I(L)
//     (the reference reports v_min / u_min, factorizer_core.hpp:91,100,105; in RC mode
//     best_fwd_start, :284,360 -- the minimum over I(L) in both of the cases DESIGN.md 3 lists);
//   reverse-complement factor (kRC): the reference keeps the smallest T-coordinate END among the
//     rc-strand suffixes of the node (rc_ends = 2N - SA, factorizer_core.hpp:226-228, 270,
//     294-296), i.e. the LARGEST SA value in I(L); ref = RC_MASK | (end - L + 1)  (:362-364).
// kRebase (merged batch, plain mode): start and ref leave relative to the first position of the factor's
// record (a separate pass over the records cost 2.3 ms per 7*10^7 factors).
template <bool kRC, bool kRebase>
__global__ __launch_bounds__(kThreads) void factor_kernel(const uint32_t *__restrict__ fpos, uint32_t z,
                                                          const uint32_t *__restrict__ lstar,
                                                          const uint32_t *__restrict__ isa,
                                                          const uint32_t *__restrict__ sa,
                                                          const uint32_t *__restrict__ lcp, Pyramid Psa,
                                                          Pyramid Plcp, Pyramid Pmax, uint32_t rcN,
                                                          FactorRec *__restrict__ out, TermTable terms, uint32_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < z; k += stride) {
        const uint32_t i = fpos[k];
        // the factors tile the text: the length is the distance to the next factor (a coalesced read) -- the
        // length code itself is needed only to tell a literal from a match of one symbol, and for the
        // reverse-complement flag
        const uint32_t nxt = k + 1 < z ? fpos[k + 1] : n;
        uint32_t code = nxt - i;
        if (kRC || code == 1u) code = lstar[i];
        const uint32_t L = code & kLenMask;
        const bool is_rc = kRC && (code >> 31);
        FactorRec f;
        f.start = i;
        if (L == 0) {  // literal: factorizer_core.hpp:83-88 / :305-316, 354-357
            f.length = 1;
            f.ref = i;
        } else {
            const uint32_t r = isa[i] - 1u;  // (1-based, pipeline.hpp)
            uint32_t lo = r, hi = r + 1, steps = 0;
            while (lcp[lo] >= L) {  // lcp[0] = 0 stops the scan
                --lo;
                if (++steps == kEmitScan) {
                    lo = (uint32_t)pyr_nearest_left<false>(Plcp, lo, L);
                    break;
                }
            }
            steps = 0;
            while (lcp[hi] >= L) {  // lcp[n] = 0 stops the scan
                ++hi;
                if (++steps == kEmitScan) {
                    hi = pyr_nearest_right<false>(Plcp, hi, L);
                    break;
                }
            }
            hi -= 1;
            f.length = L;
            if (!is_rc) {
                uint32_t mn;
                if (hi - lo < 2 * kEmitScan) {
                    mn = kNone;
                    for (uint32_t q = lo; q <= hi; ++q) {
                        const uint32_t v = sa[q];
                        mn = v < mn ? v : mn;
                    }
                } else {
                    mn = pyr_range<false>(Psa, lo, hi);
                }
                f.ref = mn;
            } else {
                uint32_t mx;
                if (hi - lo < 2 * kEmitScan) {
                    mx = 0;
                    for (uint32_t q = lo; q <= hi; ++q) {
                        const uint32_t v = sa[q];
                        mx = v > mx ? v : mx;
                    }
                } else {
                    mx = pyr_range<true>(Pmax, lo, hi);
                }
                const uint64_t end = 2ull * rcN - mx;  // T-coordinate of the last matched base
                f.ref = (1ull << 63) | (end - L + 1);
            }
        }
        if (kRebase) {
            const uint32_t t = term_lower_bound(terms, i);
            const uint64_t base = t ? (uint64_t)terms.pos[t - 1] + 1 : 0;
            f.start -= base;
            f.ref -= base;
        }
        out[k] = f;
    }
}

inline unsigned grid_for(size_t items, unsigned cap = 256u * 16u) {
    size_t g = div_up(items, (size_t)kThreads);
    if (g < 1) g = 1;
    return (unsigned)(g > cap ? cap : g);
}

}  // namespace

// Returns z; if d_factors_out != nullptr the z factor records are left in arena memory (this
// stage's temporaries are then NOT released: the caller rewinds after copying the records out).
// d_fpos_out (optional) receives the z factor starts, ascending, under the same rule -- without
// d_factors_out nothing else is built.  rebase (optional, plain mode): the terminator table of a text of
// independent records; start and ref of every factor are then relative to its record.
// Plain mode: rcN = 0, Pmax unused.  RC mode: n = N (factorized prefix of S), rcN = N and Pmax
// is the max pyramid over SA.
uint32_t resolve_chain(Context &ctx, uint32_t n, uint32_t start_pos, const uint32_t *lstar, const uint32_t *sa,
                       const uint32_t *isa, const uint32_t *lcp, const Pyramid &Psa, const Pyramid &Plcp,
                       void **d_factors_out, uint32_t rcN, const Pyramid *Pmax, uint32_t **d_fpos_out,
                       const TermTable *rebase) {
    hipStream_t s = ctx.stream;
    Arena &arena = ctx.arena;
    if (d_factors_out) *d_factors_out = nullptr;
    if (d_fpos_out) *d_fpos_out = nullptr;
    if (start_pos >= n) return 0;

    const uint32_t num_tiles = (uint32_t)div_up(n, kTile);
    const uint32_t tbits_words = (uint32_t)div_up((size_t)n + 1, 32);

    // persistent across the call: factor records (if requested) live below `mark`
    uint32_t *fpos = nullptr;
    const size_t mark = arena.mark();
    uint32_t *exit0 = arena.alloc<uint32_t>(n);
    uint32_t *tbits = arena.alloc<uint32_t>(tbits_words);
    uint32_t *tprefix = arena.alloc<uint32_t>(tbits_words);
    uint32_t *d_total = arena.alloc<uint32_t>(2);
    HIP_CHECK(hipMemsetAsync(tbits, 0, sizeof(uint32_t) * tbits_words, s));
    {
        ProfScope ps(ctx.profiler(), "chain_exit", s);
        chain_exit_kernel<<<num_tiles, kThreads, 0, s>>>(lstar, n, start_pos, exit0, tbits);
        KERNEL_CHECK();
    }
    {
        ProfScope ps(ctx.profiler(), "chain_nodes", s);
        popcount_kernel<<<grid_for(tbits_words), kThreads, 0, s>>>(tbits, tbits_words, tprefix);
        KERNEL_CHECK();
        scan_exclusive_add_u32(tprefix, tprefix, tbits_words, d_total, arena, s);
    }
    uint32_t K = 0;
    ctx.read_back(d_total, &K, 1);

    uint32_t *node_pos = arena.alloc<uint32_t>(K + 1);
    uint32_t *nxt[2] = {arena.alloc<uint32_t>(K + 1), arena.alloc<uint32_t>(K + 1)};
    uint32_t *reach = arena.alloc<uint32_t>(K + 1);
    uint32_t *entry = arena.alloc<uint32_t>(num_tiles);
    {
        ProfScope ps(ctx.profiler(), "chain_nodes", s);
        node_fill_kernel<<<grid_for(tbits_words), kThreads, 0, s>>>(tbits, tprefix, tbits_words, node_pos);
        KERNEL_CHECK();
        node_edges_kernel<<<grid_for((size_t)K + 1), kThreads, 0, s>>>(node_pos, K, exit0, n, tbits, tprefix,
                                                                       start_pos, nxt[0], reach);
        KERNEL_CHECK();
    }
    {
        ProfScope ps(ctx.profiler(), "chain_wyllie", s);
        int rounds = 1;
        while ((1ull << rounds) < (uint64_t)K + 1) ++rounds;
        int cur = 0;
        for (int r = 0; r < rounds; ++r) {
            wyllie_kernel<<<grid_for(K), kThreads, 0, s>>>(nxt[cur], nxt[cur ^ 1], K, reach);
            KERNEL_CHECK();
            cur ^= 1;
        }
        HIP_CHECK(hipMemsetAsync(entry, 0xff, sizeof(uint32_t) * num_tiles, s));
        tile_entries_kernel<<<grid_for(K), kThreads, 0, s>>>(node_pos, reach, K, entry);
        KERNEL_CHECK();
    }
    unsigned long long *cbits = arena.alloc<unsigned long long>((size_t)num_tiles * (kTile / 64));
    uint32_t *tile_count = arena.alloc<uint32_t>(num_tiles);
    {
        ProfScope ps(ctx.profiler(), "chain_mark", s);
        chain_mark_kernel<<<num_tiles, 64, 0, s>>>(lstar, n, entry, cbits, tile_count);
        KERNEL_CHECK();
        scan_exclusive_add_u32(tile_count, tile_count, num_tiles, d_total + 1, arena, s);
    }
    uint32_t z = 0;
    ctx.read_back(d_total + 1, &z, 1);
    if ((!d_factors_out && !d_fpos_out) || z == 0) {
        arena.rewind(mark);
        return z;
    }
    fpos = arena.alloc<uint32_t>(z);
    if (!d_factors_out) {  // the factor starts only (per-record counts of the merged batch)
        ProfScope ps(ctx.profiler(), "factor_emit", s);
        emit_positions_kernel<<<num_tiles, 64, 0, s>>>(cbits, tile_count, fpos);
        KERNEL_CHECK();
        *d_fpos_out = fpos;
        return z;
    }
    FactorRec *recs_tmp = arena.alloc<FactorRec>(z);
    {
        ProfScope ps(ctx.profiler(), "factor_emit", s);
        emit_positions_kernel<<<num_tiles, 64, 0, s>>>(cbits, tile_count, fpos);
        KERNEL_CHECK();
        if (rcN) {
            if (rebase) throw HipError("resolve_chain: record-relative output exists in plain mode only");
            factor_kernel<true, false><<<grid_for(z), kThreads, 0, s>>>(fpos, z, lstar, isa, sa, lcp, Psa, Plcp, *Pmax,
                                                                        rcN, recs_tmp, TermTable{}, n);
        } else if (rebase) {
            factor_kernel<false, true><<<grid_for(z), kThreads, 0, s>>>(fpos, z, lstar, isa, sa, lcp, Psa, Plcp, Psa,
                                                                        0u, recs_tmp, *rebase, n);
        } else {
            factor_kernel<false, false><<<grid_for(z), kThreads, 0, s>>>(fpos, z, lstar, isa, sa, lcp, Psa, Plcp, Psa,
                                                                         0u, recs_tmp, TermTable{}, n);
        }
        KERNEL_CHECK();
    }
    // the caller owns the arena mark: stage temporaries stay allocated until it rewinds
    *d_factors_out = recs_tmp;
    if (d_fpos_out) *d_fpos_out = fpos;
    return z;
}

}  // namespace nolzss
