// group_sort.hpp -- the SECOND direct round of the suffix-array construction (included by suffix_array.hip
// inside its anonymous namespace, behind the first direct round's kernel).
//
// The first direct round (group_refine_kernel) finishes groups of up to 64 suffixes by comparing pairs,
// 1024 bases deep.  What it leaves on ordinary sequence data is little -- 0.6 % of the suffixes of the 2^30-base
// benchmark text -- but it used to cost the whole machinery behind it: rank[] written for every suffix (a full
// random permutation, 13 ms) so that six prefix-doubling rounds could look up rank[i + h] for that 0.6 %.
// What is left are (a) groups of more than 64 members -- regions with dozens to hundreds of copies, which
// diverge after a few dozen bases --, (b) groups whose pairs did not fit the first round's pair list, and (c) the
// few ties deeper than its cap.  None of them needs ranks: here ONE workgroup takes ONE group, whatever its size
// up to kGroupSortMax, and sorts it by the text itself, round by round:
//   * the members of a still-tied segment measure how far they agree with the segment's first member (a scan of
//     at most kScanWords words): the segment's depth jumps to the first symbol at which somebody differs -- a
//     deep tie costs a few rounds, not one round per window;
//   * every tied member fetches the 128 text bits at that depth (masked at its terminator; the distance to the
//     terminator and the terminator's index break ties the way text.hpp defines) and finds its place inside its
//     segment by counting the members with a smaller (window, terminator);
//   * neighbours that differ become segment boundaries, with LCP = depth + common prefix of the two windows.
// Output is what the first round hands to the regroup kernel: the new order in sa, per list position the number
// of strictly smaller members of its group (out_lo) and the LCP of every boundary that appeared (lcp_list).
// A group that runs out of its budget keeps what it has separated; the rest stays tied for the doubling rounds,
// which then pay for rank[] as before.
#pragma once

constexpr uint32_t kGroupSortSmall = 64;   // groups up to here: by tiles of the list, one wavefront per tile
constexpr uint32_t kGroupSortMid = 256;    // up to here: one wavefront per group (two queues: up to 128, up to 256 --
                                           // the LDS of a workgroup, i.e. how many groups a CU works on at a time)
constexpr uint32_t kGroupSortMid0 = 128;
constexpr uint32_t kGroupSortMax = 1024;   // up to here: 256 threads per group; beyond: left to the doubling rounds
constexpr uint32_t kScanWords = 8;         // 64-bit words a member scans per round to find where its segment splits
constexpr uint32_t kGroupSortRounds = 24;  // rounds before a group gives up (each: kScanWords words + one window)

// PIVOT rounds (round 4; kPivot): collections of similar sequences -- dozens to hundreds of copies of a genome, a
// fraction of a percent apart -- tie in groups as large as the collection, and somebody in such a group differs
// every few symbols: the window rounds above advance to the NEXT difference of ANY member, a hundred rounds for a
// thousand symbols.  A pivot round compares every member of a segment with the segment's FIRST member R only, as
// far as they agree (up to kPivotWords words per round, kPivotBatch words per trip to the text, all loads of a
// batch in flight): member x learns l = lcp(x, R), which of the two is smaller, and its own symbol at l.  That
// orders the whole segment at once -- the members below R by ascending l (x differs from R where y still agrees:
// x < y), then R and what equals it as far as was compared, then the members above R by descending l; equal l:
// by the symbol at l (the end of a suffix in front of every symbol, ends by terminator index, text.hpp) -- and
// neighbours of different classes have LCP = the smaller l.  Members with the same (side, l, symbol) agree on l + 1
// symbols and go on as a smaller segment with a pivot of their own.  With k copies that mutate independently a
// round finishes every member that differs from the consensus before the pivot does -- half of the segment --
// so a group is done in about log2(k) rounds, and every member reads the text about as far as its own LCP.
// Segments that agree up to depth_cap are left tied for the doubling rounds, which start there.
constexpr uint32_t kPivotWords = 32;  // words a member compares with its pivot per round (1024 bases of DNA)
constexpr uint32_t kPivotBatch = 8;   // ... of which this many per dependent trip to the text

// one queue entry per LARGE group of the leftover list (items = first list position, items2 = members); groups
// of up to kGroupSortSmall members are found by the tile kernel itself
__global__ __launch_bounds__(kThreads) void group_dir_kernel(const uint32_t *__restrict__ act_slot,
                                                             const uint32_t *__restrict__ act_grp, uint32_t m,
                                                             uint32_t h0, uint32_t *__restrict__ out_lo,
                                                             uint32_t *__restrict__ lcp_list, ShardQueue q_mid0,
                                                             ShardQueue q_mid, ShardQueue q_big,
                                                             uint32_t *__restrict__ min_depth,
                                                             const uint32_t *__restrict__ lcp_mark) {
    const uint32_t shard = blockIdx.x % kQShards;
    const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // (the grid covers the list once: every lane
    bool mid0 = false, mid = false, big = false;                     //  stays for the ballots of shard_slot)
    uint32_t first = 0, sz = 0;
    if (a < m) {
        out_lo[a] = 0;  // (a group nobody takes stays one group, in place)
        lcp_list[a] = kLcpPending;
        const uint32_t g = act_grp[a], j = act_slot[a] - g;
        // (lcp_mark, the equalising round: only the groups the first direct round did not touch -- the boundary behind
        // the group's first member still holds the plain pending code)
        if ((a + 1 == m || act_grp[a + 1] != g) && (lcp_mark == nullptr || lcp_mark[g + 1] == kLcpPending)) {  // the last member knows the size
            sz = j + 1;
            first = (uint32_t)a - j;
            mid0 = sz > kGroupSortSmall && sz <= kGroupSortMid0;
            mid = sz > kGroupSortMid0 && sz <= kGroupSortMid;
            big = sz > kGroupSortMid && sz <= kGroupSortMax;
            if (sz > kGroupSortMax) lower_min(min_depth, h0);

        }
    }
    const uint32_t s0 = shard_slot(q_mid0, shard, mid0);
    if (mid0) {
        q_mid0.items[s0] = first;
        q_mid0.items2[s0] = sz;
    }
    const uint32_t s1 = shard_slot(q_mid, shard, mid);
    const uint32_t s2 = shard_slot(q_big, shard, big);
    if (mid) {
        q_mid.items[s1] = first;
        q_mid.items2[s1] = sz;
    }
    if (big) {
        q_big.items[s2] = first;
        q_big.items2[s2] = sz;
    }
}

// 128 text bits of suffix p at depth h (two words), zero behind the member's terminator; tt = symbols before the
// terminator (capped at the window) << 16 | index of the terminator for a suffix that ends inside the window.
// Two suffixes that agree on everything in front of the window order by (k0, k1, tt): text.hpp, "nearer
// terminator first, then the lower index" -- a masked window equals a longer one only where that one continues
// with zero symbols, and then the shorter suffix (smaller tt) is the smaller one.
template <int BITS>
__device__ __forceinline__ void group_window(const uint64_t *__restrict__ words, uint32_t p, uint32_t h, uint32_t lim,
                                             uint32_t term, uint64_t &k0, uint64_t &k1, uint32_t &tt) {
    constexpr uint32_t kPer = 64 / BITS;
    const uint32_t rem = lim > h ? lim - h : 0u;
    const uint32_t tag = rem < 2 * kPer ? rem : 2 * kPer;
    uint64_t w0 = 0, w1 = 0;
    if (tag) {
        // three consecutive words cover both windows
        const uint64_t bit = ((uint64_t)p + h) * BITS;
        const uint64_t *src = words + (bit >> 6);
        const int o = (int)(bit & 63);
        const uint64_t a = src[0], b = src[1], c = src[2];
        w0 = o ? ((a << o) | (b >> (64 - o))) : a;
        w1 = o ? ((b << o) | (c >> (64 - o))) : b;
        if (tag < kPer) {
            w0 &= ~((~0ull) >> (tag * BITS));
            w1 = 0;
        } else if (tag == kPer) {
            w1 = 0;
        } else if (tag < 2 * kPer) {
            w1 &= ~((~0ull) >> ((tag - kPer) * BITS));
        }
    }
    k0 = w0;
    k1 = w1;
    tt = (tag << 16) | (tag < 2 * kPer ? (term & 0xffffu) : 0u);
}

// pivot rounds: the sort key of a member relative to its pivot, one 64-bit word (one LDS read and one comparison per
// step of the ranking): [side : 2 | l - D, descending above the pivot : 11 | symbol : 9] in the high half, the
// terminator index + 1 of a suffix that ends at l in the low half (D = depth of the segment, l - D <= kPivotWords
// words of symbols; symbol 0 = the suffix ends, else 1 + the member's symbol at l)
static_assert(kPivotWords * 32 <= 1024, "l - D fits 11 bits");
__device__ __forceinline__ uint64_t pivot_key(uint32_t side, uint32_t lrel, uint32_t sym, uint32_t tword) {
    const uint32_t k = (side << 30) | ((side == 2 ? 2047u - lrel : (side == 0 ? lrel : 0u)) << 9) | sym;
    return ((uint64_t)k << 32) | tword;
}
// l - D of a key (0xffffffff for the pivot's own class: every other class decides an LCP with it)
__device__ __forceinline__ uint32_t pivot_lrel_of(uint64_t key) {
    const uint32_t k = (uint32_t)(key >> 32), side = k >> 30, v = (k >> 9) & 2047u;
    return side == 1 ? 0xffffffffu : (side == 2 ? 2047u - v : v);
}
// kPivotBatch words of suffix p from symbol h on, as 2 * kPivotBatch 32-bit words in text order (the high half of
// a 64-bit word first): kPivotBatch + 1 loads, one funnel shift per 32 bits
template <int BITS>
__device__ __forceinline__ void pivot_fetch(const uint64_t *__restrict__ words, uint64_t p, uint32_t h, uint32_t (&win)[2 * kPivotBatch]) {
    const uint64_t bit = (p + h) * BITS;
    const uint64_t *src = words + (bit >> 6);
    uint32_t r[2 * kPivotBatch + 2];
#pragma unroll
    for (uint32_t k = 0; k <= kPivotBatch; ++k) {
        const uint64_t v = src[k];
        r[2 * k] = (uint32_t)(v >> 32);
        r[2 * k + 1] = (uint32_t)v;
    }
    const uint32_t skip = (bit & 32) ? 0xffffffffu : 0u;
    const uint32_t o = (uint32_t)bit & 31;
    uint32_t q[2 * kPivotBatch + 1];
#pragma unroll
    for (uint32_t k = 0; k <= 2 * kPivotBatch; ++k) q[k] = (r[k + 1] & skip) | (r[k] & ~skip);
#pragma unroll
    for (uint32_t k = 0; k < 2 * kPivotBatch; ++k) win[k] = o ? __builtin_amdgcn_alignbit(q[k], q[k + 1], 32 - o) : q[k];
}

// kTiled = false: one workgroup per queue entry (a group of up to NMAX members), grid (kQShards, Y).
// kTiled = true: one workgroup per tile of NMAX / 2 list positions; it takes the groups of up to NMAX / 2 members
//   that START in its tile -- all of them at once, as the segments the rounds begin with: sequence data leaves
//   hundreds of thousands of groups of two to eight members, and a wavefront per group would idle on their
//   round trips to the text.  List positions are sorted positions: a group's members hold consecutive list
//   positions (and slots), and sorting only ever moves members inside their group.
// kPivot: pivot rounds instead of window rounds (above); depth_cap: segments that agree that far stay tied.
template <int BITS, int THREADS, int NMAX, bool kTiled, bool kPivot = false>
__global__ __launch_bounds__(THREADS) void group_sort_kernel(ShardQueue q, const uint32_t *__restrict__ act_slot,
                                                             const uint32_t *__restrict__ act_grp, uint32_t m,
                                                             uint32_t *sa, const uint64_t *__restrict__ words,
                                                             TermTable terms, uint32_t h0,
                                                             uint32_t *__restrict__ out_lo,
                                                             uint32_t *__restrict__ lcp_list,
                                                             uint32_t *__restrict__ min_depth,
                                                             const uint32_t *__restrict__ lcp_mark, uint32_t max_rounds,
                                                             uint32_t depth_cap = 0xffffffffu) {
    constexpr uint32_t kPer = 64 / BITS;
    constexpr int kWavesB = THREADS / 64;
    constexpr uint32_t kTile = NMAX / 2;  // (tiled) list positions whose groups this workgroup takes
    static_assert(!kTiled || kTile == kGroupSortSmall, "a tile's span holds every small group that starts in it");
    // by member (= the position it was loaded from; never changes)
    __shared__ uint32_t s_pos[NMAX], s_lim[NMAX];
    __shared__ uint16_t s_term[NMAX], s_seg[NMAX];  // s_seg: sorted position of the first member of my segment
    __shared__ uint8_t s_act[NMAX];                 // my segment has more than one member
    // by sorted position
    __shared__ uint64_t s_key[NMAX], s_key2[NMAX];  // the window of the member at this position, and its
    __shared__ uint32_t s_tt[NMAX];                  // terminator word (group_window)
    __shared__ uint16_t s_ord[NMAX];       // member at this position
    __shared__ uint16_t s_gstart[NMAX];    // first position of the GROUP this position belongs to; 0xffff: not mine
    __shared__ uint8_t s_head[NMAX + 1];   // a segment starts here
    __shared__ uint8_t s_split[NMAX];      // [segment starts] a boundary appeared inside this segment in this round
    __shared__ uint8_t s_scanit[NMAX];     // [segment starts] the last round did not split it: scan ahead first
    __shared__ uint32_t s_blcp[NMAX];      // [new heads] LCP with the position in front
    __shared__ uint32_t s_depth[NMAX];     // [segment starts] symbols all members of the segment agree on
    __shared__ uint32_t s_tmp[NMAX];       // scratch: minimum of the scans, then the depth of the next round
    __shared__ uint32_t s_scan[kWavesB + 1];
    __shared__ uint32_t s_any;
    const uint32_t t = threadIdx.x;
    const uint32_t shard = blockIdx.x;
    const uint32_t count = kTiled ? 1u : q.counts[shard * kQPad];
    for (uint32_t item = kTiled ? 0u : blockIdx.y; item < count; item += kTiled ? 1u : gridDim.y) {
        // list positions [base, base + span): position e of the arrays is list element base + e
        uint32_t base, span, N;
        if (kTiled) {
            base = blockIdx.x * kTile;
            span = m - base < (uint32_t)NMAX ? m - base : (uint32_t)NMAX;
            N = NMAX;
        } else {
            base = q.items[(size_t)shard * q.cap + item];
            span = q.items2[(size_t)shard * q.cap + item];
            N = 2;
            while (N < span) N <<= 1;
        }
        __syncthreads();  // (the arrays of the group before this one are done with)
        if (kTiled) {  // sizes of the groups that start in the tile, told by their last member
            for (uint32_t e = t; e < (uint32_t)NMAX; e += THREADS) s_tmp[e] = 0;
            __syncthreads();
            for (uint32_t e = t; e < span; e += THREADS) {
                const uint32_t a = base + e, g = act_grp[a], j = act_slot[a] - g;
                if ((a + 1 == m || act_grp[a + 1] != g) && j <= e && e - j < kTile) s_tmp[e - j] = j + 1;
            }
            __syncthreads();
        }
        for (uint32_t e = t; e < N; e += THREADS) {
            bool mine = e < span;
            uint32_t j = e, slot = 0;
            if (kTiled && mine) {
                const uint32_t a = base + e, g = act_grp[a];
                slot = act_slot[a];
                j = slot - g;
                mine = j <= e && e - j < kTile;                    // my group starts in this tile ...
                if (mine) {
                    const uint32_t gs = s_tmp[e - j];
                    mine = gs != 0 && gs <= kGroupSortSmall;       // ... and is small
                    if (mine && lcp_mark != nullptr) mine = lcp_mark[g + 1] == kLcpPending;  // ... and was not compared yet
                }
            } else if (mine) {
                slot = act_grp[base] + e;  // (the group's members hold the slots from its head slot on)
            }
            uint32_t p = 0, lim = 0, term = 0;
            if (mine) {
                p = sa[slot];
                lim = term_limit(terms, p, term);
            }
            s_pos[e] = p;
            s_lim[e] = lim;
            s_term[e] = (uint16_t)term;
            // what is not mine is a segment of its own (it never moves) -- behind the span: sorts behind everything
            s_seg[e] = mine ? (uint16_t)(e - j) : (e < span ? (uint16_t)e : (uint16_t)0xffff);
            s_gstart[e] = mine ? (uint16_t)(e - j) : (uint16_t)0xffff;
            s_act[e] = mine ? 1 : 0;  // (a group of the list has at least two members)
            s_key[e] = 0;
            s_key2[e] = 0;
            s_tt[e] = 0;
            s_scanit[e] = 0;
            s_ord[e] = (uint16_t)e;
            s_head[e] = (!mine || j == 0) ? 1 : 0;
            s_blcp[e] = kLcpPending;
            s_depth[e] = h0;
        }
        if (t == 0) s_head[N] = 1;
        __syncthreads();
        bool tied = true;
        for (uint32_t round = 0; tied && round < max_rounds; ++round) {
            constexpr int kE = NMAX / THREADS;  // positions per thread: e = t + k * THREADS
            uint64_t rk0[kE], rk1[kE];
            uint32_t rtt[kE], rmem[kE], rhs[kE];
            bool ract[kE];
            if constexpr (kPivot) {
                // ---- 1p. every member against the first member of its segment
#pragma unroll
                for (int k = 0; k < kE; ++k) {
                    const uint32_t e = t + (uint32_t)k * THREADS;
                    ract[k] = false;
                    rk0[k] = rk1[k] = 0;
                    rtt[k] = rmem[k] = rhs[k] = 0;
                    if (e >= span) continue;
                    const uint32_t mem = s_ord[e];
                    if (!s_act[mem]) continue;
                    ract[k] = true;
                    rmem[k] = mem;
                    const uint32_t hs = s_seg[mem];
                    rhs[k] = hs;
                    s_split[e] = 0;
                    uint64_t key = pivot_key(1u, 0u, 0u, 0u);  // the pivot itself, and what equals it as far as compared
                    if (e != hs) {
                        const uint32_t D = s_depth[hs];
                        const uint32_t lead = s_ord[hs];
                        const uint32_t la = s_lim[mem], lb = s_lim[lead];
                        const uint32_t limit = la < lb ? la : lb;  // symbols before the nearer terminator
                        uint32_t stop = D + kPivotWords * kPer;
                        stop = stop < depth_cap ? stop : depth_cap;
                        const uint32_t stop2 = limit < stop ? limit : stop;
                        const uint64_t pa = s_pos[mem], pb = s_pos[lead];
                        constexpr uint32_t kPer32 = 32 / BITS;
                        uint32_t h = D, xw = 0, yw = 0;
                        bool found = false;
                        while (h < stop2) {
                            uint32_t x[2 * kPivotBatch], y[2 * kPivotBatch];
                            pivot_fetch<BITS>(words, pa, h, x);
                            pivot_fetch<BITS>(words, pb, h, y);
                            uint32_t fb = 2 * kPivotBatch;
#pragma unroll
                            for (int b = 2 * (int)kPivotBatch - 1; b >= 0; --b) {
                                const bool diff = x[b] != y[b];
                                fb = diff ? (uint32_t)b : fb;
                                xw = diff ? x[b] : xw;
                                yw = diff ? y[b] : yw;
                            }
                            if (fb < 2 * kPivotBatch) {
                                h += fb * kPer32 + (uint32_t)__clz((int)(xw ^ yw)) / BITS;
                                found = true;
                                break;
                            }
                            h += kPivotBatch * kPer;
                        }
                        if (found && h < stop2) {  // a difference in front of both terminators and of the round's reach
                            const uint32_t in_word = (uint32_t)__clz((int)(xw ^ yw)) / BITS;
                            const uint32_t sym = (xw >> (32 - BITS * (in_word + 1))) & ((1u << BITS) - 1u);
                            key = pivot_key(xw < yw ? 0u : 2u, h - D, sym + 1u, 0u);
                        } else if (limit <= stop) {  // a terminator is reached first
                            if (la < lb) {  // mine: the suffix that ends is the smaller one
                                key = pivot_key(0u, limit - D, 0u, (uint32_t)s_term[mem] + 1u);
                            } else if (la > lb) {  // the pivot's: I go on with a symbol of my own
                                const uint32_t sym = (uint32_t)(sym_word<BITS>(words, pa + limit) >> (64 - BITS));
                                key = pivot_key(2u, limit - D, sym + 1u, 0u);
                            } else {  // both end here: the lower terminator index first
                                key = pivot_key(s_term[mem] < s_term[lead] ? 0u : 2u, limit - D, 0u, (uint32_t)s_term[mem] + 1u);
                            }
                        }
                        // (else: equal as far as this round looks: the pivot's class)
                    }
                    rk0[k] = key;
                    s_key[e] = key;
                }
                __syncthreads();
            } else {
            // ---- 1. a segment that the last round did not split: how far does every member agree with its first
            // member?  (a scan of up to kScanWords words; the others take their window where they stand)
            for (uint32_t e = t; e < span; e += THREADS) {
                const uint32_t mem = s_ord[e];
                s_split[e] = 0;
                s_tmp[e] = (s_act[mem] && s_seg[mem] == e && !s_scanit[e]) ? s_depth[e] : 0xffffffffu;
            }
            __syncthreads();
            for (uint32_t e = t; e < span; e += THREADS) {
                const uint32_t mem = s_ord[e];
                if (!s_act[mem]) continue;
                const uint32_t hs = s_seg[mem];
                if (hs == e || !s_scanit[hs]) continue;  // (the first member itself: somebody else reports)
                const uint32_t d0 = s_depth[hs];
                const uint32_t lead = s_ord[hs];
                const uint32_t la = s_lim[mem], lb = s_lim[lead];
                const uint32_t limit = la < lb ? la : lb;
                const uint32_t stop = d0 + kScanWords * kPer;
                uint32_t h = d0;
                const uint64_t pa = s_pos[mem], pb = s_pos[lead];
                while (h < limit && h < stop) {
                    const uint64_t x = sym_word<BITS>(words, pa + h);
                    const uint64_t y = sym_word<BITS>(words, pb + h);
                    if (x != y) {
                        h += (uint32_t)__clzll((long long)(x ^ y)) / BITS;
                        break;
                    }
                    h += kPer;
                }
                h = h < limit ? h : limit;
                h = h < stop ? h : stop;
                atomicMin(&s_tmp[hs], h);
            }
            __syncthreads();
            // ---- 2. the window at the segment's new depth: kept in registers and parked BY POSITION for the ranking
#pragma unroll
            for (int k = 0; k < kE; ++k) {
                const uint32_t e = t + (uint32_t)k * THREADS;
                ract[k] = false;
                rk0[k] = rk1[k] = 0;
                rtt[k] = rmem[k] = rhs[k] = 0;
                if (e >= span) continue;
                const uint32_t mem = s_ord[e];
                if (!s_act[mem]) continue;
                ract[k] = true;
                rmem[k] = mem;
                rhs[k] = s_seg[mem];
                const uint32_t D = s_tmp[rhs[k]];
                group_window<BITS>(words, s_pos[mem], D, s_lim[mem], s_term[mem], rk0[k], rk1[k], rtt[k]);
                s_key[e] = rk0[k];
                s_key2[e] = rk1[k];
                s_tt[e] = rtt[k];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kE; ++k) {  // (s_depth of a segment start: read above, written here)
                const uint32_t e = t + (uint32_t)k * THREADS;
                if (ract[k] && rhs[k] == e) s_depth[e] = s_tmp[e];
            }
            }  // (!kPivot)
            // ---- 3. my place inside my segment: members with a smaller (window, terminator), plus the equal ones in
            // front of me.  Every lane walks its own segment (the members of a segment read the same entries: LDS
            // broadcasts); segments shrink round by round, so the walks get short quickly -- a sorting network over
            // the whole group paid log^2 N dependent LDS round trips in every round.
            uint32_t npos[kE];
#pragma unroll
            for (int k = 0; k < kE; ++k) {
                const uint32_t e = t + (uint32_t)k * THREADS;
                npos[k] = e;
                if (!ract[k]) continue;
                uint32_t less = 0, eqb = 0;
                if constexpr (kPivot) {  // one key word per member
                    for (uint32_t x = rhs[k];; ++x) {
                        if (x > rhs[k] && s_head[x]) break;
                        const uint64_t xk = s_key[x];
                        less += xk < rk0[k] ? 1u : 0u;
                        eqb += (xk == rk0[k] && x < e) ? 1u : 0u;
                    }
                    npos[k] = rhs[k] + less + eqb;
                    continue;
                }
                for (uint32_t x = rhs[k];; ++x) {
                    if (x > rhs[k] && s_head[x]) break;  // (s_head[N] = 1 ends the last segment)
                    const uint64_t xk0 = s_key[x], xk1 = s_key2[x];
                    const uint32_t xtt = s_tt[x];
                    const bool lt = xk0 != rk0[k] ? xk0 < rk0[k] : (xk1 != rk1[k] ? xk1 < rk1[k] : xtt < rtt[k]);
                    const bool eq = xk0 == rk0[k] && xk1 == rk1[k] && xtt == rtt[k];
                    less += lt ? 1u : 0u;
                    eqb += (eq && x < e) ? 1u : 0u;
                }
                npos[k] = rhs[k] + less + eqb;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kE; ++k) {
                if (!ract[k]) continue;
                s_ord[npos[k]] = (uint16_t)rmem[k];
                s_key[npos[k]] = rk0[k];
                s_key2[npos[k]] = rk1[k];
                s_tt[npos[k]] = rtt[k];
            }
            __syncthreads();
            // ---- 4. new boundaries inside the old segments, with their LCP; depth of the next round
            for (uint32_t e = t; e < span; e += THREADS) {
                const uint32_t mem = s_ord[e];
                if (!s_act[mem]) continue;
                const uint32_t hs = s_seg[mem];
                const uint32_t D = s_depth[hs];
                if constexpr (kPivot) {
                    // members that stay together: the pivot's class agrees as far as the round looked, the others
                    // on their common l and the symbol behind it
                    const uint64_t ka = s_key[e];
                    const uint32_t la = pivot_lrel_of(ka);
                    uint32_t stop = D + kPivotWords * kPer;
                    stop = stop < depth_cap ? stop : depth_cap;
                    s_tmp[e] = la == 0xffffffffu ? stop : D + la + 1u;
                    if (e == hs) continue;
                    const uint64_t kb = s_key[e - 1];
                    if (ka != kb) {
                        const uint32_t lb = pivot_lrel_of(kb);
                        s_head[e] = 1;
                        s_split[hs] = 1;
                        s_blcp[e] = D + (la < lb ? la : lb);
                    }
                    continue;
                }
                s_tmp[e] = D + 2 * kPer;  // members that stay together agree on the whole window
                if (e == hs) continue;
                const uint64_t ka = s_key[e], kb = s_key[e - 1], ka2 = s_key2[e], kb2 = s_key2[e - 1];
                const uint32_t ta = s_tt[e], tb = s_tt[e - 1];
                if (ka != kb || ka2 != kb2 || ta != tb) {
                    uint32_t d = 2 * kPer;
                    if (ka != kb)
                        d = (uint32_t)__clzll((long long)(ka ^ kb)) / BITS;
                    else if (ka2 != kb2)
                        d = kPer + (uint32_t)__clzll((long long)(ka2 ^ kb2)) / BITS;
                    const uint32_t va = ta >> 16, vb = tb >> 16;
                    const uint32_t valid = va < vb ? va : vb;
                    s_head[e] = 1;
                    s_split[hs] = 1;
                    s_blcp[e] = D + (d < valid ? d : valid);
                }
            }
            __syncthreads();
            // ---- 5. segments of the next round: start of my segment = nearest head at or in front of me
            {
                const uint32_t K = N > (uint32_t)THREADS ? N / THREADS : 1u;
                const uint32_t e0 = t * K;
                uint32_t last = 0;  // (position + 1) of the last head in my chunk
                if (e0 < N)
                    for (uint32_t k = 0; k < K; ++k)
                        if (s_head[e0 + k]) last = e0 + k + 1;
                uint32_t total;
                uint32_t run = block_scan_exclusive<kWavesB>(last, OpMax<uint32_t>(), s_scan, total);
                if (t == 0) s_any = 0;
                __syncthreads();
                bool any = false;
                if (e0 < N)
                    for (uint32_t k = 0; k < K; ++k) {
                        const uint32_t e = e0 + k;
                        if (s_head[e]) run = e + 1;
                        if (e >= span) continue;
                        const uint32_t mem = s_ord[e];
                        const bool was = s_act[mem] != 0;
                        if (!was) continue;  // (finished members, and what is not mine, keep their state)
                        const uint32_t hs = run - 1;
                        const uint32_t old_hs = s_seg[mem];
                        s_seg[mem] = (uint16_t)hs;
                        bool act = !(s_head[e] && s_head[e + 1]);
                        if constexpr (kPivot) {
                            // a segment that agrees up to the cap stays tied: the doubling rounds start at its depth
                            // (s_tmp of a position: the depth of the class it sits in, equal for the whole new segment)
                            if (act && s_tmp[hs] >= depth_cap) {
                                act = false;
                                if (hs == e) lower_min(min_depth, s_tmp[e]);
                            }
                        }
                        s_act[mem] = act ? 1 : 0;
                        any |= act;
                        if (hs == e) {  // a segment start (old or new) of a segment that was tied
                            s_depth[e] = s_tmp[e];
                            s_scanit[e] = kPivot ? 0 : (s_split[old_hs] ? 0 : 1);
                        }
                    }
                if (any) s_any = 1;
                __syncthreads();
                tied = s_any != 0;
            }
        }
        // ---- output
        if (tied) {  // the depth every group that stays tied agrees on (the doubling rounds start there)
            uint32_t dmin = 0xffffffffu;
            for (uint32_t e = t; e < span; e += THREADS) {
                const uint32_t mem = s_ord[e];
                if (s_act[mem] && s_seg[mem] == e) dmin = s_depth[e] < dmin ? s_depth[e] : dmin;
            }
            dmin = wave_reduce(dmin, OpMinU32x());
            if (lane_id() == 0 && dmin != 0xffffffffu) lower_min(min_depth, dmin);
        }
        for (uint32_t e = t; e < span; e += THREADS) {
            const uint32_t gs = s_gstart[e];
            if (gs == 0xffffu) continue;
            const uint32_t mem = s_ord[e];
            const uint32_t a = base + e;
            sa[kTiled ? act_slot[a] : act_grp[base] + e] = s_pos[mem];
            out_lo[a] = (uint32_t)s_seg[mem] - gs;  // strictly smaller members of my group
            lcp_list[a] = e == gs ? 0u : (s_head[e] ? s_blcp[e] : kLcpPending);
        }
    }
}
