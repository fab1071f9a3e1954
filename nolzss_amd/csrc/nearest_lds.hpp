// nearest_lds.hpp -- work-efficient "nearest qualifying suffix" searches on an LDS tile.
//
// Problem: for every rank r of a tile, find the nearest rank above / below whose suffix start
// qualifies (kGreater ? SA[q] > x : SA[q] < x) and the minimum LCP crossed on the way.  The
// distance to that rank is heavy-tailed (P(distance > k) ~ 1/(k+1) on random data), so a
// lock-step scan makes every wavefront pay for its slowest lane.
//
// Scheme (per wavefront, no workgroup barriers after staging -- shared work lists with a barrier
// per round were tried and lose: independent waves hide each other's LDS latency):
//   * the workgroup stages kLdsTile ranks of SA and LCP plus a halo of kLdsReach on both sides;
//   * round 0: every lane advances each of its searches by 4 steps, branch-free (ends 4 of 5);
//   * unfinished searches are compacted (ballot + popcount) into a per-wave work list in LDS
//     and the wave keeps taking 64 list items at a time, 8 more steps each, while the list holds
//     more than 24 items: total work ~ n * H(reach) instead of n * reach;
//   * the last items, each about as far from done as it has come, go to the four 16-lane rows of
//     the wave, one search per row, a row-wide DPP prefix minimum per iteration;
//   * a search that leaves the reach keeps its running LCP minimum as a bound (far_mark): the rank
//     is finished from global memory with the pyramids only if that bound can still win;
//   * PRUNING: the callers only use the LONGER of the two directions of a pair of searches (and queue the
//     rank for the exact search when that one overlaps), so a search whose running LCP minimum has dropped
//     to what the other direction has already FOUND can never matter and is dropped (length 0).  The
//     distance to the nearest qualifying rank is heavy-tailed, the LCP falls as the search moves away,
//     and the other direction usually ends within a few steps: on the benchmark text this removes 47 %
//     of all steps beyond the 20th and 43 % of the searches that leave the reach.
#pragma once
#include "nearest.hpp"

namespace nolzss {

constexpr int kLdsThreads = 256;
constexpr int kLdsWaves = kLdsThreads / 64;
constexpr int kLdsTile = 1024;                      // ranks per workgroup
constexpr int kLdsPerWave = kLdsTile / kLdsWaves;   // 256 ranks per wavefront
constexpr int kLdsReach = 256;                      // steps each way that stay inside LDS
constexpr int kLdsSpan = kLdsTile + 2 * kLdsReach;
constexpr int kLdsStep = 8;                         // steps per round
constexpr int kLdsStep0 = 4;                        // steps of the first round (4/5 of the searches end there)
constexpr uint32_t kFarLen = 0xffffffffu;           // res_len marker: search left the reach
// A search that leaves the reach still knows the running LCP minimum m it got to, and whatever it
// would find further out cannot be longer than m.  With far_bit = 0x80000000 (texts of at most 2^31
// symbols, where every length fits 31 bits) the marker is far_bit | m, and the caller sends a rank
// to the far queue only if that bound can still beat what the other direction found -- after 250
// ranks the bound is tiny, so 9 of 10 "far" searches need no second look.  far_bit = 0: plain marker.
__device__ __forceinline__ uint32_t far_mark(uint32_t m, uint32_t far_bit) {
    return far_bit ? (far_bit | (m & ~far_bit)) : kFarLen;
}
__device__ __forceinline__ bool far_is(uint32_t v, uint32_t far_bit) {
    return far_bit ? (v & far_bit) != 0 : v == kFarLen;
}
__device__ __forceinline__ uint32_t far_bound(uint32_t v, uint32_t far_bit) {
    return far_bit ? (v & ~far_bit) : 0xffffffffu;
}
// Combine the two directions of one kind of search: true if the rank must go to the far queue;
// otherwise a far side (which cannot win) is replaced by "nothing found".
__device__ __forceinline__ bool far_resolve(uint32_t &up, uint32_t &down, uint32_t far_bit, uint32_t floor_len) {
    const bool fu = far_is(up, far_bit), fd = far_is(down, far_bit);
    if (!fu && !fd) return false;
    const uint32_t bu = fu ? far_bound(up, far_bit) : 0u, bd = fd ? far_bound(down, far_bit) : 0u;
    uint32_t known = floor_len;  // lengths up to floor_len can never matter to the caller
    if (!fu) known = up > known ? up : known;
    if (!fd) known = down > known ? down : known;
    if ((fu && bu > known) || (fd && bd > known)) return true;
    if (fu) up = 0;
    if (fd) down = 0;
    return false;
}
constexpr uint32_t kLdsSparse = 24;                 // work-list length below which the wave gangs up

struct OpMinU32 {
    __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a < b ? a : b; }
};

// stage SA[base - reach, base + tile + reach) and the matching LCP entries; out-of-range = 0
__device__ __forceinline__ void stage_tile(const uint32_t *__restrict__ sa, const uint32_t *__restrict__ lcp,
                                           uint32_t n, uint32_t base, uint32_t *s_sa, uint32_t *s_lcp) {
    const int64_t first = (int64_t)base - kLdsReach;
    // every load of the thread goes out before the first LDS store (one round trip to HBM per
    // workgroup instead of one per row)
    constexpr int kRows = (kLdsSpan + 1 + kLdsThreads - 1) / kLdsThreads;
    uint32_t a[kRows], c[kRows];
#pragma unroll
    for (int k = 0; k < kRows; ++k) {
        const int j = k * kLdsThreads + (int)threadIdx.x;
        const int64_t g = first + j;
        const bool in_sa = j < kLdsSpan && g >= 0 && g < (int64_t)n;
        const bool in_lcp = j <= kLdsSpan && g >= 0 && g <= (int64_t)n;
        a[k] = in_sa ? sa[g] : 0u;
        c[k] = in_lcp ? lcp[g] : 0u;
    }
#pragma unroll
    for (int k = 0; k < kRows; ++k) {
        const int j = k * kLdsThreads + (int)threadIdx.x;
        if (j < kLdsSpan) s_sa[j] = a[k];
        if (j <= kLdsSpan) s_lcp[j] = c[k];
    }
}

// One round of kLdsStep steps of one search, branch-free.  li = local index of rank r in the
// staged tile, s0 = steps already taken, m = running LCP minimum (in/out).
// Returns 0 = finished without a match (len 0), 1 = match (m = its LCP, pos = its suffix start),
// 2 = still searching.  No bounds logic is needed: LCP[0] = LCP[n] = 0 (and 0 is staged outside
// the array), so the running minimum dies exactly when a search would leave the array.
template <int kSteps>
__device__ __forceinline__ int lds_scan_round(const uint32_t *s_sa, const uint32_t *s_lcp, int li, int s0,
                                              bool greater, bool up, uint32_t x, uint32_t &m, uint32_t &pos) {
    uint32_t c[kSteps], v[kSteps];
    const int dir = up ? -1 : 1;
#pragma unroll
    for (int k = 0; k < kSteps; ++k) {
        const int q = li + dir * (s0 + k + 1);
        c[k] = s_lcp[q + (up ? 1 : 0)];
        v[k] = s_sa[q];
    }
    // "v > x" as "~v < ~x": one comparison form for both kinds of search
    const uint32_t flip = greater ? 0xffffffffu : 0u;
    const uint32_t xf = x ^ flip;
#pragma unroll
    for (int k = 0; k < kSteps; ++k) {  // running minima
        m = c[k] < m ? c[k] : m;
        c[k] = m;
    }
    int status = 2;
    uint32_t len = m;
#pragma unroll
    for (int k = kSteps - 1; k >= 0; --k) {  // the earliest stopping step wins
        const bool dead = c[k] == 0;
        const bool stop = dead || (v[k] ^ flip) < xf;
        status = stop ? (dead ? 0 : 1) : status;
        len = stop ? c[k] : len;
        pos = stop ? v[k] : pos;
    }
    m = len;
    return status;
}

// Runs NS searches for each of the wavefront's kLdsPerWave ranks.  Search k is "up" for even k,
// "down" for odd k; searches 0/1 look for smaller values (threshold = own suffix start i),
// searches 2/3 (NS == 4) for values greater than thr_gt(i).  A rank takes part iff active(i).
// Results: res_len[k * kLdsTile + t] (0: none, far_mark(): left the reach) and, for the first NP
// searches only, res_pos[k * kLdsTile + t] (suffix start of the match).
// list0/list1: this wave's two work lists (NS * kLdsPerWave items each).
// kPrune: bit 0 = drop in the first round, bit 1 = in the work-list rounds, bit 2 = in the tail
template <int NS, int NP, int kPerLane, int kPrune, typename Active, typename ThrGt>
__device__ __forceinline__ void lds_search_wave(const uint32_t *s_sa, const uint32_t *s_lcp, uint32_t n,
                                                uint32_t base, uint32_t *res_len, uint32_t *res_pos,
                                                uint16_t *list0, uint16_t *list1, Active active, ThrGt thr_gt,
                                                uint32_t far_bit, unsigned long long *phase_clock = nullptr) {
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const uint64_t lt = lanemask_lt();
    uint16_t *lists[2] = {list0, list1};
    uint32_t cnt = 0;

    // ---- round 0: every rank, every search, steps 1..kLdsStep --------------------------------
#pragma unroll 1
    for (int row = 0; row < kLdsPerWave / 64; ++row) {
        const int tl = row * 64 + lane;
        const int t = w * kLdsPerWave + tl;
        const uint64_t rr = (uint64_t)base + t;
        const int li = t + kLdsReach;
        const uint32_t i = s_sa[li];
        const bool valid = rr < n && active(i);
        uint32_t m[NS], pos[NS];
        int st[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const bool greater = k >= 2, up = (k & 1) == 0;
            m[k] = 0xffffffffu;
            pos[k] = kNoPos;
            st[k] = 0;
            if (valid) st[k] = lds_scan_round<kLdsStep0>(s_sa, s_lcp, li, 0, greater, up, greater ? thr_gt(i) : i, m[k], pos[k]);
        }
        // a pending search that cannot beat what its partner has found already is dropped
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int o = k ^ 1;
            if ((kPrune & 1) && st[k] == 2 && st[o] == 1 && m[k] <= m[o]) st[k] = 0;
        }
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            res_len[k * kLdsTile + t] = (st[k] == 0) ? 0u : m[k];
            if (k < NP) res_pos[k * kLdsTile + t] = (st[k] == 1) ? pos[k] : kNoPos;
            const bool pending = st[k] == 2;
            const uint64_t bal = __ballot(pending);
            // item = rank in the wave | search << 8 | (steps taken / kLdsStep0) << 10
            if (pending) lists[0][cnt + (uint32_t)__popcll(bal & lt)] = (uint16_t)(tl | (k << 8) | (1 << 10));
            cnt += (uint32_t)__popcll(bal);
        }
    }

    if (phase_clock) phase_clock[0] = __builtin_readcyclecounter();
    // ---- drain the work list: 64 items at a time, kLdsStep more steps each --------------------
    // (while the list is long enough to keep most lanes busy)
    int cur = 0;
    while (cnt > kLdsSparse) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t next_cnt = 0;
        for (uint32_t c0 = 0; c0 < cnt; c0 += 64) {
            const bool have = c0 + lane < cnt;
            const uint32_t item = have ? lists[cur][c0 + lane] : 0u;
            const int tl = item & 255, k = (item >> 8) & 3, done = (int)(item >> 10) * kLdsStep0;
            const int t = w * kLdsPerWave + tl;
            const int li = t + kLdsReach;
            const bool greater = k >= 2, up = (k & 1) == 0;
            bool pending = false;
            if (have) {
                const uint32_t i = s_sa[li];
                uint32_t m = res_len[k * kLdsTile + t], pos = kNoPos;
                const int st = lds_scan_round<kLdsStep>(s_sa, s_lcp, li, done, greater, up,
                                                        greater ? thr_gt(i) : i, m, pos);
                int stp = st;
                if ((kPrune & 2) && st == 2 && (k | 1) < NP) {  // (the partner's position tells a found partner from a pending one)
                    const uint32_t ol = res_len[(k ^ 1) * kLdsTile + t], op = res_pos[(k ^ 1) * kLdsTile + t];
                    if (op != kNoPos && m <= ol) stp = 0;  // cannot beat what the other direction found
                }
                const bool at_reach = done + 2 * kLdsStep > kLdsReach;  // the next round would leave the halo
                pending = stp == 2 && !at_reach;
                res_len[k * kLdsTile + t] = (stp == 0) ? 0u : ((stp == 2 && at_reach) ? far_mark(m, far_bit) : m);
                if (k < NP) res_pos[k * kLdsTile + t] = (stp == 1) ? pos : kNoPos;
            }
            const uint64_t bal = __ballot(pending);
            if (pending)
                lists[cur ^ 1][next_cnt + (uint32_t)__popcll(bal & lt)] =
                    (uint16_t)(tl | (k << 8) | ((done + kLdsStep) / kLdsStep0 << 10));
            next_cnt += (uint32_t)__popcll(bal);
        }
        cur ^= 1;
        cnt = next_cnt;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    if (phase_clock) phase_clock[1] = __builtin_readcyclecounter();
    // ---- the long tail: few searches left, each possibly far from done (a search that has taken
    // s steps needs about s more).  Rounds of 8 steps would finish almost nothing per round, and
    // the kernel is bound by instruction issue, so the tail is organised for few instructions per
    // step: the wavefront splits into its four 16-lane rows, every row works on its own search,
    // each lane inspects kPerLane consecutive steps (1 for the forward-only searches of plain mode,
    // which mostly end within a few more steps; 4 = 64 steps per row and iteration for the
    // reverse-complement kernel, whose searches for greater values run longer), a row-wide
    // prefix minimum over the lanes' own minima (DPP) gives every step its running LCP, the first
    // lane that stops writes the result, and a row that is done takes the next search.
    {
        const int row = lane >> 4, rl = lane & 15;
        uint32_t next = 0;   // next list entry to hand out (wave-uniform)
        bool busy = false;   // my row has a search (row-uniform)
        int t = 0, k = 0, li = kLdsReach, s0 = 0;
        bool greater = false, up = false;
        uint32_t x = 0, m = 0;
        for (;;) {
            // idle rows take the next entries of the list, lowest row first
            const uint64_t idle = __ballot(!busy);
            const uint32_t idle_rows = (uint32_t)((idle & 1ull) | ((idle >> 15) & 2ull) | ((idle >> 30) & 4ull) |
                                                  ((idle >> 45) & 8ull));
            if (!busy) {
                const uint32_t e = next + (uint32_t)__popc(idle_rows & ((1u << row) - 1u));
                if (e < cnt) {
                    const uint32_t item = lists[cur][e];
                    k = (int)((item >> 8) & 3u);
                    t = w * kLdsPerWave + (int)(item & 255u);
                    s0 = (int)(item >> 10) * kLdsStep0;
                    li = t + kLdsReach;
                    greater = k >= 2;
                    up = (k & 1) == 0;
                    const uint32_t i = s_sa[li];
                    x = greater ? thr_gt(i) : i;
                    m = res_len[k * kLdsTile + t];
                    busy = true;
                }
            }
            next += (uint32_t)__popc(idle_rows);
            if ((kPrune & 4) && busy && (k | 1) < NP) {  // dropped if it cannot beat what the other direction has found
                const uint32_t ol = res_len[(k ^ 1) * kLdsTile + t], op = res_pos[(k ^ 1) * kLdsTile + t];
                if (op != kNoPos && m <= ol) {
                    if (rl == 0) res_len[k * kLdsTile + t] = 0u;  // (its position entry is still kNoPos)
                    busy = false;
                }
            }
            if (!__ballot(busy)) {
                if (next >= cnt) break;
                continue;  // every row dropped its search: take the next entries
            }
            // my four steps: s0 + 4 rl + 1 .. s0 + 4 rl + 4
            uint32_t c[kPerLane], v[kPerLane];
            bool in[kPerLane];
#pragma unroll
            for (int j = 0; j < kPerLane; ++j) {
                const int step = s0 + rl * kPerLane + j + 1;
                in[j] = step <= kLdsReach;
                const int q = in[j] ? (up ? li - step : li + step) : li;
                c[j] = (busy && in[j]) ? s_lcp[q + (up ? 1 : 0)] : 0xffffffffu;
                v[j] = s_sa[q];
            }
#pragma unroll
            for (int j = 1; j < kPerLane; ++j) c[j] = c[j] < c[j - 1] ? c[j] : c[j - 1];  // minima inside the lane
            // minimum over the lanes in front of me in my row (and over the steps taken before)
            uint32_t inc = c[kPerLane - 1];
            inc = OpMinU32()(inc, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)inc, 0x111, 0xf, 0xf, false));  // row_shr:1
            inc = OpMinU32()(inc, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)inc, 0x112, 0xf, 0xf, false));  // row_shr:2
            inc = OpMinU32()(inc, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)inc, 0x114, 0xf, 0xf, false));  // row_shr:4
            inc = OpMinU32()(inc, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)inc, 0x118, 0xf, 0xf, false));  // row_shr:8
            uint32_t pre = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)inc, 0x111, 0xf, 0xf, false);  // exclusive
            pre = pre < m ? pre : m;
            // the first of my steps that stops, if any
            bool lane_stop = false, f_hit = false, f_dead = false;
            uint32_t f_len = 0, f_pos = kNoPos;
#pragma unroll
            for (int j = kPerLane - 1; j >= 0; --j) {
                const uint32_t mk = c[j] < pre ? c[j] : pre;
                const bool dead = in[j] && mk == 0;  // LCP[0] = LCP[n] = 0 end every search in range
                const bool hit = in[j] && !dead && (greater ? (v[j] > x) : (v[j] < x));
                const bool stop = dead || hit || !in[j];
                lane_stop = stop || lane_stop;
                f_hit = stop ? hit : f_hit;
                f_dead = stop ? dead : f_dead;
                f_len = stop ? mk : f_len;
                f_pos = stop ? v[j] : f_pos;
            }
            const uint64_t stopb = __ballot(busy && lane_stop);
            const uint32_t mine = (uint32_t)(stopb >> (row * 16)) & 0xffffu;
            // running minimum at the end of each row, for the rows that go on
            const uint32_t incm = inc < m ? inc : m;
            const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)incm, 15);
            const uint32_t e1 = (uint32_t)__builtin_amdgcn_readlane((int)incm, 31);
            const uint32_t e2 = (uint32_t)__builtin_amdgcn_readlane((int)incm, 47);
            const uint32_t e3 = (uint32_t)__builtin_amdgcn_readlane((int)incm, 63);
            if (mine) {
                if (rl == __ffs((int)mine) - 1) {  // the first lane that stops decides
                    uint32_t out_len, out_pos = kNoPos;
                    if (f_hit) {
                        out_len = f_len;
                        out_pos = f_pos;
                    } else if (f_dead) {
                        out_len = 0;
                    } else {  // the end of the reach
                        out_len = far_mark(f_len, far_bit);
                    }
                    res_len[k * kLdsTile + t] = out_len;
                    if (k < NP) res_pos[k * kLdsTile + t] = out_pos;
                }
                busy = false;
            } else if (busy) {
                m = row == 0 ? e0 : (row == 1 ? e1 : (row == 2 ? e2 : e3));
                s0 += 16 * kPerLane;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}


// ---------------------------------------------------------------------------------------------------
// The same searches WITHOUT the long tail: block minima let a search skip 16 ranks per step.
//
// Phase timing of the list / row scheme above (clock per wavefront, 2^28 bases): staging 3.7 k cycles,
// first round 6.2 k, work-list rounds 6.5 k, the tail of <= 24 long searches 29 k, epilogue 6 k -- more than
// half of the kernel goes to a handful of searches that walk up to 256 ranks 16 at a time.  Here the
// workgroup also keeps, per aligned block of 16 staged ranks, the minimum (maximum) suffix start and the
// minimum LCP crossed when the block is passed upwards / downwards.  A search that is still going after
//   round 0   4 steps, every rank, branch-free (as above), and
//   round A   16 more steps, the unfinished searches 64 at a time (as above),
// continues by BLOCKS:
//   round B   the next 16 blocks away from the rank, nearest first: running minimum of the block LCP
//             minima, stop at the first block that holds a qualifying suffix or drives the minimum to 0;
//   round C   the 16 ranks of that block, nearest first, from the minimum carried so far: ends inside.
// Blocks that overlap ranks already passed are harmless (a minimum is idempotent and those ranks do not
// qualify).  No stop within the 16 blocks (>= 261 ranks away): the search leaves the reach with its bound,
// exactly as before.  Every search takes at most four rounds, all of them 64 searches wide.
constexpr int kBlk = 16;
constexpr int kNumBlk = kLdsSpan / kBlk;  // 96
static_assert(kLdsSpan % kBlk == 0 && kLdsReach % kBlk == 0 && kLdsTile % kBlk == 0, "aligned blocks");
constexpr int kStepA = 16;

struct BlockTables {
    uint32_t *mn, *mx;    // min / max suffix start of the block (mx may be null)
    uint32_t *lup, *ldn;  // min LCP crossed passing the block upwards (entries 16B+1 .. 16B+16) / downwards (16B .. 16B+15)
};

// run by the whole workgroup after the tile has been staged (and a barrier); followed by a barrier
template <bool kMax>
__device__ __forceinline__ void build_block_tables(const uint32_t *s_sa, const uint32_t *s_lcp, const BlockTables &T) {
    const int B = (int)threadIdx.x;
    if (B < kNumBlk) {
        const uint4 *p = reinterpret_cast<const uint4 *>(s_sa + kBlk * B);
        const uint4 *c = reinterpret_cast<const uint4 *>(s_lcp + kBlk * B);
        uint32_t mn = 0xffffffffu, mx = 0u, lo = 0xffffffffu;
        uint32_t first = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 a = p[q], l = c[q];
            const uint32_t av[4] = {a.x, a.y, a.z, a.w}, lv[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                mn = av[e] < mn ? av[e] : mn;
                mx = av[e] > mx ? av[e] : mx;
                if (q == 0 && e == 0)
                    first = lv[e];
                else
                    lo = lv[e] < lo ? lv[e] : lo;  // entries 16B+1 .. 16B+15
            }
        }
        const uint32_t last = s_lcp[kBlk * B + kBlk];
        T.mn[B] = mn;
        if (kMax) T.mx[B] = mx;
        T.ldn[B] = lo < first ? lo : first;
        T.lup[B] = lo < last ? lo : last;
    }
}

template <int NS, int NP, typename Active, typename ThrGt>
__device__ __forceinline__ void lds_search_wave_blocks(const uint32_t *s_sa, const uint32_t *s_lcp,
                                                       const BlockTables &T, uint32_t n, uint32_t base,
                                                       uint32_t *res_len, uint32_t *res_pos, uint16_t *list0,
                                                       uint16_t *list1, Active active, ThrGt thr_gt,
                                                       uint32_t far_bit, unsigned long long *phase_clock = nullptr) {
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const uint64_t lt = lanemask_lt();
    uint32_t cnt = 0;

    // ---- round 0: every rank, every search, steps 1..kLdsStep0 --------------------------------
#pragma unroll 1
    for (int row = 0; row < kLdsPerWave / 64; ++row) {
        const int tl = row * 64 + lane;
        const int t = w * kLdsPerWave + tl;
        const uint64_t rr = (uint64_t)base + t;
        const int li = t + kLdsReach;
        const uint32_t i = s_sa[li];
        const bool valid = rr < n && active(i);
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const bool greater = k >= 2, up = (k & 1) == 0;
            uint32_t m = 0xffffffffu, pos = kNoPos;
            int st = 0;
            if (valid) st = lds_scan_round<kLdsStep0>(s_sa, s_lcp, li, 0, greater, up, greater ? thr_gt(i) : i, m, pos);
            res_len[k * kLdsTile + t] = (st == 0) ? 0u : m;
            if (k < NP) res_pos[k * kLdsTile + t] = (st == 1) ? pos : kNoPos;
            const bool pending = st == 2;
            const uint64_t bal = __ballot(pending);
            if (pending) list0[cnt + (uint32_t)__popcll(bal & lt)] = (uint16_t)(tl | (k << 8));  // rank in the wave | search << 8
            cnt += (uint32_t)__popcll(bal);
        }
    }
    if (phase_clock) phase_clock[0] = __builtin_readcyclecounter();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- round A: kStepA more steps for the searches still going ---------------------------------
    uint32_t cnt_b = 0;
    for (uint32_t c0 = 0; c0 < cnt; c0 += 64) {
        const bool have = c0 + lane < cnt;
        const uint32_t item = have ? list0[c0 + lane] : 0u;
        const int tl = item & 255, k = (item >> 8) & 3;
        const int t = w * kLdsPerWave + tl;
        const int li = t + kLdsReach;
        const bool greater = k >= 2, up = (k & 1) == 0;
        bool pending = false;
        if (have) {
            const uint32_t i = s_sa[li];
            uint32_t m = res_len[k * kLdsTile + t], pos = kNoPos;
            const int st = lds_scan_round<kStepA>(s_sa, s_lcp, li, kLdsStep0, greater, up, greater ? thr_gt(i) : i, m, pos);
            pending = st == 2;
            res_len[k * kLdsTile + t] = (st == 0) ? 0u : m;
            if (k < NP) res_pos[k * kLdsTile + t] = (st == 1) ? pos : kNoPos;
        }
        const uint64_t bal = __ballot(pending);
        if (pending) list1[cnt_b + (uint32_t)__popcll(bal & lt)] = (uint16_t)item;
        cnt_b += (uint32_t)__popcll(bal);
    }
    if (phase_clock) phase_clock[1] = __builtin_readcyclecounter();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- rounds B + C: by blocks, then inside the block that stops the search ---------------------
    for (uint32_t c0 = 0; c0 < cnt_b; c0 += 64) {
        const bool have = c0 + lane < cnt_b;
        const uint32_t item = have ? list1[c0 + lane] : 0u;
        const int tl = item & 255, k = (item >> 8) & 3;
        const int t = w * kLdsPerWave + tl;
        const int li = t + kLdsReach;
        const bool greater = k >= 2, up = (k & 1) == 0;
        const uint32_t i = s_sa[li];
        const uint32_t x = greater ? thr_gt(i) : i;
        const uint32_t flip = greater ? 0xffffffffu : 0u;
        const uint32_t xf = x ^ flip;
        const uint32_t m_in = have ? res_len[k * kLdsTile + t] : 0xffffffffu;
        // first block beyond the kLdsStep0 + kStepA ranks already passed (it may overlap them)
        const int b0 = up ? (li - (kLdsStep0 + kStepA + 1)) >> 4 : (li + (kLdsStep0 + kStepA + 1)) >> 4;
        const uint32_t *tv = greater ? T.mx : T.mn;
        const uint32_t *tc = up ? T.lup : T.ldn;
        uint32_t c[kBlk], v[kBlk];
#pragma unroll
        for (int j = 0; j < kBlk; ++j) {
            const int B = up ? b0 - j : b0 + j;
            const bool in = have && B >= 0 && B < kNumBlk;
            const int Bc = in ? B : 0;
            const uint32_t cc = tc[Bc], vv = tv[Bc];
            c[j] = in ? cc : 0xffffffffu;
            v[j] = in ? (vv ^ flip) : 0xffffffffu;  // (flipped: "qualifies" is "< xf" for both kinds; out of range never does)
        }
        uint32_t run = m_in;
#pragma unroll
        for (int j = 0; j < kBlk; ++j) {
            run = c[j] < run ? c[j] : run;
            c[j] = run;
        }
        int jstar = -1;
        uint32_t m_before = run;  // no stop: the bound the search leaves the reach with
#pragma unroll
        for (int j = kBlk - 1; j >= 0; --j) {
            const bool stop = c[j] == 0 || v[j] < xf;
            jstar = stop ? j : jstar;
            m_before = stop ? (j ? c[j - 1] : m_in) : m_before;
        }
        const bool inside = have && jstar >= 0;
        // round C: the ranks of block Bs, nearest first: anchor = the rank just in front of the block
        const int Bs = up ? b0 - jstar : b0 + jstar;
        const int anchor = inside ? (up ? kBlk * Bs + kBlk : kBlk * Bs - 1) : li;
        uint32_t m = m_before, pos = kNoPos;
        const int st = lds_scan_round<kBlk>(s_sa, s_lcp, anchor, 0, greater, up, x, m, pos);
        if (have) {
            uint32_t out_len, out_pos = kNoPos;
            if (inside && st == 1) {
                out_len = m;
                out_pos = pos;
            } else if (inside && st == 0) {
                out_len = 0;
            } else {  // left the reach (st == 2 inside a stopping block cannot happen; it would be treated the same way)
                out_len = far_mark(m_before, far_bit);
            }
            res_len[k * kLdsTile + t] = out_len;
            if (k < NP) res_pos[k * kLdsTile + t] = out_pos;
        }
    }
    if (phase_clock) phase_clock[2] = __builtin_readcyclecounter();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

}  // namespace nolzss
