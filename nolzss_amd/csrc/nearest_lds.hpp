// nearest_lds.hpp -- work-efficient "nearest qualifying suffix" searches on an LDS tile.
//
// Problem: for every rank r of a tile, find the nearest rank above / below whose suffix start
// qualifies (kGreater ? SA[q] > x : SA[q] < x) and the minimum LCP crossed on the way.  The
// distance to that rank is heavy-tailed (P(distance > k) ~ 1/(k+1) on random data), so a
// lock-step scan makes every wavefront pay for its slowest lane.
//
// Scheme (per wavefront, no workgroup barriers after staging):
//   * the workgroup stages kLdsTile ranks of SA and LCP plus a halo of kLdsReach on both sides, and per
//     aligned block of 16 staged ranks the min (max) suffix start and the min LCP crossed passing it;
//   * round 0: every lane advances each of its searches by 4 steps, branch-free (ends 4 of 5);
//   * unfinished searches are compacted (ballot + popcount) into a per-wave work list in LDS and taken 64
//     at a time: 12 more steps (round A), then 16 BLOCKS nearest first (round B), then the 16 ranks of the
//     block that stops the search (round C) -- see lds_search_wave_blocks below;
//   * a search that leaves the reach keeps its running LCP minimum as a bound (far_mark): the rank
//     is finished from global memory with the pyramids only if that bound can still win.
// Round 1 ran the unfinished searches in work-list rounds of 8 steps and a tail of one search per 16-lane
// row; phase clocks (NOLZSS_LPF_PHASES) showed that tail taking 29 k of the 52 k cycles of a wavefront.
#pragma once
#include "nearest.hpp"

namespace nolzss {

constexpr int kLdsThreads = 256;
constexpr int kLdsWaves = kLdsThreads / 64;
constexpr int kLdsTile = 1024;                      // ranks per workgroup
constexpr int kLdsPerWave = kLdsTile / kLdsWaves;   // 256 ranks per wavefront
constexpr int kLdsReach = 256;                      // steps each way that stay inside LDS
constexpr int kLdsSpan = kLdsTile + 2 * kLdsReach;
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

struct OpMinU32 {
    __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a < b ? a : b; }
};

// stage SA[base - reach, base + tile + reach) and the matching LCP entries; out-of-range = 0
// (flag, optional: *flag |= 1 if a staged LCP entry is >= flag_min, i.e. still a "pending" code of the suffix-array
// construction -- the check build_lcp_pyramid makes on its first level, for callers that build that level here)
__device__ __forceinline__ void stage_tile(const uint32_t *__restrict__ sa, const uint32_t *__restrict__ lcp,
                                           uint32_t n, uint32_t base, uint32_t *s_sa, uint32_t *s_lcp,
                                           uint32_t flag_min = 0, uint32_t *flag = nullptr) {
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
    bool pending = false;
#pragma unroll
    for (int k = 0; k < kRows; ++k) {
        const int j = k * kLdsThreads + (int)threadIdx.x;
        if (j < kLdsSpan) s_sa[j] = a[k];
        if (j <= kLdsSpan) s_lcp[j] = c[k];
        pending |= flag != nullptr && c[k] >= flag_min;
    }
    if (pending) atomicOr(flag, 1u);  // (never on a finished LCP array)
}

// One round of kSteps steps of one search, branch-free.  li = local index of rank r in the
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
constexpr int kStepA = 12;
// (the first block of round B is the aligned block that holds the first rank not passed yet: it may reach back over
// passed ranks, never to the rank itself)
static_assert(kLdsStep0 + kStepA + 1 >= kBlk, "round B would look at the rank itself and at ranks on its other side");

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

// kCompact: only some of the ranks search (reverse-complement mode: the ranks of the original strand, half of
// them) -- they are gathered first, so that round 0 runs over full rows of searching ranks instead of spending
// its instructions on rows that are half idle.
template <int NS, int NP, bool kCompact = false, typename Active, typename ThrGt>
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
    uint32_t rows = kLdsPerWave / 64, n_act = kLdsPerWave;
    if (kCompact) {  // (list1 is free until round A)
        n_act = 0;
#pragma unroll
        for (int row = 0; row < kLdsPerWave / 64; ++row) {
            const int tl = row * 64 + lane;
            const int t = w * kLdsPerWave + tl;
            const bool valid = (uint64_t)base + t < n && active(s_sa[t + kLdsReach]);
            const uint64_t bal = __ballot(valid);
            if (valid) list1[n_act + (uint32_t)__popcll(bal & lt)] = (uint16_t)tl;
            n_act += (uint32_t)__popcll(bal);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        rows = (n_act + 63) / 64;
    }
#pragma unroll 1
    for (uint32_t row = 0; row < rows; ++row) {
        const bool have = !kCompact || row * 64 + lane < n_act;
        const int tl = kCompact ? (have ? (int)list1[row * 64 + lane] : 0) : (int)(row * 64 + lane);
        const int t = w * kLdsPerWave + tl;
        const uint64_t rr = (uint64_t)base + t;
        const int li = t + kLdsReach;
        const uint32_t i = s_sa[li];
        const bool valid = kCompact ? have : (rr < n && active(i));
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const bool greater = k >= 2, up = (k & 1) == 0;
            uint32_t m = 0xffffffffu, pos = kNoPos;
            int st = 0;
            if (valid) st = lds_scan_round<kLdsStep0>(s_sa, s_lcp, li, 0, greater, up, greater ? thr_gt(i) : i, m, pos);
            if (have) {  // (kCompact: the results of ranks that do not search are never read)
                res_len[k * kLdsTile + t] = (st == 0) ? 0u : m;
                if (k < NP) res_pos[k * kLdsTile + t] = (st == 1) ? pos : kNoPos;
            }
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
