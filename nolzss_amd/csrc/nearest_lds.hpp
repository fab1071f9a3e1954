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
//   * unfinished searches are compacted (ballot + popcount) into per-wave work lists in LDS, ONE LIST PER KIND
//     of search (smaller / greater suffix start, towards smaller / larger ranks: kind and direction are then
//     compile-time constants of every round, and an LDS address is the rank's own plus an immediate), and taken
//     64 at a time: 12 more steps (round A), then 16 BLOCKS nearest first (round B), then the 16 ranks of the
//     block that stops the search (round C) -- see lds_search_wave_blocks below;
//   * results per rank and search: the length (4 bytes) and the LOCAL INDEX of the match (2 bytes, match_pos);
//   * a search that leaves the reach keeps its running LCP minimum as a bound (far_mark): the rank
//     is finished from global memory with the pyramids only if that bound can still win.
// Round 1 ran the unfinished searches in work-list rounds of 8 steps and a tail of one search per 16-lane
// row; phase clocks (NOLZSS_LPF_PHASES) showed that tail taking 29 k of the 52 k cycles of a wavefront.
#pragma once
#include "nearest.hpp"

#include <type_traits>

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

// One round of kSteps steps of one search, branch-free; kind (kGreater) and direction (kUp) are compile-time, so
// every LDS address is the rank's own plus an immediate.  li = local index of rank r in the staged tile, s0 = steps
// already taken, m = running LCP minimum (in/out).
// Returns 0 = finished without a match (m = 0), 1 = match (m = its LCP, pos = its suffix start), 2 = still
// searching.  No bounds logic is needed: LCP[0] = LCP[n] = 0 (and 0 is staged outside the array), so the running
// minimum dies exactly when a search would leave the array.
// Four VALU instructions per step (minimum, comparison, two selects; the "still going" predicate lives in scalar
// registers).  A search does not stop when its minimum reaches 0: the minimum stays 0, whatever qualifies later is
// discarded by the m == 0 test at the end.  (Round 2 selected status, length and position per step from the last
// step backwards, with the direction a per-lane value: 12 VALU instructions per step; SQ_INSTS_VALU x 4 cycles /
// 1024 SIMDs was the whole 10.5 ms of lpf_tile_kernel.)
// at: local index (into the staged tile) of the match when 1 is returned -- the results keep that index, two bytes,
// instead of the suffix start (kNoMatch: none)
constexpr uint16_t kNoMatch = 0xffffu;
__device__ __forceinline__ uint32_t match_pos(const uint32_t *s_sa, uint16_t at) { return at == kNoMatch ? kNoPos : s_sa[at]; }
template <int kSteps, bool kGreater, bool kUp>
__device__ __forceinline__ int lds_scan_round(const uint32_t *s_sa, const uint32_t *s_lcp, int li, int s0, uint32_t x,
                                              uint32_t &m, uint32_t &at) {
    uint32_t c[kSteps], v[kSteps];
#pragma unroll
    for (int k = 0; k < kSteps; ++k) {
        const int q = kUp ? li - (s0 + k + 1) : li + (s0 + k + 1);
        c[k] = s_lcp[q + (kUp ? 1 : 0)];
        v[k] = s_sa[q];
    }
    bool alive = true;
    uint32_t kk = 0;
#pragma unroll
    for (int k = 0; k < kSteps; ++k) {
        const uint32_t mk = c[k] < m ? c[k] : m;
        const bool qual = kGreater ? v[k] > x : v[k] < x;
        const bool stop = alive && qual;
        m = alive ? mk : m;
        kk = stop ? (uint32_t)k : kk;
        alive = alive != stop;  // (alive && !qual, without a second comparison)
    }
    at = (uint32_t)(kUp ? li - (s0 + (int)kk + 1) : li + (s0 + (int)kk + 1));
    return m == 0 ? 0 : (alive ? 2 : 1);
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
//             minima, stop at the first block that holds a qualifying suffix (a minimum that reaches 0 stays 0:
//             whatever the search finds behind it is discarded, lds_scan_round);
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

// Every table has one entry in front of block 0 and one behind the last block that never stops a search (round B
// reads 16 blocks from b0 without a bounds test): kBlkTableLen words per table, the pointers below point at block 0.
constexpr int kBlkTableLen = kNumBlk + 2;
struct BlockTables {
    uint32_t *mn, *mx;    // min / max suffix start of the block (mx may be null)
    uint32_t *lup, *ldn;  // min LCP crossed passing the block upwards (entries 16B+1 .. 16B+16) / downwards (16B .. 16B+15)
};
// the tables of a kernel in one LDS array of kTables * kBlkTableLen words (kTables = 3: mn, lup, ldn; 4: mn, mx, lup, ldn)
template <bool kMax> __device__ __forceinline__ BlockTables block_tables(uint32_t *mem) {
    if (kMax) return BlockTables{mem + 1, mem + kBlkTableLen + 1, mem + 2 * kBlkTableLen + 1, mem + 3 * kBlkTableLen + 1};
    return BlockTables{mem + 1, nullptr, mem + kBlkTableLen + 1, mem + 2 * kBlkTableLen + 1};
}

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
    } else if (B < kNumBlk + 2) {  // the two entries outside: nothing qualifies there, no minimum changes
        const int o = B == kNumBlk ? -1 : kNumBlk;
        T.mn[o] = 0xffffffffu;
        if (kMax) T.mx[o] = 0u;
        T.ldn[o] = 0xffffffffu;
        T.lup[o] = 0xffffffffu;
    }
}

// Rounds A, B and C of ONE kind of search (kGreater, kUp) for the unfinished searches of a wavefront, 64 at a time.
// list: ranks (index in the wave) still searching after round 0, cnt of them; list_b: kListCapB >= 64 entries for the
// searches still going after round A -- when the next 64 might not fit, rounds B + C run on what is there first.
// (kListOverflow, a result of round 0 in lds_search_wave_blocks: "left the reach, nothing known", for a search that
// found list0 full -- the caller sends the rank to the searches from global memory.)
constexpr uint32_t kListOverflow = 0xffffffffu;  // (= far_mark(0x7fffffff, far_bit) and the plain far marker)
template <bool kGreater, bool kUp, bool kHasPos, int kListCapB, typename ThrGt>
__device__ __forceinline__ void lds_search_tail(const uint32_t *s_sa, const uint32_t *s_lcp, const BlockTables &T, int w,
                                                uint32_t *res_len, uint16_t *res_pos, const uint16_t *list, uint32_t cnt,
                                                uint16_t *list_b, ThrGt thr_gt, uint32_t far_bit) {
    static_assert(kListCapB >= 64, "a step of round A appends up to 64 searches");
    const int lane = lane_id();
    const uint64_t lt = lanemask_lt();
    const uint32_t *tv = kGreater ? T.mx : T.mn;
    const uint32_t *tc = kUp ? T.lup : T.ldn;
    // ---- rounds B + C for the first cnt_b entries of list_b: by blocks, then inside the block that stops the search ----
    auto rounds_bc = [&](uint32_t cnt_b) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t c0 = 0; c0 < cnt_b; c0 += 64) {
            const bool have = c0 + lane < cnt_b;
            const int tl = have ? (int)list_b[c0 + lane] : 0;
            const int t = w * kLdsPerWave + tl;
            const int li = t + kLdsReach;
            const uint32_t i = s_sa[li];
            const uint32_t x = kGreater ? thr_gt(i) : i;
            // first block beyond the kLdsStep0 + kStepA ranks already passed (it may overlap them); the 16 blocks from
            // there reach at most one block outside the staged span (the padded table entries)
            const int b0 = kUp ? (li - (kLdsStep0 + kStepA + 1)) >> 4 : (li + (kLdsStep0 + kStepA + 1)) >> 4;
            uint32_t c[kBlk], v[kBlk];
#pragma unroll
            for (int j = 0; j < kBlk; ++j) {
                const int B = kUp ? b0 - j : b0 + j;
                c[j] = tc[B];
                v[j] = tv[B];
            }
            // run: the minimum in front of the block that stops the search (the first one that holds a qualifying
            // suffix), or over all 16 blocks -- the bound the search leaves the reach with
            uint32_t run = have ? res_len[t] : 0xffffffffu;
            int jstar = -1;
            bool alive = true;
#pragma unroll
            for (int j = 0; j < kBlk; ++j) {
                const uint32_t rk = c[j] < run ? c[j] : run;
                const bool qual = kGreater ? v[j] > x : v[j] < x;
                const bool stop = alive && qual;
                jstar = stop ? j : jstar;
                alive = alive != stop;  // (alive && !qual, without a second comparison)
                run = alive ? rk : run;
            }
            const bool inside = have && jstar >= 0;
            // round C: the ranks of block Bs, nearest first: anchor = the rank just in front of the block
            const int Bs = kUp ? b0 - jstar : b0 + jstar;
            const int anchor = inside ? (kUp ? kBlk * Bs + kBlk : kBlk * Bs - 1) : li;
            uint32_t m = run, at = 0;
            const int st = lds_scan_round<kBlk, kGreater, kUp>(s_sa, s_lcp, anchor, 0, x, m, at);
            if (have) {
                // inside: the block holds a qualifying suffix, so the scan ends there (st 0 or 1); otherwise the
                // search left the reach
                res_len[t] = inside ? m : far_mark(run, far_bit);
                if (kHasPos) res_pos[t] = (inside && st == 1) ? (uint16_t)at : kNoMatch;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // ---- round A: kStepA more steps ------------------------------------------------------------
    uint32_t cnt_b = 0;
    for (uint32_t c0 = 0; c0 < cnt; c0 += 64) {
        if (kListCapB < kLdsPerWave && cnt_b + 64 > (uint32_t)kListCapB) {  // (wave-uniform) room for this step's searches
            rounds_bc(cnt_b);
            cnt_b = 0;
        }
        const bool have = c0 + lane < cnt;
        const int tl = have ? (int)list[c0 + lane] : 0;
        const int t = w * kLdsPerWave + tl;
        const int li = t + kLdsReach;
        bool pending = false;
        if (have) {
            const uint32_t i = s_sa[li];
            uint32_t m = res_len[t], at = 0;
            const int st = lds_scan_round<kStepA, kGreater, kUp>(s_sa, s_lcp, li, kLdsStep0, kGreater ? thr_gt(i) : i, m, at);
            pending = st == 2;
            res_len[t] = m;
            if (kHasPos) res_pos[t] = (st == 1) ? (uint16_t)at : kNoMatch;
        }
        const uint64_t bal = __ballot(pending);
        if (pending) list_b[cnt_b + (uint32_t)__popcll(bal & lt)] = (uint16_t)tl;
        cnt_b += (uint32_t)__popcll(bal);
    }
    rounds_bc(cnt_b);
}

// kCompact: only some of the ranks search (reverse-complement mode: the ranks of the original strand, half of
// them) -- they are gathered first, so that round 0 runs over full rows of searching ranks instead of spending
// its instructions on rows that are half idle.
// list0 / list1: NS lists of kListCap / kListCapB entries each (kCompact: NS * kListCapB >= kLdsPerWave, the gathered
// ranks), one per kind of search (search k: greater = k >= 2,
// up = k even), so that every round runs with kind and direction known at compile time.
template <int NS, int NP, bool kCompact = false, int kListCap = kLdsPerWave, int kListCapB = kListCap, typename Active, typename ThrGt>
__device__ __forceinline__ void lds_search_wave_blocks(const uint32_t *s_sa, const uint32_t *s_lcp,
                                                       const BlockTables &T, uint32_t n, uint32_t base,
                                                       uint32_t *res_len, uint16_t *res_pos, uint16_t *list0,
                                                       uint16_t *list1, Active active, ThrGt thr_gt,
                                                       uint32_t far_bit, unsigned long long *phase_clock = nullptr) {
    static_assert(NS == 2 || NS == 4, "searches: smaller up / down, then greater up / down");
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const uint64_t lt = lanemask_lt();
    uint32_t cnt[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) cnt[k] = 0;

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
        auto search = [&](auto kk) {
            constexpr int k = decltype(kk)::value;
            constexpr bool greater = k >= 2, up = (k & 1) == 0;
            uint32_t m = 0xffffffffu, at = 0;
            int st = 0;
            if (valid) st = lds_scan_round<kLdsStep0, greater, up>(s_sa, s_lcp, li, 0, greater ? thr_gt(i) : i, m, at);
            if (have) {  // (kCompact: the results of ranks that do not search are never read)
                res_len[k * kLdsTile + t] = (st == 0) ? 0u : m;
                if (k < NP) res_pos[k * kLdsTile + t] = (st == 1) ? (uint16_t)at : kNoMatch;
            }
            const bool pending = st == 2;
            const uint64_t bal = __ballot(pending);
            const uint32_t slot = cnt[k] + (uint32_t)__popcll(bal & lt);
            if (pending) {
                if (kListCap >= kLdsPerWave || slot < (uint32_t)kListCap)
                    list0[k * kListCap + slot] = (uint16_t)tl;
                else
                    res_len[k * kLdsTile + t] = kListOverflow;
            }
            cnt[k] += (uint32_t)__popcll(bal);
            if (kListCap < kLdsPerWave && cnt[k] > (uint32_t)kListCap) cnt[k] = (uint32_t)kListCap;
        };
        search(std::integral_constant<int, 0>());
        search(std::integral_constant<int, 1>());
        if constexpr (NS > 2) {
            search(std::integral_constant<int, 2>());
            search(std::integral_constant<int, 3>());
        }
    }
    if (phase_clock) phase_clock[0] = __builtin_readcyclecounter();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- rounds A, B, C per kind of search ---------------------------------------------------------
    static_assert(!kCompact || NS * kListCapB >= kLdsPerWave, "list1 first holds the gathered ranks");
    lds_search_tail<false, true, (0 < NP), kListCapB>(s_sa, s_lcp, T, w, res_len, res_pos, list0, cnt[0], list1, thr_gt, far_bit);
    lds_search_tail<false, false, (1 < NP), kListCapB>(s_sa, s_lcp, T, w, res_len + kLdsTile, res_pos + kLdsTile, list0 + kListCap,
                                                       cnt[1], list1 + kListCapB, thr_gt, far_bit);
    if constexpr (NS > 2) {
        lds_search_tail<true, true, (2 < NP), kListCapB>(s_sa, s_lcp, T, w, res_len + 2 * kLdsTile, res_pos + (2 < NP ? 2 : 0) * kLdsTile,
                                                         list0 + 2 * kListCap, cnt[2], list1 + 2 * kListCapB, thr_gt, far_bit);
        lds_search_tail<true, false, (3 < NP), kListCapB>(s_sa, s_lcp, T, w, res_len + 3 * kLdsTile, res_pos + (3 < NP ? 3 : 0) * kLdsTile,
                                                          list0 + 3 * kListCap, cnt[3], list1 + 3 * kListCapB, thr_gt, far_bit);
    }
    if (phase_clock) phase_clock[1] = phase_clock[2] = __builtin_readcyclecounter();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

}  // namespace nolzss
