// local_sort.hpp -- the sub-buckets of two most-significant-digit passes sorted in LDS (local_sort_kernel), with the
// regroup of round 0 on the way where it pays, and the host side of it (local_sort_sub_buckets).  A part of radix_sort.hip:
// included there, inside namespace nolzss, behind the radix kernels it builds on (rs_hist_kernel, rs_scatter_kernel, the
// segment-list kernels).
#pragma once
#ifndef NOLZSS_RADIX_SORT_HIP
#error "local_sort.hpp is a part of radix_sort.hip"
#endif

namespace {

// ---- sub-buckets sorted in LDS (round 4) -----------------------------------------------------------------
// After TWO most-significant-digit passes a text of 2^27 .. 2^30 suffixes lies in 65 536 sub-buckets of a few thousand
// pairs each: small enough for ONE workgroup to hold in registers and order by the remaining key digits through LDS --
// read once, written once in full lines -- where every further bucket-segmented pass over HBM costs a histogram
// (4 B per pair) and a scatter (16 B per pair, in runs of 64 bytes).  One workgroup of kLocalThreads threads per CU
// takes every gridDim-th sub-bucket, kLocalRows pairs per thread at most; a larger sub-bucket (skewed texts) is put on
// a list and goes through segmented passes afterwards.  With one workgroup per CU nothing else hides its memory
// phases: the pairs of the NEXT sub-bucket are loaded while this one is ranked, the stores of the last one drain
// meanwhile, and the barriers wait for the LDS only.
//
// Ranking: rows of 64 pairs in input order, ONE returning LDS atomic per pair on the wave's counter of its digit.
// That is stable only if the LDS serves the lanes of an instruction that meet at one address in lane order.  gfx950
// does (every wave checks it on its first row of every pass against the ballot form rs_scatter_kernel uses: the
// counters start at zero there, so a lane must be handed the number of lanes below it with its digit); if a check ever
// fails the kernel says so, the sub-buckets are redone by the segmented passes and the path is switched off.  The
// ballot form costs ~60 VALU instructions per row and made this kernel issue-bound at 7.2 ms per 2^30 pairs.
#ifndef NOLZSS_LOCAL_THREADS
#define NOLZSS_LOCAL_THREADS 768
#endif
#ifndef NOLZSS_LOCAL_ROWS
#define NOLZSS_LOCAL_ROWS 24
#endif
constexpr int kLocalThreads = NOLZSS_LOCAL_THREADS;
constexpr int kLocalWaves = kLocalThreads / 64;
constexpr int kLocalRows = NOLZSS_LOCAL_ROWS;
constexpr uint32_t kLocalCap = (uint32_t)kLocalThreads * kLocalRows;
static_assert(kLocalThreads >= kBins && kLocalThreads % 64 == 0, "the first kBins threads own one bin each in the offset phase");

// first element of every sub-bucket (bucket b, digit d) from the scanned table of the pass that made them
__global__ __launch_bounds__(kBins) void sub_starts_kernel(const uint32_t *__restrict__ scanned, const uint32_t *__restrict__ tile0,
                                                           const uint32_t *__restrict__ bstart, uint32_t n,
                                                           uint32_t *__restrict__ sub_start) {
    const uint32_t b = blockIdx.x, d = threadIdx.x;
    const uint32_t t0 = tile0[b], nt = tile0[b + 1] - t0;
    sub_start[b * kBins + d] = nt ? scanned[(size_t)t0 * kBins + (size_t)d * nt] : bstart[b];
    if (b == 0 && d == 0) sub_start[(size_t)gridDim.x * kBins] = n;
}

// A barrier that waits for the wave's LDS operations only.  __syncthreads() also waits for every global load and store
// the wave has in flight -- the prefetched pairs and the draining stores this kernel wants to leave in flight.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// block_scan_exclusive (common.hpp) on lds_barrier(); w / lane: the caller's wave number and lane
template <int NW>
__device__ __forceinline__ uint32_t block_scan_exclusive_add_lds(uint32_t v, uint32_t *lds, int w, int lane) {
    const uint32_t inc = wave_scan_inclusive_dpp(v, 0u, OpAdd<uint32_t>());
    // (the address is made anew every time: hoisted out of the loop over the sub-buckets it was spilled, and the reload from
    // scratch made the wave wait for every memory operation it had in flight, twice per sub-bucket)
    int ww = w;
    asm volatile("" : "+v"(ww));
    if (lane == 63) lds[ww] = inc;
    lds_barrier();
    uint32_t prefix = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const uint32_t t = lds[k];
        if (k < w) prefix += t;
    }
    lds_barrier();
    return prefix + inc - v;
}

// the next sub-bucket of this workgroup that fits it (empty ones skipped, larger ones put on the list); count = 0: none left
__device__ __forceinline__ void local_next(const uint32_t *__restrict__ sub_start, uint32_t num_sub, uint32_t &sub, uint32_t &first,
                                           uint32_t &count, uint32_t *__restrict__ ctl, uint32_t *__restrict__ large_list, int tid,
                                           uint32_t &which) {
    count = 0;
    first = 0;
    which = 0;
    while (sub < num_sub) {
        const uint32_t a = sub_start[sub], c = sub_start[sub + 1] - a;
        const uint32_t this_sub = sub;
        sub += gridDim.x;
        if (c == 0) continue;
        if (c > kLocalCap) {
            if (tid == 0) large_list[atomicAdd(&ctl[1], 1u)] = this_sub;
            continue;
        }
        first = a;
        count = c;
        which = this_sub;
        return;
    }
}

// one array of a sub-bucket's pairs into registers: a wave's stretch is rows * 64 pairs, row r of it the 64 pairs from
// r * 64.  ALL kLocalRows loads are issued whatever `rows` is (the places behind the end load the first pair again; their
// keys are set to all ones when they are ranked): a load inside a branch leaves the compiler unable to count what is in
// flight, and every wait for memory became a wait for everything -- the stores of the last sub-bucket included.
__device__ __forceinline__ void local_load(const uint32_t *__restrict__ in, uint32_t first, uint32_t count, int rows,
                                           uint32_t (&reg)[kLocalRows], int tid) {
    const uint32_t wbase = (uint32_t)(tid >> 6) * (uint32_t)(rows * 64) + (uint32_t)(tid & 63);
#pragma unroll
    for (int r = 0; r < kLocalRows; ++r) {
        const uint32_t local = wbase + (uint32_t)r * 64u;
        reg[r] = in[first + (local < count ? local : 0u)];
    }
}

// elements [lo, hi) of the staging buffer to the same elements of out (out 16-byte aligned): whole quads with one store,
// the up to three elements in front of the first whole quad and behind the last one by the first six threads.  No
// branch around a store (a lane without work is masked off): the number of stores in flight stays countable.
__device__ __forceinline__ void local_store(const uint32_t *s_stage, uint32_t *__restrict__ out, uint32_t lo, uint32_t hi, int tid) {
    constexpr int kQuadIters = (int)((kLocalCap / 4 + 1 + kLocalThreads - 1) / kLocalThreads);
    const uint32_t qlo = (lo + 3u) & ~3u, qhi = hi & ~3u;  // whole quads: [qlo, qhi)
#pragma unroll
    for (int j = 0; j < kQuadIters; ++j) {
        const uint32_t i0 = 4u * ((uint32_t)j * kLocalThreads + (uint32_t)tid);
        if (i0 >= qlo && i0 + 4u <= qhi) *reinterpret_cast<uint4 *>(out + i0) = *reinterpret_cast<const uint4 *>(s_stage + i0);
    }
    // (qlo > qhi only for a sub-bucket inside one quad: then [lo, hi) is its head)
    const uint32_t head_end = qlo < hi ? qlo : hi;
    const uint32_t e = (uint32_t)tid < 3u ? lo + (uint32_t)tid : (qhi > head_end ? qhi : head_end) + (uint32_t)tid - 3u;
    const bool on = (uint32_t)tid < 3u ? e < head_end : ((uint32_t)tid < 6u && e < hi && e >= head_end);
    if (on) out[e] = s_stage[e];
}

// ---- the regroup of round 0 inside local_sort_kernel ---------------------------------------------------------
// After its last digit a sub-bucket lies sorted in LDS: everything regroup_kernel<true, 3> (suffix_array.hip) would read
// back from HBM is at hand.  Groups never span two sub-buckets (their members differ in the first eight bases), so a
// workgroup finds the group heads, the LCP of every boundary the keys decide and the elements that stay tied by itself;
// what it needs from the others is the number of tied elements in front (a decoupled look-back over one descriptor per
// non-empty sub-bucket, lookback.hpp) and, for the LCP of its first boundary, the last key of the sub-bucket in front.
// The keys are not written at all: 4 bytes per suffix less out, 4 less in, and a kernel less.
constexpr uint32_t kLcpPendingCode = 0xffffffffu;  // (suffix_array.hip: kLcpPending)
// The sub-buckets are dealt out statically (that is what lets a workgroup ask for the next one's keys a turn ahead), so a
// look-back can wait for a workgroup that is not resident -- if another process or stream holds CUs with a kernel of its
// own that waits the same way, for as long as it likes.  The walk therefore gives up after ~0.2 s (2^17 polls of a
// microsecond and more); the host then sorts the text again with the plain kernel and the regroup kernel.
constexpr uint32_t kLocalSpinLimit = 1u << 17;
struct LocalFuse {
    uint32_t *lcp = nullptr, *new_slot = nullptr, *new_grp = nullptr, *d_total = nullptr;
    uint64_t *desc = nullptr;     // [non-empty sub-buckets] look-back descriptors of the numbers of tied elements
    uint64_t *lastkey = nullptr;  // [non-empty sub-buckets] [ready : 32 | last key : 32]
    const uint32_t *dense = nullptr;     // [sub-buckets] index among the non-empty ones
    const uint32_t *prev_sub = nullptr;  // [sub-buckets] nearest non-empty sub-bucket in front
    uint32_t nq = 0;
};

// dense index and predecessor of every non-empty sub-bucket; info = {sub-buckets beyond a workgroup's capacity,
// non-empty sub-buckets}.  One workgroup; num_sub is a multiple of its 1024 threads.
__global__ __launch_bounds__(1024) void sub_classify_kernel(const uint32_t *__restrict__ sub_start, uint32_t num_sub,
                                                            uint32_t *__restrict__ dense, uint32_t *__restrict__ prev_sub,
                                                            uint32_t *__restrict__ info) {
    __shared__ uint32_t s_scan[16];
    __shared__ uint32_t s_large;
    const uint32_t per = num_sub / 1024u, lo = threadIdx.x * per;
    if (threadIdx.x == 0) s_large = 0;
    __syncthreads();
    uint32_t ne = 0, last = 0, large = 0;  // last: index + 1 of the last non-empty sub-bucket of my stretch
    for (uint32_t k = lo; k < lo + per; ++k) {
        const uint32_t c = sub_start[k + 1] - sub_start[k];
        if (c) {
            ++ne;
            last = k + 1;
        }
        if (c > kLocalCap) ++large;
    }
    if (large) atomicAdd(&s_large, large);
    uint32_t all_ne, all_last;
    uint32_t q = block_scan_exclusive<16>(ne, OpAdd<uint32_t>(), s_scan, all_ne);
    uint32_t prev = block_scan_exclusive<16>(last, OpMax<uint32_t>(), s_scan, all_last);
    for (uint32_t k = lo; k < lo + per; ++k) {
        const uint32_t c = sub_start[k + 1] - sub_start[k];
        dense[k] = q;
        prev_sub[k] = prev - 1u;  // (0xffffffff: none in front)
        if (c) {
            ++q;
            prev = k + 1;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        info[0] = s_large;
        info[1] = all_ne;
    }
}

// local_store with buffer stores: a lane without work is sent out of bounds (the hardware drops the store) instead of being
// branched around -- straight-line code, a number of stores the compiler knows
__device__ __forceinline__ void local_store_buf(const uint32_t *s_stage, uint32_t *out, uint32_t lo, uint32_t hi, int tid) {
    constexpr int kQuadIters = (int)((kLocalCap / 4 + 1 + kLocalThreads - 1) / kLocalThreads);
    // (the addresses below depend on the thread alone: hoisted out of the loop over the sub-buckets they were spilled, and
    // a reload from scratch is a memory operation the wave then waits for with everything else in flight)
    asm volatile("" : "+v"(tid));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)(hi * 4u), 0x00020000);
    const uint32_t qlo = (lo + 3u) & ~3u, qhi = hi & ~3u;  // whole quads: [qlo, qhi)
    constexpr uint32_t kOut = 0xfffffff0u;
#pragma unroll
    for (int j = 0; j < kQuadIters; ++j) {
        const uint32_t i0 = 4u * ((uint32_t)j * kLocalThreads + (uint32_t)tid);
        const bool on = i0 >= qlo && i0 + 4u <= qhi;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = *reinterpret_cast<const u32x4 *>(s_stage + (on ? i0 : 0u));
        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, on ? i0 * 4u : kOut, 0, 0);
    }
    const uint32_t head_end = qlo < hi ? qlo : hi;
    const uint32_t e = (uint32_t)tid < 3u ? lo + (uint32_t)tid : (qhi > head_end ? qhi : head_end) + (uint32_t)tid - 3u;
    const bool on = (uint32_t)tid < 3u ? e < head_end : ((uint32_t)tid < 6u && e < hi && e >= head_end);
    __builtin_amdgcn_raw_buffer_store_b32(s_stage[on ? e : 0u], rsrc, on ? e * 4u : kOut, 0, 0);
}

// four zeros to LDS, the zeros made on the spot: kept in four registers from the start of the kernel they were spilled, and
// the reload from scratch made the wave wait for every memory operation it had in flight, in every pass
__device__ __forceinline__ void local_zero4(uint4 *p) {
    uint32_t z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    *p = make_uint4(z, z, z, z);
}

template <int NPASS, bool kFuse = false>
__global__ __launch_bounds__(kLocalThreads) void local_sort_kernel(
    const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, uint32_t *__restrict__ keys_out,
    uint32_t *__restrict__ vals_out, const uint32_t *__restrict__ sub_start, uint32_t num_sub, int shift0,
    uint32_t *__restrict__ ctl /* [1] sub-buckets on the list, [2] a lane-order check failed */, uint32_t *__restrict__ large_list,
    unsigned long long *__restrict__ phases, LocalFuse F = LocalFuse{}) {
#ifdef NOLZSS_LOCAL_TIMED
    unsigned long long ck_last = __builtin_readcyclecounter();
#define LOCAL_CK(slot)                                               \
    if (tid == 0) {                                                  \
        const unsigned long long now = __builtin_readcyclecounter(); \
        atomicAdd(phases + (slot), now - ck_last);                   \
        ck_last = now;                                               \
    }
#else
#define LOCAL_CK(slot)
#endif
    __shared__ __align__(16) uint32_t s_stage[kLocalCap + 4];  // keys, then values, take turns
    __shared__ __align__(16) uint32_t s_whist[kLocalWaves * kBins];
    __shared__ uint32_t s_scan[kLocalWaves];
    // (kFuse) per 64 places in place order: tied elements, place of the last head + 1 (then their exclusive prefixes);
    // first key, last key, tied elements in front of the sub-bucket
    __shared__ uint32_t s_cnt[kFuse ? (kLocalRows + 4) * kLocalWaves + 64 : 1], s_lh[kFuse ? (kLocalRows + 4) * kLocalWaves + 64 : 1], s_edge[3];
    __shared__ uint32_t s_giveup;
    __shared__ uint64_t s_mask[kFuse ? 2 * ((kLocalRows + 4) * kLocalWaves + 64) : 1];  // (kFuse) heads / tied elements of the 64 places, as lane masks
    const int tid = threadIdx.x;
    const int w = tid >> 6;
    const int lane = tid & 63;
    uint32_t *wcount = s_whist + w * kBins;  // this wave's counters: zeroed by the wave itself after every use
    local_zero4(reinterpret_cast<uint4 *>(wcount) + lane);
    static_assert(kBins == 256, "four counters per lane");

    uint32_t sub = blockIdx.x, first, count, cur_sub;
    local_next(sub_start, num_sub, sub, first, count, ctl, large_list, tid, cur_sub);
    uint32_t key[kLocalRows];
    local_load(keys_in, first, count, (int)((count + kLocalThreads - 1) / kLocalThreads), key, tid);
    bool order_ok = true;
    bool preset_failure = false;  // (test hook NOLZSS_TEST_LOCAL_LOOKBACK_FAILS: the flag is up before the kernel starts)
    if constexpr (kFuse) preset_failure = __hip_atomic_load(F.d_total + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    while (count) {  // (uniform)
        // this sub-bucket's values (wanted when its keys have been ranked once) and the NEXT one's keys are asked for
        // before anything else (the values of the next one, too, took the registers over the edge: a prefetched value
        // that is spilled is a value waited for)
        const int rows = (int)((count + kLocalThreads - 1) / kLocalThreads);
        uint32_t val[kLocalRows], nkey[kLocalRows];
        uint32_t nfirst = 0, ncount = 0, next_sub = 0;
        if constexpr (kFuse) {  // (the form with the regroup on the way has no register to spare for the later place below)
            local_load(vals_in, first, count, rows, val, tid);
            local_next(sub_start, num_sub, sub, nfirst, ncount, ctl, large_list, tid, next_sub);
            local_load(keys_in, nfirst, ncount, (int)((ncount + kLocalThreads - 1) / kLocalThreads), nkey, tid);
        }

        const uint32_t wbase = (uint32_t)w * (uint32_t)(rows * 64) + (uint32_t)lane;
        // the sorted pairs lie in the staging buffer from element `skew` = first & 3 on: a 16-byte quad of the buffer is a
        // 16-byte quad of the output, and a thread stores four pairs with one instruction (50 single stores per thread and
        // sub-bucket filled the queue of the memory pipeline that the loads behind them wait in)
        const uint32_t skew = first & 3u;
        uint32_t *stage = s_stage + skew;
        uint32_t lrank[kLocalRows];
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int shift = shift0 + pass * kRadixBits;
            // (keys of all ones behind the end of the sub-bucket: last in input order and in the last bin of every digit,
            // they stay behind its pairs through every stable pass -- no masks)
            if (pass == 0) {
#pragma unroll
                for (int r = 0; r < kLocalRows; ++r)
                    if (r < rows && wbase + (uint32_t)r * 64u >= count) key[r] = 0xffffffffu;
            }
#pragma unroll
            for (int r = 0; r < kLocalRows; ++r)
                if (r < rows) lrank[r] = atomicAdd(&wcount[digit_of(key[r], shift)], 1u);
            if (pass == 0 && !kFuse) {
                // This sub-bucket's values and the next one's keys are asked for HERE, behind the first ranking's atomics:
                // finding the next sub-bucket (two scalar loads) and issuing 48 loads took 8 k cycles at the start of
                // every turn with nothing else in flight.
                local_load(vals_in, first, count, rows, val, tid);
                local_next(sub_start, num_sub, sub, nfirst, ncount, ctl, large_list, tid, next_sub);
                LOCAL_CK(0)  // values asked for, the next sub-bucket found
                local_load(keys_in, nfirst, ncount, (int)((ncount + kLocalThreads - 1) / kLocalThreads), nkey, tid);  // (ncount = 0: the first key of the array, 24 times)
                LOCAL_CK(1)  // its keys asked for
            }
            {  // the lane-order check on row 0 (its counters started at zero)
                const uint32_t d = digit_of(key[0], shift);
                uint32_t diff_lo = 0, diff_hi = 0;
#pragma unroll
                for (int b = 0; b < kRadixBits; ++b) {
                    const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)d, (unsigned)b, 1u);
                    const uint64_t bal = __ballot((int)m < 0);
                    diff_lo = __builtin_amdgcn_bitop3_b32(m, diff_lo, (uint32_t)bal, 0xde);
                    diff_hi = __builtin_amdgcn_bitop3_b32(m, diff_hi, (uint32_t)(bal >> 32), 0xde);
                }
                const uint64_t below = ~(((uint64_t)diff_hi << 32) | diff_lo) & ((1ull << lane) - 1ull);
                order_ok = order_ok && lrank[0] == (uint32_t)__popcll(below);
            }
            LOCAL_CK(2 + 8 * pass)  // ranked (wave 0)
            lds_barrier();
            LOCAL_CK(3 + 8 * pass)  // ... everybody
            {  // thread = bin (the first kBins threads): per-wave counts -> start positions in the sub-bucket
                const int d = tid & (kBins - 1);
                const bool owner = tid < kBins;
                uint32_t c[kLocalWaves], total = 0;
#pragma unroll
                for (int k = 0; k < kLocalWaves; ++k) {
                    c[k] = owner ? s_whist[k * kBins + d] : 0u;
                    total += c[k];
                }
                const uint32_t bin_start = block_scan_exclusive_add_lds<kLocalWaves>(total, s_scan, w, lane);
                if (owner) {
                    uint32_t run = bin_start;
#pragma unroll
                    for (int k = 0; k < kLocalWaves; ++k) {
                        s_whist[k * kBins + d] = run;
                        run += c[k];
                    }
                }
            }
            lds_barrier();
            LOCAL_CK(4 + 8 * pass)  // offsets
#pragma unroll
            for (int r = 0; r < kLocalRows; ++r)
                if (r < rows) lrank[r] += wcount[digit_of(key[r], shift)];
            local_zero4(reinterpret_cast<uint4 *>(wcount) + lane);  // (after the wave's own reads, before its next atomics)
            if constexpr (kFuse) {
                if (pass + 1 == NPASS) {
                    // The last digit with the regroup of round 0 on the way.  The VALUES go first: their stores are
                    // buffer stores a lane without work sends out of bounds (no branch: the compiler can count them, and the
                    // wait for the next sub-bucket's keys below lets them drain).
#pragma unroll
                    for (int r = 0; r < kLocalRows; ++r)
                        if (r < rows) stage[lrank[r]] = val[r];
                    lds_barrier();
                    LOCAL_CK(18)  // (fused) values staged
                    local_store_buf(s_stage, vals_out + (first - skew), skew, skew + count, tid);
                    lds_barrier();
                    LOCAL_CK(19)  // values out
#pragma unroll
                    for (int r = 0; r < kLocalRows; ++r)
                        if (r < rows) stage[lrank[r]] = key[r];
#pragma unroll
                    for (int r = 0; r < kLocalRows; ++r) key[r] = nkey[r];  // (the next sub-bucket's keys take over)
                    lds_barrier();
                    LOCAL_CK(20)  // keys staged, next keys taken over
                    // What follows walks the sorted keys in the staging buffer in plain loops (place p = p0 + thread): no
                    // arrays in registers -- an unrolled form with 25 more registers spilled whatever was tried.
                    // Loop 1: heads, LCP of the boundaries, and per 64 places the tied elements and the last head.
                    const uint32_t bucket = cur_sub >> 8;
                    constexpr int kU = 4;  // (places of four rows per turn: their LDS reads go out together)
                    for (uint32_t p0 = 0, e0 = (uint32_t)w; p0 < count; p0 += kU * kLocalThreads, e0 += kU * kLocalWaves) {
                        uint32_t k[kU], edge[kU];
#pragma unroll
                        for (int u = 0; u < kU; ++u) {
                            const uint32_t pl = p0 + (uint32_t)u * kLocalThreads + (uint32_t)tid;
                            const bool valid = pl < count;
                            k[u] = stage[valid ? pl : 0u];
                            // (the neighbours are the neighbouring lanes' keys; lane 0 reads the place in front of the wave's
                            // 64, lane 63 the one behind them)
                            const uint32_t ep = lane == 0 ? pl - 1u : pl + 1u;
                            edge[u] = stage[(valid && ep < count) ? ep : 0u];
                        }
#pragma unroll
                        for (int u = 0; u < kU; ++u) {
                            const uint32_t pl = p0 + (uint32_t)u * kLocalThreads + (uint32_t)tid;
                            const bool valid = pl < count;
                            const uint32_t pk = (uint32_t)__builtin_amdgcn_update_dpp((int)edge[u], (int)k[u], 0x138, 0xf, 0xf, false);  // wave_shr:1
                            const uint32_t nk = (uint32_t)__builtin_amdgcn_update_dpp((int)edge[u], (int)k[u], 0x130, 0xf, 0xf, false);  // wave_shl:1
                            // (a suffix that ends inside the key window -- tag < 16 -- ties with nobody: a head, and so is
                            // whoever follows it; the first place of a sub-bucket is a head, and so is the one behind its last.
                            // Bitwise & and | on purpose: with && and || every condition became a branch on the lane mask,
                            // a hundred instructions per row of places.)
                            const bool head = valid & ((pl == 0u) | (k[u] != pk) | ((k[u] & 0xffu) < (uint32_t)kP16Syms));
                            const bool nhead = (pl + 1u >= count) | (nk != k[u]) | ((nk & 0xffu) < (uint32_t)kP16Syms);
                            const bool keep = valid & !(head & nhead);
                            {  // (place 0: below, with the last key of the sub-bucket in front)
                                const uint32_t y = (k[u] ^ pk) >> kP16TagBits, ta = k[u] & 0xffu, tb = pk & 0xffu;
                                uint32_t ls = (uint32_t)__clz((int)y) >> 1;  // (y = 0: 16, and no tag is larger)
                                ls = ls < ta ? ls : ta;
                                ls = ls < tb ? ls : tb;
                                const uint32_t l = head ? ls : kLcpPendingCode;
                                if (valid & (pl > 0u)) F.lcp[first + pl] = l;
                            }
                            const uint64_t hmask = __ballot(head), kmask = __ballot(keep);
                            if (lane == 0) {
                                const uint32_t e = e0 + (uint32_t)u * kLocalWaves;
                                s_cnt[e] = (uint32_t)__popcll(kmask);
                                s_lh[e] = hmask ? p0 + (uint32_t)u * kLocalThreads + (uint32_t)w * 64u + (uint32_t)(63 - __builtin_clzll(hmask)) + 1u : 0u;
                                s_mask[2 * e] = hmask;  // (the second loop reads the masks, not the keys)
                                s_mask[2 * e + 1] = kmask;
                            }
                            if (valid && pl == 0u) s_edge[0] = k[u];
                            if (valid && pl + 1u == count) s_edge[1] = k[u];
                        }
                    }
                    LOCAL_CK(21)  // loop 1 (wave 0)
                    lds_barrier();
                    LOCAL_CK(22)  // ... everybody
                    if (w == 0) {
                        // the first wave: prefixes over the (rows x waves) entries in place order, the look-back for the tied
                        // elements in front of the sub-bucket, the first boundary
                        const uint32_t ne = (uint32_t)rows * kLocalWaves;
                        constexpr int kPer = (kLocalRows * kLocalWaves + 63) / 64;
                        uint32_t c[kPer], h[kPer], sum = 0, mx = 0;
#pragma unroll
                        for (int j = 0; j < kPer; ++j) {
                            const uint32_t e = (uint32_t)lane * kPer + (uint32_t)j;
                            c[j] = e < ne ? s_cnt[e] : 0u;
                            h[j] = e < ne ? s_lh[e] : 0u;
                            sum += c[j];
                            mx = h[j] ? h[j] : mx;  // (positions grow along the entries: the last one that has a head)
                        }
                        const uint32_t isum = wave_scan_inclusive_dpp(sum, 0u, OpAdd<uint32_t>());
                        const uint32_t imax = wave_scan_inclusive_dpp(mx, 0u, OpMax<uint32_t>());
                        uint32_t run = isum - sum, lasth = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)imax, 0x138, 0xf, 0xf, false);  // (wave_shr:1)
                        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)isum, 63);
#pragma unroll
                        for (int j = 0; j < kPer; ++j) {
                            const uint32_t e = (uint32_t)lane * kPer + (uint32_t)j;
                            if (e < ne) {
                                s_cnt[e] = run;
                                s_lh[e] = lasth;
                            }
                            run += c[j];
                            lasth = h[j] ? h[j] : lasth;
                        }
                        const uint32_t q = F.dense[cur_sub];
                        if (lane == 0) desc_store(F.lastkey + q, (1ull << 32) | (uint64_t)s_edge[1]);
                        bool failed = preset_failure;
                        const uint32_t xs = lookback_exclusive_add_wide<4>(F.desc, q, total, F.d_total + 1, kLocalSpinLimit, &failed);
                        if (lane == 0) {
                            s_edge[2] = xs;
                            // (the look-back gave up: this workgroup stops behind this turn; the others give up on their own
                            // next walk.  Looking at the flag in memory every turn cost 0.5 ms here, 13 ms at the top of the turn.)
                            s_giveup = failed ? 1u : 0u;
                            if (q + 1u == F.nq) F.d_total[0] = xs + total;
                            uint32_t l0 = 0;  // (the very first suffix of the order)
                            if (q > 0u) {
                                uint64_t d;
                                uint32_t spins = 0;
                                while (((d = desc_load(F.lastkey + (q - 1u))) >> 32) == 0) {
                                    if (++spins > kLocalSpinLimit) {  // (never hang the GPU)
                                        atomicExch(F.d_total + 1, 1u);
                                        s_giveup = 1u;
                                        break;
                                    }
                                    __builtin_amdgcn_s_sleep(1);
                                }
                                const uint32_t pk = (uint32_t)d, k = s_edge[0];
                                const uint32_t y = ((bucket ^ (F.prev_sub[cur_sub] >> 8)) << 24) | ((k ^ pk) >> kP16TagBits);
                                const uint32_t ta = k & 0xffu, tb = pk & 0xffu;
                                uint32_t ls = y ? (uint32_t)__builtin_clz(y) >> 1 : 0xffffffffu;
                                ls = ls < ta ? ls : ta;
                                l0 = ls < tb ? ls : tb;
                            }
                            F.lcp[first] = l0;
                        }
                    }
                    LOCAL_CK(23)  // prefixes + look-back + first boundary (wave 0)
                    lds_barrier();
                    LOCAL_CK(24)  // ... everybody
                    // Loop 2: the tied elements -- slot and slot of the group's head, in slot order
                    const uint32_t xsum = s_edge[2];
                    const bool giveup = s_giveup != 0u;  // (uniform)
                    for (uint32_t p0 = 0, e = (uint32_t)w; p0 < count; p0 += kLocalThreads, e += kLocalWaves) {
                        const uint64_t hmask = s_mask[2 * e], kmask = s_mask[2 * e + 1];  // (wave-uniform)
                        if ((kmask >> lane) & 1ull) {
                            const uint32_t pl = p0 + (uint32_t)tid;
                            const uint64_t upto = hmask & ((2ull << lane) - 1ull);
                            const uint32_t hidx = upto ? p0 + (uint32_t)w * 64u + (uint32_t)(63 - __builtin_clzll(upto)) : s_lh[e] - 1u;
                            const uint32_t pos = xsum + s_cnt[e] + (uint32_t)__popcll(kmask & ((1ull << lane) - 1ull));
                            F.new_slot[pos] = first + pl;
                            F.new_grp[pos] = first + hidx;
                        }
                    }
                    LOCAL_CK(25)  // loop 2 (wave 0)
                    lds_barrier();  // (the staging buffer is free again)
                    LOCAL_CK(26)  // ... everybody
                    if (giveup) ncount = 0;  // (the loop over the sub-buckets ends)
                    continue;
                }
            }
            // keys through the staging buffer: back into the registers in the new order, or out; then the values
#pragma unroll
            for (int r = 0; r < kLocalRows; ++r)
                if (r < rows) stage[lrank[r]] = key[r];
            lds_barrier();
            LOCAL_CK(5 + 8 * pass)  // keys staged
            if (pass + 1 < NPASS) {
#pragma unroll
                for (int r = 0; r < kLocalRows; ++r)
                    if (r < rows) key[r] = stage[wbase + (uint32_t)r * 64u];
            } else {
                // the next sub-bucket's keys take over the registers HERE, before this one's stores are issued: waiting for
                // them later would mean waiting for every store in front of them (the memory counter runs in order)
#pragma unroll
                for (int r = 0; r < kLocalRows; ++r) key[r] = nkey[r];
                local_store(s_stage, keys_out + (first - skew), skew, skew + count, tid);
            }
            lds_barrier();
            LOCAL_CK(6 + 8 * pass)  // keys back / out
#pragma unroll
            for (int r = 0; r < kLocalRows; ++r)
                if (r < rows) stage[lrank[r]] = val[r];
            lds_barrier();
            LOCAL_CK(7 + 8 * pass)  // values staged
            if (pass + 1 < NPASS) {
#pragma unroll
                for (int r = 0; r < kLocalRows; ++r)
                    if (r < rows) val[r] = stage[wbase + (uint32_t)r * 64u];
            } else {
                local_store(s_stage, vals_out + (first - skew), skew, skew + count, tid);
            }
            lds_barrier();  // (the staging buffer is free again)
            LOCAL_CK(8 + 8 * pass)  // values back / out
        }
        first = nfirst;
        count = ncount;
        cur_sub = next_sub;
        LOCAL_CK(30)  // registers handed over
#ifdef NOLZSS_LOCAL_TIMED
        if (tid == 0) atomicAdd(phases + 31, 1ull);
#endif
    }
#undef LOCAL_CK
    if (!order_ok) atomicOr(&ctl[2], 1u);
}

std::atomic<bool> local_sort_off{false};  // a lane-order check of local_sort_kernel failed on this machine

// The buckets of a bucketed view (first elements bstart[0 .. num_buckets], first tiles tile0[]) have just been partitioned
// by one more digit (scanned = the scanned table of that pass, its result in keys_in / vals_in): every sub-bucket is sorted
// by npass further digits from shift0 up, into keys_out / vals_out.  local_sort_kernel does it in LDS; sub-buckets beyond
// a workgroup's capacity go through segmented passes.  keys_in / vals_in are scratch afterwards.
void local_sort_sub_buckets(uint32_t *keys_in, uint32_t *vals_in, uint32_t *keys_out, uint32_t *vals_out, const uint32_t *scanned,
                            const uint32_t *tile0, const uint32_t *bstart, uint32_t num_buckets, int shift0, int npass, size_t n,
                            Arena &arena, hipStream_t stream, Profiler *prof, Round0Regroup *rg = nullptr) {
    const uint32_t num_sub = num_buckets * (uint32_t)kBins;
    uint32_t *sub_start = arena.alloc<uint32_t>((size_t)num_sub + 1);
    uint32_t *large_list = arena.alloc<uint32_t>(num_sub);
    uint32_t *ctl = arena.alloc<uint32_t>(4);
    sub_starts_kernel<<<num_buckets, kBins, 0, stream>>>(scanned, tile0, bstart, (uint32_t)n, sub_start);
    KERNEL_CHECK();
    HIP_CHECK(hipMemsetAsync(ctl, 0, 4 * sizeof(uint32_t), stream));
    int dev = 0, cus = 0;
    HIP_CHECK(hipGetDevice(&dev));
    HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const unsigned grid = (unsigned)std::min<uint32_t>(num_sub, (uint32_t)(cus > 0 ? cus : 256));
    unsigned long long *d_ph = nullptr;
#ifdef NOLZSS_LOCAL_TIMED
    d_ph = arena.alloc<unsigned long long>(32);
    HIP_CHECK(hipMemsetAsync(d_ph, 0, 32 * sizeof(unsigned long long), stream));
#endif
#ifdef NOLZSS_LOCAL_TIMED
    auto print_phases = [&] {
        unsigned long long h[32];
        HIP_CHECK(hipMemcpyAsync(h, d_ph, sizeof(h), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        const double wn = h[31] ? (double)h[31] : 1.0;
        fprintf(stderr, "[nolzss] local_sort phases (cycles per sub-bucket, %llu sub-buckets): next found %.0f its loads issued %.0f |", h[31], h[0] / wn, h[1] / wn);
        for (int p = 0; p < npass; ++p)
            fprintf(stderr, " pass %d: rank(wave 0) %.0f rank(all) %.0f offsets %.0f stage keys %.0f keys back/out %.0f stage values %.0f values back/out %.0f |",
                    p, h[2 + 8 * p] / wn, h[3 + 8 * p] / wn, h[4 + 8 * p] / wn, h[5 + 8 * p] / wn, h[6 + 8 * p] / wn,
                    h[7 + 8 * p] / wn, h[8 + 8 * p] / wn);
        fprintf(stderr, " fused: stage values %.0f values out %.0f stage keys + take over %.0f loop1 %.0f (all %.0f) look-back %.0f (all %.0f) loop2 %.0f (all %.0f) |",
                h[18] / wn, h[19] / wn, h[20] / wn, h[21] / wn, h[22] / wn, h[23] / wn, h[24] / wn, h[25] / wn, h[26] / wn);
        fprintf(stderr, " hand-over %.0f\n", h[30] / wn);
    };
#endif
    static const bool fail_order = getenv("NOLZSS_TEST_LOCAL_ORDER_FAILS") != nullptr;  // (test hook: the redo path)
    static const bool no_fuse = getenv("NOLZSS_NO_LOCAL_REGROUP") != nullptr;            // (A/B switch)
    // (it pays where the regroup kernel is expensive -- many tied elements -- and the turns are long: 2^30 bases of the
    // benchmark text 21.7 -> 20.0 ms for sort + regroup, but random DNA of 2^29 bases 10.7 -> 11.2 and of 2^28 bases 6.3 -> 7.5:
    // the look-back and the two loops are a fixed cost per sub-bucket.  NOLZSS_LOCAL_REGROUP_MIN = smallest text that takes it.)
    static const size_t fuse_min = getenv("NOLZSS_LOCAL_REGROUP_MIN") ? (size_t)atoll(getenv("NOLZSS_LOCAL_REGROUP_MIN")) : (size_t(3) << 28);
    if (rg && !no_fuse && !fail_order && npass == 2 && num_sub % 1024u == 0 && n >= fuse_min) {
        // the regroup of round 0 on the way -- if no sub-bucket overflows a workgroup (the segmented passes that finish
        // those come after the kernel) and the kernel's checks hold; otherwise the plain form below runs from the same input
        uint32_t *dense = arena.alloc<uint32_t>(num_sub), *prev_sub = arena.alloc<uint32_t>(num_sub), *info = arena.alloc<uint32_t>(2);
        sub_classify_kernel<<<1, 1024, 0, stream>>>(sub_start, num_sub, dense, prev_sub, info);
        KERNEL_CHECK();
        uint32_t h_info[2];
        HIP_CHECK(hipMemcpyAsync(h_info, info, sizeof(h_info), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        if (h_info[0] == 0 && h_info[1] > 0) {
            LocalFuse F;
            F.lcp = rg->lcp; F.new_slot = rg->new_slot; F.new_grp = rg->new_grp; F.d_total = rg->d_total;
            F.nq = h_info[1];
            uint64_t *dd = arena.alloc<uint64_t>(2 * (size_t)F.nq);
            HIP_CHECK(hipMemsetAsync(dd, 0, 2 * (size_t)F.nq * sizeof(uint64_t), stream));
            HIP_CHECK(hipMemsetAsync(rg->d_total, 0, 2 * sizeof(uint32_t), stream));
            static const bool fail_lookback = getenv("NOLZSS_TEST_LOCAL_LOOKBACK_FAILS") != nullptr;  // (test hook: the way back)
            if (fail_lookback) HIP_CHECK(hipMemsetAsync(rg->d_total + 1, 1, 1, stream));  // (the flag a look-back sets when it gives up)
            F.desc = dd; F.lastkey = dd + F.nq; F.dense = dense; F.prev_sub = prev_sub;
            {
                ProfScope ps(prof, "rs_local_sort", stream, 16.0 * (double)n);  // (pairs in; suffixes and LCP out; the tied elements come on top)
                local_sort_kernel<2, true><<<grid, kLocalThreads, 0, stream>>>(keys_in, vals_in, keys_out, vals_out, sub_start, num_sub, shift0,
                                                                               ctl, large_list, d_ph, F);
                KERNEL_CHECK();
            }
            uint32_t h_c[4], h_t[2];
            HIP_CHECK(hipMemcpyAsync(h_c, ctl, sizeof(h_c), hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipMemcpyAsync(h_t, rg->d_total, sizeof(h_t), hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
#ifdef NOLZSS_LOCAL_TIMED
            print_phases();
#endif
            if (h_c[2] == 0 && h_t[1] == 0) {
                if (prof) prof->add_bytes("rs_local_sort", 8.0 * (double)h_t[0]);  // (slot and group of every tied element)
                rg->done = true;
                return;
            }
            if (h_t[1] && !fail_lookback) fprintf(stderr, "[nolzss] local_sort_kernel: look-back timed out; sorted again without the regroup on the way\n");
            HIP_CHECK(hipMemsetAsync(ctl, 0, 4 * sizeof(uint32_t), stream));  // (a failed lane-order check shows again below)
        }
    }
    {
        ProfScope ps(prof, "rs_local_sort", stream, 16.0 * (double)n);
        if (npass == 2)
            local_sort_kernel<2><<<grid, kLocalThreads, 0, stream>>>(keys_in, vals_in, keys_out, vals_out, sub_start, num_sub, shift0, ctl,
                                                                     large_list, d_ph);
        else if (npass == 3)
            local_sort_kernel<3><<<grid, kLocalThreads, 0, stream>>>(keys_in, vals_in, keys_out, vals_out, sub_start, num_sub, shift0, ctl,
                                                                     large_list, d_ph);
        else
            throw HipError("local_sort_sub_buckets: two or three digits");
        KERNEL_CHECK();
    }
#ifdef NOLZSS_LOCAL_TIMED
    print_phases();
#endif
    uint32_t h_ctl[4];
    HIP_CHECK(hipMemcpyAsync(h_ctl, ctl, sizeof(h_ctl), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    const bool redo_all = h_ctl[2] != 0 || fail_order;
    if (h_ctl[2]) {
        // the lane-order check failed somewhere: nothing the kernel wrote is trusted, and it is not asked again
        local_sort_off.store(true);
        fprintf(stderr, "[nolzss] local_sort_kernel: LDS atomics not served in lane order on this device; sorted by segmented passes instead\n");
    }
    const uint32_t nl = redo_all ? num_sub : h_ctl[1];
    if (nl == 0) return;
    // sub-buckets beyond a workgroup's capacity (skewed texts): segmented passes over them alone, in -> out -> in ..,
    // and (an even number of passes) their pairs copied to where the others already are
    std::vector<uint32_t> h_list(nl), h_sub((size_t)num_sub + 1);
    if (!redo_all) HIP_CHECK(hipMemcpyAsync(h_list.data(), large_list, sizeof(uint32_t) * nl, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(h_sub.data(), sub_start, sizeof(uint32_t) * h_sub.size(), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    if (redo_all) {
        h_list.clear();
        for (uint32_t k = 0; k < num_sub; ++k)
            if (h_sub[k + 1] > h_sub[k]) h_list.push_back(k);
    }
    std::sort(h_list.begin(), h_list.end());
    const uint32_t ns = (uint32_t)h_list.size();
    if (ns == 0) return;
    // first element, end, distance to its place among the segments of the list, first tile of every segment (+ the tile count)
    std::vector<uint32_t> h_seg(4 * (size_t)ns + 1);
    uint32_t *h_first = h_seg.data(), *h_end = h_first + ns, *h_shift = h_end + ns, *h_t0 = h_shift + ns;
    h_t0[0] = 0;
    uint32_t in_front = 0;
    for (uint32_t k = 0; k < ns; ++k) {
        h_first[k] = h_sub[h_list[k]];
        h_end[k] = h_sub[h_list[k] + 1];
        h_shift[k] = h_first[k] - in_front;
        in_front += h_end[k] - h_first[k];
        h_t0[k + 1] = h_t0[k] + (uint32_t)div_up((size_t)(h_end[k] - h_first[k]), kTile);
    }
    uint32_t *d_seg = arena.alloc<uint32_t>(h_seg.size());
    HIP_CHECK(hipMemcpyAsync(d_seg, h_seg.data(), sizeof(uint32_t) * h_seg.size(), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipStreamSynchronize(stream));  // h_seg is a local vector
    SegView big;
    big.num_tiles = h_t0[ns];
    uint32_t *big_mem = arena.alloc<uint32_t>((size_t)kSegDescWords * big.num_tiles + 4);
    const uint32_t *d_t0 = d_seg + 3 * (size_t)ns;
    seg_desc_list_kernel<<<(unsigned)div_up(big.num_tiles, kThreads), kThreads, 0, stream>>>(d_seg, d_seg + ns, d_t0, big.num_tiles, big_mem, ns);
    KERNEL_CHECK();
    big.desc = big_mem;
    uint32_t *big_hist = arena.alloc<uint32_t>((size_t)kBins * big.num_tiles);
    uint32_t *kbuf[2] = {keys_in, keys_out}, *vbuf[2] = {vals_in, vals_out};
    for (int p = 0; p < npass; ++p) {
        const ArraySrc<uint32_t> src{kbuf[p & 1], vbuf[p & 1]};
        const int shift = shift0 + 8 * p;
        rs_hist_kernel<uint32_t, ArraySrc<uint32_t>><<<xcd_grid(big.num_tiles), kThreads, 0, stream>>>(src, n, shift, big_hist, big.num_tiles, big);
        KERNEL_CHECK();
        scan_exclusive_add_u32(big_hist, big_hist, (size_t)kBins * big.num_tiles, nullptr, arena, stream);
        seg_table_shift_kernel<<<big.num_tiles, kBins, 0, stream>>>(big_hist, d_t0, d_seg + 2 * (size_t)ns, ns);
        KERNEL_CHECK();
        rs_scatter_kernel<uint32_t, uint32_t, ArraySrc<uint32_t>, uint32_t><<<xcd_grid(big.num_tiles), kThreads, 0, stream>>>(
            src, kbuf[(p & 1) ^ 1], vbuf[(p & 1) ^ 1], n, shift, big_hist, big.num_tiles, big);
        KERNEL_CHECK();
    }
    if ((npass & 1) == 0) {
        seg_copy_kernel<<<big.num_tiles, kThreads, 0, stream>>>(keys_in, vals_in, keys_out, vals_out, big);
        KERNEL_CHECK();
    }
}
}  // namespace
