// lookback.hpp -- decoupled look-back over per-tile descriptors in HBM: [status : value] in one 64-bit word, status 1 =
// the tile's own aggregate, 2 = inclusive prefix.  Shared by the regroup kernel (suffix_array.hip) and the sub-bucket
// sort that does the regroup of round 0 on the way (radix_sort.hip: local_sort_kernel).
#pragma once
#include "common.hpp"

namespace nolzss {

constexpr uint32_t kSpinLimit = 1u << 24;  // look-back polls before the kernel gives up (sets err)

__device__ __forceinline__ uint64_t desc_load(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void desc_store(uint64_t *p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Run by the first wavefront of a workgroup: publishes the tile's aggregate, combines the
// descriptors of the tiles in front (64 per step, nearest first) up to the first inclusive one,
// publishes the tile's inclusive prefix and returns its exclusive prefix.
template <typename Op>
__device__ __forceinline__ uint32_t lookback_exclusive(uint64_t *desc, uint32_t tile, uint32_t aggregate, Op op,
                                                       uint32_t *err) {
    const int lane = lane_id();
    if (tile == 0) {
        if (lane == 0) desc_store(desc, (2ull << 32) | aggregate);
        return Op::identity();
    }
    if (lane == 0) desc_store(desc + tile, (1ull << 32) | aggregate);
    uint32_t excl = Op::identity();
    int64_t look = (int64_t)tile - 1;
    for (;;) {
        const int64_t idx = look - lane;
        uint64_t d, need, inc;
        uint32_t spins = 0;
        for (;;) {
            d = idx >= 0 ? desc_load(desc + idx) : (2ull << 32);  // in front of tile 0: inclusive identity
            const uint32_t st = (uint32_t)(d >> 32);
            inc = __ballot(st == 2);
            // every lane up to and including the first inclusive one must have been published
            need = inc ? (((inc & (~inc + 1ull)) << 1) - 1ull) : ~0ull;
            const uint64_t missing = __ballot(st == 0) & need;
            if (!missing) break;
            if (++spins > kSpinLimit) {  // cannot happen with ticket order; never hang the GPU
                if (lane == 0) atomicExch(err, 1u);
                return excl;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        const uint32_t v = ((need >> lane) & 1ull) ? (uint32_t)d : Op::identity();
        excl = op(excl, wave_reduce(v, op));
        if (inc) break;  // an inclusive prefix was reached
        look -= 64;
    }
    if (lane == 0) desc_store(desc + tile, (2ull << 32) | op(excl, aggregate));
    return excl;
}

// The same for sums with kWin windows of 64 descriptors per round trip, nearest first: a walk that goes back over a few
// hundred tiles in flight (one workgroup per CU, every one of them a predecessor) takes one trip instead of four.
// (spin_limit: polls before it gives up and sets err -- a caller whose tiles are NOT dealt out in start order passes a small
// one: a tile in front may belong to a workgroup that is not resident yet, and the walk then has to end in a fallback)
template <int kWin>
__device__ __forceinline__ uint32_t lookback_exclusive_add_wide(uint64_t *desc, uint32_t tile, uint32_t aggregate, uint32_t *err,
                                                                uint32_t spin_limit = kSpinLimit, bool *gave_up = nullptr) {
    const int lane = lane_id();
    if (tile == 0) {
        if (lane == 0) desc_store(desc, (2ull << 32) | aggregate);
        return 0u;
    }
    if (lane == 0) desc_store(desc + tile, (1ull << 32) | aggregate);
    uint32_t excl = 0, spins = 0;
    int64_t look = (int64_t)tile - 1;
    for (;;) {
        uint64_t d[kWin];
#pragma unroll
        for (int j = 0; j < kWin; ++j) {
            const int64_t idx = look - 64 * j - lane;
            d[j] = idx >= 0 ? desc_load(desc + idx) : (2ull << 32);  // in front of tile 0: inclusive identity
        }
        bool done = false, stalled = false;
#pragma unroll
        for (int j = 0; j < kWin; ++j) {
            if (done || stalled) continue;  // (wave-uniform)
            const uint32_t st = (uint32_t)(d[j] >> 32);
            const uint64_t inc = __ballot(st == 2);
            // every lane up to and including the first inclusive one must have been published
            const uint64_t need = inc ? (((inc & (~inc + 1ull)) << 1) - 1ull) : ~0ull;
            const uint64_t missing = __ballot(st == 0) & need;
            if (missing) {  // not published yet: wait and read again from this window on
                stalled = true;
                continue;
            }
            const uint32_t v = ((need >> lane) & 1ull) ? (uint32_t)d[j] : 0u;
            excl += wave_reduce(v, OpAdd<uint32_t>());
            look -= 64;
            if (inc) done = true;  // an inclusive prefix was reached
        }
        if (done) break;
        if (stalled) {
            if (++spins > spin_limit) {  // (never hang the GPU)
                if (lane == 0) atomicExch(err, 1u);
                if (gave_up) *gave_up = true;
                return excl;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    if (lane == 0) desc_store(desc + tile, (2ull << 32) | (uint64_t)(excl + aggregate));
    return excl;
}

}  // namespace nolzss
