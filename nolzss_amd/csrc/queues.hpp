// queues.hpp -- sharded device work queues.
//
// A queue that every wavefront of a large kernel appends to through ONE counter is bound by the rate of
// atomics on a single address: about 13 ns each on MI355X whatever the number of CUs (the memory side
// executes them one after the other).  The candidate kernel of a 2^30-base text made 1.9 M such appends
// (one per wavefront with a far search) = 25 of its 29 ms, hidden behind "waiting" in every counter.
// Here a queue has kQShards counters, 128 bytes apart, and as many regions; a workgroup appends to the
// shard blockIdx.x % kQShards with one atomic per wavefront.  A region can never overflow when every
// workgroup appends at most `per_block` items: cap = ceil(blocks / kQShards) * per_block.
// Consumers are launched on a (kQShards, Y) grid and walk their shard's region (shard_begin / _end).
#pragma once
#include "common.hpp"

namespace nolzss {

constexpr uint32_t kQShards = 256;
constexpr uint32_t kQPad = 32;  // counters 128 bytes apart

struct ShardQueue {
    uint32_t *items = nullptr;   // kQShards regions of `cap` entries
    uint32_t *items2 = nullptr;  // optional second payload, same slots
    uint32_t *counts = nullptr;  // kQShards * kQPad words, zeroed before use
    uint32_t cap = 0;
};

inline size_t shard_queue_cap(size_t blocks, size_t per_block) { return div_up(blocks, kQShards) * per_block; }

// Slot for this lane's item (0xffffffff for lanes with want == false): one atomic per wavefront.
// Every lane that is active at the call site takes part; lanes may have left the enclosing loop.
__device__ __forceinline__ uint32_t shard_slot(const ShardQueue &q, uint32_t shard, bool want) {
    const uint64_t bal = __ballot(want);
    if (!bal) return 0xffffffffu;
    const int first = __builtin_ctzll(bal);
    uint32_t base = 0;
    if (lane_id() == first) base = atomicAdd(q.counts + shard * kQPad, (uint32_t)__popcll(bal));
    base = (uint32_t)__shfl((int)base, first, 64);
    return want ? shard * q.cap + base + (uint32_t)__popcll(bal & lanemask_lt()) : 0xffffffffu;
}

// The same for several rounds of items at once (kRounds ballots of one wavefront): ONE atomic for all of
// them; slot[r] as shard_slot would give for round r.  All lanes of the wavefront must be active.
template <int kRounds>
__device__ __forceinline__ void shard_slots(const ShardQueue &q, uint32_t shard, const bool (&want)[kRounds],
                                            uint32_t (&slot)[kRounds]) {
    uint64_t bal[kRounds];
    uint32_t total = 0;
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        bal[r] = __ballot(want[r]);
        total += (uint32_t)__popcll(bal[r]);
    }
    uint32_t base = 0;
    if (total) {
        if (lane_id() == 0) base = atomicAdd(q.counts + shard * kQPad, total);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    }
    base += shard * q.cap;
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        slot[r] = want[r] ? base + (uint32_t)__popcll(bal[r] & lanemask_lt()) : 0xffffffffu;
        base += (uint32_t)__popcll(bal[r]);
    }
}

// totals[k] = number of items in queue k (k < nq), for the host
__global__ void shard_totals_kernel(const uint32_t *const *counts, int nq, uint32_t *totals);

}  // namespace nolzss
