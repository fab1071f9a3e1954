// scan.hip -- device-wide prefix scans (reduce-then-scan).
//
// HBM-bound streaming kernels: each 256-thread workgroup owns a 4096-item tile, read as four
// 1024-item rows (4 consecutive items per lane, so a wave instruction covers 1 KiB contiguous).
// Level k+1 scans the per-tile sums of level k; three levels cover 2^36 items.
#include "scan.hpp"

namespace nolzss {
namespace {

constexpr int kThreads = 256;
constexpr int kItemsPerThread = 4;
constexpr int kRows = 4;
constexpr int kTile = kThreads * kItemsPerThread * kRows;  // 4096

template <typename Op>
__global__ __launch_bounds__(kThreads) void reduce_tiles_kernel(const uint32_t *__restrict__ in,
                                                                uint32_t *__restrict__ sums,
                                                                size_t n) {
    __shared__ uint32_t lds[kThreads / 64];
    Op op;
    const size_t base = (size_t)blockIdx.x * kTile;
    uint32_t acc = Op::identity();
    // every load first (elements past the end load the last one again and are not counted): with the
    // accumulation inside the guarded load the compiler waited for each load on its own
    uint32_t v[kRows][kItemsPerThread];
#pragma unroll
    for (int row = 0; row < kRows; ++row) {
        const size_t idx = base + ((size_t)row * kThreads + threadIdx.x) * kItemsPerThread;
#pragma unroll
        for (int e = 0; e < kItemsPerThread; ++e) v[row][e] = in[idx + e < n ? idx + e : n - 1];
    }
#pragma unroll
    for (int row = 0; row < kRows; ++row) {
        const size_t idx = base + ((size_t)row * kThreads + threadIdx.x) * kItemsPerThread;
#pragma unroll
        for (int e = 0; e < kItemsPerThread; ++e)
            if (idx + e < n) acc = op(acc, v[row][e]);
    }
    acc = wave_reduce(acc, op);
    if (lane_id() == 0) lds[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = Op::identity();
        for (int k = 0; k < kThreads / 64; ++k) t = op(t, lds[k]);
        sums[blockIdx.x] = t;
    }
}

// carry: exclusive scan of the tile sums (nullptr => single tile, carry = identity)
template <typename Op, bool kInclusive>
__global__ __launch_bounds__(kThreads) void scan_tiles_kernel(const uint32_t *in, uint32_t *out,
                                                              size_t n, const uint32_t *carry,
                                                              uint32_t *d_total) {
    __shared__ uint32_t lds[kThreads / 64];
    Op op;
    const size_t base = (size_t)blockIdx.x * kTile;
    uint32_t running = carry ? carry[blockIdx.x] : Op::identity();
#pragma unroll 1
    for (int row = 0; row < kRows; ++row) {
        size_t idx = base + ((size_t)row * kThreads + threadIdx.x) * kItemsPerThread;
        uint32_t v[kItemsPerThread];
        uint32_t local = Op::identity();
#pragma unroll
        for (int e = 0; e < kItemsPerThread; ++e) {
            v[e] = (idx + e < n) ? in[idx + e] : Op::identity();
            local = op(local, v[e]);
        }
        uint32_t row_total;
        uint32_t pre = op(running, block_scan_exclusive<kThreads / 64>(local, op, lds, row_total));
#pragma unroll
        for (int e = 0; e < kItemsPerThread; ++e) {
            uint32_t inc = op(pre, v[e]);
            if (idx + e < n) out[idx + e] = kInclusive ? inc : pre;
            pre = inc;
        }
        running = op(running, row_total);
    }
    if (d_total && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *d_total = running;
}

template <typename Op, bool kInclusive>
void scan_impl(const uint32_t *in, uint32_t *out, size_t n, uint32_t *d_total, Arena &arena,
               hipStream_t stream) {
    if (n == 0) {
        if (d_total) HIP_CHECK(hipMemsetAsync(d_total, 0, sizeof(uint32_t), stream));
        return;
    }
    const size_t nb = div_up(n, kTile);
    if (nb == 1) {
        scan_tiles_kernel<Op, kInclusive><<<1, kThreads, 0, stream>>>(in, out, n, nullptr, d_total);
        KERNEL_CHECK();
        return;
    }
    const size_t m = arena.mark();
    uint32_t *sums = arena.alloc<uint32_t>(nb);
    reduce_tiles_kernel<Op><<<(unsigned)nb, kThreads, 0, stream>>>(in, sums, n);
    KERNEL_CHECK();
    // exclusive scan of the tile sums; its grand total is the grand total of the input
    scan_impl<Op, false>(sums, sums, nb, d_total, arena, stream);
    scan_tiles_kernel<Op, kInclusive><<<(unsigned)nb, kThreads, 0, stream>>>(in, out, n, sums, nullptr);
    KERNEL_CHECK();
    arena.rewind(m);
}

}  // namespace

void scan_exclusive_add_u32(const uint32_t *in, uint32_t *out, size_t n, uint32_t *d_total,
                            Arena &arena, hipStream_t stream) {
    scan_impl<OpAdd<uint32_t>, false>(in, out, n, d_total, arena, stream);
}

void scan_inclusive_max_u32(const uint32_t *in, uint32_t *out, size_t n, Arena &arena,
                            hipStream_t stream) {
    scan_impl<OpMax<uint32_t>, true>(in, out, n, nullptr, arena, stream);
}

}  // namespace nolzss
