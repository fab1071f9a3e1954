// radix_sort.hip -- stable LSD radix sort of (u64 key, u32 value) pairs for gfx950.
//
// One pass = three launches:
//   rs_hist_kernel     per 4096-key tile, a 256-bin digit histogram (LDS atomics) written
//                      bin-major, so one linear exclusive scan yields every (bin, tile) base;
//   scan               exclusive add-scan of the 256 x num_tiles table (scan.hip);
//   rs_scatter_kernel  re-reads the tile, ranks every key inside its wavefront with
//                      ballot-based peer matching (64-lane match-any over the 8 digit bits),
//                      sorts the tile by digit through LDS, and writes each bin's run with
//                      consecutive lanes on consecutive addresses.
// HBM-bound: algorithmic traffic per pass = n * (8 + 4) bytes read + the same written by the
// scatter kernel, plus n * 8 bytes read by the histogram kernel.
#include "radix_sort.hpp"

#include "scan.hpp"

namespace nolzss {
namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kKeysPerThread = 16;
constexpr int kTile = kThreads * kKeysPerThread;  // 4096
constexpr int kBins = 1 << kRadixBits;
constexpr int kWaveSpan = kTile / kWaves;  // 1024 keys per wavefront, 16 rows of 64

static_assert(kBins == kThreads, "one thread per bin in the offset phase");

__global__ __launch_bounds__(kThreads) void rs_hist_kernel(const uint64_t *__restrict__ keys,
                                                           size_t n, int shift,
                                                           uint32_t *__restrict__ tile_hist,
                                                           uint32_t num_tiles) {
    __shared__ uint32_t hist[kBins];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * kTile;
#pragma unroll
    for (int j = 0; j < kKeysPerThread; ++j) {
        size_t idx = base + (size_t)j * kThreads + threadIdx.x;
        if (idx < n) atomicAdd(&hist[(uint32_t)(keys[idx] >> shift) & (kBins - 1)], 1u);
    }
    __syncthreads();
    tile_hist[(size_t)threadIdx.x * num_tiles + blockIdx.x] = hist[threadIdx.x];
}

__global__ __launch_bounds__(kThreads) void rs_scatter_kernel(
    const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
    uint64_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, size_t n, int shift,
    const uint32_t *__restrict__ tile_base, uint32_t num_tiles) {
    __shared__ uint64_t s_keys[kTile];
    __shared__ uint32_t s_vals[kTile];
    __shared__ uint32_t s_whist[kWaves * kBins];
    __shared__ uint32_t s_glob[kBins];
    __shared__ uint32_t s_scan[kWaves];

    const int tid = threadIdx.x;
    const int w = tid >> 6;
    const int lane = tid & 63;
    volatile uint32_t *whist = s_whist + w * kBins;

#pragma unroll
    for (int k = 0; k < kWaves; ++k) s_whist[k * kBins + tid] = 0;
    __syncthreads();

    const size_t base = (size_t)blockIdx.x * kTile;
    uint64_t key[kKeysPerThread];
    uint32_t val[kKeysPerThread];
    uint32_t lrank[kKeysPerThread];

    // rank inside the wavefront: rows of 64 keys in input order (keeps the sort stable)
#pragma unroll
    for (int row = 0; row < kKeysPerThread; ++row) {
        const size_t idx = base + (size_t)w * kWaveSpan + (size_t)row * 64 + lane;
        const bool valid = idx < n;
        key[row] = valid ? keys_in[idx] : 0;
        val[row] = valid ? vals_in[idx] : 0;
        const uint32_t d = (uint32_t)(key[row] >> shift) & (kBins - 1);
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < kRadixBits; ++b) {
            const bool bit = (d >> b) & 1;
            const uint64_t bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        if (valid) {
            const uint64_t below = peers & lanemask_lt();
            const uint32_t old = whist[d];
            lrank[row] = old + (uint32_t)__popcll(below);
            if (below == 0) whist[d] = old + (uint32_t)__popcll(peers);
        }
    }
    __syncthreads();

    // thread = bin: turn per-wave counts into tile-local start positions
    {
        const int d = tid;
        uint32_t c[kWaves], total = 0;
#pragma unroll
        for (int k = 0; k < kWaves; ++k) {
            c[k] = s_whist[k * kBins + d];
            total += c[k];
        }
        uint32_t tile_total;
        const uint32_t bin_start =
            block_scan_exclusive<kWaves>(total, OpAdd<uint32_t>(), s_scan, tile_total);
        uint32_t run = bin_start;
#pragma unroll
        for (int k = 0; k < kWaves; ++k) {
            s_whist[k * kBins + d] = run;
            run += c[k];
        }
        s_glob[d] = tile_base[(size_t)d * num_tiles + blockIdx.x] - bin_start;
    }
    __syncthreads();

#pragma unroll
    for (int row = 0; row < kKeysPerThread; ++row) {
        const size_t idx = base + (size_t)w * kWaveSpan + (size_t)row * 64 + lane;
        if (idx < n) {
            const uint32_t d = (uint32_t)(key[row] >> shift) & (kBins - 1);
            const uint32_t pos = s_whist[w * kBins + d] + lrank[row];
            s_keys[pos] = key[row];
            s_vals[pos] = val[row];
        }
    }
    __syncthreads();

    const uint32_t count = (uint32_t)((n - base < (size_t)kTile) ? (n - base) : (size_t)kTile);
#pragma unroll
    for (int j = 0; j < kKeysPerThread; ++j) {
        const uint32_t p = (uint32_t)j * kThreads + tid;
        if (p < count) {
            const uint64_t k = s_keys[p];
            const uint32_t d = (uint32_t)(k >> shift) & (kBins - 1);
            const uint32_t g = s_glob[d] + p;
            keys_out[g] = k;
            vals_out[g] = s_vals[p];
        }
    }
}

}  // namespace

int radix_sort_pairs(uint64_t *keys[2], uint32_t *vals[2], size_t n, const int *shifts,
                     int npasses, Arena &arena, hipStream_t stream, Profiler *prof) {
    if (n == 0 || npasses == 0) return 0;
    const size_t m = arena.mark();
    const uint32_t num_tiles = (uint32_t)div_up(n, kTile);
    uint32_t *hist = arena.alloc<uint32_t>((size_t)kBins * num_tiles);
    int cur = 0;
    for (int p = 0; p < npasses; ++p) {
        {
            ProfScope ps(prof, "rs_hist", stream, 8.0 * (double)n);
            rs_hist_kernel<<<num_tiles, kThreads, 0, stream>>>(keys[cur], n, shifts[p], hist, num_tiles);
            KERNEL_CHECK();
        }
        {
            ProfScope ps(prof, "rs_scan", stream, 8.0 * (double)kBins * num_tiles);
            scan_exclusive_add_u32(hist, hist, (size_t)kBins * num_tiles, nullptr, arena, stream);
        }
        {
            // algorithmic bytes of one scatter launch: every (key, value) pair read once and
            // written once = 2 * (8 + 4) bytes per pair
            ProfScope ps(prof, "rs_scatter", stream, 24.0 * (double)n);
            rs_scatter_kernel<<<num_tiles, kThreads, 0, stream>>>(keys[cur], vals[cur], keys[cur ^ 1],
                                                                  vals[cur ^ 1], n, shifts[p], hist,
                                                                  num_tiles);
            KERNEL_CHECK();
        }
        cur ^= 1;
    }
    arena.rewind(m);
    return cur;
}

}  // namespace nolzss
