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

#include <cstdlib>

namespace nolzss {
namespace {

constexpr int kKeysPerThread = 16;
constexpr int kBins = 1 << kRadixBits;
constexpr int kWaveSpan = 64 * kKeysPerThread;  // 1024 keys per wavefront, 16 rows of 64

// Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 shares an XCD).  Give each
// XCD a contiguous range of tiles so that the bin runs of neighbouring tiles -- which are
// adjacent in the output -- meet in the same L2 and partial cache lines merge there.
__device__ __forceinline__ uint32_t xcd_tile(uint32_t b, uint32_t num_tiles, bool remap) {
    if (!remap) return b;
    const uint32_t per = num_tiles / 8, rem = num_tiles % 8;
    const uint32_t x = b % 8, k = b / 8;
    // XCD x owns tiles [x*per + min(x, rem), ...) with one extra tile for x < rem
    const uint32_t start = x * per + (x < rem ? x : rem);
    const uint32_t cnt = per + (x < rem ? 1u : 0u);
    return k < cnt ? start + k : 0xffffffffu;
}

template <int kThreads>
__global__ __launch_bounds__(kThreads) void rs_hist_kernel(const uint64_t *__restrict__ keys,
                                                           size_t n, int shift,
                                                           uint32_t *__restrict__ tile_hist,
                                                           uint32_t num_tiles) {
    constexpr int kTile = kThreads * kKeysPerThread;
    __shared__ uint32_t hist[kBins];
    for (int d = threadIdx.x; d < kBins; d += kThreads) hist[d] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * kTile;
#pragma unroll
    for (int j = 0; j < kKeysPerThread; ++j) {
        size_t idx = base + (size_t)j * kThreads + threadIdx.x;
        if (idx < n) atomicAdd(&hist[(uint32_t)(keys[idx] >> shift) & (kBins - 1)], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < kBins; d += kThreads)
        tile_hist[(size_t)d * num_tiles + blockIdx.x] = hist[d];
}

template <int kThreads>
__global__ __launch_bounds__(kThreads) void rs_scatter_kernel(
    const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
    uint64_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, size_t n, int shift,
    const uint32_t *__restrict__ tile_base, uint32_t num_tiles, bool remap) {
    constexpr int kWaves = kThreads / 64;
    constexpr int kTile = kThreads * kKeysPerThread;
    const uint32_t tile = xcd_tile(blockIdx.x, num_tiles, remap);
    if (tile == 0xffffffffu) return;
    // keys and values take turns in one staging buffer (8 B x tile): 37 KiB of LDS per 256-thread
    // workgroup instead of 53 KiB lets a fourth workgroup share the CU
    __shared__ uint64_t s_keys[kTile];
    uint32_t *s_vals = reinterpret_cast<uint32_t *>(s_keys);
    __shared__ uint32_t s_whist[kWaves * kBins];
    __shared__ uint32_t s_glob[kBins];
    __shared__ uint32_t s_scan[kWaves];

    const int tid = threadIdx.x;
    const int w = tid >> 6;
    const int lane = tid & 63;
    volatile uint32_t *whist = s_whist + w * kBins;

    for (int k = tid; k < kWaves * kBins; k += kThreads) s_whist[k] = 0;
    __syncthreads();

    const size_t base = (size_t)tile * kTile;
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
        if (d < kBins) {
#pragma unroll
            for (int k = 0; k < kWaves; ++k) {
                c[k] = s_whist[k * kBins + d];
                total += c[k];
            }
        }
        uint32_t tile_total;
        const uint32_t bin_start =
            block_scan_exclusive<kWaves>(total, OpAdd<uint32_t>(), s_scan, tile_total);
        if (d < kBins) {
            uint32_t run = bin_start;
#pragma unroll
            for (int k = 0; k < kWaves; ++k) {
                s_whist[k * kBins + d] = run;
                run += c[k];
            }
            s_glob[d] = tile_base[(size_t)d * num_tiles + tile] - bin_start;
        }
    }
    __syncthreads();

    // tile-local sorted position of every element (reuses lrank)
#pragma unroll
    for (int row = 0; row < kKeysPerThread; ++row) {
        const uint32_t d = (uint32_t)(key[row] >> shift) & (kBins - 1);
        lrank[row] += s_whist[w * kBins + d];
    }
#pragma unroll
    for (int row = 0; row < kKeysPerThread; ++row) {
        const size_t idx = base + (size_t)w * kWaveSpan + (size_t)row * 64 + lane;
        if (idx < n) s_keys[lrank[row]] = key[row];
    }
    __syncthreads();

    const uint32_t count = (uint32_t)((n - base < (size_t)kTile) ? (n - base) : (size_t)kTile);
    uint32_t gpos[kKeysPerThread];
#pragma unroll
    for (int j = 0; j < kKeysPerThread; ++j) {
        const uint32_t p = (uint32_t)j * kThreads + tid;
        if (p < count) {
            const uint64_t k = s_keys[p];
            const uint32_t d = (uint32_t)(k >> shift) & (kBins - 1);
            gpos[j] = s_glob[d] + p;
            keys_out[gpos[j]] = k;
        }
    }
    __syncthreads();
#pragma unroll
    for (int row = 0; row < kKeysPerThread; ++row) {
        const size_t idx = base + (size_t)w * kWaveSpan + (size_t)row * 64 + lane;
        if (idx < n) s_vals[lrank[row]] = val[row];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kKeysPerThread; ++j) {
        const uint32_t p = (uint32_t)j * kThreads + tid;
        if (p < count) vals_out[gpos[j]] = s_vals[p];
    }
}

template <int kThreads>
int radix_sort_impl(uint64_t *keys[2], uint32_t *vals[2], size_t n, const int *shifts, int npasses,
                    Arena &arena, hipStream_t stream, Profiler *prof, bool remap) {
    constexpr int kTile = kThreads * kKeysPerThread;
    const size_t m = arena.mark();
    const uint32_t num_tiles = (uint32_t)div_up(n, kTile);
    uint32_t *hist = arena.alloc<uint32_t>((size_t)kBins * num_tiles);
    int cur = 0;
    for (int p = 0; p < npasses; ++p) {
        {
            ProfScope ps(prof, "rs_hist", stream, 8.0 * (double)n);
            rs_hist_kernel<kThreads><<<num_tiles, kThreads, 0, stream>>>(keys[cur], n, shifts[p], hist, num_tiles);
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
            const uint32_t grid = remap ? (uint32_t)div_up(num_tiles, 8) * 8 : num_tiles;
            rs_scatter_kernel<kThreads><<<grid, kThreads, 0, stream>>>(keys[cur], vals[cur], keys[cur ^ 1],
                                                                       vals[cur ^ 1], n, shifts[p], hist,
                                                                       num_tiles, remap);
            KERNEL_CHECK();
        }
        cur ^= 1;
    }
    arena.rewind(m);
    return cur;
}

}  // namespace

int radix_sort_pairs(uint64_t *keys[2], uint32_t *vals[2], size_t n, const int *shifts,
                     int npasses, Arena &arena, hipStream_t stream, Profiler *prof) {
    if (n == 0 || npasses == 0) return 0;
    // tuning knob (bit 0: XCD-contiguous tile mapping, bit 1: 512-thread / 8192-key tiles);
    // default = the fastest measured on MI355X: 256 threads, XCD mapping on
    static const int variant = [] {
        const char *e = getenv("NOLZSS_RS_VARIANT");
        return e ? atoi(e) : 1;
    }();
    const bool remap = (variant & 1) != 0;
    if (variant & 2) return radix_sort_impl<512>(keys, vals, n, shifts, npasses, arena, stream, prof, remap);
    return radix_sort_impl<256>(keys, vals, n, shifts, npasses, arena, stream, prof, remap);
}

}  // namespace nolzss
