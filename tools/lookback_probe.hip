// lookback_probe.hip -- what would a decoupled look-back ("onesweep") cost the bucket-segmented radix passes?
//
// Today a pass is rs_hist_kernel (reads the keys, 0.98 ms at 2^30 pairs) + scan (0.08 ms) + rs_scatter_kernel (4.1 ms).
// A single-read histogram of all four digits up front would remove the four histogram launches IF every tile could
// get the 256 per-bin offsets of its bucket from its predecessors while it runs: tile t publishes its 256 bin counts
// (AGGREGATE), walks back over the status rows of t-1, t-2, .. summing aggregates until it meets a row that already
// holds an inclusive PREFIX, and publishes its own prefix.
//
// This program measures that mechanism alone at the scatter kernel's own tile rate: 2^18 tiles of 256 threads
// (thread = bin), the scatter kernel's LDS footprint (so the same 3 workgroups per CU are resident), the XCD chunk
// mapping of radix_sort.hip, and in place of the loads / ranking / writes of a tile two timed waits that make the
// launch WITHOUT look-back as long as the real kernel.  Reported: launch time without and with the look-back, rows
// read per tile and bin, the longest walk.  Every spin is bounded: a walk that waits too long sets an error flag and
// the tile leaves (no launch can hang).
//   hipcc --offload-arch=gfx950 -O3 tools/lookback_probe.hip -o gpurun_out/lookback_probe && gpurun_out/lookback_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kBins = 256;
constexpr uint32_t kAgg = 1u << 30, kPre = 2u << 30, kVal = (1u << 30) - 1u;

__device__ __forceinline__ uint32_t chunk_tile(uint32_t b, uint32_t num_tiles, uint32_t chunk) {
    if (chunk == 0) return b < num_tiles ? b : 0xffffffffu;  // tiles in block order
    const uint32_t x = b % 8, k = b / 8;
    const uint32_t c = (k / chunk) * 8 + x;
    const uint32_t t = c * chunk + k % chunk;
    return t < num_tiles ? t : 0xffffffffu;
}

__device__ __forceinline__ void wait_ticks(uint64_t ticks) {  // 100 MHz counter
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

struct Stats {
    unsigned long long rows;     // status rows read by bin 0 of every tile
    unsigned long long spins;    // reads that found a row not yet published (bin 0)
    unsigned int longest;        // longest walk
    unsigned int errors;         // walks given up
    unsigned long long wait_ticks;  // 100 MHz ticks bin 0 spent in the walk
};

template <bool kLook, int kAhead>
__global__ __launch_bounds__(256) void probe_kernel(uint32_t *__restrict__ status, uint32_t num_tiles,
                                                    uint32_t tiles_per_bucket, uint32_t chunk, uint32_t *ticket,
                                                    uint32_t pre_ticks, uint32_t post_ticks, Stats *stats) {
    extern __shared__ uint32_t s_pad[];
    __shared__ uint32_t s_tile;
    uint32_t b = blockIdx.x;
    if (ticket) {  // tiles in the order the workgroups START
        if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
        __syncthreads();
        b = s_tile;
    }
    const uint32_t tile = chunk_tile(b, num_tiles, chunk);
    if (tile == 0xffffffffu) return;
    s_pad[threadIdx.x] = tile;
    wait_ticks(pre_ticks);  // loads + ranking
    const uint32_t bin = threadIdx.x;
    const uint32_t count = 16u;
    if (kLook) {
        const uint32_t first = tile - tile % tiles_per_bucket;
        uint32_t *row = status + (size_t)tile * kBins;
        if (tile == first) {
            __hip_atomic_store(row + bin, kPre | count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_store(row + bin, kAgg | count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t sum = 0, walked = 0, spins = 0;
            bool bad = false;
            const uint64_t t0 = wall_clock64();
            uint32_t t = tile;
            bool done = false;
            while (!done && !bad) {
                // kAhead rows in flight at a time (a walk that takes one round trip per row is hopeless from the start)
                uint32_t v[kAhead];
#pragma unroll
                for (int k = 0; k < kAhead; ++k) {
                    const uint32_t tk = t - 1 - (uint32_t)k;
                    v[k] = (t >= first + 1 + (uint32_t)k)
                               ? __hip_atomic_load(status + (size_t)tk * kBins + bin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                               : kPre;  // in front of the bucket: a prefix of 0
                }
#pragma unroll
                for (int k = 0; k < kAhead; ++k) {
                    if (done || bad) break;
                    uint32_t x = v[k];
                    uint32_t tries = 0;
                    while (x == 0) {  // not published yet
                        __builtin_amdgcn_s_sleep(2);
                        x = __hip_atomic_load(status + (size_t)(t - 1 - (uint32_t)k) * kBins + bin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ++spins;
                        if (++tries > (1u << 15)) { bad = true; break; }
                    }
                    if (bad) break;
                    sum += x & kVal;
                    ++walked;
                    if (x & kPre) done = true;
                }
                t -= kAhead;
            }
            __hip_atomic_store(row + bin, kPre | ((sum + count) & kVal), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (bin == 0) {
                atomicAdd(&stats->rows, (unsigned long long)walked);
                atomicAdd(&stats->spins, (unsigned long long)spins);
                atomicMax(&stats->longest, walked);
                atomicAdd(&stats->wait_ticks, (unsigned long long)(wall_clock64() - t0));
                if (bad) atomicAdd(&stats->errors, 1u);
            }
            // check: every tile of a bucket holds 16 keys per bin
            if (!bad && sum != 16u * (tile - first)) atomicAdd(&stats->errors, 1u);
        }
    }
    __syncthreads();
    wait_ticks(post_ticks);  // scattered writes
    if (s_pad[threadIdx.x] == 0xfffffffeu) status[0] = 1;
}

template <bool kLook, int kAhead>
float launch(uint32_t *status, uint32_t num_tiles, uint32_t per_bucket, uint32_t chunk, uint32_t *ticket, uint32_t pre,
             uint32_t post, Stats *d_stats, size_t lds) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipMemset(status, 0, (size_t)num_tiles * kBins * 4));
    CK(hipMemset(d_stats, 0, sizeof(Stats)));
    if (ticket) CK(hipMemset(ticket, 0, 4));
    uint32_t grid = num_tiles;
    if (chunk) grid = ((num_tiles + chunk - 1) / chunk + 7) / 8 * 8 * chunk;
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    probe_kernel<kLook, kAhead><<<grid, 256, lds>>>(status, num_tiles, per_bucket, chunk, ticket, pre, post, d_stats);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

int main(int argc, char **argv) {
    const uint32_t num_tiles = 1u << 18;    // 2^30 pairs in tiles of 4096
    const uint32_t per_bucket = 1u << 10;   // 256 buckets of 2^22 pairs
    const size_t lds = 48 * 1024;           // three workgroups per CU, as the registers of the scatter kernel allow
    uint32_t *status, *ticket;
    Stats *d_stats, h;
    CK(hipMalloc(&status, (size_t)num_tiles * kBins * 4));
    CK(hipMalloc(&ticket, 4));
    CK(hipMalloc(&d_stats, sizeof(Stats)));
    CK(hipFuncSetAttribute((const void *)probe_kernel<true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)probe_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)probe_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // waits that make the launch without look-back as long as rs_scatter_kernel<u32, u32> at 2^30 pairs (4.1 ms):
    // 768 resident workgroups, 341 tiles each, 12 us per tile
    const uint32_t pre = argc > 1 ? (uint32_t)atoi(argv[1]) : 500, post = argc > 2 ? (uint32_t)atoi(argv[2]) : 600;
    printf("2^18 tiles x 256 bins, buckets of %u tiles, waits %u + %u ticks of 10 ns per tile\n", per_bucket, pre, post);
    launch<false, 1>(status, num_tiles, per_bucket, 64, nullptr, pre, post, d_stats, lds);  // warm-up
    const float base = launch<false, 1>(status, num_tiles, per_bucket, 64, nullptr, pre, post, d_stats, lds);
    printf("%-44s %7.3f ms\n", "no look-back (chunks of 64 tiles per XCD)", base);
    struct Case { const char *what; uint32_t chunk; bool ticket; int ahead; };
    const Case cases[] = {
        {"look-back, chunks of 64, 8 rows in flight", 64, false, 8}, {"look-back, chunks of 8, 8 rows in flight", 8, false, 8},
        {"look-back, block order, 8 rows in flight", 0, false, 8},   {"look-back, ticket order, 8 rows in flight", 0, true, 8},
        {"look-back, ticket order, 1 row in flight", 0, true, 1},    {"look-back, chunks of 8, 1 row in flight", 8, false, 1},
    };
    for (const Case &c : cases) {
        float ms = c.ahead == 8 ? launch<true, 8>(status, num_tiles, per_bucket, c.chunk, c.ticket ? ticket : nullptr, pre, post, d_stats, lds)
                                : launch<true, 1>(status, num_tiles, per_bucket, c.chunk, c.ticket ? ticket : nullptr, pre, post, d_stats, lds);
        CK(hipMemcpy(&h, d_stats, sizeof(h), hipMemcpyDeviceToHost));
        printf("%-44s %7.3f ms (+%.3f)  rows per tile and bin %.1f  longest walk %u  reads of an unpublished row %.1f per tile  "
               "in the walk %.2f us per tile  errors %u\n",
               c.what, ms, ms - base, (double)h.rows / num_tiles, h.longest, (double)h.spins / num_tiles,
               (double)h.wait_ticks / num_tiles / 100.0, h.errors);
    }
    printf("status table: %.0f MB written twice and read (rows per tile) times per pass, beside 17.2 GB of pairs\n",
           (double)num_tiles * kBins * 4 / 1e6);
    return 0;
}
