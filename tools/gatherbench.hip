// gatherbench.hip -- how many random text-window fetches per second does an MI355X sustain?
// The direct-comparison round of the suffix sort (group_refine_kernel) reads one unaligned window
// of the packed text per tied suffix per step; this measures the ceiling for that access pattern:
// every thread reads W consecutive 64-bit words at a pseudo-random word offset of a table.
//   hipcc --offload-arch=gfx950 -O3 tools/gatherbench.hip -o gpurun_out/gatherbench && gpurun_out/gatherbench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

template <int W, int ROUNDS>
__global__ __launch_bounds__(256) void gather_kernel(const uint64_t *__restrict__ table, uint64_t mask, uint64_t n,
                                                     uint64_t *__restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t at = mix(i) & mask;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {  // dependent steps, like successive comparison rounds
            uint64_t v = 0;
#pragma unroll
            for (int k = 0; k < W; ++k) v ^= table[at + k];
            acc += v;
            at = (at + W + (v & 1)) & mask;  // next window right behind this one (same or next line)
        }
    }
    if (acc == 0x1234567) out[0] = acc;
}

template <int W, int ROUNDS>
void run(const uint64_t *table, uint64_t words, uint64_t n, uint64_t *out, const char *what) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const uint64_t mask = words - 1;
    gather_kernel<W, ROUNDS><<<256 * 32, 256>>>(table, mask, n, out);
    CK(hipEventRecord(a));
    gather_kernel<W, ROUNDS><<<256 * 32, 256>>>(table, mask, n, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("%-28s table %4llu MiB  W=%d words  rounds=%d : %7.2f ms  %6.2f G windows/s  %7.1f GB/s useful\n", what,
           (unsigned long long)(words * 8 >> 20), W, ROUNDS, ms, n * ROUNDS / ms / 1e6, n * ROUNDS * W * 8 / ms / 1e6);
}

int main() {
    const uint64_t n = 1ull << 29;
    uint64_t *table, *out;
    const uint64_t max_words = 1ull << 27;  // 1 GiB
    CK(hipMalloc(&table, (max_words + 64) * 8));
    CK(hipMalloc(&out, 8));
    CK(hipMemset(table, 0x5a, (max_words + 64) * 8));
    for (uint64_t words : {1ull << 23, 1ull << 25, 1ull << 27}) {  // 64 MiB, 256 MiB (2^30 bases), 1 GiB
        run<1, 1>(table, words, n, out, "1 word");
        run<3, 1>(table, words, n, out, "128-bit unaligned (3 words)");
        run<5, 1>(table, words, n, out, "256-bit unaligned (5 words)");
        run<9, 1>(table, words, n, out, "512-bit unaligned (9 words)");
        run<3, 3>(table, words, n / 2, out, "3 dependent 128-bit steps");
    }
    return 0;
}
