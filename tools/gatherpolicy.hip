// gatherpolicy.hip -- does a cache policy change what a random window fetch costs?  (round 4)
// group_refine_kernel reads 0.85 G random 40-byte windows of the packed text per step and the counters show
// 1.08 G line fills = 139 GB (profiles/r04_pmc_step.json): it is bound by 128-byte line fills at 6.1 TB/s.  This
// probe repeats tools/gatherbench.hip's 5-word window gather from a 256 MiB table with plain loads, non-temporal
// loads (__builtin_nontemporal_load) and buffer loads with the sc0 / sc1 / nt bits, so that FETCH_SIZE per window
// (rocprofv3 --pmc FETCH_SIZE) and the rate can be compared.
//   hipcc --offload-arch=gfx950 -O3 tools/gatherpolicy.hip -o gpurun_out/gatherpolicy && gpurun_out/gatherpolicy
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void gather_kernel(const uint64_t *__restrict__ table, uint64_t mask, uint64_t n,
                                                     uint64_t *__restrict__ out, uint64_t table_bytes) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t acc = 0;
    // buffer resource over the whole table (MODE >= 2)
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t *>(table), 0, 0x7fffffff, 0x00020000);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t at = mix(i) & mask;
        uint64_t v = 0;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            if (MODE == 0) v ^= table[at + k];
            else if (MODE == 1) v ^= __builtin_nontemporal_load(table + at + k);
            else {
                constexpr int aux = MODE == 2 ? 1 : (MODE == 3 ? 2 : (MODE == 4 ? 16 : 17));  // sc0, nt, sc1, sc0+sc1
                const v2i x = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)((at + k) * 8), 0, aux);
                v ^= ((uint64_t)(uint32_t)x[1] << 32) | (uint32_t)x[0];
            }
        }
        acc += v;
    }
    if (acc == 0x1234567) out[0] = acc;
}

template <int MODE> void run(const uint64_t *table, uint64_t words, uint64_t n, uint64_t *out, const char *what) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    gather_kernel<MODE><<<256 * 32, 256>>>(table, words - 1, n, out, words * 8);
    CK(hipEventRecord(a));
    gather_kernel<MODE><<<256 * 32, 256>>>(table, words - 1, n, out, words * 8);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("%-34s table %4llu MiB : %7.2f ms  %6.2f G windows/s\n", what, (unsigned long long)(words * 8 >> 20), ms, n / ms / 1e6);
}

int main() {
    const uint64_t n = 1ull << 28;
    uint64_t *table, *out;
    const uint64_t words = 1ull << 25;  // 256 MiB
    CK(hipMalloc(&table, (words + 64) * 8));
    CK(hipMalloc(&out, 8));
    CK(hipMemset(table, 0x5a, (words + 64) * 8));
    run<0>(table, words, n, out, "plain global loads");
    run<1>(table, words, n, out, "non-temporal loads");
    run<2>(table, words, n, out, "buffer loads sc0");
    run<3>(table, words, n, out, "buffer loads nt");
    run<4>(table, words, n, out, "buffer loads sc1");
    run<5>(table, words, n, out, "buffer loads sc0 sc1");
    return 0;
}
