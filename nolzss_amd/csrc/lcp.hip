// lcp.hip -- LCP array from the suffix array by direct comparison of bit-packed suffixes.
//
// Replaces the LCP / tree-depth information of the reference's compressed suffix tree
// (cst.depth, cst.lca: /root/reference/src/cpp/factorizer_helpers.hpp:20-24).
// One thread per rank r: compares suffixes sa[r-1] and sa[r] 64 bits (32 DNA bases) at a time.
// Reads of sa are coalesced; the two text probes per rank hit the packed text (n/4 bytes for
// DNA), which is Infinity-Cache resident up to 2^30 bases.
#include "pipeline.hpp"

namespace nolzss {
namespace {

constexpr int kThreads = 256;

template <int BITS>
__global__ __launch_bounds__(kThreads) void lcp_kernel(const uint64_t *__restrict__ words, uint32_t n,
                                                       const uint32_t *__restrict__ sa,
                                                       uint32_t *__restrict__ lcp) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r <= n; r += stride) {
        uint32_t v = 0;
        if (r > 0 && r < n) v = suffix_lcp<BITS>(words, n, sa[r - 1], sa[r], 0);
        lcp[r] = v;
    }
}

}  // namespace

void build_lcp(Context &ctx, const PackedText &text, const uint32_t *sa, uint32_t *lcp) {
    ProfScope ps(ctx.profiler(), "lcp", ctx.stream);
    size_t g = div_up((size_t)text.n + 1, kThreads);
    if (g > 256u * 32u) g = 256u * 32u;
    switch (text.bits) {
    case 2: lcp_kernel<2><<<(unsigned)g, kThreads, 0, ctx.stream>>>(text.words, text.n, sa, lcp); break;
    case 4: lcp_kernel<4><<<(unsigned)g, kThreads, 0, ctx.stream>>>(text.words, text.n, sa, lcp); break;
    default: lcp_kernel<8><<<(unsigned)g, kThreads, 0, ctx.stream>>>(text.words, text.n, sa, lcp); break;
    }
    KERNEL_CHECK();
}

}  // namespace nolzss
