// pyramid.hip -- builds the upper levels of a 16-ary min / max pyramid (see pyramid.hpp).
// Streaming kernel: each thread reduces one aligned group of 16 entries (64 bytes).
#include "pyramid.hpp"

namespace nolzss {
namespace {

constexpr int kThreads = 256;

// flag_min / flag (optional, first level of the LCP pyramid): *flag is set if any base entry is
// >= flag_min, i.e. still holds a "pending" code of the suffix-array construction
template <bool kMax>
__global__ __launch_bounds__(kThreads) void pyramid_level_kernel(const uint32_t *__restrict__ in, uint32_t len_in,
                                                                 uint32_t *__restrict__ out, uint32_t len_out,
                                                                 uint32_t flag_min, uint32_t *__restrict__ flag) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < len_out; j += stride) {
        const size_t base = j << kPyrShift;
        uint32_t res = kMax ? 0u : 0xffffffffu, top = 0u;
        if (base + kPyrFan <= len_in) {
            const uint4 *v = reinterpret_cast<const uint4 *>(in + base);  // level arrays are 256-B aligned
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint4 x = v[k];
                res = pyr_op<kMax>(res, pyr_op<kMax>(pyr_op<kMax>(x.x, x.y), pyr_op<kMax>(x.z, x.w)));
                if (!kMax) top = pyr_op<true>(top, pyr_op<true>(pyr_op<true>(x.x, x.y), pyr_op<true>(x.z, x.w)));
            }
        } else {
            for (size_t q = base; q < len_in; ++q) {
                res = pyr_op<kMax>(res, in[q]);
                top = pyr_op<true>(top, in[q]);
            }
        }
        out[j] = res;
        if (!kMax && flag && top >= flag_min) atomicOr(flag, 1u);
    }
}

}  // namespace

Pyramid alloc_pyramid(const uint32_t *base, uint32_t len, Arena &arena) {
    Pyramid P{};
    P.lvl[0] = base;
    P.len[0] = len;
    P.nlev = 1;
    if (((uintptr_t)base & 15) != 0) throw HipError("pyramid: base array must be 16-byte aligned");
    while (P.len[P.nlev - 1] > 1 && P.nlev < kPyrMaxLevels) {
        const uint32_t len_out = (P.len[P.nlev - 1] + kPyrFan - 1) >> kPyrShift;
        P.lvl[P.nlev] = arena.alloc<uint32_t>(len_out);
        P.len[P.nlev] = len_out;
        ++P.nlev;
    }
    return P;
}

void fill_pyramid(const Pyramid &P, int first_level, bool is_max, hipStream_t stream, uint32_t flag_min, uint32_t *flag) {
    for (int lev = first_level < 1 ? 1 : first_level; lev < P.nlev; ++lev) {
        const uint32_t len_in = P.len[lev - 1], len_out = P.len[lev];
        uint32_t *out = const_cast<uint32_t *>(P.lvl[lev]);
        size_t g = div_up(len_out, kThreads);
        if (g > 8192) g = 8192;
        if (is_max)
            pyramid_level_kernel<true><<<(unsigned)g, kThreads, 0, stream>>>(P.lvl[lev - 1], len_in, out, len_out, 0u, nullptr);
        else
            pyramid_level_kernel<false><<<(unsigned)g, kThreads, 0, stream>>>(P.lvl[lev - 1], len_in, out, len_out, flag_min,
                                                                              lev == 1 ? flag : nullptr);
        KERNEL_CHECK();
    }
}

void fill_pyramid_tail(const Pyramid &P, uint32_t first_entry, bool is_max, hipStream_t stream) {
    if (P.nlev < 2 || first_entry >= P.len[1]) return;
    const size_t skip = (size_t)first_entry << kPyrShift;  // base entries in front of the tail (a multiple of 16: aligned)
    const uint32_t *in = P.lvl[0] + skip;
    uint32_t *out = const_cast<uint32_t *>(P.lvl[1]) + first_entry;
    const uint32_t len_in = (uint32_t)(P.len[0] - skip), len_out = P.len[1] - first_entry;
    if (is_max)
        pyramid_level_kernel<true><<<1, kThreads, 0, stream>>>(in, len_in, out, len_out, 0u, nullptr);
    else
        pyramid_level_kernel<false><<<1, kThreads, 0, stream>>>(in, len_in, out, len_out, 0u, nullptr);
    KERNEL_CHECK();
}

Pyramid build_pyramid(const uint32_t *base, uint32_t len, bool is_max, Arena &arena, hipStream_t stream,
                      uint32_t flag_min, uint32_t *flag) {
    const Pyramid P = alloc_pyramid(base, len, arena);
    fill_pyramid(P, 1, is_max, stream, flag_min, flag);
    return P;
}

}  // namespace nolzss
