// text.hpp -- bit-packed text: dense order-preserving symbol codes, BITS per symbol,
// big-endian inside 64-bit words so that lexicographic order == integer order and
// clz(x ^ y) / BITS is the length of a common prefix.
//
// DNA (sigma <= 4) packs at 2 bits/base: a 2^30-base text is 256 MiB and stays resident
// in the 256 MiB Infinity Cache for the random probes of the LCP and key kernels.
#pragma once
#include "common.hpp"

namespace nolzss {

struct PackedText {
    const uint64_t *words = nullptr;  // ceil(n*bits/64) + 4 zero pad words
    uint32_t n = 0;
    int bits = 0;    // 2, 4 or 8
    int sigma = 0;   // distinct byte values present
};

// Symbols per initial sort key and width of the length tag that breaks ties between a
// suffix that ends inside the key window and its zero-padded longer neighbours.
template <int BITS> struct KeyLayout;
template <> struct KeyLayout<2> { static constexpr int kSyms = 29, kTagBits = 6; };
template <> struct KeyLayout<4> { static constexpr int kSyms = 15, kTagBits = 4; };
template <> struct KeyLayout<8> { static constexpr int kSyms = 7, kTagBits = 8; };

// 64 bits of text starting at symbol `pos` (zero padded past the end).
template <int BITS>
__device__ __forceinline__ uint64_t sym_word(const uint64_t *__restrict__ w, uint64_t pos) {
    const uint64_t bit = pos * BITS;
    const uint64_t wi = bit >> 6;
    const int o = (int)(bit & 63);
    const uint64_t a = w[wi];
    const uint64_t b = w[wi + 1];
    return o ? ((a << o) | (b >> (64 - o))) : a;
}

// Common prefix length of suffixes a and b of the packed text, starting the comparison at
// offset h0 (caller guarantees the first h0 symbols match), capped at n - max(a, b).
template <int BITS>
__device__ __forceinline__ uint32_t suffix_lcp(const uint64_t *__restrict__ w, uint32_t n,
                                               uint32_t a, uint32_t b, uint32_t h0) {
    constexpr uint32_t kPerWord = 64 / BITS;
    const uint32_t limit = n - (a > b ? a : b);
    uint32_t h = h0;
    while (h < limit) {
        const uint64_t x = sym_word<BITS>(w, (uint64_t)a + h);
        const uint64_t y = sym_word<BITS>(w, (uint64_t)b + h);
        if (x != y) {
            h += (uint32_t)__clzll((long long)(x ^ y)) / BITS;
            break;
        }
        h += kPerWord;
    }
    return h < limit ? h : limit;
}

// Three-way comparison of suffixes a and b that are known to agree on their first h0 symbols,
// looking at most `cap` symbols deep.  Returns -1 / +1 if suffix a is smaller / greater, 0 if
// the first `cap` symbols are equal (undecided); lcp receives the common prefix length
// (>= cap when undecided).  A suffix that ends first is the smaller one, as with the
// reference's appended terminator.
template <int BITS>
__device__ __forceinline__ int suffix_compare(const uint64_t *__restrict__ w, uint32_t n, uint32_t a, uint32_t b,
                                              uint32_t h0, uint32_t cap, uint32_t &lcp) {
    constexpr uint32_t kPerWord = 64 / BITS;
    const uint32_t limit = n - (a > b ? a : b);  // length of the shorter suffix
    const uint32_t stop = limit < cap ? limit : cap;
    uint32_t h = h0;
    while (h < stop) {
        const uint64_t x = sym_word<BITS>(w, (uint64_t)a + h);
        const uint64_t y = sym_word<BITS>(w, (uint64_t)b + h);
        if (x != y) {
            const uint32_t d = h + (uint32_t)__clzll((long long)(x ^ y)) / BITS;
            if (d < stop) {
                lcp = d;
                return x < y ? -1 : 1;
            }
            break;  // first difference lies past the shorter suffix or past the cap
        }
        h += kPerWord;
    }
    if (limit <= cap) {  // the shorter suffix is a prefix of the other one
        lcp = limit;
        return a > b ? -1 : 1;
    }
    lcp = cap;
    return 0;
}

}  // namespace nolzss
