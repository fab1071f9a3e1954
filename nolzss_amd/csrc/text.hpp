// text.hpp -- bit-packed text: dense order-preserving symbol codes, BITS per symbol,
// big-endian inside 64-bit words so that lexicographic order == integer order and
// clz(x ^ y) / BITS is the length of a common prefix.
//
// DNA (sigma <= 4) packs at 2 bits/base: a 2^30-base text is 256 MiB and stays resident
// in the 256 MiB Infinity Cache for the random probes of the LCP and key kernels.
#pragma once
#include "common.hpp"

namespace nolzss {

// Terminators cut the text into segments that no match may cross: the end of the text (always the
// last entry, position n) and, in SEGMENTED texts, the unique sentinel bytes of a prepared
// multi-sequence / reverse-complement string (/root/reference/src/cpp/factorizer.cpp:110-147).
// A segmented text packs only its nucleotides (2 bits); a suffix comparison runs at most to the
// nearer terminator, and at a terminator the suffix that reaches it first is the smaller one
// (equal distance: the lower terminator index) -- the order the reference's unique sentinels give
// up to a relabelling of symbols, which the factorization does not depend on.
//
// INDEPENDENT sequences (the merged per-sequence batch, batch.hip): seq_shift != 0 puts the terminator
// index of a suffix -- the number of its sequence -- above bit seq_shift of its round-0 sort key.
// Suffixes then order by (sequence, suffix): the suffix array is the concatenation of the suffix
// arrays of the sequences, the LCP between neighbours of different sequences is 0, and every later
// stage (candidates, cursor, factor records) stays inside one sequence without knowing about it.
constexpr int kTermBlockShift = 12;
// their key: [record number][12 bases][4-bit length tag].  Records are short (batch.hip merges records
// below 2^21 bases), so 12 bases separate as well as 17 do in a 2^30-base text, and every 8 key bits
// less is a radix pass less.
constexpr int kIndSyms = 12, kIndTagBits = 4, kIndKeyBits = kIndSyms * 2 + kIndTagBits;
// LONG records (at most n / 2^16 of them): the record is the bucket of the segmented key sort
// (radix_sort_record_keys) and the 32-bit key holds [14 bases][4-bit length tag] only
constexpr int kRecSyms = 14, kRecTagBits = 4;
struct TermTable {
    const uint32_t *pos = nullptr;  // sorted terminator positions, pos[count-1] = n
    uint32_t count = 0;
    uint32_t end = 0;               // n (= pos[count - 1]): a text of one segment needs no table look-up
    // optional (tables with many terminators): coarse[b] = smallest k with pos[k] >= b << kTermBlockShift,
    // clamped to count - 1; (n >> kTermBlockShift) + 3 entries
    const uint32_t *coarse = nullptr;
    uint32_t seq_shift = 0;
    // independent sequences WITH their reverse complements (T1 $ .. Tk $ rc(Tk) $ .. rc(T1) $): segment t and
    // segment 2k - 1 - t belong to the same sequence
    uint32_t mirror = 0;
    // tables of at most kTermFew entries travel in the kernel arguments too (few[k] = pos[k], 0xffffffff behind the
    // last one; nfew = count, 0: not filled): a prepared reverse-complement string T $ rc(T) $ has three, and the first
    // radix pass, the direct rounds and the histogram of a segmented text ask for the terminator of every suffix
    uint32_t nfew = 0;
    uint32_t few[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
};
constexpr uint32_t kTermFew = 4;

// smallest k with pos[k] >= p and that terminator's position, for a table that travels in the kernel arguments
__device__ __forceinline__ uint32_t term_few(const TermTable &t, uint32_t p, uint32_t &pos_k) {
    // entries in front of p (the last entry is the end of the text, >= every suffix start)
    const uint32_t k = (t.few[0] < p ? 1u : 0u) + (t.few[1] < p ? 1u : 0u) + (t.few[2] < p ? 1u : 0u);
    pos_k = k == 0 ? t.few[0] : (k == 1 ? t.few[1] : (k == 2 ? t.few[2] : t.few[3]));
    return k;
}

// smallest k with pos[k] >= p (exists for every suffix start p < n)
__device__ __forceinline__ uint32_t term_lower_bound(const TermTable &t, uint32_t p) {
    if (t.nfew) {
        uint32_t unused;
        return term_few(t, p, unused);
    }
    uint32_t lo = 0, hi = t.count - 1;
    if (t.coarse) {
        lo = t.coarse[p >> kTermBlockShift];
        hi = t.coarse[(p >> kTermBlockShift) + 1];
    }
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (t.pos[mid] >= p)
            hi = mid;
        else
            lo = mid + 1;
    }
    return lo;
}

// index of the next terminator of suffix p and the symbols in front of it (one segment: no look-up at all)
__device__ __forceinline__ uint32_t term_limit(const TermTable &t, uint32_t p, uint32_t &k) {
    if (t.count == 1) {
        k = 0;
        return t.end - p;
    }
    if (t.nfew) {
        uint32_t pos_k;
        k = term_few(t, p, pos_k);
        return pos_k - p;
    }
    k = term_lower_bound(t, p);
    return t.pos[k] - p;
}

struct PackedText {
    const uint64_t *words = nullptr;  // ceil(n*bits/64) + 12 zero pad words
    uint32_t n = 0;
    int bits = 0;    // 2, 4 or 8
    int sigma = 0;   // distinct byte values present
    TermTable terms;
    bool segmented = false;  // terminators inside the text (packed as code 0, never compared)
};

// segmented texts: key = [17 symbols][5-bit length tag][8-bit terminator index]
constexpr int kSegSyms = 17, kSegTagBits = 5, kSegTermBits = 8;  // 47 key bits: 6 radix passes

// Symbols per initial sort key and width of the length tag that breaks ties between a
// suffix that ends inside the key window and its zero-padded longer neighbours.
template <int BITS> struct KeyLayout;
// 2-bit DNA: 17 bases + tag = 40 key bits -> 5 radix passes instead of 8.  At 2^30 bases about 6 %
// of the suffixes of random DNA collide by chance in 17 bases; they are pairs that the
// direct-comparison round (which real repeats need anyway) separates in its first step, which is
// cheaper than three more passes over all n keys (measured on MI355X: 29 -> 21 -> 17 bases =
// 335 -> 326 -> 322 ms on the 40 %-repeat text, 212 ms on random DNA).
template <> struct KeyLayout<2> { static constexpr int kSyms = 17, kTagBits = 6; };
template <> struct KeyLayout<4> { static constexpr int kSyms = 15, kTagBits = 4; };
template <> struct KeyLayout<8> { static constexpr int kSyms = 7, kTagBits = 8; };

// Plain one-segment 2-bit DNA, round 4: the SORTED key is 16 bases = 32 bits and nothing else -- the
// first four bases are the bucket of the most-significant-digit pass, the other twelve (24 bits) are
// sorted by three segmented passes: one pass less than the 40-bit key [17 bases][6-bit tag] takes.
// The length tag still travels, in the low byte of the stored key word [24 key bits][8-bit tag], but
// OUTSIDE the sorted digits: it only tells the regroup kernel which suffixes end inside the key window
// (they become groups of their own) and caps the LCPs read off the keys.  What the tag did for the
// ORDER -- a suffix that ends inside the window sorts in front of the longer suffixes that continue its
// zero-padded key -- comes from the stability of the sort: the first pass takes the last 16 suffixes of
// the text first, shortest first (element e < 16 is suffix n - 1 - e, element e >= 16 is suffix e - 16).
constexpr int kP16Syms = 16, kP16TagBits = 8;

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
// offset h0 (caller guarantees the first min(h0, limit) symbols match), capped at the nearer
// terminator and at `cap`.
template <int BITS>
__device__ __forceinline__ uint32_t suffix_lcp(const uint64_t *__restrict__ w, const TermTable &terms,
                                               uint32_t a, uint32_t b, uint32_t h0, uint32_t cap = 0xffffffffu) {
    constexpr uint32_t kPerWord = 64 / BITS;
    const uint32_t la = terms.pos[term_lower_bound(terms, a)] - a;
    const uint32_t lb = terms.pos[term_lower_bound(terms, b)] - b;
    uint32_t limit = la < lb ? la : lb;
    limit = limit < cap ? limit : cap;  // (the result is capped too)
    if (h0 >= limit) return limit;
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
// (>= cap when undecided).  A suffix that reaches its terminator first is the smaller one, as
// with the reference's appended terminator.
template <int BITS>
__device__ __forceinline__ int suffix_compare(const uint64_t *__restrict__ w, const TermTable &terms, uint32_t a,
                                              uint32_t b, uint32_t h0, uint32_t cap, uint32_t &lcp) {
    constexpr uint32_t kPerWord = 64 / BITS;
    const uint32_t ka = term_lower_bound(terms, a), kb = term_lower_bound(terms, b);
    const uint32_t la = terms.pos[ka] - a, lb = terms.pos[kb] - b;
    const uint32_t limit = la < lb ? la : lb;  // symbols before the nearer terminator
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
    if (limit <= cap) {  // a terminator is reached: nearer one first, then lower index first
        lcp = limit;
        if (la != lb) return la < lb ? -1 : 1;
        return ka < kb ? -1 : 1;
    }
    lcp = cap;
    return 0;
}

// Round-0 sort key of suffix i: its first K symbols and a length tag (plain texts), or
// [kSegSyms symbols][5-bit tag][8-bit terminator index] for segmented texts: a suffix that meets a
// terminator inside the key window gets a key of its own, so every group left after the sort
// consists of suffixes that agree on kSegSyms real nucleotides.
// (w = the 64 text bits at suffix i, sym_word: callers that want every load of a tile in flight before the
// first key is assembled fetch the two words themselves)
template <int BITS>
__device__ __forceinline__ uint64_t initial_key_of(uint64_t w, const TermTable &terms, bool segmented, uint32_t i) {
    // (one segment: no search and, above all, no load that the key would have to wait for behind its text
    // window -- the first radix pass generates 2^30 keys)
    uint32_t k;
    const uint32_t lim = term_limit(terms, i, k);  // symbols before the next terminator
    if (BITS == 2 && segmented && terms.seq_shift == 0) {
        const uint32_t tag = lim < (uint32_t)kSegSyms ? lim : (uint32_t)kSegSyms;
        uint64_t sym = w >> (64 - kSegSyms * 2);
        if (tag < (uint32_t)kSegSyms) sym &= ~((1ull << (2 * (kSegSyms - tag))) - 1ull);
        return (sym << (kSegTagBits + kSegTermBits)) | ((uint64_t)tag << kSegTermBits) |
               (tag < (uint32_t)kSegSyms ? (uint64_t)(k & 255u) : 0ull);
    }
    if (BITS == 2 && terms.seq_shift != 0) {  // independent sequences
        const uint32_t tag = lim < (uint32_t)kIndSyms ? lim : (uint32_t)kIndSyms;
        uint64_t sym = w >> (64 - kIndSyms * 2);
        if (tag < (uint32_t)kIndSyms) sym &= ~((1ull << (2 * (kIndSyms - (int)tag))) - 1ull);
        uint32_t seq = k;
        if (terms.mirror) {
            const uint32_t other = terms.count - 2 - k;
            seq = k < other ? k : other;
        }
        return (sym << kIndTagBits) | tag | ((uint64_t)seq << terms.seq_shift);
    }
    constexpr int K = KeyLayout<BITS>::kSyms;
    constexpr int TAG = KeyLayout<BITS>::kTagBits;
    const uint32_t tag = lim < (uint32_t)K ? lim : (uint32_t)K;
    uint64_t sym = w >> (64 - K * BITS);
    // symbols behind a terminator belong to the next segment (zero behind the end of the text)
    if (tag < (uint32_t)K) sym &= ~((1ull << (BITS * (K - (int)tag))) - 1ull);
    return (sym << TAG) | tag;
}

template <int BITS>
__device__ __forceinline__ uint64_t initial_key(const uint64_t *__restrict__ words, const TermTable &terms,
                                                bool segmented, uint32_t i) {
    return initial_key_of<BITS>(sym_word<BITS>(words, i), terms, segmented, i);
}

// the two words behind sym_word, and the window assembled from them
struct SymWords {
    uint64_t a, b;
};
template <int BITS> __device__ __forceinline__ SymWords sym_words(const uint64_t *__restrict__ w, uint64_t pos) {
    const uint64_t wi = (pos * BITS) >> 6;
    return SymWords{w[wi], w[wi + 1]};
}
template <int BITS> __device__ __forceinline__ uint64_t sym_word_of(const SymWords &x, uint64_t pos) {
    const int o = (int)((pos * BITS) & 63);
    return o ? ((x.a << o) | (x.b >> (64 - o))) : x.a;
}

}  // namespace nolzss
