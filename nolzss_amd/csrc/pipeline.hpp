// pipeline.hpp -- per-device context and the stage entry points of the factorization pipeline.
#pragma once
#include "common.hpp"
#include "text.hpp"

namespace nolzss {

struct Context {
    int device = 0;
    hipStream_t stream = nullptr;
    Arena arena;
    Profiler prof;
    uint32_t *h_pinned = nullptr;  // 64 words of pinned host memory for small read-backs

    Profiler *profiler() { return prof.enabled() ? &prof : nullptr; }
    // copy `count` (<= 64) device words to host and wait for them
    void read_back(const uint32_t *d_src, uint32_t *dst, int count);
};

// ---- stage 1: text packing -------------------------------------------------------------
// Scans the byte text for its alphabet, builds dense codes and packs it (2/4/8 bits/symbol).
PackedText pack_text(Context &ctx, const uint8_t *d_text, size_t n);

// ---- stage 2: suffix array (prefix doubling over radix sorts) ----------------------------
// sa[r] = start of the r-th smallest suffix, isa[i] = rank of suffix i.  Both arrays (n u32)
// are caller-allocated.  Returns the number of doubling rounds run.
int build_suffix_array(Context &ctx, const PackedText &text, uint32_t *sa, uint32_t *isa);

// ---- stage 3: LCP array ------------------------------------------------------------------
// lcp[0] = 0, lcp[r] = lcp(suffix sa[r-1], suffix sa[r]); lcp has n+1 entries, lcp[n] = 0.
void build_lcp(Context &ctx, const PackedText &text, const uint32_t *sa, uint32_t *lcp);

}  // namespace nolzss
