// pipeline.hpp -- per-device context and the stage entry points of the factorization pipeline.
#pragma once
#include "common.hpp"
#include "pyramid.hpp"
#include "text.hpp"

#include <vector>

namespace nolzss {

struct Context {
    int device = 0;
    hipStream_t stream = nullptr;
    Arena arena;
    Profiler prof;
    uint32_t *h_pinned = nullptr;  // 64 words of pinned host memory for small read-backs
    uint8_t *h_stage = nullptr;    // pinned host staging for uploads (grow-only, merged batch)
    size_t h_stage_cap = 0;
    // merged batch of long records (set for the duration of one run): the two permutation scatters of the
    // pipeline -- rank[sa[r]] and L*[sa[r]] -- stay inside the records (radix_sort.hpp, RecordScatterPlan)
    const struct RecordScatterPlan *rec_plan = nullptr;

    Profiler *profiler() { return prof.enabled() ? &prof : nullptr; }
    // copy `count` (<= 64) device words to host and wait for them
    void read_back(const uint32_t *d_src, uint32_t *dst, int count);
};

// ---- stage 1: text packing -------------------------------------------------------------
// Scans the byte text for its alphabet, builds dense codes and packs it (2/4/8 bits/symbol).
// A text of upper-case nucleotides plus at most 250 byte values that occur exactly once each is
// packed SEGMENTED at 2 bits per base, the unique bytes becoming terminators (text.hpp).
PackedText pack_text(Context &ctx, const uint8_t *d_text, size_t n);
// Many INDEPENDENT nucleotide sequences in one text (text.hpp, TermTable::seq_shift): records of
// A/C/G/T with one separator byte at each of the sorted positions.  false if the text holds other bytes.
// mirror: the text is T1 $ .. Tk $ rc(Tk) $ .. rc(T1) $ and segment t belongs with segment 2k - 1 - t.
bool pack_independent_text(Context &ctx, const uint8_t *d_text, size_t n, const std::vector<uint32_t> &separators,
                           PackedText &out, bool mirror = false);

// ---- stages 2+3: suffix array (prefix doubling over radix sorts) and LCP array -------------
// sa[r] = start of the r-th smallest suffix, isa[i] = rank of suffix i PLUS ONE (n u32 each);
// lcp[0] = 0, lcp[r] = lcp(suffix sa[r-1], suffix sa[r]), lcp[n] = 0 (n + 1 u32).  All
// caller-allocated.  LCP entries between suffixes that round 0 already separates come straight
// from the sort keys; only the others compare packed text.  Returns the doubling rounds run.
// isa_deferred (optional): the caller can do without isa[] until the factor-length codes have been brought into
// text order -- if the direct rounds finish the suffix array (no doubling round needs rank[]), isa[] is then NOT
// written here and *isa_deferred = true: the caller hands isa to the permutation of the codes, which delivers it
// as a second value (build_lstar's isa_fill; radix_sort.hpp, bucketed_scatter with out2).
int build_suffix_array(Context &ctx, const PackedText &text, uint32_t *sa, uint32_t *isa, uint32_t *lcp,
                       bool *isa_deferred = nullptr);
// The range-minimum pyramid over lcp[0..n]; checks on the way that the construction left no boundary
// undecided (and compares those suffixes in the text if it did).
Pyramid build_lcp_pyramid(Context &ctx, const PackedText &text, const uint32_t *sa, uint32_t *lcp);

struct Pyramid;

// ---- stage 4: per-position factor length codes (lpnf.hip) ---------------------------------
// lstar[i] = L*[i] (0 = literal).  Returns the number of positions that needed the exact search.
// isa_fill (optional): isa[] has not been written yet (build_suffix_array, isa_deferred): it is filled here.
// fill_pyramids (optional): Psa / Plcp are allocated (alloc_pyramid over sa / lcp) but not computed: the tile kernel
// of this stage writes their first level from the blocks it has in LDS and the upper levels are filled here -- the
// caller skips build_pyramid / build_lcp_pyramid (whose check for undecided LCP entries happens here too).
uint32_t build_lstar(Context &ctx, uint32_t n, const uint32_t *sa, const uint32_t *isa, const uint32_t *lcp,
                     const Pyramid &Psa, const Pyramid &Plcp, uint32_t *lstar, uint32_t *isa_fill = nullptr,
                     const PackedText *fill_pyramids = nullptr);
// pieces of build_lcp_pyramid for that form (suffix_array.hip): the code above which an LCP entry counts as
// undecided, the comparison of the suffixes around every undecided entry, and the test hook that leaves one undecided
uint32_t pending_threshold();
void finish_pending_lcp(Context &ctx, const PackedText &text, const uint32_t *sa, uint32_t *lcp);
void inject_pending_for_test(Context &ctx, uint32_t *lcp, uint32_t n);

// ---- stage 5: greedy cursor + factor records (chain.hip) ------------------------------------
uint32_t resolve_chain(Context &ctx, uint32_t n, uint32_t start_pos, const uint32_t *lstar, const uint32_t *sa,
                       const uint32_t *isa, const uint32_t *lcp, const Pyramid &Psa, const Pyramid &Plcp,
                       void **d_factors_out, uint32_t rcN = 0, const Pyramid *Pmax = nullptr,
                       uint32_t **d_fpos_out = nullptr, const TermTable *rebase = nullptr);

// ---- reverse-complement mode (rc.hip): whole pipeline over the prepared string S -------------
uint32_t run_rc_pipeline(Context &ctx, const uint8_t *d_S, size_t m, size_t start_pos, void **d_factors_out);
// the same over a text that has already been packed (merged batch)
uint32_t run_rc_pipeline_packed(Context &ctx, const PackedText &text, size_t start_pos, void **d_factors_out);
// d_S (2n + 2 bytes) = T' sep revcomp(T') sep for the n bytes d_T = upper-case records with separator bytes
// between them; bytes that are not nucleotides (the separators) are copied to their mirror position.
void prepare_batch_rc_on_device(Context &ctx, const uint8_t *d_T, uint32_t n, uint8_t separator, uint8_t *d_S);
// d_S (2n + 2 bytes) = prepared string of the single sequence d_T; returns the index of the first
// invalid nucleotide or 0xffffffff.
uint32_t prepare_single_rc_on_device(Context &ctx, const uint8_t *d_T, uint32_t n, uint8_t *d_S);

}  // namespace nolzss
