// batch.hip -- the per-sequence batch: merged runs of many records, the plan that deals records to runs and devices, its C entry points
// (part of the C ABI layer of libnolzss_hip.so, include/nolzss_hip.h; shared declarations: api_internal.hpp)
#include "api_internal.hpp"

namespace nolzss {
namespace api {

// ---- merged batch: many short nucleotide records in ONE pipeline run ----------------------------
// A record of a few thousand bases cannot fill the GPU, and one pipeline run costs the same ~100
// launches and a dozen read-backs whatever its size (0.12 ms per 4 Ki-base record with four lanes:
// 35 Mbases/s).  Short records are therefore concatenated, one separator byte between them, and
// factorized together as INDEPENDENT sequences (text.hpp, TermTable::seq_shift): suffixes order by
// (record, suffix), so no match crosses a record, the separators become literal factors, and the
// records' factor lists are the stretches between them, rebased to the record's start.
constexpr uint8_t kBatchSeparator = 0x01;

// Runs of long records: for the duration of the run the context carries the plan that keeps the two permutation
// scatters of the pipeline inside the records (radix_sort.hpp; NOLZSS_NO_RECORD_SCATTER=1 switches it off)
struct RecordPlanScope {
    Context &ctx;
    RecordScatterPlan plan;
    RecordPlanScope(Context &c, const std::vector<uint32_t> &seps, uint32_t n) : ctx(c) {
        static const bool off = getenv("NOLZSS_NO_RECORD_SCATTER") != nullptr;
        if (off || seps.empty()) return;
        std::vector<uint32_t> terms(seps);
        terms.push_back(n);
        if (record_scatter_plan(terms, n, ctx.arena, ctx.stream, plan)) ctx.rec_plan = &plan;
    }
    ~RecordPlanScope() { ctx.rec_plan = nullptr; }
    RecordPlanScope(const RecordPlanScope &) = delete;
    RecordPlanScope &operator=(const RecordPlanScope &) = delete;
};
constexpr size_t kMergeChunkBases = size_t(1) << 25;  // bases per merged run (5.5 Gbases/s on the device from 2^24 up)
constexpr size_t kMergeLanes = 2;  // runs in flight per device: one gathers / downloads while the other computes

// smallest j with recs[j].start >= the position of separator k: the separator's own literal factor
__global__ void batch_bounds_kernel(const nolzss_factor *__restrict__ recs, uint32_t z,
                                    const uint32_t *__restrict__ seps, uint32_t nsep, uint32_t *__restrict__ fidx,
                                    uint32_t *__restrict__ err) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nsep) return;
    const uint64_t target = seps[k];
    uint32_t lo = 0, hi = z;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (recs[mid].start >= target)
            hi = mid;
        else
            lo = mid + 1;
    }
    fidx[k] = lo;
    // a separator matches nothing: it must be a factor of length 1 that starts exactly there
    if (lo >= z || recs[lo].start != target || recs[lo].length != 1) atomicOr(err, 1u);
}

// smallest j with fpos[j] >= the position of separator k (its own literal factor)
__global__ void batch_bounds_pos_kernel(const uint32_t *__restrict__ fpos, uint32_t z,
                                        const uint32_t *__restrict__ seps, uint32_t nsep, uint32_t *__restrict__ fidx,
                                        uint32_t *__restrict__ err) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nsep) return;
    const uint32_t target = seps[k];
    uint32_t lo = 0, hi = z;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (fpos[mid] >= target)
            hi = mid;
        else
            lo = mid + 1;
    }
    fidx[k] = lo;
    // a separator matches nothing: a factor of length 1 starts exactly there
    if (lo >= z || fpos[lo] != target || (lo + 1 < z && fpos[lo + 1] != target + 1)) atomicOr(err, 1u);
}

// record-relative coordinates: start and ref minus the first position of the factor's record
__global__ void batch_rebase_kernel(nolzss_factor *__restrict__ recs, uint32_t z, TermTable terms) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= z) return;
    const uint64_t p = recs[j].start;
    const uint32_t k = term_lower_bound(terms, (uint32_t)p);
    const uint64_t base = k ? (uint64_t)terms.pos[k - 1] + 1 : 0;
    recs[j].start = p - base;
    // (reverse-complement factors carry NOLZSS_RC_MASK in the top bit of ref, over a position of T)
    const uint64_t ref = recs[j].ref, flag = ref & (1ull << 63);
    recs[j].ref = ((ref ^ flag) - base) | flag;
}

// Factorizes records ids[0..c) (all non-empty) in one run.  Returns false, with nothing written, when
// the records hold anything but A/C/G/T.  Factors of all records arrive in ONE malloc'ed block
// (appended to `blocks`); fs[id] points into it.
// host memory for a block of factor records: large blocks on transparent huge pages, where the
// first touch of the download costs one fault per 2 MiB instead of one per 4 KiB
void *alloc_factor_block(size_t bytes) {
    constexpr size_t kHuge = size_t(1) << 21;
    if (bytes >= 4 * kHuge) {
        void *p = nullptr;
        if (posix_memalign(&p, kHuge, (bytes + kHuge - 1) & ~(kHuge - 1)) == 0 && p) {
            (void)madvise(p, (bytes + kHuge - 1) & ~(kHuge - 1), MADV_HUGEPAGE);
            return p;
        }
    }
    return std::malloc(bytes);
}

bool run_merged_chunk(Context &ctx, const uint8_t *const *texts, const size_t *lens, const std::vector<size_t> &ids,
                      bool with_rc, size_t *zs, nolzss_factor **fs, std::vector<void *> *blocks) {
    const bool trace = getenv("NOLZSS_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    const size_t c = ids.size();
    size_t n = c - 1;
    for (size_t j : ids) n += lens[j];
    // long records go up one by one, straight into their place in the device text; short ones are gathered
    // in the pinned staging buffer first (a copy call per 4 KiB record would cost more than the gather)
    const bool direct = n / c >= (size_t(1) << 16);
    uint8_t *host = direct ? nullptr : host_stage(ctx, n);
    std::vector<uint32_t> seps;
    seps.reserve(c - 1);
    {
        size_t at = 0;
        for (size_t k = 0; k < c; ++k) {
            if (!direct) std::memcpy(host + at, texts[ids[k]], lens[ids[k]]);
            at += lens[ids[k]];
            if (k + 1 < c) {
                seps.push_back((uint32_t)at);
                if (!direct) host[at] = kBatchSeparator;
                ++at;
            }
        }
    }
    const double t_concat = since();
    Arena &arena = ctx.arena;
    hipStream_t s = ctx.stream;
    const size_t m2 = 2 * n + 2;  // with_rc: T' sep revcomp(T') sep
    reserve_arena_for(ctx, with_rc ? m2 : n, (with_rc ? m2 : 0) + n + 32 * c + (size_t(1) << 20));
    const size_t mark = arena.mark();
    struct Rewind {
        Arena &a;
        size_t m;
        ~Rewind() { a.rewind(m); }
    } rewind{arena, mark};
    uint8_t *d_text = arena.alloc<uint8_t>(n);
    if (direct) {
        ProfScope ps(ctx.profiler(), "batch_upload", s, (double)n);
        size_t at = 0;
        for (size_t k = 0; k < c; ++k) {
            upload_bytes(ctx, d_text + at, texts[ids[k]], lens[ids[k]]);
            at += lens[ids[k]];
            if (k + 1 < c) HIP_CHECK(hipMemsetAsync(d_text + at++, kBatchSeparator, 1, s));
        }
    } else {
        HIP_CHECK(hipMemcpyAsync(d_text, host, n, hipMemcpyHostToDevice, s));
    }
    PackedText text;
    void *d_recs = nullptr;
    uint32_t *d_fpos = nullptr;
    uint32_t z = 0;
    if (with_rc) {
        // the layout of prepare_multiple_dna_sequences_w_rc (factorizer.cpp:128-169) for any number of
        // records: segment t and segment 2c - 1 - t are a record and its reverse complement
        uint8_t *d_S = arena.alloc<uint8_t>(m2);
        prepare_batch_rc_on_device(ctx, d_text, (uint32_t)n, kBatchSeparator, d_S);
        std::vector<uint32_t> terms(seps);
        terms.reserve(2 * c);
        terms.push_back((uint32_t)n);
        for (size_t k = seps.size(); k-- > 0;) terms.push_back((uint32_t)(2 * n - seps[k]));
        terms.push_back((uint32_t)(2 * n + 1));
        if (!pack_independent_text(ctx, d_S, m2, terms, text, true)) return false;
        z = run_rc_pipeline_packed(ctx, text, 0, &d_recs);
    } else {
        if (!pack_independent_text(ctx, d_text, n, seps, text)) return false;
        RecordPlanScope plan_scope(ctx, seps, (uint32_t)n);
        uint32_t *sa = arena.alloc<uint32_t>(n);
        uint32_t *isa = arena.alloc<uint32_t>(n);
        uint32_t *lcp = arena.alloc<uint32_t>(n + 1);
        // (isa: left to the permutation of the codes when the direct rounds finish the suffix array, pipeline.hpp)
        bool isa_deferred = false;
        build_suffix_array(ctx, text, sa, isa, lcp, &isa_deferred);
        // (pyramids: allocated here, filled by build_lstar -- first level from the candidate kernel)
        const Pyramid Psa = alloc_pyramid(sa, (uint32_t)n, arena), Plcp = alloc_pyramid(lcp, (uint32_t)n + 1, arena);
        uint32_t *lstar = arena.alloc<uint32_t>(n);
        build_lstar(ctx, (uint32_t)n, sa, isa, lcp, Psa, Plcp, lstar, isa_deferred ? isa : nullptr, &text);
        // counts come from the factor starts; records are built only when the caller wants them, and leave
        // the factor kernel in record coordinates
        z = resolve_chain(ctx, (uint32_t)n, 0, lstar, sa, isa, lcp, Psa, Plcp, fs ? &d_recs : nullptr, 0, nullptr,
                          &d_fpos, fs ? &text.terms : nullptr);
    }
    nolzss_factor *recs = static_cast<nolzss_factor *>(d_recs);
    // where the records' factor lists start and end
    std::vector<uint32_t> fidx(c, z);
    if (c > 1) {
        uint32_t *d_fidx = arena.alloc<uint32_t>(c);
        HIP_CHECK(hipMemsetAsync(d_fidx + (c - 1), 0, sizeof(uint32_t), s));  // error flag
        if (with_rc)
            batch_bounds_kernel<<<(unsigned)div_up(c - 1, 256), 256, 0, s>>>(recs, z, text.terms.pos, (uint32_t)(c - 1),
                                                                             d_fidx, d_fidx + (c - 1));
        else
            batch_bounds_pos_kernel<<<(unsigned)div_up(c - 1, 256), 256, 0, s>>>(d_fpos, z, text.terms.pos,
                                                                                 (uint32_t)(c - 1), d_fidx, d_fidx + (c - 1));
        KERNEL_CHECK();
        HIP_CHECK(hipMemcpyAsync(fidx.data(), d_fidx, c * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        if (fidx[c - 1]) throw HipError("merged batch: a separator is not a literal factor");
        fidx[c - 1] = z;
    }
    nolzss_factor *block = nullptr;
    if (fs && z) {
        if (with_rc) {
            batch_rebase_kernel<<<(unsigned)div_up(z, 256), 256, 0, s>>>(recs, z, text.terms);
            KERNEL_CHECK();
        }
        block = static_cast<nolzss_factor *>(alloc_factor_block(sizeof(nolzss_factor) * (size_t)z));
        if (!block) throw std::bad_alloc();
        try {
            download_bytes(ctx, block, recs, sizeof(nolzss_factor) * (size_t)z);
        } catch (...) {
            std::free(block);
            throw;
        }
    }
    const hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        std::free(block);
        HIP_CHECK(e);
    }
    ctx.prof.collect();
    if (trace)
        fprintf(stderr, "[nolzss] merged batch run: %zu records, %zu symbols, %u factors: gather %.1f ms, device%s %.1f ms\n",
                c, n, z, t_concat, block ? " + download" : "", since() - t_concat);
    if (block) blocks->push_back(block);
    for (size_t k = 0; k < c; ++k) {
        const uint32_t a = k ? fidx[k - 1] + 1 : 0, b = fidx[k];
        zs[ids[k]] = b - a;
        if (fs) fs[ids[k]] = (block && b > a) ? block + a : nullptr;
    }
    return true;
}

// ---- the same with the records already resident in device memory ----------------------------
struct GatherRec {
    const uint8_t *src;
    uint64_t off, len;  // record k goes to d_text[off, off + len), its separator (all but the last) behind it
};
// grid (pieces, records): a workgroup copies 4 KiB pieces of its record
__global__ __launch_bounds__(256) void gather_records_kernel(const GatherRec *__restrict__ recs, uint32_t c,
                                                             uint8_t *__restrict__ dst, uint8_t sep) {
    for (uint32_t k = blockIdx.y; k < c; k += gridDim.y) {
        const GatherRec r = recs[k];
        for (uint64_t p0 = (uint64_t)blockIdx.x * 4096; p0 < r.len; p0 += (uint64_t)gridDim.x * 4096) {
            uint8_t b[16];  // (all loads first, none behind a branch: bytes past the end read the last byte again)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint64_t p = p0 + (uint64_t)j * 256 + threadIdx.x;
                b[j] = r.src[p < r.len ? p : r.len - 1];
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint64_t p = p0 + (uint64_t)j * 256 + threadIdx.x;
                if (p < r.len) dst[r.off + p] = b[j];
            }
        }
        if (blockIdx.x == 0 && threadIdx.x == 0 && k + 1 < c) dst[r.off + r.len] = sep;
    }
}

// Records ids[0..c) (device pointers, all non-empty) as ONE run of independent sequences: gathered on the
// device, factorized together, counted per record (emit: the factor records are also built in device
// memory, in record-relative coordinates, as the per-record runs leave them).  Returns false, with nothing
// written, when the records hold anything but A/C/G/T.
bool run_merged_chunk_device(Context &ctx, const void *const *d_texts, const size_t *lens, const std::vector<size_t> &ids,
                             bool emit, size_t *zs) {
    const size_t c = ids.size();
    size_t n = c - 1;
    for (size_t j : ids) n += lens[j];
    Arena &arena = ctx.arena;
    hipStream_t s = ctx.stream;
    reserve_arena_for(ctx, n, n + 64 * c + (size_t(1) << 20));
    const size_t mark = arena.mark();
    struct Rewind {
        Arena &a;
        size_t m;
        ~Rewind() { a.rewind(m); }
    } rewind{arena, mark};
    std::vector<GatherRec> table(c);
    std::vector<uint32_t> seps;
    seps.reserve(c - 1);
    {
        size_t at = 0;
        for (size_t k = 0; k < c; ++k) {
            table[k] = GatherRec{static_cast<const uint8_t *>(d_texts[ids[k]]), at, lens[ids[k]]};
            at += lens[ids[k]];
            if (k + 1 < c) seps.push_back((uint32_t)at++);
        }
    }
    uint8_t *d_text = arena.alloc<uint8_t>(n);
    GatherRec *d_table = arena.alloc<GatherRec>(c);
    HIP_CHECK(hipMemcpyAsync(d_table, table.data(), sizeof(GatherRec) * c, hipMemcpyHostToDevice, s));
    {
        ProfScope ps(ctx.profiler(), "batch_gather", s, 2.0 * (double)n);
        const size_t longest = lens[*std::max_element(ids.begin(), ids.end(), [&](size_t a, size_t b) { return lens[a] < lens[b]; })];
        const unsigned gy = (unsigned)std::min<size_t>(c, 32768);
        const unsigned gx = (unsigned)std::max<size_t>(1, std::min<size_t>(div_up(longest, 4096), div_up((size_t)65536, gy)));
        gather_records_kernel<<<dim3(gx, gy), 256, 0, s>>>(d_table, (uint32_t)c, d_text, kBatchSeparator);
        KERNEL_CHECK();
    }
    HIP_CHECK(hipStreamSynchronize(s));  // table is a local vector
    PackedText text;
    if (!pack_independent_text(ctx, d_text, n, seps, text)) return false;
    RecordPlanScope plan_scope(ctx, seps, (uint32_t)n);
    uint32_t *sa = arena.alloc<uint32_t>(n);
    uint32_t *isa = arena.alloc<uint32_t>(n);
    uint32_t *lcp = arena.alloc<uint32_t>(n + 1);
    bool isa_deferred = false;  // (as in run_merged_chunk)
    build_suffix_array(ctx, text, sa, isa, lcp, &isa_deferred);
    // (pyramids: allocated here, filled by build_lstar -- first level from the candidate kernel)
    const Pyramid Psa = alloc_pyramid(sa, (uint32_t)n, arena), Plcp = alloc_pyramid(lcp, (uint32_t)n + 1, arena);
    uint32_t *lstar = arena.alloc<uint32_t>(n);
    build_lstar(ctx, (uint32_t)n, sa, isa, lcp, Psa, Plcp, lstar, isa_deferred ? isa : nullptr, &text);
    void *d_recs = nullptr;
    uint32_t *d_fpos = nullptr;
    const uint32_t z = resolve_chain(ctx, (uint32_t)n, 0, lstar, sa, isa, lcp, Psa, Plcp, emit ? &d_recs : nullptr, 0,
                                     nullptr, &d_fpos, emit ? &text.terms : nullptr);
    std::vector<uint32_t> fidx(c, z);
    if (c > 1) {
        uint32_t *d_fidx = arena.alloc<uint32_t>(c);
        HIP_CHECK(hipMemsetAsync(d_fidx + (c - 1), 0, sizeof(uint32_t), s));  // error flag
        batch_bounds_pos_kernel<<<(unsigned)div_up(c - 1, 256), 256, 0, s>>>(d_fpos, z, text.terms.pos, (uint32_t)(c - 1),
                                                                             d_fidx, d_fidx + (c - 1));
        KERNEL_CHECK();
        HIP_CHECK(hipMemcpyAsync(fidx.data(), d_fidx, c * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        if (fidx[c - 1]) throw HipError("merged batch: a separator is not a literal factor");
        fidx[c - 1] = z;
    }
    HIP_CHECK(hipStreamSynchronize(s));
    ctx.prof.collect();
    for (size_t k = 0; k < c; ++k) {
        const uint32_t a = k ? fidx[k - 1] + 1 : 0, b = fidx[k];
        zs[ids[k]] = b - a;
    }
    return true;
}

std::atomic<uint64_t> g_merged_records{0}, g_single_records{0};

// a worker thread failed: the same kind of error, with the same text, for the calling thread
[[noreturn]] void rethrow_worker_error(int status, const std::string &message) {
    switch (status) {
    case NOLZSS_ERR_INVALID_ARGUMENT: throw std::invalid_argument(message);
    case NOLZSS_ERR_NOMEM: throw std::bad_alloc();
    case NOLZSS_ERR_DEVICE: throw HipError(message);
    default: throw std::runtime_error(message);
    }
}

// blocks behind the factor arrays of a batch result (nolzss_free_batch)
std::mutex g_batch_mu;
std::map<nolzss_factor **, std::vector<void *>> g_batch_blocks;

size_t merge_below() {  // records shorter than this are merged (0: never)
    const char *e = getenv("NOLZSS_BATCH_MERGE_BELOW");
    return e ? (size_t)atoll(e) : (size_t(1) << 21);
}

// The shared worker of the batch entry points: factorizes m records, zs[j] factors each; fs (optional)
// receives the arrays, every malloc'ed block behind them is appended to `blocks`.
// The static part of the batch plan: which records share a merged run (`chunks`, in the order the lanes take them from
// one work queue) and which take a pipeline run of their own (`singles`); empty records are in neither (z = 0).
// Every record is in exactly one place (tests/test_host_logic.py::test_batch_plan_deals_every_record_once, through
// nolzss_debug_batch_plan).
BatchPlan plan_batch(const size_t *lens, size_t m, bool with_rc) {
    std::vector<size_t> singles;
    std::vector<std::vector<size_t>> chunks;
    // 1. which records are merged: short, non-empty ones, in chunks of consecutive records
    // (with_rc: each record as T s0 revcomp(T) s1, dna_w_rc_common; a run holds both strands)
    const size_t below = with_rc ? merge_below() / 2 : merge_below();
    static const size_t chunk_bases = [] {
        const char *e = getenv("NOLZSS_BATCH_MERGE_BASES");
        const long long v = e ? atoll(e) : 0;
        return v > 0 ? (size_t)v : kMergeChunkBases;
    }();
    const size_t run_bases = with_rc ? chunk_bases / 2 : chunk_bases;
    // Long records (plain mode): runs of about 2^28 bases, each record uploaded straight into the run's device
    // text; one lane uploads while the other computes.
    // (NOLZSS_BATCH_MERGE_LONG_BELOW=0: long records one pipeline run each, as before round 2; read per call)
    const size_t long_below = [] {
        const char *e = getenv("NOLZSS_BATCH_MERGE_LONG_BELOW");
        return e ? (size_t)atoll(e) : (size_t(1) << 27);
    }();
    const size_t long_run_bases = [] {
        const char *e = getenv("NOLZSS_BATCH_MERGE_LONG_BASES");
        const long long v = e ? atoll(e) : 0;
        return v > 0 ? (size_t)v : (size_t(1) << 28);
    }();
    // (with reverse complement a run holds both strands: half the bases per run, 128 records of 4 Mi bases
    // 3.5 -> 4.0 Gbases/s)
    const size_t long_cut = with_rc ? long_below / 2 : long_below;
    const size_t long_run = with_rc ? long_run_bases / 4 : long_run_bases;  // (2^24 / 2^26 / 2^27 bases per run: 3.7 / 4.0 / 3.7)
    std::vector<size_t> longs;
    if (below > 0)
        for (size_t j = 0; j < m; ++j)
            if (lens[j] >= below && lens[j] < long_cut) longs.push_back(j);
    if (longs.size() >= 2) {
        size_t total = 0;
        for (size_t j : longs) total += lens[j] + 1;
        size_t runs = div_up(total - longs.size(), long_run);  // (the separators do not count)
        if (runs < 2 && total >= (with_rc ? size_t(1) << 26 : size_t(1) << 27)) runs = 2;
        const size_t share = div_up(total, runs);
        std::vector<size_t> cur;
        size_t cur_bases = 0;
        for (size_t j : longs) {
            cur.push_back(j);
            cur_bases += lens[j] + 1;
            if (cur_bases >= share) {
                chunks.push_back(std::move(cur));
                cur.clear();
                cur_bases = 0;
            }
        }
        if (!cur.empty()) chunks.push_back(std::move(cur));
    } else {
        longs.clear();
    }
    const bool merge_longs = !longs.empty();
    {
        // equal shares: as many runs as the limit asks for, each with its part of the bases
        size_t short_bases = 0;
        for (size_t j = 0; j < m; ++j)
            if (lens[j] && lens[j] < below) short_bases += lens[j] + 1;
        const size_t runs = div_up(short_bases ? short_bases : 1, run_bases);
        const size_t share = div_up(short_bases, runs);
        std::vector<size_t> cur;
        size_t cur_bases = 0;
        for (size_t j = 0; j < m; ++j) {
            if (lens[j] == 0) continue;  // z = 0
            if (lens[j] >= below) {
                if (!(merge_longs && lens[j] < long_cut)) singles.push_back(j);
                continue;
            }
            cur.push_back(j);
            cur_bases += lens[j] + 1;
            if (cur_bases >= share || cur.size() >= (size_t(1) << 23)) {
                chunks.push_back(std::move(cur));
                cur.clear();
                cur_bases = 0;
            }
        }
        if (!cur.empty()) chunks.push_back(std::move(cur));
        for (auto it = chunks.begin(); it != chunks.end();)
            if (it->size() < 2) {  // nothing to merge with
                singles.push_back((*it)[0]);
                it = chunks.erase(it);
            } else {
                ++it;
            }
    }
    return BatchPlan{std::move(chunks), std::move(singles)};
}

// longest-processing-time-first assignment of the single records to n_dev devices (singles sorted by length first)
std::vector<std::vector<size_t>> lpt_plan_singles(std::vector<size_t> &singles, const size_t *lens, size_t n_dev) {
    std::stable_sort(singles.begin(), singles.end(), [&](size_t a, size_t b) { return lens[a] > lens[b]; });
    std::vector<std::vector<size_t>> plan(n_dev);
    std::vector<size_t> load(n_dev, 0);
    for (size_t j : singles) {
        const size_t d = (size_t)(std::min_element(load.begin(), load.end()) - load.begin());
        plan[d].push_back(j);
        load[d] += lens[j];
    }
    return plan;
}

void factorize_many(const uint8_t *const *texts, const size_t *lens, size_t m, const int *devices, size_t n_dev,
                    bool with_rc, size_t *zs, nolzss_factor **fs, std::vector<void *> &blocks) {
    // 1. which records are merged: short, non-empty ones, in chunks of consecutive records (plan_batch)
    BatchPlan bp = plan_batch(lens, m, with_rc);
    std::vector<std::vector<size_t>> &chunks = bp.chunks;
    std::vector<size_t> &singles = bp.singles;
    std::mutex out_mu;
    // 2. merged chunks: kMergeLanes host threads per device, each with its own stream and arena
    if (!chunks.empty()) {
        const size_t workers = n_dev * kMergeLanes;
        std::vector<int> status(workers, NOLZSS_OK);
        std::vector<std::string> messages(workers);
        std::atomic<size_t> next{0};
        auto worker = [&](size_t w) {
            status[w] = guarded([&] {
                Session ses(devices[w % n_dev], nullptr, (int)(w / n_dev));  // the devices first, then their second lanes
                for (;;) {
                    const size_t k = next.fetch_add(1);
                    if (k >= chunks.size()) break;
                    std::vector<void *> mine;
                    const bool ok = run_merged_chunk(ses.ctx(), texts, lens, chunks[k], with_rc, zs, fs, &mine);
                    std::lock_guard<std::mutex> lk(out_mu);
                    if (ok) {
                        blocks.insert(blocks.end(), mine.begin(), mine.end());
                        g_merged_records += chunks[k].size();
                    } else  // other alphabets: one by one
                        singles.insert(singles.end(), chunks[k].begin(), chunks[k].end());
                }
            });
            if (status[w] != NOLZSS_OK) messages[w] = g_error;
        };
        std::vector<std::thread> threads;
        for (size_t w = 0; w < workers && w < chunks.size(); ++w) threads.emplace_back(worker, w);
        for (auto &t : threads) t.join();
        for (size_t w = 0; w < workers; ++w)
            if (status[w] != NOLZSS_OK) rethrow_worker_error(status[w], messages[w]);
    }
    if (singles.empty()) return;
    // 3. the others one by one: longest-processing-time-first assignment of sequences to devices
    std::vector<std::vector<size_t>> plan = lpt_plan_singles(singles, lens, n_dev);
    // Several pipelines per device: a 4 Mi-base sequence neither fills the GPU for long nor
    // hides its own launch / read-back gaps, so each device runs `lanes` sequences at a time,
    // every lane with its own stream and arena, fed from the device's queue.
    static const size_t lanes = [] {
        const char *e = getenv("NOLZSS_BATCH_LANES");
        const long v = e ? atol(e) : 8;
        return (size_t)(v < 1 ? 1 : (v > kMaxLanes ? kMaxLanes : v));
    }();
    std::vector<int> status(n_dev * lanes, NOLZSS_OK);
    std::vector<std::string> messages(n_dev * lanes);
    std::vector<std::atomic<size_t>> next(n_dev);
    for (auto &a : next) a.store(0);
    auto worker = [&](size_t d, size_t lane) {
        const size_t w = d * lanes + lane;
        status[w] = guarded([&] {
            std::unique_ptr<Session> ses;  // (dna_w_rc_common opens the lane's session itself)
            if (!with_rc) ses.reset(new Session(devices[d], nullptr, (int)lane));
            for (;;) {
                const size_t k = next[d].fetch_add(1);
                if (k >= plan[d].size()) break;
                const size_t j = plan[d][k];
                if (with_rc)
                    dna_w_rc_common(texts[j], nullptr, lens[j], devices[d], nullptr, fs ? 2 : 0, fs ? &fs[j] : nullptr, &zs[j], (int)lane);
                else
                    zs[j] = run_plain_host(ses->ctx(), texts[j], lens[j], 0, fs ? &fs[j] : nullptr, nullptr);
                ++g_single_records;
                if (fs && fs[j]) {
                    std::lock_guard<std::mutex> lk(out_mu);
                    blocks.push_back(fs[j]);
                }
            }
        });
        if (status[w] != NOLZSS_OK) messages[w] = g_error;
    };
    {
        std::vector<std::thread> threads;
        for (size_t d = 0; d < n_dev; ++d) {
            // no more lanes than arenas for the device's longest record fit its memory
            size_t fit = lanes;
            if (!plan[d].empty()) {
                const size_t longest = lens[plan[d][0]];
                const size_t need = arena_bytes_for(with_rc ? 2 * longest + 2 : longest) + 3 * longest;
                size_t free_b = 0, total_b = 0;
                if (hipSetDevice(devices[d]) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                    fit = (size_t)((double)total_b * 0.85) / need;
                    fit = fit < 1 ? 1 : (fit > lanes ? lanes : fit);
                }
            }
            for (size_t lane = 0; lane < fit; ++lane)
                if (lane < plan[d].size()) threads.emplace_back(worker, d, lane);
        }
        for (auto &t : threads) t.join();
    }
    for (size_t d = 0; d < status.size(); ++d)
        if (status[d] != NOLZSS_OK) rethrow_worker_error(status[d], messages[d]);
}

}  // namespace api
}  // namespace nolzss

using namespace nolzss;
using namespace nolzss::api;

extern "C" {

static int factorize_batch_impl(const uint8_t *const *texts, const size_t *lens, size_t m, const int *devices,
                                size_t n_dev, bool with_rc, nolzss_factor ***out, size_t **z) {
    return guarded([&] {
        if (!z) throw std::invalid_argument("output pointer is null");
        *z = nullptr;
        if (out) *out = nullptr;
        if (m && (!texts || !lens)) throw std::invalid_argument("sequence array is null");
        if (!devices || n_dev == 0) throw std::invalid_argument("device list is empty");
        for (size_t j = 0; j < m; ++j) check_text_args(texts[j], lens[j], 0);
        size_t *zs = static_cast<size_t *>(std::calloc(m ? m : 1, sizeof(size_t)));
        nolzss_factor **fs = out ? static_cast<nolzss_factor **>(std::calloc(m ? m : 1, sizeof(nolzss_factor *)))
                                 : nullptr;
        if (!zs || (out && !fs)) {
            std::free(zs);
            std::free(fs);
            throw std::bad_alloc();
        }
        std::vector<void *> blocks;
        try {
            factorize_many(texts, lens, m, devices, n_dev, with_rc, zs, fs, blocks);
            if (fs) {
                std::lock_guard<std::mutex> lk(g_batch_mu);
                g_batch_blocks[fs] = std::move(blocks);
            }
        } catch (...) {
            for (void *b : blocks) std::free(b);
            std::free(fs);
            std::free(zs);
            throw;
        }
        *z = zs;
        if (out) *out = fs;
    });
}

int nolzss_factorize_batch(const uint8_t *const *texts, const size_t *lens, size_t m, const int *devices,
                           size_t n_dev, nolzss_factor ***out, size_t **z) {
    return factorize_batch_impl(texts, lens, m, devices, n_dev, false, out, z);
}

int nolzss_factorize_batch_dna_w_rc(const uint8_t *const *texts, const size_t *lens, size_t m, const int *devices,
                                    size_t n_dev, nolzss_factor ***out, size_t **z) {
    return factorize_batch_impl(texts, lens, m, devices, n_dev, true, out, z);
}

// The per-sequence batch with the records already resident in device memory (the measurement form: no
// PCIe leg inside).  Every record takes its own pipeline run; `lanes` runs are in flight on the device,
// each lane with its own stream and arena, as for the host-buffer batch.
int nolzss_factorize_batch_device(const void *const *d_texts, const size_t *lens, size_t m, int device, int emit,
                                  size_t *z) {
    return guarded([&] {
        if (m && (!d_texts || !lens || !z)) throw std::invalid_argument("sequence array is null");
        if (emit != 0 && emit != 1) throw std::invalid_argument("emit must be 0 (count) or 1 (records built in HBM)");
        for (size_t j = 0; j < m; ++j) check_text_args(d_texts[j], lens[j], 0);
        HIP_CHECK(hipSetDevice(device));
        // Records shorter than dev_merge_below are gathered into runs of independent sequences of about
        // dev_run_bases bases (run_merged_chunk_device): a 4 Mi-base record neither fills the GPU nor hides
        // the ~100 launches and dozen read-backs of its pipeline run, eight of them in flight on eight
        // streams reach 7 Gbases/s; one run over 256 of them works at the speed of a 2^30-base text.
        static const size_t dev_merge_below = [] {
            const char *e = getenv("NOLZSS_DEVICE_MERGE_BELOW");
            return e ? (size_t)atoll(e) : (size_t(1) << 28);
        }();
        static const size_t dev_run_bases = [] {
            const char *e = getenv("NOLZSS_DEVICE_MERGE_BASES");
            const long long v = e ? atoll(e) : 0;
            return v > 0 ? (size_t)v : (size_t(1) << 30);
        }();
        std::vector<size_t> order;  // the records that take a pipeline run of their own
        std::vector<std::vector<size_t>> chunks;
        {
            size_t short_bases = 0, short_count = 0;
            for (size_t j = 0; j < m; ++j)
                if (lens[j] && lens[j] < dev_merge_below) {
                    short_bases += lens[j] + 1;
                    ++short_count;
                }
            // (the separators do not count: 512 records of 2^22 bases are two runs of 2^30, not three;
            // two runs in flight fill each other's launch and read-back gaps: 2^28 bases go as two runs of
            // 2^27 rather than one)
            size_t runs = div_up(short_bases > short_count ? short_bases - short_count : 1, dev_run_bases);
            if (runs < 2 && short_bases >= (size_t(1) << 27)) runs = 2;
            const size_t share = div_up(short_bases, runs);
            std::vector<size_t> cur;
            size_t cur_bases = 0;
            for (size_t j = 0; j < m; ++j) {
                z[j] = 0;
                if (lens[j] == 0) continue;
                if (lens[j] >= dev_merge_below) {
                    order.push_back(j);
                    continue;
                }
                // (a run is one text: below the text limit, and the record table below 2^23 entries)
                if (!cur.empty() && (cur_bases + lens[j] + 1 > kMaxText / 2 || cur.size() >= (size_t(1) << 23))) {
                    chunks.push_back(std::move(cur));
                    cur.clear();
                    cur_bases = 0;
                }
                cur.push_back(j);
                cur_bases += lens[j] + 1;
                if (cur_bases >= share) {
                    chunks.push_back(std::move(cur));
                    cur.clear();
                    cur_bases = 0;
                }
            }
            if (!cur.empty()) chunks.push_back(std::move(cur));
            for (auto it = chunks.begin(); it != chunks.end();)
                if (it->size() < 2) {  // nothing to merge with
                    order.push_back((*it)[0]);
                    it = chunks.erase(it);
                } else {
                    ++it;
                }
        }
        if (!chunks.empty()) {
            static const size_t merge_lanes = [] {
                const char *e = getenv("NOLZSS_DEVICE_MERGE_LANES");
                const long v = e ? atol(e) : 2;
                return (size_t)(v < 1 ? 1 : (v > kMaxLanes ? kMaxLanes : v));
            }();
            const size_t workers = std::min(merge_lanes, chunks.size());
            std::vector<int> status(workers, NOLZSS_OK);
            std::vector<std::string> messages(workers);
            std::atomic<size_t> next{0};
            std::mutex mu;
            // Two runs of the same size started together stay in step: both in their radix passes (HBM-bound), then
            // both in the direct round (issue-bound), and gain nothing from each other -- or drift apart and overlap
            // well: 512 records of 4 Mi bases took 144 or 170 ms, whichever way a call happened to fall.  The
            // second lane therefore starts 20 ms late (NOLZSS_DEVICE_MERGE_STAGGER_MS; in bench.py: 145 / 162 / 164 ms
            // without, 146 / 146 / 145 ms with it, profiles/r03_fasta512_stagger.txt).
            // (20 ms is a sixth of a run of 2^30 bases; shorter runs wait in proportion -- a flat 20 ms made the two runs of
            // 64 records x 4 Mi bases, 10 ms each, follow each other on one lane: 18.8 -> 20.4 ms)
            static const long stagger_ms = getenv("NOLZSS_DEVICE_MERGE_STAGGER_MS") ? atol(getenv("NOLZSS_DEVICE_MERGE_STAGGER_MS")) : 20;
            // (only runs of SIMILAR size fall into lock-step: the wait applies when the first two runs are within a factor
            // of two of each other, and the lanes are spread over that one interval however many there are -- it does not
            // grow with the lane index.  include/nolzss_hip.h says that this call may sleep.)
            size_t first_bases = 0, second_bases = 0;
            for (size_t j : chunks[0]) first_bases += lens[j];
            if (chunks.size() > 1)
                for (size_t j : chunks[1]) second_bases += lens[j];
            const bool similar = second_bases * 2 >= first_bases && first_bases * 2 >= second_bases;
            const long stagger_us = !similar ? 0 : (long)((double)stagger_ms * 1000.0 * std::min(1.0, (double)first_bases / (double)(size_t(1) << 30)) /
                                                          (double)std::max<size_t>(1, workers - 1));
            auto worker = [&](size_t w) {
                status[w] = guarded([&] {
                    Session ses(device, nullptr, (int)w);
                    if (w > 0 && stagger_us > 0 && chunks.size() > 1)
                        std::this_thread::sleep_for(std::chrono::microseconds(stagger_us * (long)w));
                    for (;;) {
                        const size_t k = next.fetch_add(1);
                        if (k >= chunks.size()) break;
                        const bool ok = run_merged_chunk_device(ses.ctx(), d_texts, lens, chunks[k], emit == 1, z);
                        std::lock_guard<std::mutex> lk(mu);
                        if (ok)
                            g_merged_records += chunks[k].size();
                        else  // other alphabets: one by one
                            order.insert(order.end(), chunks[k].begin(), chunks[k].end());
                    }
                });
                if (status[w] != NOLZSS_OK) messages[w] = g_error;
            };
            std::vector<std::thread> threads;
            for (size_t w = 0; w < workers; ++w) threads.emplace_back(worker, w);
            for (auto &t : threads) t.join();
            for (size_t w = 0; w < workers; ++w)
                if (status[w] != NOLZSS_OK) rethrow_worker_error(status[w], messages[w]);
        }
        m = order.size();  // what is left takes the per-record path below
        if (m == 0) return;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return lens[a] > lens[b]; });
        static const size_t lanes_env = [] {
            const char *e = getenv("NOLZSS_BATCH_LANES");
            const long v = e ? atol(e) : 8;
            return (size_t)(v < 1 ? 1 : (v > kMaxLanes ? kMaxLanes : v));
        }();
        size_t lanes = std::min(lanes_env, m ? m : (size_t)1);
        if (m) {  // no more lanes than arenas for the longest record fit the device
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                const size_t fit = (size_t)((double)total_b * 0.85) / arena_bytes_for(lens[order[0]]);
                lanes = std::max<size_t>(1, std::min(lanes, fit));
            }
        }
        std::atomic<size_t> next{0};
        std::vector<int> status(lanes, NOLZSS_OK);
        std::vector<std::string> messages(lanes);
        auto worker = [&](size_t lane) {
            status[lane] = guarded([&] {
                Session ses(device, nullptr, (int)lane);
                for (;;) {
                    const size_t k = next.fetch_add(1);
                    if (k >= m) break;
                    const size_t j = order[k];
                    reserve_arena_for(ses.ctx(), lens[j]);
                    z[j] = run_plain(ses.ctx(), static_cast<const uint8_t *>(d_texts[j]), lens[j], 0, nullptr, nullptr,
                                     emit == 1);
                    ++g_single_records;
                }
            });
            if (status[lane] != NOLZSS_OK) messages[lane] = g_error;
        };
        std::vector<std::thread> threads;
        for (size_t lane = 0; lane < lanes; ++lane) threads.emplace_back(worker, lane);
        for (auto &t : threads) t.join();
        for (size_t lane = 0; lane < lanes; ++lane)
            if (status[lane] != NOLZSS_OK) rethrow_worker_error(status[lane], messages[lane]);
    });
}

// out[j] may point INTO a block shared by many records: only this function knows what to free
void nolzss_free_batch(nolzss_factor **out, size_t *z, size_t m) {
    (void)m;
    if (out) {
        std::vector<void *> blocks;
        {
            std::lock_guard<std::mutex> lk(g_batch_mu);
            auto it = g_batch_blocks.find(out);
            if (it != g_batch_blocks.end()) {
                blocks = std::move(it->second);
                g_batch_blocks.erase(it);
            }
        }
        for (void *b : blocks) nolzss::api::free_block(b);
        std::free(out);
    }
    std::free(z);
}

}  // extern "C"
