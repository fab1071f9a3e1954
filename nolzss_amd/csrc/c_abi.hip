// c_abi.hip -- device sessions, arenas, copies; the plain, reverse-complement and reference + target entry points
// (part of the C ABI layer of libnolzss_hip.so, include/nolzss_hip.h; shared declarations: api_internal.hpp)
#include "api_internal.hpp"

namespace nolzss {

namespace api {

thread_local std::string g_error;

int set_error(int code, const std::string &msg) {
    g_error = msg;
    return code;
}

std::mutex g_ctx_mu;
std::map<int, std::unique_ptr<DeviceContext>> g_ctx;

// Device memory per text symbol.  The pipeline peaks at 49 bytes per symbol (candidate
// stage) unless many suffixes are still tied after the direct round: the rounds that resolve those take 60
// bytes per tied suffix on top of 32 per symbol -- 96 when EVERY suffix is tied (a periodic text).  The arena
// asks for the worst case when the device has it and settles for what is there down to kArenaMinPerSymbol;
// a text that then needs more fails in the suffix-array rounds with a message that says so.
size_t arena_bytes_for(size_t n) { return kArenaBytesPerSymbol * n + kArenaSlack; }
size_t arena_min_bytes_for(size_t n) { return kArenaMinPerSymbol * n + kArenaSlack; }

DeviceContext &get_context(int device, int lane) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        throw HipError("no HIP device available: libnolzss_hip has no CPU fallback");
    if (device < 0 || device >= count) throw std::invalid_argument("device ordinal out of range");
    if (lane < 0 || lane >= kMaxLanes) throw std::invalid_argument("lane out of range");
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    auto &slot = g_ctx[device * kMaxLanes + lane];
    if (!slot) slot = std::make_unique<DeviceContext>();
    return *slot;
}

// Arenas are kept between calls and never shrink.  When the device runs out of memory, the idle ones
// (other lanes of the device that no call holds at the moment) are given back and the reservation is
// tried once more.
size_t trim_idle_arenas(int device, const Context *keep) {
    size_t released = 0;
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    for (auto &kv : g_ctx) {
        DeviceContext &dc = *kv.second;
        if (&dc.ctx == keep || kv.first / kMaxLanes != device) continue;
        std::unique_lock<std::mutex> idle(dc.mu, std::try_to_lock);
        if (!idle.owns_lock()) continue;
        released += dc.ctx.arena.capacity();
        dc.ctx.arena.release();
    }
    return released;
}

// The arena for a text of n symbols plus `extra` bytes (uploads): the worst-case size if the device has
// it, else as much as there is, but never less than the minimum the pipeline needs on ordinary texts --
// below that the input is refused up front, with the sizes, as an argument error (ValueError in Python)
// instead of failing late with a device out-of-memory error.
void reserve_arena_for(Context &ctx, size_t n, size_t extra) {
    const size_t want = arena_bytes_for(n) + extra, least = arena_min_bytes_for(n) + extra;
    if (want <= ctx.arena.capacity()) return;
    auto available = [&]() -> size_t {  // what a fresh reservation could get: free memory + the slab it replaces
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return ~size_t(0);
        return (size_t)((double)(free_b + ctx.arena.capacity()) * 0.97);
    };
    size_t avail = available();
    if (avail < want) {
        trim_idle_arenas(ctx.device, &ctx);
        avail = available();
    }
    if (avail < least) {
        if (least <= ctx.arena.capacity()) return;  // (what is reserved already will have to do)
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        char buf[320];
        snprintf(buf, sizeof buf,
                 "input too large for this device: a text of %zu symbols needs at least %.1f GiB of device memory "
                 "(%zu bytes per symbol), %.1f GiB are free of %.1f GiB (one MI355X takes about %.1f Gi symbols in plain "
                 "mode, half of that with reverse complement)",
                 n, (double)least / 1073741824.0, kArenaMinPerSymbol, (double)free_b / 1073741824.0,
                 (double)total_b / 1073741824.0, (double)total_b * 0.97 / (double)kArenaMinPerSymbol / 1073741824.0);
        throw std::invalid_argument(buf);
    }
    const size_t take = want <= avail ? want : avail;
    if (take <= ctx.arena.capacity()) return;
    try {
        ctx.arena.reserve(take);
        return;
    } catch (const HipError &) {
        (void)hipGetLastError();
    }
    trim_idle_arenas(ctx.device, &ctx);
    ctx.arena.reserve(least > ctx.arena.capacity() ? least : ctx.arena.capacity());
}

// pinned upload buffer of the context (kept: pinning costs more than the copy it speeds up)
uint8_t *host_stage(Context &ctx, size_t bytes) {
    if (bytes > ctx.h_stage_cap) {
        if (ctx.h_stage) (void)hipHostFree(ctx.h_stage);
        ctx.h_stage = nullptr;
        ctx.h_stage_cap = 0;
        const size_t cap = (bytes + (size_t(1) << 22)) & ~((size_t(1) << 22) - 1);
        HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&ctx.h_stage), cap));
        ctx.h_stage_cap = cap;
    }
    return ctx.h_stage;
}

// ---- device -> host copies into pageable memory ------------------------------------------------
// A plain hipMemcpy into a fresh malloc'ed block moves 1.25 GB of factor records in ~100 ms: the runtime
// stages it through pinned memory with one copying thread, which also takes every first-touch page fault of
// the block.  Large downloads are therefore staged here: a few host threads take 8 MiB chunks in turn, each
// through its own pinned buffer (host_stage), the DMA of one chunk running while the others are emptied
// (2^30 bases, host bytes in / factor array out: 265 -> 219 ms; NOLZSS_COPY_THREADS, default 4; 1 = plain
// hipMemcpyAsync).  Uploads stay plain copies: pageable host memory already goes up at 50 GB/s (1 GiB of text
// in 19-21 ms either way).
constexpr size_t kCopyChunk = size_t(8) << 20;
constexpr size_t kCopyThreshold = size_t(32) << 20;
int copy_threads() {
    static const int t = [] {
        const char *e = getenv("NOLZSS_COPY_THREADS");
        const long v = e ? atol(e) : 4;
        return (int)(v < 1 ? 1 : (v > 16 ? 16 : v));
    }();
    return t;
}

// host -> device, ordered on ctx.stream
void upload_bytes(Context &ctx, void *d_dst, const void *h_src, size_t n) {
    HIP_CHECK(hipMemcpyAsync(d_dst, h_src, n, hipMemcpyHostToDevice, ctx.stream));
}

// device -> host, behind everything queued on ctx.stream; the bytes have arrived when this returns
void download_bytes(Context &ctx, void *h_dst, const void *d_src, size_t n) {
    const int T = copy_threads();
    if (n < kCopyThreshold || T <= 1) {
        HIP_CHECK(hipMemcpyAsync(h_dst, d_src, n, hipMemcpyDeviceToHost, ctx.stream));
        HIP_CHECK(hipStreamSynchronize(ctx.stream));
        return;
    }
    uint8_t *ring = host_stage(ctx, (size_t)T * kCopyChunk);
    const size_t chunks = div_up(n, kCopyChunk);
    std::atomic<size_t> next{0};
    std::vector<hipError_t> err((size_t)T, hipSuccess);
    auto work = [&](int t) {
        hipError_t e = hipSetDevice(ctx.device);
        hipEvent_t ev = nullptr;
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        uint8_t *buf = ring + (size_t)t * kCopyChunk;
        while (e == hipSuccess) {
            const size_t c = next.fetch_add(1);
            if (c >= chunks) break;
            const size_t off = c * kCopyChunk, len = n - off < kCopyChunk ? n - off : kCopyChunk;
            e = hipMemcpyAsync(buf, static_cast<const uint8_t *>(d_src) + off, len, hipMemcpyDeviceToHost, ctx.stream);
            if (e == hipSuccess) e = hipEventRecord(ev, ctx.stream);
            if (e == hipSuccess) e = hipEventSynchronize(ev);
            if (e == hipSuccess) std::memcpy(static_cast<uint8_t *>(h_dst) + off, buf, len);
        }
        if (ev) (void)hipEventDestroy(ev);
        err[(size_t)t] = e;
    };
    std::vector<std::thread> threads;
    for (int t = 1; t < T; ++t) threads.emplace_back(work, t);
    work(0);
    for (auto &th : threads) th.join();
    for (hipError_t e : err) HIP_CHECK(e);
}

// The library's own stream is non-blocking: order it behind the work already queued on the legacy default
// stream (where torch's default stream and plain hipMemcpyAsync(.., 0) producers of a device-resident text
// run).  Producers on other non-blocking streams must be synchronised by the caller or pass their stream.
void order_behind_default_stream(Context &ctx) {
    hipEvent_t ev = nullptr;
    HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, nullptr);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx.stream, ev, 0);
    (void)hipEventDestroy(ev);
    HIP_CHECK(e);
}

void copy_out(Context &ctx, uint32_t *host, const uint32_t *dev, size_t count) {
    if (!host || !count) return;
    HIP_CHECK(hipMemcpyAsync(host, dev, count * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx.stream));
}

// The plain-mode pipeline on a device-resident text.  Returns z; *out_host (optional) receives
// a malloc'ed array of z factors.
size_t run_plain(Context &ctx, const uint8_t *d_text, size_t n, size_t start_pos, nolzss_factor **out_host,
                 DebugOut *dbg, bool records_on_device_only) {
    if (out_host) *out_host = nullptr;
    if (n == 0 || start_pos >= n) return 0;
    Arena &arena = ctx.arena;
    const size_t mark = arena.mark();
    hipStream_t s = ctx.stream;

    PackedText text = pack_text(ctx, d_text, n);
    uint32_t *sa = arena.alloc<uint32_t>(n);
    uint32_t *isa = arena.alloc<uint32_t>(n);
    uint32_t *lcp = arena.alloc<uint32_t>(n + 1);
    bool isa_deferred = false;
    build_suffix_array(ctx, text, sa, isa, lcp, &isa_deferred);
    // (the pyramids are only allocated here: the candidate kernel writes their first level from the blocks it holds
    // in LDS anyway, build_lstar fills the rest)
    const Pyramid Psa = alloc_pyramid(sa, (uint32_t)n, arena), Plcp = alloc_pyramid(lcp, (uint32_t)n + 1, arena);
    uint32_t *lstar = arena.alloc<uint32_t>(n);
    build_lstar(ctx, (uint32_t)n, sa, isa, lcp, Psa, Plcp, lstar, isa_deferred ? isa : nullptr, &text);
    if (dbg) {
        copy_out(ctx, dbg->sa, sa, n);
        copy_out(ctx, dbg->isa, isa, n);  // (1-based on the device; nolzss_debug_arrays subtracts the one)
        copy_out(ctx, dbg->lcp, lcp, n + 1);
        copy_out(ctx, dbg->lstar, lstar, n);
    }
    void *d_recs = nullptr;
    const uint32_t z = resolve_chain(ctx, (uint32_t)n, (uint32_t)start_pos, lstar, sa, isa, lcp, Psa, Plcp,
                                     (out_host || records_on_device_only) ? &d_recs : nullptr);
    if (out_host && z) {
        nolzss_factor *h = static_cast<nolzss_factor *>(alloc_factor_block(sizeof(nolzss_factor) * (size_t)z));
        if (!h) throw std::bad_alloc();
        ProfScope ps(ctx.profiler(), "factors_d2h", s);
        try {
            download_bytes(ctx, h, d_recs, sizeof(nolzss_factor) * (size_t)z);
        } catch (...) {
            std::free(h);
            throw;
        }
        *out_host = h;
    }
    {
        const hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess) {  // a late device error: the caller gets an error status and no live pointer
            if (out_host && *out_host) {
                std::free(*out_host);
                *out_host = nullptr;
            }
            HIP_CHECK(e);
        }
    }
    ctx.prof.collect();
    arena.rewind(mark);
    return z;
}

size_t run_plain_host(Context &ctx, const uint8_t *text, size_t n, size_t start_pos, nolzss_factor **out,
                      DebugOut *dbg) {
    if (out) *out = nullptr;
    if (n == 0 || start_pos >= n) return 0;
    reserve_arena_for(ctx, n, n);
    const size_t mark = ctx.arena.mark();
    uint8_t *d_text = ctx.arena.alloc<uint8_t>(n);
    {
        ProfScope ps(ctx.profiler(), "text_h2d", ctx.stream);
        upload_bytes(ctx, d_text, text, n);
    }
    size_t z;
    try {
        z = run_plain(ctx, d_text, n, start_pos, out, dbg);
    } catch (...) {
        ctx.arena.rewind(mark);
        throw;
    }
    ctx.arena.rewind(mark);
    return z;
}

void check_text_args(const void *text, size_t n, size_t start_pos) {
    if (n && !text) throw std::invalid_argument("text pointer is null");
    if (n > kMaxText) throw std::invalid_argument("text too long: the device pipeline uses 32-bit indices");
    if (start_pos > n) throw std::invalid_argument("start_pos beyond the end of the text");
}

// ---- reverse-complement preparation (host side, O(n)) ---------------------------------------
// restates prepare_multiple_dna_sequences_w_rc, /root/reference/src/cpp/factorizer.cpp:54-172
uint8_t rc_sentinel(size_t index) {  // factorizer.cpp:110-125
    uint8_t s = 1;
    size_t count = 0;
    for (;;) {
        if (s != 0 && s != 'A' && s != 'C' && s != 'G' && s != 'T') {
            if (count == index) return s;
            ++count;
        }
        ++s;
        if (s == 0) s = 1;
    }
}

inline uint8_t upper_base(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 'a' + 'A') : c; }

inline uint8_t complement_base(uint8_t c) {  // factorizer.cpp:17-27
    switch (c) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    default: return 'A';  // 'T' (input validated before)
    }
}

// Bulk helpers of the two prepare functions (the strings they build are as long as the texts the device
// factorizes at several Gbases/s: no per-byte push_back, no per-byte chain of comparisons).
struct DnaTables {
    uint8_t invalid[256], comp[256];  // invalid: 1 unless [ACGTacgt]; comp: complement of the upper-cased base
    DnaTables() {
        for (int c = 0; c < 256; ++c) {
            invalid[c] = 1;
            comp[c] = 0;
        }
        const char *b = "ACGT", *r = "TGCA";
        for (int i = 0; i < 4; ++i) {
            invalid[(unsigned char)b[i]] = invalid[(unsigned char)(b[i] + 32)] = 0;
            comp[(unsigned char)b[i]] = comp[(unsigned char)(b[i] + 32)] = (uint8_t)r[i];
        }
    }
};
const DnaTables &dna_tables() {
    static const DnaTables t;
    return t;
}

// index of the first byte that is not a nucleotide, n if there is none
size_t first_invalid_nucleotide(const char *s, size_t n) {
    const DnaTables &t = dna_tables();
    std::atomic<size_t> first{n};
    host_parallel(n, [&](size_t lo, size_t hi) {
        constexpr size_t kBlock = 4096;
        for (size_t at = lo; at < hi && at < first.load(std::memory_order_relaxed); at += kBlock) {
            const size_t stop = std::min(hi, at + kBlock);
            uint8_t bad = 0;
            for (size_t j = at; j < stop; ++j) bad |= t.invalid[(unsigned char)s[j]];
            if (bad)
                for (size_t j = at; j < stop; ++j)
                    if (t.invalid[(unsigned char)s[j]]) {
                        size_t cur = first.load();
                        while (j < cur && !first.compare_exchange_weak(cur, j)) {
                        }
                        return;
                    }
        }
    });
    return first.load();
}
// dst = upper(src) for validated nucleotides (clearing bit 5 turns acgt into ACGT)
void copy_upper(uint8_t *dst, const char *src, size_t n) {
    host_parallel(n, [&](size_t lo, size_t hi) {
        for (size_t j = lo; j < hi; ++j) dst[j] = (uint8_t)src[j] & 0xdfu;
    });
}
void copy_reverse_complement(uint8_t *dst, const char *src, size_t n) {
    const DnaTables &t = dna_tables();
    host_parallel(n, [&](size_t lo, size_t hi) {
        for (size_t j = lo; j < hi; ++j) dst[j] = t.comp[(unsigned char)src[n - 1 - j]];
    });
}

void prepare_w_rc(const char *const *seqs, const size_t *lens, size_t k, HostBytes &S,
                  size_t &original_length, std::vector<uint64_t> &sentinels) {
    S.clear();
    sentinels.clear();
    original_length = 0;
    if (k == 0) return;  // :55-57
    size_t non_empty = 0, empty = 0, total = 0;
    for (size_t i = 0; i < k; ++i) (lens[i] ? ++non_empty : ++empty);
    if (empty)  // :70-72
        fprintf(stderr, "Warning: Skipping %zu empty sequence(s) in prepare_multiple_dna_sequences_w_rc\n", empty);
    if (non_empty == 0) throw std::runtime_error("All sequences are empty - cannot prepare for factorization");
    if (non_empty > 125)
        throw std::invalid_argument(
            "Too many sequences: maximum 125 sequences supported (due to sentinel character limitations)");
    for (size_t i = 0; i < k; ++i) {
        const size_t j = first_invalid_nucleotide(seqs[i], lens[i]);
        if (j < lens[i])
            throw std::runtime_error("Invalid nucleotide '" + std::string(1, seqs[i][j]) + "' found in sequence " +
                                     std::to_string(i));
    }
    for (size_t i = 0; i < k; ++i) total += 2 * lens[i];
    total += 2 * non_empty;
    S.resize(total);
    size_t sidx = 0, at = 0;
    for (size_t i = 0; i < k; ++i) {  // :128-147
        if (!lens[i]) continue;
        copy_upper(S.data() + at, seqs[i], lens[i]);
        at += lens[i];
        sentinels.push_back(at);
        S[at++] = rc_sentinel(sidx++);
    }
    original_length = at;
    for (size_t i = k; i-- > 0;) {  // :150-169
        if (!lens[i]) continue;
        copy_reverse_complement(S.data() + at, seqs[i], lens[i]);
        at += lens[i];
        sentinels.push_back(at);
        S[at++] = rc_sentinel(sidx++);
    }
}

// guards of detail::nolzss_multiple_dna_w_rc, factorizer_core.hpp:180-205
// returns false when the reference returns 0 factors without building anything
bool rc_guards(size_t S_len, size_t start_pos) {
    if (S_len == 0) return false;
    if (S_len < 4) {
        fprintf(stderr,
                "Warning: Input string too short for factorization with reverse complement (size=%zu). "
                "Returning 0 factors.\n",
                S_len);
        return false;
    }
    const size_t N = S_len / 2 - 1;
    if (N == 0) return false;
    if (start_pos >= N) throw std::invalid_argument("start_pos must be less than the original sequence length");
    return true;
}

size_t run_rc_host(Context &ctx, const uint8_t *S, size_t m, size_t start_pos, nolzss_factor **out) {
    if (out) *out = nullptr;
    if (m > kMaxText) throw std::invalid_argument("text too long: the device pipeline uses 32-bit indices");
    if (!rc_guards(m, start_pos)) return 0;
    reserve_arena_for(ctx, m, m);
    const size_t mark = ctx.arena.mark();
    size_t z = 0;
    try {
        uint8_t *d_S = ctx.arena.alloc<uint8_t>(m);
        upload_bytes(ctx, d_S, S, m);
        void *d_recs = nullptr;
        z = run_rc_pipeline(ctx, d_S, m, start_pos, out ? &d_recs : nullptr);
        if (out && z) {
            nolzss_factor *h = static_cast<nolzss_factor *>(alloc_factor_block(sizeof(nolzss_factor) * z));
            if (!h) throw std::bad_alloc();
            *out = h;  // (freed below if the download fails)
            download_bytes(ctx, h, d_recs, sizeof(nolzss_factor) * z);
        }
        HIP_CHECK(hipStreamSynchronize(ctx.stream));
        ctx.prof.collect();
    } catch (...) {
        ctx.arena.rewind(mark);
        if (out && *out) {
            std::free(*out);
            *out = nullptr;
        }
        throw;
    }
    ctx.arena.rewind(mark);
    return z;
}

}  // namespace api
}  // namespace nolzss

namespace nolzss {
namespace api {

// noLZSS::factorize_dna_w_rc: one sequence; the prepared string is built on the device so that only
// the n input bytes cross PCIe.  `d_resident` (optional) is the text already in device memory: nothing is
// uploaded then and `text` is not read.
// emit 0: count; 1: records built in HBM and left there; 2: records downloaded into *out.
void dna_w_rc_common(const uint8_t *text, const uint8_t *d_resident, size_t n, int device, void *stream, int emit,
                            nolzss_factor **out, size_t *z, int lane) {
    *z = 0;
    if (out) *out = nullptr;
    if (n == 0) return;  // factorizer_core.hpp:143
    if (!text && !d_resident) throw std::invalid_argument("text pointer is null");
    const size_t m = 2 * n + 2;
    if (m > kMaxText) throw std::invalid_argument("text too long: the device pipeline uses 32-bit indices");
    if (!rc_guards(m, 0)) return;
    Session ses(device, stream, lane);
    Context &ctx = ses.ctx();
    if (d_resident && !stream) order_behind_default_stream(ctx);
    reserve_arena_for(ctx, m, m + (d_resident ? 0 : n));
    const uint8_t *d_T = d_resident;
    if (!d_resident) {
        uint8_t *up = ctx.arena.alloc<uint8_t>(n);
        upload_bytes(ctx, up, text, n);
        d_T = up;
    }
    uint8_t *d_S = ctx.arena.alloc<uint8_t>(m);
    const uint32_t bad = prepare_single_rc_on_device(ctx, d_T, (uint32_t)n, d_S);
    if (bad != 0xffffffffu) {  // factorizer.cpp:86-95
        uint8_t c = 0;
        if (text) c = text[bad];
        else HIP_CHECK(hipMemcpy(&c, d_T + bad, 1, hipMemcpyDeviceToHost));
        throw std::runtime_error("Invalid nucleotide '" + std::string(1, (char)c) + "' found in sequence 0");
    }
    void *d_recs = nullptr;
    const size_t count = run_rc_pipeline(ctx, d_S, m, 0, emit ? &d_recs : nullptr);
    if (emit == 2 && count) {
        nolzss_factor *h = static_cast<nolzss_factor *>(alloc_factor_block(sizeof(nolzss_factor) * count));
        if (!h) throw std::bad_alloc();
        try {
            download_bytes(ctx, h, d_recs, sizeof(nolzss_factor) * count);
        } catch (...) {
            std::free(h);
            throw;
        }
        *out = h;
    }
    hipError_t e = hipStreamSynchronize(ctx.stream);
    if (e != hipSuccess) {
        if (out && *out) {
            std::free(*out);
            *out = nullptr;
        }
        HIP_CHECK(e);
    }
    ctx.prof.collect();
    *z = count;
}

}  // namespace api
}  // namespace nolzss

using namespace nolzss;
using namespace nolzss::api;

extern "C" {

const char *nolzss_last_error(void) { return g_error.c_str(); }
const char *nolzss_version(void) { return "0.1.0+gfx950"; }
void nolzss_free(void *p) { nolzss::api::free_block(p); }

int nolzss_device_count(int *count) {
    if (!count) return set_error(NOLZSS_ERR_INVALID_ARGUMENT, "count is null");
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
    *count = c;
    return NOLZSS_OK;
}

int nolzss_factorize(const uint8_t *text, size_t n, size_t start_pos, int device, nolzss_factor **out,
                     size_t *z) {
    return guarded([&] {
        if (!out || !z) throw std::invalid_argument("output pointer is null");
        *out = nullptr;
        *z = 0;
        check_text_args(text, n, start_pos);
        Session ses(device, nullptr);
        *z = run_plain_host(ses.ctx(), text, n, start_pos, out, nullptr);
    });
}

int nolzss_count_factors(const uint8_t *text, size_t n, size_t start_pos, int device, size_t *z) {
    return guarded([&] {
        if (!z) throw std::invalid_argument("output pointer is null");
        *z = 0;
        check_text_args(text, n, start_pos);
        Session ses(device, nullptr);
        *z = run_plain_host(ses.ctx(), text, n, start_pos, nullptr, nullptr);
    });
}

int nolzss_factorize_file(const char *path, size_t start_pos, int device, nolzss_factor **out, size_t *z) {
    return guarded([&] {
        if (!out || !z) throw std::invalid_argument("output pointer is null");
        *out = nullptr;
        *z = 0;
        const FileBytes data = read_file(path);
        check_text_args(data.data(), data.size(), start_pos);
        Session ses(device, nullptr);
        *z = run_plain_host(ses.ctx(), data.data(), data.size(), start_pos, out, nullptr);
    });
}

int nolzss_count_factors_file(const char *path, size_t start_pos, int device, size_t *z) {
    return guarded([&] {
        if (!z) throw std::invalid_argument("output pointer is null");
        *z = 0;
        const FileBytes data = read_file(path);
        check_text_args(data.data(), data.size(), start_pos);
        Session ses(device, nullptr);
        *z = run_plain_host(ses.ctx(), data.data(), data.size(), start_pos, nullptr, nullptr);
    });
}

int nolzss_factorize_device(const void *d_text, size_t n, size_t start_pos, int device, void *stream,
                            int emit, nolzss_factor **out_host, size_t *z) {
    return guarded([&] {
        if (!z) throw std::invalid_argument("output pointer is null");
        *z = 0;
        if (out_host) *out_host = nullptr;
        if (emit < 0 || emit > 2) throw std::invalid_argument("emit must be 0, 1 or 2");
        if (emit == 2 && !out_host) throw std::invalid_argument("emit = 2 needs out_host");
        check_text_args(d_text, n, start_pos);
        Session ses(device, stream);
        if (!stream) order_behind_default_stream(ses.ctx());
        reserve_arena_for(ses.ctx(), n);
        *z = run_plain(ses.ctx(), static_cast<const uint8_t *>(d_text), n, start_pos,
                       emit == 2 ? out_host : nullptr, nullptr, emit == 1);
    });
}

int nolzss_prepare_multiple_dna_w_rc(const char *const *seqs, const size_t *lens, size_t k, uint8_t **S,
                                     size_t *S_len, size_t *original_length, uint64_t **sentinel_positions,
                                     size_t *n_sentinels) {
    return guarded([&] {
        if (!S || !S_len || !original_length || !sentinel_positions || !n_sentinels)
            throw std::invalid_argument("output pointer is null");
        *S = nullptr;
        *sentinel_positions = nullptr;
        *S_len = *original_length = *n_sentinels = 0;
        if (k && (!seqs || !lens)) throw std::invalid_argument("sequence array is null");
        HostBytes buf;
        std::vector<uint64_t> sent;
        size_t orig = 0;
        prepare_w_rc(seqs, lens, k, buf, orig, sent);
        uint8_t *s = static_cast<uint8_t *>(std::malloc(buf.size() ? buf.size() : 1));
        uint64_t *p = static_cast<uint64_t *>(std::malloc(sent.size() ? sent.size() * sizeof(uint64_t) : 8));
        if (!s || !p) {
            std::free(s);
            std::free(p);
            throw std::bad_alloc();
        }
        if (!buf.empty()) std::memcpy(s, buf.data(), buf.size());
        if (!sent.empty()) std::memcpy(p, sent.data(), sent.size() * sizeof(uint64_t));
        *S = s;
        *S_len = buf.size();
        *original_length = orig;
        *sentinel_positions = p;
        *n_sentinels = sent.size();
    });
}

int nolzss_factorize_multiple_dna_w_rc(const uint8_t *S, size_t S_len, size_t start_pos, int device,
                                       nolzss_factor **out, size_t *z) {
    return guarded([&] {
        if (!out || !z) throw std::invalid_argument("output pointer is null");
        *out = nullptr;
        *z = 0;
        if (S_len && !S) throw std::invalid_argument("text pointer is null");
        if (!rc_guards(S_len, start_pos)) return;
        Session ses(device, nullptr);
        *z = run_rc_host(ses.ctx(), S, S_len, start_pos, out);
    });
}

int nolzss_count_factors_multiple_dna_w_rc(const uint8_t *S, size_t S_len, size_t start_pos, int device,
                                           size_t *z) {
    return guarded([&] {
        if (!z) throw std::invalid_argument("output pointer is null");
        *z = 0;
        if (S_len && !S) throw std::invalid_argument("text pointer is null");
        if (!rc_guards(S_len, start_pos)) return;
        Session ses(device, nullptr);
        *z = run_rc_host(ses.ctx(), S, S_len, start_pos, nullptr);
    });
}

int nolzss_factorize_dna_w_rc(const uint8_t *text, size_t n, int device, nolzss_factor **out, size_t *z) {
    return guarded([&] {
        if (!out || !z) throw std::invalid_argument("output pointer is null");
        dna_w_rc_common(text, nullptr, n, device, nullptr, 2, out, z);
    });
}

int nolzss_count_factors_dna_w_rc(const uint8_t *text, size_t n, int device, size_t *z) {
    return guarded([&] {
        if (!z) throw std::invalid_argument("output pointer is null");
        dna_w_rc_common(text, nullptr, n, device, nullptr, 0, nullptr, z);
    });
}

int nolzss_factorize_dna_w_rc_device(const void *d_text, size_t n, int device, void *stream, int emit,
                                     nolzss_factor **out_host, size_t *z) {
    return guarded([&] {
        if (!z) throw std::invalid_argument("output pointer is null");
        if (emit < 0 || emit > 2) throw std::invalid_argument("emit must be 0, 1 or 2");
        if (emit == 2 && !out_host) throw std::invalid_argument("emit = 2 needs out_host");
        if (n && !d_text) throw std::invalid_argument("text pointer is null");
        dna_w_rc_common(nullptr, static_cast<const uint8_t *>(d_text), n, device, stream, emit, out_host, z);
    });
}

}  // extern "C"

// ---- reference + target factorization and v2 binary files ("next" rows) ---------
namespace nolzss {
namespace api {

size_t w_reference(const uint8_t *ref, size_t ref_len, const uint8_t *tgt, size_t tgt_len, int device,
                   nolzss_factor **out) {
    if ((ref_len && !ref) || (tgt_len && !tgt)) throw std::invalid_argument("sequence pointer is null");
    std::vector<uint8_t> combined;  // factorizer.cpp:942: reference + '\x01' + target
    combined.reserve(ref_len + tgt_len + 1);
    combined.insert(combined.end(), ref, ref + ref_len);
    combined.push_back(1);
    combined.insert(combined.end(), tgt, tgt + tgt_len);
    check_text_args(combined.data(), combined.size(), ref_len + 1);
    Session ses(device, nullptr);
    return run_plain_host(ses.ctx(), combined.data(), combined.size(), ref_len + 1, out, nullptr);
}

size_t dna_w_reference(const char *ref, size_t ref_len, const char *tgt, size_t tgt_len, int device,
                       nolzss_factor **out) {
    if ((ref_len && !ref) || (tgt_len && !tgt)) throw std::invalid_argument("sequence pointer is null");
    const char *seqs[2] = {ref, tgt};
    const size_t lens[2] = {ref_len, tgt_len};
    HostBytes S;
    std::vector<uint64_t> sent;
    size_t orig = 0;
    prepare_w_rc(seqs, lens, 2, S, orig, sent);  // factorizer.cpp:827-828
    const size_t start = ref_len + 1;            // :833
    if (!rc_guards(S.size(), start)) return 0;
    Session ses(device, nullptr);
    return run_rc_host(ses.ctx(), S.data(), S.size(), start, out);
}

}  // namespace api
}  // namespace nolzss

extern "C" {

int nolzss_factorize_w_reference(const uint8_t *reference_seq, size_t reference_len, const uint8_t *target_seq,
                                 size_t target_len, int device, nolzss_factor **out, size_t *z) {
    return guarded([&] {
        if (!out || !z) throw std::invalid_argument("output pointer is null");
        *out = nullptr;
        *z = w_reference(reference_seq, reference_len, target_seq, target_len, device, out);
    });
}

int nolzss_factorize_dna_w_reference_seq(const char *reference_seq, size_t reference_len, const char *target_seq,
                                         size_t target_len, int device, nolzss_factor **out, size_t *z) {
    return guarded([&] {
        if (!out || !z) throw std::invalid_argument("output pointer is null");
        *out = nullptr;
        *z = dna_w_reference(reference_seq, reference_len, target_seq, target_len, device, out);
    });
}

int nolzss_write_factors_binary_file(const char *in_path, const char *out_path, int device, size_t *z) {
    return guarded([&] {
        if (!z) throw std::invalid_argument("output pointer is null");
        *z = 0;
        const FileBytes data = read_file(in_path);
        check_text_args(data.data(), data.size(), 0);
        nolzss_factor *f = nullptr;
        size_t count;
        {
            Session ses(device, nullptr);
            count = run_plain_host(ses.ctx(), data.data(), data.size(), 0, &f, nullptr);
        }
        std::unique_ptr<nolzss_factor, decltype(&std::free)> hold(f, &std::free);
        write_v2_file(out_path, f, count, 0, 0, data.size(), std::string());  // factorizer.cpp:447-456
        *z = count;
    });
}

int nolzss_write_factors_binary_file_dna_w_rc(const char *in_path, const char *out_path, int device, size_t *z) {
    return guarded([&] {
        if (!z) throw std::invalid_argument("output pointer is null");
        *z = 0;
        const FileBytes data = read_file(in_path);
        nolzss_factor *f = nullptr;
        size_t count = 0;
        dna_w_rc_common(data.data(), nullptr, data.size(), device, nullptr, 2, &f, &count);
        std::unique_ptr<nolzss_factor, decltype(&std::free)> hold(f, &std::free);
        // one empty sequence name, num_sequences = 1 (factorizer.cpp:621-629)
        write_v2_file(out_path, f, count, 1, 0, data.size(), std::string(1, '\0'));
        *z = count;
    });
}

int nolzss_factorize_w_reference_file(const uint8_t *reference_seq, size_t reference_len, const uint8_t *target_seq,
                                      size_t target_len, const char *out_path, int device, size_t *z) {
    return guarded([&] {
        if (!z) throw std::invalid_argument("output pointer is null");
        *z = 0;
        if (!out_path) throw std::invalid_argument("output path is null");
        {  // the reference opens the output first (factorizer.cpp:982-985)
            std::ofstream probe(out_path, std::ios::binary);
            if (!probe) throw std::runtime_error(std::string("Cannot create output file: ") + out_path);
        }
        nolzss_factor *f = nullptr;
        const size_t count = w_reference(reference_seq, reference_len, target_seq, target_len, device, &f);
        std::unique_ptr<nolzss_factor, decltype(&std::free)> hold(f, &std::free);
        write_v2_file(out_path, f, count, 2, 1, target_len, std::string());  // :1005-1015
        *z = count;
    });
}

int nolzss_factorize_dna_w_reference_seq_file(const char *reference_seq, size_t reference_len,
                                              const char *target_seq, size_t target_len, const char *out_path,
                                              int device, size_t *z) {
    return guarded([&] {
        if (!z) throw std::invalid_argument("output pointer is null");
        *z = 0;
        if (!out_path) throw std::invalid_argument("output path is null");
        {
            std::ofstream probe(out_path, std::ios::binary);
            if (!probe) throw std::runtime_error(std::string("Cannot create output file: ") + out_path);
        }
        nolzss_factor *f = nullptr;
        const size_t count = dna_w_reference(reference_seq, reference_len, target_seq, target_len, device, &f);
        std::unique_ptr<nolzss_factor, decltype(&std::free)> hold(f, &std::free);
        write_v2_file(out_path, f, count, 2, 1, target_len, std::string());  // factorizer.cpp:866-876
        *z = count;
    });
}

}  // extern "C"
