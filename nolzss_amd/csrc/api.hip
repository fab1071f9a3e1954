// api.hip -- C ABI of libnolzss_hip.so (include/nolzss_hip.h) and the host-side orchestration
// of the device pipeline.  There is no CPU fallback anywhere in this library.
#include "../../include/nolzss_hip.h"

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cerrno>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <sys/mman.h>
#include <fstream>
#include <iterator>
#include <memory>
#include <mutex>
#include <numeric>
#include <thread>

#include <malloc.h>

#include "pipeline.hpp"
#include "pyramid.hpp"
#include "radix_sort.hpp"
#include "scan.hpp"

namespace nolzss {

namespace {

thread_local std::string g_error;

void *alloc_factor_block(size_t bytes);  // host memory for results: huge pages for large blocks (defined with the batch worker)

int set_error(int code, const std::string &msg) {
    g_error = msg;
    return code;
}

struct DeviceContext {
    std::mutex mu;
    Context ctx;
    bool ready = false;
};

std::mutex g_ctx_mu;
std::map<int, std::unique_ptr<DeviceContext>> g_ctx;

constexpr size_t kMaxText = 0xffffffffull - (1ull << 19);  // 32-bit index pipeline (sharded queue slots stay below 2^32)

// Device memory per text symbol.  The pipeline peaks at 49 bytes per symbol (candidate
// stage) unless many suffixes are still tied after the direct round: the rounds that resolve those take 60
// bytes per tied suffix on top of 32 per symbol -- 96 when EVERY suffix is tied (a periodic text).  The arena
// asks for the worst case when the device has it and settles for what is there down to kArenaMinPerSymbol;
// a text that then needs more fails in the suffix-array rounds with a message that says so.
constexpr size_t kArenaBytesPerSymbol = 96;
constexpr size_t kArenaMinPerSymbol = 52;
constexpr size_t kArenaSlack = size_t(64) << 20;
size_t arena_bytes_for(size_t n) { return kArenaBytesPerSymbol * n + kArenaSlack; }
size_t arena_min_bytes_for(size_t n) { return kArenaMinPerSymbol * n + kArenaSlack; }

constexpr int kMaxLanes = 16;  // concurrent pipelines (stream + arena each) per device

DeviceContext &get_context(int device, int lane = 0) {
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

// RAII: lock the device context, make it current, make sure stream / pinned scratch exist
struct Session {
    DeviceContext &dc;
    std::unique_lock<std::mutex> lk;
    hipStream_t own_stream;
    Session(int device, void *user_stream, int lane = 0) : dc(get_context(device, lane)), lk(dc.mu) {
        HIP_CHECK(hipSetDevice(device));
        if (!dc.ready) {
            dc.ctx.device = device;
            HIP_CHECK(hipStreamCreateWithFlags(&dc.ctx.stream, hipStreamNonBlocking));
            HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&dc.ctx.h_pinned), 64 * sizeof(uint32_t)));
            dc.ready = true;
        }
        own_stream = dc.ctx.stream;
        if (user_stream) dc.ctx.stream = static_cast<hipStream_t>(user_stream);
    }
    ~Session() {
        dc.ctx.stream = own_stream;
        dc.ctx.arena.rewind(0);  // every call starts from an empty arena, also after an exception
    }
    Context &ctx() { return dc.ctx; }
};

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
void reserve_arena_for(Context &ctx, size_t n, size_t extra = 0) {
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

struct DebugOut {
    uint32_t *sa = nullptr, *isa = nullptr, *lcp = nullptr, *lstar = nullptr;
};

void copy_out(Context &ctx, uint32_t *host, const uint32_t *dev, size_t count) {
    if (!host || !count) return;
    HIP_CHECK(hipMemcpyAsync(host, dev, count * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx.stream));
}

// The plain-mode pipeline on a device-resident text.  Returns z; *out_host (optional) receives
// a malloc'ed array of z factors.
size_t run_plain(Context &ctx, const uint8_t *d_text, size_t n, size_t start_pos, nolzss_factor **out_host,
                 DebugOut *dbg, bool records_on_device_only = false) {
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

template <typename F> int guarded(F &&f) {
    try {
        f();
        return NOLZSS_OK;
    } catch (const HipError &e) {
        return set_error(NOLZSS_ERR_DEVICE, e.what());
    } catch (const std::bad_alloc &) {
        return set_error(NOLZSS_ERR_NOMEM, "out of host memory");
    } catch (const std::invalid_argument &e) {
        return set_error(NOLZSS_ERR_INVALID_ARGUMENT, e.what());
    } catch (const std::ios_base::failure &e) {
        return set_error(NOLZSS_ERR_IO, e.what());
    } catch (const std::exception &e) {
        return set_error(NOLZSS_ERR_RUNTIME, e.what());
    }
}

void check_text_args(const void *text, size_t n, size_t start_pos) {
    if (n && !text) throw std::invalid_argument("text pointer is null");
    if (n > kMaxText) throw std::invalid_argument("text too long: the device pipeline uses 32-bit indices");
    if (start_pos > n) throw std::invalid_argument("start_pos beyond the end of the text");
}

// A file in host memory.  Large files are read by several threads (pread of 16 MiB pieces: a single
// read() of a cached 1 GiB file takes twice as long as its factorization) into a block on transparent
// huge pages that nothing zero-fills first.
// Giving memory back costs too: free() of a 2 GiB block of 4 KiB pages spends 0.2 s in munmap (a third of the
// time from a 2.17 GB FASTA file to its 512 factor counts).  Blocks of 256 MiB and more are released by a
// detached thread; the caller does not wait for the page tables.
void free_block(void *p) {
    if (!p) return;
    if (malloc_usable_size(p) >= (size_t(256) << 20)) {
        try {
            std::thread([p] { std::free(p); }).detach();
            return;
        } catch (...) {  // no thread to be had: release it here
        }
    }
    std::free(p);
}

struct FileBytes {
    std::unique_ptr<uint8_t, void (*)(void *)> block{nullptr, &free_block};
    size_t bytes = 0;
    const uint8_t *data() const { return block.get(); }
    size_t size() const { return bytes; }
    bool empty() const { return bytes == 0; }
};

FileBytes read_file(const char *path) {
    if (!path) throw std::invalid_argument("path is null");
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) throw std::runtime_error(std::string("Cannot open input file: ") + path);
    struct Close {
        int fd;
        ~Close() { ::close(fd); }
    } closer{fd};
    struct stat st;
    FileBytes out;
    if (::fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size <= 0) {
        // not a regular file (or empty): take what comes
        std::vector<uint8_t> all;
        uint8_t buf[1 << 16];
        for (;;) {
            const ssize_t got = ::read(fd, buf, sizeof buf);
            if (got < 0 && errno == EINTR) continue;
            if (got <= 0) break;
            all.insert(all.end(), buf, buf + got);
        }
        if (!all.empty()) {
            out.block.reset(static_cast<uint8_t *>(std::malloc(all.size())));
            if (!out.block) throw std::bad_alloc();
            std::memcpy(out.block.get(), all.data(), all.size());
            out.bytes = all.size();
        }
        return out;
    }
    const size_t size = (size_t)st.st_size;
    out.block.reset(static_cast<uint8_t *>(alloc_factor_block(size)));
    if (!out.block) throw std::bad_alloc();
    constexpr size_t kPiece = size_t(16) << 20;
    const size_t pieces = div_up(size, kPiece);
    unsigned hw = std::thread::hardware_concurrency();
    const size_t threads = std::min<size_t>({pieces, hw ? hw : 1u, 8u});
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
    auto worker = [&] {
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= pieces || failed.load()) break;
            size_t at = k * kPiece;
            const size_t stop = std::min(size, at + kPiece);
            while (at < stop) {
                const ssize_t got = ::pread(fd, out.block.get() + at, stop - at, (off_t)at);
                if (got < 0 && errno == EINTR) continue;
                if (got <= 0) {  // shorter than fstat said, or an I/O error
                    failed.store(true);
                    return;
                }
                at += (size_t)got;
            }
        }
    };
    if (threads <= 1) {
        worker();
    } else {
        std::vector<std::thread> pool;
        for (size_t t = 0; t < threads; ++t) pool.emplace_back(worker);
        for (auto &t : pool) t.join();
    }
    if (failed.load()) throw std::ios_base::failure(std::string("Cannot read input file: ") + path);
    out.bytes = size;
    return out;
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
// Bytes that are all written right after the allocation: no value-initialisation (resize() of a std::vector<uint8_t>
// zeroes -- and page-faults -- half a gigabyte on one thread for a reference + target pair of 2^27 bases each).
template <typename T> struct DefaultInitAllocator : std::allocator<T> {
    template <typename U> struct rebind { using other = DefaultInitAllocator<U>; };
    using std::allocator<T>::allocator;
    template <typename U> void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void *>(p)) U; }
    template <typename U, typename... Args> void construct(U *p, Args &&...args) { ::new (static_cast<void *>(p)) U(std::forward<Args>(args)...); }
};
using HostBytes = std::vector<uint8_t, DefaultInitAllocator<uint8_t>>;

// fn(lo, hi) over [0, n) in contiguous pieces on up to 16 host threads (one piece on the caller's thread for short
// inputs): validation, case folding and reverse complement of sequences as long as the device factorizes in tens of
// milliseconds were 250 ms on one core for that pair
template <typename Fn> void host_parallel(size_t n, Fn fn) {
    constexpr size_t kMinPiece = size_t(4) << 20;
    unsigned hw = std::thread::hardware_concurrency();
    size_t pieces = std::min<size_t>({(size_t)(hw ? hw : 1u), (size_t)16, n / kMinPiece});
    if (pieces <= 1) {
        fn((size_t)0, n);
        return;
    }
    const size_t per = (n + pieces - 1) / pieces;
    std::vector<std::thread> pool;
    std::exception_ptr err;
    std::mutex mu;
    for (size_t k = 1; k < pieces; ++k)
        pool.emplace_back([&, k] {
            try {
                fn(std::min(n, k * per), std::min(n, (k + 1) * per));
            } catch (...) {
                std::lock_guard<std::mutex> g(mu);
                err = std::current_exception();
            }
        });
    try {
        fn((size_t)0, std::min(n, per));
    } catch (...) {
        std::lock_guard<std::mutex> g(mu);
        err = std::current_exception();
    }
    for (auto &t : pool) t.join();
    if (err) std::rethrow_exception(err);
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

}  // namespace
}  // namespace nolzss

using namespace nolzss;

extern "C" {

const char *nolzss_last_error(void) { return g_error.c_str(); }
const char *nolzss_version(void) { return "0.1.0+gfx950"; }
void nolzss_free(void *p) { nolzss::free_block(p); }

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

// noLZSS::factorize_dna_w_rc: one sequence; the prepared string is built on the device so that only
// the n input bytes cross PCIe.  `d_resident` (optional) is the text already in device memory: nothing is
// uploaded then and `text` is not read.
// emit 0: count; 1: records built in HBM and left there; 2: records downloaded into *out.
static void dna_w_rc_common(const uint8_t *text, const uint8_t *d_resident, size_t n, int device, void *stream, int emit,
                            nolzss_factor **out, size_t *z, int lane = 0) {
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
namespace {

#pragma pack(push, 1)
struct FileFooter {  // FactorFileFooter, /root/reference/src/cpp/factorizer.hpp:64-77
    char magic[8];
    uint64_t num_factors, num_sequences, num_sentinels, footer_size, total_length;
};
#pragma pack(pop)
static_assert(sizeof(FileFooter) == 48, "v2 footer is 48 bytes");

// records, then `extra` metadata bytes, then the footer (footer_size counts extra + 48)
void write_v2_file(const char *out_path, const nolzss_factor *f, size_t z, uint64_t num_sequences,
                   uint64_t num_sentinels, uint64_t total_length, const std::string &extra) {
    if (!out_path) throw std::invalid_argument("output path is null");
    std::ofstream os(out_path, std::ios::binary);
    if (!os) throw std::runtime_error(std::string("Cannot create output file: ") + out_path);
    if (z) os.write(reinterpret_cast<const char *>(f), (std::streamsize)(sizeof(nolzss_factor) * z));
    if (!extra.empty()) os.write(extra.data(), (std::streamsize)extra.size());
    FileFooter ft;
    std::memcpy(ft.magic, "noLZSSv2", 8);
    ft.num_factors = z;
    ft.num_sequences = num_sequences;
    ft.num_sentinels = num_sentinels;
    ft.footer_size = sizeof(FileFooter) + extra.size();
    ft.total_length = total_length;
    os.write(reinterpret_cast<const char *>(&ft), sizeof ft);
    if (!os) throw std::runtime_error(std::string("Error writing output file: ") + out_path);
}

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

}  // namespace
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

// ---- concatenated multi-sequence FASTA -----------------------------------------
namespace nolzss {
namespace {

// The records of a FASTA file: views into the buffer the file was read into (the bases are compacted
// in place, in front of the read position; nothing is copied or allocated per record).
struct SeqView {
    const char *ptr = nullptr;
    size_t len = 0;
    const char *data() const { return ptr; }
    size_t size() const { return len; }
};
struct FastaParse {
    std::vector<SeqView> sequences;
    std::vector<std::string> ids;
    std::vector<std::shared_ptr<char>> buffers;  // what the views point into
};

inline bool is_canonical_dna(char c) {
    return c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'a' || c == 'c' || c == 'g' || c == 't';
}

// restates parse_fasta_sequences_and_ids, /root/reference/src/cpp/fasta_processor.cpp:28-128
// (same records, ids, warnings and errors).  The reference reads line by line and appends base by
// base (0.2 GB/s); the device side takes 3-7 Gbases/s, so the file is read in one piece and every
// line goes through a 256-entry table: upper-case base, white space to skip, or anything else.
FastaParse parse_fasta(const char *path, bool strict) {
    if (!path) throw std::invalid_argument("path is null");
    std::ifstream file(path, std::ios::binary);
    if (!file.is_open()) throw std::runtime_error(std::string("Cannot open FASTA file: ") + path);
    std::shared_ptr<char> data;  // (no zero fill in front of the read; large files on huge pages)
    size_t data_size = 0;
    {
        file.seekg(0, std::ios::end);
        const std::streamoff len = file.tellg();
        file.seekg(0, std::ios::beg);
        if (len > 0) {
            char *raw = static_cast<char *>(alloc_factor_block((size_t)len));
            if (!raw) throw std::bad_alloc();
            data.reset(raw, [](char *q) { std::free(q); });
            file.read(raw, len);
            data_size = (size_t)file.gcount();
        } else {  // not seekable: take what comes
            const std::string all((std::istreambuf_iterator<char>(file)), std::istreambuf_iterator<char>());
            char *raw = static_cast<char *>(std::malloc(all.size() + 1));
            if (!raw) throw std::bad_alloc();
            data.reset(raw, [](char *q) { std::free(q); });
            std::memcpy(raw, all.data(), all.size());
            data_size = all.size();
        }
    }
    constexpr uint8_t kSpace = 0, kOther = 0xff;
    uint8_t kind[256];
    for (int c = 0; c < 256; ++c) kind[c] = kOther;
    for (unsigned char c : {' ', '\t', '\n', '\v', '\f', '\r'}) kind[c] = kSpace;  // std::isspace, "C" locale
    for (unsigned char c : {'A', 'C', 'G', 'T'}) kind[c] = kind[c - 'A' + 'a'] = c;
    uint8_t plain[256];  // 0 for an upper-case base: a line of those is copied as it is
    for (int c = 0; c < 256; ++c) plain[c] = 1;
    for (unsigned char c : {'A', 'C', 'G', 'T'}) plain[c] = 0;

    FastaParse res;
    res.buffers.push_back(data);
    std::string cur_id;
    // the bases of the current record are compacted to [rec, rec + cur_len): never beyond the read
    // position, since a byte of the file yields at most one base
    char *rec = data.get();
    size_t cur_len = 0;
    size_t empty_count = 0, removed = 0;
    auto finish = [&] {
        if (cur_id.empty()) return;  // (bases in front of the first header go on into the first record, as in the reference)
        if (cur_len) {
            res.sequences.push_back(SeqView{rec, cur_len});
            res.ids.push_back(cur_id);
        } else {
            fprintf(stderr, "Warning: Skipping empty sequence with ID: %s\n", cur_id.c_str());
            ++empty_count;
        }
        rec += cur_len;
        cur_len = 0;
    };
    const char *p = data.get(), *const end = p + data_size;
    while (p < end) {
        const char *nl = static_cast<const char *>(std::memchr(p, '\n', (size_t)(end - p)));
        const char *line = p;
        size_t len = (size_t)((nl ? nl : end) - p);
        p = nl ? nl + 1 : end;
        while (len && kind[(unsigned char)line[len - 1]] == kSpace) --len;  // trailing white space
        if (!len) continue;
        if (line[0] == '>') {
            finish();
            size_t start = 1;
            while (start < len && kind[(unsigned char)line[start]] == kSpace) ++start;
            size_t stop = start;
            while (stop < len && kind[(unsigned char)line[stop]] != kSpace) ++stop;
            if (start >= len) throw std::runtime_error("Empty sequence header in FASTA file");
            cur_id.assign(line + start, stop - start);
        } else {
            char *out = rec + cur_len;  // <= line
            uint8_t mixed = 0;
            for (size_t i = 0; i < len; ++i) mixed |= plain[(unsigned char)line[i]];
            if (!mixed) {  // the usual line
                if (out != line) std::memmove(out, line, len);
                cur_len += len;
                continue;
            }
            size_t k = 0;
            for (size_t i = 0; i < len; ++i) {
                const uint8_t t = kind[(unsigned char)line[i]];
                if (t == kOther) {
                    if (strict)
                        throw std::runtime_error("Invalid nucleotide '" + std::string(1, line[i]) +
                                                 "' found in sequence with ID: " + cur_id);
                    ++removed;
                } else if (t != kSpace) {
                    out[k++] = (char)t;
                }
            }
            cur_len += k;
        }
    }
    finish();
    if (empty_count) fprintf(stderr, "Warning: Skipped %zu empty sequence(s) in FASTA file\n", empty_count);
    if (!strict && removed)
        fprintf(stderr, "Warning: Removed %zu ambiguous nucleotide(s) from FASTA input\n", removed);
    if (res.sequences.empty()) throw std::runtime_error("No valid sequences found in FASTA file");
    return res;
}

// restates prepare_multiple_dna_sequences_no_rc, /root/reference/src/cpp/factorizer.cpp:199-294
void prepare_no_rc(const char *const *seqs, const size_t *lens, size_t k, HostBytes &S,
                   size_t &original_length, std::vector<uint64_t> &sentinels) {
    S.clear();
    sentinels.clear();
    original_length = 0;
    if (k == 0) return;
    size_t non_empty = 0, empty = 0, total = 0;
    for (size_t i = 0; i < k; ++i) (lens[i] ? ++non_empty : ++empty);
    if (empty)
        fprintf(stderr, "Warning: Skipping %zu empty sequence(s) in prepare_multiple_dna_sequences_no_rc\n", empty);
    if (non_empty == 0) throw std::runtime_error("All sequences are empty - cannot prepare for factorization");
    if (non_empty > 250)
        throw std::invalid_argument(
            "Too many sequences: maximum 250 sequences supported (due to sentinel character limitations)");
    for (size_t i = 0; i < k; ++i) {
        const size_t j = first_invalid_nucleotide(seqs[i], lens[i]);
        if (j < lens[i])
            throw std::runtime_error("Invalid nucleotide '" + std::string(1, seqs[i][j]) + "' found in sequence " +
                                     std::to_string(i));
    }
    for (size_t i = 0; i < k; ++i) total += lens[i];
    S.resize(total + non_empty - 1);
    size_t sidx = 0, done = 0, at = 0;
    for (size_t i = 0; i < k; ++i) {
        if (!lens[i]) continue;
        copy_upper(S.data() + at, seqs[i], lens[i]);
        at += lens[i];
        if (++done < non_empty) {  // sentinels only BETWEEN sequences (:280-288)
            sentinels.push_back(at);
            S[at++] = rc_sentinel(sidx++);
        }
    }
    original_length = at;
}

// restates identify_sentinel_factors, fasta_processor.cpp:131-163
std::vector<uint64_t> sentinel_factors(const nolzss_factor *f, size_t z, const std::vector<uint64_t> &positions) {
    std::vector<uint64_t> idx;
    size_t s = 0;
    for (size_t i = 0; i < z; ++i) {
        while (s < positions.size() && positions[s] < f[i].start) ++s;
        if (s < positions.size() && f[i].start == positions[s]) {
            if (f[i].length != 1)
                throw std::runtime_error("Sentinel factor has unexpected length: " + std::to_string(f[i].length));
            if (f[i].ref != f[i].start)
                throw std::runtime_error("Sentinel factor reference mismatch: ref=" + std::to_string(f[i].ref) +
                                         ", pos=" + std::to_string(f[i].start));
            idx.push_back(i);
            ++s;
        }
    }
    return idx;
}

struct FastaFactors {
    FastaParse parse;
    nolzss_factor *factors = nullptr;
    size_t z = 0;
    std::vector<uint64_t> sentinel_idx;
    ~FastaFactors() { std::free(factors); }
};

void factorize_fasta(const char *path, bool with_rc, bool strict, int device, FastaFactors &out) {
    out.parse = parse_fasta(path, strict);
    std::vector<const char *> ptrs;
    std::vector<size_t> lens;
    for (const auto &q : out.parse.sequences) {
        ptrs.push_back(q.data());
        lens.push_back(q.size());
    }
    HostBytes S;
    std::vector<uint64_t> sent;
    size_t orig = 0;
    if (with_rc) {
        prepare_w_rc(ptrs.data(), lens.data(), ptrs.size(), S, orig, sent);  // fasta_processor.cpp:308
        if (rc_guards(S.size(), 0)) {
            Session ses(device, nullptr);
            out.z = run_rc_host(ses.ctx(), S.data(), S.size(), 0, &out.factors);  // :311
        }
    } else {
        prepare_no_rc(ptrs.data(), lens.data(), ptrs.size(), S, orig, sent);  // :331
        check_text_args(S.data(), S.size(), 0);
        Session ses(device, nullptr);
        out.z = run_plain_host(ses.ctx(), S.data(), S.size(), 0, &out.factors, nullptr);  // :334
    }
    out.sentinel_idx = sentinel_factors(out.factors, out.z, sent);  // :314 / :337
}

// factorize_dna_rc_w_ref_fasta_files, fasta_processor.cpp:240-287, 362-378
void factorize_ref_target_fasta(const char *ref_path, const char *tgt_path, bool strict, int device,
                                FastaFactors &out) {
    FastaParse ref = parse_fasta(ref_path, strict);
    FastaParse tgt = parse_fasta(tgt_path, strict);
    size_t target_start = 0;
    for (const auto &q : ref.sequences) target_start += q.size() + 1;  // +1 for each sentinel (:249-252)
    out.parse.sequences = ref.sequences;
    out.parse.ids = ref.ids;
    out.parse.buffers = ref.buffers;
    out.parse.buffers.insert(out.parse.buffers.end(), tgt.buffers.begin(), tgt.buffers.end());
    out.parse.sequences.insert(out.parse.sequences.end(), tgt.sequences.begin(), tgt.sequences.end());
    out.parse.ids.insert(out.parse.ids.end(), tgt.ids.begin(), tgt.ids.end());
    std::vector<const char *> ptrs;
    std::vector<size_t> lens;
    for (const auto &q : out.parse.sequences) {
        ptrs.push_back(q.data());
        lens.push_back(q.size());
    }
    HostBytes S;
    std::vector<uint64_t> sent;
    size_t orig = 0;
    prepare_w_rc(ptrs.data(), lens.data(), ptrs.size(), S, orig, sent);
    if (rc_guards(S.size(), target_start)) {
        Session ses(device, nullptr);
        out.z = run_rc_host(ses.ctx(), S.data(), S.size(), target_start, &out.factors);
    }
    out.sentinel_idx = sentinel_factors(out.factors, out.z, sent);
}

void fill_fasta_result(FastaFactors &ff, nolzss_fasta_result *out) {
    std::string blob;
    for (const auto &id : ff.parse.ids) blob.append(id).push_back('\0');
    uint64_t *sidx = static_cast<uint64_t *>(std::malloc(ff.sentinel_idx.size() * sizeof(uint64_t) + 8));
    char *ids = static_cast<char *>(std::malloc(blob.size() + 1));
    if (!sidx || !ids) {
        std::free(sidx);
        std::free(ids);
        throw std::bad_alloc();
    }
    if (!ff.sentinel_idx.empty())
        std::memcpy(sidx, ff.sentinel_idx.data(), ff.sentinel_idx.size() * sizeof(uint64_t));
    std::memcpy(ids, blob.data(), blob.size());
    out->factors = ff.factors;
    ff.factors = nullptr;  // ownership moves to the caller
    out->num_factors = ff.z;
    out->sentinel_factor_indices = sidx;
    out->num_sentinels = ff.sentinel_idx.size();
    out->sequence_ids = ids;
    out->sequence_ids_bytes = blob.size();
    out->num_sequences = ff.parse.ids.size();
}

// write_fasta_metadata, parallel_fasta_processor.cpp:29-62: names, sentinel indices, footer
void write_fasta_file(const char *out_path, const FastaFactors &ff) {
    std::string extra;
    for (const auto &id : ff.parse.ids) extra.append(id).push_back('\0');
    extra.append(reinterpret_cast<const char *>(ff.sentinel_idx.data()), ff.sentinel_idx.size() * sizeof(uint64_t));
    uint64_t total = 0;
    for (size_t i = 0; i < ff.z; ++i) total += ff.factors[i].length;
    write_v2_file(out_path, ff.factors, ff.z, ff.parse.ids.size(), ff.sentinel_idx.size(), total, extra);
}

// the batch worker (defined with the merged batch further down)
void factorize_many(const uint8_t *const *texts, const size_t *lens, size_t m, const int *devices, size_t n_dev,
                    bool with_rc, size_t *zs, nolzss_factor **fs, std::vector<void *> &blocks);

}  // namespace
}  // namespace nolzss

extern "C" {

int nolzss_factorize_dna_rc_w_ref_fasta_files(const char *reference_fasta_path, const char *target_fasta_path,
                                              int sanitize_mode, int device, nolzss_fasta_result *out) {
    return guarded([&] {
        if (!out) throw std::invalid_argument("output pointer is null");
        std::memset(out, 0, sizeof *out);
        if (sanitize_mode != 0 && sanitize_mode != 1) throw std::invalid_argument("sanitize_mode must be 0 or 1");
        FastaFactors ff;
        factorize_ref_target_fasta(reference_fasta_path, target_fasta_path, sanitize_mode == 1, device, ff);
        fill_fasta_result(ff, out);
    });
}

int nolzss_write_factors_dna_w_reference_fasta_files_to_binary(const char *reference_fasta_path,
                                                               const char *target_fasta_path, const char *out_path,
                                                               int sanitize_mode, int device, size_t *z) {
    return guarded([&] {
        if (!z) throw std::invalid_argument("output pointer is null");
        *z = 0;
        if (sanitize_mode != 0 && sanitize_mode != 1) throw std::invalid_argument("sanitize_mode must be 0 or 1");
        FastaFactors ff;
        factorize_ref_target_fasta(reference_fasta_path, target_fasta_path, sanitize_mode == 1, device, ff);
        write_fasta_file(out_path, ff);
        *z = ff.z;
    });
}

int nolzss_write_factor_file(const char *out_path, const nolzss_factor *factors, size_t z, uint64_t num_sequences,
                             uint64_t num_sentinels, uint64_t total_length, const void *extra, size_t extra_len) {
    return guarded([&] {
        if (z && !factors) throw std::invalid_argument("factor array is null");
        write_v2_file(out_path, factors, z, num_sequences, num_sentinels, total_length,
                      std::string(static_cast<const char *>(extra ? extra : ""), extra ? extra_len : 0));
    });
}

int nolzss_prepare_multiple_dna_no_rc(const char *const *seqs, const size_t *lens, size_t k, uint8_t **S,
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
        prepare_no_rc(seqs, lens, k, buf, orig, sent);
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

int nolzss_factorize_fasta_multiple_dna(const char *fasta_path, int with_rc, int sanitize_mode, int device,
                                        nolzss_fasta_result *out) {
    return guarded([&] {
        if (!out) throw std::invalid_argument("output pointer is null");
        std::memset(out, 0, sizeof *out);
        if (sanitize_mode != 0 && sanitize_mode != 1) throw std::invalid_argument("sanitize_mode must be 0 or 1");
        FastaFactors ff;
        factorize_fasta(fasta_path, with_rc != 0, sanitize_mode == 1, device, ff);
        fill_fasta_result(ff, out);
    });
}

void nolzss_free_fasta_result(nolzss_fasta_result *r) {
    if (!r) return;
    std::free(r->factors);
    std::free(r->sentinel_factor_indices);
    std::free(r->sequence_ids);
    std::memset(r, 0, sizeof *r);
}

int nolzss_write_factors_binary_file_fasta_multiple_dna(const char *fasta_path, const char *out_path, int with_rc,
                                                        int sanitize_mode, int device, size_t *z) {
    return guarded([&] {
        if (!z) throw std::invalid_argument("output pointer is null");
        *z = 0;
        if (sanitize_mode != 0 && sanitize_mode != 1) throw std::invalid_argument("sanitize_mode must be 0 or 1");
        FastaFactors ff;
        factorize_fasta(fasta_path, with_rc != 0, sanitize_mode == 1, device, ff);
        write_fasta_file(out_path, ff);
        *z = ff.z;
    });
}

void nolzss_free_fasta_per_sequence_result(nolzss_fasta_per_sequence_result *r) {
    if (!r) return;
    if (r->factors)
        for (size_t j = 0; j < r->num_sequences; ++j) std::free(r->factors[j]);
    std::free(r->factors);
    std::free(r->counts);
    std::free(r->sequence_ids);
    std::memset(r, 0, sizeof *r);
}

int nolzss_factorize_fasta_per_sequence(const char *fasta_path, int with_rc, int sanitize_mode, int want_factors,
                                        const char *out_dir, int device, nolzss_fasta_per_sequence_result *out) {
    return guarded([&] {
        if (!out) throw std::invalid_argument("output pointer is null");
        std::memset(out, 0, sizeof *out);
        if (sanitize_mode != 0 && sanitize_mode != 1) throw std::invalid_argument("sanitize_mode must be 0 or 1");
        FastaParse parse = parse_fasta(fasta_path, sanitize_mode == 1);
        const size_t m = parse.sequences.size();
        if (out_dir) {  // fs::create_directories(out_dir), parallel_fasta_processor.cpp:343
            std::string cmd_path(out_dir);
            for (size_t pos = 1; pos <= cmd_path.size(); ++pos)
                if (pos == cmd_path.size() || cmd_path[pos] == '/') {
                    const std::string part = cmd_path.substr(0, pos);
                    if (!part.empty() && ::mkdir(part.c_str(), 0777) != 0 && errno != EEXIST)
                        throw std::runtime_error("Cannot create output directory: " + part);
                }
        }
        nolzss_fasta_per_sequence_result res;
        std::memset(&res, 0, sizeof res);
        res.num_sequences = m;
        res.counts = static_cast<size_t *>(std::calloc(m ? m : 1, sizeof(size_t)));
        const bool keep = want_factors != 0;
        res.factors = keep ? static_cast<nolzss_factor **>(std::calloc(m ? m : 1, sizeof(nolzss_factor *))) : nullptr;
        std::string blob;
        for (const auto &id : parse.ids) blob.append(id).push_back('\0');
        res.sequence_ids = static_cast<char *>(std::malloc(blob.size() + 1));
        if (!res.counts || (keep && !res.factors) || !res.sequence_ids) {
            nolzss_free_fasta_per_sequence_result(&res);
            throw std::bad_alloc();
        }
        std::memcpy(res.sequence_ids, blob.data(), blob.size());
        res.sequence_ids_bytes = blob.size();
        // the records go through the batch worker (short ones merged into one device run)
        std::vector<size_t> plain_z(m, 0);
        std::vector<nolzss_factor *> plain_f(m, nullptr);
        std::vector<void *> blocks;
        struct FreeBlocks {
            std::vector<void *> &b;
            ~FreeBlocks() {
                for (void *p : b) std::free(p);
            }
        } free_blocks{blocks};
        try {
            const bool need_f = keep || out_dir;
            if (m) {
                // with rc: prepare({seq}) + factorize_multiple_dna_w_rc per record (fasta_processor.cpp:446-451);
                // without: the reference strips the last base of every record (:469-471)
                std::vector<const uint8_t *> ptrs(m);
                std::vector<size_t> lens(m);
                for (size_t j = 0; j < m; ++j) {
                    ptrs[j] = reinterpret_cast<const uint8_t *>(parse.sequences[j].data());
                    lens[j] = with_rc ? parse.sequences[j].size() : parse.sequences[j].size() - 1;
                    if (lens[j] > 0) check_text_args(ptrs[j], lens[j], 0);
                }
                factorize_many(ptrs.data(), lens.data(), m, &device, 1, with_rc != 0, plain_z.data(),
                               need_f ? plain_f.data() : nullptr, blocks);
            }
            for (size_t j = 0; j < m; ++j) {
                nolzss_factor *f = nullptr;
                const size_t z = plain_z[j];
                if (need_f && z) {  // a block of the batch worker may hold many records: own copy
                    f = static_cast<nolzss_factor *>(std::malloc(sizeof(nolzss_factor) * z));
                    if (!f) throw std::bad_alloc();
                    std::memcpy(f, plain_f[j], sizeof(nolzss_factor) * z);
                }
                std::unique_ptr<nolzss_factor, decltype(&std::free)> hold(f, &std::free);
                res.counts[j] = z;
                if (out_dir) {  // write_single_sequence_factors, parallel_fasta_processor.cpp:262-290
                    std::string safe = parse.ids[j];
                    for (char &c : safe)
                        if (c == '/' || c == '\\' || c == ':' || c == '*' || c == '?' || c == '"' || c == '<' ||
                            c == '>' || c == '|' || c == ' ')
                            c = '_';
                    uint64_t total = 0;
                    for (size_t i = 0; i < z; ++i) total += f[i].length;
                    write_v2_file((std::string(out_dir) + "/" + safe + ".bin").c_str(), f, z, 1, 0, total,
                                  parse.ids[j] + std::string(1, '\0'));
                }
                if (keep) res.factors[j] = hold.release();
            }
        } catch (...) {
            nolzss_free_fasta_per_sequence_result(&res);
            throw;
        }
        *out = res;
    });
}

}  // extern "C"

namespace nolzss {
namespace {

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
struct BatchPlan {
    std::vector<std::vector<size_t>> chunks;
    std::vector<size_t> singles;
};
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

}  // namespace
}  // namespace nolzss

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
        for (void *b : blocks) nolzss::free_block(b);
        std::free(out);
    }
    std::free(z);
}

}  // extern "C"

// ---- genomics.read_nucleotide_fasta: the per-sequence FASTA batch, file in, per-record factors out ----
namespace nolzss {
namespace {

struct NucleotideFasta {
    FileBytes data;  // the records' bases are compacted in place in here
    std::vector<std::string> ids;
    std::vector<size_t> off, len;
};

[[noreturn]] void fasta_error(const std::string &msg) { throw std::runtime_error(msg); }

// restates _parse_fasta_content and the nucleotide check of read_nucleotide_fasta,
// /root/reference/src/noLZSS/genomics/fasta.py:28-76 and :110-115, for files of ASCII bytes (a file with
// other bytes is handed back to the Python reader: false).  Python's rules, kept: lines end at \n, \r\n,
// \r, \v, \f, \x1c, \x1d, \x1e (universal newlines + str.splitlines); white space (str.strip, re \s) is
// \t \n \v \f \r \x1c-\x1f and the blank; the id is the first word of the header; bases are upper-cased;
// a repeated id keeps its first place in the order and takes the LAST record's bases (dict semantics).
namespace nucfasta {
enum : uint8_t { kBase = 0, kLower = 1, kSpace = 2, kBreak = 3, kOther = 4 };
struct Kinds {
    uint8_t kind[256];
    Kinds() {
        for (int c = 0; c < 256; ++c) kind[c] = kOther;
        for (unsigned char c : {'A', 'C', 'G', 'T'}) kind[c] = kBase;
        for (int c = 'a'; c <= 'z'; ++c) kind[c] = kLower;
        for (int c : {(int)'\t', (int)' ', 0x1f}) kind[c] = kSpace;
        for (int c : {(int)'\r', (int)'\v', (int)'\f', 0x1c, 0x1d, 0x1e}) kind[c] = kBreak;  // (\n is what the scan splits at)
    }
};
const Kinds &kinds() {
    static const Kinds k;
    return k;
}
// one piece of the file: starts at the beginning of the file or at a line that starts with '>'
struct Piece {
    const uint8_t *begin = nullptr, *end = nullptr;
    std::vector<std::string> ids;   // records in order of appearance (repeats included)
    std::vector<size_t> off, len;
    bool failed = false;
    std::string error;              // "... at line " is completed with the line number in the file
    size_t error_line = 0;          // line inside the piece
    size_t lines = 0;               // lines of the piece (when it was read to its end)
};

void parse_piece(uint8_t *base, Piece &P, bool file_end_after) {
    const uint8_t *kind = kinds().kind;
    auto is_space = [&](uint8_t c) { return kind[c] == kSpace || kind[c] == kBreak || c == '\n'; };
    bool have_id = false;
    std::string cur_id;
    uint8_t *rec = const_cast<uint8_t *>(P.begin);  // the bases of the current record go to [rec, rec + cur_len), never beyond the read position
    size_t cur_len = 0, line_num = 0;
    auto store = [&] {
        P.ids.push_back(cur_id);
        P.off.push_back((size_t)(rec - base));
        P.len.push_back(cur_len);
        rec += cur_len;
        cur_len = 0;
    };
    auto fail = [&](const char *what) {
        P.failed = true;
        P.error = what;
        P.error_line = line_num;
    };
    auto one_line = [&](const uint8_t *line, size_t len) {  // a line without any line break inside
        ++line_num;
        while (len && is_space(line[len - 1])) --len;
        while (len && is_space(line[0])) ++line, --len;
        if (!len) return;
        if (line[0] == '>') {
            if (have_id) store();
            size_t start = 1;
            while (start < len && is_space(line[start])) ++start;
            if (start >= len) return fail("Empty sequence header at line ");
            size_t stop = start;
            while (stop < len && !is_space(line[stop])) ++stop;
            cur_id.assign(reinterpret_cast<const char *>(line) + start, stop - start);
            have_id = true;
            return;
        }
        if (!have_id) return fail("Sequence data before header at line ");
        uint8_t *out = rec + cur_len;  // <= line
        uint8_t mixed = 0;
        for (size_t i = 0; i < len; ++i) mixed |= kind[line[i]];
        if (!mixed) {  // the usual line: upper-case bases only
            if (out != line) std::memmove(out, line, len);
            cur_len += len;
            return;
        }
        size_t k = 0;
        for (size_t i = 0; i < len; ++i) {
            const uint8_t c = line[i], t = kind[c];
            if (t == kSpace) continue;
            out[k++] = t == kLower ? (uint8_t)(c - 32) : c;  // (anything that is not a base fails the check later)
        }
        cur_len += k;
    };
    const uint8_t *p = P.begin, *const end = P.end;
    while (p < end && !P.failed) {
        const uint8_t *nl = static_cast<const uint8_t *>(std::memchr(p, '\n', (size_t)(end - p)));
        const uint8_t *line = p;
        size_t len = (size_t)((nl ? nl : end) - p);
        p = nl ? nl + 1 : end;
        // (a piece that is not the last one ends right behind a \n: its last line has one)
        const bool terminated = nl != nullptr || !file_end_after;
        if (terminated && len && line[len - 1] == '\r') --len;  // \r\n is one line end
        uint8_t brk = 0;
        for (size_t i = 0; i < len; ++i) brk |= (uint8_t)(kind[line[i]] == kBreak);
        if (!brk) {
            one_line(line, len);
            continue;
        }
        size_t at = 0;  // (rare) other line ends inside: \r, \v, \f, \x1c-\x1e
        for (size_t i = 0; i <= len && !P.failed; ++i)
            if (i == len || kind[line[i]] == kBreak) {
                // (a break that is the last byte of the FILE ends the last line; in front of a \n it is followed by an empty line)
                if (i < len || i > at || terminated) one_line(line + at, i - at);
                at = i + 1;
            }
    }
    if (have_id && !P.failed) store();
    P.lines = line_num;
}
}  // namespace nucfasta

bool parse_nucleotide_fasta(const char *path, NucleotideFasta &res) {
    using namespace nucfasta;
    if (!path) throw std::invalid_argument("path is null");
    res.data = read_file(path);
    uint8_t *const base = const_cast<uint8_t *>(res.data.data());
    const size_t size = res.data.size();
    const uint8_t *kind = kinds().kind;

    // pieces for the host threads: cut in front of lines that start with '>' (such a line is a header
    // whatever came before it, so every piece can be read on its own)
    unsigned hw = std::thread::hardware_concurrency();
    const size_t max_threads = std::min<size_t>(hw ? hw : 1u, 16u);
    const size_t want = std::max<size_t>(1, std::min<size_t>(max_threads, size / (size_t(8) << 20)));
    std::vector<Piece> pieces;
    {
        size_t at = 0;
        for (size_t k = 1; k <= want && at < size; ++k) {
            size_t stop = size;
            if (k < want) {
                size_t from = std::max(at + 1, size / want * k);
                stop = size;
                while (from < size) {
                    const uint8_t *nl = static_cast<const uint8_t *>(std::memchr(base + from, '\n', size - from));
                    if (!nl || (size_t)(nl - base) + 1 >= size) break;
                    if (nl[1] == '>') {
                        stop = (size_t)(nl - base) + 1;
                        break;
                    }
                    from = (size_t)(nl - base) + 1;
                }
            }
            Piece P;
            P.begin = base + at;
            P.end = base + stop;
            pieces.push_back(std::move(P));
            at = stop;
        }
        if (pieces.empty()) pieces.emplace_back();
    }
    std::atomic<bool> non_ascii{false};
    auto work = [&](size_t k) {
        Piece &P = pieces[k];
        uint8_t any = 0;
        for (const uint8_t *q = P.begin; q < P.end; ++q) any |= *q;
        if (any & 0x80) {
            non_ascii.store(true);
            return;
        }
        parse_piece(base, P, k + 1 == pieces.size());
    };
    if (pieces.size() == 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (size_t k = 0; k < pieces.size(); ++k) pool.emplace_back(work, k);
        for (auto &t : pool) t.join();
    }
    if (non_ascii.load()) return false;
    {
        size_t lines_before = 0;  // (the pieces in front were read to their ends; their bytes have been compacted since)
        for (size_t k = 0; k < pieces.size(); ++k) {  // the first error in file order, with its line number in the file
            if (pieces[k].failed) fasta_error(pieces[k].error + std::to_string(lines_before + pieces[k].error_line));
            lines_before += pieces[k].lines;
        }
    }
    std::map<std::string, size_t> index;  // id -> position in res.ids (a repeated id keeps its place, takes the last record)
    for (const Piece &P : pieces)
        for (size_t j = 0; j < P.ids.size(); ++j) {
            auto it = index.find(P.ids[j]);
            if (it == index.end()) {
                index.emplace(P.ids[j], res.ids.size());
                res.ids.push_back(P.ids[j]);
                res.off.push_back(P.off[j]);
                res.len.push_back(P.len[j]);
            } else {
                res.off[it->second] = P.off[j];
                res.len[it->second] = P.len[j];
            }
        }
    if (res.ids.empty()) fasta_error("No valid sequences found in FASTA file");
    // ^[ACGT]+$ (fasta.py:112), records in order; the scans run on the host threads
    std::vector<uint8_t> bad(res.ids.size(), 0);
    {
        std::atomic<size_t> next{0};
        auto check = [&] {
            for (;;) {
                const size_t j = next.fetch_add(1);
                if (j >= res.ids.size()) break;
                const uint8_t *q = base + res.off[j];
                uint8_t b = 0;
                for (size_t i = 0; i < res.len[j]; ++i) b |= kind[q[i]];
                bad[j] = (b || !res.len[j]) ? 1 : 0;
            }
        };
        const size_t threads = std::min<size_t>(max_threads, res.ids.size());
        if (threads <= 1 || size < (size_t(8) << 20)) {
            check();
        } else {
            std::vector<std::thread> pool;
            for (size_t t = 0; t < threads; ++t) pool.emplace_back(check);
            for (auto &t : pool) t.join();
        }
    }
    for (size_t j = 0; j < res.ids.size(); ++j) {
        if (!bad[j]) continue;
        const uint8_t *q = base + res.off[j];
        bool present[128] = {false};
        for (size_t i = 0; i < res.len[j]; ++i) present[q[i] & 127] = true;
        std::string set;
        for (int c = 0; c < 128; ++c)
            if (present[c] && kind[c] != kBase) {
                if (!set.empty()) set += ", ";
                set += "'";
                set += (char)c;
                set += "'";
            }
        fasta_error("Sequence '" + res.ids[j] + "' contains invalid nucleotides: " + (set.empty() ? "set()" : "{" + set + "}"));
    }
    return true;
}

// Longest-processing-time-first bin packing, the plan every rank of a sharded job computes for itself:
// records by (length descending, index), bins by (load, index).
std::vector<size_t> lpt_owner(const std::vector<size_t> &lens, size_t bins) {
    std::vector<size_t> order(lens.size()), owner(lens.size(), 0), load(bins, 0);
    std::iota(order.begin(), order.end(), (size_t)0);
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return lens[a] > lens[b]; });
    for (size_t j : order) {
        const size_t b = (size_t)(std::min_element(load.begin(), load.end()) - load.begin());
        owner[j] = b;
        load[b] += lens[j];
    }
    return owner;
}

struct NucleotideFastaKeep {
    NucleotideFasta parse;
    std::vector<void *> blocks;
    ~NucleotideFastaKeep() {
        for (void *b : blocks) free_block(b);
    }
};

}  // namespace
}  // namespace nolzss

extern "C" {

int nolzss_read_nucleotide_fasta(const char *path, const int *devices, size_t n_dev, int want_factors,
                                 size_t shard_index, size_t shard_count, nolzss_nucleotide_fasta *out) {
    if (out) std::memset(out, 0, sizeof *out);
    bool ascii = true;
    const int rc = guarded([&] {
        if (!out) throw std::invalid_argument("output pointer is null");
        if (!devices || n_dev == 0) throw std::invalid_argument("device list is empty");
        if (shard_count == 0 || shard_index >= shard_count) throw std::invalid_argument("shard index out of range");
        std::unique_ptr<NucleotideFastaKeep> keep(new NucleotideFastaKeep);
        NucleotideFasta &P = keep->parse;
        const bool trace = getenv("NOLZSS_TRACE") != nullptr;
        const auto t_begin = std::chrono::steady_clock::now();
        auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
        ascii = parse_nucleotide_fasta(path, P);
        if (!ascii) return;
        const double t_parse = since();
        const size_t m = P.ids.size();
        const std::vector<size_t> owner = lpt_owner(P.len, shard_count);
        std::vector<const uint8_t *> texts;
        std::vector<size_t> lens, mine;
        for (size_t j = 0; j < m; ++j)
            if (owner[j] == shard_index) {
                check_text_args(P.data.data() + P.off[j], P.len[j], 0);
                mine.push_back(j);
                texts.push_back(P.data.data() + P.off[j]);
                lens.push_back(P.len[j]);
            }
        std::vector<size_t> zs(mine.size() ? mine.size() : 1, 0);
        std::vector<nolzss_factor *> fs(mine.size() ? mine.size() : 1, nullptr);
        try {
            factorize_many(texts.data(), lens.data(), mine.size(), devices, n_dev, false, zs.data(),
                           want_factors ? fs.data() : nullptr, keep->blocks);
        } catch (const std::exception &e) {  // fasta.py:121-122
            throw std::runtime_error(std::string("Failed to factorize sequences of '") + path + "': " + e.what());
        }
        if (trace)
            fprintf(stderr, "[nolzss] nucleotide fasta '%s': %zu records, read + parse %.1f ms, factorize %.1f ms\n", path, m,
                    t_parse, since() - t_parse);
        std::string blob;
        for (const auto &id : P.ids) blob.append(id).push_back('\0');
        out->sequence_ids = static_cast<char *>(std::malloc(blob.size() + 1));
        out->lengths = static_cast<size_t *>(std::calloc(m, sizeof(size_t)));
        out->counts = static_cast<size_t *>(std::calloc(m, sizeof(size_t)));
        out->owners = static_cast<size_t *>(std::calloc(m, sizeof(size_t)));
        out->factors = want_factors ? static_cast<nolzss_factor **>(std::calloc(m, sizeof(nolzss_factor *))) : nullptr;
        if (!out->sequence_ids || !out->lengths || !out->counts || !out->owners || (want_factors && !out->factors)) {
            nolzss_free_nucleotide_fasta(out);
            throw std::bad_alloc();
        }
        std::memcpy(out->sequence_ids, blob.data(), blob.size());
        out->sequence_ids_bytes = blob.size();
        out->num_sequences = m;
        for (size_t j = 0; j < m; ++j) {
            out->lengths[j] = P.len[j];
            out->owners[j] = owner[j];
        }
        for (size_t k = 0; k < mine.size(); ++k) {
            out->counts[mine[k]] = zs[k];
            if (want_factors) out->factors[mine[k]] = fs[k];
        }
        const double t_out = since();
        P.data = FileBytes{};  // the text is not needed any more; the factor blocks are
        out->keep = keep.release();
        if (trace) fprintf(stderr, "[nolzss] nucleotide fasta: results %.1f ms, text released %.1f ms\n", t_out, since());
    });
    if (rc == NOLZSS_OK && !ascii)
        return set_error(NOLZSS_ERR_UNSUPPORTED, "the file holds non-ASCII bytes: the native FASTA reader takes ASCII files only");
    return rc;
}

void nolzss_free_nucleotide_fasta(nolzss_nucleotide_fasta *r) {
    if (!r) return;
    delete static_cast<NucleotideFastaKeep *>(r->keep);
    std::free(r->sequence_ids);
    std::free(r->lengths);
    std::free(r->counts);
    std::free(r->owners);
    std::free(r->factors);
    std::memset(r, 0, sizeof *r);
}

int nolzss_debug_parse_fasta(const char *path, int sanitize_mode, char **ids, size_t *ids_bytes, char **sequences,
                             size_t *sequences_bytes, size_t *count) {
    return guarded([&] {
        if (!ids || !ids_bytes || !sequences || !sequences_bytes || !count)
            throw std::invalid_argument("output pointer is null");
        *ids = *sequences = nullptr;
        *ids_bytes = *sequences_bytes = *count = 0;
        const FastaParse parse = parse_fasta(path, sanitize_mode == 1);
        std::string a, b;
        for (const auto &id : parse.ids) a.append(id).push_back('\0');
        for (const auto &seq : parse.sequences) b.append(seq.data(), seq.size()).push_back('\0');
        char *pa = static_cast<char *>(std::malloc(a.size() + 1)), *pb = static_cast<char *>(std::malloc(b.size() + 1));
        if (!pa || !pb) {
            std::free(pa);
            std::free(pb);
            throw std::bad_alloc();
        }
        std::memcpy(pa, a.data(), a.size());
        std::memcpy(pb, b.data(), b.size());
        *ids = pa;
        *ids_bytes = a.size();
        *sequences = pb;
        *sequences_bytes = b.size();
        *count = parse.sequences.size();
    });
}

int nolzss_debug_parse_nucleotide_fasta(const char *path, char **ids, size_t *ids_bytes, char **sequences,
                                        size_t *sequences_bytes, size_t *count) {
    bool ascii = true;
    const int rc = guarded([&] {
        if (!ids || !ids_bytes || !sequences || !sequences_bytes || !count)
            throw std::invalid_argument("output pointer is null");
        *ids = *sequences = nullptr;
        *ids_bytes = *sequences_bytes = *count = 0;
        NucleotideFasta P;
        ascii = parse_nucleotide_fasta(path, P);
        if (!ascii) return;
        std::string a, b;
        for (size_t j = 0; j < P.ids.size(); ++j) {
            a.append(P.ids[j]).push_back('\0');
            b.append(reinterpret_cast<const char *>(P.data.data()) + P.off[j], P.len[j]).push_back('\0');
        }
        char *pa = static_cast<char *>(std::malloc(a.size() + 1)), *pb = static_cast<char *>(std::malloc(b.size() + 1));
        if (!pa || !pb) {
            std::free(pa);
            std::free(pb);
            throw std::bad_alloc();
        }
        std::memcpy(pa, a.data(), a.size());
        std::memcpy(pb, b.data(), b.size());
        *ids = pa;
        *ids_bytes = a.size();
        *sequences = pb;
        *sequences_bytes = b.size();
        *count = P.ids.size();
    });
    if (rc == NOLZSS_OK && !ascii)
        return set_error(NOLZSS_ERR_UNSUPPORTED, "the file holds non-ASCII bytes: the native FASTA reader takes ASCII files only");
    return rc;
}

int nolzss_debug_lpt_plan(const size_t *lens, size_t m, size_t bins, size_t *owners) {
    return guarded([&] {
        if ((m && (!lens || !owners)) || bins == 0) throw std::invalid_argument("bad plan arguments");
        const std::vector<size_t> o = lpt_owner(std::vector<size_t>(lens, lens + m), bins);
        for (size_t j = 0; j < m; ++j) owners[j] = o[j];
    });
}

int nolzss_debug_batch_plan(const size_t *lens, size_t m, size_t n_dev, int with_rc, int32_t *chunk_of, int32_t *device_of,
                            size_t *n_chunks) {
    return guarded([&] {
        if (!lens || !chunk_of || !device_of || n_dev == 0) throw std::invalid_argument("null argument");
        BatchPlan bp = plan_batch(lens, m, with_rc != 0);
        for (size_t j = 0; j < m; ++j) chunk_of[j] = device_of[j] = -1;
        for (size_t k = 0; k < bp.chunks.size(); ++k)
            for (size_t j : bp.chunks[k]) {
                if (chunk_of[j] != -1) throw std::logic_error("batch plan: a record sits in two runs");
                chunk_of[j] = (int32_t)k;
            }
        const std::vector<std::vector<size_t>> plan = lpt_plan_singles(bp.singles, lens, n_dev);
        for (size_t d = 0; d < n_dev; ++d)
            for (size_t j : plan[d]) {
                if (chunk_of[j] != -1 || device_of[j] != -1) throw std::logic_error("batch plan: a record is dealt twice");
                device_of[j] = (int32_t)d;
            }
        if (n_chunks) *n_chunks = bp.chunks.size();
    });
}

int nolzss_debug_trim_arenas(int device, size_t *released) {
    return guarded([&] {
        HIP_CHECK(hipSetDevice(device));
        const size_t r = trim_idle_arenas(device, nullptr);
        if (released) *released = r;
    });
}

void nolzss_debug_batch_counters(uint64_t *merged_records, uint64_t *single_records) {
    if (merged_records) *merged_records = g_merged_records.load();
    if (single_records) *single_records = g_single_records.load();
}

int nolzss_profile_enable(int device, int on) {
    return guarded([&] {
        Session ses(device, nullptr);
        ses.ctx().prof.enable(on != 0);
    });
}

int nolzss_profile_reset(int device) {
    return guarded([&] {
        Session ses(device, nullptr);
        ses.ctx().prof.reset();
    });
}

int nolzss_profile_report(int device, char *buf, size_t cap) {
    return guarded([&] {
        if (!buf || cap == 0) throw std::invalid_argument("buffer is null");
        Session ses(device, nullptr);
        std::string text;
        for (const auto &kv : ses.ctx().prof.stats()) {
            char line[256];
            snprintf(line, sizeof line, "%s %llu %.6f %.0f\n", kv.first.c_str(),
                     (unsigned long long)kv.second.count, kv.second.total_ms, kv.second.bytes);
            text += line;
        }
        const size_t len = std::min(text.size(), cap - 1);
        std::memcpy(buf, text.data(), len);
        buf[len] = 0;
    });
}

int nolzss_debug_arrays(const uint8_t *text, size_t n, int device, uint32_t *sa, uint32_t *isa, uint32_t *lcp,
                        uint32_t *lstar) {
    return guarded([&] {
        check_text_args(text, n, 0);
        if (n == 0) return;
        Session ses(device, nullptr);
        DebugOut dbg;
        dbg.sa = sa;
        dbg.isa = isa;
        dbg.lcp = lcp;
        dbg.lstar = lstar;
        run_plain_host(ses.ctx(), text, n, 0, nullptr, &dbg);
        HIP_CHECK(hipStreamSynchronize(ses.ctx().stream));
        if (isa)
            for (size_t i = 0; i < n; ++i) isa[i] -= 1u;  // the device array holds rank + 1
    });
}

int nolzss_debug_sort_pairs(uint64_t *keys, uint32_t *vals, size_t n, int device) {
    return guarded([&] {
        if (n == 0) return;
        if (!keys || !vals) throw std::invalid_argument("null array");
        Session ses(device, nullptr);
        Context &ctx = ses.ctx();
        ctx.arena.reserve(n * 28 + (size_t(64) << 20));
        const size_t mark = ctx.arena.mark();
        uint64_t *k[2] = {ctx.arena.alloc<uint64_t>(n), ctx.arena.alloc<uint64_t>(n)};
        uint32_t *v[2] = {ctx.arena.alloc<uint32_t>(n), ctx.arena.alloc<uint32_t>(n)};
        HIP_CHECK(hipMemcpyAsync(k[0], keys, n * 8, hipMemcpyHostToDevice, ctx.stream));
        HIP_CHECK(hipMemcpyAsync(v[0], vals, n * 4, hipMemcpyHostToDevice, ctx.stream));
        const int shifts[8] = {0, 8, 16, 24, 32, 40, 48, 56};
        const int cur = radix_sort_pairs(k, v, n, shifts, 8, ctx.arena, ctx.stream, ctx.profiler());
        HIP_CHECK(hipMemcpyAsync(keys, k[cur], n * 8, hipMemcpyDeviceToHost, ctx.stream));
        HIP_CHECK(hipMemcpyAsync(vals, v[cur], n * 4, hipMemcpyDeviceToHost, ctx.stream));
        HIP_CHECK(hipStreamSynchronize(ctx.stream));
        ctx.prof.collect();
        ctx.arena.rewind(mark);
    });
}

int nolzss_debug_arena(int device, size_t *capacity, size_t *peak) {
    return guarded([&] {
        if (!capacity || !peak) throw std::invalid_argument("output pointer is null");
        Session ses(device, nullptr);
        *capacity = ses.ctx().arena.capacity();
        *peak = ses.ctx().arena.peak();
    });
}

int nolzss_debug_scan(uint32_t *data, size_t n, int mode, int device) {
    return guarded([&] {
        if (n == 0) return;
        if (!data) throw std::invalid_argument("null array");
        Session ses(device, nullptr);
        Context &ctx = ses.ctx();
        ctx.arena.reserve(n * 8 + (size_t(64) << 20));
        const size_t mark = ctx.arena.mark();
        uint32_t *d = ctx.arena.alloc<uint32_t>(n);
        HIP_CHECK(hipMemcpyAsync(d, data, n * 4, hipMemcpyHostToDevice, ctx.stream));
        if (mode == 0)
            scan_exclusive_add_u32(d, d, n, nullptr, ctx.arena, ctx.stream);
        else
            scan_inclusive_max_u32(d, d, n, ctx.arena, ctx.stream);
        HIP_CHECK(hipMemcpyAsync(data, d, n * 4, hipMemcpyDeviceToHost, ctx.stream));
        HIP_CHECK(hipStreamSynchronize(ctx.stream));
        ctx.arena.rewind(mark);
    });
}

}  // extern "C"
