// common.hpp -- shared host/device helpers for the gfx950 noLZSS pipeline.
//
// Everything in csrc/ is written for CDNA4 only: 64-lane wavefronts, 160 KiB LDS per CU,
// 256 CUs in 8 XCDs.  No other target is supported.
#pragma once

#include <hip/hip_runtime.h>

#include "divup.hpp"

#include <cstdint>
#include <cstdio>
#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace nolzss {

constexpr int kWave = 64;

struct HipError : std::runtime_error {
    explicit HipError(const std::string &m) : std::runtime_error(m) {}
};

inline void hip_check(hipError_t e, const char *what, const char *file, int line) {
    if (e != hipSuccess) {
        char buf[512];
        snprintf(buf, sizeof buf, "HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file,
                 line, what);
        throw HipError(buf);
    }
}
#define HIP_CHECK(x) ::nolzss::hip_check((x), #x, __FILE__, __LINE__)
#define KERNEL_CHECK() ::nolzss::hip_check(hipGetLastError(), "kernel launch", __FILE__, __LINE__)


// Bump allocator over one device slab.  The pipeline sizes the slab once from n (all arrays
// are 32-bit index arrays over the text, so the footprint is a small multiple of n) and
// carves stage buffers out of it; mark()/rewind() free stage temporaries in LIFO order.
class Arena {
  public:
    Arena() = default;
    ~Arena() { release(); }
    Arena(const Arena &) = delete;
    Arena &operator=(const Arena &) = delete;

    void reserve(size_t bytes) {
        if (bytes <= cap_) return;
        release();
        HIP_CHECK(hipMalloc(&base_, bytes));
        cap_ = bytes;
    }
    void release() {
        if (base_) (void)hipFree(base_);
        base_ = nullptr;
        cap_ = off_ = 0;
    }
    template <typename T> T *alloc(size_t count) {
        size_t bytes = (count * sizeof(T) + 255) & ~size_t(255);
        if (off_ + bytes > cap_) {
            char buf[160];
            snprintf(buf, sizeof buf, "device arena exhausted: need %zu more bytes (cap %zu, used %zu)",
                     bytes, cap_, off_);
            throw HipError(buf);
        }
        T *p = reinterpret_cast<T *>(static_cast<char *>(base_) + off_);
        off_ += bytes;
        if (off_ > peak_) peak_ = off_;
        return p;
    }
    size_t mark() const { return off_; }
    void rewind(size_t m) { off_ = m; }
    size_t capacity() const { return cap_; }
    size_t peak() const { return peak_; }

  private:
    void *base_ = nullptr;
    size_t cap_ = 0, off_ = 0, peak_ = 0;
};

// ---------------------------------------------------------------------------------------
// Stage profiler: HIP events recorded on the pipeline's own stream around named launches.
// Off by default (zero overhead); bench.py switches it on through nolzss_profile_enable().
// ---------------------------------------------------------------------------------------
class Profiler {
  public:
    struct Stat {
        uint64_t count = 0;
        double total_ms = 0.0;
        double bytes = 0.0;  // algorithmic bytes moved by the launches (0 if not stated)
    };
    ~Profiler() { drop(); }
    void enable(bool on) { on_ = on; }
    bool enabled() const { return on_; }
    void start(const char *name, hipStream_t s, double bytes = 0.0) {
        if (!on_) return;
        Entry e;
        e.name = name;
        e.bytes = bytes;
        e.a = take();  // events are pooled: creating and destroying two per scope cost ~1 % of a run
        e.b = take();
        HIP_CHECK(hipEventRecord(e.a, s));
        open_.push_back(e);
    }
    void stop(hipStream_t s) {
        if (!on_ || open_.empty()) return;
        Entry e = open_.back();
        open_.pop_back();
        (void)hipEventRecord(e.b, s);  // called from a destructor: never throws
        done_.push_back(e);
    }
    // algorithmic bytes that are only known after the launch (the latest finished scope of that name)
    void add_bytes(const char *name, double bytes) {
        if (!on_) return;
        for (size_t k = done_.size(); k-- > 0;)
            if (done_[k].name == name) {
                done_[k].bytes += bytes;
                return;
            }
        stats_[name].bytes += bytes;
    }
    // call after the stream has been synchronised
    void collect() {
        for (auto &e : done_) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
                Stat &st = stats_[e.name];
                st.count += 1;
                st.total_ms += ms;
                st.bytes += e.bytes;
            }
            pool_.push_back(e.a);
            pool_.push_back(e.b);
        }
        done_.clear();
    }
    void reset() {
        collect();
        stats_.clear();
    }
    const std::map<std::string, Stat> &stats() const { return stats_; }

  private:
    struct Entry {
        std::string name;
        double bytes = 0.0;
        hipEvent_t a = nullptr, b = nullptr;
    };
    void drop() {
        for (auto &e : open_) {
            (void)hipEventDestroy(e.a);
            (void)hipEventDestroy(e.b);
        }
        for (auto &e : done_) {
            (void)hipEventDestroy(e.a);
            (void)hipEventDestroy(e.b);
        }
        open_.clear();
        done_.clear();
        for (auto ev : pool_) (void)hipEventDestroy(ev);
        pool_.clear();
    }
    hipEvent_t take() {
        if (!pool_.empty()) {
            hipEvent_t ev = pool_.back();
            pool_.pop_back();
            return ev;
        }
        hipEvent_t ev = nullptr;
        HIP_CHECK(hipEventCreate(&ev));
        return ev;
    }
    std::vector<hipEvent_t> pool_;
    bool on_ = false;
    std::vector<Entry> open_, done_;
    std::map<std::string, Stat> stats_;
};

struct ProfScope {
    Profiler *p;
    hipStream_t s;
    ProfScope(Profiler *prof, const char *name, hipStream_t stream, double bytes = 0.0) : p(prof), s(stream) {
        if (p) p->start(name, s, bytes);
    }
    ~ProfScope() {
        if (p) p->stop(s);
    }
};

// ---------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ uint64_t lanemask_lt() {
    return (1ull << lane_id()) - 1ull;
}

template <typename T> struct OpAdd {
    __device__ __forceinline__ T operator()(T a, T b) const { return a + b; }
    static __device__ __forceinline__ T identity() { return T(0); }
};
template <typename T> struct OpMax {
    __device__ __forceinline__ T operator()(T a, T b) const { return a > b ? a : b; }
    static __device__ __forceinline__ T identity() { return T(0); }
};

// Inclusive scan of a u32 across the 64 lanes with DPP only (no LDS crossbar): four shifts
// inside each row of 16 lanes, then the last lane of row 0/2 is broadcast into row 1/3
// (row_bcast:15) and the last lane of row 1 into rows 2-3 (row_bcast:31).  Lanes without a
// source keep `identity` (bound_ctrl off).
template <typename Op>
__device__ __forceinline__ uint32_t wave_scan_inclusive_dpp(uint32_t v, uint32_t identity, Op op) {
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, 0x111, 0xf, 0xf, false));  // row_shr:1
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, 0x112, 0xf, 0xf, false));  // row_shr:2
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, 0x114, 0xf, 0xf, false));  // row_shr:4
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, 0x118, 0xf, 0xf, false));  // row_shr:8
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, 0x142, 0xa, 0xf, false));  // row_bcast:15
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, 0x143, 0xc, 0xf, false));  // row_bcast:31
    return v;
}

// inclusive scan across the 64 lanes of a wavefront
template <typename T, typename Op> __device__ __forceinline__ T wave_scan_inclusive(T v, Op op) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T o = __shfl_up(v, d, 64);
        if (lane_id() >= d) v = op(v, o);
    }
    return v;
}

template <typename T, typename Op> __device__ __forceinline__ T wave_reduce(T v, Op op) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = op(v, __shfl_xor(v, d, 64));
    return v;
}

// Exclusive scan of one value per thread across a workgroup of NW wavefronts; lds needs NW
// words.  Every thread must call it (contains barriers).  block_total = op over all threads.
template <int NW, typename Op>
__device__ __forceinline__ uint32_t block_scan_exclusive(uint32_t v, Op op, uint32_t *lds,
                                                         uint32_t &block_total) {
    uint32_t inc = wave_scan_inclusive(v, op);
    const int w = threadIdx.x >> 6;
    if (lane_id() == 63) lds[w] = inc;
    __syncthreads();
    uint32_t prefix = Op::identity();
    uint32_t total = Op::identity();
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        uint32_t s = lds[k];
        if (k < w) prefix = op(prefix, s);
        total = op(total, s);
    }
    __syncthreads();
    uint32_t exc = __shfl_up(inc, 1, 64);
    if (lane_id() == 0) exc = Op::identity();
    block_total = total;
    return op(prefix, exc);
}

}  // namespace nolzss
