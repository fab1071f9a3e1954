// debug_api.hip -- debug hooks (plans, intermediate arrays, primitives) and the stage profiler
// (part of the C ABI layer of libnolzss_hip.so, include/nolzss_hip.h; shared declarations: api_internal.hpp)
#include "api_internal.hpp"

using namespace nolzss;
using namespace nolzss::api;

extern "C" {

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
