// api_internal.hpp -- what the translation units of the C ABI layer share (round 4: api.hip split into c_abi.hip --
// device sessions, arenas, copies and the plain / reverse-complement / reference-target entry points --, batch.hip --
// the merged batch and its scheduler --, fasta_api.hip -- the FASTA entry points --, fasta_reader.cpp and factor_file.cpp
// -- host-only parsers and the v2 file writer --, debug_api.hip -- debug hooks and the stage profiler).
#pragma once
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
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <thread>

#include <malloc.h>

#include "fasta_reader.hpp"
#include "host_util.hpp"
#include "pipeline.hpp"
#include "pyramid.hpp"
#include "radix_sort.hpp"
#include "scan.hpp"

namespace nolzss {
namespace api {

// ---- errors, device sessions, arenas (c_abi.hip) ---------------------------------------------------------------
extern thread_local std::string g_error;
int set_error(int code, const std::string &msg);

struct DeviceContext {
    std::mutex mu;
    Context ctx;
    bool ready = false;
};

constexpr size_t kMaxText = 0xffffffffull - (1ull << 19);  // 32-bit index pipeline (sharded queue slots stay below 2^32)
constexpr size_t kArenaBytesPerSymbol = 96;
constexpr size_t kArenaMinPerSymbol = 52;
constexpr size_t kArenaSlack = size_t(64) << 20;
size_t arena_bytes_for(size_t n);
size_t arena_min_bytes_for(size_t n);
constexpr int kMaxLanes = 16;  // concurrent pipelines (stream + arena each) per device
DeviceContext &get_context(int device, int lane = 0);

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

size_t trim_idle_arenas(int device, const Context *keep);
void reserve_arena_for(Context &ctx, size_t n, size_t extra = 0);
uint8_t *host_stage(Context &ctx, size_t bytes);
void upload_bytes(Context &ctx, void *d_dst, const void *h_src, size_t n);
void download_bytes(Context &ctx, void *h_dst, const void *d_src, size_t n);
void order_behind_default_stream(Context &ctx);

struct DebugOut {
    uint32_t *sa = nullptr, *isa = nullptr, *lcp = nullptr, *lstar = nullptr;
};

// the plain-mode pipeline on a device-resident text / on a host buffer (upload first); returns z
size_t run_plain(Context &ctx, const uint8_t *d_text, size_t n, size_t start_pos, nolzss_factor **out_host,
                 DebugOut *dbg, bool records_on_device_only = false);
size_t run_plain_host(Context &ctx, const uint8_t *text, size_t n, size_t start_pos, nolzss_factor **out,
                      DebugOut *dbg);

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

void check_text_args(const void *text, size_t n, size_t start_pos);

// ---- reverse-complement preparation and pipeline (c_abi.hip) -------------------------------------------------------
uint8_t rc_sentinel(size_t index);
size_t first_invalid_nucleotide(const char *s, size_t n);
void copy_upper(uint8_t *dst, const char *src, size_t n);
void copy_reverse_complement(uint8_t *dst, const char *src, size_t n);
void prepare_w_rc(const char *const *seqs, const size_t *lens, size_t k, HostBytes &S,
                  size_t &original_length, std::vector<uint64_t> &sentinels);
bool rc_guards(size_t S_len, size_t start_pos);
size_t run_rc_host(Context &ctx, const uint8_t *S, size_t m, size_t start_pos, nolzss_factor **out);
void dna_w_rc_common(const uint8_t *text, const uint8_t *d_resident, size_t n, int device, void *stream, int emit,
                     nolzss_factor **out, size_t *z, int lane = 0);

// ---- reference + target, v2 files (c_abi.hip, factor_file.cpp) ----------------------------------------------------
void write_v2_file(const char *out_path, const nolzss_factor *f, size_t z, uint64_t num_sequences,
                   uint64_t num_sentinels, uint64_t total_length, const std::string &extra);
size_t w_reference(const uint8_t *ref, size_t ref_len, const uint8_t *tgt, size_t tgt_len, int device,
                   nolzss_factor **out);
size_t dna_w_reference(const char *ref, size_t ref_len, const char *tgt, size_t tgt_len, int device,
                       nolzss_factor **out);

// ---- concatenated multi-sequence FASTA (fasta_api.hip) -------------------------------------------------------------
void prepare_no_rc(const char *const *seqs, const size_t *lens, size_t k, HostBytes &S,
                   size_t &original_length, std::vector<uint64_t> &sentinels);
std::vector<uint64_t> sentinel_factors(const nolzss_factor *f, size_t z, const std::vector<uint64_t> &positions);
struct FastaFactors {
    FastaParse parse;
    nolzss_factor *factors = nullptr;
    size_t z = 0;
    std::vector<uint64_t> sentinel_idx;
    ~FastaFactors() { std::free(factors); }
};
void factorize_fasta(const char *path, bool with_rc, bool strict, int device, FastaFactors &out);
void factorize_ref_target_fasta(const char *ref_path, const char *tgt_path, bool strict, int device,
                                FastaFactors &out);
void fill_fasta_result(FastaFactors &ff, nolzss_fasta_result *out);
void write_fasta_file(const char *out_path, const FastaFactors &ff);

// ---- the batch (batch.hip) ---------------------------------------------------------------------------------------
struct BatchPlan {
    std::vector<std::vector<size_t>> chunks;
    std::vector<size_t> singles;
};
BatchPlan plan_batch(const size_t *lens, size_t m, bool with_rc);
std::vector<std::vector<size_t>> lpt_plan_singles(std::vector<size_t> &singles, const size_t *lens, size_t n_dev);
void factorize_many(const uint8_t *const *texts, const size_t *lens, size_t m, const int *devices, size_t n_dev,
                    bool with_rc, size_t *zs, nolzss_factor **fs, std::vector<void *> &blocks);
extern std::atomic<uint64_t> g_merged_records, g_single_records;

struct NucleotideFastaKeep {
    NucleotideFasta parse;
    std::vector<void *> blocks;
    ~NucleotideFastaKeep() {
        for (void *b : blocks) free_block(b);
    }
};

}  // namespace api
}  // namespace nolzss
