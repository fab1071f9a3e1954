// fasta_api.hip -- the FASTA entry points: concatenated multi-sequence factorization, per-sequence batch, genomics.read_nucleotide_fasta
// (part of the C ABI layer of libnolzss_hip.so, include/nolzss_hip.h; shared declarations: api_internal.hpp)
#include "api_internal.hpp"

namespace nolzss {
namespace api {

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

}  // namespace api
}  // namespace nolzss

using namespace nolzss;
using namespace nolzss::api;

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

}  // extern "C"
