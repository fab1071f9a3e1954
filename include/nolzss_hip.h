/*
 * nolzss_hip.h -- C ABI of libnolzss_hip.so, the MI355X (gfx950) drop-in for the compiled core
 * of OmerKerner/noLZSS on the factorize path.
 *
 * Each entry point replaces one function that the reference's pybind11 module `_noLZSS`
 * (reference: src/cpp/bindings.cpp) binds for this path; the citation next to a declaration
 * names the reference interface it stands in for.  Plain pointers and sizes only: no C++,
 * pybind or torch types cross this boundary.  INTEGRATION.md shows the pybind11 / ctypes stub
 * a maintainer of the reference would add to bind these.
 *
 * Conventions
 *   - every function returns a status code (NOLZSS_OK == 0); on failure
 *     nolzss_last_error() returns a thread-local message.  NOLZSS_ERR_INVALID_ARGUMENT maps to
 *     the reference's std::invalid_argument (Python ValueError), NOLZSS_ERR_RUNTIME to
 *     std::runtime_error (RuntimeError)  -- bindings.cpp relies on pybind11's default mapping.
 *   - input buffers are borrowed for the duration of the call (bindings.cpp:66-67);
 *   - output arrays are allocated by the library and released with nolzss_free();
 *   - `device` is a HIP device ordinal; the library keeps one context (stream + device arena)
 *     per device and serialises calls on it, so calls are safe from any host thread
 *     (the reference releases the GIL around compute, bindings.cpp:70).
 *   - there is NO CPU fallback: without a usable GPU every compute entry point fails with
 *     NOLZSS_ERR_DEVICE.
 */
#ifndef NOLZSS_HIP_H
#define NOLZSS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference: struct Factor, src/cpp/factorizer.hpp:147-151 (24-byte POD, also the on-disk record) */
typedef struct nolzss_factor {
    uint64_t start;
    uint64_t length;
    uint64_t ref;
} nolzss_factor;

/* reference: RC_MASK, src/cpp/factorizer.hpp:41 */
#define NOLZSS_RC_MASK (1ULL << 63)

enum {
    NOLZSS_OK = 0,
    NOLZSS_ERR_INVALID_ARGUMENT = 1,
    NOLZSS_ERR_RUNTIME = 2,
    NOLZSS_ERR_NOMEM = 3,
    NOLZSS_ERR_DEVICE = 4,
    NOLZSS_ERR_IO = 5,
    NOLZSS_ERR_UNSUPPORTED = 6 /* the caller's own (host) path must handle this input */
};

const char *nolzss_last_error(void);
/* reference: m.attr("__version__"), bindings.cpp:1513-1517 */
const char *nolzss_version(void);
void nolzss_free(void *p);
int nolzss_device_count(int *count);

/* ---- plain mode ------------------------------------------------------------------------ */
/* reference: noLZSS::factorize(string_view, start_pos), factorizer.cpp:378-384;
 *            bound as _noLZSS.factorize, bindings.cpp:56-77 */
int nolzss_factorize(const uint8_t *text, size_t n, size_t start_pos, int device,
                     nolzss_factor **out, size_t *z);
/* reference: noLZSS::count_factors, factorizer.cpp:337-343; bindings.cpp:122-141 */
int nolzss_count_factors(const uint8_t *text, size_t n, size_t start_pos, int device, size_t *z);
/* reference: noLZSS::factorize_file, factorizer.cpp:401-406; bindings.cpp:96-105 */
int nolzss_factorize_file(const char *path, size_t start_pos, int device, nolzss_factor **out,
                          size_t *z);
/* reference: noLZSS::count_factors_file, factorizer.cpp:359-363; bindings.cpp:157-164 */
int nolzss_count_factors_file(const char *path, size_t start_pos, int device, size_t *z);

/* Same computation with the text already resident in device memory (d_text is a device
 * pointer on `device`); `stream` is a hipStream_t or NULL for the context's own stream.  With NULL the
 * call is ordered behind everything already queued on the legacy default stream (torch's default
 * stream); a producer of d_text on another non-blocking stream must be synchronised by the caller or
 * hand in its stream.
 * emit = 0: count only (the count_factors path);
 * emit = 1: build all z factor records in HBM and stop there (no PCIe transfer);
 * emit = 2: also copy them into a malloc'ed host array returned through out_host.
 * Used by bench.py (inputs resident in HBM when the clock starts) and by the shard dispatcher. */
int nolzss_factorize_device(const void *d_text, size_t n, size_t start_pos, int device,
                            void *stream, int emit, nolzss_factor **out_host, size_t *z);

/* ---- reverse-complement DNA mode ---------------------------------------------------------- */
/* reference: prepare_multiple_dna_sequences_w_rc, factorizer.cpp:54-172; bindings.cpp:732-740.
 * S (malloc'ed, may hold any byte value) = T1 s0 ... Tk s(k-1) rc(Tk) sk ... rc(T1) s(2k-1). */
int nolzss_prepare_multiple_dna_w_rc(const char *const *seqs, const size_t *lens, size_t k,
                                     uint8_t **S, size_t *S_len, size_t *original_length,
                                     uint64_t **sentinel_positions, size_t *n_sentinels);
/* reference: noLZSS::factorize_multiple_dna_w_rc, factorizer.cpp:651-656 over
 *            detail::nolzss_multiple_dna_w_rc, factorizer_core.hpp:177-383; bindings.cpp:361-382 */
int nolzss_factorize_multiple_dna_w_rc(const uint8_t *S, size_t S_len, size_t start_pos,
                                       int device, nolzss_factor **out, size_t *z);
/* reference: count_factors_multiple_dna_w_rc, factorizer.cpp:700-705; bindings.cpp:427-446 */
int nolzss_count_factors_multiple_dna_w_rc(const uint8_t *S, size_t S_len, size_t start_pos,
                                           int device, size_t *z);
/* reference: noLZSS::factorize_dna_w_rc, factorizer.cpp:519-523; bindings.cpp:207-228 */
int nolzss_factorize_dna_w_rc(const uint8_t *text, size_t n, int device, nolzss_factor **out,
                              size_t *z);
/* reference: noLZSS::count_factors_dna_w_rc, factorizer.cpp:559-561; bindings.cpp:276-295 */
int nolzss_count_factors_dna_w_rc(const uint8_t *text, size_t n, int device, size_t *z);
/* noLZSS::factorize_dna_w_rc (factorizer.cpp:519-523) with the text already resident in device memory: the
 * counterpart of nolzss_factorize_device for the reverse-complement mode (same `stream` and `emit` meaning; the
 * prepared string T s0 rc(T) s1 of factorizer.cpp:54-172 is built on the device).  Used by bench.py (BASELINE
 * config 5 with the input in HBM when the clock starts). */
int nolzss_factorize_dna_w_rc_device(const void *d_text, size_t n, int device, void *stream, int emit,
                                     nolzss_factor **out_host, size_t *z);

/* ---- reference + target factorization (the chain simply starts at start_pos) */
/* reference: noLZSS::factorize_w_reference, factorizer.cpp:940-955; bindings.cpp:868-880.
 * combined = reference 0x01 target, factorized from |reference| + 1; positions are absolute. */
int nolzss_factorize_w_reference(const uint8_t *reference_seq, size_t reference_len,
                                 const uint8_t *target_seq, size_t target_len, int device,
                                 nolzss_factor **out, size_t *z);
/* reference: noLZSS::factorize_dna_w_reference_seq, factorizer.cpp:825-842; bindings.cpp:800-808.
 * prepare({reference, target}) with reverse complements, factorized from |reference| + 1. */
int nolzss_factorize_dna_w_reference_seq(const char *reference_seq, size_t reference_len,
                                         const char *target_seq, size_t target_len, int device,
                                         nolzss_factor **out, size_t *z);

/* ---- v2 binary factor files -------------------------------------------- */
/* File = z 24-byte records, optional metadata, 48-byte footer "noLZSSv2", num_factors,
 * num_sequences, num_sentinels, footer_size, total_length (factorizer.hpp:64-77).
 * reference: write_factors_binary_file, factorizer.cpp:424-459 (input FILE -> output file) */
int nolzss_write_factors_binary_file(const char *in_path, const char *out_path, int device, size_t *z);
/* reference: write_factors_binary_file_dna_w_rc, factorizer.cpp:597-635 */
int nolzss_write_factors_binary_file_dna_w_rc(const char *in_path, const char *out_path, int device,
                                              size_t *z);
/* reference: factorize_w_reference_file, factorizer.cpp:980-1021 */
int nolzss_factorize_w_reference_file(const uint8_t *reference_seq, size_t reference_len,
                                      const uint8_t *target_seq, size_t target_len,
                                      const char *out_path, int device, size_t *z);
/* reference: factorize_dna_w_reference_seq_file, factorizer.cpp:851-883 */
int nolzss_factorize_dna_w_reference_seq_file(const char *reference_seq, size_t reference_len,
                                              const char *target_seq, size_t target_len,
                                              const char *out_path, int device, size_t *z);

/* Writes z records + `extra` metadata bytes (may be NULL) + the 48-byte v2 footer. */
int nolzss_write_factor_file(const char *out_path, const nolzss_factor *factors, size_t z,
                             uint64_t num_sequences, uint64_t num_sentinels, uint64_t total_length,
                             const void *extra, size_t extra_len);

/* ---- concatenated multi-sequence FASTA with sentinel bookkeeping ---------- */
/* reference: prepare_multiple_dna_sequences_no_rc, factorizer.cpp:199-294; bindings.cpp (same shape
 * as the w_rc variant): S = T1 s0 T2 s1 ... Tk (no sentinel after the last sequence), <= 250. */
int nolzss_prepare_multiple_dna_no_rc(const char *const *seqs, const size_t *lens, size_t k,
                                      uint8_t **S, size_t *S_len, size_t *original_length,
                                      uint64_t **sentinel_positions, size_t *n_sentinels);

typedef struct nolzss_fasta_result {   /* FastaFactorizationResult, fasta_processor.hpp */
    nolzss_factor *factors;
    size_t num_factors;
    uint64_t *sentinel_factor_indices; /* indices into factors[] of the sentinel literals */
    size_t num_sentinels;
    char *sequence_ids;                /* num_sequences NUL-terminated ids, back to back */
    size_t sequence_ids_bytes;
    size_t num_sequences;
} nolzss_fasta_result;

/* reference: factorize_fasta_multiple_dna_w_rc / _no_rc, fasta_processor.cpp:298-341 over
 * parse_fasta_sequences_and_ids (:28-128) and identify_sentinel_factors (:131-163).
 * sanitize_mode: 0 = "remove_ambiguous" (default of the bindings), 1 = "strict". */
int nolzss_factorize_fasta_multiple_dna(const char *fasta_path, int with_rc, int sanitize_mode,
                                        int device, nolzss_fasta_result *out);
void nolzss_free_fasta_result(nolzss_fasta_result *r);
/* reference: write_factors_binary_file_fasta_multiple_dna_w_rc / _no_rc (fasta_processor.cpp:345-360
 * -> parallel_fasta_processor.cpp:64-257): records, names, sentinel indices, footer with
 * total_length = sum of factor lengths. */
int nolzss_write_factors_binary_file_fasta_multiple_dna(const char *fasta_path, const char *out_path,
                                                        int with_rc, int sanitize_mode, int device,
                                                        size_t *z);

/* reference: factorize_dna_rc_w_ref_fasta_files, fasta_processor.cpp:362-378 over
 * prepare_ref_target_dna_w_rc_from_fasta (:240-287): all reference records, then all target records,
 * prepared with reverse complements; factorization starts at the first target base. */
int nolzss_factorize_dna_rc_w_ref_fasta_files(const char *reference_fasta_path, const char *target_fasta_path,
                                              int sanitize_mode, int device, nolzss_fasta_result *out);
/* reference: write_factors_dna_w_reference_fasta_files_to_binary, fasta_processor.cpp:381-390 */
int nolzss_write_factors_dna_w_reference_fasta_files_to_binary(const char *reference_fasta_path,
                                                               const char *target_fasta_path,
                                                               const char *out_path, int sanitize_mode,
                                                               int device, size_t *z);

/* Per-sequence FASTA factorization (each record on its own; no concatenation).
 * reference: factorize_/count_factors_/write_factors_binary_file_fasta_dna_{w,no}_rc_per_sequence,
 * fasta_processor.cpp:430-561, parallel_fasta_processor.cpp:262-465.  factors[j] / counts[j] per
 * record (factors is NULL when want_factors == 0); with out_dir != NULL every record is also
 * written to out_dir/<sanitised id>.bin (one name, no sentinels, total_length = sum of lengths).
 * Kept from the reference: the no-rc variants drop the last base of every record
 * (fasta_processor.cpp:469-471). */
typedef struct nolzss_fasta_per_sequence_result {
    nolzss_factor **factors;
    size_t *counts;
    char *sequence_ids;
    size_t sequence_ids_bytes;
    size_t num_sequences;
} nolzss_fasta_per_sequence_result;
int nolzss_factorize_fasta_per_sequence(const char *fasta_path, int with_rc, int sanitize_mode,
                                        int want_factors, const char *out_dir, int device,
                                        nolzss_fasta_per_sequence_result *out);
void nolzss_free_fasta_per_sequence_result(nolzss_fasta_per_sequence_result *r);

/* ---- per-sequence batch (the FASTA shard unit) ------------------------------------------- */
/* reference: the per-sequence factorize() loop of genomics.read_nucleotide_fasta,
 *            src/noLZSS/genomics/fasta.py:110-122 (C++ analogue:
 *            parallel_fasta_processor.cpp:360-385).  Sequence j is factorized on
 *            devices[j % n_dev]-th device of the list in longest-first order; out[j] / z[j]
 *            are per sequence (out may be NULL for counts only).
 *            Short records (fewer than NOLZSS_BATCH_MERGE_BELOW bases, default 2^21) that hold only
 *            A/C/G/T are factorized TOGETHER, as independent sequences of one device run (same
 *            results; 35 -> 3000 Mbases/s for 4 Ki-base records): out[j] may therefore point into a
 *            block shared with other records.  Free ONLY with nolzss_free_batch(). */
int nolzss_factorize_batch(const uint8_t *const *texts, const size_t *lens, size_t m,
                           const int *devices, size_t n_dev, nolzss_factor ***out, size_t **z);
/* The same with the reverse complement of every record: record j as factorize_dna_w_rc would
 * (prepare_multiple_dna_sequences_w_rc({seq}) + factorize_multiple_dna_w_rc, the per-record step of
 * factorize_fasta_dna_w_rc_per_sequence, fasta_processor.cpp:446-451; lower case accepted, refs of
 * reverse-complement factors carry NOLZSS_RC_MASK).  Short records are merged in the layout of
 * factorizer.cpp:128-169 (T1 s T2 s .. Tk s rc(Tk) s .. rc(T1) s) for any number of records, each
 * record seeing only itself and its own reverse complement.  An invalid nucleotide fails the call
 * like the reference ("Invalid nucleotide ... found in sequence 0"). */
int nolzss_factorize_batch_dna_w_rc(const uint8_t *const *texts, const size_t *lens, size_t m,
                                    const int *devices, size_t n_dev, nolzss_factor ***out, size_t **z);
void nolzss_free_batch(nolzss_factor **out, size_t *z, size_t m);
/* The plain per-sequence batch with the records already in the memory of `device` (d_texts[j] = device
 * pointer to lens[j] bytes): no PCIe leg.  emit = 0 counts, emit = 1 also builds the factor records of
 * every record in HBM and stops there.  z[j] (caller-allocated, m entries) = factors of record j.  Used by
 * bench.py for the FASTA shard workload (inputs resident in HBM when the clock starts).
 * THIS CALL MAY SLEEP: records are merged into runs that several host threads ("lanes") submit, and when the first two
 * runs are of similar size the later lanes start up to NOLZSS_DEVICE_MERGE_STAGGER_MS (default 20 ms, scaled by the run
 * size) after the first, so that their bandwidth-bound and issue-bound phases overlap instead of coinciding. */
int nolzss_factorize_batch_device(const void *const *d_texts, const size_t *lens, size_t m, int device, int emit,
                                  size_t *z);

/* ---- genomics.read_nucleotide_fasta: FASTA file in, per-record factors out ---------------------- */
/* reference: read_nucleotide_fasta + _parse_fasta_content, src/noLZSS/genomics/fasta.py:28-126: parse
 * (id = first header word, bases upper-cased, white space dropped, a repeated id keeps its place and
 * takes the last record), check ^[ACGT]+$, then factorize() every record on its own (:110-122) -- here
 * as ONE per-sequence batch over the listed devices, the records read in one piece and handed to the
 * device as views of the read buffer.  Errors carry the reference's FASTAError texts ("Empty sequence
 * header at line N", "Sequence data before header at line N", "No valid sequences found in FASTA file",
 * "Sequence 'id' contains invalid nucleotides: {...}") with NOLZSS_ERR_RUNTIME; a file with non-ASCII
 * bytes returns NOLZSS_ERR_UNSUPPORTED (the Python layer then parses it itself).
 * Sharding (one process per GPU): with shard_count > 1 only the records that the longest-processing-
 * time-first plan gives to shard_index are factorized (owners[j] = shard of record j, the same on every
 * rank); counts[j] = 0 and factors[j] = NULL for the others, and the caller all-gathers the counts. */
typedef struct nolzss_nucleotide_fasta {
    char *sequence_ids;       /* num_sequences NUL-terminated ids, back to back, first-appearance order */
    size_t sequence_ids_bytes;
    size_t num_sequences;
    size_t *lengths;          /* bases per record */
    size_t *counts;           /* factors per record */
    size_t *owners;           /* shard that factorized the record */
    nolzss_factor **factors;  /* per record (NULL array when want_factors == 0) */
    void *keep;               /* owns the factor blocks */
} nolzss_nucleotide_fasta;
int nolzss_read_nucleotide_fasta(const char *path, const int *devices, size_t n_dev, int want_factors,
                                 size_t shard_index, size_t shard_count, nolzss_nucleotide_fasta *out);
void nolzss_free_nucleotide_fasta(nolzss_nucleotide_fasta *r);

/* ---- measurement hooks -------------------------------------------------------------------- */
/* HIP-event timing of every pipeline stage on the context's stream (off by default). */
int nolzss_profile_enable(int device, int on);
int nolzss_profile_reset(int device);
/* Writes lines "name count total_ms algorithmic_bytes\n" into buf (NUL-terminated, truncated
 * to cap).  Nested scopes are reported separately (a stage and the kernels inside it). */
int nolzss_profile_report(int device, char *buf, size_t cap);

/* ---- introspection used by the parity tests of the intermediate arrays -------------------- */
/* Any output pointer may be NULL.  sa/isa/lstar: n entries; lcp: n + 1 entries. */
int nolzss_debug_arrays(const uint8_t *text, size_t n, int device, uint32_t *sa, uint32_t *isa,
                        uint32_t *lcp, uint32_t *lstar);
/* Sorts n (key, value) pairs in place on the device by all 64 key bits (stable). */
int nolzss_debug_sort_pairs(uint64_t *keys, uint32_t *vals, size_t n, int device);
/* mode 0: exclusive add-scan, mode 1: inclusive max-scan, in place. */
int nolzss_debug_scan(uint32_t *data, size_t n, int mode, int device);
/* Capacity and high-water mark (bytes) of the device arena of `device` (lane 0). */
int nolzss_debug_arena(int device, size_t *capacity, size_t *peak);
/* The FASTA reader behind the nolzss_*fasta* entry points (host only, no device needed): records and
 * ids as NUL-terminated strings back to back; sanitize_mode 0 = remove ambiguous, 1 = strict.
 * reference: parse_fasta_sequences_and_ids, fasta_processor.cpp:28-128.  Free both with nolzss_free(). */
int nolzss_debug_parse_fasta(const char *path, int sanitize_mode, char **ids, size_t *ids_bytes,
                             char **sequences, size_t *sequences_bytes, size_t *count);
/* The reader behind nolzss_read_nucleotide_fasta alone (host only): the records after the nucleotide
 * check, or the error the reference's Python reader raises.  reference: genomics/fasta.py:28-76, 110-115. */
int nolzss_debug_parse_nucleotide_fasta(const char *path, char **ids, size_t *ids_bytes, char **sequences,
                                        size_t *sequences_bytes, size_t *count);
/* The shard plan of nolzss_read_nucleotide_fasta: owners[j] = shard of a record of lens[j] bases. */
int nolzss_debug_lpt_plan(const size_t *lens, size_t m, size_t bins, size_t *owners);
/* The static plan of nolzss_factorize_batch / _dna_w_rc for m records on n_dev devices, without touching a device
 * (host logic only): chunk_of[j] = index of the merged run record j shares (-1: none), device_of[j] = slot in the device
 * list of the pipeline run record j takes on its own (-1: none; both -1: an empty record).  The merged runs are taken by
 * n_dev x 2 lanes from ONE work queue (lane w on device w % n_dev), the single records are dealt to the devices
 * longest-processing-time first.  n_chunks (optional) = number of merged runs. */
int nolzss_debug_batch_plan(const size_t *lens, size_t m, size_t n_dev, int with_rc, int32_t *chunk_of, int32_t *device_of,
                            size_t *n_chunks);
/* Gives the device arenas that no call is using back to the driver (they are otherwise kept between
 * calls and only grow; the library does this by itself when a reservation fails). */
int nolzss_debug_trim_arenas(int device, size_t *released_bytes);
/* Records factorized since the library was loaded by merged runs / one pipeline run each. */
void nolzss_debug_batch_counters(uint64_t *merged_records, uint64_t *single_records);

#ifdef __cplusplus
}
#endif
#endif /* NOLZSS_HIP_H */
