/*
 * nolzss_oracle.h -- CPU restatement of the reference noLZSS algorithm.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The product path
 * (nolzss_amd/ + libnolzss_hip.so) never links, imports or calls it.
 *
 * Parity pinning: the reference extension cannot be built here (its CST/RMQ
 * come from sdsl-lite v3.0.3, fetched at CMake configure time; no copy is
 * available offline).  This restatement is therefore
 * pinned by the reference's own known-answer vectors (tests/golden/kats.json:
 * README.md:48-50, docs/examples.md:13-16, tests/test_cpp_bindings.py:715-747,
 * tests/test_genomics.py:268-290) and by its invariant tests.  Beyond those
 * vectors parity is UNPINNED ("parity unpinned": no reference-generated
 * golden exists for large inputs).
 *
 * All file:line citations are into /root/reference.
 */
#ifndef NOLZSS_ORACLE_H
#define NOLZSS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/cpp/factorizer.hpp:147-151 (struct Factor), :41 (RC_MASK) */
typedef struct {
    uint64_t start;
    uint64_t length;
    uint64_t ref;
} oracle_factor;

#define ORACLE_RC_MASK (1ULL << 63)

/* status codes */
#define ORACLE_OK 0
#define ORACLE_ERR_INVALID_ARGUMENT 1 /* std::invalid_argument in the reference */
#define ORACLE_ERR_RUNTIME 2          /* std::runtime_error in the reference */
#define ORACLE_ERR_NOMEM 3

const char *oracle_last_error(void);
void oracle_free(void *p);

/* Suffix array of text[0..n) under "shorter suffix first" order (what the
 * reference gets from sdsl::construct_im, which appends a 0 terminator:
 * factorizer.cpp:381).  sa has room for n entries. */
int oracle_suffix_array(const uint8_t *text, size_t n, int32_t *sa);

/* lcp[0] = 0, lcp[r] = lcp(text[sa[r-1]..], text[sa[r]..]) (Kasai). */
int oracle_lcp_array(const uint8_t *text, size_t n, const int32_t *sa, int32_t *lcp);

/* noLZSS::factorize / count_factors (factorizer.cpp:337-343, 378-384) over
 * detail::nolzss (factorizer_core.hpp:51-119).  *out is malloc'ed. */
int oracle_factorize(const uint8_t *text, size_t n, size_t start_pos,
                     oracle_factor **out, size_t *z);
int oracle_count_factors(const uint8_t *text, size_t n, size_t start_pos, size_t *z);

/* Per-position variant used to test GPU intermediates: the factor that the
 * reference loop would emit if its cursor stood at every position i
 * (len[i] >= 1, ref[i]); same tree walk, applied to all i. */
int oracle_lpnf_all(const uint8_t *text, size_t n, uint32_t *len, uint32_t *ref);

/* prepare_multiple_dna_sequences_w_rc (factorizer.cpp:54-172).
 * S is malloc'ed (may contain any byte value); sentinel_pos is malloc'ed with
 * *n_sentinels entries. */
int oracle_prepare_multiple_dna_w_rc(const char *const *seqs, const size_t *lens, size_t k,
                                     uint8_t **S, size_t *S_len, size_t *original_length,
                                     uint64_t **sentinel_pos, size_t *n_sentinels);

/* detail::nolzss_multiple_dna_w_rc (factorizer_core.hpp:177-383) on a prepared S. */
int oracle_factorize_multiple_dna_w_rc(const uint8_t *S, size_t S_len, size_t start_pos,
                                       oracle_factor **out, size_t *z);
int oracle_count_factors_multiple_dna_w_rc(const uint8_t *S, size_t S_len, size_t start_pos,
                                           size_t *z);

/* noLZSS::factorize_dna_w_rc (factorizer.cpp:519-523 via factorizer_core.hpp:140-151). */
int oracle_factorize_dna_w_rc(const uint8_t *text, size_t n, oracle_factor **out, size_t *z);

/* Per-position RC variant for GPU intermediate tests: factor the RC loop would
 * emit with its cursor at every i < N (ref carries ORACLE_RC_MASK when RC). */
int oracle_lpnf_all_rc(const uint8_t *S, size_t S_len, uint32_t *len, uint64_t *ref);

#ifdef __cplusplus
}
#endif
#endif
