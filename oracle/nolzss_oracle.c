/*
 * nolzss_oracle.c -- CPU restatement of the reference noLZSS algorithm.
 *
 * TEST INFRASTRUCTURE ONLY (see nolzss_oracle.h).  Parity beyond the
 * reference's own known-answer vectors is UNPINNED: the reference cannot be
 * built offline (sdsl-lite v3.0.3 is fetched at configure time).
 *
 * What is restated, and from where (all paths under /root/reference):
 *   - the compressed suffix tree the reference gets from sdsl::cst_sada<>
 *     (src/cpp/factorizer_helpers.hpp:6, construct_im at factorizer.cpp:381)
 *     is replaced by plain arrays: suffix array (SA-IS), Kasai LCP, and the
 *     LCP-interval tree (explicit internal nodes of the suffix tree, with
 *     string depth, parent, and the per-node minimum the reference obtains
 *     with rmq_succinct_sct: factorizer_core.hpp:53,72,231-232,264-270);
 *   - detail::nolzss                       src/cpp/factorizer_core.hpp:51-119
 *   - detail::nolzss_multiple_dna_w_rc     src/cpp/factorizer_core.hpp:177-383
 *   - detail::nolzss_dna_w_rc              src/cpp/factorizer_core.hpp:140-151
 *   - lcp()                                src/cpp/factorizer_helpers.hpp:20-24
 *   - prepare_multiple_dna_sequences_w_rc  src/cpp/factorizer.cpp:54-172
 *   - complement/revcomp                   src/cpp/factorizer.cpp:17-32
 *
 * The walks below visit the same nodes in the same (top-down) order as the
 * reference and take the same branches; only the data structure differs.
 */
#include "nolzss_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char g_err[512];

const char *oracle_last_error(void) { return g_err; }
void oracle_free(void *p) { free(p); }

static int fail(int code, const char *msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}

/* ------------------------------------------------------------------ */
/* Suffix array: SA-IS (Nong, Zhang, Chan 2009) over an int alphabet.  */
/* s[n-1] must be a unique smallest symbol (the terminator that        */
/* sdsl::construct_im appends, factorizer.cpp:381).                    */
/* ------------------------------------------------------------------ */
typedef int32_t idx_t;

#define T_GET(i) ((tp[(size_t)(i) >> 3] >> ((i)&7)) & 1) /* 1 = S-type */
#define T_SET(i, b)                                                                   \
    do {                                                                              \
        if (b)                                                                        \
            tp[(size_t)(i) >> 3] |= (unsigned char)(1u << ((i)&7));                   \
        else                                                                          \
            tp[(size_t)(i) >> 3] &= (unsigned char)~(1u << ((i)&7));                  \
    } while (0)
#define IS_LMS(i) ((i) > 0 && T_GET(i) && !T_GET((i)-1))

static void bucket_bounds(const idx_t *s, idx_t *bkt, idx_t n, idx_t K, int ends) {
    idx_t i, sum = 0;
    for (i = 0; i <= K; ++i) bkt[i] = 0;
    for (i = 0; i < n; ++i) bkt[s[i]]++;
    for (i = 0; i <= K; ++i) {
        sum += bkt[i];
        bkt[i] = ends ? sum : sum - bkt[i];
    }
}

static void induce_l(const unsigned char *tp, idx_t *SA, const idx_t *s, idx_t *bkt, idx_t n,
                     idx_t K) {
    idx_t i, j;
    bucket_bounds(s, bkt, n, K, 0);
    for (i = 0; i < n; ++i) {
        j = SA[i] - 1;
        if (SA[i] > 0 && !T_GET(j)) SA[bkt[s[j]]++] = j;
    }
}

static void induce_s(const unsigned char *tp, idx_t *SA, const idx_t *s, idx_t *bkt, idx_t n,
                     idx_t K) {
    idx_t i, j;
    bucket_bounds(s, bkt, n, K, 1);
    for (i = n - 1; i >= 0; --i) {
        j = SA[i] - 1;
        if (SA[i] > 0 && T_GET(j)) SA[--bkt[s[j]]] = j;
    }
}

/* K = largest symbol value.  Returns 0 on success. */
static int sais(const idx_t *s, idx_t *SA, idx_t n, idx_t K) {
    idx_t i, j, n1, name, prev;
    unsigned char *tp;
    idx_t *bkt, *SA1, *s1;

    if (n == 1) {
        SA[0] = 0;
        return 0;
    }
    tp = (unsigned char *)calloc((size_t)n / 8 + 1, 1);
    bkt = (idx_t *)malloc(sizeof(idx_t) * ((size_t)K + 1));
    if (!tp || !bkt) {
        free(tp);
        free(bkt);
        return -1;
    }
    T_SET(n - 2, 0);
    T_SET(n - 1, 1);
    for (i = n - 3; i >= 0; --i)
        T_SET(i, (s[i] < s[i + 1] || (s[i] == s[i + 1] && T_GET(i + 1))) ? 1 : 0);

    /* stage 1: sort LMS substrings */
    bucket_bounds(s, bkt, n, K, 1);
    for (i = 0; i < n; ++i) SA[i] = -1;
    for (i = 1; i < n; ++i)
        if (IS_LMS(i)) SA[--bkt[s[i]]] = i;
    induce_l(tp, SA, s, bkt, n, K);
    induce_s(tp, SA, s, bkt, n, K);

    n1 = 0;
    for (i = 0; i < n; ++i)
        if (IS_LMS(SA[i])) SA[n1++] = SA[i];
    for (i = n1; i < n; ++i) SA[i] = -1;
    name = 0;
    prev = -1;
    for (i = 0; i < n1; ++i) {
        idx_t pos = SA[i], d;
        int diff = 0;
        for (d = 0; d < n; ++d) {
            if (prev == -1 || s[pos + d] != s[prev + d] || T_GET(pos + d) != T_GET(prev + d)) {
                diff = 1;
                break;
            } else if (d > 0 && (IS_LMS(pos + d) || IS_LMS(prev + d)))
                break;
        }
        if (diff) {
            ++name;
            prev = pos;
        }
        SA[n1 + pos / 2] = name - 1;
    }
    for (i = n - 1, j = n - 1; i >= n1; --i)
        if (SA[i] >= 0) SA[j--] = SA[i];

    /* stage 2: solve the reduced problem */
    SA1 = SA;
    s1 = SA + n - n1;
    if (name < n1) {
        if (sais(s1, SA1, n1, name - 1) != 0) {
            free(tp);
            free(bkt);
            return -1;
        }
    } else {
        for (i = 0; i < n1; ++i) SA1[s1[i]] = i;
    }

    /* stage 3: induce the full order */
    bucket_bounds(s, bkt, n, K, 1);
    for (i = 1, j = 0; i < n; ++i)
        if (IS_LMS(i)) s1[j++] = i;
    for (i = 0; i < n1; ++i) SA1[i] = s1[SA1[i]];
    for (i = n1; i < n; ++i) SA[i] = -1;
    for (i = n1 - 1; i >= 0; --i) {
        j = SA[i];
        SA[i] = -1;
        SA[--bkt[s[j]]] = j;
    }
    induce_l(tp, SA, s, bkt, n, K);
    induce_s(tp, SA, s, bkt, n, K);
    free(bkt);
    free(tp);
    return 0;
}

int oracle_suffix_array(const uint8_t *text, size_t n, int32_t *sa) {
    idx_t *s, *SA;
    size_t i;
    if (n == 0) return ORACLE_OK;
    if (n >= 0x7ffffff0u) return fail(ORACLE_ERR_INVALID_ARGUMENT, "oracle: text too long");
    s = (idx_t *)malloc(sizeof(idx_t) * (n + 1));
    SA = (idx_t *)malloc(sizeof(idx_t) * (n + 1));
    if (!s || !SA) {
        free(s);
        free(SA);
        return fail(ORACLE_ERR_NOMEM, "oracle: out of memory");
    }
    for (i = 0; i < n; ++i) s[i] = (idx_t)text[i] + 1;
    s[n] = 0; /* the terminator construct_im appends */
    if (sais(s, SA, (idx_t)(n + 1), 256) != 0) {
        free(s);
        free(SA);
        return fail(ORACLE_ERR_NOMEM, "oracle: out of memory");
    }
    memcpy(sa, SA + 1, sizeof(idx_t) * n); /* SA[0] is the terminator suffix */
    free(s);
    free(SA);
    return ORACLE_OK;
}

int oracle_lcp_array(const uint8_t *text, size_t n, const int32_t *sa, int32_t *lcp) {
    int32_t *isa;
    size_t i, h = 0;
    if (n == 0) return ORACLE_OK;
    isa = (int32_t *)malloc(sizeof(int32_t) * n);
    if (!isa) return fail(ORACLE_ERR_NOMEM, "oracle: out of memory");
    for (i = 0; i < n; ++i) isa[sa[i]] = (int32_t)i;
    lcp[0] = 0;
    for (i = 0; i < n; ++i) {
        int32_t r = isa[i];
        if (r > 0) {
            size_t j = (size_t)sa[r - 1];
            while (i + h < n && j + h < n && text[i + h] == text[j + h]) ++h;
            lcp[r] = (int32_t)h;
            if (h > 0) --h;
        } else {
            h = 0;
        }
    }
    free(isa);
    return ORACLE_OK;
}

/* ------------------------------------------------------------------ */
/* LCP-interval tree = explicit internal nodes of the suffix tree.     */
/* ------------------------------------------------------------------ */
#define U32_INF 0xffffffffu

typedef struct {
    size_t n;            /* text length (without terminator) */
    const uint8_t *text; /* borrowed */
    int32_t *sa, *isa, *lcp;
    int32_t n_nodes;
    int32_t *depth;       /* string depth, cst.depth(v)  */
    int32_t *parent;      /* cst.parent(v); root = 0     */
    int32_t *leaf_parent; /* parent of leaf at rank r    */
    uint32_t *minA;       /* plain: min SA; RC: min fwd_starts   (rmq / rmqF) */
    uint32_t *minB;       /* RC only: min rc_ends                (rmqRcEnd)   */
} tree_t;

static void tree_free(tree_t *t) {
    free(t->sa);
    free(t->isa);
    free(t->lcp);
    free(t->depth);
    free(t->parent);
    free(t->leaf_parent);
    free(t->minA);
    free(t->minB);
    memset(t, 0, sizeof *t);
}

static inline uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

/* valA/valB give the per-rank values whose range minimum the reference
 * queries per node (valB may be NULL). */
static int tree_build(tree_t *t, const uint8_t *text, size_t n, int two_values,
                      uint32_t (*valA)(const tree_t *, size_t), uint32_t (*valB)(const tree_t *, size_t)) {
    size_t r;
    int32_t *stack = NULL, sp = 0;
    memset(t, 0, sizeof *t);
    t->n = n;
    t->text = text;
    t->sa = (int32_t *)malloc(sizeof(int32_t) * (n + 1));
    t->isa = (int32_t *)malloc(sizeof(int32_t) * (n + 1));
    t->lcp = (int32_t *)malloc(sizeof(int32_t) * (n + 1));
    t->depth = (int32_t *)malloc(sizeof(int32_t) * (n + 2));
    t->parent = (int32_t *)malloc(sizeof(int32_t) * (n + 2));
    t->leaf_parent = (int32_t *)malloc(sizeof(int32_t) * (n + 1));
    t->minA = (uint32_t *)malloc(sizeof(uint32_t) * (n + 2));
    t->minB = two_values ? (uint32_t *)malloc(sizeof(uint32_t) * (n + 2)) : NULL;
    stack = (int32_t *)malloc(sizeof(int32_t) * (n + 2));
    if (!t->sa || !t->isa || !t->lcp || !t->depth || !t->parent || !t->leaf_parent || !t->minA ||
        (two_values && !t->minB) || !stack) {
        free(stack);
        tree_free(t);
        return fail(ORACLE_ERR_NOMEM, "oracle: out of memory");
    }
    if (oracle_suffix_array(text, n, t->sa) != ORACLE_OK ||
        oracle_lcp_array(text, n, t->sa, t->lcp) != ORACLE_OK) {
        free(stack);
        tree_free(t);
        return ORACLE_ERR_NOMEM;
    }
    for (r = 0; r < n; ++r) t->isa[t->sa[r]] = (int32_t)r;

#define NEW_NODE(d)                                                                   \
    (t->depth[t->n_nodes] = (d), t->parent[t->n_nodes] = -1, t->minA[t->n_nodes] = U32_INF, \
     (two_values ? (t->minB[t->n_nodes] = U32_INF) : 0), t->n_nodes++)
#define ADD_LEAF(node, rank)                                                          \
    do {                                                                              \
        t->leaf_parent[rank] = (node);                                                \
        t->minA[node] = umin(t->minA[node], valA(t, rank));                           \
        if (two_values) t->minB[node] = umin(t->minB[node], valB(t, rank));           \
    } while (0)

    stack[sp++] = NEW_NODE(0); /* root */
    for (r = 1; r <= n; ++r) {
        int32_t cur = (r < n) ? t->lcp[r] : 0;
        int32_t prev = t->lcp[r - 1];
        if (cur < prev) ADD_LEAF(stack[sp - 1], r - 1);
        while (t->depth[stack[sp - 1]] > cur) {
            int32_t x = stack[--sp];
            int32_t y = stack[sp - 1];
            if (t->depth[y] < cur) { /* a new node of depth cur becomes x's parent */
                y = NEW_NODE(cur);
                stack[sp++] = y;
            }
            t->parent[x] = y;
            t->minA[y] = umin(t->minA[y], t->minA[x]);
            if (two_values) t->minB[y] = umin(t->minB[y], t->minB[x]);
        }
        if (t->depth[stack[sp - 1]] < cur) stack[sp++] = NEW_NODE(cur);
        if (cur >= prev) ADD_LEAF(stack[sp - 1], r - 1);
    }
#undef NEW_NODE
#undef ADD_LEAF
    free(stack);
    return ORACLE_OK;
}

/* min(lcp(text[a..], text[b..]), cap): string depth of the LCA of the two
 * leaves (factorizer_helpers.hpp:20-24), read off the text directly. */
static size_t lcp_capped(const tree_t *t, size_t a, size_t b, size_t cap) {
    size_t h = 0, lim = t->n - (a > b ? a : b);
    if (lim > cap) lim = cap;
    while (h < lim && t->text[a + h] == t->text[b + h]) ++h;
    return h;
}

/* ------------------------------------------------------------------ */
/* Plain mode: detail::nolzss, factorizer_core.hpp:51-119              */
/* ------------------------------------------------------------------ */
static uint32_t val_sa(const tree_t *t, size_t r) { return (uint32_t)t->sa[r]; }

typedef struct {
    int32_t *buf;
    size_t cap;
} chain_t;

static int chain_reserve(chain_t *c, size_t need) {
    if (need <= c->cap) return 0;
    size_t nc = c->cap ? c->cap * 2 : 64;
    while (nc < need) nc *= 2;
    int32_t *nb = (int32_t *)realloc(c->buf, nc * sizeof(int32_t));
    if (!nb) return -1;
    c->buf = nb;
    c->cap = nc;
    return 0;
}

/* Collect the explicit proper ancestors of leaf(i) below the root, deepest
 * first; the reference enumerates them shallowest first with
 * level_anc(lambda, node_depth - d), d = 1,2,... (factorizer_core.hpp:71). */
static long collect_ancestors(const tree_t *t, size_t i, chain_t *c) {
    long k = 0;
    int32_t v = t->leaf_parent[t->isa[i]];
    while (v != 0) {
        if (chain_reserve(c, (size_t)k + 1)) return -1;
        c->buf[k++] = v;
        v = t->parent[v];
    }
    return k;
}

/* One iteration of the outer while loop of detail::nolzss with the cursor at
 * lambda_sufnum = i.  u_min_io mirrors u_min_leaf_sufnum, which the reference
 * declares outside the loop (factorizer_core.hpp:62). */
static int plain_factor_at(const tree_t *t, size_t i, chain_t *c, uint64_t *u_min_io,
                           oracle_factor *f) {
    long k = collect_ancestors(t, i, c), d;
    uint64_t u_min = *u_min_io, u_depth = 0;
    int u_is_root = 1;
    if (k < 0) return -1;
    for (d = k - 1; d >= -1; --d) { /* d == -1: v is the leaf lambda itself */
        uint64_t v_min, l;
        if (d >= 0) {
            int32_t v = c->buf[d];
            v_min = t->minA[v];           /* cst.csa[rmq(lb(v), rb(v))]   :72 */
            l = (uint64_t)t->depth[v];    /* cst.depth(v)                 :73 */
        } else {
            v_min = i;
            l = (uint64_t)(t->n - i) + 1; /* leaf depth counts the terminator */
        }
        if (v_min + l - 1 < i) { /* :75 */
            u_min = v_min;
            u_depth = l;
            u_is_root = 0;
            continue;
        }
        if (v_min == i) { /* :82 */
            if (u_is_root) {
                f->start = i, f->length = 1, f->ref = i; /* :83-88 */
            } else {
                f->start = i, f->length = u_depth, f->ref = u_min; /* :89-94 */
            }
        } else {
            uint64_t cap = i - v_min; /* :96-97 */
            l = lcp_capped(t, i, (size_t)v_min, (size_t)cap);
            if (l <= u_depth) {
                f->start = i, f->length = u_depth, f->ref = u_min; /* :98-102 */
            } else {
                f->start = i, f->length = l, f->ref = v_min; /* :104-107 */
            }
        }
        break;
    }
    *u_min_io = u_min;
    return 0;
}

static int plain_run(const uint8_t *text, size_t n, size_t start_pos, oracle_factor **out,
                     size_t *z, int every_position, uint32_t *len_all, uint32_t *ref_all) {
    tree_t t;
    chain_t c = {0, 0};
    size_t i, count = 0, cap = 0;
    oracle_factor *fs = NULL;
    uint64_t u_min = 0;
    int rc;
    if (z) *z = 0;
    if (out) *out = NULL;
    if (start_pos > n) return fail(ORACLE_ERR_INVALID_ARGUMENT, "start_pos beyond end of text");
    if (n == 0 || start_pos == n) return ORACLE_OK;
    if ((rc = tree_build(&t, text, n, 0, val_sa, NULL)) != ORACLE_OK) return rc;
    i = start_pos;
    while (i < n) { /* factorizer_core.hpp:66 */
        oracle_factor f;
        if (plain_factor_at(&t, i, &c, &u_min, &f)) goto nomem;
        if (every_position) {
            len_all[i] = (uint32_t)f.length;
            ref_all[i] = (uint32_t)f.ref;
            ++i;
            continue;
        }
        if (out) {
            if (count == cap) {
                size_t nc = cap ? cap * 2 : 1024;
                oracle_factor *nf = (oracle_factor *)realloc(fs, nc * sizeof *fs);
                if (!nf) goto nomem;
                fs = nf;
                cap = nc;
            }
            fs[count] = f;
        }
        ++count;
        i += f.length; /* next_leaf(lambda, l)  :113 */
    }
    free(c.buf);
    tree_free(&t);
    if (out) *out = fs;
    if (z) *z = count;
    return ORACLE_OK;
nomem:
    free(c.buf);
    free(fs);
    tree_free(&t);
    return fail(ORACLE_ERR_NOMEM, "oracle: out of memory");
}

int oracle_factorize(const uint8_t *text, size_t n, size_t start_pos, oracle_factor **out,
                     size_t *z) {
    return plain_run(text, n, start_pos, out, z, 0, NULL, NULL);
}

int oracle_count_factors(const uint8_t *text, size_t n, size_t start_pos, size_t *z) {
    return plain_run(text, n, start_pos, NULL, z, 0, NULL, NULL);
}

int oracle_lpnf_all(const uint8_t *text, size_t n, uint32_t *len, uint32_t *ref) {
    return plain_run(text, n, 0, NULL, NULL, 1, len, ref);
}

/* ------------------------------------------------------------------ */
/* prepare_multiple_dna_sequences_w_rc, factorizer.cpp:54-172          */
/* ------------------------------------------------------------------ */
static int complement(uint8_t c, uint8_t *o) { /* factorizer.cpp:17-27 */
    switch (c) {
    case 'A': *o = 'T'; return 0;
    case 'C': *o = 'G'; return 0;
    case 'G': *o = 'C'; return 0;
    case 'T': *o = 'A'; return 0;
    default: return -1;
    }
}

/* get_sentinel lambda, factorizer.cpp:110-125: k-th value of 1,2,3,...
 * skipping 0,'A','C','G','T'; a signed char wraps 127 -> -128 ... -1 -> 0 -> 1. */
static uint8_t sentinel_for(size_t index) {
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

int oracle_prepare_multiple_dna_w_rc(const char *const *seqs, const size_t *lens, size_t k,
                                     uint8_t **S_out, size_t *S_len, size_t *original_length,
                                     uint64_t **sentinel_pos, size_t *n_sentinels) {
    size_t i, j, non_empty = 0, empty = 0, total = 0, pos = 0, sidx = 0;
    uint8_t *S;
    uint64_t *sp;
    *S_out = NULL, *S_len = 0, *original_length = 0, *sentinel_pos = NULL, *n_sentinels = 0;
    if (k == 0) return ORACLE_OK; /* :55-57 */
    for (i = 0; i < k; ++i) (lens[i] ? ++non_empty : ++empty);
    if (empty)
        fprintf(stderr, "Warning: Skipping %zu empty sequence(s) in prepare_multiple_dna_sequences_w_rc\n",
                empty); /* :70-72 */
    if (non_empty == 0)
        return fail(ORACLE_ERR_RUNTIME, "All sequences are empty - cannot prepare for factorization");
    if (non_empty > 125)
        return fail(ORACLE_ERR_INVALID_ARGUMENT,
                    "Too many sequences: maximum 125 sequences supported (due to sentinel character limitations)");
    for (i = 0; i < k; ++i)
        for (j = 0; j < lens[i]; ++j) {
            char ch = seqs[i][j];
            if (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T' && ch != 'a' && ch != 'c' &&
                ch != 'g' && ch != 't') {
                snprintf(g_err, sizeof g_err, "Invalid nucleotide '%c' found in sequence %zu", ch, i);
                return ORACLE_ERR_RUNTIME; /* :86-95 */
            }
        }
    for (i = 0; i < k; ++i) total += 2 * lens[i];
    total += 2 * non_empty;
    S = (uint8_t *)malloc(total ? total : 1);
    sp = (uint64_t *)malloc(sizeof(uint64_t) * 2 * non_empty);
    if (!S || !sp) {
        free(S);
        free(sp);
        return fail(ORACLE_ERR_NOMEM, "oracle: out of memory");
    }
    for (i = 0; i < k; ++i) { /* :128-147 */
        if (!lens[i]) continue;
        for (j = 0; j < lens[i]; ++j) {
            uint8_t ch = (uint8_t)seqs[i][j];
            if (ch >= 'a' && ch <= 'z') ch = (uint8_t)(ch - 'a' + 'A');
            S[pos++] = ch;
        }
        sp[sidx] = pos;
        S[pos++] = sentinel_for(sidx);
        ++sidx;
    }
    *original_length = pos;
    for (i = k; i-- > 0;) { /* :150-169 */
        if (!lens[i]) continue;
        for (j = 0; j < lens[i]; ++j) {
            uint8_t ch = (uint8_t)seqs[i][lens[i] - 1 - j], o = 0;
            if (ch >= 'a' && ch <= 'z') ch = (uint8_t)(ch - 'a' + 'A');
            complement(ch, &o);
            S[pos++] = o;
        }
        sp[sidx] = pos;
        S[pos++] = sentinel_for(sidx);
        ++sidx;
    }
    *S_out = S;
    *S_len = pos;
    *sentinel_pos = sp;
    *n_sentinels = sidx;
    return ORACLE_OK;
}

/* ------------------------------------------------------------------ */
/* RC mode: detail::nolzss_multiple_dna_w_rc, factorizer_core.hpp:177-383 */
/* ------------------------------------------------------------------ */
static size_t g_rc_N; /* set per call; tree_build callbacks read it (single-threaded use) */

static uint32_t val_fwd(const tree_t *t, size_t r) { /* fwd_starts[k]  :221-223 */
    size_t posS = (size_t)t->sa[r];
    return posS < g_rc_N ? (uint32_t)posS : U32_INF;
}
static uint32_t val_rcend(const tree_t *t, size_t r) { /* rc_ends[k]  :224-229 */
    size_t posS = (size_t)t->sa[r];
    size_t R_beg = g_rc_N + 1, R_end = t->n - 1;
    if (posS >= R_beg && posS < R_end) {
        size_t jR0 = posS - R_beg;
        return (uint32_t)(g_rc_N - jR0 - 1);
    }
    return U32_INF;
}

static int rc_factor_at(const tree_t *t, size_t N, size_t i, chain_t *c, oracle_factor *f) {
    long k = collect_ancestors(t, i, c), d;
    int have_fwd = 0, have_rc = 0;
    uint64_t best_fwd_start = 0, best_fwd_depth = 0;
    uint64_t best_rc_end = 0, best_rc_posS = 0, best_rc_depth = 0;
    uint64_t fwd_true_len = 0, rc_true_len = 0, emit_len = 1, emit_ref = i;
    int use_fwd = 0, use_literal = 0;
    if (k < 0) return -1;
    for (d = k - 1; d >= -1; --d) { /* :256-300, shallowest ancestor first, leaf last */
        uint64_t ell, jF, endRC;
        int okF, okR;
        if (d >= 0) {
            int32_t v = c->buf[d];
            ell = (uint64_t)t->depth[v];
            jF = t->minA[v];
            endRC = t->minB[v];
        } else {
            ell = (uint64_t)(t->n - i) + 1;
            jF = i;          /* i < N: the leaf's own fwd_starts entry */
            endRC = U32_INF; /* and it has no rc_ends entry            */
        }
        if (ell == 0) break;                                   /* :259 */
        okF = (jF != U32_INF) && (jF + ell - 1 < i);           /* :266 */
        okR = (endRC != U32_INF) && (endRC < i);               /* :271 */
        if (!okF && !okR) break;                               /* :273-277 */
        if (okF) {                                             /* :280-287 */
            if (ell > best_fwd_depth ||
                (ell == best_fwd_depth && (jF + ell - 1) < (best_fwd_start + best_fwd_depth - 1))) {
                best_fwd_depth = ell;
                best_fwd_start = jF;
                have_fwd = 1;
            }
        }
        if (okR) {                                             /* :290-299 */
            if (ell > best_rc_depth || (ell == best_rc_depth && endRC < best_rc_end)) {
                best_rc_depth = ell;
                best_rc_end = endRC;
                best_rc_posS = (N + 1) + (N - 1 - endRC); /* cst.csa[kR]: inverse of :226-228 */
                have_rc = 1;
            }
        }
    }
    if (!have_fwd && !have_rc) { /* :305-316 */
        f->start = i, f->length = 1, f->ref = i;
        return 0;
    }
    if (have_fwd) { /* :322-326 */
        uint64_t cap = i - best_fwd_start;
        fwd_true_len = lcp_capped(t, i, (size_t)best_fwd_start, (size_t)cap);
    }
    if (have_rc) /* :328-330 */
        rc_true_len = lcp_capped(t, i, (size_t)best_rc_posS, (size_t)-1);
    if (have_fwd && fwd_true_len >= 1) { /* :338-352 */
        use_fwd = !(have_rc && rc_true_len > fwd_true_len);
    } else {
        if (have_rc && rc_true_len > 1)
            use_fwd = 0;
        else
            use_literal = 1;
    }
    if (use_literal) { /* :354-365 */
        emit_len = 1;
        emit_ref = i;
    } else if (use_fwd) {
        emit_len = fwd_true_len;
        emit_ref = best_fwd_start;
    } else {
        emit_len = rc_true_len;
        emit_ref = ORACLE_RC_MASK | (best_rc_end - emit_len + 1);
    }
    if (emit_len == 0) return -2; /* :368-370 */
    f->start = i, f->length = emit_len, f->ref = emit_ref;
    return 0;
}

static int rc_run(const uint8_t *S, size_t S_len, size_t start_pos, oracle_factor **out,
                  size_t *z, int every_position, uint32_t *len_all, uint64_t *ref_all) {
    tree_t t;
    chain_t c = {0, 0};
    size_t N, i, count = 0, cap = 0;
    oracle_factor *fs = NULL;
    int rc;
    if (z) *z = 0;
    if (out) *out = NULL;
    if (S_len == 0) return ORACLE_OK; /* :180 */
    if (S_len < 4) {                  /* :189-193 */
        fprintf(stderr,
                "Warning: Input string too short for factorization with reverse complement (size=%zu). Returning 0 factors.\n",
                S_len);
        return ORACLE_OK;
    }
    N = S_len / 2 - 1; /* :195 */
    if (N == 0) return ORACLE_OK;
    if (start_pos >= N) /* :203-205 */
        return fail(ORACLE_ERR_INVALID_ARGUMENT,
                    "start_pos must be less than the original sequence length");
    g_rc_N = N;
    if ((rc = tree_build(&t, S, S_len, 1, val_fwd, val_rcend)) != ORACLE_OK) return rc;
    i = start_pos;
    while (i < N) { /* :241 */
        oracle_factor f;
        int e = rc_factor_at(&t, N, i, &c, &f);
        if (e == -1) goto nomem;
        if (e == -2) {
            free(c.buf);
            free(fs);
            tree_free(&t);
            return fail(ORACLE_ERR_RUNTIME, "emit_len must be positive to ensure factorization progress");
        }
        if (every_position) {
            len_all[i] = (uint32_t)f.length;
            ref_all[i] = f.ref;
            ++i;
            continue;
        }
        if (out) {
            if (count == cap) {
                size_t nc = cap ? cap * 2 : 1024;
                oracle_factor *nf = (oracle_factor *)realloc(fs, nc * sizeof *fs);
                if (!nf) goto nomem;
                fs = nf;
                cap = nc;
            }
            fs[count] = f;
        }
        ++count;
        i += f.length; /* :377-379 */
    }
    free(c.buf);
    tree_free(&t);
    if (out) *out = fs;
    if (z) *z = count;
    return ORACLE_OK;
nomem:
    free(c.buf);
    free(fs);
    tree_free(&t);
    return fail(ORACLE_ERR_NOMEM, "oracle: out of memory");
}

int oracle_factorize_multiple_dna_w_rc(const uint8_t *S, size_t S_len, size_t start_pos,
                                       oracle_factor **out, size_t *z) {
    return rc_run(S, S_len, start_pos, out, z, 0, NULL, NULL);
}

int oracle_count_factors_multiple_dna_w_rc(const uint8_t *S, size_t S_len, size_t start_pos,
                                           size_t *z) {
    return rc_run(S, S_len, start_pos, NULL, z, 0, NULL, NULL);
}

int oracle_lpnf_all_rc(const uint8_t *S, size_t S_len, uint32_t *len, uint64_t *ref) {
    return rc_run(S, S_len, 0, NULL, NULL, 1, len, ref);
}

/* detail::nolzss_dna_w_rc, factorizer_core.hpp:140-151 */
int oracle_factorize_dna_w_rc(const uint8_t *text, size_t n, oracle_factor **out, size_t *z) {
    const char *seqs[1];
    size_t lens[1], S_len, orig, ns;
    uint8_t *S;
    uint64_t *sp;
    int rc;
    *out = NULL;
    *z = 0;
    if (n == 0) return ORACLE_OK; /* :143 */
    seqs[0] = (const char *)text;
    lens[0] = n;
    rc = oracle_prepare_multiple_dna_w_rc(seqs, lens, 1, &S, &S_len, &orig, &sp, &ns);
    if (rc != ORACLE_OK) return rc;
    rc = rc_run(S, S_len, 0, out, z, 0, NULL, NULL);
    free(S);
    free(sp);
    return rc;
}
