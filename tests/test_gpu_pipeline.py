"""GPU parity of the plain factorize path against the oracle: intermediate arrays
(SA, LCP, L*) and final factors, through the C ABI (nolzss_amd._noLZSS -> libnolzss_hip.so)."""
import json
import random
from pathlib import Path

import numpy as np
import pytest

import gen
import oracle_lib as oracle

pytestmark = pytest.mark.gpu

KATS = json.loads((Path(__file__).parent / "golden" / "kats.json").read_text())


@pytest.fixture(scope="module")
def native():
    from nolzss_amd import _noLZSS
    assert _noLZSS.device_count() >= 1, "no MI355X visible"
    return _noLZSS


def _text(v):
    if "input" in v:
        return v["input"].encode("ascii")
    s, k = v["input_repeat"]
    return s.encode("ascii") * k


def _cases():
    rng = random.Random(99)
    cases = {
        "abracadabra": b"abracadabra",
        "abracadabra_x1000": b"abracadabra" * 1000,
        "single": b"A",
        "two": b"AA",
        "aaaa_5000": b"a" * 5000,
        "ab_period": b"ab" * 3000 + b"b",
        "acgt_period7": (b"ACGTTGA" * 2000)[:13001],
        "dna_100": gen.random_dna(100, 1).tobytes(),
        "dna_5000": gen.random_dna(5000, 2).tobytes(),
        "dna_70k": gen.random_dna(70_000, 3).tobytes(),
        "dna_1M": gen.random_dna(1 << 20, 4).tobytes(),
        "repeat_300k": gen.repeat_dna(300_000, 5, lo=16, hi=4096).tobytes(),
        "binary_20k": bytes(rng.choice(b"01") for _ in range(20_000)),
        "protein_30k": bytes(rng.choice(b"ACDEFGHIKLMNPQRSTVWY") for _ in range(30_000)),
        "bytes_50k": bytes(rng.randrange(1, 256) for _ in range(50_000)),
        "bytes_with_nul_tail": bytes(rng.randrange(1, 256) for _ in range(999)) + b"\x00",
        "all256": bytes(range(256)) * 40,
        "fib": None,
    }
    # families of 40 and of 64 near-identical copies: every group of the key sort has 33..64
    # members, so the pair list of the direct comparison round overflows (groups left to the
    # doubling rounds) and the ones that fit run many pairs per member
    blk = gen.random_dna(6000, 11)
    fam = []
    for c in range(40):
        b = blk.copy()
        b[(c * 131) % 6000::997] = ord("ACGT"[c % 4])
        fam.append(b.tobytes())
    cases["family_of_40_with_edits"] = b"".join(fam)
    cases["family_of_64_exact"] = gen.random_dna(3000, 12).tobytes() * 64
    cases["family_of_65_exact"] = gen.random_dna(2000, 13).tobytes() * 65
    # mixture: group sizes 2..20 side by side with singletons (balanced pair list, several rounds)
    parts = []
    r2 = random.Random(5)
    base = [gen.random_dna(700, 20 + k).tobytes() for k in range(30)]
    for k in range(400):
        b = bytearray(base[r2.randrange(30)] if r2.random() < 0.7 else gen.random_dna(700, 1000 + k).tobytes())
        if r2.random() < 0.5:
            b[r2.randrange(700)] = ord("ACGT"[r2.randrange(4)])
        parts.append(bytes(b))
    cases["mixed_families_280k"] = b"".join(parts)
    a, b = b"a", b"ab"
    while len(b) < 40_000:
        a, b = b, b + a
    cases["fib"] = b
    return cases


CASES = _cases()


@pytest.mark.parametrize("name", list(CASES))
def test_intermediate_arrays(native, name):
    t = CASES[name]
    n = len(t)
    d = native.debug_arrays(t)
    sa = oracle.suffix_array(t)
    assert np.array_equal(d["sa"].astype(np.int64), sa.astype(np.int64)), "suffix array"
    isa = np.empty(n, dtype=np.int64)
    isa[sa] = np.arange(n)
    assert np.array_equal(d["isa"].astype(np.int64), isa), "inverse suffix array"
    lcp = oracle.lcp_array(t, sa)
    assert np.array_equal(d["lcp"][:n].astype(np.int64), lcp.astype(np.int64)), "LCP"
    assert d["lcp"][n] == 0
    ln, _ = oracle.lpnf_all(t)
    got = d["lstar"].astype(np.int64)
    got_len = np.where(got == 0, 1, got)
    assert np.array_equal(got_len, ln.astype(np.int64)), "L* (per-position factor length)"


@pytest.mark.parametrize("name", list(CASES))
def test_factorize_matches_oracle(native, name):
    t = CASES[name]
    got = native.factorize_array(t)
    exp = oracle.factors_array(t)
    assert len(got) == len(exp)
    assert np.array_equal(got["start"], exp["start"])
    assert np.array_equal(got["length"], exp["length"])
    assert np.array_equal(got["ref"], exp["ref"])
    assert native.count_factors(t) == len(exp)


@pytest.mark.parametrize("v", KATS["plain"] + KATS["derived_plain"], ids=lambda v: v["source"][:40])
def test_reference_kats(native, v):
    assert native.factorize(_text(v)) == [tuple(f) for f in v["factors"]]


def test_start_pos(native):
    t = CASES["repeat_300k"]
    for sp in (0, 1, 4095, 4096, 123_457, len(t) - 1):
        got = native.factorize_array(t, start_pos=sp)
        exp = oracle.factors_array(t, start_pos=sp)
        assert np.array_equal(got["start"], exp["start"]) and np.array_equal(got["ref"], exp["ref"])
        assert np.array_equal(got["length"], exp["length"])
    assert len(native.factorize_array(t, start_pos=len(t))) == 0


def test_many_random_small(native):
    rng = random.Random(5)
    for _ in range(150):
        n = rng.randint(1, 300)
        alpha = rng.choice([b"A", b"AC", b"ACGT", b"abcdefghijklmnopqrstuvwxyz"])
        t = bytes(rng.choice(alpha) for _ in range(n))
        assert native.factorize(t) == oracle.factorize(t), t


@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097, 8191, 8192, 8193,
                               65535, 65536, 65537, (1 << 20) - 1, 1 << 20, (1 << 20) + 1])
def test_sizes_around_tile_boundaries(native, n):
    """tile sizes of the kernels are 256, 1024, 4096; 2^20 is where the bucketed key sort starts"""
    t = gen.repeat_dna(n, 40 + n % 7, lo=8, hi=512).tobytes()
    assert native.factorize(t) == oracle.factorize(t)


@pytest.mark.parametrize("extra_env", [{}, {"NOLZSS_LOCAL_SORT_MIN": "1"},
                                       {"NOLZSS_LOCAL_SORT_MIN": "1", "NOLZSS_TEST_LOCAL_ORDER_FAILS": "1"},
                                       {"NOLZSS_LOCAL_SORT_MIN": "1", "NOLZSS_TEST_LOCAL_LOOKBACK_FAILS": "1"},
                                       {"NOLZSS_LOCAL_SORT_MIN": "1", "NOLZSS_NO_LOCAL_REGROUP": "1"}],
                         ids=["segmented-passes", "sub-buckets-in-lds", "sub-buckets-redone", "regroup-falls-back", "regroup-kernel"])
def test_bucketed_key_sort_on_small_texts(extra_env):
    """The bucketed (most-significant-digit first) key sort normally starts at 2^20 bases; one child
    process with NOLZSS_DNA_FAST_MIN=1 runs it on small DNA cases: empty buckets, buckets smaller
    than a tile, suffixes near the end of the text.  NOLZSS_LOCAL_SORT_MIN=1 sends the same cases through
    the form for 2^28 bases and more -- two most-significant-digit passes, the 65 536 sub-buckets sorted
    in LDS (local_sort_kernel) --, with sub-buckets beyond a workgroup's capacity (a run of 30 000 A's, a
    period-4 text) on the list for the segmented passes; NOLZSS_TEST_LOCAL_ORDER_FAILS pretends the
    kernel's lane-order check failed, so every sub-bucket is redone by those passes.  Where no sub-bucket
    overflows, that kernel also does the regroup of round 0 (heads, LCP from the keys, tied elements with a
    look-back across the sub-buckets): NOLZSS_TEST_LOCAL_LOOKBACK_FAILS raises the flag a look-back sets
    when it gives up (the text is sorted again by the plain kernel), NOLZSS_NO_LOCAL_REGROUP takes the
    plain kernel and the regroup kernel from the start."""
    import os
    import subprocess
    import sys
    code = r'''
import sys
sys.path.insert(0, "tests")
import numpy as np
import gen, oracle_lib as oracle
from nolzss_amd import _noLZSS as native
sizes = [1, 2, 3, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097, 8191, 8192, 8193, 65535, 65536, 65537]
cases = [gen.repeat_dna(m, 40 + m % 7, lo=8, hi=512).tobytes() for m in sizes] + [b"ACGT", b"A" * 5000, (b"ACGTTGA" * 2000)[:13001], gen.random_dna(100, 1).tobytes(),
         gen.random_dna(70_000, 3).tobytes(), gen.repeat_dna(300_000, 5, lo=16, hi=4096).tobytes(),
         b"AC" * 40000 + b"G", gen.random_dna(3000, 12).tobytes() * 64, b"T" * 4097 + gen.random_dna(5000, 9).tobytes(),
         b"A" * 30000 + gen.random_dna(20000, 4).tobytes(), b"ACGT" * 25000 + b"T"]
for t in cases:
    got = native.factorize_array(t)
    exp = oracle.factors_array(t)
    assert len(got) == len(exp), (len(t), len(got), len(exp))
    for k in ("start", "length", "ref"):
        assert np.array_equal(got[k], exp[k]), (len(t), k)
# segmented texts on the same sort: many sequences with IDENTICAL ends (copies of a short suffix at
# up to 124 terminators tie on the key and must come out in terminator order), reverse complement
import random
rng = random.Random(3)
tail = "ACGTTGCAAGGCTA"
groups = [["ACGT" * 3 + tail] * 40,
          ["".join(rng.choice("ACGT") for _ in range(rng.randint(1, 60))) + tail[-rng.randint(1, 14):] for _ in range(62)],
          ["A", "A", "AA", "AAA", "C", "CA", "AAAA"] * 8,
          ["".join(rng.choice("ACGT") for _ in range(300)) for _ in range(5)]]
for seqs in groups:
    S, orig, sent = native.prepare_multiple_dna_sequences_w_rc_bytes(seqs)
    assert native.factorize_multiple_dna_w_rc(S) == oracle.factorize_multiple_dna_w_rc(S), len(seqs)
    S2, orig2, sent2 = native.prepare_multiple_dna_sequences_no_rc_bytes(seqs) if hasattr(native, "prepare_multiple_dna_sequences_no_rc_bytes") else (None, None, None)
    if S2 is not None:
        assert native.factorize(S2) == oracle.factorize(S2), len(seqs)
print("ok", len(cases))
'''
    # (NOLZSS_TEST_INJECT_PENDING: one LCP entry per text is left "pending" on purpose, so the safety
    # net that compares those suffixes in the packed text runs as well)
    # (NOLZSS_LOCAL_REGROUP_MIN=1: the regroup on the way normally starts at 3 * 2^28 bases)
    env = dict(os.environ, NOLZSS_DNA_FAST_MIN="1", NOLZSS_TEST_INJECT_PENDING="1", NOLZSS_LOCAL_REGROUP_MIN="1", **extra_env)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_pair_runs_on_long_exact_repeats():
    """Long exact repeats leave pairs of suffixes tied beyond the cap of the direct round; along runs of
    text positions they are finished arithmetically (suffix_array.hip, pair_delta_kernel ...).  One child
    process with a tiny cap (NOLZSS_REFINE_WORDS=1: 49 symbols) and NOLZSS_PAIR_RUNS_MIN=1 sends every
    repeat longer than that through this path: two and three copies, copies with substitutions, nested and
    overlapping repeats, repeats that end at the end of the text, other alphabets, reverse complement."""
    import os
    import subprocess
    import sys
    code = r'''
import sys
sys.path.insert(0, "tests")
import numpy as np
import gen, oracle_lib as oracle
from nolzss_amd import _noLZSS as native
rng = np.random.default_rng(5)
def rnd(n, seed): return gen.random_dna(n, seed)
def mutate(x, k, seed):
    y = x.copy(); r = np.random.default_rng(seed)
    idx = r.integers(0, len(y), size=k); y[idx] = np.frombuffer(b"ACGT", dtype=np.uint8)[r.integers(0, 4, size=k)]
    return y
a, b, c = rnd(5000, 1), rnd(3000, 2), rnd(777, 3)
cases = [
    np.concatenate([a, a]),                                  # two copies, the repeat ends with the text
    np.concatenate([a, b, a]),                               # two copies apart
    np.concatenate([a, a, a]),                               # three copies: groups of three (left to doubling)
    np.concatenate([a, b, a, c, a[:2000], b]),               # nested / partial copies
    np.concatenate([a, mutate(a, 5, 9)]),                    # a copy with substitutions: several runs
    np.concatenate([a, mutate(a, 60, 10), mutate(a, 3, 11)]),
    np.concatenate([a[:100], a[:100]]),                      # repeats barely longer than the cap
    np.concatenate([a[:60], c, a[:60], c[:50], a[:60]]),
    np.concatenate([c, c, b, b, c]),
    np.tile(a[:300], 7),                                     # period 300
    gen.repeat_dna(200000, 77, lo=64, hi=20000),
    np.concatenate([rnd(100000, 4), rnd(100000, 4)]),
    np.concatenate([rnd(70000, 5), mutate(rnd(70000, 5), 40, 6), rnd(1000, 7)]),
]
texts = [bytes(x) for x in cases]
texts += [b"abracadabra, " * 300 + b"simsalabim" * 100 + b"abracadabra, " * 300,   # 8-bit alphabet
          bytes(np.concatenate([rnd(4000, 8), rnd(4000, 8)]) + 1)]               # other symbols, sigma = 4
for t in texts:
    got = native.factorize_array(t)
    exp = oracle.factors_array(t)
    assert len(got) == len(exp), (len(t), len(got), len(exp))
    for k in ("start", "length", "ref"):
        assert np.array_equal(got[k], exp[k]), (len(t), k)
    d = native.debug_arrays(t)
    sa = oracle.suffix_array(t)
    assert np.array_equal(d["sa"].astype(np.int64), sa.astype(np.int64)), len(t)
    assert np.array_equal(d["lcp"][:len(t)].astype(np.int64), oracle.lcp_array(t, sa).astype(np.int64)), len(t)
    isa = np.empty(len(t), dtype=np.int64); isa[sa] = np.arange(len(t))
    assert np.array_equal(d["isa"].astype(np.int64), isa), len(t)
# reverse complement: the prepared string holds every sequence twice
for seqs in ([bytes(a)], [bytes(a), bytes(a)], [bytes(a[:3000]), bytes(b), bytes(a[:3000])]):
    S, orig, sent = native.prepare_multiple_dna_sequences_w_rc_bytes(seqs)
    assert native.factorize_multiple_dna_w_rc(S) == oracle.factorize_multiple_dna_w_rc(S), len(seqs)
# the merged batch: independent records that are copies of each other or hold repeats
recs = [a, a.copy(), np.concatenate([b, b]), np.concatenate([c, a[:1000], c]), b]
counts, arrays = native.factorize_batch(recs, want_factors=True)
for r, f in zip(recs, arrays):
    e = oracle.factors_array(r)
    assert len(f) == len(e) and all(np.array_equal(f[k], e[k]) for k in ("start", "length", "ref"))
print("ok", len(texts))
'''
    env = dict(os.environ, NOLZSS_REFINE_WORDS="1", NOLZSS_PAIR_RUNS_MIN="1", NOLZSS_TRACE="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    done = [int(line.split("pair runs:")[1].split()[0]) for line in r.stderr.splitlines() if "pair runs:" in line]
    assert len(done) >= 15 and sum(done) > 100000, done  # the path ran and finished pairs


@pytest.mark.parametrize("copies,base_len", [(5, 40_000), (17, 12_000), (20, 40_000), (24, 30_000), (33, 6_000), (65, 4_000),
                                             (130, 2_500), (300, 1_200)])
def test_collections_of_similar_genomes(native, copies, base_len):
    """k genomes a few substitutions apart (the reference's use case for whole collections,
    /root/reference/src/cpp/fasta_processor.cpp:298-341): nearly every suffix ties with k - 1 others for hundreds
    to thousands of symbols.  Groups of up to 64 members go through the pair comparisons of the direct round, what
    it leaves and every larger group through the doubling rounds -- groups of 65 .. 1024 members sorted in LDS
    (mid_sort_kernel), larger ones by the radix sort; with 20 and 24 copies a part of the groups does not fit the
    pair list of the direct round and goes through the equalising round first (suffix_array.hip).  Factors, suffix
    array, LCP and inverse against the oracle."""
    rng = np.random.default_rng(1000 + copies)
    base = gen.random_dna(base_len, 500 + copies)
    parts = [base]
    for _ in range(copies - 1):
        y = base.copy()
        idx = rng.integers(0, len(y), size=max(1, len(y) // 1000))
        y[idx] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=len(idx))]
        parts.append(y)
    t = bytes(np.concatenate(parts))
    got = native.factorize_array(t)
    exp = oracle.factors_array(t)
    assert len(got) == len(exp)
    for k in ("start", "length", "ref"):
        assert np.array_equal(got[k], exp[k]), k
    d = native.debug_arrays(t)
    sa = oracle.suffix_array(t)
    assert np.array_equal(d["sa"].astype(np.int64), sa.astype(np.int64))
    assert np.array_equal(d["lcp"][:len(t)].astype(np.int64), oracle.lcp_array(t, sa).astype(np.int64))
    isa = np.empty(len(t), dtype=np.int64)
    isa[sa] = np.arange(len(t))
    assert np.array_equal(d["isa"].astype(np.int64), isa)


@pytest.mark.parametrize("depth", ["2048", "96"])
def test_pivot_rounds(depth):
    """Pivot rounds (group_sort.hpp, kPivot): every tied group is sorted against the first member of its segment.
    One child process with NOLZSS_PIVOT_MIN=1 (every text that leaves ties behind the first direct round takes them)
    and the cap of the first round lowered to 49 symbols: collections of 3 .. 300 similar sequences (groups in every
    size class of the kernels: tiles of small groups, one workgroup per group of up to 128 / 256 / 1024 members),
    exact copies (segments that agree up to the cap stay tied for the doubling rounds; NOLZSS_PIVOT_DEPTH=96 makes
    that the common case), suffixes that end inside a comparison (copies at the end of the text, prepared strings with
    sentinels, reverse complement), runs (groups beyond the kernels' reach stay as they are), other alphabets.
    Factors, suffix array, LCP and inverse suffix array against the oracle."""
    import os
    import subprocess
    import sys
    code = r'''
import sys
sys.path.insert(0, "tests")
import numpy as np
import gen, oracle_lib as oracle
from nolzss_amd import _noLZSS as native
def rnd(n, seed): return gen.random_dna(n, seed)
def mutate(x, k, seed):
    y = x.copy(); r = np.random.default_rng(seed)
    idx = r.integers(0, len(y), size=k); y[idx] = np.frombuffer(b"ACGT", dtype=np.uint8)[r.integers(0, 4, size=k)]
    return y
def collection(copies, base_len, per, seed):
    base = rnd(base_len, seed)
    return np.concatenate([base] + [mutate(base, max(1, base_len // per), seed * 100 + k) for k in range(copies - 1)])
a, b, c = rnd(5000, 1), rnd(3000, 2), rnd(777, 3)
cases = [
    collection(3, 20000, 1000, 11), collection(5, 20000, 300, 12), collection(17, 6000, 1000, 13),
    collection(40, 3000, 500, 14), collection(70, 2000, 1000, 15), collection(100, 1500, 200, 16),
    collection(140, 1200, 1000, 17), collection(300, 700, 500, 18), collection(600, 300, 300, 19),
    collection(1100, 150, 100, 20),                             # groups beyond 1024 members: left alone
    np.concatenate([a, a, a, a]),                               # exact copies: tied up to the cap
    np.concatenate([a, b, a, c, a[:2000], b, a[:2000]]),        # copies that end where the text ends
    np.concatenate([collection(20, 1000, 100, 21), np.tile(a[:7], 400), collection(9, 900, 50, 22)]),
    np.tile(a[:300], 40),                                       # period 300: 40 copies, every suffix in a group
    np.concatenate([np.tile(a[:31], 200), b, np.tile(a[:31], 150)]),
    gen.repeat_dna(150000, 78, lo=64, hi=9000),
]
texts = [bytes(x) for x in cases]
texts += [b"abracadabra, " * 300 + b"simsalabim" * 100 + b"abracadabra, " * 300,       # 8-bit alphabet
          bytes(collection(30, 800, 100, 23) + 1),                                       # other symbols, sigma = 4
          bytes(np.concatenate([np.frombuffer(b"ACGTNRYK", dtype=np.uint8)[np.random.default_rng(5).integers(0, 8, 3000)]] * 12))]  # 4-bit
for t in texts:
    got = native.factorize_array(t)
    exp = oracle.factors_array(t)
    assert len(got) == len(exp), (len(t), len(got), len(exp))
    for k in ("start", "length", "ref"):
        assert np.array_equal(got[k], exp[k]), (len(t), k)
    d = native.debug_arrays(t)
    sa = oracle.suffix_array(t)
    assert np.array_equal(d["sa"].astype(np.int64), sa.astype(np.int64)), len(t)
    assert np.array_equal(d["lcp"][:len(t)].astype(np.int64), oracle.lcp_array(t, sa).astype(np.int64)), len(t)
    isa = np.empty(len(t), dtype=np.int64); isa[sa] = np.arange(len(t))
    assert np.array_equal(d["isa"].astype(np.int64), isa), len(t)
# prepared strings: sentinels between the sequences, with and without reverse complement
fam = [bytes(mutate(a[:2500], 3, 40 + k)) for k in range(9)]
for seqs in (fam, fam[:3], [bytes(a[:3000]), bytes(b), bytes(a[:3000]), bytes(a[:1500])]):
    S, orig, sent = native.prepare_multiple_dna_sequences_w_rc_bytes(seqs)
    assert native.factorize_multiple_dna_w_rc(S) == oracle.factorize_multiple_dna_w_rc(S), len(seqs)
print("ok", len(texts))
'''
    env = dict(os.environ, NOLZSS_REFINE_WORDS="1", NOLZSS_PIVOT_MIN="1", NOLZSS_PIVOT_DEPTH=depth, NOLZSS_TRACE="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    done = [int(line.split("pivot rounds:")[1].split()[0]) for line in r.stderr.splitlines() if "pivot rounds:" in line]
    assert len(done) >= 15 and sum(done) > 100000, done  # the rounds ran and finished suffixes


def test_periodic_runs_pass():
    """Runs of a short period tie the suffixes of a run in groups that only log2(run length) doubling
    rounds would resolve; the periodic-run pass orders them arithmetically (suffix_array.hip, "Periodic
    runs").  One child process with a tiny direct-round cap (NOLZSS_REFINE_WORDS=1: 49 symbols) and
    NOLZSS_PAIR_RUNS_MIN=1 sends every run longer than that through the pass: single runs, several runs of
    the same and of different periods, runs that break upwards / downwards / at the end of the text, runs
    inside random text, periods longer than the depth compared so far (with runs that only look alike), Fibonacci words, other alphabets, reverse complement.  SA, ISA, LCP and the factors
    are compared with the oracle."""
    import os
    import subprocess
    import sys
    code = r'''
import sys
sys.path.insert(0, "tests")
import numpy as np
import gen, oracle_lib as oracle
from nolzss_amd import _noLZSS as native
rng = np.random.default_rng(9)
def R(k, seed): return gen.random_dna(k, seed).tobytes()
def fib(k):
    a, b = b"A", b"AC"
    while len(b) < k: a, b = b, b + a
    return b[:k]
texts = [
    b"A" * 5000, b"AC" * 3000, b"ACG" * 2500 + b"T", (b"ACGTTGA" * 1500)[:9001], b"A" * 3000 + b"C" + b"A" * 2000,
    b"A" * 2000 + b"C" + b"A" * 3000, b"AC" * 2000 + b"AA" + b"AC" * 1500, R(500, 1) + b"A" * 4000 + R(500, 2),
    R(300, 3) + b"AC" * 1200 + R(200, 4) + b"AC" * 2100 + b"G" + b"CA" * 900 + R(100, 5),
    R(100, 6) + b"T" * 1000 + R(50, 7) + b"T" * 1500 + R(50, 8) + b"T" * 1200 + R(70, 9) + b"T" * 999,
    (R(23, 10) * 300) + R(40, 11) + (R(23, 10) * 200), (R(97, 12) * 90), (R(97, 12) * 90)[:-13] + b"G",
    fib(20000), fib(6765) + b"C" + fib(4181), b"abcab" * 2000, b"ab" * 3000 + b"c" + b"ab" * 2500 + b"a",
    bytes(range(65, 75)) * 800, (b"AAC" * 2000 + b"AAT" * 2000) * 2, b"G" * 7000 + b"A",
    b"ACGT" * 1500 + b"ACGA" * 1500 + b"ACGT" * 700, R(2000, 13) * 5,
]
# periods longer than the depth compared so far (cap 49): taken only if the members really agree on a period
U = R(60, 30); V = bytearray(U); V[50] = ord("A") if V[50] != ord("A") else ord("C"); V = bytes(V)
W = R(171, 31)
texts += [U * 50, U * 40 + R(77, 32) + U * 35, U * 40 + R(77, 33) + V * 40, V * 30 + U * 45, (U * 30)[:-7], W * 40,
          W * 25 + R(30, 34) + W * 30 + b"A", R(40, 35) + (U * 33)[5:] + R(3, 36) + U * 20]
for k in range(12):   # random mixtures of runs
    parts = []
    for _ in range(int(rng.integers(2, 9))):
        unit = R(int(rng.integers(1, 30)), 100 + int(rng.integers(0, 5)))
        parts.append(unit * int(rng.integers(1, 400)))
        if rng.random() < 0.6: parts.append(R(int(rng.integers(1, 200)), 200 + k))
    texts.append(b"".join(parts))
for t in texts:
    d = native.debug_arrays(t)
    sa = oracle.suffix_array(t)
    assert np.array_equal(d["sa"].astype(np.int64), sa.astype(np.int64)), ("sa", len(t), t[:40])
    lcp = oracle.lcp_array(t, sa)
    assert np.array_equal(d["lcp"][:len(t)].astype(np.int64), lcp.astype(np.int64)), ("lcp", len(t), t[:40])
    f, e = native.factorize_array(t), oracle.factors_array(t)
    assert len(f) == len(e) and all(np.array_equal(f[k], e[k]) for k in ("start", "length", "ref")), ("factors", len(t), t[:40])
for t in [b"A" * 3000 + b"T" * 3000, b"AC" * 2500 + b"GT" * 2500, R(50, 20) + b"AAG" * 1500 + R(60, 21) + b"CTT" * 1500]:
    assert native.factorize_dna_w_rc(t) == oracle.factorize_dna_w_rc(t), ("rc", len(t))
print("ok", len(texts))
'''
    env = dict(os.environ, NOLZSS_REFINE_WORDS="1", NOLZSS_PAIR_RUNS_MIN="1", NOLZSS_TRACE="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    done = [int(line.split("periodic runs")[1].split(":")[1].split()[0]) for line in r.stderr.splitlines() if "periodic runs" in line]
    assert len(done) >= 20 and sum(done) > 50000, done  # the pass ran and finished suffixes
