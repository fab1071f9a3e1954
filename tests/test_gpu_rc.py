"""GPU parity of the reverse-complement DNA mode against the oracle (through the C ABI)."""
import json
import random
from pathlib import Path

import numpy as np
import pytest

import bruteforce as bf
import gen
import oracle_lib as oracle

pytestmark = pytest.mark.gpu

KATS = json.loads((Path(__file__).parent / "golden" / "kats.json").read_text())


@pytest.fixture(scope="module")
def native():
    from nolzss_amd import _noLZSS
    assert _noLZSS.device_count() >= 1, "no MI355X visible"
    return _noLZSS


@pytest.mark.parametrize("v", KATS["dna_w_rc"] + KATS["derived_dna_w_rc"], ids=lambda v: v["input"])
def test_rc_kats(native, v):
    t = v["input"].encode()
    assert native.factorize_dna_w_rc(t) == [tuple(f) for f in v["factors"]]
    assert native.count_factors_dna_w_rc(t) == len(v["factors"])


def test_rc_forward_preferred_on_tie(native):
    v = KATS["dna_w_rc_partial"][0]
    assert native.factorize_dna_w_rc(v["input"].encode())[v["index"]] == tuple(v["factor"])


def test_doc_example_follows_the_code(native):
    """docs/RC_ALGORITHM.md:298-323 (ATCGATCG): the GPU path gives what the code of the reference gives
    (trace in kats.json), not the simplified table of the doc."""
    v = KATS["reference_doc_example_contradicted_by_the_code"][0]
    assert native.factorize_dna_w_rc(v["input"].encode()) == [tuple(f) for f in v["code_trace"]]


def _mixed(rng, n, p_copy=0.5, maxlen=40):
    s = ""
    while len(s) < n:
        if s and rng.random() < p_copy:
            a = rng.randrange(len(s))
            seg = s[a:a + rng.randint(1, maxlen)]
            if rng.random() < 0.5:
                seg = bf.revcomp(seg)
            s += seg
        else:
            s += rng.choice("ACGT")
    return s[:n]


def _cases():
    rng = random.Random(77)
    c = {
        "A": "A",
        "AAAA": "A" * 64,
        "AT_period": "AT" * 500,
        "ACAG_tandem": ("ACAG" * 300 + "AGAGAT") * 3,
        "mixed_2k": _mixed(rng, 2000),
        "mixed_50k": _mixed(rng, 50_000, maxlen=300),
        "random_100k": gen.random_dna(100_000, 21).tobytes().decode(),
        "repeat_400k": gen.repeat_dna(400_000, 22, lo=16, hi=4096).tobytes().decode(),
        "palindromic": "ACGT" * 700 + "TTAA" * 200,
    }
    return c


CASES = _cases()


@pytest.mark.parametrize("name", list(CASES))
def test_rc_single_sequence(native, name):
    t = CASES[name].encode()
    got = native.factorize_dna_w_rc_array(t)
    S, _, _ = oracle.prepare_multiple_dna_w_rc([t])
    exp = oracle.factors_array_multiple_dna_w_rc(S)
    assert len(got) == len(exp)
    for k in ("start", "length", "ref"):
        assert np.array_equal(got[k], exp[k]), k
    assert native.count_factors_dna_w_rc(t) == len(exp)


def test_rc_multiple_sequences(native):
    rng = random.Random(5)
    for trial in range(25):
        k = rng.randint(1, 9)
        seqs = [_mixed(rng, rng.randint(1, 400)) for _ in range(k)]
        S, orig, sent = native.prepare_multiple_dna_sequences_w_rc_bytes(seqs)
        S2, orig2, sent2 = oracle.prepare_multiple_dna_w_rc(seqs)
        assert (S, orig, sent) == (S2, orig2, sent2)
        assert native.factorize_multiple_dna_w_rc(S) == oracle.factorize_multiple_dna_w_rc(S)
        assert native.count_factors_multiple_dna_w_rc(S) == oracle.count_factors_multiple_dna_w_rc(S)
        for p in sent[:k - 1]:  # sentinels inside T come out as literals
            lst = native.factorize_multiple_dna_w_rc(S)
            assert (p, 1, p, False) in lst


def test_rc_many_small(native):
    rng = random.Random(6)
    for _ in range(200):
        t = _mixed(rng, rng.randint(1, 120), maxlen=12).encode()
        assert native.factorize_dna_w_rc(t) == oracle.factorize_dna_w_rc(t), t


def test_rc_every_factor_is_a_true_match(native):
    """reference invariant: tests/test_factorization_validation.py:118-175"""
    t = CASES["mixed_50k"]
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    pos = 0
    for s, l, r, is_rc in native.factorize_dna_w_rc(t.encode()):
        assert s == pos
        if r == s and not is_rc:
            assert l == 1
        elif not is_rc:
            assert r + l <= s and t[r:r + l] == t[s:s + l]
        else:
            src = t[r:r + l]
            assert r + l <= s and "".join(comp[c] for c in reversed(src)) == t[s:s + l]
        pos += l
    assert pos == len(t)


def test_rc_guards(native):
    assert native.factorize_dna_w_rc(b"") == []
    assert native.factorize_multiple_dna_w_rc(b"") == []
    assert native.factorize_multiple_dna_w_rc(b"A\x01") == []
    S, _, _ = native.prepare_multiple_dna_sequences_w_rc_bytes(["ACGTACGT"])
    with pytest.raises(ValueError):
        native.factorize_multiple_dna_w_rc_array(S, start_pos=8)
    with pytest.raises(RuntimeError):
        native.factorize_dna_w_rc(b"ACGN")


@pytest.mark.parametrize("k", [41, 62, 100, 125])
def test_rc_many_sequences_up_to_the_limit(native, k):
    """More than 40 sequences put sentinel bytes above 'T' into the prepared string (up to 250
    sentinels at the limit of 125 sequences); sequences share their last bases, so copies of one
    short suffix sit at many terminators."""
    rng = random.Random(100 + k)
    tail = "ACGTTGCAAGGCTA"
    seqs = ["".join(rng.choice("ACGT") for _ in range(rng.randint(1, 80))) + tail[-rng.randint(1, 14):]
            for _ in range(k)]
    S, orig, sent = native.prepare_multiple_dna_sequences_w_rc_bytes(seqs)
    S2, orig2, sent2 = oracle.prepare_multiple_dna_w_rc(seqs)
    assert (S, orig, sent) == (S2, orig2, sent2)
    assert max(S) > ord("T")
    assert native.factorize_multiple_dna_w_rc(S) == oracle.factorize_multiple_dna_w_rc(S)
    assert native.count_factors_multiple_dna_w_rc(S) == oracle.count_factors_multiple_dna_w_rc(S)


def test_concatenation_without_rc_many_sequences(native):
    """the no-rc concatenation (one sentinel per sequence) with 200 sequences, plain factorization of
    the prepared string against the oracle"""
    rng = random.Random(77)
    seqs = ["".join(rng.choice("ACGT") for _ in range(rng.randint(1, 60))) + "GATTACA"[-rng.randint(1, 7):]
            for _ in range(200)]
    S, total, sent = native.prepare_multiple_dna_sequences_no_rc_bytes(seqs)
    assert max(S) > ord("T") and len(sent) >= 199
    assert native.factorize(S) == oracle.factorize(S)


@pytest.mark.timeout(600)
def test_rc_six_million_bases_every_factor(native):
    """6 M bases with copied blocks (and their reverse complements, which RC mode finds): above 2^22 bases the
    candidate kernel hands only the ranks of the original strand to the permutation that brings codes and ranks
    into text order, and that permutation takes its two partition passes and the LDS windows (rc.hip,
    radix_sort.hip: permute_packed) -- the path of BASELINE config 5, here at a size the oracle finishes in seconds."""
    t = gen.repeat_dna(6_000_000, seed=0x5EED0005 + 17)
    got = native.factorize_dna_w_rc_array(t)
    S, _, _ = oracle.prepare_multiple_dna_w_rc([t.tobytes()])
    exp = oracle.factors_array_multiple_dna_w_rc(S)
    assert len(got) == len(exp)
    for k in ("start", "length", "ref"):
        assert np.array_equal(got[k], exp[k]), k
    assert (got["ref"] >> np.uint64(63)).any()  # reverse-complement factors are there
