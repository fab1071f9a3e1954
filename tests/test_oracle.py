"""CPU tests of the oracle (oracle/): pinned to the reference's known-answer vectors
(tests/golden/kats.json) and cross-checked against definition-level brute force."""
import json
import random
from pathlib import Path

import numpy as np
import pytest

import bruteforce as bf
import oracle_lib as oracle

KATS = json.loads((Path(__file__).parent / "golden" / "kats.json").read_text())


def _text(v):
    if "input" in v:
        return v["input"].encode("ascii")
    s, k = v["input_repeat"]
    return s.encode("ascii") * k


@pytest.mark.parametrize("v", KATS["plain"] + KATS["derived_plain"], ids=lambda v: v["source"][:40])
def test_plain_kats(v):
    assert oracle.factorize(_text(v)) == [tuple(f) for f in v["factors"]]
    assert oracle.count_factors(_text(v)) == len(v["factors"])


@pytest.mark.parametrize("v", KATS["dna_w_rc"] + KATS["derived_dna_w_rc"], ids=lambda v: v["input"])
def test_rc_kats(v):
    assert oracle.factorize_dna_w_rc(_text(v)) == [tuple(f) for f in v["factors"]]


def test_rc_forward_preferred_on_tie():
    v = KATS["dna_w_rc_partial"][0]
    got = oracle.factorize_dna_w_rc(_text(v))
    assert got[v["index"]] == tuple(v["factor"])


def test_doc_example_the_code_contradicts():
    """docs/RC_ALGORITHM.md:298-323 tabulates ATCGATCG as four literals + (4,4,0); a node-by-node trace of
    factorizer_core.hpp:256-352 (kept with the vector in kats.json) gives an RC factor at 3, as in the
    reference's own test vector ATGCAT.  The oracle and both brute-force models follow the code."""
    v = KATS["reference_doc_example_contradicted_by_the_code"][0]
    expected = [tuple(f) for f in v["code_trace"]]
    assert oracle.factorize_dna_w_rc(_text(v)) == expected
    assert bf.rc_factorize(v["input"]) == expected
    assert expected != [tuple(f) for f in v["doc_table"]]
    # the plain-mode parse of the same text is what the doc table shows
    assert oracle.factorize(_text(v)) == [tuple(f[:3]) for f in v["doc_table"]]


def _gen(rng, kind, n):
    if kind == 0:
        return "".join(rng.choice("ACGT") for _ in range(n))
    if kind == 1:
        return "".join(rng.choice("AC") for _ in range(n))
    if kind == 2:
        u = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 4)))
        return (u * n)[:n]
    if kind == 3:
        s = ""
        while len(s) < n:
            if s and rng.random() < 0.5:
                a = rng.randrange(len(s))
                s += s[a:a + rng.randint(1, 8)]
            else:
                s += rng.choice("ACGT")
        return s[:n]
    return "".join(rng.choice("abcdefgh") for _ in range(n))


def test_plain_vs_bruteforce():
    rng = random.Random(11)
    for _ in range(400):
        kind, n = rng.randrange(5), rng.randint(1, 36)
        t = _gen(rng, kind, n)
        got = oracle.factorize(t.encode())
        assert got == bf.plain_factorize(t, walk=True), t
        assert got == bf.plain_factorize(t, walk=False), t
        ln, rf = oracle.lpnf_all(t.encode())
        for i in range(n):
            f = bf.plain_closed_form_at(t, i)
            assert (int(ln[i]), int(rf[i])) == (f[1], f[2]), (t, i)


def test_rc_vs_bruteforce():
    rng = random.Random(12)
    for _ in range(300):
        t = _gen(rng, rng.randrange(4), rng.randint(1, 30))
        assert oracle.factorize_dna_w_rc(t.encode()) == bf.rc_factorize(t), t


def test_multi_rc_and_prepare_vs_bruteforce():
    rng = random.Random(13)
    for _ in range(120):
        seqs = [_gen(rng, rng.randrange(4), rng.randint(1, 12)) for _ in range(rng.randint(1, 4))]
        if rng.random() < 0.3:
            seqs = [s.lower() for s in seqs]
        S, orig, sent = bf.prepare_w_rc(seqs)
        S2, orig2, sent2 = oracle.prepare_multiple_dna_w_rc(seqs)
        assert (S.encode("latin-1"), orig, sent) == (S2, orig2, sent2)
        assert oracle.factorize_multiple_dna_w_rc(S2) == bf.rc_factorize_prepared(S)
        assert oracle.count_factors_multiple_dna_w_rc(S2) == len(bf.rc_factorize_prepared(S))


def test_prepare_errors_and_limits():
    with pytest.raises(oracle.OracleError):
        oracle.prepare_multiple_dna_w_rc(["ACGN"])            # factorizer.cpp:86-95
    with pytest.raises(oracle.OracleError):
        oracle.prepare_multiple_dna_w_rc(["", ""])            # :76-78
    with pytest.raises(oracle.OracleInvalidArgument):
        oracle.prepare_multiple_dna_w_rc(["A"] * 126)         # :81-83
    S, orig, sent = oracle.prepare_multiple_dna_w_rc(["A"] * 125)
    assert len(S) == 500 and orig == 250 and len(set(S[1::2])) == 250
    assert not (set(S[1::2]) & set(b"\x00ACGT"))
    assert oracle.prepare_multiple_dna_w_rc([]) == (b"", 0, [])


def test_rc_start_pos_guard():
    S, _, _ = oracle.prepare_multiple_dna_w_rc(["ACGTACGT"])
    with pytest.raises(oracle.OracleInvalidArgument):         # factorizer_core.hpp:203-205
        oracle.factorize_multiple_dna_w_rc(S, start_pos=8)
    assert oracle.factorize_multiple_dna_w_rc(b"") == []      # :180
    assert oracle.factorize_multiple_dna_w_rc(b"A\x01") == [] # :189-193


def test_sa_lcp_vs_naive():
    rng = random.Random(14)
    for _ in range(100):
        n = rng.randint(1, 150)
        t = bytes(rng.choice(b"\x01\x02\xff\x00ab") for _ in range(n))
        sa = oracle.suffix_array(t)
        assert sa.tolist() == sorted(range(n), key=lambda i: t[i:])
        lcp = oracle.lcp_array(t, sa)
        for r in range(1, n):
            a, b, h = int(sa[r - 1]), int(sa[r]), 0
            while a + h < n and b + h < n and t[a + h] == t[b + h]:
                h += 1
            assert lcp[r] == h


def test_invariants_medium_random():
    """tiling, count == len, every factor a true earlier non-overlapping occurrence
    (reference invariants: tests/test_cpp_bindings.py:9-21,37-104)."""
    rng = np.random.default_rng(5)
    t = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 200_000)].tobytes()
    f = oracle.factors_array(t)
    assert oracle.count_factors(t) == len(f)
    assert f["start"][0] == 0
    assert np.all(f["start"][1:] == f["start"][:-1] + f["length"][:-1])
    assert f["start"][-1] + f["length"][-1] == len(t)
    for s, l, r in zip(f["start"][:3000].tolist(), f["length"][:3000].tolist(), f["ref"][:3000].tolist()):
        if r == s:
            assert l == 1
        else:
            assert r + l <= s and t[r:r + l] == t[s:s + l]


def test_start_pos():
    t = b"abracadabra" * 20
    full = oracle.factorize(t)
    for sp in (0, 7, 11, 100, len(t) - 1):
        got = oracle.factorize(t, start_pos=sp)
        assert got[0][0] == sp and got[-1][0] + got[-1][1] == len(t)
        assert got == bf.plain_factorize(t.decode(), walk=False, start_pos=sp)
    assert oracle.factorize(t, start_pos=len(t)) == []
    assert full[0] == (0, 1, 0)
