"""CPU: the oracle on the real-sequence fixtures the reference's tests hold (tests/golden/genomes/).

The reference cannot be built here (sdsl-lite is not available offline), so no reference-made factor
list exists for these files; what pins the oracle on them is (a) the invariants the reference's own
test checks on the same files (tests/test_factorization_validation.py:118-211), (b) the
definition-level brute-force models on prefixes, and (c) PARTIAL agreement with the two stale factor
files the reference ships (old "noLZSSv1" layout, RC-preferred tie-break of an older version).
"""
import hashlib
import json

import pytest

import bruteforce as bf
import genomes
import oracle_lib as oracle

SMALL = ["short_dna1", "short_dna2", "T3", "T7", "test_viral_dna", "test_bacterial_dna"]


def test_fixtures_match_manifest():
    man = json.loads((genomes.DIR / "MANIFEST.json").read_text())
    for fname, meta in man.items():
        data = (genomes.DIR / fname).read_bytes()
        if fname.endswith(".gz"):
            assert hashlib.sha256(genomes.raw(fname[:-len(".fna.gz")])).hexdigest() == meta["sha256_uncompressed"]
        else:
            assert hashlib.sha256(data).hexdigest() == meta["sha256"]
    for name in genomes.NAMES:
        recs = genomes.records(name)
        assert recs and all(seq and set(seq) <= set(b"ACGT") for _, seq in recs)


@pytest.mark.parametrize("name", SMALL)
def test_oracle_plain_invariants(name):
    for _, seq in genomes.records(name):
        f = oracle.factorize(seq)
        genomes.check_plain_invariants(seq, f)
        assert oracle.count_factors(seq) == len(f)


@pytest.mark.parametrize("name", SMALL)
def test_oracle_rc_invariants_multi_sequence(name):
    """the check of the reference's TestFactorizationCorrectness, on the same files"""
    seqs = [seq for _, seq in genomes.records(name)]
    S, orig, sent = oracle.prepare_multiple_dna_w_rc(seqs)
    f = oracle.factorize_multiple_dna_w_rc(S)
    genomes.check_rc_invariants(S, orig, f, sent[:len(seqs)])
    assert sum(1 for x in f if not x[3]) > 0
    if name in ("T7", "test_viral_dna"):
        assert sum(1 for x in f if x[3]) > 0  # real genomes do contain reverse-complement repeats


@pytest.mark.parametrize("name", ["short_dna1", "short_dna2", "T7", "test_bacterial_dna"])
def test_oracle_matches_bruteforce_on_prefix(name):
    seq = genomes.records(name)[0][1][:700].decode()
    assert oracle.factorize(seq.encode()) == bf.plain_factorize(seq, walk=False)
    assert oracle.factorize(seq.encode()) == bf.plain_factorize(seq, walk=True)
    short = seq[:260]
    assert oracle.factorize_dna_w_rc(short.encode()) == bf.rc_factorize(short)


def test_stale_dna1_w_dna2_factor_file_partial_agreement():
    """dna1_factors_w_dna2_ref.bin (reference fixture, v1 layout): all 7 (start, length) pairs agree
    with the oracle; the 2 refs that differ are ties the old version gave to the reverse complement
    (today: forward preferred, factorizer_core.hpp:338-352) -- partial corroboration, labelled as such."""
    old = genomes.read_v1_factor_file("dna1_factors_w_dna2_ref.bin")
    ref = [s for _, s in genomes.records("short_dna2")]
    tgt = [s for _, s in genomes.records("short_dna1")]
    S, _, _ = oracle.prepare_multiple_dna_w_rc(ref + tgt)
    new = oracle.factorize_multiple_dna_w_rc(S, start_pos=sum(len(s) + 1 for s in ref))
    assert len(old) == len(new) == 7
    assert [x[:2] for x in old] == [x[:2] for x in new]
    differing = [(a, b) for a, b in zip(old, new) if a != b]
    assert len(differing) == 2
    for a, b in differing:
        assert a[3] and not b[3]  # old: RC, new: forward, same length
        s, l, r, _ = b
        assert S[r:r + l] == S[s:s + l] and r + l <= s


def test_stale_t7_w_t3_factor_file_partial_agreement():
    """T7_factors_w_T3_ref.bin: 3911 factors in the old file, 3910 from the oracle.  Every factor of
    the old file is a true (RC) match; the first 35 (start, length) pairs agree, 3908 factor starts
    are common, and among those exactly ONE length differs: at 38509 the old version took an RC match
    of 9 where a forward match of 13 ends exactly at the cursor (ref + len == start, allowed by
    factorizer_core.hpp:264-266 today).  Refs differ where the old version preferred RC on ties.
    Partial corroboration only (SURVEY section 4 calls the file stale)."""
    old = genomes.read_v1_factor_file("T7_factors_w_T3_ref.bin")
    t3 = genomes.records("T3")[0][1]
    t7 = genomes.records("T7")[0][1]
    S, orig, sent = oracle.prepare_multiple_dna_w_rc([t3, t7])
    new = oracle.factorize_multiple_dna_w_rc(S, start_pos=len(t3) + 1)
    assert len(old) == 3911 and len(new) == 3910
    genomes.check_rc_invariants(S, orig, new, sent[:2], start_pos=len(t3) + 1)
    assert [x[:2] for x in old[:35]] == [x[:2] for x in new[:35]]
    by_start = {x[0]: x for x in new}
    common = [x for x in old if x[0] in by_start]
    assert len(common) == 3908
    longer = [(x, by_start[x[0]]) for x in common if by_start[x[0]][1] != x[1]]
    assert longer == [((38509, 9, 29076, True), (38509, 13, 38496, False))]
    for a in common:
        b = by_start[a[0]]
        if a != b and a[1] == b[1]:  # same factor, different source: the old RC-preferred tie-break
            assert a[3] and not b[3]
