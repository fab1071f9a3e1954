"""GPU parity on the real-sequence fixtures the reference's tests hold (tests/golden/genomes/,
copied by tests/golden/make_genome_fixtures.py from the reference's tests/resources/): plain, reverse
complement, concatenated multi-sequence (with and without RC) and reference/target modes --
HIP path == oracle bit-exact, plus the reference's own invariants on the same files
(reference: tests/test_factorization_validation.py:87-211, tests/test_reference_seq.py)."""
import numpy as np
import pytest

import genomes
import oracle_lib as oracle

pytestmark = pytest.mark.gpu

SMALL = ["short_dna1", "short_dna2", "T3", "T7", "test_viral_dna", "test_bacterial_dna"]
ALL = SMALL + ["Vibrio_cholerae"]


@pytest.fixture(scope="module")
def native():
    from nolzss_amd import _noLZSS
    assert _noLZSS.device_count() >= 1, "no MI355X visible"
    return _noLZSS


def _same(got, exp):
    return len(got) == len(exp) and all(np.array_equal(got[k], exp[k]) for k in ("start", "length", "ref"))


@pytest.mark.parametrize("name", ALL)
def test_plain_every_record(native, name):
    import nolzss_amd
    for _, seq in genomes.records(name):
        exp = oracle.factors_array(seq)
        assert _same(native.factorize_array(seq), exp)
        assert nolzss_amd.count_factors(seq) == len(exp)
        if len(seq) < 500_000:
            tuples = nolzss_amd.factorize(seq)
            genomes.check_plain_invariants(seq, tuples)
            assert tuples == list(zip(exp["start"].tolist(), exp["length"].tolist(), exp["ref"].tolist()))


@pytest.mark.parametrize("name", ALL)
def test_rc_every_record(native, name):
    for _, seq in genomes.records(name):
        S, _, _ = oracle.prepare_multiple_dna_w_rc([seq])
        exp = oracle.factors_array_multiple_dna_w_rc(S)
        assert _same(native.factorize_dna_w_rc_array(seq), exp)
        assert native.count_factors_dna_w_rc(seq) == len(exp)


@pytest.mark.parametrize("name", ALL)
def test_fasta_multiple_dna_w_rc_and_no_rc(native, name, tmp_path):
    """the reference's TestFactorizationCorrectness on its own files, through the FASTA entry point"""
    path = genomes.materialize(name, tmp_path)
    recs = genomes.records(name)
    seqs = [s for _, s in recs]
    factors, sentinels, ids = native.factorize_fasta_multiple_dna_w_rc(path)
    S, orig, sent_pos = oracle.prepare_multiple_dna_w_rc(seqs)
    assert factors == oracle.factorize_multiple_dna_w_rc(S)
    assert ids == [rid for rid, _ in recs]
    assert [factors[i][0] for i in sentinels] == sent_pos[:len(seqs) - 1]
    genomes.check_rc_invariants(S, orig, factors, sent_pos[:len(seqs)])
    # bindings-level twin on the prepared string (reference: bindings.cpp:361-382)
    assert native.factorize_multiple_dna_w_rc(S) == factors
    assert native.count_factors_multiple_dna_w_rc(S) == len(factors)

    f2, s2, ids2 = native.factorize_fasta_multiple_dna_no_rc(path)
    S2, _, sent2 = native.prepare_multiple_dna_sequences_no_rc_bytes([s.decode() for s in seqs])
    plain = [(s, l, r) for s, l, r, _ in f2]
    assert plain == oracle.factorize(S2) and ids2 == ids
    assert [f2[i][0] for i in s2] == sent2
    genomes.check_plain_invariants(S2, plain)


def test_t7_with_t3_as_reference(native, tmp_path):
    """reference/target mode on the pair the reference ships a (stale) factor file for:
    factorize_dna_w_reference_seq (factorizer.cpp:825-842) and the FASTA-file form
    (fasta_processor.cpp:240-287)"""
    from nolzss_amd.genomics import factorize_dna_w_reference_seq
    t3 = genomes.records("T3")[0][1]
    t7 = genomes.records("T7")[0][1]
    S, orig, sent = oracle.prepare_multiple_dna_w_rc([t3, t7])
    exp = oracle.factorize_multiple_dna_w_rc(S, start_pos=len(t3) + 1)
    got = factorize_dna_w_reference_seq(t3.decode(), t7.decode())
    assert got == exp and len(got) == 3910
    genomes.check_rc_invariants(S, orig, got, sent[:2], start_pos=len(t3) + 1)
    f, _, ids = native.factorize_dna_rc_w_ref_fasta_files(genomes.materialize("T3", tmp_path),
                                                          genomes.materialize("T7", tmp_path))
    assert f == exp and ids == ["NC_047864.1", "V01146.1"]
    # the stale fixture agrees on the first 35 (start, length) pairs (tests/test_golden_genomes.py)
    old = genomes.read_v1_factor_file("T7_factors_w_T3_ref.bin")
    assert [x[:2] for x in old[:35]] == [x[:2] for x in got[:35]]


def test_short_dna1_with_short_dna2_as_reference(native, tmp_path):
    f, _, ids = native.factorize_dna_rc_w_ref_fasta_files(genomes.materialize("short_dna2", tmp_path),
                                                          genomes.materialize("short_dna1", tmp_path))
    old = genomes.read_v1_factor_file("dna1_factors_w_dna2_ref.bin")
    assert [x[:2] for x in f] == [x[:2] for x in old]
    assert ids == ["short_dna_2_seq1", "short_dna_2_seq2", "short_dna_1_seq1", "short_dna_1_seq2"]


@pytest.mark.parametrize("name", ["short_dna1", "test_bacterial_dna", "Vibrio_cholerae"])
def test_per_sequence_batch_and_read_nucleotide_fasta(native, name, tmp_path):
    """the per-sequence FASTA batch (reference: genomics/fasta.py:79-126) on real records"""
    from nolzss_amd.genomics import read_nucleotide_fasta
    path = genomes.materialize(name, tmp_path)
    recs = genomes.records(name)
    counts, arrays = native.factorize_batch([s for _, s in recs], want_factors=True)
    for (_, seq), c, a in zip(recs, counts, arrays):
        exp = oracle.factors_array(seq)
        assert c == len(exp) and _same(a, exp)
    if name != "Vibrio_cholerae":  # (tuple lists of a 4 Mb genome are slow to build, not to compute)
        got = read_nucleotide_fasta(path)
        assert [rid for rid, _ in got] == [rid for rid, _ in recs]
        for (_, seq), (_, f) in zip(recs, got):
            assert f == oracle.factorize(seq)
