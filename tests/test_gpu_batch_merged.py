"""GPU parity of the MERGED batch: short nucleotide records factorized together in one pipeline run
as independent sequences (batch.hip, run_merged_chunk) must give, record by record, exactly what the
oracle -- and the one-record-at-a-time path -- gives.
reference: the per-sequence loop of genomics.read_nucleotide_fasta (src/noLZSS/genomics/fasta.py:110-122)
"""
import os

import numpy as np
import pytest

import gen
import oracle_lib as oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    from nolzss_amd import _noLZSS
    assert _noLZSS.device_count() >= 1, "no MI355X visible"
    return _noLZSS


class merge_below:
    """NOLZSS_BATCH_MERGE_BELOW for the duration of a block (the library reads it on every call)."""

    def __init__(self, value):
        self.value = value

    def __enter__(self):
        self.old = os.environ.get("NOLZSS_BATCH_MERGE_BELOW")
        if self.value is None:
            os.environ.pop("NOLZSS_BATCH_MERGE_BELOW", None)
        else:
            os.environ["NOLZSS_BATCH_MERGE_BELOW"] = str(self.value)

    def __exit__(self, *exc):
        if self.old is None:
            os.environ.pop("NOLZSS_BATCH_MERGE_BELOW", None)
        else:
            os.environ["NOLZSS_BATCH_MERGE_BELOW"] = self.old


def _same(a, b):
    return len(a) == len(b) and all(np.array_equal(a[k], b[k]) for k in ("start", "length", "ref"))


def _records(rng, count, lo, hi, repeat=True):
    recs = []
    for k in range(count):
        n = int(rng.integers(lo, hi + 1))
        recs.append(gen.repeat_dna(n, seed=7000 + k) if repeat and n >= 64 else gen.random_dna(n, seed=7000 + k))
    return recs


def test_merged_equals_oracle_record_by_record(native):
    rng = np.random.default_rng(11)
    recs = _records(rng, 300, 1, 3000)
    # records that are copies of each other must not see each other; runs; the four one-base records
    recs += [recs[5].copy(), recs[5].copy(), recs[17][:100].copy()]
    recs += [np.frombuffer(b"A" * 777, dtype=np.uint8), np.frombuffer(b"ACGT" * 300, dtype=np.uint8)]
    recs += [np.frombuffer(c, dtype=np.uint8) for c in (b"A", b"C", b"G", b"T", b"AA", b"TTTTTTTTTTTTTTTTTTTTTTTTT")]
    merged0, single0 = native.debug_batch_counters()
    counts, arrays = native.factorize_batch(recs, want_factors=True)
    merged1, single1 = native.debug_batch_counters()
    assert (merged1 - merged0, single1 - single0) == (len(recs), 0)  # the merged path is what ran
    for j, r in enumerate(recs):
        exp = oracle.factors_array(r)
        assert counts[j] == len(exp), j
        assert _same(arrays[j], exp), j
    counts2, none = native.factorize_batch(recs, want_factors=False)
    assert counts2 == counts and none is None


def test_merged_equals_one_by_one_on_larger_records(native):
    rng = np.random.default_rng(12)
    recs = _records(rng, 40, 20000, 300000)
    recs += [recs[3].copy()]
    merged0, single0 = native.debug_batch_counters()
    with merge_below(0):
        c1, a1 = native.factorize_batch(recs, want_factors=True)
    merged1, single1 = native.debug_batch_counters()
    assert (merged1 - merged0, single1 - single0) == (0, len(recs))
    c2, a2 = native.factorize_batch(recs, want_factors=True)
    assert native.debug_batch_counters() == (merged1 + len(recs), single1)
    assert c1 == c2
    for j in range(len(recs)):
        assert _same(a1[j], a2[j]), j
    for j in (0, 19, 40):
        assert _same(a2[j], oracle.factors_array(recs[j])), j


def test_few_records_and_empty_ones(native):
    """fewer than 256 separators (plain terminator search), empty records in between, a single record"""
    recs = [gen.random_dna(500, 1), np.zeros(0, dtype=np.uint8), gen.repeat_dna(4000, 2), gen.random_dna(1, 3),
            np.zeros(0, dtype=np.uint8)]
    counts, arrays = native.factorize_batch(recs, want_factors=True)
    for j, r in enumerate(recs):
        exp = oracle.factors_array(r) if len(r) else None
        assert counts[j] == (len(exp) if exp is not None else 0)
        if exp is not None:
            assert _same(arrays[j], exp)
        else:
            assert len(arrays[j]) == 0
    one = [gen.repeat_dna(3000, 9)]
    counts, arrays = native.factorize_batch(one, want_factors=True)
    assert _same(arrays[0], oracle.factors_array(one[0]))


def test_other_alphabets_fall_back(native):
    """a record with anything but A/C/G/T sends its chunk through the one-by-one path: same results"""
    rng = np.random.default_rng(13)
    recs = _records(rng, 50, 10, 2000)
    recs[20] = np.frombuffer(b"ACGTNNNNACGTNACGTTTGACN" * 20, dtype=np.uint8)
    recs[31] = np.frombuffer(b"abracadabra" * 9, dtype=np.uint8)
    recs[32] = np.frombuffer(b"AC\x01GT\x01AC\x01GT", dtype=np.uint8)  # the separator byte inside a record
    merged0, single0 = native.debug_batch_counters()
    counts, arrays = native.factorize_batch(recs, want_factors=True)
    assert native.debug_batch_counters() == (merged0, single0 + len(recs))
    for j, r in enumerate(recs):
        assert _same(arrays[j], oracle.factors_array(r)), j


def test_chunks_split_and_mix_with_long_records(native):
    """a merge limit in the middle of the size range: the records below it merged in runs of their own, the
    longer ones in runs of long records (uploaded one by one into the run's device text) or, with
    NOLZSS_BATCH_MERGE_LONG_BELOW=0, one pipeline run each"""
    rng = np.random.default_rng(14)
    recs = _records(rng, 120, 100, 9000)
    long_ones = sum(1 for r in recs if len(r) >= 4000)
    assert 0 < long_ones < len(recs)
    expected = [oracle.factors_array(r) for r in recs]
    merged0, single0 = native.debug_batch_counters()
    with merge_below(4000):
        counts, arrays = native.factorize_batch(recs, want_factors=True)
    assert native.debug_batch_counters() == (merged0 + len(recs), single0)
    for j in range(len(recs)):
        assert _same(arrays[j], expected[j]), j
    os.environ["NOLZSS_BATCH_MERGE_LONG_BELOW"] = "0"
    try:
        merged0, single0 = native.debug_batch_counters()
        with merge_below(4000):
            counts, arrays = native.factorize_batch(recs, want_factors=True)
        assert native.debug_batch_counters() == (merged0 + len(recs) - long_ones, single0 + long_ones)
    finally:
        os.environ.pop("NOLZSS_BATCH_MERGE_LONG_BELOW", None)
    for j in range(len(recs)):
        assert _same(arrays[j], expected[j]), j


@pytest.mark.timeout(600)
def test_many_short_records(native):
    """4096 records of 150..6000 bases (one merged run of ~12 Mi bases): a sample against the oracle,
    every count against the one-by-one path"""
    rng = np.random.default_rng(15)
    recs = _records(rng, 4096, 150, 6000)
    counts, arrays = native.factorize_batch(recs, want_factors=True)
    for j in rng.choice(len(recs), size=200, replace=False).tolist():
        assert _same(arrays[j], oracle.factors_array(recs[j])), j
    with merge_below(0):
        c1, _ = native.factorize_batch(recs, want_factors=False)
    assert c1 == counts
    for j, r in enumerate(recs):  # tilings
        f = arrays[j]
        assert f["start"][0] == 0 and int(f["start"][-1] + f["length"][-1]) == len(r)


@pytest.mark.timeout(900)
def test_read_sized_records_by_the_hundred_thousand(native):
    """200 000 records of 30..300 bases: 18 bits of record number in the sort key (8 radix passes),
    dozens of separators per 4096-symbol block of the coarse terminator index"""
    rng = np.random.default_rng(16)
    base = gen.repeat_dna(1 << 20, seed=99)
    lens = rng.integers(30, 301, size=200000)
    offs = rng.integers(0, (1 << 20) - 300, size=200000)
    recs = [base[o:o + l] for o, l in zip(offs.tolist(), lens.tolist())]
    merged0, single0 = native.debug_batch_counters()
    counts, arrays = native.factorize_batch(recs, want_factors=True)
    assert native.debug_batch_counters() == (merged0 + len(recs), single0)
    for j in rng.choice(len(recs), size=1500, replace=False).tolist():
        assert _same(arrays[j], oracle.factors_array(recs[j])), j
    total = 0
    for j, f in enumerate(arrays):
        assert len(f) == counts[j] and f["start"][0] == 0
        total += int(f["length"].sum())
    assert total == int(lens.sum())


def test_differential_fuzz_of_tiny_records(native):
    """a few seconds of random batches of very short records over sub-alphabets (long runs, periodic
    records, records that are prefixes / copies of each other) against the oracle"""
    import time
    rng = np.random.default_rng(17)
    t_end = time.time() + 6.0
    cases = 0
    while time.time() < t_end:
        sigma = int(rng.integers(1, 5))
        letters = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.permutation(4)[:sigma]]
        m = int(rng.integers(2, 400))
        recs = []
        for _ in range(m):
            kind = rng.integers(0, 4)
            n = int(rng.integers(1, 60))
            if kind == 0 and recs:
                src = recs[int(rng.integers(0, len(recs)))]
                r = src[:max(1, int(rng.integers(1, len(src) + 1)))].copy()
            elif kind == 1:
                unit = letters[rng.integers(0, sigma, size=int(rng.integers(1, 5)))]
                r = np.tile(unit, n)[:max(1, n)]
            else:
                r = letters[rng.integers(0, sigma, size=n)]
            recs.append(np.ascontiguousarray(r, dtype=np.uint8))
        counts, arrays = native.factorize_batch(recs, want_factors=True)
        for j, r in enumerate(recs):
            assert _same(arrays[j], oracle.factors_array(r)), (cases, j, bytes(r))
        cases += 1
    assert cases >= 5


# ---- with reverse complement: record j as factorize_dna_w_rc(record j) -------------------------------

def _rc_expected(rec):
    S, _, _ = oracle.prepare_multiple_dna_w_rc([bytes(rec)])
    return oracle.factors_array_multiple_dna_w_rc(S)


def _palindromic(rng, n):
    """a record whose second half is the reverse complement of its first (rc factors for sure)"""
    comp = np.zeros(256, dtype=np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    half = gen.random_dna(max(1, n // 2), int(rng.integers(1 << 30)))
    return np.concatenate([half, comp[half[::-1]]])


def test_rc_merged_equals_oracle_record_by_record(native):
    rng = np.random.default_rng(21)
    recs = _records(rng, 200, 1, 2500)
    recs += [_palindromic(rng, int(rng.integers(2, 3000))) for _ in range(60)]
    recs += [recs[5].copy(), recs[205].copy(), recs[17][:100].copy()]
    recs += [np.frombuffer(c, dtype=np.uint8) for c in (b"A", b"T", b"AT", b"ACGT", b"A" * 300, b"AT" * 200, b"acgtTTgca")]
    merged0, single0 = native.debug_batch_counters()
    counts, arrays = native.factorize_batch(recs, want_factors=True, with_rc=True)
    assert native.debug_batch_counters() == (merged0 + len(recs), single0)
    n_rc = 0
    for j, r in enumerate(recs):
        exp = _rc_expected(r)
        assert counts[j] == len(exp), j
        assert _same(arrays[j], exp), j
        n_rc += int((exp["ref"] >> 63).sum())
    assert n_rc > 100  # reverse-complement factors were exercised
    counts2, none = native.factorize_batch(recs, want_factors=False, with_rc=True)
    assert counts2 == counts and none is None


def test_rc_merged_equals_one_by_one(native):
    rng = np.random.default_rng(22)
    recs = _records(rng, 30, 5000, 120000) + [_palindromic(rng, 50000)]
    with merge_below(0):
        c1, a1 = native.factorize_batch(recs, want_factors=True, with_rc=True)
    merged0, _ = native.debug_batch_counters()
    c2, a2 = native.factorize_batch(recs, want_factors=True, with_rc=True)
    assert native.debug_batch_counters()[0] == merged0 + len(recs)
    assert c1 == c2
    for j in range(len(recs)):
        assert _same(a1[j], a2[j]), j
    assert _same(a2[30], _rc_expected(recs[30]))


def test_rc_invalid_nucleotide_fails_like_the_reference(native):
    recs = [gen.random_dna(100, 1), np.frombuffer(b"ACGTNACGT", dtype=np.uint8), gen.random_dna(50, 2)]
    with pytest.raises(RuntimeError, match="Invalid nucleotide 'N'"):
        native.factorize_batch(recs, want_factors=True, with_rc=True)


def test_rc_differential_fuzz_of_tiny_records(native):
    import time
    rng = np.random.default_rng(23)
    t_end = time.time() + 6.0
    cases = 0
    while time.time() < t_end:
        sigma = int(rng.integers(1, 5))
        letters = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.permutation(4)[:sigma]]
        m = int(rng.integers(2, 300))
        recs = []
        for _ in range(m):
            kind = rng.integers(0, 5)
            n = int(rng.integers(1, 60))
            if kind == 0 and recs:
                src = recs[int(rng.integers(0, len(recs)))]
                r = src[:max(1, int(rng.integers(1, len(src) + 1)))].copy()
            elif kind == 1:
                unit = letters[rng.integers(0, sigma, size=int(rng.integers(1, 5)))]
                r = np.tile(unit, n)[:max(1, n)]
            elif kind == 2:
                r = _palindromic(rng, n + 1)
            else:
                r = letters[rng.integers(0, sigma, size=n)]
            recs.append(np.ascontiguousarray(r, dtype=np.uint8))
        counts, arrays = native.factorize_batch(recs, want_factors=True, with_rc=True)
        for j, r in enumerate(recs):
            assert _same(arrays[j], _rc_expected(r)), (cases, j, bytes(r))
        cases += 1
    assert cases >= 5


@pytest.mark.parametrize("count", [255, 256, 257, 258, 512, 513])
def test_record_counts_around_table_and_key_width_boundaries(native, count):
    """256 / 257 terminators switch the coarse terminator index on; powers of two change the number of
    record bits in the sort key"""
    rng = np.random.default_rng(300 + count)
    recs = _records(rng, count, 1, 400)
    counts, arrays = native.factorize_batch(recs, want_factors=True)
    for j, r in enumerate(recs):
        assert _same(arrays[j], oracle.factors_array(r)), j
    half = recs[:(count + 1) // 2]  # with rc the table holds two terminators per record
    counts, arrays = native.factorize_batch(half, want_factors=True, with_rc=True)
    for j, r in enumerate(half):
        assert _same(arrays[j], _rc_expected(r)), j


@pytest.mark.timeout(600)
def test_several_runs_in_flight(native):
    """~90 Mi bases of records: three merged runs, two of them in flight at a time on their own lanes"""
    rng = np.random.default_rng(31)
    base = gen.repeat_dna(1 << 22, seed=5)
    lens = rng.integers(20000, 130000, size=1200)
    offs = rng.integers(0, (1 << 22) - 130000, size=1200)
    recs = [base[o:o + l] for o, l in zip(offs.tolist(), lens.tolist())]
    assert int(lens.sum()) > (2 << 25)
    merged0, single0 = native.debug_batch_counters()
    counts, arrays = native.factorize_batch(recs, want_factors=True)
    assert native.debug_batch_counters() == (merged0 + len(recs), single0)
    with merge_below(0):
        c1, _ = native.factorize_batch(recs, want_factors=False)
    assert c1 == counts
    for j in rng.choice(len(recs), size=40, replace=False).tolist():
        assert _same(arrays[j], oracle.factors_array(recs[j])), j
    crc, arc = native.factorize_batch(recs[:500], want_factors=True, with_rc=True)
    for j in rng.choice(500, size=20, replace=False).tolist():
        assert _same(arc[j], _rc_expected(recs[j])), j


@pytest.mark.parametrize("extra_env", [{}, {"NOLZSS_LOCAL_SORT_MIN": "1"},
                                       {"NOLZSS_LOCAL_SORT_MIN": "1", "NOLZSS_TEST_LOCAL_ORDER_FAILS": "1"}],
                         ids=["segmented-passes", "sub-buckets-in-lds", "sub-buckets-redone"])
def test_record_bucket_sort_on_small_records(extra_env):
    """Runs of LONG records (on average >= 2^16 bases) sort with the records as the buckets of the segmented
    key sort (radix_sort.hip, radix_sort_record_keys; key = 14 bases + length tag).  One child process with
    NOLZSS_REC_BUCKET_MIN=1 and NOLZSS_DNA_FAST_MIN=1 sends small runs through it: records shorter than a tile
    and shorter than the key, records that end inside the key window, copies of each other, one-base records,
    record counts around 256; host-buffer batch (every factor against the oracle) and the device-resident
    batch (counts, with and without records built in device memory).  NOLZSS_LOCAL_SORT_MIN=1: the form for
    records of a megabase and more (most significant digit from the text, the 256 sub-buckets of every record
    sorted in LDS by local_sort_kernel; a record of 30 000 A's overflows a workgroup and takes the segmented
    passes); NOLZSS_TEST_LOCAL_ORDER_FAILS: every sub-bucket redone by those passes."""
    import subprocess
    import sys
    code = r'''
import sys
sys.path.insert(0, "tests")
import numpy as np, torch
import gen, oracle_lib as oracle
from nolzss_amd import _noLZSS as native
rng = np.random.default_rng(21)
def recs_of(count, lo, hi):
    out = []
    for k in range(count):
        n = int(rng.integers(lo, hi + 1))
        out.append(gen.repeat_dna(n, seed=9000 + k) if n >= 64 else gen.random_dna(n, seed=9000 + k))
    return out
sets = [recs_of(2, 5000, 9000), recs_of(7, 1, 40), recs_of(60, 1, 6000), recs_of(255, 10, 300), recs_of(256, 10, 300),
        recs_of(257, 10, 300), recs_of(9, 4090, 4100), recs_of(3, 60000, 140000)]
base = gen.repeat_dna(7000, seed=5)
sets.append([base, base.copy(), base[:3000].copy(), base[100:].copy(), np.frombuffer(b"A" * 5000, dtype=np.uint8),
             np.frombuffer(b"ACGT" * 1100, dtype=np.uint8), np.frombuffer(b"ACGTTGCATTGACC", dtype=np.uint8),
             np.frombuffer(b"ACGTTGCATTGAC", dtype=np.uint8), np.frombuffer(b"ACGTTGCATTGACCA", dtype=np.uint8)]
            + [np.frombuffer(c, dtype=np.uint8) for c in (b"A", b"C", b"G", b"T", b"AA", b"T" * 29)])
sets.append([np.frombuffer(b"A" * 30000, dtype=np.uint8), gen.repeat_dna(40000, seed=6), np.frombuffer(b"AC" * 25000, dtype=np.uint8)])
total = 0
for recs in sets:
    m0, s0 = native.debug_batch_counters()
    counts, arrays = native.factorize_batch(recs, want_factors=True)
    m1, s1 = native.debug_batch_counters()
    assert (m1 - m0, s1 - s0) == (len(recs), 0), (m1 - m0, s1 - s0)
    for j, r in enumerate(recs):
        exp = oracle.factors_array(r)
        assert counts[j] == len(exp), (len(recs), j)
        for k in ("start", "length", "ref"):
            assert np.array_equal(arrays[j][k], exp[k]), (len(recs), j, k)
    d = [torch.from_numpy(np.ascontiguousarray(r)).cuda() for r in recs]
    torch.cuda.synchronize()
    for emit in (0, 1):
        m0, s0 = native.debug_batch_counters()
        got = native.factorize_batch_device([t.data_ptr() for t in d], [len(r) for r in recs], emit=emit)
        m1, s1 = native.debug_batch_counters()
        assert got == counts, (len(recs), emit)
        assert (m1 - m0, s1 - s0) == (len(recs), 0), (m1 - m0, s1 - s0)
    total += len(recs)
print("ok", total)
'''
    env = dict(os.environ, NOLZSS_DNA_FAST_MIN="1", NOLZSS_REC_BUCKET_MIN="1", NOLZSS_TEST_INJECT_PENDING="1", **extra_env)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_device_batch_merges_and_matches_one_by_one(native):
    """the device-resident batch gathers its records into runs of independent sequences on the device; counts
    equal the one-run-per-record path (NOLZSS_DEVICE_MERGE_BELOW is read once per process: the comparison run
    is nolzss_factorize_device per record), other alphabets fall back to it, empty records count zero"""
    import torch
    rng = np.random.default_rng(31)
    recs = _records(rng, 24, 30000, 200000) + [np.zeros(0, dtype=np.uint8), gen.random_dna(1, 3)]
    d = [torch.from_numpy(np.ascontiguousarray(r)).cuda() for r in recs]
    torch.cuda.synchronize()
    ptrs = [t.data_ptr() if len(r) else 0 for t, r in zip(d, recs)]
    lens = [len(r) for r in recs]
    one_by_one = [native.factorize_device(p, n, emit=1)[0] if n else 0 for p, n in zip(ptrs, lens)]
    m0, s0 = native.debug_batch_counters()
    for emit in (0, 1):
        assert native.factorize_batch_device(ptrs, lens, emit=emit) == one_by_one
    m1, s1 = native.debug_batch_counters()
    assert m1 - m0 == 2 * (len(recs) - 1)  # everything but the empty record went through merged runs
    # a record with another alphabet: the whole run falls back to one run per record
    other = recs[:3] + [np.frombuffer(b"ACGTNACGT" * 50, dtype=np.uint8)]
    d2 = [torch.from_numpy(np.ascontiguousarray(r)).cuda() for r in other]
    torch.cuda.synchronize()
    got = native.factorize_batch_device([t.data_ptr() for t in d2], [len(r) for r in other], emit=0)
    assert got == [oracle.count_factors(r) for r in other]


def test_far_ranks_that_need_the_exact_search(native):
    """Regression (round 3, found by tools/fuzz_batch.py 25 17, case 131): a merged run whose far ranks -- searches
    that leave the candidate kernel's LDS reach -- mostly need the exact search (periodic records).  A far kernel that
    appended them to the sharded exact queue in list order overflowed a shard's region; they now come back through a
    list of their own (lpnf.hip, far_exact)."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    import fuzz_batch as fb
    rng = np.random.default_rng(17)
    for _ in range(132):
        recs = fb.make_records(rng)
    counts, arrays = native.factorize_batch(recs, want_factors=True)
    for j, r in enumerate(recs):
        exp = oracle.factors_array(r)
        assert counts[j] == len(exp), j
        for k in ("start", "length", "ref"):
            assert np.array_equal(arrays[j][k], exp[k]), (j, k)
    # the same records as one text each: long periodic texts whose far ranks all need the exact search
    for t in (b"CT" * 40_000, b"CCCT" * 30_000 + b"G" + b"CCCT" * 30_000):
        got, exp = native.factorize_array(t), oracle.factors_array(t)
        assert len(got) == len(exp) and all(np.array_equal(got[k], exp[k]) for k in ("start", "length", "ref"))

