"""GPU parity at the BASELINE.json sizes: exact against the oracle where the oracle finishes in
seconds (config 2, FASTA records, a 4 Mi RC case), size-independent properties at the maximum sizes.
Configs 3 and 5 at their full size, every factor against the oracle: tests/test_zz_gpu_fullsize_exact.py."""
import numpy as np
import pytest

import gen
import oracle_lib as oracle

pytestmark = pytest.mark.gpu

COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGT", b"TGCA"):
    COMP[a] = b


@pytest.fixture(scope="module")
def native():
    from nolzss_amd import _noLZSS
    assert _noLZSS.device_count() >= 1, "no MI355X visible"
    return _noLZSS


def _check_tiling(f, n, start=0):
    assert f["start"][0] == start
    assert np.array_equal(f["start"][1:], f["start"][:-1] + f["length"][:-1])
    assert int(f["start"][-1] + f["length"][-1]) == n
    assert f["length"].min() >= 1


def _check_matches(text, f, sample, rng, rc_mode=False):
    idx = rng.choice(len(f), size=min(sample, len(f)), replace=False)
    for k in idx.tolist():
        s, l, r = int(f["start"][k]), int(f["length"][k]), int(f["ref"][k])
        is_rc = bool(r >> 63)
        r &= (1 << 63) - 1
        if not is_rc and r == s:
            assert l == 1
            continue
        assert r + l <= s, (s, l, r, is_rc)
        src = text[r:r + l]
        if is_rc:
            assert rc_mode
            src = COMP[src[::-1]]
        assert np.array_equal(src, text[s:s + l]), (s, l, r, is_rc)


@pytest.mark.timeout(900)
def test_config2_random_64Mi_exact(native):
    """BASELINE config 2: 64 Mi iid ACGT, every factor compared with the oracle."""
    n = 1 << 26
    text = gen.random_dna(n)
    got = native.factorize_array(text)
    exp = oracle.factors_array(text)
    assert len(got) == len(exp)
    for k in ("start", "length", "ref"):
        assert np.array_equal(got[k], exp[k]), k
    assert native.count_factors(text) == len(exp)


@pytest.mark.timeout(900)
def test_config4_fasta_records_batch(native, tmp_path):
    """BASELINE config 4 shape at reduced record count (16 x 4 Mi instead of 512 x 4 Mi; the
    full count only repeats the same independent unit): batch / FASTA path vs oracle."""
    from nolzss_amd.genomics import read_nucleotide_fasta
    recs = gen.fasta_records(16, 1 << 22)
    counts, arrays = native.factorize_batch([s for _, s in recs], want_factors=True)
    for j in (0, 7, 15):
        exp = oracle.factors_array(recs[j][1])
        assert counts[j] == len(exp)
        for k in ("start", "length", "ref"):
            assert np.array_equal(arrays[j][k], exp[k])
    counts2, none = native.factorize_batch([s for _, s in recs], want_factors=False)
    assert counts2 == counts and none is None
    small = [(f"s{k}", gen.random_dna(3000 + 17 * k, 900 + k)) for k in range(5)]
    path = tmp_path / "small.fa"
    gen.write_fasta(path, small)
    res = read_nucleotide_fasta(path)
    assert [rid for rid, _ in res] == [rid for rid, _ in small]
    for (rid, factors), (_, seq) in zip(res, small):
        assert factors == oracle.factorize(seq)


@pytest.mark.timeout(1500)
def test_config4_fasta_512_records_end_to_end(native, tmp_path):
    """BASELINE config 4 at its stated size through its stated caller: a FASTA file of 512 records x 4 Mi
    bases (generator of config 2, seeds 0x4000 + k, 80-column lines, 2.2 GB) through the
    read_nucleotide_fasta path (file -> native reader -> per-sequence batch; reference:
    genomics/fasta.py:79-126) on one GPU.  ALL 512 factor counts are checked against the oracle run per
    record in a host thread pool, the complete factor lists of a sample of records bit by bit, every record
    for tiling; the tuple-list form of the API on the first records of the same file."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    from pathlib import Path
    from nolzss_amd.genomics import read_nucleotide_fasta
    from nolzss_amd.genomics.fasta import read_nucleotide_fasta_arrays
    m, L = 512, 1 << 22
    workers = min(16, os.cpu_count() or 8)
    with ThreadPoolExecutor(max_workers=workers) as pool:
        seqs = list(pool.map(lambda k: gen.random_dna(L, 0x4000 + k), range(m)))
    shm = Path("/dev/shm")
    path = (shm if shm.is_dir() else tmp_path) / f"nolzss_config4_{os.getpid()}.fa"
    try:
        gen.write_fasta_fast(path, [(f"seq{k}", s) for k, s in enumerate(seqs)])
        assert path.stat().st_size > 2_170_000_000
        ids, counts, arrays = read_nucleotide_fasta_arrays(path)
        ids2, counts2, none = read_nucleotide_fasta_arrays(path, want_factors=False)
    finally:
        path.unlink(missing_ok=True)
    assert ids == ids2 == [f"seq{k}" for k in range(m)] and counts2 == counts and none is None
    with ThreadPoolExecutor(max_workers=workers) as pool:  # (the oracle releases the GIL: ctypes)
        expected = list(pool.map(oracle.count_factors, seqs))
    assert counts == expected
    for k in range(m):
        assert len(arrays[k]) == counts[k]
        _check_tiling(arrays[k], L)
    for k in (0, 1, 100, 255, 256, 300, 510, 511):
        exp = oracle.factors_array(seqs[k])
        for key in ("start", "length", "ref"):
            assert np.array_equal(arrays[k][key], exp[key]), (k, key)
    # the reference's result shape (lists of int tuples) on the head of the same data
    head = tmp_path / "head.fa"
    gen.write_fasta_fast(head, [(f"seq{k}", seqs[k]) for k in range(3)])
    res = read_nucleotide_fasta(head)
    assert [rid for rid, _ in res] == ["seq0", "seq1", "seq2"]
    for k, (_, tuples) in enumerate(res):
        assert len(tuples) == counts[k] and tuples[:1000] == list(zip(arrays[k]["start"][:1000].tolist(),
                                                                       arrays[k]["length"][:1000].tolist(),
                                                                       arrays[k]["ref"][:1000].tolist()))
        assert tuples[-1] == tuple(int(arrays[k][key][-1]) for key in ("start", "length", "ref"))


@pytest.mark.timeout(900)
def test_rc_4Mi_exact(native):
    text = gen.repeat_dna(1 << 22, seed=0x5EED0005, lo=16, hi=8192)
    got = native.factorize_dna_w_rc_array(text)
    S, _, _ = oracle.prepare_multiple_dna_w_rc([text.tobytes()])
    exp = oracle.factors_array_multiple_dna_w_rc(S)
    assert len(got) == len(exp)
    for k in ("start", "length", "ref"):
        assert np.array_equal(got[k], exp[k]), k


def _fib(n):
    a, b = b"a", b"ab"
    while len(b) < n:
        a, b = b, b + a
    return b[:n]


@pytest.mark.timeout(1200)
def test_beyond_2Gi_positions(native):
    """Maximum sizes: 2^31 + 2^21 bases, so starts, references and ranks exceed 2^31 (32-bit index
    arithmetic must be unsigned everywhere).  The last 2^20 bases copy a block that straddles
    position 2^31: the parse must end with that factor.  Plus tiling, sampled true-match
    checks and the exact prefix of the parse."""
    n = (1 << 31) + (1 << 21)
    text = gen.random_dna(n, seed=77)
    src = (1 << 31) - (1 << 19)
    text[n - (1 << 20):] = text[src:src + (1 << 20)]
    f = native.factorize_array(text)
    _check_tiling(f, n)
    # (the factor in front may run a few bases into the copy by chance)
    k = int(f["start"][-1]) - (n - (1 << 20))
    assert 0 <= k < 64
    assert (int(f["length"][-1]), int(f["ref"][-1])) == ((1 << 20) - k, src + k)
    _check_matches(text, f, 20_000, np.random.default_rng(3))
    tail = np.flatnonzero(f["start"] > np.uint64(1 << 31))
    _check_matches(text, f[tail[0]:], 20_000, np.random.default_rng(4))
    P = 1 << 23
    exp = oracle.factors_array(text[:P])
    cut = int(np.searchsorted(f["start"] + f["length"], P, side="right"))
    assert cut > 100_000 and cut <= len(exp)
    for k in ("start", "length", "ref"):
        assert np.array_equal(f[k][:cut], exp[k][:cut]), k
    assert native.count_factors(text) == len(f)


@pytest.mark.timeout(1500)
def test_genome_scale_capacity(native):
    """The documented capacity of one MI355X: a 3.2 Gbase text in plain mode -- the
    size of a human genome, 49 bytes per base of device memory at the peak, 52 reserved at the least -- and a
    1.55 Gbase text with its reverse complement (3.1 G symbols).  Tiling, sampled true-match checks, the
    exact prefix of the parse against the oracle, and an input beyond the limits refused up front."""
    n = 3_200_000_000
    text = gen.random_dna(n, seed=91)
    text[n - (1 << 20):] = text[5:5 + (1 << 20)]          # the parse must end with (about) this copy
    f = native.factorize_array(text)
    _check_tiling(f, n)
    assert int(f["length"][-1]) > (1 << 20) - 64 and abs(int(f["ref"][-1]) - 5) < 64
    _check_matches(text, f, 20_000, np.random.default_rng(5))
    far = np.flatnonzero(f["start"] > np.uint64(3_000_000_000))
    _check_matches(text, f[far[0]:], 20_000, np.random.default_rng(6))
    P = 1 << 22
    exp = oracle.factors_array(text[:P])
    cut = int(np.searchsorted(f["start"] + f["length"], P, side="right"))
    assert cut > 100_000 and cut <= len(exp)
    for k in ("start", "length", "ref"):
        assert np.array_equal(f[k][:cut], exp[k][:cut]), k
    z = len(f)
    del f
    assert native.count_factors(text) == z
    cap, peak = native.debug_arena()
    assert peak <= 52 * n + (64 << 20), (cap, peak)
    # reverse-complement mode at 1.55 Gbases: S has 3.1 G symbols
    n2 = 1_550_000_000
    f2 = native.factorize_dna_w_rc_array(text[:n2])
    _check_tiling(f2, n2)
    _check_matches(text[:n2], f2, 20_000, np.random.default_rng(7), rc_mode=True)
    assert (f2["ref"] >> np.uint64(63)).any()
    del f2
    # beyond the 32-bit index range: an argument error up front, not a device out-of-memory error
    with pytest.raises(ValueError, match="text too long"):
        native.count_factors_dna_w_rc(np.zeros((1 << 31) + 5, dtype=np.uint8))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", ["allA_8M", "period3_4M", "fib_4M", "abracadabra_x400k", "two_long_copies_8M",
                                  "long_copy_with_edits_6M"])
def test_pathological_repeats(native, name):
    """Huge LCPs (runs, periodic texts, long exact duplications): the direct round bails out,
    doubling rounds + range-minimum LCPs take over; results must still be bit-exact and fast."""
    import time
    if name == "allA_8M":
        t = b"A" * (1 << 23)
    elif name == "period3_4M":
        t = (b"ACG" * ((1 << 22) // 3 + 1))[:1 << 22]
    elif name == "fib_4M":
        t = _fib(1 << 22)
    elif name == "abracadabra_x400k":
        t = b"abracadabra" * 400_000
    elif name == "two_long_copies_8M":
        x = gen.random_dna(1 << 22, 71).tobytes()
        t = x + x
    else:
        x = bytearray(gen.random_dna(3 << 20, 72).tobytes())
        y = bytearray(x)
        for p in range(1000, len(y), 400_003):   # a handful of point edits in the second copy
            y[p] = ord("A") if y[p] != ord("A") else ord("C")
        t = bytes(x + y)
    native.count_factors(b"ACGT" * 100)
    t0 = time.time()
    got = native.factorize_array(t)
    dt = time.time() - t0
    exp = oracle.factors_array(t)
    assert len(got) == len(exp)
    for k in ("start", "length", "ref"):
        assert np.array_equal(got[k], exp[k]), k
    assert dt < 5.0, f"{name}: {dt:.2f} s"


@pytest.mark.timeout(900)
@pytest.mark.parametrize("copies,base_log2", [(24, 21), (96, 19)])
def test_collection_of_similar_genomes_at_scale(native, copies, base_log2):
    """24 genomes of 2 Mi bases and 96 of 512 Ki, 0.1 % apart (the reference's use case for collections,
    /root/reference/src/cpp/fasta_processor.cpp:298-341) at a size where the default thresholds decide the path: the
    16-base key sort, the first direct round, the pivot PASSES (2048, 8192, 32768 symbols deep: group_sort.hpp, kPivot)
    and the inverse suffix array delivered by the permutation of the codes -- no environment switch.  Every factor,
    the suffix array and the LCP array against the oracle."""
    rng = np.random.default_rng(7000 + copies)
    base = gen.random_dna(1 << base_log2, 600 + copies)
    parts = [base]
    for _ in range(copies - 1):
        y = base.copy()
        idx = rng.integers(0, len(y), size=len(y) // 1000)
        y[idx] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=len(idx))]
        parts.append(y)
    t = np.concatenate(parts)
    got = native.factorize_array(t)
    exp = oracle.factors_array(t)
    assert len(got) == len(exp)
    for k in ("start", "length", "ref"):
        assert np.array_equal(got[k], exp[k]), k
    d = native.debug_arrays(t)
    sa = oracle.suffix_array(t)
    assert np.array_equal(d["sa"].astype(np.int64), sa.astype(np.int64))
    assert np.array_equal(d["lcp"][:len(t)].astype(np.int64), oracle.lcp_array(t, sa).astype(np.int64))


@pytest.mark.timeout(900)
def test_rc_long_palindromic_repeat(native):
    """reverse-complement mode with a long exact inverted repeat (LCP in the joint text ~ 1 Mi)"""
    x = gen.random_dna(1 << 20, 73)
    t = np.concatenate([x, gen.random_dna(1000, 74), COMP[x[::-1]]])
    got = native.factorize_dna_w_rc_array(t)
    S, _, _ = oracle.prepare_multiple_dna_w_rc([t.tobytes()])
    exp = oracle.factors_array_multiple_dna_w_rc(S)
    assert len(got) == len(exp)
    for k in ("start", "length", "ref"):
        assert np.array_equal(got[k], exp[k]), k
    assert (got["ref"] >> np.uint64(63)).sum() >= 1


@pytest.mark.timeout(900)
@pytest.mark.parametrize("kind", ["protein_4Mi", "bytes_2Mi", "sigma12_repeats_3Mi"])
def test_other_alphabets_at_scale(native, kind):
    """4-bit and 8-bit packed texts (5 to 8 sort passes on 64-bit keys, direct round on wider
    symbols) at sizes where every kernel runs many tiles; exact against the oracle."""
    rng = np.random.default_rng(11)
    if kind == "protein_4Mi":
        alphabet = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
        t = alphabet[rng.integers(0, 20, size=1 << 22)]
        t[1 << 21:(1 << 21) + 300_000] = t[1000:301_000]  # one long copy
    elif kind == "bytes_2Mi":
        t = rng.integers(1, 256, size=1 << 21, dtype=np.uint8)
        t[1 << 20:(1 << 20) + 100_000] = t[5:100_005]
    else:
        alphabet = np.frombuffer(b"abcdefghijkl", dtype=np.uint8)
        base = alphabet[rng.integers(0, 12, size=50_000)]
        t = np.concatenate([base if rng.random() < 0.5 else alphabet[rng.integers(0, 12, size=50_000)]
                            for _ in range(60)])
    got = native.factorize_array(t)
    exp = oracle.factors_array(t)
    assert len(got) == len(exp)
    for k in ("start", "length", "ref"):
        assert np.array_equal(got[k], exp[k]), k


@pytest.mark.timeout(600)
def test_differential_fuzz_short():
    """tools/fuzz.py for a few seconds: random texts of many shapes (tandem repeats, runs, copies with
    edits, small and large alphabets, prepared multi-sequence strings), plain and RC, against the
    oracle; the child process also takes the bucketed sort for every 2-bit text."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (third run: RC mode hands only the ranks of the original strand to its permutation whatever the size, rc.hip)
    for env_extra, seed in (({}, "11"), ({"NOLZSS_DNA_FAST_MIN": "1"}, "12"), ({"NOLZSS_RC_COMPACT_MIN": "1"}, "13")):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz.py"), "8", seed], cwd=root,
                           env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "no mismatch" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_file_entry_points_read_large_files_in_pieces(native, tmp_path):
    """files beyond 16 MiB are read by several threads in 16 MiB pieces: same result as the buffer"""
    text = gen.repeat_dna((40 << 20) + 12345, seed=77)
    path = tmp_path / "big.txt"
    path.write_bytes(text.tobytes())
    z = native.count_factors(text)
    assert native.count_factors_file(str(path)) == z
    f = native.factorize_file(str(path))
    assert len(f) == z and f[0] == (0, 1, 0)
    assert sum(l for _, l, _ in f) == len(text)
