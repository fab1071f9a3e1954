"""GPU parity of the "next" rows: reference + target factorization and
the v2 binary factor files, against the oracle (start_pos form) and the format definition."""
import random
import struct

import numpy as np
import pytest

import gen
import oracle_lib as oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import nolzss_amd
    from nolzss_amd import _noLZSS
    assert _noLZSS.device_count() >= 1, "no MI355X visible"
    return nolzss_amd


def test_factorize_w_reference(pkg):
    """reference semantics: factorizer.cpp:940-955; tests/test_reference_seq.py:198-250"""
    rng = random.Random(3)
    for _ in range(20):
        ref = "".join(rng.choice("abcdefgh") for _ in range(rng.randint(1, 400)))
        tgt = "".join(rng.choice("abcdefgh") for _ in range(rng.randint(1, 400)))
        got = pkg.factorize_w_reference(ref, tgt)
        combined = (ref + "\x01" + tgt).encode()
        assert got == oracle.factorize(combined, start_pos=len(ref) + 1)
        assert got[0][0] == len(ref) + 1 and got[-1][0] + got[-1][1] == len(combined)
    big_ref = gen.repeat_dna(300_000, 41, lo=16, hi=2048).tobytes().decode()
    big_tgt = big_ref[1000:90_000] + gen.random_dna(50_000, 42).tobytes().decode()
    got = pkg.factorize_w_reference(big_ref, big_tgt)
    assert got == oracle.factorize((big_ref + "\x01" + big_tgt).encode(), start_pos=len(big_ref) + 1)
    assert got[0][1] > 80_000  # the copied stretch is one factor pointing into the reference


def test_factorize_dna_w_reference_seq(pkg):
    """reference semantics: factorizer.cpp:825-842; tests/test_reference_seq.py:43-120"""
    from nolzss_amd.genomics import factorize_dna_w_reference_seq
    rng = random.Random(4)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    for _ in range(15):
        ref = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 300)))
        tgt = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 200)))
        if rng.random() < 0.5 and len(ref) > 20:
            tgt += "".join(comp[c] for c in reversed(ref[5:20]))
        got = factorize_dna_w_reference_seq(ref, tgt)
        S, _, _ = oracle.prepare_multiple_dna_w_rc([ref, tgt])
        assert got == oracle.factorize_multiple_dna_w_rc(S, start_pos=len(ref) + 1)
    with pytest.raises(ValueError):
        factorize_dna_w_reference_seq("ACGT", "ACGN")


def _footer(path):
    raw = open(path, "rb").read()
    magic, nf, nseq, nsent, fsize, total = struct.unpack("<8sQQQQQ", raw[-48:])
    return raw, magic, nf, nseq, nsent, fsize, total


def test_write_factors_binary_file_roundtrip(pkg, tmp_path):
    """format: factorizer.hpp:64-77, factorizer.cpp:424-459; reader utils.py:106-155"""
    from nolzss_amd import _noLZSS
    text = gen.repeat_dna(100_000, 51, lo=16, hi=1024).tobytes()
    src = tmp_path / "in.txt"
    src.write_bytes(text)
    out = tmp_path / "out.bin"
    z = _noLZSS.write_factors_binary_file(str(src), str(out))
    exp = oracle.factorize(text)
    assert z == len(exp)
    raw, magic, nf, nseq, nsent, fsize, total = _footer(out)
    assert (magic, nf, nseq, nsent, fsize, total) == (b"noLZSSv2", z, 0, 0, 48, len(text))
    assert len(raw) == 24 * z + 48
    assert pkg.read_factors_binary_file(out) == exp
    meta = pkg.read_binary_file_metadata(out)
    assert meta["num_factors"] == z and meta["sequence_names"] == [] and meta["total_length"] == len(text)
    # the Python wrapper keeps the reference's data-as-path behaviour (core.py:110-132)
    out2 = tmp_path / "sub" / "out2.bin"
    pkg.write_factors_binary_file(str(src), out2)
    assert out2.read_bytes() == raw
    with pytest.raises(RuntimeError):
        _noLZSS.write_factors_binary_file(str(tmp_path / "missing"), str(out))


def test_write_factors_binary_file_dna_w_rc(pkg, tmp_path):
    """factorizer.cpp:597-635: one empty sequence name, num_sequences = 1"""
    from nolzss_amd import _noLZSS
    text = gen.repeat_dna(60_000, 52, lo=16, hi=1024).tobytes()
    src = tmp_path / "dna.txt"
    src.write_bytes(text)
    out = tmp_path / "dna.bin"
    z = _noLZSS.write_factors_binary_file_dna_w_rc(str(src), str(out))
    raw, magic, nf, nseq, nsent, fsize, total = _footer(out)
    assert (magic, nf, nseq, nsent, fsize, total) == (b"noLZSSv2", z, 1, 0, 49, len(text))
    meta = pkg.read_factors_binary_file_with_metadata(out)
    assert meta["sequence_names"] == [""] and meta["factors"] == oracle.factorize_dna_w_rc(text)


def test_reference_files(pkg, tmp_path):
    """factorizer.cpp:980-1021 and :851-883: num_sequences = 2, num_sentinels = 1,
    total_length = |target|"""
    from nolzss_amd.genomics import factorize_dna_w_reference_seq_file, factorize_dna_w_reference_seq
    ref, tgt = "abcabcabcxyz" * 30, "xyzabc" * 25
    out = tmp_path / "a" / "gen.bin"
    z = pkg.factorize_w_reference_file(ref, tgt, out)
    raw, magic, nf, nseq, nsent, fsize, total = _footer(out)
    assert (nf, nseq, nsent, fsize, total) == (z, 2, 1, 48, len(tgt))
    assert pkg.read_factors_binary_file(out) == pkg.factorize_w_reference(ref, tgt)
    dref, dtgt = gen.random_dna(5000, 61).tobytes().decode(), gen.random_dna(3000, 62).tobytes().decode()
    out2 = tmp_path / "dna_ref.bin"
    z2 = factorize_dna_w_reference_seq_file(dref, dtgt, out2)
    raw, magic, nf, nseq, nsent, fsize, total = _footer(out2)
    assert (nf, nseq, nsent, fsize, total) == (z2, 2, 1, 48, len(dtgt))
    rc_mask = 1 << 63
    got = [(s, l, r & (rc_mask - 1), bool(r & rc_mask)) for s, l, r in pkg.read_factors_binary_file(out2)]
    assert got == factorize_dna_w_reference_seq(dref, dtgt)
    # reference quirk kept: these files declare 2 sequences but carry no names, so the metadata
    # reader rejects them (the reference's own reader, utils.py:214-221, does the same)
    with pytest.raises(pkg.NoLZSSError):
        pkg.read_binary_file_metadata(out2)


def _write_fasta(path, recs, width=60):
    with open(path, "w") as f:
        for rid, seq in recs:
            f.write(f">{rid} some description\n")
            for i in range(0, len(seq), width):
                f.write(seq[i:i + width] + "\n")


def test_factorize_fasta_multiple_dna(pkg, tmp_path):
    """fasta_processor.cpp:28-163, 298-341 -- parse, prepare, factorize the
    concatenation, identify the sentinel literals"""
    from nolzss_amd import _noLZSS
    rng = random.Random(9)
    recs = [(f"chr{k}", "".join(rng.choice("ACGT") for _ in range(rng.randint(50, 900)))) for k in range(6)]
    recs[3] = (recs[3][0], recs[1][1][10:300] + recs[3][1])       # shared material across sequences
    path = tmp_path / "multi.fa"
    _write_fasta(path, recs)
    seqs = [s for _, s in recs]

    factors, sentinels, ids = _noLZSS.factorize_fasta_multiple_dna_w_rc(str(path))
    S, orig, sent_pos = oracle.prepare_multiple_dna_w_rc(seqs)
    assert factors == oracle.factorize_multiple_dna_w_rc(S)
    assert ids == [rid for rid, _ in recs]
    assert [factors[i][0] for i in sentinels] == sent_pos[:len(seqs) - 1]   # those inside [0, N)
    assert all(factors[i][1] == 1 and factors[i][2] == factors[i][0] for i in sentinels)

    factors2, sentinels2, ids2 = _noLZSS.factorize_fasta_multiple_dna_no_rc(str(path))
    S2, _, sent2 = _noLZSS.prepare_multiple_dna_sequences_no_rc_bytes(seqs)
    assert [(s, l, r) for s, l, r, rc in factors2] == oracle.factorize(S2)
    assert not any(rc for _, _, _, rc in factors2)
    assert [factors2[i][0] for i in sentinels2] == sent2 and ids2 == ids

    # sanitisation modes (fasta_processor.cpp:86-98)
    dirty = tmp_path / "dirty.fa"
    dirty.write_text(">a\nACGTNNACGT\n>b\nacgtRYacgt\n>empty\n>c\nGGGG\n")
    f3, s3, ids3 = _noLZSS.factorize_fasta_multiple_dna_no_rc(str(dirty))
    assert ids3 == ["a", "b", "c"]
    S3, _, _ = _noLZSS.prepare_multiple_dna_sequences_no_rc_bytes(["ACGTACGT", "ACGTACGT", "GGGG"])
    assert [(s, l, r) for s, l, r, _ in f3] == oracle.factorize(S3)
    with pytest.raises(RuntimeError, match="Invalid nucleotide"):
        _noLZSS.factorize_fasta_multiple_dna_no_rc(str(dirty), "strict")
    hdr = tmp_path / "hdr.fa"
    hdr.write_text(">\nACGT\n")
    with pytest.raises(RuntimeError, match="Empty sequence header"):
        _noLZSS.factorize_fasta_multiple_dna_w_rc(str(hdr))


def test_fasta_binary_files_and_parallel_aliases(pkg, tmp_path):
    """write_fasta_metadata (parallel_fasta_processor.cpp:29-62) and SURVEY 8f.4: the parallel
    entry points give the sequential result (reference: tests/test_parallel_fasta.py:294-323)"""
    from nolzss_amd import _noLZSS, parallel
    recs = [(f"s{k}", gen.random_dna(3000 + 411 * k, 500 + k).tobytes().decode()) for k in range(3)]
    path = tmp_path / "three.fa"
    _write_fasta(path, recs)
    out = tmp_path / "three.bin"
    z = _noLZSS.write_factors_binary_file_fasta_multiple_dna_w_rc(str(path), str(out))
    factors, sentinels, ids = _noLZSS.factorize_fasta_multiple_dna_w_rc(str(path))
    meta = pkg.read_factors_binary_file_with_metadata(out)
    assert z == len(factors) and meta["factors"] == factors
    assert meta["sequence_names"] == ids and meta["sentinel_factor_indices"] == sentinels
    assert meta["total_length"] == sum(f[1] for f in factors)
    out2 = tmp_path / "three_par.bin"
    assert _noLZSS.parallel_write_factors_binary_file_fasta_multiple_dna_w_rc(str(path), str(out2), 4) == z
    assert out2.read_bytes() == out.read_bytes()

    text = gen.repeat_dna(250_000, 77, lo=16, hi=2048).tobytes()
    seq = oracle.factorize(text)
    got = parallel.parallel_factorize(text, num_threads=4)
    assert [tuple(f) for f in got] == seq
    pout = tmp_path / "par.bin"
    n_par = parallel.parallel_factorize_to_file(text, pout, num_threads=3, start_pos=1000)
    assert n_par == len(oracle.factorize(text, start_pos=1000))
    raw = pout.read_bytes()
    magic, nf, nseq, nsent, fsize, total = struct.unpack("<8sQQQQQ", raw[-48:])
    assert (magic, nseq, nsent, fsize, total) == (b"noLZSSv2", 0, 0, 48, len(text) - 1000)
    with pytest.raises(ValueError):
        _noLZSS.parallel_factorize_to_file(b"ACGT", str(pout), 2, 4)
    dout = tmp_path / "par_rc.bin"
    zr = parallel.parallel_factorize_dna_w_rc_to_file(text[:50_000], dout, num_threads=2)
    rc_mask = 1 << 63
    got_rc = [(s, l, r & (rc_mask - 1), bool(r & rc_mask)) for s, l, r in pkg.read_factors_binary_file(dout)]
    assert zr == len(got_rc) and got_rc == oracle.factorize_dna_w_rc(text[:50_000])


def test_per_sequence_fasta_and_ref_target_fasta(pkg, tmp_path):
    """fasta_processor.cpp:430-561 (per record; the no-rc variants drop the last base, kept) and
    :240-287, 362-378 (reference FASTA + target FASTA)"""
    from nolzss_amd import _noLZSS
    recs = [(f"id{k}/x extra words", gen.random_dna(1200 + 97 * k, 800 + k).tobytes().decode()) for k in range(4)]
    path = tmp_path / "ps.fa"
    with open(path, "w") as f:
        for rid, seq in recs:
            f.write(f">{rid}\n{seq}\n")
    ids_expected = [rid.split()[0] for rid, _ in recs]
    per_seq, ids = _noLZSS.factorize_fasta_dna_w_rc_per_sequence(str(path))
    assert ids == ids_expected
    for (_, seq), got in zip(recs, per_seq):
        assert got == oracle.factorize_dna_w_rc(seq.encode())
    per_seq2, ids2 = _noLZSS.factorize_fasta_dna_no_rc_per_sequence(str(path))
    for (_, seq), got in zip(recs, per_seq2):
        assert [(s, l, r) for s, l, r, _ in got] == oracle.factorize(seq[:-1].encode())   # last base dropped
    counts, ids3, total = _noLZSS.count_factors_fasta_dna_w_rc_per_sequence(str(path))
    assert counts == [len(x) for x in per_seq] and total == sum(counts) and ids3 == ids_expected
    counts2, _, total2 = _noLZSS.count_factors_fasta_dna_no_rc_per_sequence(str(path))
    assert counts2 == [len(x) for x in per_seq2] and total2 == sum(counts2)
    out_dir = tmp_path / "per" / "seq"
    assert _noLZSS.write_factors_binary_file_fasta_dna_w_rc_per_sequence(str(path), str(out_dir)) == total
    for rid, got in zip(ids_expected, per_seq):
        meta = pkg.read_factors_binary_file_with_metadata(out_dir / (rid.replace("/", "_") + ".bin"))
        assert meta["factors"] == got and meta["sequence_names"] == [rid] and meta["num_sentinels"] == 0

    ref_fa, tgt_fa = tmp_path / "ref.fa", tmp_path / "tgt.fa"
    ref_seqs = [gen.random_dna(2000, 901).tobytes().decode(), gen.random_dna(1500, 902).tobytes().decode()]
    tgt_seqs = [ref_seqs[0][100:900] + gen.random_dna(300, 903).tobytes().decode()]
    ref_fa.write_text("".join(f">r{k}\n{s}\n" for k, s in enumerate(ref_seqs)))
    tgt_fa.write_text("".join(f">t{k}\n{s}\n" for k, s in enumerate(tgt_seqs)))
    factors, sentinels, ids4 = _noLZSS.factorize_dna_rc_w_ref_fasta_files(str(ref_fa), str(tgt_fa))
    S, _, sent_pos = oracle.prepare_multiple_dna_w_rc(ref_seqs + tgt_seqs)
    start = sum(len(s) + 1 for s in ref_seqs)
    assert factors == oracle.factorize_multiple_dna_w_rc(S, start_pos=start)
    assert ids4 == ["r0", "r1", "t0"] and factors[0][0] == start and factors[0][1] >= 800
    out = tmp_path / "rt.bin"
    assert _noLZSS.write_factors_dna_w_reference_fasta_files_to_binary(str(ref_fa), str(tgt_fa), str(out)) == len(factors)
    meta = pkg.read_factors_binary_file_with_metadata(out)
    assert meta["factors"] == factors and meta["sequence_names"] == ids4

    # file variants of the RC entry points (factorizer.cpp:525-545, 575-577, 658-732, 751-790)
    dna = tmp_path / "dna.txt"
    dna.write_bytes(recs[0][1].encode())
    assert _noLZSS.factorize_file_dna_w_rc(str(dna)) == per_seq[0]
    assert _noLZSS.count_factors_file_dna_w_rc(str(dna)) == len(per_seq[0])
    prepared = tmp_path / "prepared.bin"
    prepared.write_bytes(S)
    assert _noLZSS.factorize_file_multiple_dna_w_rc(str(prepared)) == oracle.factorize_multiple_dna_w_rc(S)
    assert _noLZSS.count_factors_file_multiple_dna_w_rc(str(prepared)) == oracle.count_factors_multiple_dna_w_rc(S)
    pout = tmp_path / "prepared_out.bin"
    z = _noLZSS.write_factors_binary_file_multiple_dna_w_rc(str(prepared), str(pout))
    raw = pout.read_bytes()
    assert struct.unpack("<8sQQQQQ", raw[-48:]) == (b"noLZSSv2", z, 0, 0, 48, len(S))
    with pytest.raises(RuntimeError, match="Cannot open input file"):
        _noLZSS.factorize_file_dna_w_rc(str(tmp_path / "nope"))


def test_concurrent_host_threads(pkg):
    """The bindings release the GIL around the native call (bindings.cpp:70): several Python threads
    factorizing different texts at once must each get their own answer."""
    import threading
    from nolzss_amd import _noLZSS as native
    texts = [gen.repeat_dna(300_000 + 37_000 * k, seed=900 + k, lo=16, hi=4096) for k in range(6)]
    texts += [gen.random_dna(500_000 + 11 * k, seed=800 + k) for k in range(3)]
    expected = [oracle.count_factors(t) for t in texts]
    errors = []

    def worker(tid):
        try:
            for rep in range(3):
                for k in range(len(texts)):
                    j = (k + tid) % len(texts)
                    z = native.count_factors(texts[j])
                    if z != expected[j]:
                        errors.append((tid, rep, j, z, expected[j]))
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


def test_arenas_are_given_back_and_taken_again(pkg):
    """the device arenas are kept between calls; trimming them (what the library does by itself when a
    reservation fails) must leave every lane usable"""
    native = pkg._noLZSS
    recs = [gen.random_dna(1 << 18, 40 + k) for k in range(8)]
    import os
    os.environ["NOLZSS_BATCH_MERGE_BELOW"] = "0"   # one run per record: four lanes take arenas
    try:
        c1, _ = native.factorize_batch(recs, want_factors=False)
        assert native.debug_arena()[0] > 0
        released = native.debug_trim_arenas()
        assert released >= 4 * 96 * (1 << 18)   # four lanes, the worst-case reservation of a 2^18-base record each
        assert native.debug_arena()[0] == 0
        c2, _ = native.factorize_batch(recs, want_factors=False)
    finally:
        os.environ.pop("NOLZSS_BATCH_MERGE_BELOW", None)
    assert c1 == c2
    assert native.count_factors(recs[0]) == c1[0]


def test_factorize_device_is_ordered_behind_the_default_stream(pkg):
    """nolzss_factorize_device with stream = NULL runs on the library's own non-blocking stream; it must wait
    for what torch's default stream has queued (an asynchronous upload of the text, a kernel that writes it)
    without the caller synchronising.  Also: importing the package first and torch second
    must leave torch with its GPU (nolzss_amd/_lib.py shares torch's HIP runtime)."""
    import torch
    assert torch.cuda.is_available(), "torch lost the GPU to a second HIP runtime"
    native = pkg._noLZSS
    n = 1 << 24
    text = gen.repeat_dna(n, 77, lo=16, hi=4096)
    exp = oracle.count_factors(text)
    pinned = torch.from_numpy(text).pin_memory()
    for _ in range(3):
        d = torch.empty(n, dtype=torch.uint8, device="cuda")
        d.zero_()                                   # a kernel on the default stream ...
        d.copy_(pinned, non_blocking=True)          # ... and an asynchronous upload behind it
        z, _ = native.factorize_device(d.data_ptr(), n, emit=1)   # no torch.cuda.synchronize() in between
        assert z == exp
        del d


def test_native_fasta_shards_and_device_batch(pkg, tmp_path):
    """the sharded form of the native FASTA entry point (nolzss_read_nucleotide_fasta with shard_index /
    shard_count: what every rank of a multi-GPU job calls) and the device-resident batch of bench.py, on one
    GPU: the shards partition the records by the LPT plan and together give the counts of the unsharded call"""
    import torch
    native = pkg._noLZSS
    recs = [(f"r{k}", gen.repeat_dna(30_000 + 7_919 * (k % 5), 300 + k, lo=16, hi=2048)) for k in range(11)]
    path = tmp_path / "shards.fa"
    gen.write_fasta(path, recs)
    ids, lens, counts, owners, arrays = native.read_nucleotide_fasta_arrays(path)
    assert ids == [r for r, _ in recs] and lens == [len(s) for _, s in recs] and owners == [0] * len(recs)
    expected = [oracle.count_factors(s) for _, s in recs]
    assert counts == expected
    for a, (_, s) in zip(arrays, recs):
        e = oracle.factors_array(s)
        assert all(np.array_equal(a[k], e[k]) for k in ("start", "length", "ref"))
    world = 3
    seen = [0] * len(recs)
    total = [0] * len(recs)
    for rank in range(world):
        ids_r, lens_r, counts_r, owners_r, arrays_r = native.read_nucleotide_fasta_arrays(
            path, want_factors=True, shard_index=rank, shard_count=world)
        assert ids_r == ids and lens_r == lens
        assert owners_r == native.debug_lpt_plan(lens, world)          # the same plan on every rank
        for j, o in enumerate(owners_r):
            if o == rank:
                seen[j] += 1
                total[j] += counts_r[j]
                assert len(arrays_r[j]) == expected[j]
            else:
                assert counts_r[j] == 0 and arrays_r[j] is None
    assert seen == [1] * len(recs) and total == expected               # disjoint cover, same counts
    with pytest.raises(ValueError):
        native.read_nucleotide_fasta_arrays(path, shard_index=3, shard_count=3)
    # records resident in device memory (bench.py's fasta512 form)
    d = [torch.from_numpy(s).cuda() for _, s in recs]
    torch.cuda.synchronize()
    for emit in (0, 1):
        assert native.factorize_batch_device([t.data_ptr() for t in d], lens, emit=emit) == expected


def test_concurrent_calls_with_staged_downloads(pkg):
    """factor arrays of 32 MiB and more come down through pinned chunks emptied by several host threads
    (c_abi.hip, download_bytes), and blocks of 256 MiB and more are released by a detached thread: four callers at
    once, each on its own lane, get what a lone caller gets; results of a 2^24-base random text (1.4 M factors,
    34 MB of records) are also checked as a tiling with true earlier occurrences"""
    import threading
    native = pkg._noLZSS
    texts = [gen.random_dna((1 << 24) + 1000 * k, 7100 + k) for k in range(4)]
    alone = [native.factorize_array(t) for t in texts]
    for t, f in zip(texts, alone):
        assert len(f) * 24 >= 32 << 20
        assert int(f["start"][0]) == 0 and int(f["start"][-1] + f["length"][-1]) == len(t)
        assert np.array_equal(f["start"][1:], f["start"][:-1] + f["length"][:-1])
        for j in np.random.default_rng(1).integers(0, len(f), size=200):
            s, l, r = int(f["start"][j]), int(f["length"][j]), int(f["ref"][j])
            assert r + l <= s or l == 1
            if r != s:
                assert np.array_equal(t[s:s + l], t[r:r + l])
    errors = []

    def worker(k):
        try:
            for rep in range(3):
                j = (k + rep) % len(texts)
                got = native.factorize_array(texts[j])
                if not all(np.array_equal(got[c], alone[j][c]) for c in ("start", "length", "ref")):
                    errors.append((k, rep, j))
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
