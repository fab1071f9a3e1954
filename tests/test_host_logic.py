"""CPU tests (no GPU): the C-ABI library loads and exports every declared symbol, the host-side
mirror of the reference interface behaves like the reference (validation, FASTA parsing, error
types), and the product path fails loudly without a device."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    from nolzss_amd import _lib
    header = (ROOT / "include" / "nolzss_hip.h").read_text()
    declared = set(re.findall(r"\b(nolzss_[a-z_0-9]+)\s*\(", header))
    declared -= {"nolzss_factor"}
    assert declared, "no declarations parsed"
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/nolzss_hip.h but not exported"
    assert declared == set(_lib.EXPORTED_SYMBOLS)


def test_version_and_error_string():
    from nolzss_amd import _noLZSS
    assert _noLZSS.__version__.startswith("0.")
    assert isinstance(_noLZSS.lib.nolzss_last_error(), bytes)


def test_reference_module_names_exist():
    """names the reference imports from _noLZSS at package import time"""
    from nolzss_amd import _noLZSS
    for name in ["factorize", "factorize_file", "count_factors", "count_factors_file",
                 "write_factors_binary_file", "factorize_w_reference", "factorize_w_reference_file",
                 "factorize_dna_w_rc", "factorize_file_dna_w_rc", "count_factors_dna_w_rc",
                 "count_factors_file_dna_w_rc", "write_factors_binary_file_dna_w_rc",
                 "factorize_multiple_dna_w_rc", "factorize_file_multiple_dna_w_rc",
                 "count_factors_multiple_dna_w_rc", "count_factors_file_multiple_dna_w_rc",
                 "write_factors_binary_file_multiple_dna_w_rc", "factorize_fasta_multiple_dna_w_rc",
                 "prepare_multiple_dna_sequences_w_rc", "Factor", "__version__"]:
        assert hasattr(_noLZSS, name), name
    with pytest.raises(RuntimeError, match="Cannot open FASTA file"):      # fasta_processor.cpp:33-35
        _noLZSS.factorize_fasta_multiple_dna_w_rc("/nonexistent/a.fasta")
    with pytest.raises(ValueError, match="Invalid sanitize_mode"):          # bindings.cpp:36
        _noLZSS.factorize_fasta_multiple_dna_w_rc("a.fasta", "lenient")


def test_validate_input_mirrors_reference():
    """reference: tests/test_utils.py:26-73, src/noLZSS/utils.py:26-58"""
    from nolzss_amd import InvalidInputError, validate_input
    assert validate_input("hello") == b"hello"
    assert validate_input(b"hello") == b"hello"
    assert validate_input(b"abc\x00") == b"abc\x00"          # NUL allowed only as last byte
    with pytest.raises(InvalidInputError):
        validate_input("")
    with pytest.raises(InvalidInputError):
        validate_input(b"")
    with pytest.raises(InvalidInputError):
        validate_input(b"a\x00b")
    with pytest.raises(InvalidInputError):
        validate_input("héllo")
    with pytest.raises(TypeError):
        validate_input(123)
    with pytest.raises(TypeError):
        validate_input(bytearray(b"abc"))


def test_core_wrappers_validate_before_native(tmp_path):
    """reference: src/noLZSS/core.py:25-107 -- errors raised before the extension is touched"""
    import nolzss_amd
    with pytest.raises(nolzss_amd.InvalidInputError):
        nolzss_amd.factorize("")
    with pytest.raises(TypeError):
        nolzss_amd.count_factors(12)
    with pytest.raises(FileNotFoundError):
        nolzss_amd.factorize_file(tmp_path / "missing.txt")
    with pytest.raises(FileNotFoundError):
        nolzss_amd.count_factors_file(tmp_path / "missing.txt")


def test_buffer_contract():
    """bindings.cpp:59-64: 1-D, itemsize 1, else ValueError"""
    import numpy as np
    from nolzss_amd import _noLZSS
    with pytest.raises(ValueError):
        _noLZSS._as_buffer(np.zeros(4, dtype=np.uint16))
    with pytest.raises(ValueError):
        _noLZSS._as_buffer(np.zeros((2, 2), dtype=np.uint8))
    with pytest.raises(TypeError):
        _noLZSS._as_buffer(3.5)
    p, n, keep = _noLZSS._as_buffer(bytearray(b"abc"))
    assert n == 3


def test_no_cpu_fallback_without_device():
    """The product path must fail loudly when no GPU is usable (never route through a CPU path)."""
    import nolzss_amd
    from nolzss_amd import _noLZSS
    if _noLZSS.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        nolzss_amd.factorize(b"abracadabra")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        nolzss_amd.count_factors(b"abracadabra")
    from nolzss_amd.genomics import factorize_dna_w_rc
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        factorize_dna_w_rc(b"ACGT")


def test_product_never_imports_oracle():
    for path in (ROOT / "nolzss_amd").rglob("*"):
        if path.suffix in {".py", ".hip", ".hpp", ".h"} or path.name == "Makefile":
            text = path.read_text(errors="ignore")
            assert "oracle" not in text.lower(), f"{path} mentions the oracle"


def test_prepare_w_rc_host_side_matches_oracle():
    """prepare_* is host-side O(n) code in the C ABI (no GPU needed)."""
    import oracle_lib as oracle
    from nolzss_amd import _noLZSS
    for seqs in (["ACGT"], ["acgt", "TTGA", "C"], ["A"] * 125, ["AC", "", "GT"]):
        assert _noLZSS.prepare_multiple_dna_sequences_w_rc_bytes(seqs) == oracle.prepare_multiple_dna_w_rc(seqs)
    s, orig, sent = _noLZSS.prepare_multiple_dna_sequences_w_rc(["ACGT", "GG"])
    assert isinstance(s, str) and orig == 8 and sent == [4, 7, 10, 15]
    with pytest.raises(ValueError):                       # std::invalid_argument, factorizer.cpp:81-83
        _noLZSS.prepare_multiple_dna_sequences_w_rc(["A"] * 126)
    with pytest.raises(RuntimeError):                     # std::runtime_error, :86-95
        _noLZSS.prepare_multiple_dna_sequences_w_rc(["ACGX"])
    with pytest.raises(RuntimeError):                     # :76-78
        _noLZSS.prepare_multiple_dna_sequences_w_rc(["", ""])
    assert _noLZSS.prepare_multiple_dna_sequences_w_rc([]) == ("", 0, [])
    with pytest.raises(UnicodeDecodeError):               # sentinel bytes >= 128 cannot become str
        _noLZSS.prepare_multiple_dna_sequences_w_rc(["A"] * 70)


def test_prepare_w_rc_long_sequences_on_several_threads():
    """Sequences of tens of megabytes are validated, case-folded and reverse-complemented in pieces on several host
    threads (host_util.hpp: host_parallel): same bytes as numpy, and an invalid byte is still reported at its FIRST index."""
    import numpy as np
    from nolzss_amd import _noLZSS
    rng = np.random.default_rng(7)
    a = np.frombuffer(b"ACGTacgt", dtype=np.uint8)[rng.integers(0, 8, size=23_000_001)]
    b = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=9_500_003)]
    S, orig, sent = _noLZSS.prepare_multiple_dna_sequences_w_rc_bytes([a.tobytes().decode(), b.tobytes().decode()])
    up = lambda x: x & np.uint8(0xDF)
    comp = np.zeros(256, dtype=np.uint8)
    comp[list(b"ACGT")] = list(b"TGCA")
    rc = lambda x: comp[up(x)][::-1]
    exp = np.concatenate([up(a), [1], up(b), [2], rc(b), [3], rc(a), [4]]).astype(np.uint8)
    got = np.frombuffer(S, dtype=np.uint8)
    assert orig == len(a) + len(b) + 2 and len(got) == len(exp)
    assert np.array_equal(got, exp)
    bad = bytearray(a.tobytes())
    bad[20_000_000] = ord("N")
    bad[11_111_111] = ord("x")
    with pytest.raises(RuntimeError, match="Invalid nucleotide 'x'"):
        _noLZSS.prepare_multiple_dna_sequences_w_rc_bytes([bytes(bad).decode()])


def test_fasta_parsing_mirrors_reference(tmp_path):
    """reference: tests/test_genomics.py:96-144, src/noLZSS/genomics/fasta.py:28-76"""
    from nolzss_amd.genomics.fasta import FASTAError, _parse_fasta_content, _load_validated
    d = _parse_fasta_content(">seq1 some description\nacgt\nAC GT\n\n>seq2\nTTTT\n")
    assert d == {"seq1": "ACGTACGT", "seq2": "TTTT"}
    assert list(_parse_fasta_content(">a\nAC\n>b\nGG\n>a\nTT\n").items()) == [("a", "TT"), ("b", "GG")]
    with pytest.raises(FASTAError):
        _parse_fasta_content(">\nACGT\n")
    with pytest.raises(FASTAError):
        _parse_fasta_content("ACGT\n>x\nAC\n")
    with pytest.raises(FASTAError):
        _parse_fasta_content("\n\n")
    p = tmp_path / "bad.fa"
    p.write_text(">x\nACGTN\n")
    with pytest.raises(FASTAError):
        _load_validated(p)
    with pytest.raises(FileNotFoundError):
        _load_validated(tmp_path / "nope.fa")
    p2 = tmp_path / "ok.fa"
    p2.write_text(">x\nacgt\n>y z\nGG\n")
    assert _load_validated(p2) == [("x", b"ACGT"), ("y", b"GG")]


def test_lpt_assignment_is_deterministic_and_balanced():
    from nolzss_amd.genomics.fasta import lpt_assignment
    lens = [10, 9, 8, 7, 6, 5, 4, 3, 2, 1]
    own = lpt_assignment(lens, 3)
    assert own == lpt_assignment(lens, 3)
    loads = [sum(l for l, o in zip(lens, own) if o == b) for b in range(3)]
    assert max(loads) - min(loads) <= max(lens)
    assert lpt_assignment([5, 5, 5, 5], 4) == [0, 1, 2, 3]
    assert lpt_assignment([], 2) == []


@pytest.mark.parametrize("n_dev", [1, 2, 4, 8])
@pytest.mark.parametrize("with_rc", [False, True])
def test_batch_plan_deals_every_record_once(n_dev, with_rc):
    """The in-process multi-device path of nolzss_factorize_batch (batch.hip: plan_batch, lpt_plan_singles) has only ever
    RUN with one device; its plan is host logic and is checked here for 2, 4 and 8 devices: every non-empty record is
    dealt exactly once -- to one merged run (taken from one work queue by lane w on device w % n_dev) or to one device
    for a run of its own --, the runs hold consecutive records, and the single records are balanced within the
    longest-processing-time bound (no device carries more than the lightest one plus one record)."""
    import random
    from nolzss_amd import _noLZSS
    rng = random.Random(1000 * n_dev + with_rc)
    shapes = [
        [1 << 22] * 512,                                                   # BASELINE config 4
        [rng.randint(1, 5000) for _ in range(3000)],                       # short records only
        [0, 5, 0, 1 << 28, 7, 1 << 29, 0, 3, 1 << 27, 1 << 21, 1 << 20],   # empty, tiny and very long ones
        [rng.choice([0, 1, 100, 10_000, 1 << 20, 1 << 22, 1 << 24, 1 << 27, 3 << 27]) for _ in range(400)],
        [1 << 28] * 17, [3], [], [0, 0],
    ]
    for lens in shapes:
        chunk_of, device_of, n_chunks = _noLZSS.debug_batch_plan(lens, n_dev, with_rc)
        members = {}
        for j, L in enumerate(lens):
            placed = (chunk_of[j] >= 0) + (device_of[j] >= 0)
            assert placed == (1 if L > 0 else 0), (j, L, chunk_of[j], device_of[j])
            if chunk_of[j] >= 0:
                members.setdefault(chunk_of[j], []).append(j)
            assert device_of[j] < n_dev
        assert sorted(members) == list(range(n_chunks))
        for k, js in members.items():
            assert len(js) >= 2                                            # a run of one record is a single record
            between = [j for j in range(js[0], js[-1] + 1) if j not in js]
            # what lies between the members of a run belongs elsewhere: empty records, or records of the other size class
            assert all(chunk_of[j] != k for j in between)
        load = [sum(L for j, L in enumerate(lens) if device_of[j] == d) for d in range(n_dev)]
        singles = [L for j, L in enumerate(lens) if device_of[j] >= 0]
        if singles:
            assert max(load) - min(load) <= max(singles), (load, max(singles))


def test_v2_reader_against_hand_packed_files(tmp_path):
    """reference: tests/test_utils.py:177-290 -- footers packed by hand"""
    import struct
    import nolzss_amd as pkg
    factors = [(0, 1, 0), (1, 3, 0), (4, 2, (1 << 63) | 1)]
    body = b"".join(struct.pack("<QQQ", *f) for f in factors)
    names = b"chr1\x00chr2\x00"
    sent = struct.pack("<Q", 1)
    footer = struct.pack("<8sQQQQQ", b"noLZSSv2", 3, 2, 1, 48 + len(names) + len(sent), 6)
    p = tmp_path / "f.bin"
    p.write_bytes(body + names + sent + footer)
    assert pkg.read_factors_binary_file(p) == factors
    meta = pkg.read_factors_binary_file_with_metadata(p)
    assert meta["sequence_names"] == ["chr1", "chr2"] and meta["sentinel_factor_indices"] == [1]
    # (the reference's dictionary of this reader has no 'num_factors': tests/golden/python_ref_utils.json)
    assert "num_factors" not in meta and meta["total_length"] == 6
    assert pkg.read_binary_file_metadata(p)["num_factors"] == 3
    assert meta["factors"] == [(0, 1, 0, False), (1, 3, 0, False), (4, 2, 1, True)]
    bad = tmp_path / "bad.bin"
    bad.write_bytes(body + struct.pack("<8sQQQQQ", b"noLZSSv1", 3, 0, 0, 48, 6))
    with pytest.raises(pkg.NoLZSSError):
        pkg.read_factors_binary_file(bad)
    with pytest.raises(pkg.NoLZSSError):
        pkg.read_factors_binary_file(tmp_path / "missing.bin")
    tiny = tmp_path / "tiny.bin"
    tiny.write_bytes(b"abc")
    with pytest.raises(pkg.NoLZSSError):
        pkg.read_factors_binary_file(tiny)


def test_prepare_no_rc_host_side():
    """prepare_multiple_dna_sequences_no_rc, factorizer.cpp:199-294: sentinels only BETWEEN sequences"""
    from nolzss_amd import _noLZSS
    assert _noLZSS.prepare_multiple_dna_sequences_no_rc(["ACGT", "gg", "T"]) == ("ACGT\x01GG\x02T", 9, [4, 7])
    assert _noLZSS.prepare_multiple_dna_sequences_no_rc(["ACGT"]) == ("ACGT", 4, [])
    assert _noLZSS.prepare_multiple_dna_sequences_no_rc([]) == ("", 0, [])
    s, n, sent = _noLZSS.prepare_multiple_dna_sequences_no_rc_bytes(["A"] * 250)
    assert len(s) == 499 and len(sent) == 249 and len(set(s[1::2])) == 249
    with pytest.raises(ValueError):
        _noLZSS.prepare_multiple_dna_sequences_no_rc(["A"] * 251)
    with pytest.raises(RuntimeError):
        _noLZSS.prepare_multiple_dna_sequences_no_rc(["ACGU"])


def test_every_function_of_the_reference_module_exists():
    """the 42 m.def names of the reference's bindings.cpp, listed here as data"""
    from nolzss_amd import _noLZSS
    names = """factorize factorize_file count_factors count_factors_file write_factors_binary_file
    factorize_dna_w_rc factorize_file_dna_w_rc count_factors_dna_w_rc count_factors_file_dna_w_rc
    write_factors_binary_file_dna_w_rc factorize_multiple_dna_w_rc factorize_file_multiple_dna_w_rc
    count_factors_multiple_dna_w_rc count_factors_file_multiple_dna_w_rc
    write_factors_binary_file_multiple_dna_w_rc factorize_fasta_multiple_dna_w_rc
    factorize_dna_rc_w_ref_fasta_files factorize_fasta_multiple_dna_no_rc
    write_factors_binary_file_fasta_multiple_dna_w_rc write_factors_binary_file_fasta_multiple_dna_no_rc
    prepare_multiple_dna_sequences_w_rc prepare_multiple_dna_sequences_no_rc factorize_dna_w_reference_seq
    factorize_dna_w_reference_seq_file factorize_w_reference factorize_w_reference_file
    write_factors_dna_w_reference_fasta_files_to_binary parallel_factorize_to_file
    parallel_factorize_file_to_file parallel_factorize_dna_w_rc_to_file
    parallel_factorize_file_dna_w_rc_to_file parallel_write_factors_binary_file_fasta_multiple_dna_w_rc
    parallel_write_factors_binary_file_fasta_multiple_dna_no_rc
    parallel_write_factors_dna_w_reference_fasta_files_to_binary factorize_fasta_dna_w_rc_per_sequence
    factorize_fasta_dna_no_rc_per_sequence write_factors_binary_file_fasta_dna_w_rc_per_sequence
    write_factors_binary_file_fasta_dna_no_rc_per_sequence count_factors_fasta_dna_w_rc_per_sequence
    count_factors_fasta_dna_no_rc_per_sequence
    parallel_write_factors_binary_file_fasta_dna_w_rc_per_sequence
    parallel_write_factors_binary_file_fasta_dna_no_rc_per_sequence""".split()
    assert len(names) == 42
    for n in names:
        assert callable(getattr(_noLZSS, n, None)), n
    for cls in ("Factor", "FastaFactorizationResult", "FastaPerSequenceFactorizationResult"):
        assert isinstance(getattr(_noLZSS, cls), type)
    f = _noLZSS.Factor(5, 3, (1 << 63) | 2)
    assert (f.start, f.length, f.ref, f.is_rc) == (5, 3, 2, True)
