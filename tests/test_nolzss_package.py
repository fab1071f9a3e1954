"""The reference's import names: `import noLZSS` resolves to a compiled pybind11 module
(`noLZSS._noLZSS`, nolzss_amd/csrc/pybind_shim.cpp) over the C ABI, with core / utils / genomics /
parallel on top (reference: src/cpp/bindings.cpp:39-77, src/noLZSS/__init__.py:9-20,
src/noLZSS/core.py:12-20, src/noLZSS/genomics/__init__.py:8-22).

CPU part: the module is compiled (not Python), every name the reference package imports is there,
argument checks and error types happen before / without a device.  GPU part: the reference's
known-answer vectors through `from noLZSS import factorize`."""
import importlib.machinery
import json
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
KATS = json.loads((ROOT / "tests" / "golden" / "kats.json").read_text())


def test_compiled_module_and_names():
    import noLZSS
    import noLZSS.genomics
    import noLZSS.parallel
    from noLZSS import _noLZSS
    assert any(_noLZSS.__file__.endswith(s) for s in importlib.machinery.EXTENSION_SUFFIXES), _noLZSS.__file__
    assert type(_noLZSS.factorize).__name__ == "builtin_function_or_method"  # bound by pybind11, not Python
    assert noLZSS.__version__ == _noLZSS.__version__
    # what the reference's Python layer imports by name
    for name in ["factorize", "factorize_file", "count_factors", "count_factors_file", "write_factors_binary_file",
                 "factorize_w_reference", "factorize_w_reference_file", "factorize_dna_w_rc", "factorize_file_dna_w_rc",
                 "count_factors_dna_w_rc", "count_factors_file_dna_w_rc", "write_factors_binary_file_dna_w_rc",
                 "factorize_multiple_dna_w_rc", "factorize_file_multiple_dna_w_rc", "count_factors_multiple_dna_w_rc",
                 "count_factors_file_multiple_dna_w_rc", "write_factors_binary_file_multiple_dna_w_rc",
                 "factorize_fasta_multiple_dna_w_rc", "prepare_multiple_dna_sequences_w_rc",
                 "write_factors_binary_file_fasta_multiple_dna_w_rc", "write_factors_binary_file_fasta_multiple_dna_no_rc",
                 "parallel_factorize_to_file", "parallel_factorize_file_to_file", "parallel_factorize_dna_w_rc_to_file",
                 "parallel_factorize_file_dna_w_rc_to_file", "Factor", "__version__"]:
        assert hasattr(_noLZSS, name), name
    for name in ["factorize", "factorize_file", "count_factors", "count_factors_file", "validate_input",
                 "InvalidInputError", "NoLZSSError", "read_factors_binary_file"]:
        assert hasattr(noLZSS, name), name
    for name in ["factorize_dna_w_rc", "prepare_multiple_dna_sequences_w_rc", "read_nucleotide_fasta", "FASTAError"]:
        assert hasattr(noLZSS.genomics, name), name


def test_argument_checks_without_a_device(tmp_path):
    import numpy as np
    import noLZSS
    from noLZSS import _noLZSS
    with pytest.raises(ValueError, match="1-dimensional"):                 # bindings.cpp:62-64
        _noLZSS.factorize(np.zeros((2, 2), dtype=np.uint8))
    with pytest.raises(ValueError, match="itemsize==1"):                   # bindings.cpp:59-61
        _noLZSS.count_factors(np.zeros(4, dtype=np.uint16))
    with pytest.raises(TypeError):
        _noLZSS.factorize(3.5)
    with pytest.raises(noLZSS.InvalidInputError):                          # core.py:25-43 via utils.py:26-58
        noLZSS.factorize("")
    with pytest.raises(noLZSS.InvalidInputError):
        noLZSS.count_factors(b"a\x00b")
    with pytest.raises(TypeError):
        noLZSS.factorize(12)
    with pytest.raises(FileNotFoundError):
        noLZSS.factorize_file(tmp_path / "missing.txt")
    with pytest.raises(RuntimeError, match="Cannot open input file"):      # factorizer.cpp:401-406
        _noLZSS.factorize_file(str(tmp_path / "missing.txt"))
    # host-side preparation needs no device (factorizer.cpp:54-172)
    S, n, sent = noLZSS.genomics.prepare_multiple_dna_sequences_w_rc(["ACGT", "tt"])
    assert (S, n, sent) == ("ACGT\x01TT\x02AA\x03ACGT\x04", 8, [4, 7, 10, 15])
    with pytest.raises(RuntimeError, match="Invalid nucleotide"):
        noLZSS.genomics.prepare_multiple_dna_sequences_w_rc(["ACGN"])
    with pytest.raises(ValueError, match="Too many sequences"):
        noLZSS.genomics.prepare_multiple_dna_sequences_w_rc(["A"] * 126)
    f = _noLZSS.Factor()
    assert (f.start, f.length, f.ref, f.is_rc) == (0, 0, 0, False)


def test_no_cpu_fallback_through_the_compiled_module():
    import noLZSS
    if noLZSS._noLZSS.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        noLZSS.factorize(b"abracadabra")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        noLZSS.genomics.count_factors_dna_w_rc(b"ACGT")


@pytest.mark.gpu
@pytest.mark.parametrize("v", KATS["plain"] + KATS["derived_plain"], ids=lambda v: str(v.get("input", "repeat")))
def test_plain_kats_through_noLZSS(v, tmp_path):
    from noLZSS import factorize, count_factors, factorize_file, count_factors_file
    text = v["input"] if "input" in v else v["input_repeat"][0] * v["input_repeat"][1]
    exp = [tuple(f) for f in v["factors"]]
    assert factorize(text) == exp and factorize(text.encode()) == exp
    assert count_factors(text) == len(exp)
    p = tmp_path / "t.txt"
    p.write_bytes(text.encode())
    assert factorize_file(p) == exp and count_factors_file(str(p)) == len(exp)


@pytest.mark.gpu
@pytest.mark.parametrize("v", KATS["dna_w_rc"] + KATS["derived_dna_w_rc"], ids=lambda v: v["input"])
def test_rc_kats_through_noLZSS(v):
    from noLZSS.genomics import (factorize_dna_w_rc, count_factors_dna_w_rc, factorize_multiple_dna_w_rc,
                                 prepare_multiple_dna_sequences_w_rc)
    exp = [tuple(f) for f in v["factors"]]
    assert factorize_dna_w_rc(v["input"].encode()) == exp
    assert count_factors_dna_w_rc(v["input"].encode()) == len(exp)
    S, _, _ = prepare_multiple_dna_sequences_w_rc([v["input"]])
    assert factorize_multiple_dna_w_rc(S.encode("latin-1")) == exp


@pytest.mark.gpu
def test_noLZSS_agrees_with_the_ctypes_mirror_and_the_oracle(tmp_path):
    import gen
    import oracle_lib as oracle
    import noLZSS
    import nolzss_amd
    text = gen.repeat_dna(300_000, seed=11, lo=16, hi=4096).tobytes()
    got = noLZSS.factorize(text)
    assert got == nolzss_amd.factorize(text) == oracle.factorize(text)
    assert noLZSS.genomics.factorize_dna_w_rc(text[:100_000]) == oracle.factorize_dna_w_rc(text[:100_000])
    ref, tgt = text[:5000].decode(), text[2000:9000].decode()
    assert noLZSS.factorize_w_reference(ref, tgt) == oracle.factorize((ref + "\x01" + tgt).encode(), start_pos=len(ref) + 1)
    out = tmp_path / "f.bin"
    src = tmp_path / "in.txt"
    src.write_bytes(text)
    assert noLZSS._noLZSS.write_factors_binary_file(str(src), str(out)) == len(got)
    assert noLZSS.read_factors_binary_file(out) == got
    fa = tmp_path / "x.fa"
    gen.write_fasta(fa, [("r1", text[:3000]), ("r2", text[1000:5000])])
    res = noLZSS.genomics.read_nucleotide_fasta(fa)
    assert [rid for rid, _ in res] == ["r1", "r2"] and res[1][1] == oracle.factorize(text[1000:5000])
