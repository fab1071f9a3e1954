"""nolzss_amd.utils against what the reference's own utils.py did on the same inputs (CPU suite).

tests/golden/python_ref_utils.json was produced in the build container by tests/golden/make_python_ref_fixtures.py,
which imports /root/reference/src/noLZSS/utils.py (pure standard library) by path: the outcomes of its
`validate_input` (utils.py:26-58) and of its three v2 readers (utils.py:106, 158, 250) on files written by this repo's
host-only writer `nolzss_write_factor_file`.  Here the same inputs go through `nolzss_amd.utils`, and the writer is
checked to still produce the recorded bytes -- so the files the product writes are files the reference reads, with
the results recorded there.  No GPU, no reference at run time.
"""
import ctypes as C
import json
import struct
from pathlib import Path

import pytest

from nolzss_amd import utils

HERE = Path(__file__).resolve().parent
FX = json.loads((HERE / "golden" / "python_ref_utils.json").read_text())


def make_input(kind, v):
    if kind == "str":
        return v
    if kind == "bytes":
        return bytes.fromhex(v)
    if v.startswith("bytearray:"):
        return bytearray(bytes.fromhex(v.split(":")[1]))
    if v.startswith("memoryview:"):
        return memoryview(bytes.fromhex(v.split(":")[1]))
    if v.startswith("int:"):
        return int(v.split(":")[1])
    if v == "none":
        return None
    return ["A", "C"]


def jsonable(v):
    if isinstance(v, bytes):
        return {"bytes_hex": v.hex()}
    if isinstance(v, (tuple, list)):
        return [jsonable(x) for x in v]
    if isinstance(v, dict):
        return {k: jsonable(x) for k, x in v.items()}
    return v


def outcome(fn, *args, path=None):
    try:
        v = fn(*args)
    except Exception as e:  # noqa: BLE001
        msg = str(e)
        if path is not None:
            msg = msg.replace(str(path), "<PATH>")
        return {"exc": type(e).__name__, "msg": msg}
    return {"ok": jsonable(v)}


@pytest.mark.parametrize("case", FX["validate_input"], ids=lambda c: f"{c['kind']}:{c['value'][:12]!r}")
def test_validate_input_as_the_reference(case):
    got = outcome(utils.validate_input, make_input(case["kind"], case["value"]))
    assert got == case["outcome"]


@pytest.mark.parametrize("case", FX["files"], ids=lambda c: c["name"])
def test_v2_readers_as_the_reference(case, tmp_path):
    path = tmp_path / (case["name"] + ".bin")
    if case["file_hex"] is not None:
        path.write_bytes(bytes.fromhex(case["file_hex"]))
    for reader, expected in case["readers"].items():
        got = outcome(getattr(utils, reader), str(path), path=str(path))
        assert got == expected, reader


class Factor(C.Structure):
    _fields_ = [("start", C.c_uint64), ("length", C.c_uint64), ("ref", C.c_uint64)]


@pytest.mark.parametrize("case", [c for c in FX["files"] if c.get("written_by") == "nolzss_write_factor_file"],
                         ids=lambda c: c["name"])
def test_writer_still_produces_the_bytes_the_reference_read(case, tmp_path):
    """nolzss_write_factor_file is host-only file I/O: it runs without a GPU."""
    lib = C.CDLL(str(HERE.parent / "nolzss_amd" / "libnolzss_hip.so"))
    f = lib.nolzss_write_factor_file
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t]
    factors, names, sentinels = case["factors"], case["names"], case["sentinels"]
    arr = (Factor * max(1, len(factors)))(*[Factor(*x) for x in factors])
    if names is None:
        extra, nseq = b"", 0
    elif isinstance(names, str):
        extra, nseq = b"", int(names.split(":")[1])
    else:
        extra = b"".join(n.encode("utf-8") + b"\0" for n in names) + b"".join(struct.pack("<Q", s) for s in sentinels)
        nseq = len(names)
    path = tmp_path / "out.bin"
    rc = f(str(path).encode(), C.cast(arr, C.c_void_p) if factors else None, len(factors), nseq, len(sentinels),
           case["total_length"], extra if extra else None, len(extra))
    assert rc == 0
    assert path.read_bytes().hex() == case["file_hex"]
