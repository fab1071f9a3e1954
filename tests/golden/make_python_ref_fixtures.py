#!/usr/bin/env python3
"""Pins the host-side helpers of the path to the part of the reference that DOES run in the build container.

/root/reference/src/noLZSS/utils.py is pure standard library (the compiled `_noLZSS` module is not needed for it), so
this script -- build container only, never on the GPU box, which has no /root/reference -- imports that one file by
path and records, as DATA, in tests/golden/python_ref_utils.json:

  (i)  the outcome of the reference's `validate_input` (utils.py:26-58) on a table of inputs: the returned bytes, or
       the exception class and message;
  (ii) for v2 factor files written by THIS repo's host-only writer `nolzss_write_factor_file` (no GPU involved:
       plain file I/O in libnolzss_hip.so) -- with and without sequence names / sentinel indices, empty, and damaged
       in the ways the readers check for -- what the reference's three readers return (utils.py:106, 158, 250):
       `read_factors_binary_file`, `read_binary_file_metadata`, `read_factors_binary_file_with_metadata`.
       The file bytes themselves are kept in the fixture (hex), so the CPU test replays them without the reference.

tests/test_python_ref_fixtures.py (CPU suite) then checks `nolzss_amd.utils` against these outcomes and that the
writer still produces the recorded bytes.  Nothing of the reference's text is copied: inputs and observed outputs only.

    python tests/golden/make_python_ref_fixtures.py      (rewrites tests/golden/python_ref_utils.json)
"""
import ctypes as C
import importlib.util
import json
import os
import struct
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
REF_UTILS = Path("/root/reference/src/noLZSS/utils.py")
OUT = Path(__file__).resolve().parent / "python_ref_utils.json"


def load_reference_utils():
    spec = importlib.util.spec_from_file_location("_reference_noLZSS_utils", REF_UTILS)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def outcome(fn, *args, path=None):
    """{"ok": value} or {"exc": class name, "msg": message}; a temporary file's path is replaced by <PATH>."""
    try:
        v = fn(*args)
    except Exception as e:  # noqa: BLE001  (the class is what is recorded)
        msg = str(e)
        if path is not None:
            msg = msg.replace(str(path), "<PATH>")
        return {"exc": type(e).__name__, "msg": msg}
    return {"ok": jsonable(v)}


def jsonable(v):
    if isinstance(v, bytes):
        return {"bytes_hex": v.hex()}
    if isinstance(v, tuple):
        return [jsonable(x) for x in v]
    if isinstance(v, list):
        return [jsonable(x) for x in v]
    if isinstance(v, dict):
        return {k: jsonable(x) for k, x in v.items()}
    return v


# ---- (i) validate_input -------------------------------------------------------------------------------------
VALIDATE_INPUTS = [
    ("str", "ACGT"), ("str", "abracadabra"), ("str", ""), ("str", "café"), ("str", "A\x00C"), ("str", "AC\x00"),
    ("str", "\x00"), ("str", " \t\n"), ("str", "中"), ("str", "\x7f~"),
    ("bytes", b"ACGT"), ("bytes", b""), ("bytes", b"A\x00C"), ("bytes", b"AC\x00"), ("bytes", b"\x00"),
    ("bytes", b"\x00\x00"), ("bytes", bytes(range(1, 256))), ("bytes", b"\xff\xfe"), ("bytes", b"$"),
    ("other", "bytearray:4143"), ("other", "int:5"), ("other", "none"), ("other", "list"), ("other", "memoryview:4143"),
]


def make_input(kind, v):
    if kind != "other":
        return v
    if v.startswith("bytearray:"):
        return bytearray(bytes.fromhex(v.split(":")[1]))
    if v.startswith("memoryview:"):
        return memoryview(bytes.fromhex(v.split(":")[1]))
    if v.startswith("int:"):
        return int(v.split(":")[1])
    if v == "none":
        return None
    return ["A", "C"]


# ---- (ii) v2 factor files -----------------------------------------------------------------------------------
class Factor(C.Structure):
    _fields_ = [("start", C.c_uint64), ("length", C.c_uint64), ("ref", C.c_uint64)]


def writer():
    lib = C.CDLL(str(ROOT / "nolzss_amd" / "libnolzss_hip.so"))
    f = lib.nolzss_write_factor_file
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t]
    return f


RC = 1 << 63
FILE_CASES = [
    # name, factors, names (None: no metadata), sentinel indices, total_length
    ("plain_three_factors", [(0, 1, 0), (1, 1, 1), (2, 5, 0)], None, [], 7),
    ("no_factors", [], None, [], 0),
    ("one_literal", [(0, 1, 0)], None, [], 1),
    ("rc_flag_and_large_values", [(0, 3, 0), (3, 4, RC | 1), (7, 2 ** 40, 2 ** 62 + 5)], None, [], 7 + 2 ** 40),
    ("two_sequences_one_sentinel", [(0, 4, 0), (4, 1, 4), (5, 3, RC | 0), (8, 1, 8)], ["seq1", "seq2"], [1], 9),
    ("three_sequences_names_with_spaces_and_utf8", [(0, 2, 0), (2, 1, 2), (3, 2, 0), (5, 1, 5), (6, 2, RC | 3)],
     ["chr 1", "café", ""], [1, 3], 8),
    ("names_without_sentinels", [(0, 9, 0)], ["only"], [], 9),
    ("declared_sequences_without_names", [(0, 2, 0), (2, 2, 0)], "declare:2", [], 4),  # the reference+target files: 2 sequences, no names
]
DAMAGE = [
    ("truncated_below_footer", lambda b: b[:40]),
    ("bad_magic", lambda b: b[:-48] + b"noLZSSv1" + b[-40:]),
    ("factor_count_beyond_data", lambda b: b[:-40] + struct.pack("<Q", 1000) + b[-32:]),
    ("footer_size_beyond_file", lambda b: b[:-16] + struct.pack("<Q", 10 ** 6) + b[-8:]),
    ("empty_file", lambda b: b""),
]


def main():
    ref = load_reference_utils()
    fx = {"_how": "tests/golden/make_python_ref_fixtures.py: the reference's src/noLZSS/utils.py imported by path in the build "
                  "container; inputs and observed outputs only", "validate_input": [], "files": []}
    for kind, v in VALIDATE_INPUTS:
        fx["validate_input"].append({"kind": kind, "value": v.hex() if isinstance(v, bytes) else v,
                                     "outcome": outcome(ref.validate_input, make_input(kind, v))})
    write = writer()
    with tempfile.TemporaryDirectory() as td:
        def readers(path):
            return {name: outcome(getattr(ref, name), path, path=path)
                    for name in ("read_factors_binary_file", "read_binary_file_metadata", "read_factors_binary_file_with_metadata")}
        base_bytes = None
        for name, factors, names, sentinels, total in FILE_CASES:
            path = os.path.join(td, name + ".bin")
            arr = (Factor * max(1, len(factors)))(*[Factor(*f) for f in factors])
            if names is None:
                extra, nseq = b"", 0
            elif isinstance(names, str):
                extra, nseq = b"", int(names.split(":")[1])
            else:
                extra = b"".join(n.encode("utf-8") + b"\0" for n in names) + b"".join(struct.pack("<Q", s) for s in sentinels)
                nseq = len(names)
            rc = write(path.encode(), C.cast(arr, C.c_void_p) if factors else None, len(factors), nseq, len(sentinels), total,
                       extra if extra else None, len(extra))
            assert rc == 0, (name, rc)
            data = Path(path).read_bytes()
            if name == "two_sequences_one_sentinel":
                base_bytes = data
            fx["files"].append({"name": name, "factors": [list(f) for f in factors],
                                "names": names, "sentinels": sentinels, "total_length": total,
                                "written_by": "nolzss_write_factor_file", "file_hex": data.hex(), "readers": readers(path)})
        for name, fn in DAMAGE:
            path = os.path.join(td, name + ".bin")
            Path(path).write_bytes(fn(base_bytes))
            fx["files"].append({"name": name, "written_by": "damaged copy of two_sequences_one_sentinel",
                                "file_hex": Path(path).read_bytes().hex(), "readers": readers(path)})
        missing = os.path.join(td, "does_not_exist.bin")
        fx["files"].append({"name": "missing_file", "written_by": None, "file_hex": None, "readers": readers(missing)})
    OUT.write_text(json.dumps(fx, indent=1, ensure_ascii=True) + "\n")
    print(f"wrote {OUT}: {len(fx['validate_input'])} validate_input cases, {len(fx['files'])} files")


if __name__ == "__main__":
    sys.exit(main())
