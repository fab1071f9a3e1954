"""Input validation for the factorize path (mirror of the reference's noLZSS.utils,
reference: src/noLZSS/utils.py:16-58; only what the hot path needs)."""
import struct
from pathlib import Path
from typing import Any, Dict, List, Tuple, Union


class NoLZSSError(Exception):
    """Base exception of the package (reference: utils.py:16-18)."""


class InvalidInputError(NoLZSSError):
    """Input data is invalid for factorization (reference: utils.py:21-23)."""


def validate_input(data: Union[str, bytes]) -> bytes:
    """str -> ASCII bytes; reject non-ASCII, empty input and NUL bytes anywhere but the last
    position; TypeError for anything that is neither str nor bytes (reference: utils.py:26-58)."""
    if isinstance(data, str):
        try:
            data = data.encode("ascii")
        except UnicodeEncodeError as e:
            raise InvalidInputError(f"Input string must contain only ASCII characters (1 byte each): {e}")
    elif not isinstance(data, bytes):
        raise TypeError(f"Input must be str or bytes, got {type(data)}")
    if len(data) == 0:
        raise InvalidInputError("Input data cannot be empty")
    if b"\x00" in data[:-1]:
        raise InvalidInputError("Input data contains null bytes")
    return data


# ---- v2 binary factor files (reference: utils.py:106-357; format factorizer.hpp:64-77) --------
_FOOTER = struct.Struct("<8sQQQQQ")  # magic, num_factors, num_sequences, num_sentinels, footer_size, total_length


def _read_footer(f) -> Tuple[int, int, int, int, int]:
    # (a file shorter than the footer fails in the seek, as in the reference: the caller reports
    # "Error reading file ...: [Errno 22] Invalid argument" -- pinned by tests/golden/python_ref_utils.json)
    f.seek(-_FOOTER.size, 2)
    raw = f.read(_FOOTER.size)
    if len(raw) != _FOOTER.size:
        raise NoLZSSError("File too small to contain valid footer")
    magic, nf, nseq, nsent, fsize, total = _FOOTER.unpack(raw)
    if magic != b"noLZSSv2":
        raise NoLZSSError("Invalid file format: missing noLZSS magic footer (expected v2 format)")
    return nf, nseq, nsent, fsize, total


def read_factors_binary_file(filepath: Union[str, Path]) -> List[Tuple[int, int, int]]:
    """[(start, length, ref)] from a v2 factor file (reference: utils.py:106-155)."""
    filepath = Path(filepath)
    if not filepath.exists():
        raise NoLZSSError(f"File not found: {filepath}")
    try:
        with open(filepath, "rb") as f:
            nf = _read_footer(f)[0]
            f.seek(0)
            data = f.read(24 * nf)
    except OSError as e:
        raise NoLZSSError(f"Error reading file {filepath}: {e}")
    if len(data) != 24 * nf:
        raise NoLZSSError(f"Insufficient data for factor {len(data) // 24}")
    return [struct.unpack_from("<QQQ", data, 24 * i) for i in range(nf)]


def read_binary_file_metadata(filepath: Union[str, Path]) -> Dict[str, Any]:
    """Footer metadata only: names, sentinel factor indices, counts (reference: utils.py:158-251)."""
    filepath = Path(filepath)
    if not filepath.exists():
        raise NoLZSSError(f"File not found: {filepath}")
    try:
        with open(filepath, "rb") as f:
            nf, nseq, nsent, fsize, total = _read_footer(f)
            f.seek(-fsize, 2)
            full = f.read(fsize)
    except OSError as e:
        raise NoLZSSError(f"Error reading file {filepath}: {e}")
    if len(full) != fsize:
        raise NoLZSSError(f"Could not read full footer: expected {fsize}, got {len(full)}")
    meta = full[:fsize - _FOOTER.size]
    names, off = [], 0
    for _ in range(nseq):
        end = meta.find(b"\x00", off)
        if end < 0:
            raise NoLZSSError("Invalid sequence name format")
        names.append(meta[off:end].decode("utf-8"))
        off = end + 1
    if off + 8 * nsent > len(meta):
        raise NoLZSSError("Insufficient data for sentinel indices")
    sentinels = list(struct.unpack_from(f"<{nsent}Q", meta, off)) if nsent else []
    return {"sentinel_factor_indices": sentinels, "sequence_names": names, "num_sequences": nseq,
            "num_sentinels": nsent, "num_factors": nf, "total_length": total}


def read_factors_binary_file_with_metadata(filepath: Union[str, Path]) -> Dict[str, Any]:
    """Factors as (start, length, ref & ~RC_MASK, is_rc) plus the metadata (reference: utils.py:254-357)."""
    meta = read_binary_file_metadata(filepath)
    rc_mask = 1 << 63
    factors = [(s, l, r & (rc_mask - 1), bool(r & rc_mask)) for s, l, r in read_factors_binary_file(filepath)]
    # (the reference's dictionary has no 'num_factors' here, unlike read_binary_file_metadata: utils.py:350-357;
    # pinned by tests/golden/python_ref_utils.json)
    return {"factors": factors, "sentinel_factor_indices": meta["sentinel_factor_indices"],
            "sequence_names": meta["sequence_names"], "num_sequences": meta["num_sequences"],
            "num_sentinels": meta["num_sentinels"], "total_length": meta["total_length"]}
