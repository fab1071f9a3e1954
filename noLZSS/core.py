"""noLZSS.core over the compiled module (reference: src/noLZSS/core.py:25-189): input validation in
Python, compute in `noLZSS._noLZSS` (pybind11 -> C ABI -> HIP)."""
from pathlib import Path
from typing import List, Tuple, Union

from ._noLZSS import (
    factorize as _factorize,
    factorize_file as _factorize_file,
    count_factors as _count_factors,
    count_factors_file as _count_factors_file,
    write_factors_binary_file as _write_factors_binary_file,
    factorize_w_reference as _factorize_w_reference,
    factorize_w_reference_file as _factorize_w_reference_file,
)
from .utils import validate_input

__all__ = ["factorize", "factorize_file", "count_factors", "count_factors_file", "write_factors_binary_file",
           "factorize_w_reference", "factorize_w_reference_file"]

Factors = List[Tuple[int, int, int]]


def _existing(filepath: Union[str, Path]) -> Path:
    p = Path(filepath)
    if not p.exists():  # before the extension is touched (reference: core.py:61-63)
        raise FileNotFoundError(f"File not found: {p}")
    return p


def factorize(data: Union[str, bytes], validate: bool = True) -> Factors:
    return _factorize(validate_input(data) if validate else data)


def factorize_file(filepath: Union[str, Path], reserve_hint: int = 0) -> Factors:
    return _factorize_file(str(_existing(filepath)), reserve_hint)


def count_factors(data: Union[str, bytes], validate: bool = True) -> int:
    return _count_factors(validate_input(data) if validate else data)


def count_factors_file(filepath: Union[str, Path], validate: bool = True) -> int:
    return _count_factors_file(str(_existing(filepath)))


def write_factors_binary_file(data: Union[str, bytes], output_filepath: Union[str, Path], validate: bool = True) -> None:
    """Kept from the reference (core.py:110-132 vs bindings.cpp:180-187): the validated `data` is handed
    to a parameter of the native function that is an input PATH."""
    if validate:
        data = validate_input(data)
    out = Path(output_filepath)
    out.parent.mkdir(parents=True, exist_ok=True)
    _write_factors_binary_file(data, str(out))


def factorize_w_reference(reference_seq: Union[str, bytes], target_seq: Union[str, bytes], validate: bool = True) -> Factors:
    if validate:
        reference_seq, target_seq = validate_input(reference_seq), validate_input(target_seq)
    return _factorize_w_reference(reference_seq, target_seq)


def factorize_w_reference_file(reference_seq: Union[str, bytes], target_seq: Union[str, bytes],
                               output_path: Union[str, Path], validate: bool = True) -> int:
    if validate:
        reference_seq, target_seq = validate_input(reference_seq), validate_input(target_seq)
    out = Path(output_path)
    out.parent.mkdir(parents=True, exist_ok=True)
    return _factorize_w_reference_file(reference_seq, target_seq, str(out))
