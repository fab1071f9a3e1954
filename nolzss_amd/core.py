"""User-facing factorize API (mirror of the reference's noLZSS.core for the hot path,
reference: src/noLZSS/core.py:25-107): validate, then call the native module."""
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


def factorize(data: Union[str, bytes], validate: bool = True) -> List[Tuple[int, int, int]]:
    """reference: core.py:25-43"""
    if validate:
        data = validate_input(data)
    return _factorize(data)


def factorize_file(filepath: Union[str, Path], reserve_hint: int = 0) -> List[Tuple[int, int, int]]:
    """reference: core.py:46-65 (FileNotFoundError before the extension is touched)"""
    filepath = Path(filepath)
    if not filepath.exists():
        raise FileNotFoundError(f"File not found: {filepath}")
    return _factorize_file(str(filepath), reserve_hint)


def count_factors(data: Union[str, bytes], validate: bool = True) -> int:
    """reference: core.py:68-86"""
    if validate:
        data = validate_input(data)
    return _count_factors(data)


def count_factors_file(filepath: Union[str, Path], validate: bool = True) -> int:
    """reference: core.py:89-107"""
    filepath = Path(filepath)
    if not filepath.exists():
        raise FileNotFoundError(f"File not found: {filepath}")
    return _count_factors_file(str(filepath))


def write_factors_binary_file(data: Union[str, bytes], output_filepath: Union[str, Path]) -> None:
    """reference: core.py:110-132.  Kept bug-for-bug: the reference hands the validated *data*
    to a native parameter that is an input FILE PATH (bindings.cpp:180-187), so `data` must name
    a file (README.md:65 passes paths)."""
    data = validate_input(data)
    output_filepath = Path(output_filepath)
    output_filepath.parent.mkdir(parents=True, exist_ok=True)
    _write_factors_binary_file(data, str(output_filepath))


def factorize_w_reference(reference_seq: Union[str, bytes], target_seq: Union[str, bytes],
                          validate: bool = True) -> List[Tuple[int, int, int]]:
    """reference: core.py:164-207 -- target factorized against reference + '\\x01' + target;
    start positions are absolute in the combined string."""
    if validate:
        reference_seq = validate_input(reference_seq)
        target_seq = validate_input(target_seq)
    if isinstance(reference_seq, bytes):
        reference_seq = reference_seq.decode("ascii")
    if isinstance(target_seq, bytes):
        target_seq = target_seq.decode("ascii")
    return _factorize_w_reference(reference_seq, target_seq)


def factorize_w_reference_file(reference_seq: Union[str, bytes], target_seq: Union[str, bytes],
                               output_path: Union[str, Path], validate: bool = True) -> int:
    """reference: core.py:210-257"""
    if validate:
        reference_seq = validate_input(reference_seq)
        target_seq = validate_input(target_seq)
    if isinstance(reference_seq, bytes):
        reference_seq = reference_seq.decode("ascii")
    if isinstance(target_seq, bytes):
        target_seq = target_seq.decode("ascii")
    output_path = Path(output_path)
    output_path.parent.mkdir(parents=True, exist_ok=True)
    return _factorize_w_reference_file(reference_seq, target_seq, str(output_path))
