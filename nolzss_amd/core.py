"""User-facing factorize API (mirror of the reference's noLZSS.core for the hot path,
reference: src/noLZSS/core.py:25-107): validate, then call the native module."""
from pathlib import Path
from typing import List, Tuple, Union

from ._noLZSS import (
    factorize as _factorize,
    factorize_file as _factorize_file,
    count_factors as _count_factors,
    count_factors_file as _count_factors_file,
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
