"""The user-facing factorize API of the hot path, written once for both import names.

`bind(native)` builds the seven functions of the reference's `noLZSS.core` (reference:
src/noLZSS/core.py:25-257 -- names, argument meaning, error behaviour) over a native module:
`nolzss_amd.core` binds them to the ctypes mirror `nolzss_amd._noLZSS`, `noLZSS.core` to the compiled
pybind11 module `noLZSS._noLZSS`.  Both native modules call the same C ABI (include/nolzss_hip.h).
"""
from pathlib import Path
from typing import List, Tuple, Union

from .utils import validate_input

__all__ = ["factorize", "factorize_file", "count_factors", "count_factors_file", "write_factors_binary_file",
           "factorize_w_reference", "factorize_w_reference_file"]

Factors = List[Tuple[int, int, int]]
Text = Union[str, bytes]
PathLike = Union[str, Path]


def _existing(filepath: PathLike) -> str:
    p = Path(filepath)
    if not p.exists():  # before the extension is touched (reference: core.py:61-63, 102-104)
        raise FileNotFoundError(f"File not found: {p}")
    return str(p)


def _writable(filepath: PathLike) -> str:
    p = Path(filepath)
    p.parent.mkdir(parents=True, exist_ok=True)  # reference: core.py:130-131, 254-255
    return str(p)


def bind(native) -> dict:
    """the functions of `__all__` over `native` (a module with the reference's `_noLZSS` surface)"""

    def checked(data: Text, validate: bool):
        return validate_input(data) if validate else data

    def factorize(data: Text, validate: bool = True) -> Factors:
        """(start, length, ref) of every factor of `data` (reference: core.py:25-43)"""
        return native.factorize(checked(data, validate))

    def count_factors(data: Text, validate: bool = True) -> int:
        """number of factors of `data` (reference: core.py:68-86)"""
        return native.count_factors(checked(data, validate))

    def factorize_file(filepath: PathLike, reserve_hint: int = 0) -> Factors:
        """factors of the bytes of a file (reference: core.py:46-65)"""
        return native.factorize_file(_existing(filepath), reserve_hint)

    def count_factors_file(filepath: PathLike, validate: bool = True) -> int:
        """number of factors of the bytes of a file (reference: core.py:89-107; `validate` is unused there too)"""
        return native.count_factors_file(_existing(filepath))

    def write_factors_binary_file(data: Text, output_filepath: PathLike) -> None:
        """Kept from the reference (core.py:110-132 against bindings.cpp:180-187): the validated `data` goes to a
        parameter of the native function that is an input FILE PATH, so `data` must name a file (README.md:65).
        (No `validate` parameter: the reference's signature has none and always validates.)"""
        native.write_factors_binary_file(validate_input(data), _writable(output_filepath))

    def as_ascii_str(seq: Text) -> str:
        # the reference hands the two sequences to its binding as str: bytes are decoded as ASCII first, and bytes
        # above 0x7f raise UnicodeDecodeError there (core.py:201-205, 247-251)
        return seq.decode("ascii") if isinstance(seq, bytes) else seq

    def factorize_w_reference(reference_seq: Text, target_seq: Text, validate: bool = True) -> Factors:
        """the target factorized against reference + '\\x01' + target, positions absolute in that string
        (reference: core.py:164-207)"""
        return native.factorize_w_reference(as_ascii_str(checked(reference_seq, validate)),
                                            as_ascii_str(checked(target_seq, validate)))

    def factorize_w_reference_file(reference_seq: Text, target_seq: Text, output_path: PathLike,
                                   validate: bool = True) -> int:
        """the same into a v2 binary factor file; returns the number of factors (reference: core.py:210-257)"""
        return native.factorize_w_reference_file(as_ascii_str(checked(reference_seq, validate)),
                                                 as_ascii_str(checked(target_seq, validate)), _writable(output_path))

    fns = locals()
    return {name: fns[name] for name in __all__}


from . import _noLZSS as _mirror  # noqa: E402

globals().update(bind(_mirror))
