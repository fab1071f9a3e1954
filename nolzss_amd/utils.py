"""Input validation for the factorize path (mirror of the reference's noLZSS.utils,
reference: src/noLZSS/utils.py:16-58; only what the hot path needs)."""
from typing import Union


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
