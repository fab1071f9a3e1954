"""DNA reference + target factorization wrappers (mirror of the reference's
noLZSS.genomics.sequences for the "next" row SURVEY.md 8f.2;
reference: src/noLZSS/genomics/sequences.py:19-53, 117-221)."""
from pathlib import Path
from typing import Union

from ..utils import validate_input


def is_dna_sequence(data: Union[str, bytes]) -> bool:
    """Only A, C, G, T (case-insensitive) (reference: sequences.py:19-34)."""
    if isinstance(data, bytes):
        try:
            data = data.decode("ascii")
        except UnicodeDecodeError:
            return False
    elif not isinstance(data, str):
        return False
    return all(c in "ACGT" for c in data.upper())


def _checked(reference_seq, target_seq, validate):
    if validate:
        reference_seq = validate_input(reference_seq)
        target_seq = validate_input(target_seq)
        if not is_dna_sequence(reference_seq):
            raise ValueError("Reference sequence must contain only DNA nucleotides (A, C, T, G)")
        if not is_dna_sequence(target_seq):
            raise ValueError("Target sequence must contain only DNA nucleotides (A, C, T, G)")
    if isinstance(reference_seq, bytes):
        reference_seq = reference_seq.decode("ascii")
    if isinstance(target_seq, bytes):
        target_seq = target_seq.decode("ascii")
    return reference_seq, target_seq


def factorize_dna_w_reference_seq(reference_seq: Union[str, bytes], target_seq: Union[str, bytes],
                                  validate: bool = True):
    """[(start, length, ref, is_rc)] of the target against reference + target with reverse
    complements (reference: sequences.py:117-165)."""
    from .._noLZSS import factorize_dna_w_reference_seq as _native
    reference_seq, target_seq = _checked(reference_seq, target_seq, validate)
    return _native(reference_seq, target_seq)


def factorize_dna_w_reference_seq_file(reference_seq: Union[str, bytes], target_seq: Union[str, bytes],
                                       output_path: Union[str, Path], validate: bool = True) -> int:
    """reference: sequences.py:168-221"""
    from .._noLZSS import factorize_dna_w_reference_seq_file as _native
    reference_seq, target_seq = _checked(reference_seq, target_seq, validate)
    output_path = Path(output_path)
    output_path.parent.mkdir(parents=True, exist_ok=True)
    return _native(reference_seq, target_seq, str(output_path))
