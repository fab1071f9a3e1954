"""Thread-parallel API of the reference (reference: src/noLZSS/parallel.py).

On the MI355X the whole pipeline is data-parallel, so `num_threads` has no meaning: the functions
keep the reference's signatures and results (the reference's own tests require the parallel result
to equal the sequential one, tests/test_parallel_fasta.py:294-323, 516-553) and ignore it."""
import os
import tempfile
from collections import namedtuple
from pathlib import Path
from typing import List, Union

from . import _noLZSS
from .utils import validate_input, read_factors_binary_file

Factor = namedtuple("Factor", ["start", "length", "ref"])


def parallel_factorize_to_file(text: Union[str, bytes], output_path: Union[str, Path], num_threads: int = 0,
                               start_pos: int = 0, validate: bool = True) -> int:
    """reference: parallel.py:23-54"""
    if validate:
        text = validate_input(text)
    return _noLZSS.parallel_factorize_to_file(text, str(Path(output_path)), num_threads, start_pos)


def parallel_factorize_file_to_file(input_path: Union[str, Path], output_path: Union[str, Path],
                                    num_threads: int = 0, start_pos: int = 0) -> int:
    """reference: parallel.py:57-86"""
    input_path = Path(input_path)
    if not input_path.exists():
        raise FileNotFoundError(f"Input file not found: {input_path}")
    return _noLZSS.parallel_factorize_file_to_file(str(input_path), str(Path(output_path)), num_threads, start_pos)


def parallel_factorize(text: Union[str, bytes], num_threads: int = 0, start_pos: int = 0,
                       validate: bool = True) -> List[Factor]:
    """reference: parallel.py:89-159 (factors come back through a temporary v2 file)"""
    with tempfile.NamedTemporaryFile(mode="wb", suffix=".bin", delete=False) as tmp:
        temp_path = Path(tmp.name)
    try:
        parallel_factorize_to_file(text, temp_path, num_threads, start_pos, validate)
        return [Factor(*f) for f in read_factors_binary_file(temp_path)]
    finally:
        if temp_path.exists():
            os.unlink(temp_path)


def parallel_factorize_dna_w_rc_to_file(text: Union[str, bytes], output_path: Union[str, Path],
                                        num_threads: int = 0, validate: bool = True) -> int:
    """reference: parallel.py:162-193"""
    if validate:
        text = validate_input(text)
    return _noLZSS.parallel_factorize_dna_w_rc_to_file(text, str(Path(output_path)), num_threads)


def parallel_factorize_file_dna_w_rc_to_file(input_path: Union[str, Path], output_path: Union[str, Path],
                                             num_threads: int = 0) -> int:
    """reference: parallel.py:196-226"""
    input_path = Path(input_path)
    if not input_path.exists():
        raise FileNotFoundError(f"Input file not found: {input_path}")
    return _noLZSS.parallel_factorize_file_dna_w_rc_to_file(str(input_path), str(Path(output_path)), num_threads)
