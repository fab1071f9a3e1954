"""nolzss_amd -- MI355X-native non-overlapping LZSS factorization.

Drop-in for the factorize path of OmerKerner/noLZSS: the same `factorize`, `factorize_file`,
`count_factors`, `count_factors_file` (and, under `nolzss_amd.genomics`, `factorize_dna_w_rc`,
`prepare_multiple_dna_sequences_w_rc`, `read_nucleotide_fasta`) with identical results, computed
by hand-written HIP kernels for gfx950 behind a C ABI (include/nolzss_hip.h).
"""
from ._noLZSS import __version__
from .core import (factorize, factorize_file, count_factors, count_factors_file, write_factors_binary_file,
                   factorize_w_reference, factorize_w_reference_file)
from .utils import (NoLZSSError, InvalidInputError, validate_input, read_factors_binary_file,
                    read_binary_file_metadata, read_factors_binary_file_with_metadata)

__all__ = [
    "factorize", "factorize_file", "count_factors", "count_factors_file", "write_factors_binary_file",
    "factorize_w_reference", "factorize_w_reference_file",
    "NoLZSSError", "InvalidInputError", "validate_input", "read_factors_binary_file",
    "read_binary_file_metadata", "read_factors_binary_file_with_metadata", "__version__",
]
