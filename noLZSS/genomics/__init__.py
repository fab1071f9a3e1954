"""noLZSS.genomics on the hot path (reference: src/noLZSS/genomics/__init__.py:8-25): the
reverse-complement entry points from the compiled module, the per-sequence FASTA batch and the
reference/target wrappers from the shared host layer."""
from .. import _noLZSS
from .fasta import FASTAError, read_nucleotide_fasta, shard_nucleotide_fasta
from .sequences import is_dna_sequence, factorize_dna_w_reference_seq, factorize_dna_w_reference_seq_file

# the names the reference's package takes from its extension module at import time
_NATIVE = tuple(f"{verb}{mode}" for mode in ("_dna_w_rc", "_multiple_dna_w_rc")
                for verb in ("factorize", "factorize_file", "count_factors", "count_factors_file",
                             "write_factors_binary_file")) + (
    "factorize_fasta_multiple_dna_w_rc", "prepare_multiple_dna_sequences_w_rc")
globals().update({name: getattr(_noLZSS, name) for name in _NATIVE})

__all__ = list(_NATIVE) + ["FASTAError", "read_nucleotide_fasta", "shard_nucleotide_fasta",
                           "is_dna_sequence", "factorize_dna_w_reference_seq", "factorize_dna_w_reference_seq_file"]
