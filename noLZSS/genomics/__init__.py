"""noLZSS.genomics on the hot path (reference: src/noLZSS/genomics/__init__.py:8-25): the
reverse-complement entry points from the compiled module, the per-sequence FASTA batch and the
reference/target wrappers from the shared host layer."""
from .._noLZSS import (
    factorize_dna_w_rc,
    factorize_file_dna_w_rc,
    count_factors_dna_w_rc,
    count_factors_file_dna_w_rc,
    write_factors_binary_file_dna_w_rc,
    factorize_multiple_dna_w_rc,
    factorize_file_multiple_dna_w_rc,
    count_factors_multiple_dna_w_rc,
    count_factors_file_multiple_dna_w_rc,
    write_factors_binary_file_multiple_dna_w_rc,
    factorize_fasta_multiple_dna_w_rc,
    prepare_multiple_dna_sequences_w_rc,
)
from .fasta import FASTAError, read_nucleotide_fasta, shard_nucleotide_fasta
from .sequences import is_dna_sequence, factorize_dna_w_reference_seq, factorize_dna_w_reference_seq_file

__all__ = [
    "factorize_dna_w_rc", "factorize_file_dna_w_rc", "count_factors_dna_w_rc", "count_factors_file_dna_w_rc",
    "write_factors_binary_file_dna_w_rc", "factorize_multiple_dna_w_rc", "factorize_file_multiple_dna_w_rc",
    "count_factors_multiple_dna_w_rc", "count_factors_file_multiple_dna_w_rc",
    "write_factors_binary_file_multiple_dna_w_rc", "factorize_fasta_multiple_dna_w_rc",
    "prepare_multiple_dna_sequences_w_rc", "FASTAError", "read_nucleotide_fasta", "shard_nucleotide_fasta",
    "is_dna_sequence", "factorize_dna_w_reference_seq", "factorize_dna_w_reference_seq_file",
]
