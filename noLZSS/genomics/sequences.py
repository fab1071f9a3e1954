"""noLZSS.genomics.sequences (reference: src/noLZSS/genomics/sequences.py)."""
from nolzss_amd.genomics.sequences import (is_dna_sequence, factorize_dna_w_reference_seq,  # noqa: F401
                                           factorize_dna_w_reference_seq_file)

__all__ = ["is_dna_sequence", "factorize_dna_w_reference_seq", "factorize_dna_w_reference_seq_file"]
