"""noLZSS.genomics.fasta (reference: src/noLZSS/genomics/fasta.py:28-126): the per-sequence GPU shard
dispatcher behind the reference's `read_nucleotide_fasta`."""
from nolzss_amd.genomics.fasta import (FASTAError, _parse_fasta_content, read_nucleotide_fasta,  # noqa: F401
                                       shard_nucleotide_fasta, lpt_assignment)

__all__ = ["FASTAError", "read_nucleotide_fasta", "shard_nucleotide_fasta"]
