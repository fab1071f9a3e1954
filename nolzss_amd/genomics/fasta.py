"""FASTA batch path: parse, validate, and factorize every sequence independently.

Mirror of the reference's noLZSS.genomics.fasta for the hot path
(reference: src/noLZSS/genomics/fasta.py:28-126).  `read_nucleotide_fasta` keeps the reference's
signature and result shape; the per-sequence factorize() loop (fasta.py:110-122) is replaced by
the per-sequence GPU shard dispatcher: sequences are independent factorizations, so they are
dealt over the ranks of a torch.distributed job (one process per GPU) or, in a single process,
over the visible devices, and only the per-sequence factor counts are all-gathered (RCCL).
"""
import re
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple, Union

from .. import _noLZSS
from ..utils import NoLZSSError


class FASTAError(NoLZSSError):
    """FASTA parsing / validation failure (reference: fasta.py:23-25)."""


def _parse_fasta_content(content: str) -> Dict[str, str]:
    """{id: UPPERCASE sequence}; id = first word of the header; later duplicates overwrite
    (reference: fasta.py:28-76)."""
    sequences: Dict[str, str] = {}
    current_id: Optional[str] = None
    chunks: List[str] = []
    for line_num, raw in enumerate(content.splitlines(), 1):
        line = raw.strip()
        if not line:
            continue
        if line.startswith(">"):
            if current_id is not None:
                sequences[current_id] = "".join(chunks)
            header = line[1:].strip()
            if not header:
                raise FASTAError(f"Empty sequence header at line {line_num}")
            current_id = header.split()[0]
            chunks = []
        else:
            if current_id is None:
                raise FASTAError(f"Sequence data before header at line {line_num}")
            chunks.append(re.sub(r"\s", "", line.upper()))
    if current_id is not None:
        sequences[current_id] = "".join(chunks)
    if not sequences:
        raise FASTAError("No valid sequences found in FASTA file")
    return sequences


def _load_validated(filepath: Union[str, Path]) -> List[Tuple[str, bytes]]:
    filepath = Path(filepath)
    if not filepath.exists():
        raise FileNotFoundError(f"FASTA file not found: {filepath}")
    try:
        content = filepath.read_text(encoding="utf-8")
    except UnicodeDecodeError as e:
        raise FASTAError(f"File encoding error: {e}")
    records = []
    for seq_id, sequence in _parse_fasta_content(content).items():
        if not re.match(r"^[ACGT]+$", sequence):
            invalid = set(sequence) - set("ACGT")
            raise FASTAError(f"Sequence '{seq_id}' contains invalid nucleotides: {invalid}")
        records.append((seq_id, sequence.encode("ascii")))
    return records


def lpt_assignment(lengths: Sequence[int], n_bins: int) -> List[int]:
    """Longest-processing-time-first bin packing: bin index per sequence.  Deterministic, so
    every rank computes the same plan without communicating."""
    loads = [0] * n_bins
    owner = [0] * len(lengths)
    for j in sorted(range(len(lengths)), key=lambda j: (-lengths[j], j)):
        b = min(range(n_bins), key=lambda b: (loads[b], b))
        owner[j] = b
        loads[b] += lengths[j]
    return owner


def read_nucleotide_fasta(filepath: Union[str, Path], devices: Optional[Sequence[int]] = None
                          ) -> List[Tuple[str, List[Tuple[int, int, int]]]]:
    """[(sequence_id, [(start, length, ref), ...]), ...] in first-appearance order
    (reference: fasta.py:79-126).  `devices` (extension) spreads the sequences over several
    GPUs of this process; default: the current device only."""
    records = _load_validated(filepath)
    try:
        _, arrays = _noLZSS.factorize_batch([seq for _, seq in records], devices=devices, want_factors=True)
    except Exception as e:
        raise FASTAError(f"Failed to factorize sequences of '{filepath}': {e}")
    return [(seq_id, _noLZSS._tuples3(f)) for (seq_id, _), f in zip(records, arrays)]


def shard_nucleotide_fasta(filepath: Union[str, Path], want_factors: bool = False):
    """Multi-GPU form of read_nucleotide_fasta for a torch.distributed job (one process per GPU,
    backend nccl = RCCL on ROCm, or gloo on CPU-only ranks for tests).

    Every rank parses the file, takes the sequences the LPT plan assigns to it, factorizes them
    on its own GPU and contributes its per-sequence factor counts to ONE all-gather (the only
    collective on this path).  Returns (ids, counts, local) where counts[j] is the factor count
    of sequence j on every rank and local maps the indices owned by this rank to their factor
    arrays (None unless want_factors)."""
    import torch
    import torch.distributed as dist

    records = _load_validated(filepath)
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    owner = lpt_assignment([len(s) for _, s in records], world)
    mine = [j for j, o in enumerate(owner) if o == rank]
    counts_local, arrays = _noLZSS.factorize_batch([records[j][1] for j in mine], want_factors=want_factors)
    m = len(records)
    if world > 1:
        use_cuda = dist.get_backend() == "nccl"
        dev = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
        mine_vec = torch.zeros(m, dtype=torch.int64, device=dev)
        if mine:
            mine_vec[torch.tensor(mine, device=dev)] = torch.tensor(counts_local, dtype=torch.int64, device=dev)
        gathered = [torch.zeros_like(mine_vec) for _ in range(world)]
        dist.all_gather(gathered, mine_vec)
        counts = torch.stack(gathered).sum(dim=0).cpu().tolist()
    else:
        counts = [0] * m
        for j, c in zip(mine, counts_local):
            counts[j] = c
    local = {j: (arrays[k] if arrays is not None else None) for k, j in enumerate(mine)}
    return [rid for rid, _ in records], counts, local
