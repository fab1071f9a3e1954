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


def _factorize_file_records(filepath, devices, want_factors, shard_index=0, shard_count=1):
    """(ids, lengths, counts, owners, arrays) for the records of a FASTA file: the native reader + batch
    (C ABI nolzss_read_nucleotide_fasta) for ASCII files, else this module's parser with the same plan."""
    filepath = Path(filepath)
    if not filepath.exists():
        raise FileNotFoundError(f"FASTA file not found: {filepath}")
    try:
        return _noLZSS.read_nucleotide_fasta_arrays(filepath, devices, want_factors, shard_index, shard_count)
    except _noLZSS.UnsupportedInput:
        pass  # non-ASCII bytes (a header in another script, or a broken file): Python's own decoding rules
    except RuntimeError as e:  # the reader's errors are the reference's FASTAError texts
        raise FASTAError(str(e))
    records = _load_validated(filepath)
    owners = lpt_assignment([len(s) for _, s in records], shard_count)
    mine = [j for j, o in enumerate(owners) if o == shard_index]
    try:
        got_counts, got = _noLZSS.factorize_batch([records[j][1] for j in mine], devices=devices,
                                                  want_factors=want_factors)
    except Exception as e:
        raise FASTAError(f"Failed to factorize sequences of '{filepath}': {e}")
    counts = [0] * len(records)
    arrays = [None] * len(records) if want_factors else None
    for k, j in enumerate(mine):
        counts[j] = got_counts[k]
        if want_factors:
            arrays[j] = got[k]
    return [rid for rid, _ in records], [len(s) for _, s in records], counts, owners, arrays


def read_nucleotide_fasta(filepath: Union[str, Path], devices: Optional[Sequence[int]] = None
                          ) -> List[Tuple[str, List[Tuple[int, int, int]]]]:
    """[(sequence_id, [(start, length, ref), ...]), ...] in first-appearance order
    (reference: fasta.py:79-126).  `devices` (extension) spreads the sequences over several
    GPUs of this process; default: the current device only."""
    ids, _, _, _, arrays = _factorize_file_records(filepath, devices, True)
    return [(rid, _noLZSS._tuples3(f)) for rid, f in zip(ids, arrays)]


def read_nucleotide_fasta_arrays(filepath: Union[str, Path], devices: Optional[Sequence[int]] = None,
                                 want_factors: bool = True):
    """Extension: the same records as (ids, factor counts, NumPy arrays of (start, length, ref)) -- for
    files whose tuple lists would not fit (config 4 of the benchmark: 2 * 10^8 tuples)."""
    ids, _, counts, _, arrays = _factorize_file_records(filepath, devices, want_factors)
    return ids, counts, arrays


def shard_nucleotide_fasta(filepath: Union[str, Path], want_factors: bool = False):
    """Multi-GPU form of read_nucleotide_fasta for a torch.distributed job (one process per GPU,
    backend nccl = RCCL on ROCm, or gloo on CPU-only ranks for tests).

    Every rank reads the file, takes the sequences the LPT plan assigns to it, factorizes them
    on its own GPU and contributes its per-sequence factor counts to ONE all-gather (the only
    collective on this path).  Returns (ids, counts, local) where counts[j] is the factor count
    of sequence j on every rank and local maps the indices owned by this rank to their factor
    arrays (None unless want_factors)."""
    import torch
    import torch.distributed as dist

    grouped = dist.is_initialized()  # (a process group of one rank still takes the collective path)
    world = dist.get_world_size() if grouped else 1
    rank = dist.get_rank() if grouped else 0
    ids, _, counts_mine, owners, arrays = _factorize_file_records(filepath, None, want_factors, rank, world)
    if grouped:
        use_cuda = dist.get_backend() == "nccl"
        dev = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
        mine_vec = torch.tensor(counts_mine, dtype=torch.int64, device=dev)  # zero for records of other ranks
        gathered = [torch.zeros_like(mine_vec) for _ in range(world)]
        dist.all_gather(gathered, mine_vec)
        counts = torch.stack(gathered).sum(dim=0).cpu().tolist()
    else:
        counts = list(counts_mine)
    local = {j: (arrays[j] if arrays is not None else None) for j, o in enumerate(owners) if o == rank}
    return ids, counts, local
