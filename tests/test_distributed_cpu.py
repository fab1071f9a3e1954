"""World-size-2 gloo tests of the multi-GPU FASTA shard path (no GPU): the per-rank device call is
replaced by the oracle so that the reader, the sharding plan and the all-gather of counts are exercised.
Two variants: the library's reader + plan (as on the GPU box), and the Python reader + plan that
non-ASCII files fall back to."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fasta, outdir, variant):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import oracle_lib as oracle
    from nolzss_amd import _noLZSS
    from nolzss_amd.genomics import fasta as F

    def fake_batch(texts, devices=None, want_factors=True):
        arrays = [oracle.factors_array(bytes(t)) for t in texts]
        return [len(a) for a in arrays], (arrays if want_factors else None)

    def fake_native(path, devices=None, want_factors=True, shard_index=0, shard_count=1):
        if variant == "python":
            raise _noLZSS.UnsupportedInput("forced")
        recs = _noLZSS.debug_parse_nucleotide_fasta(path)                 # the library's reader (host only)
        lens = [len(s) for _, s in recs]
        owners = _noLZSS.debug_lpt_plan(lens, shard_count)                # the library's plan
        arrays = [oracle.factors_array(s) if o == shard_index else None for (_, s), o in zip(recs, owners)]
        counts = [len(a) if a is not None else 0 for a in arrays]
        return [i.decode() for i, _ in recs], lens, counts, owners, (arrays if want_factors else None)

    _noLZSS.factorize_batch = fake_batch  # the CPU checker stands in for the GPU on this rank
    _noLZSS.read_nucleotide_fasta_arrays = fake_native
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids, counts, local = F.shard_nucleotide_fasta(fasta, want_factors=True)
    np.save(Path(outdir) / f"counts{rank}.npy", np.array(counts, dtype=np.int64))
    np.save(Path(outdir) / f"owned{rank}.npy", np.array(sorted(local), dtype=np.int64))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("variant", ["native", "python"])
def test_shard_fasta_two_ranks_gloo(tmp_path, variant):
    import gen
    import oracle_lib as oracle
    recs = [(f"seq{k}", gen.random_dna(2000 + 531 * k, 100 + k)) for k in range(7)]
    fasta = tmp_path / "in.fa"
    gen.write_fasta(fasta, recs)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(fasta), str(tmp_path), variant), nprocs=2, join=True)
    c0 = np.load(tmp_path / "counts0.npy")
    c1 = np.load(tmp_path / "counts1.npy")
    expected = np.array([oracle.count_factors(seq) for _, seq in recs])
    assert np.array_equal(c0, expected) and np.array_equal(c1, expected)   # every rank sees all counts
    o0 = set(np.load(tmp_path / "owned0.npy").tolist())
    o1 = set(np.load(tmp_path / "owned1.npy").tolist())
    assert o0 | o1 == set(range(7)) and not (o0 & o1)                        # disjoint cover
