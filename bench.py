#!/usr/bin/env python3
"""bench.py -- bases/s factorized on synthetic sigma=4 DNA, MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json config 3, the configuration the metric is quoted on): one 2^30-base
synthetic DNA string (sigma = 4, 40 % copied blocks with 1 % substitutions, generator and seed in
tests/gen.py) PER GPU -- the multi-sequence shard of north_star with one sequence per rank, so
per-GPU work is fixed as N grows (weak scaling).  Rank r factorizes its own sequence (seed + r);
the only collective is the all-gather of the per-sequence factor counts (RCCL over xGMI).

A step = one complete factorization of the rank's sequence with the text already resident in HBM:
pack -> suffix array -> LCP -> L* -> chain -> all z factor records (start, length, ref) built in
HBM.  The PCIe download of the records is outside `value` (reported as pcie_inclusive_*).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import gen  # noqa: E402
from nolzss_amd import _noLZSS as native  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured copy
# rs_scatter_kernel: 2 * (sizeof(key) + 4) algorithmic bytes per (key, value) pair per launch
# (24 B for the u64-key sorts, 16 B for the u32-key partition passes); the library sums them.
DOMINANT = "rs_scatter"


def measured_traffic_ratio():
    """HBM bytes / algorithmic bytes of rs_scatter_kernel from the committed rocprofv3 PMC passes
    (FETCH_SIZE and WRITE_SIZE in separate runs, FETCH_SIZE doubled per the gfx950 correction);
    None if the profile summary is not there."""
    try:
        with open(ROOT / "profiles" / "r01_pmc_radix_traffic.json") as f:
            d = json.load(f)
            cur = d.get("rs_scatter_kernel<u32,u32> at 2^30 pairs (current pipeline)")  # the dominant instantiation
            return float((cur or d["rs_scatter_kernel<u64>"])["traffic_over_algorithmic"])
    except Exception:
        return None


def make_text(workload: str, n: int, rank: int) -> np.ndarray:
    if workload == "dna1g":
        return gen.repeat_dna(n, seed=0x5EED0003 + rank)
    if workload == "random":
        return gen.random_dna(n, seed=0x6E6F4C5A + rank)
    raise ValueError(workload)


def cpu_baseline(text: np.ndarray, sample: int):
    """The oracle (CPU restatement of the reference algorithm, single thread) on a bounded prefix
    of the same workload.  Reported beside the GPU number; it is a baseline, not the target."""
    import oracle_lib as oracle  # checker only: never on the measured path
    sample = min(sample, len(text))
    t0 = time.perf_counter()
    z = oracle.count_factors(text[:sample])
    dt = time.perf_counter() - t0
    return {"value": sample / dt, "unit": "bases/s", "cores": 1, "kind": "port",
            "sample": f"first {sample} bases of the rank-0 sequence, count_factors, {dt:.1f} s, z={z}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="dna1g", choices=["dna1g", "random"])
    ap.add_argument("--log2n", type=int, default=30, help="bases per GPU = 2^log2n (default 2^30)")
    ap.add_argument("--cpu-sample-log2", type=int, default=26)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL over xGMI; gloo only for rehearsals)")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses cuda:0 (needs --backend gloo)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if a.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank)
    native.set_device(local_rank)

    n = 1 << a.log2n
    text = make_text(a.workload, n, rank)
    d_text = torch.from_numpy(text).to(dev)
    cdev = dev if a.backend == "nccl" else torch.device("cpu")
    counts = torch.zeros(world, dtype=torch.int64, device=cdev)
    mine = torch.zeros(1, dtype=torch.int64, device=cdev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        z, _ = native.factorize_device(d_text.data_ptr(), n, emit=1)
        if world > 1:  # the only collective: per-sequence factor counts
            mine[0] = z
            dist.all_gather_into_tensor(counts, mine)
        return z

    for _ in range(a.warmup):
        step()
    native.profile_enable(True)
    native.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        z = step()
    barrier()
    elapsed = time.perf_counter() - t0
    stats = native.profile_report()
    native.profile_enable(False)

    t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # PCIe-inclusive variant (records downloaded into host memory), rank-local, one run
    t1 = time.perf_counter()
    z2, f = native.factorize_device(d_text.data_ptr(), n, emit=2)
    pcie_dt = time.perf_counter() - t1
    assert z2 == z
    del f

    if rank == 0:
        total_bases = float(n) * world * a.steps
        # every launch of the kernel, whatever the instantiation (the library reports them by class:
        # rs_scatter.text = first pass computing the keys, .u32 / .u64 = key width, .small = < 2^24 pairs)
        family = {k: v for k, v in stats.items() if k.startswith(DOMINANT)}
        cnt = sum(v[0] for v in family.values())
        ms = sum(v[1] for v in family.values())
        nbytes = sum(v[2] for v in family.values())
        achieved = (nbytes / (ms * 1e-3)) / 1e9 if ms > 0 else 0.0
        by_class = {k: {"launches": v[0], "ms_per_step": v[1] / a.steps,
                        "achieved_GBps": (v[2] / (v[1] * 1e-3)) / 1e9 if v[1] > 0 else 0.0}
                    for k, v in sorted(family.items())}
        nested = {"rs_hist", "rs_scan", "bucket_scatter", "window_scatter"} | set(family)
        ratio = measured_traffic_ratio()
        out = {
            "metric": "bases/sec factorized (1 GB sigma=4 DNA) + HBM GB/s fraction",
            "value": total_bases / elapsed,
            "unit": "bases/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"{a.workload}: one 2^{a.log2n}-base synthetic DNA sequence per GPU "
                                   "(sigma=4, 40% copied blocks, 1% substitutions), full factorization "
                                   "to (start,length,ref) records in HBM",
                       "bases_per_gpu": n, "factors_per_sequence": int(z),
                       "parallelism": f"{world} independent sequence shard(s), all-gather of counts"},
            "roofline": {"bound": "hbm", "kernel": "rs_scatter_kernel", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (ratio * nbytes / cnt) if (ratio and cnt) else None,
                         "traffic_source": "profiles/r01_pmc_radix_traffic.json (rocprofv3 --pmc FETCH_SIZE / "
                                           "WRITE_SIZE, separate passes): HBM bytes = 1.02 x algorithmic bytes, u32 passes at 2^30 pairs",
                         "launches": cnt,
                         "avg_launch_ms": (ms / cnt) if cnt else None,
                         "algorithmic_bytes_per_launch": (nbytes / cnt) if cnt else None,
                         "by_class": by_class},
            "pcie_inclusive_bases_per_s": n / pcie_dt,
            "stages_ms_per_step": {k: v[1] / a.steps for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1])
                                   if k not in nested},
            "kernels_ms_per_step": {k: stats[k][1] / a.steps for k in sorted(nested) if k in stats},
        }
        if not a.no_cpu_baseline and world == 1:  # the CPU leg runs on rank 0 of the one-GPU run only
            out["cpu_baseline"] = cpu_baseline(text, 1 << a.cpu_sample_log2)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
