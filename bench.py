#!/usr/bin/env python3
"""bench.py -- bases/s factorized on synthetic sigma=4 DNA, MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
  python bench.py --gpus N ...       (WORLD_SIZE unset: starts its own N ranks as child processes)

Headline workload (BASELINE.json config 3, the configuration the metric is quoted on): one 2^30-base
synthetic DNA string (sigma = 4, 40 % copied blocks with 1 % substitutions, generator and seed in
tests/gen.py) PER GPU -- the multi-sequence shard of north_star with one sequence per rank, so per-GPU
work is fixed as N grows (weak scaling).  Rank r factorizes its own sequence (seed + r); the only
collective is the all-gather of the per-sequence factor counts (RCCL over xGMI).

A step = one complete factorization of the rank's sequence with the text already resident in HBM:
pack -> suffix array -> LCP -> L* -> chain -> all z factor records (start, length, ref) built in HBM.

Beside `value` the same JSON line carries
  * "stopwatches" (N = 1): the three clocks of SURVEY.md 8d -- C ABI host buffer -> host factor array,
    count_factors from a host buffer, and the Python-visible noLZSS.factorize() (tuple list) on a prefix;
  * "rc256m": BASELINE config 5 on one GPU per rank (factorize_dna_w_rc of 2^28 bases resident in HBM, SURVEY.md
    8d's 84 B/base, stage table, its own cpu_baseline);
  * "fasta512": BASELINE config 4, the multi-sequence FASTA shard (512 records x 4 Mi bases, generator of
    config 2 with seeds 0x4000 + k) dealt over the N ranks by the longest-processing-time-first plan, one
    all-gather of the 512 factor counts -- STRONG scaling (total work fixed), records resident in HBM when
    its clock starts.  `--workload fasta512` makes this the headline instead and also times the
    file -> reader -> batch -> counts path through genomics.shard_nucleotide_fasta.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import gen  # noqa: E402  (NumPy generators of the synthetic inputs; no device code)

# torch, torch.distributed and the library are imported by main() AFTER the decision to start the ranks as
# child processes: the parent of a self-launched run never touches the GPU (and never execs).
torch = dist = native = None


def late_imports():
    global torch, dist, native
    import torch as _torch
    import torch.distributed as _dist
    from nolzss_amd import _noLZSS as _native
    torch, dist, native = _torch, _dist, _native


def self_launch(a, argv) -> int:
    """`python bench.py --gpus N` outside a launcher: start the N ranks as child processes through
    torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1) and relay their output; rank 0
    prints the JSON line to the stdout this process shares with them."""
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + list(argv)
    return subprocess.run(cmd).returncode

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured copy
ALG_BYTES_PER_BASE = 50.0  # SURVEY.md 8d, plain mode: compulsory traffic of the whole pipeline
ALG_BYTES_PER_BASE_RC = 84.0  # SURVEY.md 8d, reverse-complement mode (arrays over |S| = 2n + 2)
# rs_scatter_kernel: 2 * (sizeof(key) + 4) algorithmic bytes per (key, value) pair per launch
# (24 B for the u64-key sorts, 16 B for the u32-key partition passes); the library sums them.
# (round 4: the key sort ends with local_sort_kernel -- the same ranking on sub-buckets held in LDS, 16 algorithmic bytes
# per pair --; the library reports it as "rs_local_sort" and it is counted with the family)
DOMINANT = "rs_scatter"
FAMILY_EXTRA = {"rs_local_sort"}


def in_family(name):
    return name.startswith(DOMINANT) or name in FAMILY_EXTRA


PMC_FILES = ["r04_pmc_radix_traffic.json", "r03_pmc_radix_traffic.json", "r02_pmc_radix_traffic.json", "r01_pmc_radix_traffic.json"]
# per-kernel counters of ONE factorization of the 2^30-base benchmark text (tools/pmc_step.sh; committed summary)
STEP_PMC_FILE = "r04_pmc_step.json"
SHADER_CLOCK_HZ = 2.4e9   # MI355X peak engine clock
SIMDS = 256 * 4           # 256 CUs x 4 SIMDs; a wave64 VALU instruction occupies its SIMD for 4 cycles


def measure_copy_ceiling(dev, nbytes=1 << 30, reps=6):
    """Streaming-copy ceiling of this box, measured before the timed region ("report both"): a device-to-
    device copy of 1 GiB moves 2 GiB (read + write); best of `reps`, HIP events on torch's stream."""
    a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    b = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    a.zero_()
    b.copy_(a)
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    del a, b
    torch.cuda.empty_cache()
    return 2.0 * nbytes / (best * 1e-3) / 1e9


def step_counters():
    """The committed per-kernel counter summary (tools/pmc_step.sh) or None."""
    try:
        with open(ROOT / "profiles" / STEP_PMC_FILE) as f:
            return json.load(f)
    except Exception:
        return None


HBM_ACHIEVABLE_GBS = 6300.0  # MI355X_MICROARCH.md: what a streaming kernel reaches of the 8 TB/s spec peak

# library stage name -> (kernels of the counter summary that run inside it, what bounds it)
KERNEL_OF_STAGE = [
    ("rs_scatter.u32", ["rs_scatter_kernel<u32, u32, ArraySrc<u32>, u32>", "rs_scatter_kernel<u32, u32, RankSrc, u64>",
                        "rs_scatter_kernel<u32, u16, PairSrc, u64>"], "hbm"),
    ("sa_direct_sort", ["group_refine_kernel<2, false>"], "line fills (random windows of the text) + valu_issue"),
    ("lpf", ["lpf_tile_kernel<false>"], "lds + valu_issue"),
    ("factor_emit", ["factor_kernel<false, false>"], "line fills (random blocks of the pyramids)"),
    ("sa_regroup", ["regroup_kernel<true, 3>", "regroup_kernel<false, 0>", "compact_survivors_kernel"], "valu_issue + look-back latency"),
    ("window_scatter", ["window_scatter2_kernel"], "hbm"),
    ("rs_scatter.text", ["rs_scatter_kernel<u64, u32, Text16Src, u32>"], "valu_issue + hbm writes"),
    ("rs_local_sort", ["local_sort_kernel<2, true>", "local_sort_kernel<2, false>"],
     "valu_issue + lds at 12 waves per CU (sort and the regroup of round 0 in one kernel) + hbm"),
    ("rs_hist", ["rs_hist_kernel<u32, ArraySrc<u32> >", "rs_hist_kernel<u32, PairSrc>", "rs_hist_kernel<u32, RankSrc>",
                 "rs_hist_kernel<u64, Text16Src>"], "hbm + per-tile latency"),
    ("chain_exit", ["chain_exit_kernel"], "hbm + lds"),
]


def kernel_bounds(stats, steps, pmc):
    """roofline.kernels: the largest kernels of the step, each with its own bound.  Durations are this run's HIP events
    (live); instruction, LDS and HBM byte counts come from the committed counter summary of ONE factorization of the same
    text (derived: they do not change unless the kernel does).  valu_issue_frac = VALU wave-instructions x 4 cycles /
    (1024 SIMDs x clock) over the kernel's time; lds_frac = LDS-active cycles per CU over its time; hbm_GBps = FETCH_SIZE
    (gfx950-corrected) + WRITE_SIZE over its time, as a fraction of the 8 TB/s spec peak (hbm_frac) and of the 6.3 TB/s a
    streaming kernel reaches (hbm_frac_of_achievable): for the random-access kernels these bytes are 128-byte line fills."""
    out = []
    ks = (pmc or {}).get("kernels", {})
    for stage, kernels, bound in KERNEL_OF_STAGE:
        if stage not in stats or stats[stage][1] <= 0:
            continue
        cnt, ms, nbytes = stats[stage]
        e = {"stage": stage, "kernels": kernels, "bound": bound, "ms_per_step": ms / steps, "launches_per_step": cnt / steps}
        if nbytes:
            e["algorithmic_GBps"] = nbytes / (ms * 1e-3) / 1e9
            e["algorithmic_frac_of_hbm_peak"] = e["algorithmic_GBps"] / HBM_PEAK_GBS
        cs = [ks[k] for k in kernels if k in ks]
        if cs and pmc.get("bases") == stats.get("_bases"):
            t = ms / steps * 1e-3
            tot = lambda name: sum(c.get(name, 0.0) for c in cs)  # noqa: E731  (one factorization = one step)
            if tot("SQ_INSTS_VALU"):
                e["valu_issue_frac"] = tot("SQ_INSTS_VALU") * 4.0 / (SIMDS * SHADER_CLOCK_HZ) / t
                e["salu_per_valu"] = tot("SQ_INSTS_SALU") / tot("SQ_INSTS_VALU")
            if tot("SQ_LDS_IDX_ACTIVE"):
                e["lds_frac"] = tot("SQ_LDS_IDX_ACTIVE") / 256.0 / SHADER_CLOCK_HZ / t
                e["lds_bank_conflict_share"] = tot("SQ_LDS_BANK_CONFLICT") / tot("SQ_LDS_IDX_ACTIVE")
            hbm = tot("fetch_bytes") + tot("write_bytes")
            if hbm:
                e["hbm_GBps"] = hbm / t / 1e9
                e["hbm_frac"] = e["hbm_GBps"] / HBM_PEAK_GBS
                e["hbm_frac_of_achievable"] = e["hbm_GBps"] / HBM_ACHIEVABLE_GBS
            e["counters_derived_from"] = f"profiles/{STEP_PMC_FILE}"
        out.append(e)
    out.sort(key=lambda e: -e["ms_per_step"])
    return out


def measured_traffic_ratio():
    """HBM bytes / algorithmic bytes of rs_scatter_kernel from the committed rocprofv3 PMC passes
    (FETCH_SIZE and WRITE_SIZE in separate runs, FETCH_SIZE doubled per the gfx950 correction);
    (ratio, file) or (None, None) if no profile summary is there."""
    for name in PMC_FILES:
        try:
            with open(ROOT / "profiles" / name) as f:
                d = json.load(f)
            cur = d.get("rs_scatter_kernel<u32,u32> at 2^30 pairs (current pipeline)")  # the dominant instantiation
            return float((cur or d["rs_scatter_kernel<u64>"])["traffic_over_algorithmic"]), name
        except Exception:
            continue
    return None, None


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


def cpu_baseline_fasta(a):
    """The FASTA shard on the host: the oracle on a bounded sample of the same records, one record per
    thread on all cores (the reference's own per-sequence work queue, parallel_fasta_processor.cpp:360-385)."""
    import oracle_lib as oracle  # checker only: never on the measured path
    # (the box's share of host cores for one GPU is 16, whatever os.cpu_count() says)
    cores = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    L = 1 << a.fasta_record_log2
    sample = min(a.fasta_records, 2 * cores)
    recs = [gen.random_dna(L, 0x4000 + j) for j in range(sample)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as pool:  # (the oracle releases the GIL: ctypes)
        zs = list(pool.map(oracle.count_factors, recs))
    dt = time.perf_counter() - t0
    return {"value": sample * L / dt, "unit": "bases/s", "cores": cores, "kind": "port",
            "sample": f"the first {sample} of the {a.fasta_records} records, count_factors, one record per thread, "
                      f"{dt:.1f} s, z[0]={zs[0]}"}


class Job:
    """rank / world / collectives of this process"""

    def __init__(self, a):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        if self.world != a.gpus:
            raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={self.world} (start it without WORLD_SIZE and it "
                             "launches its own ranks, or give torch.distributed.run the same number)")
        self.local_rank = 0 if a.same_device else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(self.local_rank)
        self.backend = a.backend
        # --force-collectives: a one-rank run still opens the process group and issues every collective of the
        # multi-rank path (the RCCL code path on a one-GPU box)
        self.collectives = self.world > 1 or a.force_collectives
        if self.collectives:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.world == 1:
                with socket.socket() as sock:
                    sock.bind(("127.0.0.1", 0))
                    os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
                os.environ.setdefault("RANK", "0")
                os.environ.setdefault("WORLD_SIZE", "1")
            if a.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group("gloo")
        self.dev = torch.device("cuda", self.local_rank)
        self.cdev = self.dev if a.backend == "nccl" else torch.device("cpu")
        native.set_device(self.local_rank)

    def barrier(self):
        if self.collectives:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, seconds: float) -> float:
        t = torch.tensor([seconds], dtype=torch.float64, device=self.cdev)
        if self.collectives:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())


def run_single_sequence(job: Job, a):
    """headline: one sequence per GPU, text resident in HBM, records built in HBM"""
    n = 1 << a.log2n
    text = make_text(a.workload, n, job.rank)
    d_text = torch.from_numpy(text).to(job.dev)
    counts = torch.zeros(job.world, dtype=torch.int64, device=job.cdev)
    mine = torch.zeros(1, dtype=torch.int64, device=job.cdev)

    def step():
        z, _ = native.factorize_device(d_text.data_ptr(), n, emit=1)
        if job.collectives:  # the only collective: per-sequence factor counts
            mine[0] = z
            dist.all_gather_into_tensor(counts, mine)
        return z

    copy_ceiling = measure_copy_ceiling(job.dev) if job.rank == 0 else None
    for _ in range(a.warmup):
        step()
    native.profile_enable(True)
    native.profile_reset()
    job.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        z = step()
    job.barrier()
    elapsed = time.perf_counter() - t0
    stats = native.profile_report()
    native.profile_enable(False)
    # every rank's own clock, device and factor count, for rank 0's line
    mine_row = torch.tensor([elapsed / a.steps * 1e3, float(job.local_rank), float(z)], dtype=torch.float64, device=job.cdev)
    rank_rows = torch.zeros(job.world * 3, dtype=torch.float64, device=job.cdev)
    if job.collectives:
        dist.all_gather_into_tensor(rank_rows, mine_row)
    else:
        rank_rows[:3] = mine_row
    rank_rows = rank_rows.view(job.world, 3).cpu().tolist()
    elapsed = job.max_over_ranks(elapsed)

    # PCIe-inclusive variant (records downloaded into host memory), rank-local, one run
    t1 = time.perf_counter()
    z2, f = native.factorize_device(d_text.data_ptr(), n, emit=2)
    pcie_dt = time.perf_counter() - t1
    assert z2 == z
    del f
    del d_text
    torch.cuda.empty_cache()
    if job.rank != 0:
        return None, text, z

    total_bases = float(n) * job.world * a.steps
    # every launch of the kernel, whatever the instantiation (the library reports them by class:
    # rs_scatter.text = first pass computing the keys, .u32 / .u64 = key width, .small = < 2^24 pairs)
    family = {k: v for k, v in stats.items() if in_family(k)}
    cnt = sum(v[0] for v in family.values())
    ms = sum(v[1] for v in family.values())
    nbytes = sum(v[2] for v in family.values())
    achieved = (nbytes / (ms * 1e-3)) / 1e9 if ms > 0 else 0.0
    by_class = {k: {"launches": v[0], "ms_per_step": v[1] / a.steps,
                    "achieved_GBps": (v[2] / (v[1] * 1e-3)) / 1e9 if v[1] > 0 else 0.0}
                for k, v in sorted(family.items())}
    nested = {"rs_hist", "rs_scan", "bucket_scatter", "window_scatter"} | set(family)
    ratio, ratio_file = measured_traffic_ratio()
    pmc = step_counters()
    step_s = elapsed / a.steps
    out = {
        "metric": "bases/sec factorized (1 GB sigma=4 DNA) + HBM GB/s fraction",
        "value": total_bases / elapsed,
        "unit": "bases/s",
        "n_gpus": job.world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": step_s * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": f"{a.workload}: one 2^{a.log2n}-base synthetic DNA sequence per GPU "
                               "(sigma=4, 40% copied blocks, 1% substitutions), full factorization "
                               "to (start,length,ref) records in HBM",
                   "bases_per_gpu": n, "factors_per_sequence": int(z),
                   "parallelism": f"{job.world} independent sequence shard(s), all-gather of counts"},
        "roofline": {"bound": "hbm", "kernel": "rs_scatter_kernel + local_sort_kernel (the radix sort family)", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     # the second ceiling of SURVEY.md 8d: a streaming copy measured on this box before the timed region
                     "peak_measured": copy_ceiling, "frac_of_measured": (achieved / copy_ceiling) if copy_ceiling else None,
                     "peak_measured_how": "device-to-device copy of 1 GiB (2 GiB moved), best of 6, HIP events",
                     "traffic": (ratio * nbytes / cnt) if (ratio and cnt) else None,
                     "traffic_kind": "derived" if ratio_file else None,
                     "traffic_derived_from": f"profiles/{ratio_file}" if ratio_file else None,
                     "traffic_source": (f"profiles/{ratio_file} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate "
                                        "passes): HBM bytes over algorithmic bytes of the u32 passes at 2^30 pairs")
                     if ratio_file else None,
                     "launches": cnt,
                     "avg_launch_ms": (ms / cnt) if cnt else None,
                     "algorithmic_bytes_per_launch": (nbytes / cnt) if cnt else None,
                     "by_class": by_class,
                     "kernels": kernel_bounds(dict(stats, _bases=n), a.steps, pmc)},
        # SURVEY.md 8d end-to-end figure: 50 B per base of compulsory traffic over the whole pipeline
        "pipeline_hbm": {"algorithmic_bytes_per_base": ALG_BYTES_PER_BASE,
                         "achieved_GBps_per_gpu": ALG_BYTES_PER_BASE * n / step_s / 1e9,
                         "frac_of_peak_per_gpu": ALG_BYTES_PER_BASE * n / step_s / 1e9 / HBM_PEAK_GBS,
                         # what the whole step really moves: FETCH_SIZE + WRITE_SIZE summed over every kernel of one
                         # factorization (tools/pmc_step.sh, separate passes, gfx950 correction), per base
                         "traffic_bytes_per_base": ((pmc or {}).get("step") or {}).get("hbm_bytes_per_base")
                         if (pmc and pmc.get("bases") == n) else None,
                         "traffic_GBps_per_gpu": (pmc["step"]["hbm_bytes_per_base"] * n / step_s / 1e9)
                         if (pmc and pmc.get("bases") == n and pmc["step"].get("hbm_bytes_per_base")) else None,
                         "traffic_derived_from": f"profiles/{STEP_PMC_FILE}" if pmc else None},
        "per_rank": [{"rank": k, "device": f"cuda:{int(r[1])}", "bases": n, "factors": int(r[2]), "ms_per_step": r[0]}
                     for k, r in enumerate(rank_rows)],
        "pcie_inclusive_bases_per_s": n / pcie_dt,
        "stages_ms_per_step": {k: v[1] / a.steps for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1])
                               if k not in nested},
        "kernels_ms_per_step": {k: stats[k][1] / a.steps for k in sorted(nested) if k in stats},
    }
    return out, text, z


def cpu_baseline_rc(text: np.ndarray, sample: int):
    """The oracle in reverse-complement mode (prepare + nolzss_multiple_dna_w_rc restated, single thread)
    on a bounded prefix of the rc256m text."""
    import oracle_lib as oracle  # checker only: never on the measured path
    sample = min(sample, len(text))
    t0 = time.perf_counter()
    S, _, _ = oracle.prepare_multiple_dna_w_rc([text[:sample].tobytes()])
    z = oracle.count_factors_multiple_dna_w_rc(S)
    dt = time.perf_counter() - t0
    return {"value": sample / dt, "unit": "bases/s", "cores": 1, "kind": "port",
            "sample": f"first {sample} bases of the rank-0 rc256m text (+ their reverse complement), count_factors, "
                      f"{dt:.1f} s, z={z}"}


def run_rc(job: Job, a):
    """BASELINE config 5 on one GPU per rank: factorize_dna_w_rc of a 2^28-base text (generator of config 3, seed
    0x5EED0005 + rank) resident in HBM: prepared string T s0 rc(T) s1 built on the device, suffix array of its
    2^29 + 2 symbols, forward and reverse-complement candidates, selection, all records built in HBM.  RC mode does
    not shard: every rank factorizes its own genome, weak scaling."""
    n = 1 << a.rc_log2n
    text = gen.repeat_dna(n, seed=0x5EED0005 + job.rank)
    d_text = torch.from_numpy(text).to(job.dev)
    for _ in range(a.rc_warmup):
        native.factorize_dna_w_rc_device(d_text.data_ptr(), n, emit=1)
    native.profile_enable(True)
    native.profile_reset()
    job.barrier()
    t0 = time.perf_counter()
    for _ in range(a.rc_steps):
        z, _ = native.factorize_dna_w_rc_device(d_text.data_ptr(), n, emit=1)
    job.barrier()
    elapsed = time.perf_counter() - t0
    stats = native.profile_report()
    native.profile_enable(False)
    elapsed = job.max_over_ranks(elapsed)
    t1 = time.perf_counter()
    z2 = native.count_factors_dna_w_rc(text)  # host bytes in (upload + kernels), count out
    host_dt = time.perf_counter() - t1
    assert z2 == z
    del d_text
    torch.cuda.empty_cache()
    if job.rank != 0:
        return None, text
    step_s = elapsed / a.rc_steps
    nested = {"rs_hist", "rs_scan", "bucket_scatter", "window_scatter"} | {k for k in stats if in_family(k)}
    out = {"workload": f"rc256m: factorize_dna_w_rc of one 2^{a.rc_log2n}-base synthetic DNA text per GPU (sigma=4, 40% "
                       "copied blocks, 1% substitutions, seed 0x5EED0005 + rank) with its reverse-complement strand "
                       f"(|S| = 2^{a.rc_log2n + 1} + 2 symbols), text resident in HBM, records built in HBM",
           "scaling": "weak", "n_gpus": job.world, "steps": a.rc_steps, "warmup": a.rc_warmup,
           "value": float(n) * job.world / step_s, "unit": "bases/s", "ms_per_step": step_s * 1e3,
           "factors_per_sequence": int(z),
           "pipeline_hbm": {"algorithmic_bytes_per_base": ALG_BYTES_PER_BASE_RC,
                            "achieved_GBps_per_gpu": ALG_BYTES_PER_BASE_RC * n / step_s / 1e9,
                            "frac_of_peak_per_gpu": ALG_BYTES_PER_BASE_RC * n / step_s / 1e9 / HBM_PEAK_GBS},
           "count_factors_host_ms": host_dt * 1e3,
           "stages_ms_per_step": {k: v[1] / a.rc_steps for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1])
                                  if k not in nested},
           "kernels_ms_per_step": {k: stats[k][1] / a.rc_steps for k in sorted(nested) if k in stats}}
    return out, text


def stopwatches(text: np.ndarray, z: int, py_log2: int):
    """(i) C ABI host-in / host-out, (i') count_factors, (iii) Python-visible tuples;
    (ii) kernel-only is `value`.  Best of two calls each."""
    import noLZSS  # the reference's import path: pybind11 module -> C ABI
    n = len(text)
    buf = text.tobytes()
    # (best of two: the first host-buffer call re-reserves the arena -- the upload buffer comes on top of what the
    # device-resident steps reserved)
    t_count = t_fact = float("inf")
    for _ in range(2):
        t0 = time.perf_counter()
        zc = native.count_factors(buf)
        t_count = min(t_count, time.perf_counter() - t0)
    for _ in range(2):
        t0 = time.perf_counter()
        f = native.factorize_array(buf)
        t_fact = min(t_fact, time.perf_counter() - t0)
        assert zc == z == len(f)
        del f
    m = min(n, 1 << py_log2)
    prefix = buf[:m]
    t0 = time.perf_counter()
    tuples = noLZSS.factorize(prefix)
    t_py = time.perf_counter() - t0
    zt = len(tuples)
    del tuples
    return {"c_abi_host_ms": t_fact * 1e3, "c_abi_host_bases_per_s": n / t_fact,
            "count_factors_host_ms": t_count * 1e3, "count_factors_host_bases_per_s": n / t_count,
            "python_visible_ms": t_py * 1e3, "python_visible_bases": m, "python_visible_factors": zt,
            "python_visible_bases_per_s": m / t_py,
            "note": "c_abi_host: nolzss_factorize, host bytes in -> nolzss_factor[] on the host (upload + kernels + "
                    "download); count_factors_host: nolzss_count_factors from host bytes; python_visible: "
                    f"noLZSS.factorize(bytes) -> list of int tuples on the first {m} bases (the full text would "
                    "build 5*10^7 tuples)"}


def run_fasta_shard(job: Job, a, with_file: bool):
    """BASELINE config 4: `records` FASTA records of 2^rec_log2 bases, dealt over the ranks by the LPT
    plan of the library; per step every rank factorizes its records (resident in HBM) to factor records
    in HBM and the 512 counts are all-gathered.  Strong scaling."""
    m, L = a.fasta_records, 1 << a.fasta_record_log2
    owners = native.debug_lpt_plan([L] * m, job.world)
    mine = [j for j, o in enumerate(owners) if o == job.rank]
    with ThreadPoolExecutor(max_workers=min(8, max(1, (os.cpu_count() or 8) // max(1, job.world)))) as pool:
        recs = list(pool.map(lambda j: gen.random_dna(L, 0x4000 + j), mine))
    d_recs = [torch.from_numpy(r).to(job.dev) for r in recs]
    ptrs = [t.data_ptr() for t in d_recs]
    lens = [L] * len(mine)
    vec = torch.zeros(m, dtype=torch.int64, device=job.cdev)
    gathered = torch.zeros(job.world * m, dtype=torch.int64, device=job.cdev)
    idx = torch.tensor(mine, dtype=torch.int64, device=job.cdev)

    lib_seconds = [0.0]
    gather_seconds = [0.0]

    def step():
        t_lib = time.perf_counter()
        zs = native.factorize_batch_device(ptrs, lens, emit=1)
        lib_seconds[0] += time.perf_counter() - t_lib
        vec.zero_()
        if mine:
            vec[idx] = torch.tensor(zs, dtype=torch.int64, device=job.cdev)
        if job.collectives:
            # the one collective of the path, timed on its own (it includes the wait for the slowest rank)
            if job.backend == "nccl":
                torch.cuda.synchronize()
            t_g = time.perf_counter()
            dist.all_gather_into_tensor(gathered, vec)
            if job.backend == "nccl":
                torch.cuda.synchronize()
            gather_seconds[0] += time.perf_counter() - t_g
            return gathered.view(job.world, m).sum(dim=0)
        return vec.clone()

    for _ in range(a.fasta_warmup):
        step()
    job.barrier()
    lib_seconds[0] = 0.0
    gather_seconds[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(a.fasta_steps):
        counts = step()
    job.barrier()
    elapsed = job.max_over_ranks(time.perf_counter() - t0)
    counts = counts.cpu().tolist()
    # what every rank did, for rank 0's line: compute time of its share, time in the all-gather, device, records
    mine_row = torch.tensor([lib_seconds[0] / a.fasta_steps * 1e3, gather_seconds[0] / a.fasta_steps * 1e3,
                             float(job.local_rank), float(len(mine))], dtype=torch.float64, device=job.cdev)
    rows = torch.zeros(job.world * 4, dtype=torch.float64, device=job.cdev)
    if job.collectives:
        dist.all_gather_into_tensor(rows, mine_row)
    else:
        rows[:4] = mine_row
    rows = rows.view(job.world, 4).cpu().tolist()
    step_s = elapsed / a.fasta_steps
    total = float(m) * L
    out = {"workload": f"fasta512: {m} records x 2^{a.fasta_record_log2} random ACGT bases (seeds 0x4000+k), LPT shard "
                       f"over {job.world} rank(s), records resident in HBM, factor records built in HBM, one all-gather "
                       f"of {m} counts",
           "scaling": "strong", "n_gpus": job.world, "steps": a.fasta_steps, "warmup": a.fasta_warmup,
           "value": total / step_s, "unit": "bases/s", "ms_per_step": step_s * 1e3,
           "library_call_ms_per_step": lib_seconds[0] / a.fasta_steps * 1e3,
           "records_per_rank": [owners.count(r) for r in range(job.world)],
           "per_rank_ms": [r[0] for r in rows],          # compute of the rank's share per step (the library call)
           "allgather_ms": [r[1] for r in rows],         # the all-gather of the counts per step, as each rank saw it
           "per_rank": [{"rank": k, "device": f"cuda:{int(r[2])}", "records": int(r[3]), "compute_ms_per_step": r[0],
                         "allgather_ms_per_step": r[1]} for k, r in enumerate(rows)],
           "total_factors": int(sum(counts)),
           "pipeline_hbm_frac_per_gpu": ALG_BYTES_PER_BASE * (total / job.world) / step_s / 1e9 / HBM_PEAK_GBS}
    # PCIe-inclusive: the host-buffer batch entry point (nolzss_factorize_batch), counts only
    job.barrier()
    t0 = time.perf_counter()
    zs_host, _ = native.factorize_batch(recs, want_factors=False)
    job.barrier()
    dt = job.max_over_ranks(time.perf_counter() - t0)
    assert zs_host == [counts[j] for j in mine]
    out["host_buffers_to_counts_bases_per_s"] = total / dt
    del d_recs
    torch.cuda.empty_cache()
    if with_file:
        # file -> native reader -> batch -> all-gather, through the reference's caller
        from nolzss_amd.genomics.fasta import shard_nucleotide_fasta
        tmp = Path("/dev/shm") if Path("/dev/shm").is_dir() else Path(tempfile.gettempdir())
        path = tmp / f"nolzss_bench_fasta_{os.environ.get('MASTER_PORT', 'single')}.fa"
        if job.rank == 0:
            with ThreadPoolExecutor(max_workers=8) as pool:
                allrecs = list(pool.map(lambda j: gen.random_dna(L, 0x4000 + j), range(m)))
            gen.write_fasta_fast(path, [(f"seq{j}", r) for j, r in enumerate(allrecs)])
            del allrecs
        job.barrier()
        try:
            t0 = time.perf_counter()
            ids, fcounts, local = shard_nucleotide_fasta(path, want_factors=False)
            job.barrier()
            dt = job.max_over_ranks(time.perf_counter() - t0)
            assert fcounts == counts and ids[0] == "seq0" and len(ids) == m
            out["file_to_counts_s"] = dt
            out["file_to_counts_bases_per_s"] = total / dt
            out["file_bytes"] = path.stat().st_size
        finally:
            job.barrier()
            if job.rank == 0:
                path.unlink(missing_ok=True)
    return out if job.rank == 0 else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="dna1g", choices=["dna1g", "random", "fasta512"])
    ap.add_argument("--log2n", type=int, default=30, help="bases per GPU = 2^log2n (default 2^30)")
    ap.add_argument("--cpu-sample-log2", type=int, default=26)
    ap.add_argument("--python-sample-log2", type=int, default=26)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stopwatches", action="store_true")
    ap.add_argument("--no-fasta", action="store_true", help="skip the fasta512 side measurement")
    ap.add_argument("--fasta-records", type=int, default=512)
    ap.add_argument("--fasta-record-log2", type=int, default=22)
    ap.add_argument("--fasta-steps", type=int, default=2)
    ap.add_argument("--fasta-warmup", type=int, default=1)
    ap.add_argument("--no-rc", action="store_true", help="skip the rc256m side measurement (BASELINE config 5)")
    ap.add_argument("--rc-log2n", type=int, default=28)
    ap.add_argument("--rc-steps", type=int, default=3)
    ap.add_argument("--rc-warmup", type=int, default=1)
    ap.add_argument("--rc-cpu-sample-log2", type=int, default=24)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend (nccl = RCCL over xGMI; gloo only for rehearsals)")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="with one rank: open the process group and issue every collective of the multi-rank path anyway")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: the N ranks are child processes of this one, which stays off the GPU
        sys.exit(self_launch(a, sys.argv[1:]))
    late_imports()
    job = Job(a)

    if a.workload == "fasta512":
        a.fasta_steps, a.fasta_warmup = a.steps, a.warmup
        fa = run_fasta_shard(job, a, with_file=True)
        if job.rank == 0 and job.world == 1 and not a.no_cpu_baseline:
            fa["cpu_baseline"] = cpu_baseline_fasta(a)
        if job.rank == 0:
            out = {"metric": "bases/sec factorized (multi-sequence FASTA shard, 512 x 4 Mi bases) + HBM GB/s fraction",
                   "value": fa["value"], "unit": "bases/s", "n_gpus": job.world, "steps": a.steps, "warmup": a.warmup,
                   "ms_per_step": fa["ms_per_step"], "higher_is_better": True, "scaling": "strong",
                   "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                   "config": {"workload": fa["workload"], "parallelism": f"{job.world} rank(s), LPT plan, all-gather of counts"},
                   "fasta512": fa}
            print(json.dumps(out), flush=True)
    else:
        out, text, z = run_single_sequence(job, a)
        fa = None if a.no_fasta else run_fasta_shard(job, a, with_file=False)
        rc, rc_text = (None, None) if a.no_rc else run_rc(job, a)
        if job.rank == 0:
            if job.collectives:
                out["collectives"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                      "forced_on_one_rank": bool(a.force_collectives and job.world == 1)}
            if fa is not None:
                out["fasta512"] = fa
            if rc is not None:
                out["rc256m"] = rc
            if job.world == 1 and not a.no_stopwatches:  # the host-side clocks run on the one-GPU run only
                out["stopwatches"] = stopwatches(text, z, a.python_sample_log2)
            if job.world == 1 and not a.no_cpu_baseline:  # the CPU legs run on rank 0 of the one-GPU run only
                out["cpu_baseline"] = cpu_baseline(text, 1 << a.cpu_sample_log2)
                if fa is not None:
                    fa["cpu_baseline"] = cpu_baseline_fasta(a)
                if rc is not None:
                    rc["cpu_baseline"] = cpu_baseline_rc(rc_text, 1 << a.rc_cpu_sample_log2)
            print(json.dumps(out), flush=True)
    if job.collectives:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
