"""Stage-time probe: python tools/probe.py <kind> <log2 n or n> [--check] [--factors]
Prints the HIP-event stage breakdown of one factorization with the text resident in HBM."""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import gen  # noqa: E402
from nolzss_amd import _noLZSS as native  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kind", choices=["random", "repeat"])
    ap.add_argument("n", type=str)
    ap.add_argument("--check", action="store_true", help="compare factor count with the oracle")
    ap.add_argument("--factors", action="store_true", help="also emit and download the factors")
    ap.add_argument("--reps", type=int, default=2)
    a = ap.parse_args()
    n = int(a.n) if not a.n.startswith("2^") else 1 << int(a.n[2:])
    t0 = time.time()
    text = gen.random_dna(n) if a.kind == "random" else gen.repeat_dna(n)
    print(f"generated {n} bases in {time.time()-t0:.1f}s", flush=True)
    d = torch.from_numpy(text).cuda()
    torch.cuda.synchronize()
    for rep in range(a.reps):
        native.profile_enable(True)
        native.profile_reset()
        t0 = time.time()
        z, f = native.factorize_device(d.data_ptr(), n, emit=2 if a.factors else 1)
        dt = time.time() - t0
        rep_stats = native.profile_report()
        print(f"rep {rep}: z={z} wall={dt*1e3:.1f} ms  {n/dt/1e6:.1f} Mbases/s", flush=True)
    tot = 0.0
    nested = {"rs_hist", "rs_scan", "bucket_scatter", "window_scatter", "rs_local_sort"}
    for name, (cnt, ms, nbytes) in sorted(rep_stats.items(), key=lambda kv: -kv[1][1]):
        extra = f"  {nbytes/ms/1e6:8.1f} GB/s" if nbytes else ""
        print(f"  {name:18s} x{cnt:4d} {ms:10.3f} ms{extra}")
        if name not in nested and not name.startswith("rs_scatter"):
            tot += ms
    print(f"  sum of top-level stages {tot:.1f} ms")
    cap, peak = native.debug_arena()
    print(f"  arena: capacity {cap/2**30:.2f} GiB, peak {peak/2**30:.2f} GiB = {peak/n:.1f} bytes/base")
    native.profile_enable(False)
    if a.check:
        import oracle_lib as oracle
        t0 = time.time()
        zo = oracle.count_factors(text)
        print(f"oracle z={zo} ({time.time()-t0:.1f}s) match={zo == z}")


if __name__ == "__main__":
    main()
