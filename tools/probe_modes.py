"""Stage-time probe for the other two configurations:
   python tools/probe_modes.py rc <log2 n>          reverse-complement mode, one sequence
   python tools/probe_modes.py batch <m> <log2 len> m independent sequences (FASTA shard unit)"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import gen  # noqa: E402
from nolzss_amd import _noLZSS as native  # noqa: E402


def report(stats):
    nested = {"rs_hist", "rs_scan", "bucket_scatter", "window_scatter"}
    tot = 0.0
    for name, (cnt, ms, nbytes) in sorted(stats.items(), key=lambda kv: -kv[1][1])[:18]:
        print(f"  {name:20s} x{cnt:5d} {ms:10.3f} ms")
    for name, (cnt, ms, nbytes) in stats.items():
        if name not in nested and not name.startswith("rs_scatter"):
            tot += ms
    print(f"  sum of top-level stages {tot:.1f} ms")


mode = sys.argv[1]
if mode == "rc":
    n = 1 << int(sys.argv[2])
    text = gen.repeat_dna(n, seed=0x5EED0005)
    native.count_factors_dna_w_rc(text[:1 << 20])
    for rep in range(2):
        native.profile_enable(True)
        native.profile_reset()
        t0 = time.time()
        z = native.count_factors_dna_w_rc(text)
        dt = time.time() - t0
        st = native.profile_report()
        print(f"rep {rep}: RC n={n} z={z} wall={dt*1e3:.1f} ms  {n/dt/1e6:.1f} Mbases/s (host buffer, count only)")
    report(st)
else:
    m, ln = int(sys.argv[2]), 1 << int(sys.argv[3])
    recs = [s for _, s in gen.fasta_records(m, ln)]
    native.factorize_batch(recs[:2], want_factors=False)
    for rep in range(2):
        native.profile_enable(True)
        native.profile_reset()
        t0 = time.time()
        counts, _ = native.factorize_batch(recs, want_factors=False)
        dt = time.time() - t0
        st = native.profile_report()
        print(f"rep {rep}: batch {m} x {ln}: wall={dt*1e3:.1f} ms  {m*ln/dt/1e6:.1f} Mbases/s  z[0]={counts[0]}")
    report(st)
