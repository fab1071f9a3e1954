"""Radix-sort micro-benchmark: python tools/sortbench.py [log2n] -- device sort of random
(u64, u32) pairs through nolzss_debug_sort_pairs with the stage profiler on."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from nolzss_amd import _noLZSS as native  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 27
n = 1 << log2n
rng = np.random.default_rng(1)
keys = rng.integers(0, 1 << 63, size=n, dtype=np.uint64)
vals = np.arange(n, dtype=np.uint32)
native.debug_sort_pairs(keys[:1 << 20], vals[:1 << 20])
native.profile_enable(True)
for rep in range(2):
    native.profile_reset()
    t0 = time.time()
    k2, v2 = native.debug_sort_pairs(keys, vals)
    dt = time.time() - t0
    st = native.profile_report()
    print(f"rep {rep}: n=2^{log2n} wall {dt*1e3:.1f} ms (includes H2D/D2H)")
    for name, (cnt, ms, nbytes) in sorted(st.items()):
        print(f"  {name:12s} x{cnt:3d} {ms:9.3f} ms  avg {ms/cnt*1e3:8.1f} us  {nbytes/ms/1e6:8.1f} GB/s")
assert np.all(k2[:-1] <= k2[1:])
print("sorted ok")
