"""Radix-sort micro-benchmark: python tools/sortbench.py [log2n] [random|sorted|onebin] -- device
sort of (u64, u32) pairs through nolzss_debug_sort_pairs with the stage profiler on.
"sorted" input makes every pass a near-identity permutation (long runs per bin: the streaming
ceiling of the scatter kernel), "onebin" puts all keys into one bin per pass."""
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
mode = sys.argv[2] if len(sys.argv) > 2 else "random"
keys = rng.integers(0, 1 << 63, size=n, dtype=np.uint64)
if mode == "sorted":
    keys = np.arange(n, dtype=np.uint64) << np.uint64(64 - log2n - 1)
elif mode == "onebin":
    keys = np.zeros(n, dtype=np.uint64) + np.uint64(0x0101010101010101)
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
