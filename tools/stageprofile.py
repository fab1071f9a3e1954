"""Stage times (HIP events) of one count_factors call on a chosen degenerate text."""
import sys, time
from pathlib import Path
R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / 'tests'))
import numpy as np
import gen
from nolzss_amd import _noLZSS as native
kind = sys.argv[1] if len(sys.argv) > 1 else "copies"
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 28
n = 1 << lg
if kind == "copies":
    half = gen.random_dna(n // 2, 3); t = np.concatenate([half, half])
elif kind == "mutated":   # second copy with 0.1 % substitutions
    half = gen.random_dna(n // 2, 3); c = half.copy()
    rng = np.random.default_rng(1); idx = rng.integers(0, n // 2, size=n // 2000)
    c[idx] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=len(idx))]
    t = np.concatenate([half, c])
elif kind == "copies3":   # three genomes, the second and third 0.1 % away from the first
    third = gen.random_dna(n // 3, 3); rng = np.random.default_rng(1); parts = [third]
    for _ in range(2):
        c = third.copy(); idx = rng.integers(0, len(c), size=len(c) // 1000)
        c[idx] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=len(idx))]; parts.append(c)
    t = np.concatenate(parts)
elif kind == "run":
    t = np.full(n, ord("A"), dtype=np.uint8)
else:
    t = gen.random_dna(n, 2)
native.count_factors(t[:1 << 16])
native.profile_enable(True); native.profile_reset()
t0 = time.time(); z = native.count_factors(t); dt = time.time() - t0
st = native.profile_report()
nested = ("rs_", "bucket_scatter", "window_scatter", "runs_")
for k, v in sorted(st.items()):
    if k.startswith("runs_"):
        print(f"      {k:20s} {v[0]:4d} x {v[1]:8.1f} ms")
print(f"{kind} 2^{lg}: {dt*1e3:.1f} ms, z={z}")
for k, v in sorted(st.items(), key=lambda kv: -kv[1][1]):
    if not k.startswith(nested) and v[1] > 0.5:
        print(f"   {k:22s} {v[0]:4d} x {v[1]:8.1f} ms")
