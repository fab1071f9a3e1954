"""Round-by-round trace and stage times of one degenerate text: python tools/degenerate_trace.py <log2 n> <case>
case: fib | copiesK (K genomes 0.1 % apart) | run | period1000   (set NOLZSS_TRACE=1 for the rounds)"""
import sys, time
from pathlib import Path
R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / 'tests'))
import numpy as np
import gen
from nolzss_amd import _noLZSS as native
lg, case = int(sys.argv[1]), sys.argv[2]
n = 1 << lg


def mutated(x, seed, rate=1000):
    y = x.copy(); r = np.random.default_rng(seed)
    idx = r.integers(0, len(y), size=len(y) // rate)
    y[idx] = np.frombuffer(b"ACGT", dtype=np.uint8)[r.integers(0, 4, size=len(idx))]
    return y


if case == "fib":
    a, b = b"A", b"AC"
    while len(b) < n:
        a, b = b, b + a
    t = np.frombuffer(b[:n], dtype=np.uint8)
elif case.startswith("copies"):
    k = int(case[6:])
    base = gen.random_dna(n // k, 40 + k)
    t = np.concatenate([base] + [mutated(base, 100 * k + j) for j in range(k - 1)])
elif case == "run":
    t = np.full(n, ord("A"), dtype=np.uint8)
else:
    t = np.tile(gen.random_dna(1000, 1), n // 1000 + 1)[:n]
native.count_factors(gen.random_dna(1 << 16, 2))
native.profile_enable(True)
native.profile_reset()
t0 = time.time(); z = native.count_factors(t); dt = time.time() - t0
print(f"2^{lg} {case}: {dt*1e3:.1f} ms, z={z}")
for name, (cnt, ms, nbytes) in sorted(native.profile_report().items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"  {name:22s} x{cnt:4d} {ms:10.3f} ms")
