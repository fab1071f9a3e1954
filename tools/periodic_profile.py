"""Stage profile of the pipeline on periodic texts: python tools/periodic_profile.py [log2 n] [case ...]"""
import sys, time
from pathlib import Path
R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / 'tests'))
import numpy as np, torch
import gen
from nolzss_amd import _noLZSS as native
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << lg
want = sys.argv[2:]


def fib(n):
    a, b = b"A", b"AC"
    while len(b) < n:
        a, b = b, b + a
    return np.frombuffer(b[:n], dtype=np.uint8)


cases = {
    "run": np.full(n, ord("A"), dtype=np.uint8),
    "ac": np.tile(np.frombuffer(b"AC", dtype=np.uint8), n // 2),
    "p1000": np.tile(gen.random_dna(1000, 1), n // 1000 + 1)[:n],
    "fib": fib(n),
}
native.count_factors(gen.random_dna(1 << 16, 2))
for name, t in cases.items():
    if want and name not in want:
        continue
    d = torch.from_numpy(t).cuda()
    torch.cuda.synchronize()
    native.profile_enable(True)
    native.profile_reset()
    t0 = time.time()
    z, _ = native.factorize_device(d.data_ptr(), n, emit=1)
    dt = time.time() - t0
    st = native.profile_report()
    native.profile_enable(False)
    print(f"== {name} 2^{lg}: {dt*1e3:.1f} ms, z={z}", flush=True)
    for k, (cnt, ms, _) in sorted(st.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"   {k:22s} x{cnt:5d} {ms:10.2f} ms")
