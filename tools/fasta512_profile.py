"""Stage times (HIP events) of one device-resident batch step: records x 2^log2 random bases."""
import sys, time
from pathlib import Path
R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / 'tests'))
import torch
import gen
from nolzss_amd import _noLZSS as native
m, lg = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 22)
L = 1 << lg
d = [torch.from_numpy(gen.random_dna(L, 0x4000 + j)).cuda() for j in range(m)]
ptrs = [t.data_ptr() for t in d]
lens = [L] * m
native.factorize_batch_device(ptrs, lens, emit=1)
native.profile_enable(True)
for emit in (0, 1):
    native.profile_reset()
    torch.cuda.synchronize()
    t0 = time.time(); zs = native.factorize_batch_device(ptrs, lens, emit=emit); dt = time.time() - t0
    st = native.profile_report()
    nested = ("rs_", "bucket_scatter", "window_scatter")
    top = {k: v for k, v in st.items() if not k.startswith(nested)}
    print(f"emit={emit}: wall {dt*1e3:.1f} ms = {m*L/dt/1e9:.2f} Gbases/s, sum of stages {sum(v[1] for v in top.values()):.1f} ms, z={sum(zs)}")
    for k, v in sorted(st.items(), key=lambda kv: -kv[1][1]):
        print(f"   {k:24s} {v[0]:4d} x  {v[1]:8.2f} ms")
