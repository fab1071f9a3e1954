"""fasta512 step time of the library in use, several repetitions: python tools/fasta512_ab.py [reps]
(run once per build: NOLZSS_LIB=... selects the build)"""
import sys, time
from pathlib import Path
R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / 'tests'))
import torch
import gen
from nolzss_amd import _noLZSS as native
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
m, L = 512, 1 << 22
d = [torch.from_numpy(gen.random_dna(L, 0x4000 + j)).cuda() for j in range(m)]
ptrs = [t.data_ptr() for t in d]
lens = [L] * m
native.factorize_batch_device(ptrs, lens, emit=1)
ts = []
for _ in range(reps):
    torch.cuda.synchronize()
    t0 = time.time(); zs = native.factorize_batch_device(ptrs, lens, emit=1); ts.append((time.time() - t0) * 1e3)
print("fasta512 ms per step:", " ".join(f"{t:.1f}" for t in ts), " z =", sum(zs))
