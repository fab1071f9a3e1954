import sys, time
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent / 'tests'))
import gen
from nolzss_amd import _noLZSS as native
m, ln = 64, 1 << 22
recs = [s for _, s in gen.fasta_records(m, ln)]
native.factorize_batch(recs[:2], want_factors=False)
for rep in range(3):
    t0 = time.time(); counts, _ = native.factorize_batch(recs, want_factors=False); dt = time.time() - t0
    print(f"no-profile rep {rep}: wall={dt*1e3:.1f} ms  {m*ln/dt/1e6:.1f} Mbases/s")
