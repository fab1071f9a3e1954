"""Stage times of one merged batch run (HIP events) beside the wall time of the call."""
import sys, time
from pathlib import Path
R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / 'tests'))
import gen
from nolzss_amd import _noLZSS as native
m, lg = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 14)
recs = [s for _, s in gen.fasta_records(m, 1 << lg)]
native.factorize_batch(recs, want_factors=False)
native.profile_enable(True)
for want in (False, True):
    native.profile_reset()
    t0 = time.time(); counts, arr = native.factorize_batch(recs, want_factors=want); dt = time.time() - t0
    st = native.profile_report()
    nested = ("rs_", "bucket_scatter", "window_scatter")
    top = {k: v for k, v in st.items() if not k.startswith(nested)}
    print(f"want_factors={want}: wall {dt*1e3:.1f} ms, sum of stages {sum(v[1] for v in top.values()):.1f} ms, z={sum(counts)}")
    for k, v in sorted(top.items(), key=lambda kv: -kv[1][1]):
        print(f"   {k:24s} {v[0]:4d} x  {v[1]:8.2f} ms")
