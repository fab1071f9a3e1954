"""Wall time of the per-sequence FASTA entry points on a file of short records (reader + device)."""
import sys, time, tempfile, os
from pathlib import Path
R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / 'tests'))
import gen
from nolzss_amd import _noLZSS as native
m, lg = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 14)
recs = gen.fasta_records(m, 1 << lg)
path = os.path.join(tempfile.mkdtemp(), "records.fa")
gen.write_fasta(path, recs)
size = os.path.getsize(path)
native.count_factors_fasta_dna_no_rc_per_sequence(path)
for name, fn in (("count, no rc", native.count_factors_fasta_dna_no_rc_per_sequence),
                 ("count, with rc", native.count_factors_fasta_dna_w_rc_per_sequence)):
    best = 1e9
    for rep in range(3):
        t0 = time.time(); r = fn(path); best = min(best, time.time() - t0)
    print(f"{m} x 2^{lg} ({size/1e6:.0f} MB file) {name}: {best*1e3:.1f} ms = {m*(1<<lg)/best/1e6:.0f} Mbases/s", flush=True)
t0 = time.time(); native.debug_parse_fasta(path); print(f"reader alone incl. python copies: {(time.time()-t0)*1e3:.1f} ms")
