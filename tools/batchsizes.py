"""Throughput of the batch entry point as a function of the record size: merged runs (default)
against one pipeline run per record (NOLZSS_BATCH_MERGE_BELOW=0)."""
import os, sys, time
from pathlib import Path
R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / 'tests'))
import gen
from nolzss_amd import _noLZSS as native
rc = "--rc" in sys.argv  # every record with its reverse complement (factorize_dna_w_rc per record)
native.factorize_batch([s for _, s in gen.fasta_records(2, 1 << 16)], want_factors=False, with_rc=rc)
for m, lg in ((16384, 10), (4096, 12), (4096, 14), (1024, 16), (256, 18), (64, 20), (16, 22)):
    recs = [s for _, s in gen.fasta_records(m, 1 << lg)]
    line = f"{m} x 2^{lg}:"
    for mode, env in (("merged", None), ("one by one", "0")):
        if env is None:
            os.environ.pop("NOLZSS_BATCH_MERGE_BELOW", None)
        else:
            os.environ["NOLZSS_BATCH_MERGE_BELOW"] = env
        for want in (False, True):
            best = 1e9
            for rep in range(2):
                t0 = time.time(); counts, _ = native.factorize_batch(recs, want_factors=want, with_rc=rc); best = min(best, time.time() - t0)
            line += f"  {mode}{' +factors' if want else ''}: {best*1e3:.1f} ms = {m*(1<<lg)/best/1e6:.0f} Mbases/s"
    print(line, flush=True)
