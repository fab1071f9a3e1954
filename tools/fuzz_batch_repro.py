"""Re-run ONE case of tools/fuzz_batch.py: python tools/fuzz_batch_repro.py SEED CASE  (NOLZSS_LIB selects the build)"""
import sys
from pathlib import Path
R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / 'tests')); sys.path.insert(0, str(R / 'tools'))
import numpy as np
import oracle_lib as oracle
import fuzz_batch as fb
from nolzss_amd import _noLZSS as native

seed, case = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for _ in range(case + 1):
    recs = fb.make_records(rng)
print("records", len(recs), "bases", sum(len(r) for r in recs))
for with_rc in (False, True):
    counts, arrays = native.factorize_batch(recs, want_factors=True, with_rc=with_rc)
    bad = []
    for j, r in enumerate(recs):
        exp = fb.rc_expected(r) if with_rc else oracle.factors_array(r)
        if not fb.same(arrays[j], exp):
            bad.append(j)
    print("rc" if with_rc else "plain", "mismatching records:", bad[:20], "of", len(recs))
    if bad:
        j = bad[0]
        exp = fb.rc_expected(recs[j]) if with_rc else oracle.factors_array(recs[j])
        got = arrays[j]
        k = 0
        while k < min(len(got), len(exp)) and all(got[c][k] == exp[c][k] for c in ("start", "length", "ref")):
            k += 1
        print(" record", j, "len", len(recs[j]), "first difference at factor", k, "got", [tuple(int(x) for x in got[q]) for q in range(max(0, k - 1), min(len(got), k + 3))],
              "exp", [tuple(int(exp[c][q]) for c in ("start", "length", "ref")) for q in range(max(0, k - 1), min(len(exp), k + 3))])
