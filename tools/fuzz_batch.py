"""Differential fuzzing of the merged batch (plain and reverse-complement) against the oracle.
usage: python tools/fuzz_batch.py SECONDS [SEED]"""
import sys, time
from pathlib import Path
R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / 'tests'))
import numpy as np
import gen
import oracle_lib as oracle
from nolzss_amd import _noLZSS as native

COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGT", b"TGCA"):
    COMP[a] = b


def same(a, b):
    return len(a) == len(b) and all(np.array_equal(a[k], b[k]) for k in ("start", "length", "ref"))


def rc_expected(rec):
    S, _, _ = oracle.prepare_multiple_dna_w_rc([bytes(rec)])
    return oracle.factors_array_multiple_dna_w_rc(S)


def make_records(rng):
    sigma = int(rng.integers(1, 5))
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.permutation(4)[:sigma]]
    m = int(rng.integers(2, 700))
    hi = int(rng.choice([8, 40, 300, 3000]))
    recs = []
    for _ in range(m):
        kind = int(rng.integers(0, 7))
        n = int(rng.integers(1, hi + 1))
        if kind == 0 and recs:      # prefix / copy of an earlier record
            src = recs[int(rng.integers(0, len(recs)))]
            r = src[:int(rng.integers(1, len(src) + 1))].copy()
        elif kind == 1:             # periodic
            unit = letters[rng.integers(0, sigma, size=int(rng.integers(1, 7)))]
            r = np.tile(unit, n)[:n]
        elif kind == 2:             # second half = reverse complement of the first
            half = letters[rng.integers(0, sigma, size=max(1, n // 2))]
            r = np.concatenate([half, COMP[half[::-1]]])
        elif kind == 3 and recs:    # reverse complement of an earlier record
            r = COMP[recs[int(rng.integers(0, len(recs)))][::-1]].copy()
        elif kind == 4 and n >= 64:  # repeats with substitutions
            r = gen.repeat_dna(n, seed=int(rng.integers(1 << 30)))
        else:
            r = letters[rng.integers(0, sigma, size=n)]
        recs.append(np.ascontiguousarray(r, dtype=np.uint8))
    return recs


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t_end = time.time() + seconds
    cases = records = 0
    last = time.time()
    while time.time() < t_end:
        recs = make_records(rng)
        for with_rc in (False, True):
            counts, arrays = native.factorize_batch(recs, want_factors=True, with_rc=with_rc)
            for j, r in enumerate(recs):
                exp = rc_expected(r) if with_rc else oracle.factors_array(r)
                if not same(arrays[j], exp):
                    print("MISMATCH", "rc" if with_rc else "plain", "case", cases, "record", j, bytes(r)[:200])
                    print("batch:", [bytes(x) for x in recs][:50])
                    sys.exit(1)
        cases += 1
        records += len(recs)
        if time.time() - last > 30:
            print(f"{cases} batches, {records} records, no mismatch", flush=True)
            last = time.time()
    merged, single = native.debug_batch_counters()
    print(f"done: {cases} batches, {records} records (x2 modes), merged runs took {merged} records, one-by-one {single}; no mismatch")


if __name__ == "__main__":
    main()
