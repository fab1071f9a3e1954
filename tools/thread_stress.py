"""Concurrent callers: python tools/thread_stress.py -- 8 host threads factorize different texts at
the same time (the library releases the GIL and hands every call its own lane); every result is
compared with the oracle."""
import sys, threading
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import gen, oracle_lib as oracle
from nolzss_amd import _noLZSS as native

texts = [gen.repeat_dna(600_000 + 37_000 * k, seed=900 + k, lo=16, hi=4096) for k in range(8)]
texts += [gen.random_dna(1_200_000 + 11 * k, seed=800 + k) for k in range(4)]
expected = [oracle.count_factors(t) for t in texts]
errors = []

def worker(tid):
    try:
        for rep in range(6):
            for k in range(len(texts)):
                j = (k + tid) % len(texts)
                z = native.count_factors(texts[j])
                if z != expected[j]:
                    errors.append((tid, rep, j, z, expected[j]))
    except Exception as e:  # noqa: BLE001
        errors.append((tid, repr(e)))

threads = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
[t.start() for t in threads]
[t.join() for t in threads]
print("errors:", errors[:5], "total", len(errors))
sys.exit(1 if errors else 0)
