"""Differential fuzzing against the oracle: python tools/fuzz.py [seconds] [seed] [log10 max size]
Random texts of many shapes (DNA with copies / tandem repeats / runs, small and large alphabets,
prepared multi-sequence strings with and without reverse complement), plain and RC entry points."""
import random
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib as oracle  # noqa: E402
from nolzss_amd import _noLZSS as native  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
max_log = float(sys.argv[3]) if len(sys.argv) > 3 else 5.5
min_log = 0.0 if max_log <= 5.5 else max_log - 1.3
rng = random.Random(seed)


def dna(n, alphabet="ACGT"):
    kind = rng.randrange(6)
    if kind == 0:
        return "".join(rng.choice(alphabet) for _ in range(n))
    if kind == 1:  # tandem repeats of random period
        out = []
        while len(out) < n:
            unit = [rng.choice(alphabet) for _ in range(rng.randint(1, 40))]
            out += unit * rng.randint(1, 200)
        return "".join(out[:n])
    if kind == 2:  # copies with edits
        out = [rng.choice(alphabet) for _ in range(min(n, rng.randint(10, 2000)))]
        while len(out) < n:
            if rng.random() < 0.6:
                s = rng.randrange(len(out)); l = rng.randint(1, 3000)
                chunk = out[s:s + l]
                for _ in range(len(chunk) // rng.choice([20, 100, 1000000]) ):
                    chunk[rng.randrange(len(chunk))] = rng.choice(alphabet)
                out += chunk
            else:
                out += [rng.choice(alphabet) for _ in range(rng.randint(1, 500))]
        return "".join(out[:n])
    if kind == 3:  # long runs
        out = []
        while len(out) < n:
            out += [rng.choice(alphabet)] * rng.randint(1, 5000)
        return "".join(out[:n])
    if kind == 4:  # low-complexity two-letter
        a, b = rng.sample(alphabet, 2) if len(alphabet) > 1 else (alphabet[0], alphabet[0])
        return "".join(rng.choice((a, b)) for _ in range(n))
    x = "".join(rng.choice(alphabet) for _ in range(max(1, n // 2)))  # X X' with few edits
    y = list(x)
    for _ in range(rng.randint(0, 5)):
        y[rng.randrange(len(y))] = rng.choice(alphabet)
    return (x + "".join(y))[:n]


t_end = time.time() + budget
cases = 0
while time.time() < t_end:
    mode = rng.randrange(10)
    n = int(10 ** rng.uniform(min_log, max_log))
    if mode <= 3:  # plain DNA
        t = dna(n).encode()
        assert native.factorize(t) == oracle.factorize(t), ("plain dna", seed, cases, n)
    elif mode == 4:  # other alphabets
        alpha = rng.choice(["ab", "abc", "ACDEFGHIKLMNPQRSTVWY", "".join(chr(c) for c in range(33, 127))])
        t = dna(n, alpha).encode()
        assert native.factorize(t) == oracle.factorize(t), ("plain", alpha[:5], seed, cases, n)
    elif mode == 5:  # arbitrary bytes
        t = bytes(rng.randrange(1, 256) for _ in range(min(n, 50_000)))
        assert native.factorize(t) == oracle.factorize(t), ("bytes", seed, cases, n)
    elif mode <= 7:  # single sequence with reverse complement
        t = dna(min(n, 200_000 if max_log <= 5.5 else 3_000_000)).encode()
        assert native.factorize_dna_w_rc(t) == oracle.factorize_dna_w_rc(t), ("rc", seed, cases, len(t))
    else:  # prepared multi-sequence strings
        k = rng.randint(1, 125 if mode == 8 else 250)
        tail = dna(rng.randint(1, 16))
        seqs = [dna(rng.randint(1, max(1, min(3000, n // k + 1)))) + (tail[-rng.randint(1, len(tail)):] if rng.random() < 0.5 else "")
                for _ in range(k)]
        if mode == 8:
            S, _, _ = native.prepare_multiple_dna_sequences_w_rc_bytes(seqs)
            assert native.factorize_multiple_dna_w_rc(S) == oracle.factorize_multiple_dna_w_rc(S), ("multi rc", seed, cases, k)
        else:
            S, _, _ = native.prepare_multiple_dna_sequences_no_rc_bytes(seqs)
            assert native.factorize(S) == oracle.factorize(S), ("multi", seed, cases, k)
    cases += 1
    if cases % (200 if max_log <= 5.5 else 5) == 0:
        print(f"{cases} cases ok", flush=True)
print(f"done: {cases} cases, seed {seed}, no mismatch")
