"""Wall time on degenerate texts (one run, short periods, Fibonacci words, two copies of a random text)."""
import sys, time
from pathlib import Path
R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / 'tests'))
import numpy as np
import gen
from nolzss_amd import _noLZSS as native
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << lg


def fib(n):
    a, b = b"A", b"AC"
    while len(b) < n:
        a, b = b, b + a
    return np.frombuffer(b[:n], dtype=np.uint8)


half = gen.random_dna(n // 2, 3)


def mutated(x, seed, rate=1000):
    y = x.copy(); r = np.random.default_rng(seed)
    idx = r.integers(0, len(y), size=len(y) // rate)
    y[idx] = np.frombuffer(b"ACGT", dtype=np.uint8)[r.integers(0, 4, size=len(idx))]
    return y


third = gen.random_dna(n // 3, 4)
fifth = gen.random_dna(n // 5, 5)
cases = {
    "one run A^n": np.full(n, ord("A"), dtype=np.uint8),
    "(AC)^(n/2)": np.tile(np.frombuffer(b"AC", dtype=np.uint8), n // 2),
    "period 1000": np.tile(gen.random_dna(1000, 1), n // 1000 + 1)[:n],
    "Fibonacci word": fib(n),
    "two copies of a random text": np.concatenate([half, half]),
    "two genomes 0.1 % apart": np.concatenate([half, mutated(half, 1)]),
    "three genomes 0.1 % apart": np.concatenate([third, mutated(third, 2), mutated(third, 3)]),
    "five genomes 0.1 % apart": np.concatenate([fifth] + [mutated(fifth, 10 + k) for k in range(4)]),
    "twelve genomes 0.1 % apart": np.concatenate([gen.random_dna(n // 12, 6)] + [mutated(gen.random_dna(n // 12, 6), 30 + k) for k in range(11)]),
    "random": gen.random_dna(n, 2),
}
# collections of similar genomes (the reference's reference / target and multi-FASTA entry points are made for them)
for copies in (17, 24, 48, 96):
    base = gen.random_dna(n // copies, 40 + copies)
    cases[f"{copies} genomes 0.1 % apart"] = np.concatenate([base] + [mutated(base, 100 * copies + k) for k in range(copies - 1)])
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
if only:
    cases = {k: v for k, v in cases.items() if any(o in k for o in only) or k == "random"}
native.count_factors(cases["random"][:1 << 16])
for name, t in cases.items():
    t0 = time.time(); z = native.count_factors(t); dt = time.time() - t0
    print(f"2^{lg} {name}: {dt*1e3:.1f} ms, z={z}", flush=True)
# a reference and a target of n / 2 bases each, 0.1 % apart, through the reference-sequence entry point
# (factorize_dna_w_reference_seq up to 2^24: the Python tuples of millions of factors take seconds; beyond that the
#  binary-file form of the same C entry point, nolzss_factorize_dna_w_reference_seq with a file for the records)
if not only or any("reference" in o for o in only):
    import os, tempfile
    ref = gen.random_dna(n // 2, 77)
    tgt = mutated(ref, 78)
    rs, ts = ref.tobytes().decode(), tgt.tobytes().decode()
    if lg <= 24:
        t0 = time.time()
        f = native.factorize_dna_w_reference_seq(rs, ts)
        print(f"2^{lg} reference + target through factorize_dna_w_reference_seq (tuples): {(time.time()-t0)*1e3:.1f} ms, z={len(f)}", flush=True)
    out = os.path.join(tempfile.gettempdir(), f"nolzss_degenerate_ref_{os.getpid()}.bin")
    t0 = time.time()
    z = native.factorize_dna_w_reference_seq_file(rs, ts, out)
    print(f"2^{lg} reference + target of 2^{lg-1} bases each through factorize_dna_w_reference_seq_file: {(time.time()-t0)*1e3:.1f} ms, "
          f"z={z}, {os.path.getsize(out)} bytes written", flush=True)
    os.remove(out)
