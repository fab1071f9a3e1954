"""Deterministic synthetic inputs named in SURVEY.md section 8d (generators + seeds only;
data is never committed)."""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def random_dna(n: int, seed: int = 0x6E6F4C5A) -> np.ndarray:
    """iid uniform ACGT (config 2: n = 64 Mi)."""
    rng = np.random.default_rng(seed)
    return ACGT[rng.integers(0, 4, size=n, dtype=np.uint8)]


def repeat_dna(n: int, seed: int = 0x5EED0003, p_copy: float = 0.4, lo: int = 64, hi: int = 65536,
               p_sub: float = 0.01) -> np.ndarray:
    """Config 3 generator: chunks of log-uniform length in [lo, hi]; with probability p_copy a
    chunk is a copy of an earlier region with 1 % point substitutions, else fresh iid ACGT."""
    rng = np.random.default_rng(seed)
    out = np.empty(n, dtype=np.uint8)
    pos = 0
    while pos < n:
        length = int(np.exp(rng.uniform(np.log(lo), np.log(hi))))
        length = max(1, min(length, n - pos))
        if pos > 0 and rng.random() < p_copy:
            src = int(rng.integers(0, pos))
            length = min(length, pos - src) if pos - src > 0 else length
            chunk = out[src:src + length].copy()
            nsub = rng.binomial(length, p_sub)
            if nsub:
                where = rng.integers(0, length, size=nsub)
                chunk[where] = ACGT[rng.integers(0, 4, size=nsub)]
            out[pos:pos + length] = chunk
        else:
            out[pos:pos + length] = ACGT[rng.integers(0, 4, size=length, dtype=np.uint8)]
        pos += length
    return out


def fasta_records(m: int, length: int, seed0: int = 0x4000):
    """Config 4: m records of `length` random bases, ids seq{k}, seeds seed0 + k."""
    return [(f"seq{k}", random_dna(length, seed0 + k)) for k in range(m)]


def write_fasta_fast(path, records, width: int = 80):
    """write_fasta for records of millions of bases: whole lines at a time through NumPy"""
    with open(path, "wb") as f:
        for rid, seq in records:
            f.write(b">" + rid.encode() + b"\n")
            a = np.frombuffer(seq, dtype=np.uint8) if not isinstance(seq, np.ndarray) else seq
            full = len(a) // width * width
            if full:
                lines = np.empty((full // width, width + 1), dtype=np.uint8)
                lines[:, :width] = a[:full].reshape(-1, width)
                lines[:, width] = 10
                lines.tofile(f)
            if full < len(a):
                f.write(a[full:].tobytes() + b"\n")


def write_fasta(path, records, width: int = 80):
    with open(path, "wb") as f:
        for rid, seq in records:
            f.write(b">" + rid.encode() + b"\n")
            b = seq.tobytes() if isinstance(seq, np.ndarray) else bytes(seq)
            for i in range(0, len(b), width):
                f.write(b[i:i + width] + b"\n")
