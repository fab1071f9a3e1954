"""Loader for the real-sequence fixtures under tests/golden/genomes/ (copied from the reference's
tests/resources/ by tests/golden/make_genome_fixtures.py) and the invariants the reference checks
on them (reference: tests/test_factorization_validation.py:118-211)."""
import gzip
import struct
from pathlib import Path

DIR = Path(__file__).resolve().parent / "golden" / "genomes"
_COMP = bytes.maketrans(b"ACGT", b"TGCA")
RC_MASK = 1 << 63

NAMES = ["short_dna1", "short_dna2", "T3", "T7", "test_viral_dna", "test_bacterial_dna", "Vibrio_cholerae"]
_FILES = {"short_dna1": "short_dna1.fasta", "short_dna2": "short_dna2.fasta", "T3": "T3.fasta", "T7": "T7.fasta",
          "test_viral_dna": "test_viral_dna.fna", "test_bacterial_dna": "test_bacterial_dna.fna",
          "Vibrio_cholerae": "Vibrio_cholerae.fna.gz"}


def raw(name: str) -> bytes:
    p = DIR / _FILES[name]
    return gzip.decompress(p.read_bytes()) if p.suffix == ".gz" else p.read_bytes()


def records(name: str):
    """[(id, upper-case sequence bytes)] -- plain line-by-line parse, ids = first header word."""
    out = []
    for line in raw(name).splitlines():
        line = line.strip()
        if not line:
            continue
        if line.startswith(b">"):
            out.append([line[1:].split()[0].decode(), []])
        else:
            out[-1][1].append(line.upper())
    return [(rid, b"".join(parts)) for rid, parts in out]


def materialize(name: str, tmp_path) -> str:
    """the FASTA file as a real (uncompressed) file, for the path-taking entry points"""
    p = Path(tmp_path) / (name + ".fasta")
    p.write_bytes(raw(name))
    return str(p)


def revcomp(b: bytes) -> bytes:
    return b.translate(_COMP)[::-1]


def check_plain_invariants(text: bytes, factors, start_pos=0):
    """tiling from start_pos to the end, every factor a literal or a true earlier,
    non-overlapping occurrence (reference: factorizer_core.hpp:51-119 contract)"""
    pos = start_pos
    for s, l, r in factors:
        assert s == pos and l >= 1
        if r == s:
            assert l == 1
        else:
            assert r + l <= s and text[r:r + l] == text[s:s + l]
        pos += l
    assert pos == len(text)


def check_rc_invariants(S: bytes, original_length: int, factors, sentinel_positions=(), start_pos=0):
    """the reference's own checks (tests/test_factorization_validation.py:118-211): gap-free
    coverage of [start_pos, original_length), every factor a true forward match or the reverse
    complement of the referenced stretch, sentinels literal"""
    sent = set(sentinel_positions)
    pos = start_pos
    for s, l, r, is_rc in factors:
        assert s == pos and l >= 1
        if s in sent:
            assert (l, r, is_rc) == (1, s, False)
        elif r == s and not is_rc:
            assert l == 1
        elif is_rc:
            assert revcomp(S[r:r + l]) == S[s:s + l]
        else:
            assert r + l <= s and S[r:r + l] == S[s:s + l]
        pos += l
    assert pos >= original_length - 1


def read_v1_factor_file(name: str):
    """the two stale binary fixtures use the reference's OLD layout: header first
    ("noLZSSv1", num_factors, num_sequences, num_sentinels, header_size, names, sentinel indices)"""
    b = (DIR / name).read_bytes()
    magic, nf, nseq, nsent, hsize = struct.unpack("<8sQQQQ", b[:40])
    assert magic == b"noLZSSv1"
    body = b[hsize:]
    fs = [struct.unpack("<QQQ", body[24 * k:24 * k + 24]) for k in range(nf)]
    return [(s, l, r & (RC_MASK - 1), bool(r >> 63)) for s, l, r in fs]
