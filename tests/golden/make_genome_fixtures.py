#!/usr/bin/env python3
"""Copies the DATA fixtures the reference's own tests hold for the factorize path into
tests/golden/genomes/ (inputs only -- FASTA files and the two binary factor files; no source
text).  Run in the authoring container, where /root/reference exists; the GPU box only sees
the committed copies.

Sources (reference repo, tests/resources/), used by tests/test_factorization_validation.py:87,
tests/test_reference_seq.py and tests/test_genomics.py:
  T3.fasta, T7.fasta                      bacteriophage genomes (38 kb / 40 kb)
  short_dna1.fasta, short_dna2.fasta      two records each, 12-19 bases
  test_viral_dna.fna                      one 369 kb record
  test_bacterial_dna.fna                  two records, 233 kb, AT-rich
  Vibrio_cholerae.fna                     three records, 4.1 Mb (stored gzip-compressed)
  dna1_factors_w_dna2_ref.bin             v2 factor file, STALE (RC-preferred tie-break of an
  T7_factors_w_T3_ref.bin                 older version / v1 layout): kept as partial evidence only
"""
import gzip
import hashlib
import json
import shutil
import sys
from pathlib import Path

SRC = Path("/root/reference/tests/resources")
DST = Path(__file__).resolve().parent / "genomes"
PLAIN = ["T3.fasta", "T7.fasta", "short_dna1.fasta", "short_dna2.fasta", "test_viral_dna.fna",
         "test_bacterial_dna.fna", "dna1_factors_w_dna2_ref.bin", "T7_factors_w_T3_ref.bin"]
GZIPPED = ["Vibrio_cholerae.fna"]


def main():
    if not SRC.is_dir():
        sys.exit(f"{SRC} not found: this script runs where the reference checkout is")
    DST.mkdir(parents=True, exist_ok=True)
    manifest = {}
    for name in PLAIN:
        shutil.copyfile(SRC / name, DST / name)
        manifest[name] = {"sha256": hashlib.sha256((SRC / name).read_bytes()).hexdigest(),
                          "bytes": (SRC / name).stat().st_size, "source": f"tests/resources/{name}"}
    for name in GZIPPED:
        raw = (SRC / name).read_bytes()
        with open(DST / (name + ".gz"), "wb") as f, gzip.GzipFile(fileobj=f, mode="wb", mtime=0, compresslevel=9) as g:
            g.write(raw)
        manifest[name + ".gz"] = {"sha256_uncompressed": hashlib.sha256(raw).hexdigest(), "bytes_uncompressed": len(raw),
                                  "source": f"tests/resources/{name}"}
    (DST / "MANIFEST.json").write_text(json.dumps(manifest, indent=1, sort_keys=True) + "\n")
    print(f"{len(manifest)} fixtures -> {DST}")


if __name__ == "__main__":
    main()
