"""One-line summary of a bench.py JSON line: python tools/bench_brief.py <file> [label]"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
parts = [sys.argv[2] if len(sys.argv) > 2 else "", f"ms_per_step {d['ms_per_step']:.2f}"]
for sec in ("fasta512", "rc256m"):
    if sec in d and isinstance(d[sec], dict) and "ms_per_step" in d[sec]:
        parts.append(f"{sec} {d[sec]['ms_per_step']:.1f}")
print("  ".join(parts))
if len(sys.argv) > 3:
    print(json.dumps(d.get("stages_ms_per_step", {})))
