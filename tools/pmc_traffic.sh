#!/bin/bash
# tools/pmc_traffic.sh <kernel-regex> <out-name> <probe args...>
# HBM traffic of one kernel: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes
# (kernel trace only) while tools/probe.py factorizes a text.  Run on the GPU box.
KREGEX="$1"; OUT="$2"; shift 2
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p "$ROOT/gpurun_out/$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "$KREGEX" --output-format csv \
     -d "$ROOT/gpurun_out/$OUT/$c" -o pmc -- python3 "$ROOT/tools/probe.py" "$@" > "$ROOT/gpurun_out/$OUT/$c.log" 2>&1 || echo "pass $c failed"
  echo "pass $c done"
done
python3 - "$ROOT/gpurun_out/$OUT" <<'PY'
import csv, glob, sys, collections, json
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for f in glob.glob(root + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = k[k.find("rs_"):][:70] if "rs_" in k else k[:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
out = {k: {"launches": len(calls[k]), **d} for k, d in acc.items()}
json.dump(out, open(root + "/traffic_raw.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
