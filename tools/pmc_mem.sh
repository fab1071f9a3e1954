#!/bin/bash
# tools/pmc_mem.sh <kernel-regex> <out-name> <probe args...>
# Memory-side counters (L2 / fabric) of one kernel, two or three per pass (more exceed the
# hardware's TCC counter slots), while tools/probe.py factorizes a text.  Run on the GPU box.
KREGEX="$1"; OUT="$2"; shift 2
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p "$ROOT/gpurun_out/$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for group in \
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
  "TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum" \
  "TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum" \
  "TCC_HIT_sum TCC_MISS_sum" \
  "TCC_REQ_sum TCC_BUSY_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum" \
  "TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $group --kernel-include-regex "$KREGEX" --output-format csv \
     -d "$ROOT/gpurun_out/$OUT/p$i" -o pmc -- python3 "$ROOT/tools/probe.py" "$@" > "$ROOT/gpurun_out/$OUT/p$i.log" 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
python3 - "$ROOT/gpurun_out/$OUT" <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
with open(root + "/summary.txt", "w") as out:
    for k, d in acc.items():
        out.write(k + "\n")
        for c, v in sorted(d.items()): out.write(f"  {c:40s} {v:.6g}\n")
print(open(root + "/summary.txt").read())
PY
