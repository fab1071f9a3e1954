#!/bin/bash
# tools/pmc_kernel.sh <kernel-regex> <out-name> <probe args...>
# Collects hardware counters for one kernel of the pipeline (rocprofv3 --pmc, one pass per counter
# group, kernel trace only) while tools/probe.py factorizes a text.  Run on the GPU box.
set -e
KREGEX="$1"; OUT="$2"; shift 2
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p "$ROOT/gpurun_out/$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for group in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
  "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" \
  "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
  "SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_ANY" \
  "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
  "FETCH_SIZE" "WRITE_SIZE"; do
  # Every group stays within what one pass can collect: a round-1 pass that asked for more (TA_* next to
  # eight SQ counters) made rocprofv3 fail with "Could not construct profile cfg ... error code 38: Request
  # exceeds the capabilities of the hardware to collect" and abort (SIGABRT) inside the first kernel launch
  # -- an over-subscribed counter group (gpurun_out/pmc_refine/p3.log), not a GPU fault.  No counter
  # appears in two groups, so the sums below are not doubled.
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $group --kernel-include-regex "$KREGEX" --output-format csv \
     -d "$ROOT/gpurun_out/$OUT/p$i" -o pmc -- python3 "$ROOT/tools/probe.py" "$@" > "$ROOT/gpurun_out/$OUT/p$i.log" 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
python3 - "$ROOT/gpurun_out/$OUT" <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        import re
        m = re.search(r"([A-Za-z_0-9]+_kernel)", r["Kernel_Name"])  # (the names start with "void nolzss::(anonymous namespace)::")
        k = m.group(1) if m else r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen.add((k, r["Dispatch_Id"]))
    for k, _ in seen: calls[(k, f)] += 1
with open(root + "/summary.txt", "w") as out:
    for k, d in acc.items():
        out.write(k + "\n")
        for c, v in sorted(d.items()): out.write(f"  {c:40s} {v:.6g}\n")
print(open(root + "/summary.txt").read())
PY
