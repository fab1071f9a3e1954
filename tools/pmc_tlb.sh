#!/bin/bash
# tools/pmc_tlb.sh <out-name>: address-translation counters of rs_scatter_kernel at 2^29 and 2^30 bases
# (rocprofv3 --pmc, kernel trace only, the program directly after "--").  Run on the GPU box.
OUT="$1"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p "$ROOT/gpurun_out/$OUT"
cd /tmp && export TMPDIR=/tmp
for L in 29 30; do
  i=0
  for group in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $group --kernel-include-regex "rs_scatter_kernel" --output-format csv \
       -d "$ROOT/gpurun_out/$OUT/n$L.p$i" -o pmc -- python3 "$ROOT/tools/probe.py" repeat "2^$L" --reps 1 > "$ROOT/gpurun_out/$OUT/n$L.p$i.log" 2>&1 || echo "2^$L pass $i failed"
    echo "2^$L pass $i done"
  done
done
python3 - "$ROOT/gpurun_out/$OUT" <<'PY'
import csv, glob, sys, collections, re
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(root + "/n*/**/*counter_collection.csv", recursive=True):
    size = re.search(r"/n(\d+)\.p", f).group(1)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        cls = "u32" if "IjjNS" in name or "<unsigned int, unsigned int" in name else "other"
        acc[(size, cls)][r["Counter_Name"]] += float(r["Counter_Value"])
with open(root + "/summary.txt", "w") as out:
    for k, d in sorted(acc.items()):
        out.write(f"rs_scatter_kernel {k[1]} instantiations, text of 2^{k[0]} bases\n")
        for c, v in sorted(d.items()): out.write(f"  {c:52s} {v:.6g}\n")
        if "TCP_UTCL1_TRANSLATION_MISS_sum" in d and "TCP_UTCL1_TRANSLATION_HIT_sum" in d:
            out.write(f"  {'UTCL1 miss rate':52s} {d['TCP_UTCL1_TRANSLATION_MISS_sum']/(d['TCP_UTCL1_TRANSLATION_MISS_sum']+d['TCP_UTCL1_TRANSLATION_HIT_sum']):.4f}\n")
print(open(root + "/summary.txt").read())
PY
