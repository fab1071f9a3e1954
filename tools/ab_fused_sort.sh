#!/bin/bash
# tools/ab_fused_sort.sh <out-name>: A/B of the key sort on split arrays (default) against fused 64-bit records
# (NOLZSS_FUSED_SORT=1) on the 2^30-base benchmark text: stage times of both, then the write-request counters of the
# radix kernels (TCC_EA0_WRREQ / WRREQ_64B: all write requests the L2 sends to memory / those of a full 64 bytes) and
# FETCH_SIZE / WRITE_SIZE, one counter group per pass, program directly behind "--".  Run on the GPU box.
OUT="$1"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p "$ROOT/gpurun_out/$OUT"
PY=$(readlink -f "$(command -v python3)")
cd "$ROOT"
python3 tools/probe.py repeat 2^30 --reps 3 > gpurun_out/$OUT/probe_split.log 2>&1 || exit 1
NOLZSS_FUSED_SORT=1 python3 tools/probe.py repeat 2^30 --reps 3 > gpurun_out/$OUT/probe_fused.log 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
for variant in split fused; do
  if [ $variant = fused ]; then export NOLZSS_FUSED_SORT=1; else unset NOLZSS_FUSED_SORT; fi
  i=0
  for group in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $group --kernel-include-regex "rs_scatter|rs_hist" --output-format csv \
       -d "$ROOT/gpurun_out/$OUT/${variant}_p$i" -o pmc -- "$PY" "$ROOT/tools/probe.py" repeat 2^30 --reps 1 > "$ROOT/gpurun_out/$OUT/${variant}_p$i.log" 2>&1 || echo "$variant pass $i failed"
  done
done
unset NOLZSS_FUSED_SORT
"$PY" - "$ROOT/gpurun_out/$OUT" <<'PY'
import csv, glob, re, sys, collections
root = sys.argv[1]
out = open(root + "/summary.txt", "w")
def p(*a):
    print(*a); print(*a, file=out)
for variant in ("split", "fused"):
    p(f"== {variant}: stage times (tools/probe.py repeat 2^30, third repetition)")
    for line in open(f"{root}/probe_{variant}.log"):
        if re.search(r"rep 2|sa_sort_initial|rs_scatter|rs_hist|rs_scan", line):
            p("  " + line.rstrip())
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for f in glob.glob(f"{root}/{variant}_p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            k = re.sub(r"nolzss::\(anonymous namespace\)::|nolzss::|void ", "", k)
            k = re.sub(r"unsigned int", "u32", re.sub(r"unsigned long", "u64", re.sub(r"unsigned short", "u16", k)))
            k = re.sub(r"\(.*", "", k)
            # the passes of the key sort only (2^30 pairs; the permutation passes carry 64-bit values: PairSrc / RankSrc)
            if int(r["Grid_Size"]) < (1 << 26): continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    p(f"== {variant}: counters, sums over the launches of one factorization")
    for k, d in sorted(acc.items()):
        p(f"  {k}  x{len(n[k])}")
        for c, v in sorted(d.items()):
            p(f"      {c:28s} {v:.6g}")
PY
