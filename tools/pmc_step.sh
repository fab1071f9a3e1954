#!/bin/bash
# tools/pmc_step.sh <out-name> <probe args...>
# Hardware counters of EVERY kernel of one factorization (tools/probe.py ... --reps 1): rocprofv3 --kernel-trace --pmc,
# one pass per counter group (each group within what one pass can collect, no counter twice), the program directly
# behind "--".  Summary: gpurun_out/<out-name>/step.json -- per kernel (launches, time, FETCH_SIZE / WRITE_SIZE bytes
# with the gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE doubled, SQ instruction / LDS counters) and the totals
# of the whole step (HBM bytes per base).  Run on the GPU box; copy the summary to profiles/.
OUT="$1"; shift
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p "$ROOT/gpurun_out/$OUT"
PY=$(readlink -f "$(command -v python3)")
cd /tmp && export TMPDIR=/tmp
i=0
for group in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
  "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $group --output-format csv -d "$ROOT/gpurun_out/$OUT/p$i" -o pmc \
     -- "$PY" "$ROOT/tools/probe.py" "$@" > "$ROOT/gpurun_out/$OUT/p$i.log" 2>&1 || { echo "pass $i ($group) failed"; tail -5 "$ROOT/gpurun_out/$OUT/p$i.log"; }
  echo "pass $i done"
done
"$PY" - "$ROOT/gpurun_out/$OUT" "$@" <<'PY'
import csv, glob, json, re, sys, collections
root, probe_args = sys.argv[1], sys.argv[2:]
n = None
for a in probe_args[1:2]:  # tools/probe.py <kind> <n> ...
    n = (1 << int(a[2:])) if a.startswith("2^") else int(a)


def short(name):
    name = re.sub(r"nolzss::\(anonymous namespace\)::|nolzss::|void ", "", name)
    name = re.sub(r"unsigned int", "u32", name)
    name = re.sub(r"unsigned long", "u64", name)
    name = re.sub(r"unsigned short", "u16", name)
    if name.startswith("rs_scatter_kernel<"):
        name = re.sub(r", false>", ">", name)  # (its kTimed parameter)
    return re.sub(r"\(.*", "", name).strip()


acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
dur = collections.defaultdict(float)
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k].add(r["Dispatch_Id"])
# kernel durations from the trace of the first pass (the counters of a pass do not change them much)
trace = sorted(glob.glob(root + "/p1/**/*kernel_trace.csv", recursive=True))
if trace:
    for r in csv.DictReader(open(trace[0])):
        dur[short(r["Kernel_Name"])] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6
kernels = {}
tot_f = tot_w = 0.0
for k, d in acc.items():
    e = {"launches": len(launches[k]), "ms_under_profiler": round(dur.get(k, 0.0), 3)}
    if "FETCH_SIZE" in d:
        e["fetch_bytes"] = d["FETCH_SIZE"] * 1024.0 * 2.0  # KB, doubled (gfx950: 128-B requests tallied at 64 B)
        tot_f += e["fetch_bytes"]
    if "WRITE_SIZE" in d:
        e["write_bytes"] = d["WRITE_SIZE"] * 1024.0
        tot_w += e["write_bytes"]
    for c, v in d.items():
        if c not in ("FETCH_SIZE", "WRITE_SIZE"):
            e[c] = v
    kernels[k] = e
out = {"_how": "tools/pmc_step.sh " + " ".join(probe_args) + ": rocprofv3 --kernel-trace --pmc, one pass per counter group, program "
               "directly after '--'; sums over the launches of ONE factorization; FETCH_SIZE (KB) doubled per the gfx950 "
               "correction of MI355X_MICROARCH.md, WRITE_SIZE (KB) as counted",
       "bases": n,
       "step": {"fetch_bytes": tot_f, "write_bytes": tot_w, "hbm_bytes": tot_f + tot_w,
                "hbm_bytes_per_base": (tot_f + tot_w) / n if n else None,
                "kernel_ms_under_profiler": round(sum(dur.values()), 2)},
       "kernels": dict(sorted(kernels.items(), key=lambda kv: -kv[1]["ms_under_profiler"]))}
json.dump(out, open(root + "/step.json", "w"), indent=1)
print(json.dumps(out["step"], indent=1))
for k, e in list(out["kernels"].items())[:12]:
    print(f"{k[:64]:64s} x{e['launches']:3d} {e['ms_under_profiler']:8.2f} ms  fetch {e.get('fetch_bytes', 0)/1e9:7.2f} GB  write {e.get('write_bytes', 0)/1e9:7.2f} GB"
          f"  valu {e.get('SQ_INSTS_VALU', 0):.3g}  salu {e.get('SQ_INSTS_SALU', 0):.3g}  lds-active {e.get('SQ_LDS_IDX_ACTIVE', 0):.3g}")
PY
