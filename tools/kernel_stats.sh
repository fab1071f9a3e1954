#!/bin/bash
# rocprofv3 kernel statistics of one command, summarised:  tools/kernel_stats.sh <out name> <program> [args]
# (run on the GPU box; the program comes directly after "--": no env / bash -c hop under the profiler; the
# profiler runs in /tmp, so give script paths relative to the repository root: they are made absolute here)
name=$1; shift
GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
# the program must be a real ELF binary: a shim or a "#!/usr/bin/env" script would exec under the profiler's preloaded
# runtime (which has initialised the GPU by then), and that is what takes a box down
prog=$(readlink -f "$(command -v "$1")")
if [ -z "$prog" ] || [ "$(head -c 4 "$prog" | od -An -c | tr -d ' ')" != "177ELF" ]; then
  echo "tools/kernel_stats.sh: '$1' does not resolve to an ELF binary ($prog): refusing to run it under rocprofv3" >&2
  exit 2
fi
shift; set -- "$prog" "$@"
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
args=()
for a in "$@"; do
  if [ -e "$GRAFT_REPO_ROOT/$a" ]; then args+=("$GRAFT_REPO_ROOT/$a"); else args+=("$a"); fi
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o trace -- "${args[@]}" > $out/run.log 2>&1
rc=$?
cd $GRAFT_REPO_ROOT
python3 tools/kernel_stats_summary.py $out > $out/kernel_stats.txt
head -45 $out/kernel_stats.txt
exit $rc
