#!/bin/bash
# A/B of a compile-time variant of the library on the GPU box:
#   make -C nolzss_amd/csrc variant VARIANT=<name> EXTRA=-D...      (here, before gpurun)
#   tools/ab_variant.sh <name> [bench args]                          (on the box)
# Parity of the variant on the pipeline / primitive / batch / RC tests (through NOLZSS_LIB; the tests that import
# the compiled noLZSS module are left out: it is linked against the default build), then the headline step of
# both builds back to back.  Output under gpurun_out/ab_<name>/.
set -o pipefail
name=$1; shift
out=gpurun_out/ab_$name
mkdir -p $out
lib=$PWD/nolzss_amd/libnolzss_hip.$name.so
[ -f $lib ] || { echo "no $lib"; exit 2; }
NOLZSS_LIB=$lib python -m pytest tests/test_gpu_primitives.py tests/test_gpu_pipeline.py tests/test_gpu_batch_merged.py \
    tests/test_gpu_rc.py tests/test_gpu_scale.py::test_config2_random_64Mi_exact tests/test_gpu_scale.py::test_rc_4Mi_exact \
    -x -q -m gpu > $out/pytest_variant.log 2>&1 || { tail -30 $out/pytest_variant.log; exit 1; }
tail -2 $out/pytest_variant.log
args="--steps 5 --warmup 2 --no-cpu-baseline --no-stopwatches $*"
python bench.py $args > $out/bench_default.json 2> $out/bench_default.err || exit 1
NOLZSS_LIB=$lib python bench.py $args > $out/bench_variant.json 2> $out/bench_variant.err || exit 1
python - $out <<'PY'
import json, sys
out = sys.argv[1]
a = json.loads(open(f"{out}/bench_default.json").read().strip().splitlines()[-1])
b = json.loads(open(f"{out}/bench_variant.json").read().strip().splitlines()[-1])
print(f"ms_per_step default {a['ms_per_step']:.2f}  variant {b['ms_per_step']:.2f}")
for sec in ("stages_ms_per_step", "kernels_ms_per_step"):
    for k in sorted(set(a[sec]) | set(b[sec]), key=lambda k: -a[sec].get(k, 0)):
        x, y = a[sec].get(k, 0), b[sec].get(k, 0)
        if max(x, y) >= 0.3:
            print(f"  {k:24s} {x:8.2f} {y:8.2f}  {y - x:+.2f}")
for k, v in b["roofline"]["by_class"].items():
    print(f"  {k:24s} {a['roofline']['by_class'].get(k, {}).get('achieved_GBps', 0):8.0f} {v['achieved_GBps']:8.0f} GB/s")
for sec in ("fasta512", "rc256m"):
    if sec in a and sec in b:
        print(f"  {sec}: {a[sec]['ms_per_step']:.2f} -> {b[sec]['ms_per_step']:.2f} ms")
PY
