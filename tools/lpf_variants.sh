#!/bin/bash
# tools/lpf_variants.sh [log2n]: stage time of lpf_tile_kernel for every NOLZSS_LPF_VARIANT (run on the GPU box)
L=${1:-28}
for v in 0 1 2 3 4 5 6 7; do
  echo "variant $v: $(NOLZSS_LPF_VARIANT=$v python3 tools/probe.py repeat 2^$L --reps 3 | grep -E '^  (lpf |lpf_far|lpnf_fallback)' | tr -s ' ' | tr '\n' ';')"
done
