#!/bin/bash
# round-2 session 19: do workgroups of DIFFERENT LDS sizes from two processes keep their LDS apart on one CU?  cwsr_lds_probe (72 KB per
# workgroup, pattern held for HOLD us and checked) next to poison processes whose workgroups scribble NaNs over smaller LDS allocations
set -u
mkdir -p gpurun_out
HOLD=${1:-300}
for sz in 16384 20480 53248 57344; do
  tools/ubench/poison 14 $sz 4 > /dev/null 2>&1 &
  PP=$!
  sleep 1
  tools/ubench/cwsr_lds_probe 73728 10 $HOLD 2>&1 | sed "s/^/poison LDS $sz: /"
  kill $PP 2>/dev/null; wait $PP 2>/dev/null
done | tee gpurun_out/lds_mixed_sizes.txt
