#!/bin/bash
# round-2 session 4: what binds the LVC layer -- instruction issue costs, compute-only time, staggered workgroup starts
set -u
mkdir -p gpurun_out
tools/ubench/valu_rate_probe 2>&1 | tee gpurun_out/valu_rate.txt
for rep in 1 2; do
echo "== normal";  tools/ubench/lvc_h2_bench 8 864 1 | grep lvc_f16
echo "== no HBM reads (compute + stores only)"; tools/ubench/lvc_noload 8 864 1 | grep lvc_f16
echo "== stagger 1 x s_sleep 127"; tools/ubench/lvc_stag1 8 864 1 | grep lvc_f16
echo "== stagger 2 x s_sleep 127"; tools/ubench/lvc_stag2 8 864 1 | grep lvc_f16
done 2>&1 | tee gpurun_out/lvc_bound.txt
tools/ubench/copy_mix_probe | tail -2
