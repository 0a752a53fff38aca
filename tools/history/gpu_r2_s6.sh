#!/bin/bash
# round-2 session 6: 8-wave LVC workgroups: harness (4 vs 8 waves, compute-only), parity, A/B against the previous build
set -u
mkdir -p gpurun_out
for rep in 1 2; do
echo "== 4 waves";  tools/ubench/lvc_h2_bench 8 864 1 4 | grep "lvc<"
echo "== 8 waves";  tools/ubench/lvc_h2_bench 8 864 1 8 | grep "lvc<"
done 2>&1 | tee gpurun_out/lvc_w8.txt
echo "== 8 waves, no HBM reads"; tools/ubench/lvc_noload 8 864 1 8 | grep "lvc<" | tee -a gpurun_out/lvc_w8.txt
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu.log | cut -c1-400
echo "== A/B"; cp fastdiff_amd/lib/libfastdiff_hip.so /tmp/new.so; AB_ARGS="--no-fp32-pipe" bash tools/gpu_ab.sh gpurun_ab/base.so /tmp/new.so 3 2>&1 | tee gpurun_out/ab.txt
echo "== A/B B=1"; AB_ARGS="--no-fp32-pipe --batch 1" bash tools/gpu_ab.sh gpurun_ab/base.so /tmp/new.so 2 2>&1 | tee gpurun_out/ab_b1.txt
