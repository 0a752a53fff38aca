#!/bin/bash
# round-2 session 7: the tree after the LVC experiments were taken out again: parity, A/B of a 3-waves-per-SIMD DBlock
set -u
mkdir -p gpurun_out
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log | cut -c1-300
echo "== A/B dblock occupancy"; cp fastdiff_amd/lib/libfastdiff_hip.so /tmp/new.so; AB_ARGS="--no-fp32-pipe" bash tools/gpu_ab.sh /tmp/new.so gpurun_ab/dblock3.so 3 2>&1 | tee gpurun_out/ab_dblock.txt
echo "== A/B B=1"; AB_ARGS="--no-fp32-pipe --batch 1" bash tools/gpu_ab.sh /tmp/new.so gpurun_ab/dblock3.so 2 2>&1 | tee gpurun_out/ab_dblock_b1.txt
