#!/bin/bash
# round-2 session 2: parity of the restructured LVC layer / packed predicted kernels, harness timings, A/B against the previous build
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
(rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|power (W)" | head -4) > gpurun_out/box_state.txt
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log | cut -c1-300
echo "== harness"; timeout 120 tools/ubench/lvc_h2_bench_t > gpurun_out/lvc_h2_t.txt 2>&1; cat gpurun_out/lvc_h2_t.txt
timeout 120 tools/ubench/lvc_h2_bench > gpurun_out/lvc_h2.txt 2>&1; cat gpurun_out/lvc_h2.txt
timeout 120 tools/ubench/lvc_h2_bench 8 864 0 > gpurun_out/lvc_h2_nokpre.txt 2>&1; cat gpurun_out/lvc_h2_nokpre.txt
echo "== A/B"; cp fastdiff_amd/lib/libfastdiff_hip.so /tmp/new.so; AB_ARGS="--no-fp32-pipe" bash tools/gpu_ab.sh gpurun_ab/base.so /tmp/new.so 3 2>&1 | tee gpurun_out/ab.txt
echo "== A/B B=1"; AB_ARGS="--no-fp32-pipe --batch 1" bash tools/gpu_ab.sh gpurun_ab/base.so /tmp/new.so 2 2>&1 | tee gpurun_out/ab_b1.txt
