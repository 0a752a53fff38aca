#!/bin/bash
# round-2 session 13: bit-reproducibility loop of the first micro-batch while a second process leaves NaNs in LDS and registers
set -u
mkdir -p gpurun_out
N=${1:-60}
tools/ubench/poison 100 > gpurun_out/poison.txt 2>&1 &
PP=$!
sleep 1
timeout 400 python tools/flake_hunt.py Apoison $N reuse > gpurun_out/flake_poison.txt 2>&1
kill $PP 2>/dev/null; wait $PP 2>/dev/null
grep -v "Warning\|WeightNorm\|amdgpu.ids" gpurun_out/flake_poison.txt | tail -15 | cut -c1-300
cat gpurun_out/poison.txt
