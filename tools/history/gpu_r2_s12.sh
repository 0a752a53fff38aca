#!/bin/bash
# round-2 session 12: hunt for the rare two-process mismatch (tests/test_sharded_synthesis.py): two processes on the one GPU
set -u
mkdir -p gpurun_out
N=${1:-150}
timeout 600 python tools/flake_hunt.py A $N > gpurun_out/flake_A.txt 2>&1 &
PA=$!
timeout 600 python tools/flake_hunt.py B $N > gpurun_out/flake_B.txt 2>&1 &
PB=$!
wait $PA $PB
grep -v "Warning\|WeightNorm\|amdgpu.ids" gpurun_out/flake_A.txt | tail -12
grep -v "Warning\|WeightNorm\|amdgpu.ids" gpurun_out/flake_B.txt | tail -12
