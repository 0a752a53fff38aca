#!/bin/bash
# round-3 session 27: halo columns as two extra conv tiles on the matrix pipe -- standalone A/B against HEAD and the v_dot2 form
mkdir -p gpurun_out
{
for i in 1 2 3; do
echo "== base"; tools/ubench/lvc_h2_bench_base 8 864
echo "== dot2"; tools/ubench/lvc_h2_bench_dot2 8 864
echo "== mfma"; tools/ubench/lvc_h2_bench 8 864
done
tools/ubench/lvc_h2_timeline gpurun_out/timeline_mfma.bin
} > gpurun_out/halo_ab3.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/pytest_halo3.txt 2>&1; tail -3 gpurun_out/pytest_halo3.txt
