#!/bin/bash
# round-3 session 26: halo columns on v_dot2_f32_f16 with the weights from the A operand registers -- standalone A/B, timeline, parity tests, bench
mkdir -p gpurun_out
{
for i in 1 2; do
echo "== base"; tools/ubench/lvc_h2_bench_base 8 864; tools/ubench/lvc_h8_bench_base 8 864
echo "== dot2"; tools/ubench/lvc_h2_bench 8 864; tools/ubench/lvc_h8_bench 8 864
done
tools/ubench/lvc_h2_timeline gpurun_out/timeline_wa.bin
} > gpurun_out/halo_ab2.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/pytest_halo2.txt 2>&1; tail -3 gpurun_out/pytest_halo2.txt
python bench.py --no-cpu-baseline > gpurun_out/bench_halo2.json 2> gpurun_out/bench_halo2.err; tail -c 1500 gpurun_out/bench_halo2.json
