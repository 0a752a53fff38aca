#!/bin/bash
# round-2 session 1: new parity tests, bench line (self-spawn refusal, config4), memory-mix / co-execution probes, LVC phase stamps
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
(rocm-smi --showclocks --showpower --showperflevel 2>&1 | grep -v "^=\|^$" | head -30) > gpurun_out/box_state.txt
echo "== probes"
timeout 120 tools/ubench/copy_mix_probe > gpurun_out/copy_mix.txt 2>&1; cat gpurun_out/copy_mix.txt
timeout 120 tools/ubench/coexec_probe > gpurun_out/coexec.txt 2>&1; head -12 gpurun_out/coexec.txt
timeout 120 tools/ubench/lvc_h2_bench_t > gpurun_out/lvc_h2_t.txt 2>&1; cat gpurun_out/lvc_h2_t.txt
timeout 120 tools/ubench/lvc_h2_bench > gpurun_out/lvc_h2.txt 2>&1; cat gpurun_out/lvc_h2.txt
timeout 120 tools/ubench/lvc_h8_bench > gpurun_out/lvc_h8.txt 2>&1; cat gpurun_out/lvc_h8.txt
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu.log; grep -a "max|d|\|int16 max" gpurun_out/pytest_gpu.log | head
echo "== bench --gpus 2 on a 1-GPU box (must refuse)"; python bench.py --gpus 2 --steps 2 > gpurun_out/bench_gpus2.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/bench_gpus2.log
echo "== bench default"; timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; grep '^{' gpurun_out/bench.log | cut -c1-1500
echo "== bench config4"; timeout 600 python bench.py --workload config4 --steps 3 --warmup 1 > gpurun_out/bench_config4.log 2>&1; echo "rc=$?"; grep '^{' gpurun_out/bench_config4.log | cut -c1-700; tail -3 gpurun_out/bench_config4.log | cut -c1-300
