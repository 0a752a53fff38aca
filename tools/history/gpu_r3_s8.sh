#!/bin/bash
# round-3 session 8: whole GPU suite with the Python module's new default (host-checked range fallback), smoke, bench lines
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
echo "== pytest gpu" ; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -rx > gpurun_out/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; tail -5 gpurun_out/pytest_gpu.log | cut -c1-300
echo "== smoke" ; timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 ; echo "smoke rc=$?" ; tail -1 gpurun_out/smoke.log
echo "== bench" ; timeout 900 python bench.py > gpurun_out/bench.log 2>&1 ; echo "bench rc=$?" ; grep '^{' gpurun_out/bench.log | cut -c1-400
echo "== bench N=1000 B=1" ; timeout 900 python bench.py --batch 1 --nsteps 1000 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline > gpurun_out/bench_n1000.log 2>&1 ; grep '^{' gpurun_out/bench_n1000.log | cut -c1-300
echo "== bench config4" ; timeout 900 python bench.py --workload config4 --steps 5 --warmup 2 > gpurun_out/bench_config4.log 2>&1 ; grep '^{' gpurun_out/bench_config4.log | cut -c1-300
echo "== 2 ranks sharing this GPU (FD_BENCH_OVERSUBSCRIBE, disjoint CU masks): a code-path check"
FD_BENCH_OVERSUBSCRIBE=1 timeout 900 python bench.py --gpus 2 --workload config4 --steps 3 --warmup 1 > gpurun_out/bench_config4_2ranks_1gpu.log 2>&1 ; grep '^{' gpurun_out/bench_config4_2ranks_1gpu.log | cut -c1-300
