#!/bin/bash
set -u
mkdir -p gpurun_out
echo "== pytest (fallback tests + rest)"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu.log | cut -c1-400
for mode in host graph; do echo "== N=1000 B=1 fallback=$mode"; python bench.py --batch 1 --nsteps 1000 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-fp32-pipe --no-host-io --opt fallback=$mode > gpurun_out/n1000_$mode.log 2>&1; tail -3 gpurun_out/n1000_$mode.log | cut -c1-600; done
for mode in host graph; do echo "== N=200 B=8 fallback=$mode"; python bench.py --batch 8 --nsteps 200 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-fp32-pipe --no-host-io --opt fallback=$mode > gpurun_out/n200_$mode.log 2>&1; tail -1 gpurun_out/n200_$mode.log | cut -c1-300; done
