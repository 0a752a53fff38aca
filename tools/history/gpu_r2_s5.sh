#!/bin/bash
# round-2 session 5: host-checked fallback (option fallback = host): parity, then A/B of the two modes at B=8 and B=1, N=1000
set -u
mkdir -p gpurun_out
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest_gpu.log | cut -c1-400
for b in 8 1; do for mode in host graph; do
  echo "== B=$b fallback=$mode"; python bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-pipe --no-roofline --no-host-io --opt fallback=$mode 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])"
done; done 2>&1 | tee gpurun_out/fallback_modes.txt
for mode in host graph; do echo "== N=1000 B=1 fallback=$mode"; python bench.py --batch 1 --nsteps 1000 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-fp32-pipe --no-host-io --opt fallback=$mode 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])"; done 2>&1 | tee -a gpurun_out/fallback_modes.txt
