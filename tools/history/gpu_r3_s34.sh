#!/bin/bash
# round-3 session 34: predictor hoisted per 8-step piece of a long schedule: parity tests, then N=1000 / N=200 at B=1 with hoist on and off
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hoisted or long_schedule or n1000 or sampler or graph" > gpurun_out/pytest_hoist.txt 2>&1; tail -5 gpurun_out/pytest_hoist.txt
for N in 1000 200; do for hz in on off; do
python bench.py --batch 1 --nsteps $N --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-pipe --no-b1 --no-host-io --no-roofline --opt hoist=$hz 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=$N hoist=$hz', j['ms_per_step'], j['value'])"
done; done > gpurun_out/hoist_long.txt 2>&1
cat gpurun_out/hoist_long.txt
