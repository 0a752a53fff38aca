#!/bin/bash
# round-3 session 17: the eager-framework baseline on the same GPU, lean and "like the reference" (per-forward weight-norm, host schedule work per call, CPU RNG + H2D per step)
set -u
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
for b in 8 1; do
timeout 600 python bench.py --batch $b --steps 10 --torch-eager-baseline --no-cpu-baseline --no-roofline --no-fp32-pipe --no-b1 --no-host-io 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B=%d' % d['config']['batch_per_gpu'], 'HIP path ms', d['ms_per_step'], 'eager', json.dumps(d['torch_eager_baseline']))"
done | tee gpurun_out/eager_baselines.txt
