#!/bin/bash
# round-3 session 54: fused up-sampler restructured (ConvTranspose -> parking area -> the shared staging code; skip in flight meanwhile): tests, A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "up_sampler or random_shapes or host_checked or hoisted or ragged_batch_with_lens or embedding" > gpurun_out/pytest_up2.txt 2>&1; tail -3 gpurun_out/pytest_up2.txt
for B in 1 8; do
python tools/ab_opts.py --batch $B --reps 3 --steps 30 "fuse_up=0" "fuse_up=1" "fuse_up=0" "fuse_up=1" 2>&1 | grep "^B="
done > gpurun_out/ab_up2.txt 2>&1
cat gpurun_out/ab_up2.txt
python bench.py --no-cpu-baseline --no-fp32-pipe --no-host-io 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = j['kernels']
print(j['ms_per_step'], j['b1']['ms_per_step'], j['roofline']['frac'], {n: k[n]['avg_us'] for n in k if 'lvc' in n}, j['roofline'].get('lvc_all_layers'), j['roofline'].get('lvc_first_layers_with_upsampler'))"
