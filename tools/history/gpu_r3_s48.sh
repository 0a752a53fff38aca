#!/bin/bash
# round-3 session 48: fuse_up with the weights requested before the barrier: tests, A/B at B=1 / 2 / 4 / 8, per-kernel stats at B=1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "up_sampler or random_shapes or host_checked or hoisted" > gpurun_out/pytest_up.txt 2>&1; tail -3 gpurun_out/pytest_up.txt
for B in 1 2 4 8; do
python tools/ab_opts.py --batch $B --reps 3 --steps 30 "fuse_up=off" "fuse_up=on" "fuse_up=off" "fuse_up=on" 2>&1 | grep "^B="
done > gpurun_out/ab_up.txt 2>&1
cat gpurun_out/ab_up.txt
bash tools/history/gpu_r3_s47.sh > /dev/null 2>&1
cat gpurun_out/up_kernel_stats_b1.txt | cut -c1-150
