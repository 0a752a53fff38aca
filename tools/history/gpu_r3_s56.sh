#!/bin/bash
# round-3 session 56: block 0's ConvTranspose inside the last DBlock: tests, A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "up_sampler or random_shapes or host_checked or hoisted or ragged_batch_with_lens or embedding or n1000_at_64" > gpurun_out/pytest_up3.txt 2>&1; tail -3 gpurun_out/pytest_up3.txt
for B in 1 8; do
python tools/ab_opts.py --batch $B --reps 3 --steps 30 "fuse_up=0" "fuse_up=1" "fuse_up=0" "fuse_up=1" 2>&1 | grep "^B="
done > gpurun_out/ab_up3.txt 2>&1
cat gpurun_out/ab_up3.txt
