#!/bin/bash
# round-3 session 45: the ConvTranspose of blocks 1 and 2 inside their first LVC layer: bit-equality with the separate kernel, then the step A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "up_sampler or random_shapes or host_checked or hoisted or ragged_batch_with_lens" > gpurun_out/pytest_up.txt 2>&1; tail -12 gpurun_out/pytest_up.txt
python tools/ab_opts.py --batch 8 --reps 3 --steps 20 "fuse_up=off" "fuse_up=on" "fuse_up=off" "fuse_up=on" > gpurun_out/ab_up_b8.txt 2>&1; tail -4 gpurun_out/ab_up_b8.txt
python tools/ab_opts.py --batch 1 --reps 3 --steps 40 "fuse_up=off" "fuse_up=on" "fuse_up=off" "fuse_up=on" > gpurun_out/ab_up_b1.txt 2>&1; tail -4 gpurun_out/ab_up_b1.txt
