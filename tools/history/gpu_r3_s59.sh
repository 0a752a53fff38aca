#!/bin/bash
# round-3 session 59: the end-of-step bookkeeping in the next step's first kernel (option fuse_advance): tests, A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bookkeeping or up_sampler or random_shapes or host_checked or hoisted or pipelined or n1000_at_64 or graph or embedding or sampler" > gpurun_out/pytest_adv2.txt 2>&1; tail -3 gpurun_out/pytest_adv2.txt
for B in 1 8; do
python tools/ab_opts.py --batch $B --reps 3 --steps 30 "fuse_advance=0" "fuse_advance=1" "fuse_advance=0" "fuse_advance=1" 2>&1 | grep "^B="
done > gpurun_out/ab_adv2.txt 2>&1
cat gpurun_out/ab_adv2.txt
