#!/bin/bash
# round-3 session 50: k_advance inside the step's last kernel, embedding rows kept between calls: tests, then A/B at B=1 and B=8
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bookkeeping or up_sampler or random_shapes or host_checked or hoisted or pipelined or n1000 or graph" > gpurun_out/pytest_adv.txt 2>&1; tail -4 gpurun_out/pytest_adv.txt
for B in 1 8; do
python tools/ab_opts.py --batch $B --reps 3 --steps 30 "fuse_advance=0" "fuse_advance=1" "fuse_advance=0" "fuse_advance=1" 2>&1 | grep "^B="
done > gpurun_out/ab_adv.txt 2>&1
cat gpurun_out/ab_adv.txt
