#!/bin/bash
# round-3 session 31: the predictor (front + GEMM) in front of the down path: do the first hop-8 layers stop paying for the GEMM's dirty lines?
mkdir -p gpurun_out
python tools/ab_opts.py --batch 8 --reps 4 --steps 20 "" "order=predictor" > gpurun_out/ab_order.txt 2>&1
for o in down predictor; do
  python bench.py --no-cpu-baseline --no-fp32-pipe --no-b1 --no-host-io --opt order=$o > gpurun_out/bench_order_$o.json 2>/dev/null
done
tail -4 gpurun_out/ab_order.txt
