#!/bin/bash
# round-3 session 32: order=predictor against order=down again, two model instances each, alternating (instance-to-instance spread?)
mkdir -p gpurun_out
python tools/ab_opts.py --batch 8 --reps 4 --steps 20 "order=down" "order=predictor" "order=down" "order=predictor" > gpurun_out/ab_order2.txt 2>&1
tail -5 gpurun_out/ab_order2.txt
