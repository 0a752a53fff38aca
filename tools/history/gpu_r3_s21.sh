#!/bin/bash
# round-3 session 21: hoisted predictor at larger batches (where does it stop paying?)
set -u
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
for b in 8 16; do timeout 300 python tools/ab_opts.py --batch $b --steps 15 "hoist=off" "hoist=on" 2>&1 | grep "^B=\|config"; done | tee -a gpurun_out/ab_hoist.txt
timeout 300 python tools/ab_opts.py --batch 3 --steps 30 "hoist=off" "hoist=on" 2>&1 | grep "^B=" | tee -a gpurun_out/ab_hoist.txt
