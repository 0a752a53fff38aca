#!/bin/bash
# round-3 session 20: hoisted predictor (all N steps' kernels predicted by one launch pair for small batches): parity tests, A/B
set -u
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
echo "== tests"; timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_infer_glue.py tests/test_c_host.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -6 | cut -c1-300
echo "== A/B"
for b in 1 2 4; do timeout 300 python tools/ab_opts.py --batch $b --steps 30 "hoist=off" "hoist=on" 2>&1 | grep "^B=\|config"; done | tee gpurun_out/ab_hoist.txt
timeout 300 python tools/ab_opts.py --batch 1 --nsteps 8 --steps 20 "hoist=off" "hoist=on" 2>&1 | grep "^B=\|config" | tee -a gpurun_out/ab_hoist.txt
