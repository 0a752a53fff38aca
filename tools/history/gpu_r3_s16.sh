#!/bin/bash
# round-3 session 16: option overlap = paths (down path next to the predictor) at B=1, 2, 8
set -u
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
for b in 1 2 8; do timeout 300 python tools/ab_opts.py --batch $b --steps 30 "" "overlap=paths" 2>&1 | grep "^B=\|config"; done | tee gpurun_out/ab_overlap_paths.txt
