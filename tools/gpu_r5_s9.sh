#!/bin/bash
# Round 5, session 9: micro-batch size of the one-GPU configs[3] job (64 ragged utterances, N=6, host to host).
set -u
mkdir -p gpurun_out/r5s9
O=$GRAFT_REPO_ROOT/gpurun_out/r5s9
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
for b in 8 16 32 8 16; do
  timeout 300 python bench.py --workload config4 --batch $b --steps 5 --warmup 2 --project-ranks 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('micro-batch $b: %.3f ms per job, %.0fx' % (d['ms_per_step'], d['value']))"
done 2>&1 | tee $O/config4_microbatch.txt
