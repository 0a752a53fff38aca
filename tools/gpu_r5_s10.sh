#!/bin/bash
# Round 5, session 10 (record): the two PCIe legs of a micro-batch on copy streams of their own, A/B through FD_INFER_COPY_LANES.  The code
# lived in the commit before this note only (slower by 1 %: LABBOOK R5.7); this script needs that commit.
set -u
mkdir -p gpurun_out/r5s10
O=$GRAFT_REPO_ROOT/gpurun_out/r5s10
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
for v in 0 1 0 1; do
  FD_INFER_COPY_LANES=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline --no-fp32-pipe --no-b1 --no-torch-eager-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lanes=$v  configs1: device-resident %.4f ms, host to host %.4f ms (%s)' % (d['ms_per_step'], d['ms_per_step_host_to_host'], d['host_inclusive'].get('copy_lanes')))"
  FD_INFER_COPY_LANES=$v timeout 300 python bench.py --workload config4 --steps 5 --warmup 2 --project-ranks 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lanes=$v  config4: %.3f ms per job' % d['ms_per_step'])"
  FD_INFER_COPY_LANES=$v timeout 300 python bench.py --workload config5 --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lanes=$v  config5: %.3f ms per job' % d['ms_per_step'])"
done 2>&1 | tee $O/copy_lanes.txt
echo "== driver / sharding tests with the lanes on"; timeout 900 python -m pytest tests/test_infer_glue.py tests/test_sharded_synthesis.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "config5 or config4 or test_step or synthes" 2>&1 | tail -3
