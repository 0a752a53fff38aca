#!/bin/bash
# round 4, session 12: the new concurrent two-rank test (shared compute units, no masks, no turns) ten times
set -u
mkdir -p gpurun_out/s12
O=$GRAFT_REPO_ROOT/gpurun_out/s12
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 300 python -m pytest tests/test_sharded_synthesis.py -m gpu -q -p no:cacheprovider -k "concurrent_on_shared" 2>&1 | tail -1
done | tee $O/concurrent_shared_cus_10_runs.txt
