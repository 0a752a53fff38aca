#!/bin/bash
# round 6, session 9: the extended bucketing test, the world-2 job without gather on the real vocoder, the C host
set -u
O=gpurun_out/r6s9; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -k "bucketed or without_gather or c_host or graph_cache" > $O/pytest_sel.log 2>&1; echo "rc=$?"; tail -5 $O/pytest_sel.log
