#!/bin/bash
# Round 5, session 7: `lens` on the generic-configuration kernels; the whole GPU suite once more on another box (flakiness check).
set -u
mkdir -p gpurun_out/r5s7
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s7
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
echo "== generic-configuration tests"; timeout 600 python -m pytest tests/test_generic_config.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v "Warn\|warn" | tail -12
echo "== pytest gpu (all)"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
