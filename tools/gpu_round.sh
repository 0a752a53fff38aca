#!/bin/bash
# One GPU-box session: diagnostics -> tests -> smoke -> bench -> rocprof.  Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
echo "== diag" ; timeout 900 python tools/gpu_diag.py > gpurun_out/diag.log 2>&1 ; echo "diag rc=$?" ; tail -25 gpurun_out/diag.log
echo "== pytest gpu" ; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; tail -40 gpurun_out/pytest_gpu.log
echo "== smoke" ; timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 ; echo "smoke rc=$?" ; tail -3 gpurun_out/smoke.log
echo "== bench" ; timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.log 2>&1 ; echo "bench rc=$?" ; tail -5 gpurun_out/bench.log
echo "== rocprof kernel-trace"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-roofline --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1 ; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT; ls -R gpurun_out/prof 2>/dev/null | head -20
f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -40 "$f"
# keep only the small summaries (the merge-back limit is 64 MiB)
find gpurun_out/prof -name '*kernel_trace.csv' -size +20M -delete 2>/dev/null
