#!/bin/bash
# one PMC pass (FETCH_SIZE, WRITE_SIZE need separate passes; here FETCH only) over a short eager bench run
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmcq; rm -rf $OUT; mkdir -p $OUT; cd /tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-graph"
timeout 600 rocprofv3 --kernel-trace --pmc ${1:-FETCH_SIZE} --output-format csv -d $OUT -o p2 -- $CMD > $OUT/p2.log 2>&1; echo "rc=$?"
cp $OUT/p2_kernel_trace.csv $OUT/p1_kernel_trace.csv 2>/dev/null
cd $R; python tools/pmc_summary.py $OUT | grep -v "at::native\|rocclr" | head -8
