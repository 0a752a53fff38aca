#!/bin/bash
# rocprofv3 PMC passes (kernel-trace only; never combined with sys/hip traces) for the bench workload.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-graph"
rocprofv3 -L > $OUT/counters.txt 2>&1
pass() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -o $name -- $CMD > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
pass p1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass p2 FETCH_SIZE
pass p3 WRITE_SIZE
pass p4 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA
ls -la $OUT | head -30
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt | head -80
find $OUT -name '*.csv' -size +8M -delete
