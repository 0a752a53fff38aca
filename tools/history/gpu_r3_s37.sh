#!/bin/bash
# round-3 session 37: where the training step's GPU time goes, by kernel family (rocprofv3 kernel trace of 12 steps, the first 4 left out)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trainprof; rm -rf $OUT; mkdir -p $OUT
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o train -- python $R/tools/train_step_profile.py 12 > $OUT/run.log 2>&1; cd $R
ls $OUT | head
f=$(ls $OUT/*kernel_trace.csv | head -1)
python tools/train_step_profile.py --report $f 12 > gpurun_out/train_step_families.txt 2>&1
rm -f $OUT/*kernel_trace.csv
cat gpurun_out/train_step_families.txt
