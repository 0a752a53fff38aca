#!/bin/bash
# round 4, session 25: kernel families of the training step after the frames path / fused predictor ops / skip fan-out; kconv forward step-major
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout 600 python -m pytest tests/test_lvc_op.py -m gpu -q -p no:cacheprovider -x -k "kernel_conv" 2>&1 | tail -2
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_train -o train -- python $R/tools/train_step_profile.py 12 > $R/gpurun_out/rocprof_train.log 2>&1; echo "rocprof rc=$?"
cd $R; KT=$(find gpurun_out/prof_train -name '*kernel_trace.csv' | head -1); python tools/train_step_profile.py --report $KT 12 > gpurun_out/train_step_families_s25.txt 2>&1; head -44 gpurun_out/train_step_families_s25.txt | cut -c1-150
find gpurun_out/prof_train -name '*.csv' -size +8M -delete 2>/dev/null
timeout 600 python tools/train_step_probe.py 2>&1 | grep "hipGraph\|frames:" | cut -c1-200
