#!/bin/bash
# round 4, session 3: the small-convolution operator of the training path (conv32): operator tests, the training step against the
# reference's gradients, step time, kernel families of a steady-state step
set -u
mkdir -p gpurun_out/s15
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s15
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
echo "== tests"; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -s -k "upsample or slices or conv7 or conv32 or training or hip_operator or eval_mode or kernel_conv or weight_norm or gate_operator" > $O/pytest.log 2>&1; echo "rc=$?"; tail -15 $O/pytest.log | cut -c1-400
echo "== training step probe"; timeout 600 python tools/train_step_probe.py 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu.ids" | head -6 | tee $O/train_step_probe.txt
echo "== kernel families of a steady-state step"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o train -- python $R/tools/train_step_profile.py 12 > $O/rocprof_train.log 2>&1; echo "rocprof rc=$?"
cd $R; KT=$(find $O/prof -name '*kernel_trace.csv' | head -1); python tools/train_step_profile.py --report $KT 12 2>&1 | tee $O/train_step_families.txt | head -60
find $O/prof -name '*.csv' -size +8M -delete 2>/dev/null
