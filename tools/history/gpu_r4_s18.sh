#!/bin/bash
# round 4, session 18: kernel_conv -> LVC operator through frame-major tensors (fd_kconv_*_frames, fd_lvc_*_frames): tests, training step A/B
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
echo "== tests"; timeout 1200 python -m pytest tests/test_lvc_op.py tests/test_training_path.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -15
echo "== training step"; timeout 600 python tools/train_step_probe.py 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu.ids" | head -8 | tee gpurun_out/train_step_probe_s18.txt
echo "== families"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_train -o train -- python $R/tools/train_step_profile.py 12 > $R/gpurun_out/rocprof_train.log 2>&1; echo "rocprof rc=$?"
cd $R; KT=$(find gpurun_out/prof_train -name '*kernel_trace.csv' | head -1); python tools/train_step_profile.py --report $KT 12 > gpurun_out/train_step_families_s18.txt 2>&1; head -40 gpurun_out/train_step_families_s18.txt | cut -c1-150
find gpurun_out/prof_train -name '*.csv' -size +8M -delete 2>/dev/null
