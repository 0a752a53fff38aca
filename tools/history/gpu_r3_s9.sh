#!/bin/bash
# round-3 session 9: fused gate operator (training path): parity + the training step's time and kernel profile; fixed tests; config 4 projection
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
echo "== tests"; timeout 900 python -m pytest tests/test_lvc_op.py tests/test_training_path.py tests/test_gpu_parity.py tests/test_sharded_synthesis.py -m gpu -q -p no:cacheprovider -k "gate or training or hands_over or rccl or validation_loss" 2>&1 | tail -4 | cut -c1-300
echo "== training step"; timeout 600 python tools/train_step_probe.py 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu.ids" | tee gpurun_out/train_step_probe.txt
echo "== training step kernel profile"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train -o kt -- python $R/tools/train_step_probe.py > $R/gpurun_out/rocprof_train.log 2>&1 ; echo "rocprof rc=$?"
cd $R; find gpurun_out/prof_train -name '*kernel_trace.csv' -delete 2>/dev/null
head -25 gpurun_out/prof_train/kt_kernel_stats.csv | cut -c1-150
echo "== config4"; timeout 600 python bench.py --workload config4 --steps 5 --warmup 2 > gpurun_out/bench_config4.log 2>&1 ; python - <<'PY'
import json
for line in open('gpurun_out/bench_config4.log'):
    if line.startswith('{'):
        d = json.loads(line); print(d['value'], d['ms_per_step'], d['projection'])
PY
