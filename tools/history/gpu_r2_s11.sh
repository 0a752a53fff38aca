#!/bin/bash
# round-2 session 11: LVC operator kernels after float4 staging / fused frame edges / chunk prefetch: parity, probe, per-kernel times
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_training_path.py tests/test_lvc_op.py -m gpu -q -s -p no:cacheprovider > gpurun_out/pytest_train.log 2>&1; echo "pytest rc=$?"
grep -a "passed\|failed\|worst\|loss \|^E " gpurun_out/pytest_train.log | cut -c1-400 | head -30
echo "== probe"; timeout 600 python tools/train_step_probe.py 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu.ids" | tee gpurun_out/train_step_probe.txt
bash tools/gpu_r2_s10.sh
