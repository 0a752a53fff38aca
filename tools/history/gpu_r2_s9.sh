#!/bin/bash
# round-2 session 9: the denoiser under autograd (fastdiff_amd/train.py): parity of the training step, timing against unfold+einsum
set -u
mkdir -p gpurun_out
echo "== pytest (training path, theta loss)"
timeout 900 python -m pytest tests/test_training_path.py tests/test_lvc_op.py "tests/test_gpu_parity.py::test_validation_loss_on_the_hip_denoiser" -m gpu -q -s -p no:cacheprovider > gpurun_out/pytest_train.log 2>&1; echo "pytest rc=$?"
grep -a "passed\|failed\|worst\|loss \|^E " gpurun_out/pytest_train.log | cut -c1-400 | head -30
echo "== probe"; timeout 600 python tools/train_step_probe.py 2>&1 | grep -v Warning | tee gpurun_out/train_step_probe.txt
