#!/bin/bash
# round-3 session 15: the training step replayed from a hipGraph
set -u
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout 600 python tools/train_step_probe.py 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu.ids" | grep -v "run_backward\|Consider using\|loss_graph = " | head -8 | tee gpurun_out/train_step_probe.txt
