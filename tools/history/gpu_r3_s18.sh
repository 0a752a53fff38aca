#!/bin/bash
# round-3 session 18: new tests (synthesize redo path), whole GPU suite once more
set -u
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -rx > gpurun_out/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; tail -5 gpurun_out/pytest_gpu.log | cut -c1-300
