#!/bin/bash
# round-3 session 2: bisect the aggressor (what on a fresh process's way to its first call disturbs the victim), and a victim made of
# plain torch ops next to the same aggressors; then the new sharded-path GPU tests
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
rm -f gpurun_out/xproc_hunt_fd.txt gpurun_out/xproc_hunt_plain.txt gpurun_out/xproc_*.npz
timeout 500 python tools/xproc_hunt.py 30 torchinit,torchops,model,chost,firstcall_nograph,firstcall fd 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu.ids" | tail -40
timeout 200 python tools/xproc_hunt.py 30 chost,firstcall plain 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -10
echo "== sharded-path tests"
timeout 600 python -m pytest tests/test_sharded_synthesis.py tests/test_infer_glue.py tests/test_shard.py -m gpu -q -p no:cacheprovider 2>&1 | tail -8
