#!/bin/bash
# round-3 session 19: more statistics for the CU-mask finding (c_host aggressor; control without masks, then 120 s on disjoint masks, then masks swapped)
set -u
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
rm -f gpurun_out/xproc_hunt_fd.txt gpurun_out/xproc_*.npz
filt() { grep -v "Warning\|WeightNorm\|amdgpu.ids\|iteration" | tail -2; }
timeout 200 python tools/xproc_hunt.py 40 chost fd 2>&1 | filt
HSA_CU_MASK="0:0-127" XPROC_AGG_ENV="HSA_CU_MASK=0:128-255" timeout 300 python tools/xproc_hunt.py 120 chost fd 2>&1 | filt
HSA_CU_MASK="0:128-255" XPROC_AGG_ENV="HSA_CU_MASK=0:0-127" timeout 200 python tools/xproc_hunt.py 60 chost fd 2>&1 | filt
cat gpurun_out/xproc_hunt_fd.txt
