#!/bin/bash
# round-3 session 4: same virtual addresses in both processes?  (c_host with every allocation moved by 1.5 GiB; a second long-lived sampler loop)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
rm -f gpurun_out/xproc_hunt_fd.txt gpurun_out/xproc_*.npz
timeout 300 python tools/xproc_hunt.py 25 chost,chost_pad,chost,chost_pad fd 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu.ids" | tail -60
timeout 200 python tools/xproc_hunt.py 45 fdloop,fdloop_pad fd 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu.ids" | tail -40
