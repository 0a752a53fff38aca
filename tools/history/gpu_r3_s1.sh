#!/bin/bash
# round-3 session 1: which activity of a second process disturbs a vocoding process on the same GPU (tools/xproc_hunt.py)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
(rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|power (W)" | head -4) > gpurun_out/box_state.txt
rm -f gpurun_out/xproc_hunt_fd.txt gpurun_out/xproc_hunt_plain.txt
timeout 400 python tools/xproc_hunt.py 25 idle,kernels,hostalloc,devalloc,procs,firstcall fd 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu.ids" | tail -40
timeout 120 python tools/xproc_hunt.py 15 devalloc,procs plain 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -10
