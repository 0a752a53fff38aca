#!/bin/bash
# round-3 session 3: what in the aggressor (code-object load alone? any code object?) and what in the victim (fp16 LVC kernel? CU sharing?)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
rm -f gpurun_out/xproc_hunt_fd.txt gpurun_out/xproc_*.npz
F="grep -v Warning\|WeightNorm\|amdgpu.ids\|iteration"
timeout 200 python tools/xproc_hunt.py 25 chost,modload,othermod fd 2>&1 | $F | tail -5
FD_HUNT_OPTS="lvc=fp32" timeout 120 python tools/xproc_hunt.py 25 chost fd 2>&1 | $F | tail -3
FD_HUNT_OPTS="graph=0" timeout 120 python tools/xproc_hunt.py 25 chost fd 2>&1 | $F | tail -3
echo "== disjoint CU masks"
HSA_CU_MASK="0:0-127" XPROC_AGG_ENV="HSA_CU_MASK=0:128-255" timeout 120 python tools/xproc_hunt.py 30 idle,chost fd 2>&1 | $F | tail -3
echo "== 4 vs 8 waves per workgroup at B=1 (round-2 harness, packed records)"
tools/ubench/lvc_h2_bench_r2w8 1 864 1 4 | grep "lvc<"
tools/ubench/lvc_h2_bench_r2w8 1 864 1 8 | grep "lvc<"
tools/ubench/lvc_h2_bench_r2w8 2 864 1 4 | grep "lvc<"
tools/ubench/lvc_h2_bench_r2w8 2 864 1 8 | grep "lvc<"
