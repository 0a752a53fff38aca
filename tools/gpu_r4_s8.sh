#!/bin/bash
# round 4, session 8: which part of the library's sampler makes it the only victim?  The sampler next to the generic aggressor
# (tools/ubench/xproc_repro: none of this library's code), with the victim's kernels swapped stage by stage (FD_HUNT_OPTS)
set -u
mkdir -p gpurun_out/s8
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s8
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
run() { echo "== victim options: [$1]"; FD_HUNT_OPTS="$1" timeout 300 python $R/tools/xproc_hunt.py 18 generic fd 2>&1 | grep "^victim" ; }
{
run ""
run "kernels=naive"
run "kernels.lvc=naive"
run "kernels.kp_gemm=naive,kernels.kp_front=naive"
run "kernels.dblock=naive,kernels.convt=naive,kernels.first=naive,kernels.final=naive"
run "fallback=graph,hoist=off,fuse_up=0,fuse_final=0,fuse_advance=0,embed_cache=0"
run "graph=0"
} 2>&1 | tee $O/xproc_victim_bisect.txt
