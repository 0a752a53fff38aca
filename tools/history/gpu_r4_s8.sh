#!/bin/bash
# round 4, session 8: which part of the library's sampler makes it the only victim?  The sampler next to the generic aggressor
# (tools/ubench/xproc_repro: none of this library's code), with the victim's kernels swapped stage by stage (FD_HUNT_OPTS)
set -u
mkdir -p gpurun_out/s8
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s8
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
run() { echo "== victim options: [$1] (padded batch, no lens)"; FD_HUNT_NOLENS=1 FD_HUNT_OPTS="$1" timeout 300 python $R/tools/xproc_hunt.py 18 generic fd 2>&1 | grep "^victim\|Error\|error" | tail -2 ; }
{
run "first_variant=1"
run "first_variant=2"
run "first_variant=3"
run "first_variant=0"
} 2>&1 | tee $O/xproc_victim_first_conv_variants.txt
