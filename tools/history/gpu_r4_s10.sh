#!/bin/bash
# round 4, session 10: which difference between the two forms of k_first_conv matters -- scalar data loads, or the LDS write + barrier in front?
set -u
mkdir -p gpurun_out/s10
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s10
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
run() { echo "== victim options: [$1]"; FD_HUNT_OPTS="$1" timeout 300 python $R/tools/xproc_hunt.py 18 generic fd 2>&1 | grep "^victim\|rror" | tail -2 ; }
{
run "first_variant=4"
run "first_variant=5"
run "first_variant=0"
run "first_variant=1"
} 2>&1 | tee $O/xproc_first_conv_forms.txt
