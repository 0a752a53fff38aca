#!/bin/bash
# round 4, session 9: (a) the library's sampler (default build: first conv weights through LDS) next to the generic aggressor, (b) the
# standalone victim WITH scalar data loads next to the generic aggressor, (c) the two-rank concurrent test without CU masks
set -u
mkdir -p gpurun_out/s9
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s9
X=$R/tools/ubench
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
{
echo "== (a) the library's sampler, default build, 40 s next to the generic aggressor; then first_variant=0 (round 1-3 form) 18 s as the control"
timeout 300 python $R/tools/xproc_hunt.py 40 generic fd 2>&1 | grep "^victim"
FD_HUNT_OPTS="first_variant=0" timeout 300 python $R/tools/xproc_hunt.py 18 generic fd 2>&1 | grep "^victim"
echo "== (a2) default build next to examples/c_host (round 3's trigger) and fresh Python vocoders, 25 s each"
timeout 400 python $R/tools/xproc_hunt.py 25 chost,firstcall fd 2>&1 | grep "^victim"
echo "== (b) standalone victim with 256 scalar-loaded weights (xproc_repro -DSCALAR_W): alone 15 s, then 60 s next to short-lived lean aggressors"
timeout 100 $X/xproc_repro_scalar victim 15
( end=$((SECONDS+62)); n=0; while [ $SECONDS -lt $end ]; do $X/xproc_repro aggressor 40 > /dev/null 2>&1; n=$((n+1)); done; echo "aggressor processes run: $n" ) &
AG=$!; timeout 180 $X/xproc_repro_scalar victim 60; wait $AG
} 2>&1 | tee $O/xproc_after_fix.txt
