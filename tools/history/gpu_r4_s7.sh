#!/bin/bash
# round 4, session 7: the standalone victim with (nearly) the full register budget (238 VGPRs per wave, like the library's LVC kernels)
set -u
mkdir -p gpurun_out/s7
O=$GRAFT_REPO_ROOT/gpurun_out/s7
X=$GRAFT_REPO_ROOT/tools/ubench
{
echo "== fat victim (238 VGPRs) alone, 20 s"; timeout 100 $X/xproc_repro_fat victim 20
echo "== fat victim 60 s next to short-lived lean aggressors"
( end=$((SECONDS+62)); n=0; while [ $SECONDS -lt $end ]; do $X/xproc_repro aggressor 40 > /dev/null 2>&1; n=$((n+1)); done; echo "aggressor processes run: $n" ) &
AG=$!; timeout 180 $X/xproc_repro_fat victim 60; wait $AG
echo "== fat victim 40 s next to short-lived fat aggressors"
( end=$((SECONDS+42)); n=0; while [ $SECONDS -lt $end ]; do $X/xproc_repro_fat aggressor 40 > /dev/null 2>&1; n=$((n+1)); done; echo "aggressor processes run: $n" ) &
AG=$!; timeout 180 $X/xproc_repro_fat victim 40; wait $AG
} 2>&1 | tee $O/xproc_repro_fat.txt
