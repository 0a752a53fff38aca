#!/bin/bash
# round 4, session 11: standalone: victim and aggressor BOTH read a buffer with scalar loads, at the same virtual address, different contents
set -u
mkdir -p gpurun_out/s11
O=$GRAFT_REPO_ROOT/gpurun_out/s11
X=$GRAFT_REPO_ROOT/tools/ubench
{
echo "== addresses"; $X/xproc_repro_scalar aggressor 1 | head -1; timeout 20 $X/xproc_repro_scalar victim 1 | head -2
echo "== scalar victim 60 s next to short-lived scalar aggressors (same VA, other contents)"
( end=$((SECONDS+62)); n=0; while [ $SECONDS -lt $end ]; do $X/xproc_repro_scalar aggressor 40 > /dev/null 2>&1; n=$((n+1)); done; echo "aggressor processes run: $n" ) &
AG=$!; timeout 180 $X/xproc_repro_scalar victim 60 | tail -8; wait $AG
echo "== scalar victim 40 s next to ONE long-lived scalar aggressor"
timeout 45 $X/xproc_repro_scalar aggressor 100000000 > /dev/null 2>&1 &
AG=$!; sleep 2; timeout 120 $X/xproc_repro_scalar victim 40 | tail -8; kill $AG 2>/dev/null; wait $AG 2>/dev/null
} 2>&1 | tee $O/xproc_scalar_same_va.txt
