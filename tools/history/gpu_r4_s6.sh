#!/bin/bash
# round 4, session 6: (a) the standalone two-process probe (no torch, none of this library), (b) LDS bank conflicts of the fused
# first layers, round-3 form against round-4 form (one PMC pass each), (c) B=1 option A/Bs
set -u
mkdir -p gpurun_out/s6
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s6
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
X=$R/tools/ubench/xproc_repro
{
echo "== (a1) victim alone, 30 s"; timeout 120 $X victim 30
echo "== (a2) victim 60 s next to a loop of short-lived aggressors (same kernel, 40 launches each, then exit)"
( end=$((SECONDS+62)); n=0; while [ $SECONDS -lt $end ]; do $X aggressor 40 > /dev/null 2>&1; n=$((n+1)); done; echo "aggressor processes run: $n" ) &
AG=$!
timeout 180 $X victim 60
wait $AG
echo "== (a3) the same generic victim, 40 s, next to fresh processes that run THIS LIBRARY's sampler once and exit (xproc_hunt firstcall)"
timeout 100 python $R/tools/xproc_hunt.py --aggressor firstcall > /dev/null 2>&1 &
AG=$!
sleep 8
timeout 180 $X victim 40
kill $AG 2>/dev/null; wait $AG 2>/dev/null
echo "== (a4) this library's sampler as the victim next to the generic aggressors, and (a5) next to examples/c_host (round 3's trigger), 25 s each"
timeout 400 python $R/tools/xproc_hunt.py 25 idle,generic,chost fd 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu.ids" | tail -12
} 2>&1 | tee $O/xproc_repro.txt
echo "== (b) PMC: LDS bank conflicts"
cd /tmp
for v in 1 0; do
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $O/pmc_v$v -o p4 -- python $R/bench.py --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-graph --no-fp32-pipe --no-host-io --no-b1 --opt lvc_variant=$v > $O/pmc_v$v.log 2>&1; echo "pmc variant $v rc=$?"
  (cd $R && python tools/pmc_summary.py $O/pmc_v$v 2>&1 | grep "lvc_\|kp_gemm" ) | tee $O/pmc_lds_variant$v.txt
  find $O/pmc_v$v -name '*.csv' -size +8M -delete 2>/dev/null
done
cd $R
