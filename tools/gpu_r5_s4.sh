#!/bin/bash
# Round 5, session 4: which of the two extra nt bits of session 3 pays -- 64 (first_conv's a0 stores) or 16 (the hop-8 layers' out stores) --
# on another box: base (bit 2), 2|64, 2|16, 2|16|64, alternated.
set -u
mkdir -p gpurun_out/r5s4
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5s4
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
(rocm-smi --showclocks --showpower --showperflevel --showmaxpower 2>&1 | grep -v "^=\|^$" | head -30) > $O/box_state.txt
LIB=fastdiff_amd/lib/libfastdiff_hip.so
cp $LIB /tmp/keep.so
i=0
for v in base lvc_nt66 lvc_nt18 lvc_nt82 base lvc_nt66 lvc_nt82; do
  i=$((i+1))
  cp gpurun_ab/$v.so $LIB
  rm -rf /tmp/kt_$i
  (cd /tmp && FD_BENCH_CHILD=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$i -o kt -- python $R/bench.py --steps 6 --warmup 2 > /tmp/kt_$i.log 2>&1)
  ST=$(find /tmp/kt_$i -name '*kernel_stats.csv' | head -1)
  [ -n "$ST" ] && python tools/kstats.py $ST "$v#$i" || { echo "$v: no stats"; tail -3 /tmp/kt_$i.log; }
done 2>&1 | tee $O/lvc_nt_policy_bits.txt
cp /tmp/keep.so $LIB
