#!/bin/bash
# Round 5, session 6: can x stay in the memory-side cache between two LVC layers?  Out stores WITHOUT nt (bit 2 off) with the skip loads
# streamed (bit 8) or not: 64 (no nt on out), 72 (= 64 | 8), 74 (= 64 | 8 | 2), against the shipped 66 (= 64 | 2).
set -u
mkdir -p gpurun_out/r5s6
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5s6
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
LIB=fastdiff_amd/lib/libfastdiff_hip.so
cp $LIB /tmp/keep.so
i=0
for v in base66 lvc_nt64 lvc_nt72 lvc_nt74 base66; do
  i=$((i+1))
  cp gpurun_ab/$v.so $LIB
  rm -rf /tmp/kt_$i
  (cd /tmp && FD_BENCH_CHILD=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$i -o kt -- python $R/bench.py --steps 6 --warmup 2 > /tmp/kt_$i.log 2>&1)
  ST=$(find /tmp/kt_$i -name '*kernel_stats.csv' | head -1)
  [ -n "$ST" ] && python tools/kstats.py $ST "$v#$i" || { echo "$v: no stats"; tail -3 /tmp/kt_$i.log; }
done 2>&1 | tee $O/lvc_nt_out_stores.txt
cp /tmp/keep.so $LIB
