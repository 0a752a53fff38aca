#!/bin/bash
# Round 5, session 12 (record): the hop-8 LVC layer at four workgroups per CU (probe macro FD_H8M_OCC, in the history only: commits 79c419f, 8cea3b3;
# LABBOOK R5.8).
set -u
mkdir -p gpurun_out/r5s12
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5s12
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
LIB=fastdiff_amd/lib/libfastdiff_hip.so
cp $LIB /tmp/keep.so
i=0
for v in base h8_occ4 base h8_occ4; do
  i=$((i+1))
  cp gpurun_ab/$v.so $LIB
  rm -rf /tmp/kt_$i
  (cd /tmp && FD_BENCH_CHILD=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$i -o kt -- python $R/bench.py --steps 6 --warmup 2 > /tmp/kt_$i.log 2>&1)
  ST=$(find /tmp/kt_$i -name '*kernel_stats.csv' | head -1)
  [ -n "$ST" ] && python tools/kstats.py $ST "$v#$i" || { echo "$v: no stats"; tail -3 /tmp/kt_$i.log; }
done 2>&1 | cut -c1-150 | tee $O/h8_occupancy.txt
echo "== B=1 wall clock"
for v in base h8_occ4 base h8_occ4; do
  cp gpurun_ab/$v.so $LIB
  timeout 300 python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline --no-fp32-pipe --no-host-io --no-torch-eager-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v  B=8 ms_per_step %.4f   b1 %.4f' % (d['ms_per_step'], d['b1']['ms_per_step']))"
done 2>&1 | tee -a $O/h8_occupancy.txt
cp /tmp/keep.so $LIB
