#!/bin/bash
# Round 5, session 13 (record): load order of the hop-8 LVC layer (probe macro FD_H8M_XFIRST, in the history only; LABBOOK R5.8).
set -u
mkdir -p gpurun_out/r5s13
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5s13
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
LIB=fastdiff_amd/lib/libfastdiff_hip.so
cp $LIB /tmp/keep.so
i=0
for v in base h8_x1 h8_x2 h8_x2o3 base h8_x1; do
  i=$((i+1))
  cp gpurun_ab/$v.so $LIB
  rm -rf /tmp/kt_$i
  (cd /tmp && FD_BENCH_CHILD=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$i -o kt -- python $R/bench.py --steps 6 --warmup 2 > /tmp/kt_$i.log 2>&1)
  ST=$(find /tmp/kt_$i -name '*kernel_stats.csv' | head -1)
  [ -n "$ST" ] && python tools/kstats.py $ST "$v#$i" || { echo "$v: no stats"; tail -3 /tmp/kt_$i.log; }
done 2>&1 | cut -c1-150 | tee $O/h8_load_order.txt
echo "== parity of the x1 build (hop-8 stage tests, forward, ragged)"
cp gpurun_ab/h8_x1.so $LIB
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "stage or forward or ragged or lens or batch" 2>&1 | tail -3
cp /tmp/keep.so $LIB
