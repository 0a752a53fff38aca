#!/bin/bash
# Round 5, session 3: cache policy of the other big streams on the same A/B harness (FD_LVC_NT bits: 1 x loads, 2 out stores (shipped), 8 skip
# loads, 16 hop-8 out stores, 64 first_conv out stores); the mel-bank test after its fix; the default bench line (its run time with the
# guarded all-cores CPU leg).
set -u
mkdir -p gpurun_out/r5s3
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5s3
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
(rocm-smi --showclocks --showpower --showperflevel --showmaxpower 2>&1 | grep -v "^=\|^$" | head -30) > $O/box_state.txt
echo "== nt policy of the LVC / first_conv streams (base = bit 2 only)"
LIB=fastdiff_amd/lib/libfastdiff_hip.so
cp $LIB /tmp/keep.so
i=0
for v in base lvc_nt82 lvc_nt11 lvc_nt91 base; do
  i=$((i+1))
  cp gpurun_ab/$v.so $LIB
  rm -rf /tmp/kt_$i
  (cd /tmp && FD_BENCH_CHILD=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$i -o kt -- python $R/bench.py --steps 6 --warmup 2 > /tmp/kt_$i.log 2>&1)
  ST=$(find /tmp/kt_$i -name '*kernel_stats.csv' | head -1)
  [ -n "$ST" ] && python tools/kstats.py $ST "$v#$i" || { echo "$v: no stats"; tail -3 /tmp/kt_$i.log; }
  [ -n "$ST" ] && grep "k_first_conv\|k_final_acc\|k_dblock_h2<4" $ST | awk -F, '{n=$1; sub(/\(.*/,"",n); printf "      %s avg %.1f us\n", n, $4/1000}'
done 2>&1 | tee $O/lvc_nt_policy.txt
cp /tmp/keep.so $LIB
echo "== the mel bank test"; timeout 600 python -m pytest tests/test_mel_frontend.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
echo "== bench (default line), wall time of the whole command"; T0=$(date +%s); FD_BENCH_KEEP_STATS=$O/bench_child_kernel_stats.csv timeout 1200 python bench.py > $O/bench.log 2>&1; echo "bench rc=$? in $(( $(date +%s) - T0 )) s"; grep '^{' $O/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['summary'])); print(json.dumps(d['cpu_baseline']))"
