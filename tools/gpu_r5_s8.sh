#!/bin/bash
# Round 5, session 8 (record): the predictor front on a second stream next to the DBlocks (FD_EXP_FRONT_FORK=1), A/B inside one session, and the
# parity suite with the fork on.  The experiment code lived in commit ed8f950 only (+0.6 %: LABBOOK R5.6); this script needs that commit.
set -u
mkdir -p gpurun_out/r5s8
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5s8
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
i=0
for v in 0 1 0 1; do
  i=$((i+1))
  rm -rf /tmp/kt_$i
  (cd /tmp && FD_EXP_FRONT_FORK=$v FD_BENCH_CHILD=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$i -o kt -- python $R/bench.py --steps 6 --warmup 2 > /tmp/kt_$i.log 2>&1)
  ST=$(find /tmp/kt_$i -name '*kernel_stats.csv' | head -1)
  [ -n "$ST" ] && python tools/kstats.py $ST "fork=$v#$i" || { echo "fork=$v: no stats"; tail -3 /tmp/kt_$i.log; }
done 2>&1 | cut -c1-200 | tee $O/front_fork.txt
echo "== wall clock per sample call (what the kernel sums cannot show for overlapped kernels)"
for v in 0 1 0 1; do
  FD_EXP_FRONT_FORK=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline --no-fp32-pipe --no-host-io --no-torch-eager-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fork=$v  B=8 ms_per_step %.4f   b1 %.4f' % (d['ms_per_step'], d['b1']['ms_per_step']))"
done 2>&1 | tee -a $O/front_fork.txt
echo "== parity suite with the fork on"; FD_EXP_FRONT_FORK=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider 2>&1 | tail -3
