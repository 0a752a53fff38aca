#!/bin/bash
# round-2 session 14: the two-processes-on-one-GPU test in a loop (fresh processes every time), diagnostics on a mismatch.
# usage: gpu_r2_s14.sh RUNS [OPTS] [STOP_AT_FAILURES]   OPTS = library options for every test model (FD_TEST_OPTS), e.g. graph=0
set -u
mkdir -p gpurun_out
N=${1:-14}
export FD_TEST_OPTS=${2:-}
export FD_TEST_SERIALIZE=${4:-1}
STOP=${3:-1000}
fail=0
for i in $(seq 1 $N); do
  timeout 300 python -m pytest tests/test_sharded_synthesis.py -m gpu -q -s -p no:cacheprovider -k hip_vocoder > gpurun_out/shard_loop_$i.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then fail=$((fail+1)); echo "run $i rc=$rc"; grep -a "MISMATCH" gpurun_out/shard_loop_$i.log | cut -c1-500; else rm -f gpurun_out/shard_loop_$i.log; fi
  if [ $fail -ge $STOP ]; then echo "stopping after $i runs"; break; fi
done
echo "opts '$FD_TEST_OPTS': runs $i failures $fail" | tee -a gpurun_out/shard_loop_summary.txt
