#!/bin/bash
# Round 5, session 1: the pruned tree through the GPU tests; the pair-fusion bound probes; the predictor GEMM's store policy against the
# hop-8 layers behind it; power / clock traces of the GEMM, the hop-256 layer and the whole step.
set -u
mkdir -p gpurun_out/r5s1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5s1
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
(rocm-smi --showclocks --showpower --showperflevel --showmaxpower 2>&1 | grep -v "^=\|^$" | head -30) > $O/box_state.txt
echo "== pair-fusion bound (tools/ubench/lvc_h2_bench: full / no x+skip loads / no stores / neither), B=8 then B=1"
for b in 8 1; do for v in "" _NOLOADX _NOSTORE _NOLOADX_NOSTORE; do echo "-- lvc_h2_bench$v B=$b"; timeout 120 tools/ubench/lvc_h2_bench$v $b 864; done; done 2>&1 | tee $O/pair_bound.txt
echo "== GEMM store policy vs the hop-8 layers (rocprofv3 kernel trace of the replayed step, B=8)"
LIB=fastdiff_amd/lib/libfastdiff_hip.so
cp $LIB /tmp/keep.so
for rep in 1; do
for v in base gemm_aux2 gemm_aux17 gemm_b0aux2 gemm_b0aux17 gemm_b0aux19 base; do
  cp gpurun_ab/$v.so $LIB
  rm -rf /tmp/kt_$v
  (cd /tmp && FD_BENCH_CHILD=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$v -o kt -- python $R/bench.py --steps 6 --warmup 2 > /tmp/kt_$v.log 2>&1)
  ST=$(find /tmp/kt_$v -name '*kernel_stats.csv' | head -1)
  [ -n "$ST" ] && python tools/kstats.py $ST "$v#$rep" || { echo "$v: no stats"; tail -3 /tmp/kt_$v.log; }
done; done 2>&1 | tee $O/gemm_store_policy.txt
cp /tmp/keep.so $LIB
echo "== power / clock traces"
python tools/power_trace.py $O/power_gemm.csv -- tools/ubench/gemm_h2_bench 8 864 512 4000 2>&1 | tee $O/power_gemm.txt
python tools/power_trace.py $O/power_lvc_h256.csv -- tools/ubench/lvc_h2_bench 8 864 8000 2>&1 | tee $O/power_lvc_h256.txt
python tools/power_trace.py $O/power_step.csv -- python bench.py --steps 300 --warmup 3 --no-roofline --no-cpu-baseline --no-b1 --no-fp32-pipe --no-host-io 2>&1 | grep -v "^{" | tee $O/power_step.txt
echo "== pytest gpu" ; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; tail -6 $O/pytest_gpu.log
