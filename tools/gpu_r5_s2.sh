#!/bin/bash
# Round 5, session 2: nt stores of the predictor GEMM against the old policy on a second box; the new bench line (summary scalars, CPU
# recipe, torch-eager row), BASELINE configs[4] as a bench line, eight ranks through the sharded code path on this one GPU; GPU tests.
set -u
mkdir -p gpurun_out/r5s2
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5s2
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
(rocm-smi --showclocks --showpower --showperflevel --showmaxpower 2>&1 | grep -v "^=\|^$" | head -30) > $O/box_state.txt
echo "== GEMM store policy, second box (aux0 = round 4's stores, base_aux2 = nt, the shipped default)"
LIB=fastdiff_amd/lib/libfastdiff_hip.so
cp $LIB /tmp/keep.so
i=0
for v in gemm_aux0 base_aux2 gemm_aux3 gemm_aux18 gemm_aux0 base_aux2; do
  i=$((i+1))
  cp gpurun_ab/$v.so $LIB
  rm -rf /tmp/kt_$i
  (cd /tmp && FD_BENCH_CHILD=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$i -o kt -- python $R/bench.py --steps 6 --warmup 2 > /tmp/kt_$i.log 2>&1)
  ST=$(find /tmp/kt_$i -name '*kernel_stats.csv' | head -1)
  [ -n "$ST" ] && python tools/kstats.py $ST "$v#$i" || { echo "$v: no stats"; tail -3 /tmp/kt_$i.log; }
done 2>&1 | tee $O/gemm_store_policy.txt
cp /tmp/keep.so $LIB
echo "== bench (default line)"; FD_BENCH_KEEP_STATS=$O/bench_child_kernel_stats.csv timeout 1200 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?"; grep '^{' $O/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['summary'])); print(json.dumps(d['cpu_baseline'])); print(json.dumps(d.get('torch_eager_baseline')))"
echo "== bench config5"; timeout 600 python bench.py --workload config5 --steps 10 --warmup 3 > $O/bench_config5.log 2>&1; echo "rc=$?"; grep '^{' $O/bench_config5.log | cut -c1-1500
echo "== 8 ranks sharing this GPU: config4 (sharded job, bit-equality against the single-process job)"
FD_BENCH_OVERSUBSCRIBE=1 timeout 900 python bench.py --gpus 8 --workload config4 --steps 3 --warmup 1 > $O/bench_config4_8ranks_1gpu.log 2>&1; echo "rc=$?"; grep '^{' $O/bench_config4_8ranks_1gpu.log | cut -c1-1800; grep -i "error\|Traceback" $O/bench_config4_8ranks_1gpu.log | head -5
echo "== 8 ranks sharing this GPU: configs1"
FD_BENCH_OVERSUBSCRIBE=1 timeout 900 python bench.py --gpus 8 --steps 3 --warmup 1 > $O/bench_configs1_8ranks_1gpu.log 2>&1; echo "rc=$?"; grep '^{' $O/bench_configs1_8ranks_1gpu.log | cut -c1-900; grep -i "error\|Traceback" $O/bench_configs1_8ranks_1gpu.log | head -5
echo "== pytest gpu: the new tests first, verbosely"; timeout 900 python -m pytest tests/test_generic_config.py tests/test_mel_frontend.py tests/test_training_path.py tests/test_c_host.py -m gpu -q -s -p no:cacheprovider > $O/pytest_new.log 2>&1; echo "rc=$?"; grep -v "Warning\|warn\|WeightNorm" $O/pytest_new.log | tail -25
echo "== pytest gpu (all)"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu.log
