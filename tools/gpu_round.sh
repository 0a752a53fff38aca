#!/bin/bash
# One full GPU-box session: tests -> smoke -> bench -> rocprofv3 kernel stats -> PMC passes.  Results land in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
(rocm-smi --showclocks --showpower --showperflevel --showmaxpower --showmemvendor 2>&1 | grep -v "^=\|^$" | head -30) > gpurun_out/box_state.txt
echo "== pytest gpu" ; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; tail -4 gpurun_out/pytest_gpu.log
echo "== smoke" ; timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 ; echo "smoke rc=$?" ; tail -1 gpurun_out/smoke.log
echo "== bench" ; FD_BENCH_KEEP_STATS=$R/gpurun_out/bench_child_kernel_stats.csv timeout 900 python bench.py > gpurun_out/bench.log 2>&1 ; echo "bench rc=$?" ; grep '^{' gpurun_out/bench.log | cut -c1-900
(rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|power (W)" | head -4) >> gpurun_out/box_state.txt
echo "== bench stream (the reference CLI's call pattern: 256 requests of 256 lengths)" ; timeout 1500 python bench.py --workload stream --no-cpu-baseline > gpurun_out/bench_stream.log 2>&1 ; grep '^{' gpurun_out/bench_stream.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in d if k.startswith('stream_')})"
echo "== bench N=1000 B=1 (config 3, contractive weights)" ; timeout 900 python bench.py --batch 1 --nsteps 1000 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline > gpurun_out/bench_n1000.log 2>&1 ; grep '^{' gpurun_out/bench_n1000.log | cut -c1-400
echo "== bench config4 (64 ragged utterances, N=6, host to host) on this one GPU" ; timeout 900 python bench.py --workload config4 --steps 5 --warmup 2 > gpurun_out/bench_config4.log 2>&1 ; grep '^{' gpurun_out/bench_config4.log | cut -c1-400
echo "== bench config4 without gather (every rank keeps its share, as the reference)" ; timeout 900 python bench.py --workload config4 --gather none --steps 5 --warmup 2 > gpurun_out/bench_config4_no_gather.log 2>&1 ; grep '^{' gpurun_out/bench_config4_no_gather.log | cut -c1-400
echo "== bench config5 (BASELINE configs[4]: dir of 16 Tacotron-range mels -> int16 PCM on the host)" ; timeout 600 python bench.py --workload config5 --steps 10 --warmup 3 > gpurun_out/bench_config5.log 2>&1 ; grep '^{' gpurun_out/bench_config5.log | cut -c1-400
echo "== the multi-rank path, 8 ranks sharing this GPU over gloo (FD_BENCH_OVERSUBSCRIBE: a code-path proof, not a scaling number)"
FD_BENCH_OVERSUBSCRIBE=1 timeout 900 python bench.py --gpus 8 --workload config4 --steps 3 --warmup 1 > gpurun_out/bench_config4_8ranks_1gpu.log 2>&1 ; grep '^{' gpurun_out/bench_config4_8ranks_1gpu.log | cut -c1-300
FD_BENCH_OVERSUBSCRIBE=1 timeout 900 python bench.py --gpus 8 --workload config4 --gather none --steps 3 --warmup 1 > gpurun_out/bench_config4_8ranks_1gpu_no_gather.log 2>&1 ; grep '^{' gpurun_out/bench_config4_8ranks_1gpu_no_gather.log | cut -c1-300
FD_BENCH_OVERSUBSCRIBE=1 timeout 900 python bench.py --gpus 8 --steps 3 --warmup 1 > gpurun_out/bench_configs1_8ranks_1gpu.log 2>&1 ; grep '^{' gpurun_out/bench_configs1_8ranks_1gpu.log | cut -c1-300
echo "== bench --gpus 2 without the override must refuse" ; python bench.py --gpus 2 > gpurun_out/bench_gpus2_refused.log 2>&1 ; echo "rc=$?" ; tail -1 gpurun_out/bench_gpus2_refused.log
echo "== training step (the training operators stay exercised every round)" ; timeout 600 python tools/train_step_probe.py > gpurun_out/train_step_probe.txt 2>&1 ; echo "train probe rc=$?" ; tail -4 gpurun_out/train_step_probe.txt | cut -c1-300
echo "== rocprof kernel-trace"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o kt -- python $R/bench.py --steps 5 --warmup 1 --no-roofline --no-cpu-baseline --no-fp32-pipe --no-host-io --no-b1 --no-stream --no-torch-eager-baseline > $R/gpurun_out/rocprof.log 2>&1 ; echo "rocprof rc=$?"
cd $R; find gpurun_out/prof -name '*kernel_trace.csv' -size +20M -delete 2>/dev/null
ST=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1); python tools/kernel_stats_fracs.py $ST --bench-json gpurun_out/bench.log > gpurun_out/fracs_from_kernel_stats.txt 2>&1; head -14 gpurun_out/fracs_from_kernel_stats.txt
echo "== PMC"
OUT=$R/gpurun_out/pmc; mkdir -p $OUT; cd /tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-graph --no-fp32-pipe --no-host-io --no-b1 --no-torch-eager-baseline"
pass() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -o $name -- $CMD > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
pass p1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass p2 FETCH_SIZE
pass p3 WRITE_SIZE
pass p4 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA
cd $R; PMC_SOURCE="round 6, rocprofv3 --pmc passes of bench.py --steps 2 --warmup 1 --no-graph (tools/gpu_round.sh); HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch" python tools/pmc_summary.py $OUT --json $OUT/pmc_traffic.json > $OUT/summary.txt 2>&1; cat $OUT/summary.txt | grep -v "at::native\|rocclr"
find $OUT -name '*.csv' -size +8M -delete
