#!/bin/bash
# Round 5, session 11: the config4 lines again with the micro-batch default of 16 (one GPU with the 8-rank projection; 8 ranks sharing the GPU).
set -u
mkdir -p gpurun_out/r5s11
O=$GRAFT_REPO_ROOT/gpurun_out/r5s11
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 900 python bench.py --workload config4 --steps 5 --warmup 2 > $O/bench_config4.log 2>&1; grep '^{' $O/bench_config4.log | cut -c1-300
FD_BENCH_OVERSUBSCRIBE=1 timeout 900 python bench.py --gpus 8 --workload config4 --steps 3 --warmup 1 > $O/bench_config4_8ranks_1gpu.log 2>&1; grep '^{' $O/bench_config4_8ranks_1gpu.log | cut -c1-300
