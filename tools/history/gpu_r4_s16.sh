#!/bin/bash
# round 4, session 16: option order = split (GEMM of block 0, LVC block 0, GEMM of blocks 1 and 2, LVC blocks 1 and 2) against the default
set -u
mkdir -p gpurun_out/s16
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s16
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 600 python tools/ab_opts.py --batch 8 --reps 3 --steps 20 "" "order=split" "" "order=split" 2>&1 | grep -v Warn | tee $O/ab_order_split_B8.txt
timeout 600 python bench.py --steps 10 --opt order=split --no-cpu-baseline --no-fp32-pipe --no-host-io --no-b1 > $O/bench_split.log 2>&1
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/s16"
for line in open(O+"/bench_split.log"):
    if line.startswith("{"):
        d=json.loads(line); print("order=split ms/step", d["ms_per_step"])
        for k,v in list(d.get("kernels",{}).items())[:10]: print("   ",k,{a:b for a,b in v.items() if a in ("launches_per_step","avg_us","hbm_frac")})
PY
