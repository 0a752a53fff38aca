#!/bin/bash
# round 4, session 2: the new tests (caller pin, forced collectives on RCCL), A/B of the round-4 forms of the fused LVC launches
# (option lvc_variant), the bench line with the rocprofv3 child, the GEMM's phase stamps
set -u
mkdir -p gpurun_out/s2
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s2
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
(rocm-smi --showclocks --showpower --showperflevel 2>&1 | grep -v "^=\|^$" | head -20) > $O/box_state.txt
echo "== new tests"; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -s -k "test_step or forced or up_sampler or final_conv_fused or rccl or hoisted" > $O/pytest_new.log 2>&1; echo "rc=$?"; tail -3 $O/pytest_new.log; grep -h "test_step N\|test_step list" $O/pytest_new.log
echo "== A/B lvc_variant B=8"; timeout 600 python tools/ab_opts.py --batch 8 --reps 3 --steps 20 "lvc_variant=1" "lvc_variant=0" "lvc_variant=1" "lvc_variant=0" 2>&1 | grep -v Warn | tee $O/ab_variant_B8.txt
echo "== A/B lvc_variant B=1"; timeout 600 python tools/ab_opts.py --batch 1 --reps 3 --steps 50 "lvc_variant=1" "lvc_variant=0" "lvc_variant=1" "lvc_variant=0" 2>&1 | grep -v Warn | tee $O/ab_variant_B1.txt
echo "== bench variant 0 (kernels only)"; timeout 600 python bench.py --steps 10 --opt lvc_variant=0 --no-cpu-baseline --no-fp32-pipe --no-host-io --no-b1 > $O/bench_v0.log 2>&1; echo "rc=$?"
echo "== bench default"; FD_BENCH_KEEP_STATS=$O/bench_child_kernel_stats.csv timeout 900 python bench.py > $O/bench.log 2>&1; echo "rc=$?"
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/s2"
for f in ("bench_v0.log","bench.log"):
    for line in open(O+"/"+f):
        if line.startswith("{"):
            d=json.loads(line)
            print(f, "ms/step", d["ms_per_step"], "RTF", d["value"], "b1", d.get("b1",{}).get("ms_per_step"))
            r=d.get("roofline",{})
            print("  roofline", {k:r.get(k) for k in ("kernel","achieved","frac","avg_launch_us","timing","eager","replay_error")})
            print("  all12", {k:v for k,v in (r.get("lvc_all_12_launches") or {}).items() if k!="note"})
            for k,v in list(d.get("kernels",{}).items())[:9]: print("   ",k,{a:b for a,b in v.items() if a in ("launches_per_step","avg_us","avg_us_eager","hbm_frac","share")})
PY
echo "== GEMM phase stamps"; ./tools/ubench/gemm_h2_bench_t 8 864 512 2>&1 | tee $O/gemm_phase_stamps.txt
