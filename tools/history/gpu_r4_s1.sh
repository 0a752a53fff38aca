#!/bin/bash
# round 4, session 1: new parity tests (configs[2] at T=864, configs[3] vs the oracle, ADVICE fixes), the per-variant roofline rows with
# kernel-timestamp timing, and a rocprofv3 kernel trace of the same command to lay beside them
set -u
mkdir -p gpurun_out/s1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s1
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
(rocm-smi --showclocks --showpower --showperflevel 2>&1 | grep -v "^=\|^$" | head -20) > $O/box_state.txt
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
grep -h "config3 T=864\|config4 item\|n1000 T=64" $O/pytest_gpu.log | cut -c1-700
echo "== bench (default: kernel timestamps)"; timeout 900 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?"
echo "== bench (events around launches)"; FD_BENCH_PROFILE=events timeout 600 python bench.py --steps 10 --no-cpu-baseline --no-fp32-pipe --no-host-io --no-b1 > $O/bench_events.log 2>&1; echo "rc=$?"
echo "== rocprofv3 kernel trace of the bench command (graph replays + the profiled pass)"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o kt -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-fp32-pipe --no-host-io --no-b1 > $O/rocprof_bench.log 2>&1; echo "rocprof rc=$?"
cd $R; find $O/prof -name '*kernel_trace.csv' -size +20M -delete 2>/dev/null
ST=$(find $O/prof -name '*kernel_stats.csv' | head -1)
python tools/kernel_stats_fracs.py $ST --bench-json $O/rocprof_bench.log > $O/fracs_vs_rocprof.txt 2>&1; cat $O/fracs_vs_rocprof.txt
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/s1"
for f in ("bench.log","bench_events.log"):
    for line in open(O+"/"+f):
        if line.startswith("{"):
            d=json.loads(line)
            print(f, "ms/step", d["ms_per_step"], "RTF", d["value"], "b1", d.get("b1",{}).get("ms_per_step"))
            r=d.get("roofline",{})
            print("  roofline", {k:r.get(k) for k in ("kernel","achieved","frac","avg_launch_us","traffic")})
            print("  variants", r.get("variants"))
            print("  all12", r.get("lvc_all_12_launches"))
            for k,v in list(d.get("kernels",{}).items())[:12]: print("   ",k,v)
PY
