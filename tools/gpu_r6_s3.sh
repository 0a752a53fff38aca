#!/bin/bash
# round 6, session 3: whole GPU suite on the bucketed / safe-by-default library, headline bench (value = host to host), stream with the
# synchronous legs, config4 in both gather modes, the per-utterance cost model, N=1000 on contractive weights
set -u
O=gpurun_out/r6s3; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
(rocm-smi --showclocks --showpower --showperflevel --showmaxpower 2>&1 | grep -v "^=\|^$" | head -30) > $O/box_state.txt
echo "== pytest gpu"; timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -k "not against_the_reference_trajectory" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
echo "== bench"; FD_BENCH_KEEP_STATS=$R/$O/bench_child_kernel_stats.csv timeout 900 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?"; grep '^{' $O/bench.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d['summary'])); print(d['config']['value_is']); print({k:d['roofline'][k] for k in ('kernel','frac','avg_launch_us','traffic') if k in d['roofline']}); print(d.get('cpu_baseline',{}).get('value'), d['b1'].get('host_us_per_call_median'))"
echo "== stream"; timeout 1500 python bench.py --workload stream --no-cpu-baseline > $O/bench_stream.log 2>&1; echo "stream rc=$?"
grep '^{' $O/bench_stream.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stream']
for k,v in s.items(): print(k, v)
"
echo "== config4 gather=src"; timeout 900 python bench.py --workload config4 --steps 5 --warmup 2 > $O/bench_config4_src.log 2>&1; grep '^{' $O/bench_config4_src.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], json.dumps(d['projection'])[:900])"
echo "== config4 gather=none"; timeout 900 python bench.py --workload config4 --steps 5 --warmup 2 --gather none > $O/bench_config4_none.log 2>&1; grep '^{' $O/bench_config4_none.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], json.dumps(d['projection'])[:900])"
echo "== config4 gather=none balance=frames"; timeout 900 python bench.py --workload config4 --steps 5 --warmup 2 --gather none --balance frames > $O/bench_config4_none_frames.log 2>&1; grep '^{' $O/bench_config4_none_frames.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], json.dumps(d['projection'])[:600])"
echo "== cost model"; timeout 600 python tools/cost_model.py > $O/cost_model.json 2> $O/cost_model.err; python -c "
import json; d=json.load(open('$O/cost_model.json')); print({k:v for k,v in d.items() if k!='points'})"
echo "== N=1000 B=1 (configs[2], contractive weights)"; timeout 900 python bench.py --batch 1 --nsteps 1000 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-fp32-pipe --no-torch-eager-baseline > $O/bench_n1000.log 2>&1; grep '^{' $O/bench_n1000.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('value_device_resident'), d.get('long_schedule'), d['config']['weights'])"
echo "== 8 ranks on this GPU (code path)"; FD_BENCH_OVERSUBSCRIBE=1 timeout 900 python bench.py --gpus 8 --workload config4 --gather none --steps 3 --warmup 1 > $O/bench_config4_8ranks_none.log 2>&1; grep '^{' $O/bench_config4_8ranks_none.log | cut -c1-400; tail -2 $O/bench_config4_8ranks_none.log | cut -c1-300
FD_BENCH_OVERSUBSCRIBE=1 timeout 900 python bench.py --gpus 8 --steps 3 --warmup 1 > $O/bench_configs1_8ranks.log 2>&1; grep '^{' $O/bench_configs1_8ranks.log | cut -c1-400; tail -2 $O/bench_configs1_8ranks.log | cut -c1-300
