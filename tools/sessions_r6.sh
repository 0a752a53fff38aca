#!/bin/bash
# The short GPU sessions of round 6, one case each (what each `gpurun -- bash tools/sessions_r6.sh sN` ran; the full session is tools/gpu_round.sh).
# Results: profiles/r06/sN_*, LABBOOK.md "Round 6".
set -u
case "${1:-}" in
s1)
# round 6, session 1: the reference CLI's call pattern (a new T every call) on the round-5 library -- the "before" numbers
mkdir -p gpurun_out/r6s1
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6s1/build.log 2>&1
timeout 1200 python bench.py --workload stream --no-cpu-baseline > gpurun_out/r6s1/bench_stream.log 2>&1; echo "stream rc=$?"
grep '^{' gpurun_out/r6s1/bench_stream.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d['stream'], indent=1)); print(d['value'], d['ms_per_step'])"
tail -3 gpurun_out/r6s1/bench_stream.log | cut -c1-300
;;
s2)
# round 6, session 2: frame-bucketed graphs + safe-by-default boundary: new tests first, then the whole GPU suite, then the stream bench
mkdir -p gpurun_out/r6s2
O=gpurun_out/r6s2
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "bucketed or graph_cache or c_host" > $O/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -15 $O/pytest_new.log
timeout 1200 python bench.py --workload stream --no-cpu-baseline > $O/bench_stream.log 2>&1; echo "stream rc=$?"
grep '^{' $O/bench_stream.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stream']
for k,v in s.items(): print(k, v)
"
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log
;;
s3)
# round 6, session 3: whole GPU suite on the bucketed / safe-by-default library, headline bench (value = host to host), stream with the
# synchronous legs, config4 in both gather modes, the per-utterance cost model, N=1000 on contractive weights
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
;;
s4)
# round 6, session 4: the gate of VERDICT item 6 (what do the fused up-sampler / fused final conv cost inside their hop-256 layers?) as an
# in-session A/B of probe builds, and B=1 at T=864 replayed from graphs vs launched kernel by kernel
O=gpurun_out/r6s4; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
cp fastdiff_amd/lib/libfastdiff_hip.so /tmp/ship.so
cmp /tmp/ship.so gpurun_ab/base.so && echo "base.so == the shipped build"
AB_ARGS="--no-fp32-pipe --no-torch-eager-baseline --no-b1 --no-replay-profile" AB_MORE="gpurun_ab/final_nofold.so" bash tools/gpu_ab.sh gpurun_ab/base.so gpurun_ab/up_noconvt.so 3 2>&1 | tee $O/ab_gate.txt
cp /tmp/ship.so fastdiff_amd/lib/libfastdiff_hip.so
for g in "" "--no-graph"; do
  for t in 864 539; do
    python bench.py --batch 1 --frames $t --steps 200 --warmup 20 --no-host-io --no-roofline --no-cpu-baseline --no-fp32-pipe --no-torch-eager-baseline $g > /tmp/b1.log 2>&1
    grep '^{' /tmp/b1.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B=1 T=$t graph=%s device-resident ms' % d['config']['graph'], d['ms_per_step'], 'rtf', d['value'])" | tee -a $O/b1_graph_vs_no_graph.txt
  done
done
;;
s5)
# round 6, session 5: phase timelines of the fused hop-256 variants (where do the up-sampler's 63 us and the fold's 29 us go?) + the s7 test
O=gpurun_out/r6s5; mkdir -p $O
for v in 0 1 2; do
  tools/ubench/lvc_h2_timeline_v$v /tmp/tl_v$v.bin 8 864 > $O/timeline_v$v.txt 2>&1
  python tools/timeline_fused_report.py /tmp/tl_v$v.bin $v >> $O/timeline_v$v.txt 2>&1
  cat $O/timeline_v$v.txt
done
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -s -k "against_the_reference_trajectory" > $O/pytest_s7.log 2>&1; echo "s7 rc=$?"; grep "config3 T=864 N=1000 vs\|passed\|failed\|Error" $O/pytest_s7.log | cut -c1-1200
;;
s6)
# round 6, session 6: the fused up-sampler with its B operands requested one tile ahead and the record requested behind the phase: timeline, A/B, bits
O=gpurun_out/r6s6; mkdir -p $O
tools/ubench/lvc_h2_timeline_v1 /tmp/tl_v1.bin 8 864 > $O/timeline_v1.txt 2>&1; python tools/timeline_fused_report.py /tmp/tl_v1.bin 1 >> $O/timeline_v1.txt 2>&1; cat $O/timeline_v1.txt | tail -9
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
cp fastdiff_amd/lib/libfastdiff_hip.so /tmp/ship.so
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "up_sampler or fused or graph_replay or golden or bucketed or b8_items or ragged" > $O/pytest_sel.log 2>&1; echo "selected tests rc=$?"; tail -3 $O/pytest_sel.log
AB_ARGS="--no-fp32-pipe --no-torch-eager-baseline --no-b1 --no-replay-profile" bash tools/gpu_ab.sh gpurun_ab/base.so /tmp/ship.so 3 2>&1 | tee $O/ab_up_pipe.txt
cp /tmp/ship.so fastdiff_amd/lib/libfastdiff_hip.so
for i in 1 2; do for v in gpurun_ab/base.so /tmp/ship.so; do cp $v fastdiff_amd/lib/libfastdiff_hip.so; python bench.py --batch 1 --steps 200 --warmup 20 --no-host-io --no-roofline --no-cpu-baseline --no-fp32-pipe --no-torch-eager-baseline > /tmp/b1.log 2>&1; grep '^{' /tmp/b1.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v B=1 T=864 ms', d['ms_per_step'])" | tee -a $O/ab_up_pipe.txt; done; done
cp /tmp/ship.so fastdiff_amd/lib/libfastdiff_hip.so
;;
s7)
# round 6, session 7: what sits between two one-utterance calls on the GPU's timeline; one replicate launch instead of N small D2D copies (A/B)
O=gpurun_out/r6s7; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
cp fastdiff_amd/lib/libfastdiff_hip.so /tmp/ship.so
cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/b1prof -o b1 -- python $R/tools/b1_timeline.py run 864 > $R/$O/rocprof_b1.log 2>&1; cd $R
python tools/b1_timeline.py report /tmp/b1prof > $O/b1_timeline.txt 2>&1; head -75 $O/b1_timeline.txt
for i in 1 2 3; do for v in gpurun_ab/base.so /tmp/ship.so; do cp $v fastdiff_amd/lib/libfastdiff_hip.so; python bench.py --batch 1 --steps 300 --warmup 30 --no-host-io --no-roofline --no-cpu-baseline --no-fp32-pipe --no-torch-eager-baseline > /tmp/b1.log 2>&1; grep '^{' /tmp/b1.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v B=1 T=864 ms', d['ms_per_step'])" | tee -a $O/ab_b1_replicate.txt; done; done
cp /tmp/ship.so fastdiff_amd/lib/libfastdiff_hip.so
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "bucketed or ragged or hoist or embedding or b8_items or config4 or golden or graph" > $O/pytest_sel.log 2>&1; echo "selected tests rc=$?"; tail -3 $O/pytest_sel.log
;;
s8)
# round 6, session 8: the default bench line with its short stream object
O=gpurun_out/r6s8; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
T0=$(date +%s); python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"
grep '^{' $O/bench.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d['summary'])); print(d['stream'])"
;;
s9)
# round 6, session 9: the extended bucketing test, the world-2 job without gather on the real vocoder, the C host
O=gpurun_out/r6s9; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -k "bucketed or without_gather or c_host or graph_cache" > $O/pytest_sel.log 2>&1; echo "rc=$?"; tail -5 $O/pytest_sel.log
;;
*) echo "usage: $0 s1..s9"; exit 2;;
esac
