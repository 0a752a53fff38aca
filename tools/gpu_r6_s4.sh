#!/bin/bash
# round 6, session 4: the gate of VERDICT item 6 (what do the fused up-sampler / fused final conv cost inside their hop-256 layers?) as an
# in-session A/B of probe builds, and B=1 at T=864 replayed from graphs vs launched kernel by kernel
set -u
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
