#!/bin/bash
# round 6, session 6: the fused up-sampler with its B operands requested one tile ahead and the record requested behind the phase: timeline, A/B, bits
set -u
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
