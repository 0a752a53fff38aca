#!/bin/bash
# round 6, session 7: what sits between two one-utterance calls on the GPU's timeline; one replicate launch instead of N small D2D copies (A/B)
set -u
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
