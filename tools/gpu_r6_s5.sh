#!/bin/bash
# round 6, session 5: phase timelines of the fused hop-256 variants (where do the up-sampler's 63 us and the fold's 29 us go?) + the s7 test
set -u
O=gpurun_out/r6s5; mkdir -p $O
for v in 0 1 2; do
  tools/ubench/lvc_h2_timeline_v$v /tmp/tl_v$v.bin 8 864 > $O/timeline_v$v.txt 2>&1
  python tools/timeline_fused_report.py /tmp/tl_v$v.bin $v >> $O/timeline_v$v.txt 2>&1
  cat $O/timeline_v$v.txt
done
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -s -k "against_the_reference_trajectory" > $O/pytest_s7.log 2>&1; echo "s7 rc=$?"; grep "config3 T=864 N=1000 vs\|passed\|failed\|Error" $O/pytest_s7.log | cut -c1-1200
