#!/bin/bash
# round 6, session 8: the default bench line with its short stream object
set -u
O=gpurun_out/r6s8; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
T0=$(date +%s); python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"
grep '^{' $O/bench.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d['summary'])); print(d['stream'])"
