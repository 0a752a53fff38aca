#!/bin/bash
# round 6, session 2: frame-bucketed graphs + safe-by-default boundary: new tests first, then the whole GPU suite, then the stream bench
set -u
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
