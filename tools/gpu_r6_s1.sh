#!/bin/bash
# round 6, session 1: the reference CLI's call pattern (a new T every call) on the round-5 library -- the "before" numbers
set -u
mkdir -p gpurun_out/r6s1
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6s1/build.log 2>&1
timeout 1200 python bench.py --workload stream --no-cpu-baseline > gpurun_out/r6s1/bench_stream.log 2>&1; echo "stream rc=$?"
grep '^{' gpurun_out/r6s1/bench_stream.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d['stream'], indent=1)); print(d['value'], d['ms_per_step'])"
tail -3 gpurun_out/r6s1/bench_stream.log | cut -c1-300
