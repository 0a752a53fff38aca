#!/bin/bash
# quick iteration: parity tests + bench (no rocprof)
set -u
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; tail -5 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench.log 2>&1 ; echo "bench rc=$?"
python - <<'PY'
import json
for line in open('gpurun_out/bench.log'):
    if line.startswith('{'):
        d=json.loads(line)
        print('RTF', d['value'], 'ms/step', d['ms_per_step'], 'roofline', d.get('roofline'))
        for k,v in d.get('kernels',{}).items(): print(f"  {k:20s} {v}")
PY
