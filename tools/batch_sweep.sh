#!/bin/bash
# Throughput against batch size and schedule length on one GPU (no roofline / CPU legs): writes gpurun_out/batch_sweep.jsonl
set -u
mkdir -p gpurun_out; : > gpurun_out/batch_sweep.jsonl
(rocm-smi --showclocks --showpower --showperflevel --showmaxpower 2>&1 | grep -v '^=\|^$' | head -30) > gpurun_out/box_state.txt
run() { python bench.py --no-roofline --no-cpu-baseline "$@" 2>/dev/null | grep '^{' >> gpurun_out/batch_sweep.jsonl; }
for b in 1 2 4 8 16 32 64; do run --batch $b --steps 10 --warmup 2; done
for n in 3 6 8; do run --batch 8 --nsteps $n --steps 10 --warmup 2; done
run --batch 8 --nsteps 200 --steps 2 --warmup 1
run --batch 64 --nsteps 6 --ragged --steps 5 --warmup 1              # BASELINE config 4 on one GPU
run --batch 64 --nsteps 6 --ragged --no-lens --steps 5 --warmup 1
(rocm-smi --showclocks --showpower 2>&1 | grep -i 'sclk\|power' | head -6) >> gpurun_out/box_state.txt
python - <<'PY'
import json
print(open('gpurun_out/box_state.txt').read())
for l in open('gpurun_out/batch_sweep.jsonl'):
    d = json.loads(l); c = d['config']
    print(f"B={c['batch_per_gpu']:3d} N={c['reverse_steps']:4d} ragged={bool(c['ragged'])!s:5s} lens={(c['ragged'] or {}).get('told_to_library')!s:5s} "
          f"ms/call {d['ms_per_step']:9.3f}  RTF {d['value']:9.1f}  host-inclusive {d.get('host_inclusive', {}).get('value')}")
PY
