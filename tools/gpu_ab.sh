#!/bin/bash
# A/B builds of libfastdiff_hip.so inside ONE GPU session (box-to-box variation is +-4 %): tools/gpu_ab.sh A.so B.so [reps]   (AB_MORE="C.so D.so": further variants)
set -u
LIB=fastdiff_amd/lib/libfastdiff_hip.so
cp $LIB /tmp/keep.so
for i in $(seq 1 ${3:-3}); do
  for v in $1 $2 ${AB_MORE:-}; do
    cp $v $LIB
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-io --no-stream ${AB_ARGS:-} > /tmp/ab.log 2>&1
    python - "$v" <<'PY'
import json, sys
for line in open('/tmp/ab.log'):
    if line.startswith('{'):
        d = json.loads(line)
        k = d.get('kernels', {})
        pick = {n: k[n]['avg_us'] for n in ('lvc_layer_h256', 'lvc_up_h256', 'lvc_final_h256', 'lvc_up_h64', 'kp_gemm_f16x2', 'lvc_layer_h64', 'lvc_layer_h8', 'final_update') if n in k}
        print(f"{sys.argv[1]:28s} ms/step {d['ms_per_step']:.3f}  {pick}")
PY
  done
done
cp /tmp/keep.so $LIB
