#!/bin/bash
# round-3 session 41: nt cache policy on the LVC layers' streams IN THE STEP (library built with -DFD_LVC_NT=k: 2 = out stores,
# 3 = x loads + out stores, 10 = out stores + skip loads), separate processes alternating, twice
mkdir -p gpurun_out
cp fastdiff_amd/lib/libfastdiff_hip.so /tmp/lib_orig.so
for rep in 1 2; do for nt in 0 2 3 10; do
  cp fastdiff_amd/lib/variants/libfastdiff_hip_nt$nt.so fastdiff_amd/lib/libfastdiff_hip.so
  python bench.py --no-cpu-baseline --no-fp32-pipe --no-host-io 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = j['kernels']
print('nt$nt rep$rep', j['ms_per_step'], 'b1', j['b1']['ms_per_step'], {n: k[n]['avg_us'] for n in ('lvc_layer_h256', 'lvc_layer_h64', 'lvc_layer_h8', 'convt_r4', 'convt_r8', 'kp_gemm_f16x2')})"
done; done > gpurun_out/nt_instep.txt 2>&1
cp /tmp/lib_orig.so fastdiff_amd/lib/libfastdiff_hip.so
cat gpurun_out/nt_instep.txt
