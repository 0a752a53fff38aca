#!/bin/bash
# round-3 session 44: ConvTranspose r=4 with the register budget of three workgroups per CU (168 VGPRs, no spills) against two (187)
mkdir -p gpurun_out
cp fastdiff_amd/lib/libfastdiff_hip.so /tmp/lib_orig.so
for rep in 1 2 3; do for v in occ2 occ3; do
  cp fastdiff_amd/lib/variants/libfastdiff_hip_$v.so fastdiff_amd/lib/libfastdiff_hip.so
  python bench.py --no-cpu-baseline --no-fp32-pipe --no-host-io --no-b1 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = j['kernels']
print('$v rep$rep', j['ms_per_step'], {n: k[n]['avg_us'] for n in ('convt_r4', 'convt_r8', 'lvc_layer_h256', 'lvc_layer_h64', 'lvc_layer_h8', 'kp_gemm_f16x2')})"
done; done > gpurun_out/convt_occ.txt 2>&1
cp /tmp/lib_orig.so fastdiff_amd/lib/libfastdiff_hip.so
cat gpurun_out/convt_occ.txt
