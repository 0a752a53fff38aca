#!/bin/bash
# round-2 session 8: hop-8 LVC layer on 16x16x32 matrix tiles: harness, parity, A/B against the all-VALU kernel (option lvc_h8)
set -u
mkdir -p gpurun_out
echo "== harness"; tools/ubench/lvc_h8_bench 2>&1 | tee gpurun_out/lvc_h8.txt
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log | cut -c1-300; grep -a "^E " gpurun_out/pytest_gpu.log | head -10
for b in 8 1; do for mode in mfma valu mfma valu; do
  echo "== B=$b lvc_h8=$mode"; python bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-pipe --no-host-io --opt lvc_h8=$mode 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['kernels']
print(d['ms_per_step'], d['value'], {n: k[n]['avg_us'] for n in ('lvc_layer_h8','lvc_layer_h64','lvc_layer_h256','kp_gemm_f16x2','lvc_fp32_fallback') if n in k})"
done; done 2>&1 | tee gpurun_out/ab_h8.txt
