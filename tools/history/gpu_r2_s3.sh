#!/bin/bash
# round-2 session 3: the LVC instruction diet, ingredient by ingredient (harness), then parity + A/B of the full build
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do
echo "== A: old kernel (kpre 0, lrelu25 0, mix 0)"; tools/ubench/lvc_v_a 8 864 0 | grep lvc_f16
echo "== B: kpre 1";                              tools/ubench/lvc_v_a 8 864 1 | grep lvc_f16
echo "== C: kpre 1 + lrelu25";                    tools/ubench/lvc_v_c 8 864 1 | grep lvc_f16
echo "== E: kpre 1 + mix split";                  tools/ubench/lvc_v_e 8 864 1 | grep lvc_f16
echo "== D: kpre 1 + lrelu25 + mix split";        tools/ubench/lvc_v_d 8 864 1 | grep lvc_f16
done 2>&1 | tee gpurun_out/lvc_variants.txt
echo "== D phase stamps"; tools/ubench/lvc_h2_bench_t 8 864 1 > gpurun_out/lvc_h2_t.txt 2>&1; cat gpurun_out/lvc_h2_t.txt
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log | cut -c1-300
echo "== A/B"; cp fastdiff_amd/lib/libfastdiff_hip.so /tmp/new.so; AB_ARGS="--no-fp32-pipe" bash tools/gpu_ab.sh gpurun_ab/base.so /tmp/new.so 3 2>&1 | tee gpurun_out/ab.txt
