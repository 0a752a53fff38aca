#!/bin/bash
# round-3 session 29: is the hop-256 LVC layer bound by instruction issue?  +128 / +256 VALU instructions per wave (of ~1650)
mkdir -p gpurun_out
{
for i in 1 2 3; do
echo "== pad0"; tools/ubench/lvc_h2_bench 8 864
echo "== pad128"; tools/ubench/lvc_h2_bench_pad128 8 864
echo "== pad256"; tools/ubench/lvc_h2_bench_pad256 8 864
done
} > gpurun_out/pad_ab.txt 2>&1
tail -3 gpurun_out/pad_ab.txt
