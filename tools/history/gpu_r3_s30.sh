#!/bin/bash
# round-3 session 30: is the hop-256 LVC layer bound by the CU's memory pipe?  +4 / +12 L2-hitting 16 B loads per lane (+16 / +48 KB per workgroup)
mkdir -p gpurun_out
{
for i in 1 2 3; do
echo "== ld0"; tools/ubench/lvc_h2_bench 8 864
echo "== ld4"; tools/ubench/lvc_h2_bench_ld4 8 864
echo "== ld12"; tools/ubench/lvc_h2_bench_ld12 8 864
done
} > gpurun_out/ld_ab.txt 2>&1
tail -3 gpurun_out/ld_ab.txt
