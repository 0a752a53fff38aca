#!/bin/bash
# round-3 session 40: nt (non-temporal) cache policy on the LVC layer's streams: 2 = out stores, 4 = records, 7 = x + out + records (skip stays cacheable), 15 = all, 8 = skip only
mkdir -p gpurun_out
{
for i in 1 2 3; do
for nt in 0 2 4 7 15 8 1; do echo "== nt$nt"; tools/ubench/lvc_h2_bench_nt$nt 8 864; done
done
} > gpurun_out/nt_ab.txt 2>&1
tail -3 gpurun_out/nt_ab.txt
