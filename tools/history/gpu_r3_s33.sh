#!/bin/bash
# round-3 session 33: does the row stride (T*hop*4 bytes = 27 * 32 KB at T=864) cost HBM channel conflicts?  the same kernels at other T
mkdir -p gpurun_out
{
for i in 1 2; do
for T in 864 865 866 868 872 880 896 863; do tools/ubench/lvc_h2_bench 8 $T; done
done
} > gpurun_out/stride_sweep.txt 2>&1
tail -4 gpurun_out/stride_sweep.txt
