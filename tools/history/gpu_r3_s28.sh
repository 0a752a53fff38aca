#!/bin/bash
# round-3 session 28: hop-256 LVC with the frame's record requested after x/skip (so that staging does not wait for it)
mkdir -p gpurun_out
{
for i in 1 2 3; do
echo "== early"; tools/ubench/lvc_h2_bench 8 864
echo "== late"; tools/ubench/lvc_h2_bench_late 8 864
done
tools/ubench/lvc_h2_timeline_late gpurun_out/timeline_late.bin
} > gpurun_out/late_ab.txt 2>&1
tail -6 gpurun_out/late_ab.txt
