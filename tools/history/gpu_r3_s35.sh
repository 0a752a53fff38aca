#!/bin/bash
# round-3 session 35: phase timeline of the hop-256 LVC layer at B=1 (864 tiles on 512 slots) and B=2
mkdir -p gpurun_out
tools/ubench/lvc_h2_timeline gpurun_out/timeline_b1.bin 1 864 > gpurun_out/timeline_b1.txt 2>&1
tools/ubench/lvc_h2_timeline gpurun_out/timeline_b2.bin 2 864 >> gpurun_out/timeline_b1.txt 2>&1
cat gpurun_out/timeline_b1.txt
