#!/bin/bash
# round-3 session 24: phase timeline of every workgroup of the hop-256 LVC layer
mkdir -p gpurun_out
tools/ubench/lvc_h2_timeline gpurun_out/timeline.bin > gpurun_out/timeline.txt 2>&1
cat gpurun_out/timeline.txt
