#!/bin/bash
# round-3 session 39: clocks and power while the hop-256 LVC layer / the traffic-only probe / the predictor GEMM run back to back
mkdir -p gpurun_out
smi() { rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|fclk\|socclk\|power (W)\|Average Graphics" | tr -s ' ' | tr '\n' ';'; echo; }
{
echo "idle: $(smi)"
for prog in "tools/ubench/lvc_h2_bench 8 864" "tools/ubench/copy_mix_probe" "tools/ubench/gemm_h2_bench"; do
  ( for i in 1 2 3 4 5 6 7 8; do $prog > /dev/null 2>&1; done ) &
  pid=$!
  sleep 1.0
  for k in 1 2 3 4; do echo "$prog: $(smi)"; sleep 0.4; done
  wait $pid
done
} > gpurun_out/power_clocks.txt 2>&1
cat gpurun_out/power_clocks.txt
