#!/bin/bash
# round-2 session 16: fresh-process hunt (tools/fresh_proc_hunt.py): args K ROUNDS OPTIONS
set -u
mkdir -p gpurun_out
timeout 1200 python tools/fresh_proc_hunt.py "${1:-4}" "${2:-10}" "${3:-}" 2>&1 | tail -30 | cut -c1-300 | tee -a gpurun_out/fresh_hunt.txt
