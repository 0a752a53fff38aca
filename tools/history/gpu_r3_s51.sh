#!/bin/bash
# round-3 session 51: the step-embedding rows kept between calls (option embed_cache): test, A/B at B=1 and B=8
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "embedding_table or up_sampler or hoisted or host_checked" > gpurun_out/pytest_emb.txt 2>&1; tail -3 gpurun_out/pytest_emb.txt
for B in 1 8; do
python tools/ab_opts.py --batch $B --reps 3 --steps 30 "embed_cache=0" "embed_cache=1" "embed_cache=0" "embed_cache=1" 2>&1 | grep "^B="
done > gpurun_out/ab_emb.txt 2>&1
cat gpurun_out/ab_emb.txt
