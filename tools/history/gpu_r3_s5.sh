#!/bin/bash
# round-3 session 5: whole GPU suite, overlap = gemm (per-block predictor GEMM next to the LVC layers) A/B + rocprof, new bench line
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
(rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|power (W)" | head -4) > gpurun_out/box_state.txt
echo "== pytest gpu" ; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -rx > gpurun_out/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; tail -6 gpurun_out/pytest_gpu.log | cut -c1-300
echo "== parity subset with overlap=gemm"
FD_TEST_OPTS="overlap=gemm" timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "sampler_matches or full_size_sampler or ragged_batch_with_lens or graph_replay or config4" 2>&1 | tail -3 | cut -c1-300
echo "== A/B overlap"
timeout 300 python tools/ab_opts.py --batch 8 "" "overlap=gemm" "overlap=gemm,overlap_wg=2" 2>&1 | grep -v Warn | tee gpurun_out/ab_overlap_b8.txt
timeout 300 python tools/ab_opts.py --batch 1 --steps 40 "" "overlap=gemm" "overlap=gemm,overlap_wg=2" "fallback=host" 2>&1 | grep -v Warn | tee gpurun_out/ab_overlap_b1.txt
echo "== rocprof kernel trace, overlap=gemm"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_overlap -o kt -- python $R/bench.py --steps 5 --warmup 1 --no-roofline --no-cpu-baseline --no-fp32-pipe --no-host-io --no-b1 --opt overlap=gemm > $R/gpurun_out/rocprof_overlap.log 2>&1 ; echo "rocprof rc=$?"
cd $R; find gpurun_out/prof_overlap -name '*kernel_trace.csv' -size +20M -delete 2>/dev/null
echo "== bench (default line)" ; timeout 900 python bench.py > gpurun_out/bench.log 2>&1 ; echo "bench rc=$?" ; grep '^{' gpurun_out/bench.log | cut -c1-600
echo "== bench config4 with the 8-rank projection" ; timeout 600 python bench.py --workload config4 --steps 5 --warmup 2 > gpurun_out/bench_config4.log 2>&1 ; grep '^{' gpurun_out/bench_config4.log | cut -c1-300
