#!/bin/bash
# round-3 session 7: pipelined host check (fallback = host, deferred checks) -- tests and A/B at B=8 / B=1; kernel_conv cost in the training step
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_infer_glue.py tests/test_sharded_synthesis.py tests/test_c_host.py -m gpu -q -p no:cacheprovider -k "host_check or fallback or sharded or synthesize or c_host or sampler_matches or config4 or config5" 2>&1 | tail -4 | cut -c1-300
echo "== the same selection with fallback=host for every test model"; FD_TEST_OPTS="fallback=host" timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_infer_glue.py -m gpu -q -p no:cacheprovider -k "sampler_matches or full_size or ragged_batch or config4 or config5 or synthesize or n1000" 2>&1 | tail -4 | cut -c1-300
echo "== A/B"
timeout 300 python tools/ab_opts.py --batch 8 "" "fallback=host" 2>&1 | grep "^B=\|config" | tee gpurun_out/ab_hostcheck_b8.txt
timeout 300 python tools/ab_opts.py --batch 1 --steps 40 "" "fallback=host" 2>&1 | grep "^B=\|config" | tee gpurun_out/ab_hostcheck_b1.txt
timeout 300 python tools/ab_opts.py --batch 2 --steps 40 "" "fallback=host" 2>&1 | grep "^B=" | tee -a gpurun_out/ab_hostcheck_b1.txt
echo "== host to host with fallback=host"; timeout 300 python bench.py --steps 20 --no-roofline --no-cpu-baseline --no-fp32-pipe --no-b1 --opt fallback=host 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_inclusive']['ms_per_step'])"
timeout 300 python bench.py --steps 20 --no-roofline --no-cpu-baseline --no-fp32-pipe --no-b1 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_inclusive']['ms_per_step'])"
echo "== kernel_conv (64 -> 24576, k3) on PyTorch-ROCm at the training shape B=20, T=100: forward / backward"
python - <<'PY'
import torch, time
h = torch.randn(20, 64, 100, device='cuda', requires_grad=True)
conv = torch.nn.Conv1d(64, 24576, 3, padding=1).cuda()
def t(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    print("forward %.3f ms" % t(lambda: conv(h)))
y = conv(h); g = torch.randn_like(y)
def fb():
    y = conv(h); y.backward(g)
print("forward + backward %.3f ms" % t(fb))
PY
