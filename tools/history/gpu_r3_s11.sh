#!/bin/bash
# round-3 session 11: kernel_conv operator (training path): parity vs torch autograd, training-step test, timing
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
echo "== tests"; timeout 900 python -m pytest tests/test_lvc_op.py tests/test_training_path.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -25 | cut -c1-300
echo "== kernel_conv alone, B=20 T=100 M=24576"
python - <<'PY'
import torch, time, fastdiff_amd
import torch.nn.functional as F
B, M, T = 20, 24576, 100
x = torch.randn(B, 64, T, device='cuda', requires_grad=True)
w = (torch.randn(M, 64, 3, device='cuda') / 14).requires_grad_(True)
b = torch.randn(M, device='cuda', requires_grad=True)
g = torch.randn(B, M, T, device='cuda')
def t(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for name, op in (("HIP", fastdiff_amd.kernel_conv1d), ("torch", lambda x, w, b: F.conv1d(x, w, b, padding=1))):
    with torch.no_grad():
        f = t(lambda: op(x, w, b))
    def fb():
        x.grad = w.grad = b.grad = None
        op(x, w, b).backward(g)
    print(f"{name:6s} forward {f:.3f} ms   forward + backward {t(fb):.3f} ms")
PY
echo "== training step"; timeout 600 python tools/train_step_probe.py 2>&1 | grep -v "Warning\|WeightNorm\|amdgpu.ids" | head -4
