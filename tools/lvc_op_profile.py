"""The LVC operator's kernels one by one: run under `rocprofv3 --kernel-trace --stats` (tools/history/gpu_r2_s10.sh).  B = 20, T = 100."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import fastdiff_amd

B, T = 20, 100
for hop in (8, 64, 256):
    L = T * hop
    y = torch.randn(B, 32, L, device="cuda", requires_grad=True)
    k = (0.1 * torch.randn(B, 32, 64, 3, T, device="cuda")).requires_grad_(True)
    b = torch.randn(B, 64, T, device="cuda", requires_grad=True)
    d = torch.randn(B, 64, L, device="cuda")
    for _ in range(5):
        y.grad = k.grad = b.grad = None
        fastdiff_amd.location_variable_convolution(y, k, b, 1, hop).backward(d)
    torch.cuda.synchronize()
