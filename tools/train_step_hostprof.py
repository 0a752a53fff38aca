"""Where the HOST time of an eager training step goes (cProfile over 20 steps after warm-up): the step is launch-bound on the host
(~460 launches through ~200 autograd nodes), so Python / ctypes / autograd overhead per operator call is what eager mode pays."""
import cProfile
import os
import pstats
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import fastdiff_amd                      # noqa: E402
from fastdiff_amd import train           # noqa: E402


def main():
    B, T = 20, 100
    torch.manual_seed(0)
    m = fastdiff_amd.FastDiff().cuda().train()
    mel = (torch.rand(B, 80, T) * 7.5 - 6.0).cuda()
    x = (0.3 * torch.randn(B, 1, T * 256)).cuda()
    z = torch.randn(B, 1, T * 256).cuda()
    steps = torch.randint(1000, (B, 1)).float().cuda()

    def step():
        m.zero_grad(set_to_none=True)
        F.mse_loss(train.differentiable_forward(m, (x, mel, steps)), z).backward()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()
