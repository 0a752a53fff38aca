"""Training-side timing: the reference's training shape -- max_sentences 20 x max_samples 25600 (base.yaml:50-51),
i.e. B = 20 utterance crops of T = 100 frames -- through FastDiff.forward in train() mode + loss.backward(), with the location-variable
convolution (a) on the HIP operator (fd_lvc_forward / fd_lvc_backward) and (b) as the reference states it, pad + unfold + einsum on
PyTorch-ROCm (modules.py:220-253 with dilation 1), on the same GPU; then the operator alone, forward and backward, per hop size.
Usage: python tools/train_step_probe.py [B] [T]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
import torch.nn.functional as F

import fastdiff_amd
from fastdiff_amd import train


def lvc_unfold_einsum(x, kernel, bias, dilation, hop):
    """out[b,o,l*hop+s] = bias[b,o,l] + sum_{i,k} xpad[b,i,l*hop+s+k] kernel[b,i,o,k,l] the way eager PyTorch runs it."""
    B, _, L = x.shape
    win = F.pad(x, (1, 1)).unfold(2, hop + 2, hop).unfold(3, 3, 1)           # [B, in, T, hop, 3]
    out = torch.einsum("bithk,biokt->both", win, kernel) + bias.unsqueeze(-1)
    return out.reshape(B, -1, L)


def timed(fn, warm=2, reps=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    torch.manual_seed(0)
    m = fastdiff_amd.FastDiff().cuda().train()
    mel = (torch.rand(B, 80, T) * 7.5 - 6.0).cuda()
    x = (0.3 * torch.randn(B, 1, T * 256)).cuda()
    z = torch.randn(B, 1, T * 256).cuda()
    steps = torch.randint(1000, (B, 1)).float().cuda()

    def step(lvc):
        m.zero_grad(set_to_none=True)
        eps = train.differentiable_forward(m, (x, mel, steps), lvc=lvc)
        F.mse_loss(eps, z).backward()

    def fwd(lvc):
        with torch.no_grad():
            train.differentiable_forward(m, (x, mel, steps), lvc=lvc)

    if os.environ.get("FD_LVC_DX"):      # gather (default) | copy: how the frames path's dx kernel gets its operands
        from fastdiff_amd import lvc_op
        lib, h = lvc_op._handle(torch.device("cuda"))
        assert lib.fd_set_option(h, b"lvc_dx", os.environ["FD_LVC_DX"].encode()) == 0
    print(f"training shape B={B} T={T} ({B * T * 256} samples)" + (f", lvc_dx = {os.environ['FD_LVC_DX']}" if os.environ.get("FD_LVC_DX") else ""))
    # frames: kernel_conv hands the LVC operator its frame-major operands (the product path); reference tensor: through the reference's
    # [B, layers, 32, 64, 3, T] kernels and the operator's transposes (module._train_frames = False)
    for name, lvc, frames in (("HIP operator, frames", None, True), ("HIP operator, reference tensor", None, False),
                              ("unfold+einsum (PyTorch-ROCm eager)", lvc_unfold_einsum, True)):
        try:
            m._train_frames = frames
            print(f"  forward + backward, LVC = {name}: {timed(lambda: step(lvc)):8.2f} ms   forward only (no_grad): {timed(lambda: fwd(lvc)):8.2f} ms")
        except Exception as e:      # noqa: BLE001 -- e.g. out of memory in the unfold view's backward
            print(f"  LVC = {name}: failed: {e!r}")
    m._train_frames = True
    # the same step captured once in a hipGraph (torch.cuda.graph) and replayed: what is left when the ~500 kernel launches of a step
    # cost no host time.  Variants: FD_TRAIN_VARIANTS=1 also replays the step without frames / without the fused predictor activations
    def graph_run(label, **attrs):
        try:
            for k, v in attrs.items():
                setattr(m, k, v)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    step(None)
            torch.cuda.current_stream().wait_stream(side)
            m.zero_grad(set_to_none=True)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                eps = train.differentiable_forward(m, (x, mel, steps), lvc=None)
                loss = F.mse_loss(eps, z)
                loss.backward()
            graph.replay()
            torch.cuda.synchronize()
            g_graph = {n: p.grad.clone() for n, p in m.named_parameters()}
            loss_graph = float(loss.detach())
            step(None)
            torch.cuda.synchronize()
            worst = max(float((g_graph[n] - p.grad).abs().max()) / max(float(p.grad.abs().max()), 1e-20) for n, p in m.named_parameters())
            print(f"  forward + backward, LVC = HIP operator{label}, replayed from a hipGraph: {timed(graph.replay):8.2f} ms   eager: {timed(lambda: step(None)):8.2f} ms   (loss {loss_graph:.6f}; gradients vs the eager step: max relative difference {worst:.1e})")
        except Exception as e:      # noqa: BLE001
            print(f"  hipGraph capture of the training step{label} failed: {e!r}")
        finally:
            m._train_frames, m._train_fuse_act, m._train_skip_fan, m._train_stack, m._train_wn_all, m._train_fronts = True, True, True, True, True, True

    graph_run("")
    if os.environ.get("FD_TRAIN_VARIANTS"):
        graph_run(" [predictor front ends one by one]", _train_fronts=False)
        graph_run(" [weight-norm: one operator per convolution]", _train_wn_all=False)
        graph_run(" [residual stack: one node per pair]", _train_stack=False)
        graph_run(" [skip fan-out by autograd]", _train_skip_fan=False)
        graph_run(" [predictor activations as torch nodes]", _train_fuse_act=False)
        graph_run(" [reference kernel tensor + transposes]", _train_frames=False)
        graph_run("")
    for hop in (8, 64, 256):
        L = T * hop
        y = torch.randn(B, 32, L, device="cuda", requires_grad=True)
        k = (0.1 * torch.randn(B, 32, 64, 3, T, device="cuda")).requires_grad_(True)
        b = torch.randn(B, 64, T, device="cuda", requires_grad=True)
        d = torch.randn(B, 64, L, device="cuda")
        for name, op in (("HIP", fastdiff_amd.location_variable_convolution), ("unfold+einsum", lvc_unfold_einsum)):
            def f():
                with torch.no_grad():
                    op(y, k, b, 1, hop)

            def fb():
                y.grad = k.grad = b.grad = None
                op(y, k, b, 1, hop).backward(d)
            tf, tfb = timed(f), timed(fb)
            flops = 2.0 * B * L * 64 * 96
            print(f"  hop {hop:3d} {name:14s}: forward {tf:7.3f} ms ({flops / tf / 1e9:6.1f} TFLOP/s)   forward+backward {tfb:7.3f} ms ({3 * flops / tfb / 1e9:6.1f} TFLOP/s)")


if __name__ == "__main__":
    main()
