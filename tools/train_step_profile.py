"""The training step of tools/train_step_probe.py (HIP operators) run N times, for `rocprofv3 --kernel-trace --stats`:
    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o train -- python tools/train_step_profile.py [steps]
then  python tools/train_step_profile.py --report <dir>/..._kernel_trace.csv <steps>  prints the per-step time by kernel family."""
import csv
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def family(name):
    for pat, fam in ((r"k_lvc|k_kc_|k_gate|fdk", "this repo's HIP kernels"), (r"igemm_|miopen|Sp3AsmConv|naive_conv|gridwise", "MIOpen convolutions"),
                     (r"batched_transpose|transpose", "MIOpen layout transposes"), (r"Cijk_|rocblas|gemm", "rocBLAS / hipBLASLt"),
                     (r"direct_copy|CatArray|copy_", "torch copies (contiguous, cat, stack)"), (r"reduce_kernel", "torch reductions (bias grads, norms, loss)"),
                     (r"weight_norm", "torch weight-norm"), (r"leaky_relu", "torch leaky_relu fwd/bwd"),
                     (r"elementwise|vectorized|SubTensor|fill|index", "torch elementwise / index / fill")):
        if re.search(pat, name):
            return fam
    return "other"


def report(path, steps, warm=4):
    """path: rocprofv3's *_kernel_trace.csv of `steps` training steps; the first `warm` (MIOpen's find runs, allocator growth) are left
    out."""
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "k_kc_fwd" in r[2]]
    assert len(marks) == 3 * steps, (len(marks), steps)
    # a step begins with the diffusion-step embedding (calc_diffusion_step_embedding: the only sin kernel of a step) and the zero_grad /
    # loss host work in front of it: cut at the longest launch gap among the dozen launches that precede step `warm`'s sin kernel
    sins = [i for i, r in enumerate(rows) if "sin_kernel" in r[2]]
    assert len(sins) == steps, (len(sins), steps)
    first = sins[warm]
    gaps = [(rows[i + 1][0] - rows[i][1], i + 1) for i in range(max(first - 12, 0), first)]
    start = max(gaps)[1]
    rows = rows[start:]
    n = steps - warm
    fam, top = {}, {}
    for s0, e0, name in rows:
        t = (e0 - s0) / 1e6 / n
        a = fam.setdefault(family(name), [0.0, 0.0])
        a[0] += t
        a[1] += 1.0 / n
        b = top.setdefault(name, [0.0, 0.0])
        b[0] += t
        b[1] += 1.0 / n
    total = sum(v[0] for v in fam.values())
    wall = (rows[-1][1] - rows[0][0]) / 1e6 / n
    print(f"kernel time per training step: {total:.2f} ms in {sum(v[1] for v in fam.values()):.0f} launches (first kernel to last: {wall:.2f} ms per step; steps {warm}..{steps - 1})")
    for f, (t, c) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        print(f"  {f:45s} {t:6.2f} ms  {100 * t / total:5.1f} %  {c:6.0f} launches")
    print("top kernels:")
    for name, (t, c) in sorted(top.items(), key=lambda kv: -kv[1][0])[:30]:
        print(f"  {t:6.3f} ms  {c:5.0f} x  {name[:150]}")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--report":
        return report(sys.argv[2], int(sys.argv[3]))
    import torch
    import torch.nn.functional as F

    import fastdiff_amd
    from fastdiff_amd import train
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    B, T = 20, 100
    torch.manual_seed(0)
    m = fastdiff_amd.FastDiff().cuda().train()
    mel = (torch.rand(B, 80, T) * 7.5 - 6.0).cuda()
    x = (0.3 * torch.randn(B, 1, T * 256)).cuda()
    z = torch.randn(B, 1, T * 256).cuda()
    t = torch.randint(1000, (B, 1)).float().cuda()
    for _ in range(steps):
        m.zero_grad(set_to_none=True)
        F.mse_loss(train.differentiable_forward(m, (x, mel, t)), z).backward()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
