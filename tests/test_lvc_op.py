"""The LVC operator with its gradients: oracle/lvc_grad.py (numpy) and the HIP operator behind
fastdiff_amd.location_variable_convolution against tests/golden/lvc_grad.npz -- TimeAware_LVCBlock.location_variable_convolution
(modules.py:220-253) executed on the reference module in float64, forward and, through torch.autograd, backward."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import lvc_grad as lg   # noqa: E402

HOPS = (8, 64, 256)


def _case(g, hop):
    t = f"h{hop}_"
    return {k: g[t + k] for k in ("x", "K", "bias", "dout", "out", "dx", "dK", "dbias")}


@pytest.mark.parametrize("hop", HOPS)
def test_oracle_restatement_matches_the_reference_function_and_its_autograd(hop):
    c = _case(load_golden("lvc_grad"), hop)
    x, K, b, d = (c[k].astype(np.float64) for k in ("x", "K", "bias", "dout"))
    out = lg.lvc_forward(x, K, b, hop)
    dx, dK, db = lg.lvc_backward(x, K, d, hop)
    for name, got in (("out", out), ("dx", dx), ("dK", dK), ("dbias", db)):
        ref = c[name].astype(np.float64)
        assert got.shape == ref.shape, name
        assert np.abs(got - ref).max() <= 2e-7 * max(1.0, np.abs(ref).max()), name      # the fixture is float32 of a float64 run


def test_oracle_gradients_against_finite_differences():
    rng = np.random.default_rng(0)
    B, Cin, Cout, ks, T, hop = 1, 3, 4, 3, 3, 5
    x, K, b = rng.standard_normal((B, Cin, T * hop)), rng.standard_normal((B, Cin, Cout, ks, T)), rng.standard_normal((B, Cout, T))
    d = rng.standard_normal((B, Cout, T * hop))
    dx, dK, db = lg.lvc_backward(x, K, d, hop)
    f = lambda x_, K_, b_: float((lg.lvc_forward(x_, K_, b_, hop) * d).sum())      # noqa: E731
    eps = 1e-6
    for arr, grad, which in ((x, dx, 0), (K, dK, 1), (b, db, 2)):
        for idx in [tuple(rng.integers(0, s) for s in arr.shape) for _ in range(12)]:
            ap, am = arr.copy(), arr.copy()
            ap[idx] += eps
            am[idx] -= eps
            args_p, args_m = [x, K, b], [x, K, b]
            args_p[which], args_m[which] = ap, am
            num = (f(*args_p) - f(*args_m)) / (2 * eps)
            assert abs(num - grad[idx]) < 1e-6 * max(1.0, abs(num)), (which, idx)


@pytest.mark.gpu
@pytest.mark.parametrize("hop", HOPS)
def test_hip_operator_forward_and_backward_match_the_reference(hop):
    import fastdiff_amd
    c = _case(load_golden("lvc_grad"), hop)
    x, K, b = (torch.from_numpy(c[k]).cuda().requires_grad_(True) for k in ("x", "K", "bias"))
    y = fastdiff_amd.location_variable_convolution(x, K, b, 1, hop)
    y.backward(torch.from_numpy(c["dout"]).cuda())
    for name, got in (("out", y.detach()), ("dx", x.grad), ("dK", K.grad), ("dbias", b.grad)):
        ref = c[name].astype(np.float64)
        err = np.abs(got.cpu().numpy().astype(np.float64) - ref).max()
        print(f"hop {hop} {name}: max|d| {err:.2e} of {np.abs(ref).max():.2f}")
        assert got.shape == ref.shape and err <= 2e-6 * max(1.0, np.abs(ref).max()), name      # fp32 sums of <= 256 products


@pytest.mark.gpu
def test_hip_operator_in_place_of_the_reference_method_under_autograd():
    """The operator dropped into a small autograd graph the way the reference uses it (a conv in front, the gate behind,
    modules.py:208-217): gradients of the conv weight and of the predicted kernels against the numpy oracle chain."""
    import fastdiff_amd
    torch.manual_seed(3)
    B, T, hop = 2, 4, 8
    conv = torch.nn.Conv1d(32, 32, 3, padding=1).cuda()
    xin = torch.randn(B, 32, T * hop, device="cuda")
    K = (0.1 * torch.randn(B, 32, 64, 3, T, device="cuda")).requires_grad_(True)
    bias = torch.zeros(B, 64, T, device="cuda", requires_grad=True)
    y = torch.nn.functional.leaky_relu(conv(torch.nn.functional.leaky_relu(xin, 0.2)), 0.2)
    z = fastdiff_amd.location_variable_convolution(y, K, bias, 1, hop)
    out = xin + torch.sigmoid(z[:, :32]) * torch.tanh(z[:, 32:])
    loss = (out ** 2).mean()
    loss.backward()
    # the same graph with the operator replaced by its numpy restatement (forward value and backward through lg.lvc_backward)
    yn, Kn, bn = y.detach().cpu().double().numpy(), K.detach().cpu().double().numpy(), bias.detach().cpu().double().numpy()
    zn = torch.from_numpy(lg.lvc_forward(yn, Kn, bn, hop)).requires_grad_(True)
    outn = xin.detach().cpu().double() + torch.sigmoid(zn[:, :32]) * torch.tanh(zn[:, 32:])
    (outn ** 2).mean().backward()
    dy, dK, db = lg.lvc_backward(yn, Kn, zn.grad.numpy(), hop)
    assert np.abs(K.grad.cpu().numpy() - dK).max() <= 1e-5 * max(1e-3, np.abs(dK).max())
    assert np.abs(bias.grad.cpu().numpy() - db).max() <= 1e-5 * max(1e-3, np.abs(db).max())
    assert conv.weight.grad is not None and torch.isfinite(conv.weight.grad).all() and float(conv.weight.grad.abs().max()) > 0
    with pytest.raises(AssertionError, match="not matched"):
        fastdiff_amd.location_variable_convolution(y, K, bias, 1, hop + 1)            # modules.py:236
    with pytest.raises(NotImplementedError):
        fastdiff_amd.location_variable_convolution(y, K, bias, 2, hop)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fastdiff_amd.location_variable_convolution(y.cpu(), K.cpu(), bias.cpu(), 1, hop)


@pytest.mark.gpu
@pytest.mark.parametrize("hop,T,B", [(8, 70, 2), (64, 70, 2), (256, 5, 3), (8, 1, 1), (256, 1, 1)])
def test_matrix_pipe_kernels_against_the_numpy_oracle_at_ragged_sizes(hop, T, B):
    """The model's shape (Cin 32, Cout 64, ks 3) runs on the fp32 matrix instruction behind two tiled transposes: T beyond one
    64-frame transpose tile, T not a multiple of the four frames a workgroup owns, and the single-frame case (no frame boundary)."""
    import fastdiff_amd
    rng = np.random.default_rng(hop * 1000 + T)
    x, K = rng.standard_normal((B, 32, T * hop)), 0.2 * rng.standard_normal((B, 32, 64, 3, T))
    b, d = rng.standard_normal((B, 64, T)), rng.standard_normal((B, 64, T * hop))
    ref = lg.lvc_forward(x, K, b, hop)
    rdx, rdK, rdb = lg.lvc_backward(x, K, d, hop)
    xt, Kt, bt = (torch.from_numpy(a.astype(np.float32)).cuda().requires_grad_(True) for a in (x, K, b))
    y = fastdiff_amd.location_variable_convolution(xt, Kt, bt, 1, hop)
    y.backward(torch.from_numpy(d.astype(np.float32)).cuda())
    for name, got, want in (("out", y.detach(), ref), ("dx", xt.grad, rdx), ("dK", Kt.grad, rdK), ("dbias", bt.grad, rdb)):
        err = np.abs(got.cpu().numpy().astype(np.float64) - want).max()
        assert got.shape == want.shape and err <= 3e-6 * max(1.0, np.abs(want).max()), (name, err)


@pytest.mark.gpu
def test_other_shapes_take_the_generic_kernels():
    import fastdiff_amd
    rng = np.random.default_rng(5)
    B, Cin, Cout, ks, T, hop = 2, 6, 10, 5, 7, 12
    x, K = rng.standard_normal((B, Cin, T * hop)), rng.standard_normal((B, Cin, Cout, ks, T))
    b, d = rng.standard_normal((B, Cout, T)), rng.standard_normal((B, Cout, T * hop))
    ref = lg.lvc_forward(x, K, b, hop)
    rdx, rdK, rdb = lg.lvc_backward(x, K, d, hop)
    xt, Kt, bt = (torch.from_numpy(a.astype(np.float32)).cuda().requires_grad_(True) for a in (x, K, b))
    y = fastdiff_amd.location_variable_convolution(xt, Kt, bt, 1, hop)
    y.backward(torch.from_numpy(d.astype(np.float32)).cuda())
    for name, got, want in (("out", y.detach(), ref), ("dx", xt.grad, rdx), ("dK", Kt.grad, rdK), ("dbias", bt.grad, rdb)):
        err = np.abs(got.cpu().numpy().astype(np.float64) - want).max()
        assert got.shape == want.shape and err <= 3e-6 * max(1.0, np.abs(want).max()), (name, err)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 32, 100 * 256), (3, 5, 37), (1, 32, 8 * 7)])
def test_gate_operator_forward_and_backward_match_torch_autograd(shape):
    """fastdiff_amd.gated_residual = x + sigmoid(y[:, :C]) * tanh(y[:, C:]) (modules.py:217) as one HIP pass each way, against the same
    line on torch autograd in float64 (values include saturated arguments, where tanh' and sigmoid' vanish)."""
    import fastdiff_amd
    B, C, L = shape
    g = torch.Generator().manual_seed(B * 1000 + L)
    x = torch.randn(B, C, L, generator=g)
    y = torch.randn(B, 2 * C, L, generator=g) * 4.0
    y[0, 0, :8] = torch.tensor([-120.0, 120.0, -30.0, 30.0, 0.0, 1e-4, -88.0, 88.0])
    y[0, C, :8] = torch.tensor([60.0, -60.0, 0.0, 20.0, -20.0, 1e-4, 50.0, -50.0])
    dout = torch.randn(B, C, L, generator=g)
    x64, y64 = x.double().requires_grad_(True), y.double().requires_grad_(True)
    ref = x64 + torch.sigmoid(y64[:, :C]) * torch.tanh(y64[:, C:])
    ref.backward(dout.double())
    xg, yg = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    out = fastdiff_amd.gated_residual(xg, yg)
    out.backward(dout.cuda())
    assert out.dtype == torch.float32 and torch.isfinite(out).all() and torch.isfinite(yg.grad).all()
    assert float((out.double().cpu() - ref.detach()).abs().max()) < 2e-6
    assert torch.equal(xg.grad.cpu(), dout)                                  # d out / d x is the identity
    assert float((yg.grad.double().cpu() - y64.grad).abs().max()) < 2e-6 * max(1.0, float(y64.grad.abs().max()))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fastdiff_amd.gated_residual(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 256, 100), (2, 128, 37), (1, 384, 128), (2, 128, 1), (20, 64, 100), (2, 96, 50), (3, 32, 7)])
def test_kernel_conv_operator_forward_and_backward_match_torch_autograd(shape):
    """fastdiff_amd.kernel_conv1d = the predictor's kernel_conv, Conv1d(64 -> M, k3, pad 1) (modules.py:315-318,330-331), forward and
    its three gradients on fp32-MFMA HIP kernels, against torch's conv1d and autograd in float64 (T a multiple of 4 and not, one
    frame, a last column tile of every fill; M = 64 / 96 / 32: the predictor's residual convolutions and other row counts that leave
    waves of the last 128-row workgroup without rows)."""
    import fastdiff_amd
    import torch.nn.functional as F
    B, M, T = shape
    g = torch.Generator().manual_seed(M + T)
    x = torch.randn(B, 64, T, generator=g)
    w = torch.randn(M, 64, 3, generator=g) / 14.0
    bias = torch.randn(M, generator=g)
    dout = torch.randn(B, M, T, generator=g)
    x64, w64, b64 = (t.double().requires_grad_(True) for t in (x, w, bias))
    ref = F.conv1d(x64, w64, b64, padding=1)
    ref.backward(dout.double())
    xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, bias))
    out = fastdiff_amd.kernel_conv1d(xg, wg, bg)
    out.backward(dout.cuda())
    rel = lambda got, want: float((got.double().cpu() - want).abs().max()) / max(1.0, float(want.abs().max()))      # noqa: E731
    assert rel(out, ref.detach()) < 2e-6
    assert rel(xg.grad, x64.grad) < 3e-6 and rel(wg.grad, w64.grad) < 3e-6 and rel(bg.grad, b64.grad) < 3e-6
    # only one gradient asked for: the others' kernels are not run
    x2 = x.cuda().requires_grad_(True)
    fastdiff_amd.kernel_conv1d(x2, w.cuda(), bias.cuda()).backward(dout.cuda())
    assert rel(x2.grad, x64.grad) < 3e-6
    with pytest.raises(NotImplementedError, match="128"):          # more than 128 frames: refused, never silently computed elsewhere
        fastdiff_amd.kernel_conv1d(torch.zeros(1, 64, 200).cuda(), w.cuda(), bias.cuda())


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 64, 100), (2, 64, 37), (20, 64, 100), (2, 256, 1), (1, 96, 128)])
def test_kernel_conv_with_its_activation_inside_matches_torch_autograd(shape):
    """kernel_conv1d(..., post_slope = 0.1) = `Conv1d(64 -> 64, k3), LeakyReLU(0.1)` of the predictor's residual stack (modules.py:296-314)
    as ONE operator each way: output and the three gradients against the two torch modules in float64, zeros and negative values of the
    pre-activation included; the plain operator followed by torch's leaky_relu gives the same bits (same kernels, the activation in the
    store / in the load of dout).  M > 512 (kernel_conv itself has no activation) is refused."""
    import fastdiff_amd
    import torch.nn.functional as F
    B, M, T = shape
    g = torch.Generator().manual_seed(3 * M + T)
    x = torch.randn(B, 64, T, generator=g)
    w = torch.randn(M, 64, 3, generator=g) / 14.0
    bias = torch.randn(M, generator=g)
    w[0] = 0.0
    bias[0] = 0.0                                                                   # a row whose pre-activation is exactly 0
    dout = torch.randn(B, M, T, generator=g)
    x64, w64, b64 = (t.double().requires_grad_(True) for t in (x, w, bias))
    ref = F.leaky_relu(F.conv1d(x64, w64, b64, padding=1), 0.1)
    ref.backward(dout.double())
    xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, bias))
    out = fastdiff_amd.kernel_conv1d(xg, wg, bg, 0.1)
    out.backward(dout.cuda())
    rel = lambda got, want: float((got.double().cpu() - want).abs().max()) / max(1.0, float(want.abs().max()))      # noqa: E731
    assert rel(out, ref.detach()) < 2e-6
    assert rel(xg.grad, x64.grad) < 3e-6 and rel(wg.grad, w64.grad) < 3e-6 and rel(bg.grad, b64.grad) < 3e-6
    xp, wp, bp = (t.cuda().requires_grad_(True) for t in (x, w, bias))
    out2 = F.leaky_relu(fastdiff_amd.kernel_conv1d(xp, wp, bp), 0.1)
    out2.backward(dout.cuda())
    assert torch.equal(out2.detach(), out.detach())
    assert torch.equal(xp.grad, xg.grad) and torch.equal(wp.grad, wg.grad) and torch.equal(bp.grad, bg.grad)
    with pytest.raises(NotImplementedError, match="512"):
        fastdiff_amd.kernel_conv1d(torch.zeros(1, 64, 8).cuda(), torch.zeros(1024, 64, 3).cuda(), torch.zeros(1024).cuda(), 0.1)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,n", [(3, 100, 6), (2, 37, 2), (1, 128, 1), (2, 1, 3), (20, 100, 6)])
def test_residual_stack_as_one_node_equals_the_pairs_one_by_one(B, T, n):
    """lvc_op.kernel_conv_stack = n `Conv1d(64, 64, 3), LeakyReLU(0.1)` pairs (the predictor's residual stack, modules.py:297-314) as one
    autograd node whose backward hands the gradient down already multiplied by the activation mask of the pair below: against the
    pairs one by one on kernel_conv1d(..., post_slope) -- same kernels, the mask applied one kernel earlier: bit-equal -- and against
    torch in float64."""
    import fastdiff_amd
    from fastdiff_amd import lvc_op
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(B + T + n)
    x = torch.randn(B, 64, T, generator=g)
    ws = [torch.randn(64, 64, 3, generator=g) / 9.0 for _ in range(n)]
    bs = [torch.randn(64, generator=g) * 0.3 for _ in range(n)]
    dout = torch.randn(B, 64, T, generator=g)
    x64 = x.double().requires_grad_(True)
    p64 = [(w.double().requires_grad_(True), b.double().requires_grad_(True)) for w, b in zip(ws, bs)]
    h = x64
    for w, b in p64:
        h = F.leaky_relu(F.conv1d(h, w, b, padding=1), 0.1)
    h.backward(dout.double())
    xa = x.cuda().requires_grad_(True)
    pa = [(w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)) for w, b in zip(ws, bs)]
    ya = lvc_op.kernel_conv_stack(xa, [w for w, _ in pa], [b for _, b in pa], 0.1)
    ya.backward(dout.cuda())
    xb = x.cuda().requires_grad_(True)
    pb = [(w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)) for w, b in zip(ws, bs)]
    yb = xb
    for w, b in pb:
        yb = fastdiff_amd.kernel_conv1d(yb, w, b, 0.1)
    yb.backward(dout.cuda())
    assert torch.equal(ya.detach(), yb.detach()) and torch.equal(xa.grad, xb.grad)
    for (wa, ba), (wb, bb) in zip(pa, pb):
        assert torch.equal(wa.grad, wb.grad) and torch.equal(ba.grad, bb.grad)
    rel = lambda got, want: float((got.double().cpu() - want).abs().max()) / max(1.0, float(want.abs().max()))      # noqa: E731
    assert rel(ya.detach(), h.detach()) < 5e-6 and rel(xa.grad, x64.grad) < 1e-5
    for (wa, ba), (w6, b6) in zip(pa, p64):
        assert rel(wa.grad, w6.grad) < 1e-5 and rel(ba.grad, b6.grad) < 1e-5
    # only the parameters need gradients (the stack's input does not): no dx of the bottom pair
    y2 = lvc_op.kernel_conv_stack(x.cuda(), [w.detach().requires_grad_(True) for w, _ in pa], [b for _, b in pa], 0.1)
    y2.backward(dout.cuda())


@pytest.mark.gpu
@pytest.mark.parametrize("P,n,B,T", [(3, 6, 20, 100), (2, 2, 3, 37), (3, 1, 1, 128), (8, 3, 2, 5)])
def test_predictor_front_ends_side_by_side_equal_one_by_one(P, n, B, T):
    """lvc_op.predictor_fronts: the front ends of P KernelPredictors (input convolution + activation, n Conv1d + LeakyReLU pairs, c + r) with
    one launch per chain step for all P, against input_conv + kernel_conv_stack + add per predictor -- same kernels, the pointers of the P
    convolutions as kernel arguments: outputs and every gradient bit for bit; one predictor's output unused: its gradients are zero."""
    from fastdiff_amd import lvc_op
    g = torch.Generator().manual_seed(P * 100 + n * 10 + T)
    xs = [torch.randn(B, 80, T, generator=g).cuda() for _ in range(P)]
    ic = [((torch.randn(64, 80, 5, generator=g) / 20).cuda(), torch.randn(64, generator=g).cuda()) for _ in range(P)]
    st = [[((torch.randn(64, 64, 3, generator=g) / 9).cuda(), (torch.randn(64, generator=g) * 0.3).cuda()) for _ in range(n)] for _ in range(P)]
    douts = [torch.randn(B, 64, T, generator=g).cuda() for _ in range(P)]
    for unused in (None, P - 1):
        leaf = lambda t: t.clone().requires_grad_(True)      # noqa: E731
        xa, ica, sta = [leaf(x) for x in xs], [(leaf(w), leaf(b)) for w, b in ic], [[(leaf(w), leaf(b)) for w, b in s_] for s_ in st]
        outs = lvc_op.predictor_fronts(xa, ica, sta, 0.1)
        sum((o * d).sum() for p, (o, d) in enumerate(zip(outs, douts)) if p != unused).backward()
        xb, icb, stb = [leaf(x) for x in xs], [(leaf(w), leaf(b)) for w, b in ic], [[(leaf(w), leaf(b)) for w, b in s_] for s_ in st]
        for p in range(P):
            c = lvc_op.input_conv(xb[p], icb[p][0], icb[p][1], 0.1)
            o = c + lvc_op.kernel_conv_stack(c, [w for w, _ in stb[p]], [b for _, b in stb[p]], 0.1)
            assert torch.equal(o.detach(), outs[p].detach()), p
            if p == unused:
                assert float(xa[p].grad.abs().max()) == 0.0 and float(ica[p][0].grad.abs().max()) == 0.0 and float(sta[p][0][0].grad.abs().max()) == 0.0
                continue
            (o * douts[p]).sum().backward()
            assert torch.equal(xa[p].grad, xb[p].grad), p
            assert torch.equal(ica[p][0].grad, icb[p][0].grad) and torch.equal(ica[p][1].grad, icb[p][1].grad), p
            for j in range(n):
                assert torch.equal(sta[p][j][0].grad, stb[p][j][0].grad) and torch.equal(sta[p][j][1].grad, stb[p][j][1].grad), (p, j)


@pytest.mark.gpu
@pytest.mark.parametrize("P,M,B,T", [(3, 256, 20, 100), (2, 64, 3, 37), (8, 96, 1, 5)])
def test_small_convolutions_side_by_side_equal_one_by_one(P, M, B, T):
    """lvc_op.kernel_conv1d_side_by_side (the three predictors' bias_conv in one launch per kernel) against kernel_conv1d per
    convolution: outputs and gradients bit for bit."""
    import fastdiff_amd
    from fastdiff_amd import lvc_op
    g = torch.Generator().manual_seed(P + M + T)
    xs = [torch.randn(B, 64, T, generator=g).cuda() for _ in range(P)]
    ws = [(torch.randn(M, 64, 3, generator=g) / 14).cuda() for _ in range(P)]
    bs = [torch.randn(M, generator=g).cuda() for _ in range(P)]
    douts = [torch.randn(B, M, T, generator=g).cuda() for _ in range(P)]
    leaf = lambda t: t.clone().requires_grad_(True)      # noqa: E731
    xa, wa, ba = [leaf(t) for t in xs], [leaf(t) for t in ws], [leaf(t) for t in bs]
    outs = lvc_op.kernel_conv1d_side_by_side(xa, wa, ba)
    sum((o * d).sum() for o, d in zip(outs, douts)).backward()
    for p in range(P):
        xb, wb, bb = leaf(xs[p]), leaf(ws[p]), leaf(bs[p])
        o = fastdiff_amd.kernel_conv1d(xb, wb, bb)
        o.backward(douts[p])
        assert torch.equal(o.detach(), outs[p].detach()), p
        assert torch.equal(xa[p].grad, xb.grad) and torch.equal(wa[p].grad, wb.grad) and torch.equal(ba[p].grad, bb.grad), p


@pytest.mark.gpu
@pytest.mark.parametrize("B,L,f", [(2, 1024, 4), (3, 96, 8), (1, 8, 8), (2, 30, 3), (20, 25600, 4)])
def test_skip_fan_out_picks_and_adds_up_like_autograd(B, L, f):
    """lvc_op.skip_fan(x, f) = (x[..., ::f], x, x, x, x): the DBlock's nearest pick (F.interpolate to L / f, modules.py:128-131) and one
    alias of the skip tensor per LVC layer; the five gradients come back as ONE sum.  Against torch autograd on the same readers --
    all five, and with some of them unused (their gradient is then absent, not zero)."""
    from fastdiff_amd import lvc_op
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(L + f)
    x = torch.randn(B, 32, L, generator=g).cuda()
    ws = [torch.randn(B, 32, L, generator=g).cuda() for _ in range(4)]
    wp = torch.randn(B, 32, L // f, generator=g).cuda()
    for used in ((0, 1, 2, 3, "p"), (1, 3), ("p",), (0, "p")):
        xa = x.clone().requires_grad_(True)
        picked, *al = lvc_op.skip_fan(xa, f)
        assert torch.equal(picked.detach(), F.interpolate(x, size=L // f)) and all(torch.equal(a.detach(), x) for a in al)
        loss = sum((al[i] * ws[i]).sum() for i in used if i != "p") + ((picked * wp).sum() if "p" in used else 0.0)
        loss.backward()
        xb = x.clone().requires_grad_(True)
        lossb = sum((xb * ws[i]).sum() for i in used if i != "p") + ((F.interpolate(xb, size=L // f) * wp).sum() if "p" in used else 0.0)
        lossb.backward()
        assert float((xa.grad - xb.grad).abs().max()) <= 1e-6 * float(xb.grad.abs().max()), used


@pytest.mark.gpu
@pytest.mark.parametrize("B,T", [(20, 100), (3, 37), (1, 128), (2, 1), (2, 8), (3, 9)])
def test_input_conv_operator_forward_and_backward_match_torch_autograd(B, T):
    """lvc_op.input_conv = KernelPredictor.input_conv, `Conv1d(80, 64, 5, padding=2), LeakyReLU(0.1)` (modules.py:292-295), as one HIP
    operator each way, against the two torch modules in float64: output, dx, dW, db; frame counts that are not multiples of the
    8-frame thread tiles, one frame, the training shape; the weight gradient is a fixed-order sum (same bits on a second run)."""
    from fastdiff_amd import lvc_op
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(17 * B + T)
    x = torch.randn(B, 80, T, generator=g)
    w = torch.randn(64, 80, 5, generator=g) / 20.0
    bias = torch.randn(64, generator=g)
    w[0] = 0.0
    bias[0] = 0.0                                                                   # a channel whose pre-activation is exactly 0
    dout = torch.randn(B, 64, T, generator=g)
    x64, w64, b64 = (t.double().requires_grad_(True) for t in (x, w, bias))
    ref = F.leaky_relu(F.conv1d(x64, w64, b64, padding=2), 0.1)
    ref.backward(dout.double())
    assert lvc_op.input_conv_supported(x.cuda(), w.cuda())
    outs = []
    for _ in range(2):
        xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, bias))
        out = lvc_op.input_conv(xg, wg, bg, 0.1)
        out.backward(dout.cuda())
        outs.append((out.detach(), xg.grad, wg.grad, bg.grad))
    rel = lambda got, want: float((got.double().cpu() - want).abs().max()) / max(1.0, float(want.abs().max()))      # noqa: E731
    out, dx, dw, db = outs[0]
    assert rel(out, ref.detach()) < 2e-6
    assert rel(dx, x64.grad) < 3e-6 and rel(dw, w64.grad) < 3e-6 and rel(db, b64.grad) < 3e-6
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    with pytest.raises(NotImplementedError, match="128"):
        lvc_op.input_conv(torch.zeros(1, 80, 200).cuda(), w.cuda(), bias.cuda())


@pytest.mark.gpu
@pytest.mark.parametrize("B,L,dil,skip,post", [(2, 1024, 1, True, 0.2), (3, 388, 3, True, 0.2), (1, 2560, 9, True, 0.2), (2, 904, 27, True, 0.2),
                                                (2, 640, 1, False, 1.0), (1, 132, 2, False, 1.0), (3, 260, 4, False, 1.0), (1, 4, 27, True, 0.2),
                                                (20, 25600, 27, True, 0.2)])
def test_conv32_operator_forward_and_backward_match_torch_autograd(B, L, dil, skip, post):
    """fastdiff_amd.conv32 = one small convolution of the denoiser as the reference applies it -- `x += audio_down; y =
    leaky_relu(conv(leaky_relu(x, 0.2)), 0.2)` (modules.py:209-212) or `layer(leaky_relu(x, 0.2))` (modules.py:136-137) -- one HIP pass
    forward and one backward, against the same lines on torch autograd in float64: y, xs, and the gradients of x, skip, weight and
    bias, with a gradient arriving at xs from another reader as in the LVC layer (the gate).  Tile edges (lengths that are not
    multiples of the 256 / 128-column tiles), every dilation of the model, one quad of columns, and the training shape itself."""
    import fastdiff_amd
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(L + 7 * dil)
    x = torch.randn(B, 32, L, generator=g)
    sk = torch.randn(B, 32, L, generator=g) if skip else None
    x[0, 0, :4] = torch.tensor([0.0, -0.0, 1e-30, -1e-30])                       # the activation's kink
    if skip:
        sk[0, 0, :4] = 0.0
    w = torch.randn(32, 32, 3, generator=g) / 9.8
    bias = torch.randn(32, generator=g)
    gy, gxs = torch.randn(B, 32, L, generator=g), torch.randn(B, 32, L, generator=g)
    big = B * L > 100000
    dt = torch.float32 if big else torch.float64                                  # (the training shape: float32 on the GPU as the yardstick)
    dev = "cuda" if big else "cpu"
    x64, w64, b64 = (t.to(dev, dt).requires_grad_(True) for t in (x, w, bias))
    s64 = sk.to(dev, dt).requires_grad_(True) if skip else None
    xs64 = x64 + s64 if skip else x64
    y64 = F.conv1d(F.leaky_relu(xs64, 0.2), w64, b64, padding=dil, dilation=dil)
    if post != 1.0:
        y64 = F.leaky_relu(y64, post)
    ((y64 * gy.to(dev, dt)).sum() + ((xs64 * gxs.to(dev, dt)).sum() if skip else 0.0)).backward()
    xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, bias))
    sg = sk.cuda().requires_grad_(True) if skip else None
    out = fastdiff_amd.conv32(xg, wg, bg, dil, skip=sg, post_slope=post)
    if skip:
        xs, y = out
        ((y * gy.cuda()).sum() + (xs * gxs.cuda()).sum()).backward()
        assert torch.equal(xs.detach(), (xg + sg).detach())
        assert torch.equal(xg.grad, sg.grad)                                      # x and skip receive the same gradient
    else:
        y = out
        (y * gy.cuda()).sum().backward()
    rel = lambda got, want: float((got.double().cpu() - want.double().cpu()).abs().max()) / max(1.0, float(want.abs().max()))      # noqa: E731
    tol = 2e-5 if big else 3e-6
    assert rel(y.detach(), y64.detach()) < tol
    assert rel(xg.grad, x64.grad) < tol and rel(wg.grad, w64.grad) < tol and rel(bg.grad, b64.grad) < tol
    # the weight and bias gradients are summed in a fixed order: the same bits on a second run
    xg2, wg2, bg2 = (t.cuda().requires_grad_(True) for t in (x, w, bias))
    out2 = fastdiff_amd.conv32(xg2, wg2, bg2, dil, skip=None if not skip else sk.cuda().requires_grad_(True), post_slope=post)
    if skip:
        ((out2[1] * gy.cuda()).sum() + (out2[0] * gxs.cuda()).sum()).backward()
    else:
        (out2 * gy.cuda()).sum().backward()
    assert torch.equal(wg2.grad, wg.grad) and torch.equal(bg2.grad, bg.grad) and torch.equal(xg2.grad, xg.grad)


@pytest.mark.gpu
def test_conv32_refuses_what_it_has_no_kernel_for():
    import fastdiff_amd
    w, b = torch.zeros(32, 32, 3).cuda(), torch.zeros(32).cuda()
    with pytest.raises(NotImplementedError, match="multiple of 4"):
        fastdiff_amd.conv32(torch.zeros(1, 32, 6).cuda(), w, b, 1)
    with pytest.raises(NotImplementedError, match="dilation"):
        fastdiff_amd.conv32(torch.zeros(1, 32, 64).cuda(), w, b, 5)
    from fastdiff_amd.lvc_op import conv32_supported
    assert conv32_supported(torch.zeros(1, 32, 64).cuda(), w, 27) and not conv32_supported(torch.zeros(1, 32, 6).cuda(), w, 1)
    assert not conv32_supported(torch.zeros(1, 16, 64).cuda(), w, 1) and not conv32_supported(torch.zeros(1, 32, 64), w, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(24576, 64, 3), (32, 32, 3), (64, 80, 5), (32, 1, 7), (1, 32, 7), (32, 32, 1), (256, 64, 3)])
def test_weight_norm_operator_matches_torch(shape):
    """fastdiff_amd.lvc_op.weight_norm = torch._weight_norm(v, g, 0), the fold every Conv1d of the model evaluates on each training
    forward (FastDiff_model.py:115-122), and its backward, against torch in float64 on every weight shape of the model."""
    from fastdiff_amd.lvc_op import weight_norm
    g0 = torch.Generator().manual_seed(shape[0] + shape[1])
    v = torch.randn(*shape, generator=g0)
    g = torch.rand(shape[0], 1, 1, generator=g0) + 0.5
    dw = torch.randn(*shape, generator=g0)
    v64, g64 = v.double().requires_grad_(True), g.double().requires_grad_(True)
    torch._weight_norm(v64, g64, 0).backward(dw.double())
    vg, gg = v.cuda().requires_grad_(True), g.cuda().requires_grad_(True)
    w = weight_norm(vg, gg)
    w.backward(dw.cuda())
    rel = lambda got, want: float((got.double().cpu() - want).abs().max()) / max(1e-30, float(want.abs().max()))      # noqa: E731
    assert rel(w.detach(), torch._weight_norm(v64, g64, 0).detach()) < 1e-6
    assert rel(vg.grad, v64.grad) < 2e-6 and rel(gg.grad, g64.grad) < 2e-6 and gg.grad.shape == g.shape


@pytest.mark.gpu
def test_weight_norm_of_all_convolutions_in_one_operator():
    """lvc_op.weight_norm_all = torch._weight_norm(v, g, 0) for a list of tensors in one operator (the records travel as kernel
    arguments, 28 per launch): 60 tensors of the model's shapes -- more than two launches' worth --, against the one-tensor operator
    (same per-row code: same bits, forward and backward) and torch; a weight that takes no part in the loss gets zero gradients."""
    from fastdiff_amd import lvc_op
    g = torch.Generator().manual_seed(4)
    shapes = [(24576, 64, 3), (256, 64, 3), (64, 80, 5), (32, 1, 7), (1, 32, 7), (32, 32, 1)] + [(64, 64, 3)] * 24 + [(32, 32, 3)] * 30
    vs = [torch.randn(*sh, generator=g).cuda() for sh in shapes]
    gs = [(torch.rand(sh[0], 1, 1, generator=g) + 0.5).cuda() for sh in shapes]
    douts = [torch.randn(*sh, generator=g).cuda() for sh in shapes]
    unused = {7, 41}
    va, ga = [v.clone().requires_grad_(True) for v in vs], [x.clone().requires_grad_(True) for x in gs]
    ws = lvc_op.weight_norm_all(list(zip(va, ga)))
    sum((w * d).sum() for i, (w, d) in enumerate(zip(ws, douts)) if i not in unused).backward()
    for i, sh in enumerate(shapes):
        vb, gb = vs[i].clone().requires_grad_(True), gs[i].clone().requires_grad_(True)
        wb = lvc_op.weight_norm(vb, gb)
        assert torch.equal(ws[i].detach(), wb.detach()), i
        assert torch.allclose(wb.detach(), torch._weight_norm(vs[i], gs[i], 0), rtol=1e-6, atol=1e-7), i
        if i in unused:
            assert float(va[i].grad.abs().max()) == 0.0 and float(ga[i].grad.abs().max()) == 0.0
            continue
        (wb * douts[i]).sum().backward()
        assert torch.equal(va[i].grad, vb.grad) and torch.equal(ga[i].grad, gb.grad), i


@pytest.mark.gpu
@pytest.mark.parametrize("which,B,L", [(0, 2, 1024), (1, 2, 1024), (0, 3, 260), (1, 3, 260), (0, 1, 4), (1, 1, 4), (0, 20, 25600), (1, 20, 25600)])
def test_conv7_operators_forward_and_backward_match_torch_autograd(which, B, L):
    """fastdiff_amd.lvc_op.conv7 = first_audio_conv (Conv1d(1, 32, 7, padding 3), FastDiff_model.py:34-36) and final_conv (Conv1d(32, 1,
    7, padding 3), FastDiff_model.py:67-68) forward and their three gradients on HIP kernels, against torch's conv1d and autograd in
    float64 (the training shape: float32 on the GPU as the yardstick); tile edges, one quad of columns."""
    import torch.nn.functional as F
    from fastdiff_amd.lvc_op import conv7, conv7_supported
    g = torch.Generator().manual_seed(10 * L + which)
    cin, cout = (1, 32) if which == 0 else (32, 1)
    x = torch.randn(B, cin, L, generator=g)
    w = torch.randn(cout, cin, 7, generator=g) / 3.0
    bias = torch.randn(cout, generator=g)
    dy = torch.randn(B, cout, L, generator=g)
    big = B * L > 100000
    dt, dev = (torch.float32, "cuda") if big else (torch.float64, "cpu")
    x64, w64, b64 = (t.to(dev, dt).requires_grad_(True) for t in (x, w, bias))
    ref = F.conv1d(x64, w64, b64, padding=3)
    ref.backward(dy.to(dev, dt))
    xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, bias))
    assert conv7_supported(xg, wg)
    y = conv7(xg, wg, bg)
    y.backward(dy.cuda())
    rel = lambda got, want: float((got.double().cpu() - want.double().cpu()).abs().max()) / max(1.0, float(want.abs().max()))      # noqa: E731
    tol = 2e-5 if big else 3e-6
    assert y.shape == ref.shape and rel(y.detach(), ref.detach()) < tol
    assert rel(xg.grad, x64.grad) < tol and rel(wg.grad, w64.grad) < tol and rel(bg.grad, b64.grad) < tol
    x2 = x.cuda()                                                           # the training case: the audio needs no gradient
    w2, b2 = w.cuda().requires_grad_(True), bias.cuda().requires_grad_(True)
    conv7(x2, w2, b2).backward(dy.cuda())
    assert torch.equal(w2.grad, wg.grad) and torch.equal(b2.grad, bg.grad)  # fixed-order sums: the same bits


def test_split_layers_node_on_the_cpu():
    """lvc_op.split_layers (no HIP involved): the slices are views of the input, and when their gradients do NOT come back as slices of
    the node's own buffer (anything but the HIP operator with its grad_slot) the node stacks them, zeros for an unused layer -- what
    torch's unbind would return."""
    from fastdiff_amd.lvc_op import split_layers
    g = torch.Generator().manual_seed(3)
    k = torch.randn(2, 4, 3, 5, 3, 7, generator=g, dtype=torch.float64).requires_grad_(True)
    slices, slots = split_layers(k)
    assert len(slices) == 4 and all(s_.data_ptr() == k[:, i].data_ptr() and tuple(s_.shape) == (2, 3, 5, 3, 7) for i, s_ in enumerate(slices))
    assert [s_[1] for s_ in slots] == [0, 1, 2, 3] and all(s_[2] == tuple(k.shape) for s_ in slots)
    w = [torch.randn(2, 3, 5, 3, 7, generator=g, dtype=torch.float64) for _ in range(3)]
    sum((slices[i] * w[i]).sum() for i in range(3)).backward()
    k2 = k.detach().clone().requires_grad_(True)
    sum((k2.unbind(1)[i] * w[i]).sum() for i in range(3)).backward()
    assert torch.equal(k.grad, k2.grad) and not k.grad[:, 3].any()


@pytest.mark.gpu
@pytest.mark.parametrize("hop,T", [(8, 37), (64, 100), (256, 12)])
def test_lvc_operator_on_layer_slices_without_copies(hop, T):
    """fastdiff_amd.lvc_op.split_layers + location_variable_convolution(..., grad_slot=...): the four layers' kernels are slices
    kernels[:, i] of the predictor's [B, 4, 32, 64, 3, T] output, read where they lie (batch-strided) and their gradients written into
    the slices of ONE buffer that becomes the gradient of the whole tensor.  Must equal, bit for bit, the plain route (contiguous copy
    of every slice in, torch.stack of the four gradients out)."""
    import fastdiff_amd
    from fastdiff_amd.lvc_op import split_layers
    B = 3
    g = torch.Generator().manual_seed(hop + T)
    k = (0.1 * torch.randn(B, 4, 32, 64, 3, T, generator=g)).cuda()
    xs = [torch.randn(B, 32, T * hop, generator=g).cuda() for _ in range(4)]
    bs = [torch.randn(B, 64, T, generator=g).cuda() for _ in range(4)]
    ds = [torch.randn(B, 64, T * hop, generator=g).cuda() for _ in range(4)]
    ka = k.clone().requires_grad_(True)
    outs_a = [fastdiff_amd.location_variable_convolution(xs[i], ka[:, i], bs[i], 1, hop) for i in range(4)]
    sum((o * d).sum() for o, d in zip(outs_a, ds)).backward()
    kb = k.clone().requires_grad_(True)
    slices, slots = split_layers(kb)
    assert all(not s_.is_contiguous() and s_.data_ptr() == kb[:, i].data_ptr() for i, s_ in enumerate(slices))
    outs_b = [fastdiff_amd.location_variable_convolution(xs[i], slices[i], bs[i], 1, hop, grad_slot=slots[i]) for i in range(4)]
    sum((o * d).sum() for o, d in zip(outs_b, ds)).backward()
    for a, b in zip(outs_a, outs_b):
        assert torch.equal(a, b)
    assert torch.equal(ka.grad, kb.grad) and float(kb.grad.abs().max()) > 0
    # only three of the four layers used: the fourth's gradient is zero, the others as before (the fallback route of the split node)
    kc = k.clone().requires_grad_(True)
    slices, slots = split_layers(kc)
    sum((fastdiff_amd.location_variable_convolution(xs[i], slices[i], bs[i], 1, hop, grad_slot=slots[i]) * ds[i]).sum() for i in range(3)).backward()
    assert torch.equal(kc.grad[:, :3], ka.grad[:, :3]) and not kc.grad[:, 3].any()


@pytest.mark.gpu
@pytest.mark.parametrize("B,Lin,r", [(2, 100, 8), (3, 37, 8), (2, 130, 4), (1, 1, 4), (1, 2, 8), (2, 800, 8), (20, 6400, 4)])
def test_upsample_operator_forward_and_backward_match_torch_autograd(B, Lin, r):
    """fastdiff_amd.lvc_op.upsample = an LVC block's `self.upsample(F.leaky_relu(x, 0.2))` (modules.py:163-166,205-206: ConvTranspose1d(32,
    32, 2 r, stride r, padding r / 2)) forward and its three gradients on HIP kernels, against torch in float64 (the training shape:
    float32 on the GPU as the yardstick): tile edges (lengths that are not multiples of the 64 / 32-position tiles), one position."""
    import torch.nn.functional as F
    from fastdiff_amd.lvc_op import upsample
    g = torch.Generator().manual_seed(100 * Lin + r)
    x = torch.randn(B, 32, Lin, generator=g)
    x[0, 0, 0] = 0.0                                                              # the activation's kink
    w = torch.randn(32, 32, 2 * r, generator=g) / 8.0
    bias = torch.randn(32, generator=g)
    dy = torch.randn(B, 32, Lin * r, generator=g)
    big = B * Lin * r > 100000
    dt, dev = (torch.float32, "cuda") if big else (torch.float64, "cpu")
    x64, w64, b64 = (t.to(dev, dt).requires_grad_(True) for t in (x, w, bias))
    ref = F.conv_transpose1d(F.leaky_relu(x64, 0.2), w64, b64, stride=r, padding=r // 2)
    ref.backward(dy.to(dev, dt))
    xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, bias))
    y = upsample(xg, wg, bg, r)
    y.backward(dy.cuda())
    rel = lambda got, want: float((got.double().cpu() - want.double().cpu()).abs().max()) / max(1.0, float(want.abs().max()))      # noqa: E731
    tol = 2e-5 if big else 3e-6
    assert y.shape == ref.shape and rel(y.detach(), ref.detach()) < tol
    assert rel(xg.grad, x64.grad) < tol and rel(wg.grad, w64.grad) < tol and rel(bg.grad, b64.grad) < tol
    x2, w2, b2 = (t.cuda().requires_grad_(True) for t in (x, w, bias))           # fixed-order sums: the same bits again
    upsample(x2, w2, b2, r).backward(dy.cuda())
    assert torch.equal(w2.grad, wg.grad) and torch.equal(b2.grad, bg.grad) and torch.equal(x2.grad, xg.grad)


def test_frame_orders_are_permutations_and_round_trip():
    """fastdiff_amd.lvc_op.frame_order (the Python statement of csrc/fd_frame_order.h) lists every coefficient of a frame exactly once in
    both orders, and reference -> frames -> reference is the identity."""
    from fastdiff_amd import lvc_op
    for order in ("forward", "grad"):
        assert sorted(lvc_op.frame_order(order).tolist()) == list(range(6144)), order
    k = torch.randn(2, 4, 32, 64, 3, 5)
    for order in ("forward", "grad"):
        f = lvc_op.reference_to_frames(k, order)
        assert tuple(f.shape) == (2, 4, 5, 6144) and torch.equal(lvc_op.frames_to_reference(f, order), k)
    # forward order: the float4 of lane l, group g holds four consecutive k-steps of output row 32 (g / 12) + (l & 31)
    r = lvc_op.frame_order("forward").view(24, 64, 4)
    o = (r // 3) % 64
    assert torch.equal(o, (32 * (torch.arange(24) // 12)).view(24, 1, 1) + (torch.arange(64) & 31).view(1, 64, 1) + torch.zeros(24, 64, 4, dtype=torch.long))


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,layers", [(3, 100, 4), (2, 37, 4), (1, 128, 1), (2, 1, 2), (20, 100, 4)])
def test_kernel_conv_frames_equal_the_reference_layout(B, T, layers):
    """kernel_conv writing the operator's frames (fd_kconv_forward_frames) = the reference-layout kernel_conv (fd_kconv_forward) with the
    rows gathered and the store transposed: frames_to_reference(out) must equal it bit for bit; the backward reading gradient frames
    (fd_kconv_backward_frames) must give the bits of fd_kconv_backward fed with the same gradient in the reference's layout for the
    weight and bias gradients (rows are independent), and its sums in another order for dx (the 24576 rows are added up frame group by
    frame group instead of row by row: float32 rounding apart)."""
    import fastdiff_amd
    from fastdiff_amd import lvc_op
    g = torch.Generator().manual_seed(100 * B + T)
    M = layers * 6144
    x = torch.randn(B, 64, T, generator=g).cuda()
    w = (torch.randn(M, 64, 3, generator=g) / 13.9).cuda()
    bias = torch.randn(M, generator=g).cuda()
    dfr = torch.randn(B, layers, T, 6144, generator=g).cuda()                       # a gradient in the "grad" frame order
    xa, wa, ba = (t.clone().requires_grad_(True) for t in (x, w, bias))
    fr = lvc_op.kernel_conv1d_frames(xa, wa, ba)
    assert tuple(fr.shape) == (B, layers, T, 6144)
    fr.backward(dfr)
    xb, wb, bb = (t.clone().requires_grad_(True) for t in (x, w, bias))
    ref = fastdiff_amd.kernel_conv1d(xb, wb, bb)
    ref.backward(lvc_op.frames_to_reference(dfr, "grad").reshape(B, M, T))
    assert torch.equal(lvc_op.frames_to_reference(fr.detach(), "forward").reshape(B, M, T), ref.detach())
    assert torch.equal(wa.grad, wb.grad) and torch.equal(ba.grad, bb.grad)
    assert float((xa.grad - xb.grad).abs().max()) <= 2e-6 * float(xb.grad.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("hop,T,B", [(8, 37, 2), (64, 100, 2), (256, 12, 3), (256, 1, 1), (8, 128, 1)])
@pytest.mark.parametrize("dx_mode", ["gather", "copy"])
def test_lvc_operator_on_frames_equals_the_operator_on_the_reference_layout(hop, T, B, dx_mode):
    """fd_lvc_forward_frames / fd_lvc_backward_frames take one layer's [T, 6144] block per utterance out of a [B, layers, T, 6144] tensor
    where it lies and leave the kernel gradient as frames in the layer's slice of one buffer: output, dx, dbias and (after
    frames_to_reference) dK equal fd_lvc_forward / fd_lvc_backward on the reference's tensors bit for bit -- the same kernels minus
    the transposes.  dx_mode (library option lvc_dx): the dx kernel gathers its operands out of the forward-order frames (default) or
    reads a reordered copy."""
    import fastdiff_amd
    from fastdiff_amd import lvc_op
    lib, h = lvc_op._handle(torch.device("cuda"))
    assert lib.fd_set_option(h, b"lvc_dx", dx_mode.encode()) == 0
    g = torch.Generator().manual_seed(hop + T)
    layers = 4
    k6 = (torch.randn(B, layers, 32, 64, 3, T, generator=g) / 9.8).cuda()
    bias = torch.randn(B, layers, 64, T, generator=g).cuda()
    xs = [torch.randn(B, 32, T * hop, generator=g).cuda() for _ in range(layers)]
    douts = [torch.randn(B, 64, T * hop, generator=g).cuda() for _ in range(layers)]
    # the reference layout, layer by layer
    want = []
    for i in range(layers):
        x, k, b = xs[i].clone().requires_grad_(True), k6[:, i].clone().requires_grad_(True), bias[:, i].clone().requires_grad_(True)
        y = fastdiff_amd.location_variable_convolution(x, k, b, 1, hop)
        y.backward(douts[i])
        want.append((y.detach(), x.grad, k.grad, b.grad))
    # frames: one tensor for the four layers, split without copies, gradients into one buffer
    fr = lvc_op.reference_to_frames(k6, "forward").requires_grad_(True)
    slices, slots = lvc_op.split_layers(fr)
    bias_all = bias.clone().requires_grad_(True)                                    # bias_conv's [B, layers, 64, T] output: slices where they lie
    bslices, bslots = lvc_op.split_layers(bias_all)
    loss = 0.0
    got = []
    for i in range(layers):
        x = xs[i].clone().requires_grad_(True)
        y = lvc_op.location_variable_convolution_frames(x, slices[i], bslices[i], hop, grad_slot=slots[i], bias_slot=bslots[i])
        loss = loss + (y * douts[i]).sum()
        got.append((y, x))
    loss.backward()
    assert tuple(fr.grad.shape) == (B, layers, T, 6144) and tuple(bias_all.grad.shape) == (B, layers, 64, T)
    dk6 = lvc_op.frames_to_reference(fr.grad, "grad")
    for i in range(layers):
        y, x = got[i]
        assert torch.equal(y.detach(), want[i][0]), i
        assert torch.equal(x.grad, want[i][1]) and torch.equal(bias_all.grad[:, i], want[i][3]), i
        assert torch.equal(dk6[:, i], want[i][2]), i
    # a bias tensor of its own (no slot) goes the same way
    x, b = xs[0].clone().requires_grad_(True), bias[:, 0].clone().requires_grad_(True)
    y = lvc_op.location_variable_convolution_frames(x, fr.detach()[:, 0], b, hop)
    y.backward(douts[0])
    assert torch.equal(y.detach(), want[0][0]) and torch.equal(b.grad, want[0][3]) and torch.equal(x.grad, want[0][1])
    assert lib.fd_set_option(h, b"lvc_dx", b"gather") == 0
