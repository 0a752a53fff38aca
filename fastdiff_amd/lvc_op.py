"""The location-variable convolution as a differentiable operator on the MI355X.

`location_variable_convolution(x, kernel, bias, dilation, hop_size)` has the signature, tensor layouts and assert of
`TimeAware_LVCBlock.location_variable_convolution` (modules/FastDiff/module/modules.py:220-253); forward and backward run as HIP
kernels behind the C ABI (fd_lvc_forward / fd_lvc_backward, include/fastdiff_hip.h), everything around it stays on PyTorch autograd.
Binding it in the reference is one line:

    TimeAware_LVCBlock.location_variable_convolution = lambda self, x, k, b, d, h: fastdiff_amd.location_variable_convolution(x, k, b, d, h)

which puts the twelve LVC calls of every training forward/backward (theta_timestep_loss, util.py:291-325) on these kernels.
There is no CPU path: CPU tensors raise.
"""
import ctypes as ct

import threading

import torch

from . import _capi

_handles = {}
_handles_lock = threading.Lock()


def _handle(device):
    """One library handle per device: it supplies the device, the error text and the scratch buffers the backward kernels sum their
    partial results in (grown with hipFree + hipMalloc on first use of a larger shape).  Consequence, as for any handle of the library:
    the operators of this module must be ordered on ONE stream per device -- what torch.autograd does for a forward and its backward --
    and their first call at a new shape must not sit inside a graph capture (warm up before capturing).  Two models
    training concurrently on one device from different threads or streams need a process each."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _handles:
        with _handles_lock:
            if idx not in _handles:
                lib = _capi.load()
                cfg = _capi.FdConfig()
                lib.fd_default_config(ct.byref(cfg))
                h = ct.c_void_p()
                _capi.check(lib, None, lib.fd_create(ct.byref(cfg), idx, ct.byref(h)), "fd_create")
                _handles[idx] = h
    return _capi.load(), _handles[idx]


def _stream(device):
    return ct.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _batch_strided(k):
    """True if k [B, Cin, Cout, ks, T] is contiguous apart from its batch stride (one layer's slice of a [B, layers, ...] tensor)."""
    _, ci, co, ks, T = k.shape
    return k.stride()[1:] == (co * ks * T, ks * T, T, 1) and k.stride(0) >= ci * co * ks * T


class _LVC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kernel, bias, hop_size, grad_slot=None):
        if not (x.is_cuda and kernel.is_cuda and bias.is_cuda):
            raise RuntimeError("fastdiff_amd.location_variable_convolution runs only on a HIP device (no CPU fallback)")
        # the kernels compute in float32 (the reference trains in float32); other floating types are converted on the way in and
        # the gradients go back in each input's own type, as autograd requires
        ctx.in_dtypes = (x.dtype, kernel.dtype, bias.dtype)
        # split_layers (below) hands out slices of the predictor's [B, layers, ...] output and a slot for the gradient of each
        ctx.grad_slot = grad_slot
        B, Cin, L = x.shape
        _, _, Cout, ks, T = kernel.shape
        model_shape = (Cin, Cout, ks) == (32, 64, 3) and int(hop_size) in (8, 64, 256)
        x, bias = x.contiguous().float(), bias.contiguous().float()
        # a layer's slice stays where it lies, and its gradient may go into the shared slot, only for the shape whose kernels take a
        # batch stride (fd_lvc_*_strided: the model's own); every other shape runs on a contiguous copy and owns its gradient
        ctx.strided_ok = kernel.dtype == torch.float32 and model_shape and _batch_strided(kernel)
        if not ctx.strided_ok:
            kernel = kernel.contiguous().float()
        out = torch.empty((B, Cout, L), device=x.device, dtype=torch.float32)
        lib, h = _handle(x.device)
        _capi.check(lib, h, lib.fd_lvc_forward_strided(h, x.data_ptr(), kernel.data_ptr(), kernel.stride(0), bias.data_ptr(), B, Cin, Cout, ks, T,
                                                       int(hop_size), out.data_ptr(), _stream(x.device)), "fd_lvc_forward")
        ctx.save_for_backward(x, kernel)
        ctx.hop = int(hop_size)
        return out.to(ctx.in_dtypes[0])

    @staticmethod
    def backward(ctx, dout):
        x, kernel = ctx.saved_tensors
        dout = dout.contiguous().float()
        B, Cin, L = x.shape
        _, _, Cout, ks, T = kernel.shape
        need_x, need_k, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        dx = torch.empty_like(x) if need_x else None
        dk = None
        if need_k:
            slot = ctx.grad_slot
            if slot is not None and ctx.strided_ok:      # this layer's slice of the shared gradient buffer
                holder, i, shape = slot
                if holder.get("buf") is None:
                    holder["buf"] = torch.empty(shape, device=x.device, dtype=torch.float32)
                dk = holder["buf"][:, i]
            else:
                dk = torch.empty(kernel.shape, device=x.device, dtype=torch.float32)
        db = torch.empty((B, Cout, T), device=x.device, dtype=torch.float32) if need_b else None
        lib, h = _handle(x.device)
        _capi.check(lib, h, lib.fd_lvc_backward_strided(h, x.data_ptr(), kernel.data_ptr(), kernel.stride(0), dout.data_ptr(), B, Cin, Cout, ks, T,
                                                        ctx.hop, None if dx is None else dx.data_ptr(), None if dk is None else dk.data_ptr(),
                                                        0 if dk is None else dk.stride(0), None if db is None else db.data_ptr(), _stream(x.device)),
                    "fd_lvc_backward")
        tx, tk, tb = ctx.in_dtypes
        return (None if dx is None else dx.to(tx), None if dk is None else dk.to(tk), None if db is None else db.to(tb), None, None)


class _SplitLayers(torch.autograd.Function):
    """kernels [B, layers, Cin, Cout, ks, T] -> the `layers` slices kernels[:, i] (what the reference takes with `kernels[:, i, ...]`,
    modules.py:213-214), without a copy either way: the location-variable convolution reads a slice where it lies (batch-strided) and
    writes that layer's gradient into its slice of ONE buffer of the whole shape, which is then this node's gradient as it stands.
    (torch's own unbind / select give contiguous copies to the operator and stack the four gradients: 0.5 ms per training step.)"""

    @staticmethod
    def forward(ctx, k, holder):
        ctx.holder = holder
        return tuple(k[:, i] for i in range(k.shape[1]))

    @staticmethod
    def backward(ctx, *grads):
        buf = ctx.holder.get("buf")
        ctx.holder["buf"] = None
        if buf is not None and all(g is not None and g.data_ptr() == buf[:, i].data_ptr() and g.stride() == buf[:, i].stride() and g.dtype == buf.dtype
                                   for i, g in enumerate(grads)):
            return buf, None
        ref = next(g for g in grads if g is not None)
        return torch.stack([g if g is not None else torch.zeros_like(ref) for g in grads], 1), None


def split_layers(kernels):
    """(slices, slots): the per-layer slices kernels[:, i] of the predictor's kernels [B, layers, Cin, Cout, ks, T] and, for each, the
    `grad_slot` to hand to location_variable_convolution together with it."""
    holder = {"buf": None}
    slices = _SplitLayers.apply(kernels, holder)
    return slices, tuple((holder, i, tuple(kernels.shape)) for i in range(kernels.shape[1]))


class _Gate(torch.autograd.Function):
    """out = x + sigmoid(y[:, :C]) * tanh(y[:, C:]) (modules.py:217), one HIP pass forward (fd_gate_forward) and one backward."""

    @staticmethod
    def forward(ctx, x, y):
        if not (x.is_cuda and y.is_cuda):
            raise RuntimeError("fastdiff_amd.gated_residual runs only on a HIP device (no CPU fallback)")
        ctx.in_dtypes = (x.dtype, y.dtype)
        x, y = x.contiguous().float(), y.contiguous().float()
        B, C, L = x.shape
        assert tuple(y.shape) == (B, 2 * C, L), "gate: y must hold the sigmoid half and the tanh half of every channel of x"
        out = torch.empty_like(x)
        lib, h = _handle(x.device)
        _capi.check(lib, h, lib.fd_gate_forward(h, x.data_ptr(), y.data_ptr(), B, C, L, out.data_ptr(), _stream(x.device)), "fd_gate_forward")
        ctx.save_for_backward(y)
        return out.to(ctx.in_dtypes[0])

    @staticmethod
    def backward(ctx, dout):
        (y,) = ctx.saved_tensors
        dy = None
        if ctx.needs_input_grad[1]:
            g = dout.contiguous().float()
            B, C2, L = y.shape
            dy = torch.empty_like(y)
            lib, h = _handle(y.device)
            _capi.check(lib, h, lib.fd_gate_backward(h, y.data_ptr(), g.data_ptr(), B, C2 // 2, L, dy.data_ptr(), _stream(y.device)), "fd_gate_backward")
            dy = dy.to(ctx.in_dtypes[1])
        return (dout if ctx.needs_input_grad[0] else None), dy


def gated_residual(x, y):
    """x + sigmoid(y[:, :C]) * tanh(y[:, C:]) for x [B, C, L], y [B, 2C, L]: the last line of an LVC layer (modules.py:217) as one
    differentiable operator."""
    return _Gate.apply(x, y)


class _KConv(torch.autograd.Function):
    """KernelPredictor.kernel_conv (modules.py:315-318,330-331: Conv1d(64 -> M, k3, pad 1)) forward and backward on fp32-MFMA HIP
    kernels (fd_kconv_forward / fd_kconv_backward), the reference's layouts.  post_slope != 1 (M <= 512): the LeakyReLU the predictor
    puts behind its small convolutions (modules.py:296-314) inside the same launches (fd_kconv_forward_act / fd_kconv_backward_act)."""

    @staticmethod
    def forward(ctx, x, weight, bias, post_slope=1.0):
        ctx.in_dtypes = (x.dtype, weight.dtype, bias.dtype)
        x, weight, bias = x.contiguous().float(), weight.contiguous().float(), bias.contiguous().float()
        B, _, T = x.shape
        M = weight.shape[0]
        out = torch.empty((B, M, T), device=x.device, dtype=torch.float32)
        lib, h = _handle(x.device)
        _capi.check(lib, h, lib.fd_kconv_forward_act(h, x.data_ptr(), weight.data_ptr(), bias.data_ptr(), B, M, T, float(post_slope), out.data_ptr(),
                                                     _stream(x.device)), "fd_kconv_forward")
        ctx.post = float(post_slope)
        if ctx.post != 1.0:
            ctx.save_for_backward(x, weight, out)
        else:
            ctx.save_for_backward(x, weight)
        return out.to(ctx.in_dtypes[0])

    @staticmethod
    def backward(ctx, dout):
        x, weight = ctx.saved_tensors[:2]
        y = ctx.saved_tensors[2] if ctx.post != 1.0 else None
        dout = dout.contiguous().float()
        B, _, T = x.shape
        M = weight.shape[0]
        need_x, need_w, need_b = ctx.needs_input_grad[:3]
        dx = torch.empty_like(x) if need_x else None
        dw = torch.empty_like(weight) if need_w else None
        db = torch.empty((M,), device=x.device, dtype=torch.float32) if need_b else None
        lib, h = _handle(x.device)
        _capi.check(lib, h, lib.fd_kconv_backward_act(h, x.data_ptr(), weight.data_ptr(), None if y is None else y.data_ptr(), dout.data_ptr(), B, M, T,
                                                      ctx.post, 1.0, None if dx is None else dx.data_ptr(), None if dw is None else dw.data_ptr(),
                                                      None if db is None else db.data_ptr(), _stream(x.device)), "fd_kconv_backward")
        tx, tw, tb = ctx.in_dtypes
        return (None if dx is None else dx.to(tx), None if dw is None else dw.to(tw), None if db is None else db.to(tb), None)


class _SkipFan(torch.autograd.Function):
    """A skip tensor's fan-out (FastDiff_model.py:91-98): returns (x[..., ::factor] for the DiffusionDBlock below, whose nearest
    F.interpolate picks exactly those columns (modules.py:128-131), and four aliases of x for the four layers of the LVC block that
    adds it as audio_down).  Backward: the five gradients in one pass (fd_fan_backward) instead of autograd's zero-fill + strided
    scatter + four full-size additions."""

    @staticmethod
    def forward(ctx, x, factor):
        assert x.dtype == torch.float32 and x.dim() == 3 and x.shape[-1] % int(factor) == 0
        x = x.contiguous()
        B, C, L = x.shape
        picked = torch.empty((B, C, L // int(factor)), device=x.device, dtype=torch.float32)
        lib, h = _handle(x.device)
        _capi.check(lib, h, lib.fd_fan_forward(h, x.data_ptr(), B * C, L, int(factor), picked.data_ptr(), _stream(x.device)), "fd_fan_forward")
        ctx.factor, ctx.shape = int(factor), (B, C, L)
        return (picked,) + tuple(x.view_as(x) for _ in range(4))

    @staticmethod
    def backward(ctx, gp, *gs):
        B, C, L = ctx.shape
        ref = next((g for g in (gp,) + gs if g is not None), None)
        if ref is None:
            return None, None
        dx = torch.empty((B, C, L), device=ref.device, dtype=torch.float32)
        ptr = lambda g: None if g is None else g.data_ptr()      # noqa: E731
        gs = [None if g is None else g.contiguous().float() for g in gs]
        gp = None if gp is None else gp.contiguous().float()
        lib, h = _handle(ref.device)
        _capi.check(lib, h, lib.fd_fan_backward(h, ptr(gs[0]), ptr(gs[1]), ptr(gs[2]), ptr(gs[3]), ptr(gp), B * C, L, ctx.factor, dx.data_ptr(),
                                                _stream(ref.device)), "fd_fan_backward")
        return dx, None


def skip_fan_supported(x, factor):
    return x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.shape[-1] % int(factor) == 0 and x.shape[0] * x.shape[1] <= 65535


def skip_fan(x, factor):
    """(x[..., ::factor], x, x, x, x) for a skip tensor x [B, C, L]: the picked columns for the DBlock below and one alias of x per
    layer of the LVC block that adds it; the gradients of all five come back through one kernel."""
    return _SkipFan.apply(x, factor)


class _InputConv(torch.autograd.Function):
    """KernelPredictor.input_conv (modules.py:292-295): leaky_relu(Conv1d(80 -> 64, k5, padding 2), slope), forward and backward on HIP
    kernels (fd_input_conv_forward / fd_input_conv_backward)."""

    @staticmethod
    def forward(ctx, x, weight, bias, post_slope):
        ctx.in_dtypes = (x.dtype, weight.dtype, bias.dtype)
        x, weight, bias = x.contiguous().float(), weight.contiguous().float(), bias.contiguous().float()
        B, _, T = x.shape
        y = torch.empty((B, 64, T), device=x.device, dtype=torch.float32)
        lib, h = _handle(x.device)
        _capi.check(lib, h, lib.fd_input_conv_forward(h, x.data_ptr(), weight.data_ptr(), bias.data_ptr(), B, T, float(post_slope), y.data_ptr(),
                                                      _stream(x.device)), "fd_input_conv_forward")
        ctx.post = float(post_slope)
        ctx.save_for_backward(x, weight, y)
        return y.to(ctx.in_dtypes[0])

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        dy = dy.contiguous().float()
        B, _, T = x.shape
        need_x, need_w, need_b = ctx.needs_input_grad[:3]
        dx = torch.empty_like(x) if need_x else None
        dw = torch.empty_like(weight) if need_w else None
        db = torch.empty(64, device=x.device, dtype=torch.float32) if need_b else None
        lib, h = _handle(x.device)
        _capi.check(lib, h, lib.fd_input_conv_backward(h, x.data_ptr(), weight.data_ptr(), y.data_ptr(), dy.data_ptr(), B, T, ctx.post,
                                                       None if dx is None else dx.data_ptr(), None if dw is None else dw.data_ptr(),
                                                       None if db is None else db.data_ptr(), _stream(x.device)), "fd_input_conv_backward")
        tx, tw, tb = ctx.in_dtypes
        return (None if dx is None else dx.to(tx), None if dw is None else dw.to(tw), None if db is None else db.to(tb), None)


def input_conv_supported(x, weight):
    """The predictor's input convolution as the model builds it: weight [64, 80, 5] on a HIP tensor [B, 80, T], T <= 128."""
    return x.is_cuda and x.dim() == 3 and x.shape[1] == 80 and tuple(weight.shape) == (64, 80, 5) and 1 <= x.shape[2] <= 128


def input_conv(x, weight, bias, post_slope=0.1):
    """leaky_relu(conv1d(x [B,80,T], weight [64,80,5], bias, padding=2), post_slope) as a differentiable HIP operator."""
    return _InputConv.apply(x, weight, bias, post_slope)


class _KConvStack(torch.autograd.Function):
    """A chain of `Conv1d(64, 64, 3, padding 1), LeakyReLU(slope)` pairs -- the predictor's residual stack (modules.py:297-314) -- as ONE
    autograd node: n forward launches with the activation in the store; in the backward the gradient that a pair hands down is
    multiplied by the mask of the pair below as it is written (fd_kconv_backward_act: in_slope), so only the top pair's two
    latency-bound launches read an activation mask.  Inputs: x [B, 64, T], then weight_1, bias_1, ..., weight_n, bias_n."""

    @staticmethod
    def forward(ctx, x, slope, *params):
        n = len(params) // 2
        ctx.in_dtype = x.dtype
        ctx.slope, ctx.n = float(slope), n
        hs = [x.contiguous().float()]
        ws = [params[2 * j].contiguous().float() for j in range(n)]
        B, _, T = hs[0].shape
        lib, h = _handle(x.device)
        for j in range(n):
            out = torch.empty((B, 64, T), device=x.device, dtype=torch.float32)
            _capi.check(lib, h, lib.fd_kconv_forward_act(h, hs[-1].data_ptr(), ws[j].data_ptr(), params[2 * j + 1].contiguous().float().data_ptr(), B, 64, T,
                                                         ctx.slope, out.data_ptr(), _stream(x.device)), "fd_kconv_forward")
            hs.append(out)
        ctx.save_for_backward(*hs, *ws)
        return hs[-1].to(ctx.in_dtype)

    @staticmethod
    def backward(ctx, dout):
        n = ctx.n
        hs, ws = ctx.saved_tensors[:n + 1], ctx.saved_tensors[n + 1:]
        B, _, T = hs[0].shape
        g = dout.contiguous().float()
        lib, h = _handle(g.device)
        st = _stream(g.device)
        gs = [None] * n
        # the dx chain first: pair j's gradient in front of its activation (the top pair masks dout with its own output, every other
        # pair receives it masked from the pair above) ...
        for j in range(n - 1, -1, -1):
            top, bottom = j == n - 1, j == 0
            gs[j] = g
            if bottom and not ctx.needs_input_grad[0]:
                g = None
                break
            dx = torch.empty_like(hs[j])
            _capi.check(lib, h, lib.fd_kconv_backward_act(h, hs[j].data_ptr(), ws[j].data_ptr(), hs[j + 1].data_ptr() if top else None, g.data_ptr(), B, 64, T,
                                                          ctx.slope if top else 1.0, 1.0 if bottom else ctx.slope, dx.data_ptr(), None, None, st),
                        "fd_kconv_backward")
            g = dx
        # ... then the weight and bias gradients of all pairs in two launches
        want = [j for j in range(n) if ctx.needs_input_grad[2 + 2 * j] or ctx.needs_input_grad[3 + 2 * j]]
        grads = [None] * (2 * n)
        for c0 in range(0, len(want), 8):
            js = want[c0:c0 + 8]
            dws = [torch.empty_like(ws[j]) if ctx.needs_input_grad[2 + 2 * j] else None for j in js]
            dbs = [torch.empty(64, device=gs[0].device, dtype=torch.float32) if ctx.needs_input_grad[3 + 2 * j] else None for j in js]
            arr = lambda ts: (ct.c_void_p * len(js))(*[None if t is None else t.data_ptr() for t in ts])      # noqa: E731
            _capi.check(lib, h, lib.fd_kconv_backward_w_multi(h, len(js), arr([hs[j] for j in js]), arr([gs[j] for j in js]),
                                                              arr([hs[j + 1] if j == n - 1 else None for j in js]), B, 64, T, ctx.slope,
                                                              arr(dws), arr(dbs), st), "fd_kconv_backward_w_multi")
            for j, dw, db in zip(js, dws, dbs):
                grads[2 * j], grads[2 * j + 1] = dw, db
        return (None if g is None else g.to(ctx.in_dtype), None) + tuple(grads)


def _ptrs(ts):
    """A ctypes array of the tensors' device pointers (None -> NULL): the host-side pointer lists of the *_multi entry points."""
    return (ct.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])


class _PredictorFronts(torch.autograd.Function):
    """The front ends of P KernelPredictors side by side (modules.py:292-314,328-329): for each, c = leaky_relu(input_conv(x)),
    r = residual stack(c) (n pairs of Conv1d(64, 64, 3) + LeakyReLU), output c + r.  The P chains are independent and each launch of one
    is latency-bound (B workgroups); here every step of the chain is ONE launch for all P (fd_*_multi: the pointers of the P
    convolutions as kernel arguments), forward and backward.  Inputs: slope, n, x_1 .. x_P, then per predictor w_in, b_in, w_1, b_1,
    ..., w_n, b_n.  Same kernels as input_conv / kernel_conv_stack: same bits."""

    @staticmethod
    def forward(ctx, slope, n, *args):
        P = len(args) // (2 * n + 3)
        xs = [t.contiguous().float() for t in args[:P]]
        par = [t.contiguous().float() for t in args[P:]]
        per = 2 * n + 2
        win, bin_ = [par[p * per] for p in range(P)], [par[p * per + 1] for p in range(P)]
        ws = [[par[p * per + 2 + 2 * j] for p in range(P)] for j in range(n)]
        bs = [[par[p * per + 3 + 2 * j] for p in range(P)] for j in range(n)]
        B, _, T = xs[0].shape
        dev = xs[0].device
        lib, h = _handle(dev)
        st = _stream(dev)
        H = [torch.empty((P, B, 64, T), device=dev, dtype=torch.float32) for _ in range(n + 1)]      # H[0] = c, H[j] = output of pair j
        _capi.check(lib, h, lib.fd_input_conv_forward_multi(h, P, _ptrs(xs), _ptrs(win), _ptrs(bin_), B, T, float(slope), _ptrs(list(H[0].unbind(0))), st),
                    "fd_input_conv_forward_multi")
        for j in range(n):
            _capi.check(lib, h, lib.fd_kconv_forward_act_multi(h, P, _ptrs(list(H[j].unbind(0))), _ptrs(ws[j]), _ptrs(bs[j]), B, 64, T, float(slope),
                                                               _ptrs(list(H[j + 1].unbind(0))), st), "fd_kconv_forward_act_multi")
        out = H[0] + H[n]
        ctx.save_for_backward(*xs, *win, *[w for wj in ws for w in wj], *H)
        ctx.slope, ctx.n, ctx.P = float(slope), n, P
        return tuple(out.unbind(0))

    @staticmethod
    def backward(ctx, *gout):
        n, P, slope = ctx.n, ctx.P, ctx.slope
        sv = ctx.saved_tensors
        xs, win = sv[:P], sv[P:2 * P]
        ws = [sv[2 * P + j * P:2 * P + (j + 1) * P] for j in range(n)]
        H = sv[2 * P + n * P:]
        B, _, T = xs[0].shape
        dev = xs[0].device
        lib, h = _handle(dev)
        st = _stream(dev)
        zero = None
        g = []
        for t in gout:      # a predictor whose output took no part in the loss: zeros
            if t is None:
                zero = torch.zeros((B, 64, T), device=dev, dtype=torch.float32) if zero is None else zero
                t = zero
            g.append(t.contiguous().float())
        gtop = g
        gs = [None] * n            # gs[j]: the gradients in front of pair j's activation (P tensors); the top pair's is masked in the kernels
        cur = g
        for j in range(n - 1, -1, -1):
            top, bottom = j == n - 1, j == 0
            gs[j] = cur
            DX = torch.empty((P, B, 64, T), device=dev, dtype=torch.float32)
            _capi.check(lib, h, lib.fd_kconv_backward_x_multi(h, P, _ptrs(list(H[j].unbind(0))), _ptrs(ws[j]), _ptrs(list(H[n].unbind(0))) if top else None,
                                                              _ptrs(cur), B, 64, T, slope if top else 1.0, 1.0 if bottom else slope,
                                                              _ptrs(list(DX.unbind(0))), st), "fd_kconv_backward_x_multi")
            cur = list(DX.unbind(0))
        # weight and bias gradients of all P * n pairs, eight per call
        flat = [(p, j) for p in range(P) for j in range(n)]
        dW = {k: torch.empty_like(ws[k[1]][k[0]]) for k in flat}
        dB = {k: torch.empty(64, device=dev, dtype=torch.float32) for k in flat}
        for c0 in range(0, len(flat), 8):
            ks = flat[c0:c0 + 8]
            _capi.check(lib, h, lib.fd_kconv_backward_w_multi(h, len(ks), _ptrs([H[j][p] for p, j in ks]), _ptrs([gs[j][p] for p, j in ks]),
                                                              _ptrs([H[n][p] if j == n - 1 else None for p, j in ks]), B, 64, T, slope,
                                                              _ptrs([dW[k] for k in ks]), _ptrs([dB[k] for k in ks]), st), "fd_kconv_backward_w_multi")
        # c has two readers, the stack and the sum: dc = dx of the bottom pair + the output's gradient
        torch._foreach_add_(cur, gtop)
        need_x = any(ctx.needs_input_grad[2:2 + P])
        DXin = torch.empty((P, B, 80, T), device=dev, dtype=torch.float32) if need_x else None
        dwin = [torch.empty_like(w) for w in win]
        dbin = [torch.empty(64, device=dev, dtype=torch.float32) for _ in range(P)]
        _capi.check(lib, h, lib.fd_input_conv_backward_multi(h, P, _ptrs(xs), _ptrs(win), _ptrs(list(H[0].unbind(0))), _ptrs(cur), B, T, slope,
                                                             None if DXin is None else _ptrs(list(DXin.unbind(0))), _ptrs(dwin), _ptrs(dbin), st),
                    "fd_input_conv_backward_multi")
        grads = [None, None] + ([None] * P if DXin is None else list(DXin.unbind(0)))
        for p in range(P):
            grads += [dwin[p], dbin[p]]
            for j in range(n):
                grads += [dW[(p, j)], dB[(p, j)]]
        return tuple(grads)


def predictor_fronts(xs, input_convs, stacks, slope):
    """[leaky_relu(input_conv_p(x_p)) + stack_p(.) for p] for P predictors at once: xs = P tensors [B, 80, T]; input_convs = P pairs
    (weight [64, 80, 5], bias); stacks = P lists of n pairs (weight [64, 64, 3], bias); one launch per chain step for all P."""
    n = len(stacks[0])
    flat = []
    for (w, b), st in zip(input_convs, stacks):
        flat += [w, b]
        for wj, bj in st:
            flat += [wj, bj]
    return list(_PredictorFronts.apply(slope, n, *xs, *flat))


class _KConvSide(torch.autograd.Function):
    """P independent small convolutions of one shape -- conv1d(x_p [B,64,T], w_p [M,64,3], b_p, padding 1), M <= 512: the three
    predictors' bias_conv -- with one launch per kernel for all P (fd_kconv_forward_act_multi, fd_kconv_backward_x_multi,
    fd_kconv_backward_w_multi).  Inputs: x_1 .. x_P, w_1, b_1, ..., w_P, b_P."""

    @staticmethod
    def forward(ctx, *args):
        P = len(args) // 3
        xs = [t.contiguous().float() for t in args[:P]]
        ws = [t.contiguous().float() for t in args[P::2]]
        bs = [t.contiguous().float() for t in args[P + 1::2]]
        B, _, T = xs[0].shape
        M = ws[0].shape[0]
        dev = xs[0].device
        out = torch.empty((P, B, M, T), device=dev, dtype=torch.float32)
        lib, h = _handle(dev)
        _capi.check(lib, h, lib.fd_kconv_forward_act_multi(h, P, _ptrs(xs), _ptrs(ws), _ptrs(bs), B, M, T, 1.0, _ptrs(list(out.unbind(0))), _stream(dev)),
                    "fd_kconv_forward_act_multi")
        ctx.save_for_backward(*xs, *ws)
        ctx.P = P
        return tuple(out.unbind(0))

    @staticmethod
    def backward(ctx, *gout):
        P = ctx.P
        xs, ws = ctx.saved_tensors[:P], ctx.saved_tensors[P:]
        B, _, T = xs[0].shape
        M = ws[0].shape[0]
        dev = xs[0].device
        zero = None
        g = []
        for t in gout:
            if t is None:
                zero = torch.zeros((B, M, T), device=dev, dtype=torch.float32) if zero is None else zero
                t = zero
            g.append(t.contiguous().float())
        lib, h = _handle(dev)
        st = _stream(dev)
        DX = torch.empty((P, B, 64, T), device=dev, dtype=torch.float32) if any(ctx.needs_input_grad[:P]) else None
        if DX is not None:
            _capi.check(lib, h, lib.fd_kconv_backward_x_multi(h, P, None, _ptrs(ws), None, _ptrs(g), B, M, T, 1.0, 1.0, _ptrs(list(DX.unbind(0))), st),
                        "fd_kconv_backward_x_multi")
        dws = [torch.empty_like(w) for w in ws]
        dbs = [torch.empty(M, device=dev, dtype=torch.float32) for _ in range(P)]
        _capi.check(lib, h, lib.fd_kconv_backward_w_multi(h, P, _ptrs(xs), _ptrs(g), None, B, M, T, 1.0, _ptrs(dws), _ptrs(dbs), st), "fd_kconv_backward_w_multi")
        grads = [None] * P if DX is None else list(DX.unbind(0))
        for dw, db in zip(dws, dbs):
            grads += [dw, db]
        return tuple(grads)


def kernel_conv1d_side_by_side(xs, weights, biases):
    """[conv1d(x_p, w_p, b_p, padding=1) for p] for P <= 8 inputs [B, 64, T] and weights [M, 64, 3] of one shape (M <= 512), one launch
    per kernel for all P."""
    flat = []
    for w, b in zip(weights, biases):
        flat += [w, b]
    return list(_KConvSide.apply(*xs, *flat))


def kernel_conv_stack(x, weights, biases, slope):
    """leaky_relu(conv1d(., w_j, b_j, padding=1), slope) applied n times in a row to x [B, 64, T] (every w_j [64, 64, 3]) as one
    differentiable HIP operator: the predictor's residual stack without its Dropout(p = 0) modules."""
    params = []
    for w, b in zip(weights, biases):
        params += [w, b]
    return _KConvStack.apply(x, slope, *params)


class _Conv7(torch.autograd.Function):
    """first_audio_conv (which = 0: Conv1d(1, 32, 7, padding 3)) / final_conv (which = 1: Conv1d(32, 1, 7, padding 3)), FastDiff_model.py:
    34-36,67-68, forward and backward on HIP kernels (fd_conv7_forward / fd_conv7_backward)."""

    @staticmethod
    def forward(ctx, x, weight, bias, which):
        ctx.in_dtypes = (x.dtype, weight.dtype, bias.dtype)
        x, weight, bias = x.contiguous().float(), weight.contiguous().float(), bias.contiguous().float()
        B, _, L = x.shape
        y = torch.empty((B, 32 if which == 0 else 1, L), device=x.device, dtype=torch.float32)
        lib, h = _handle(x.device)
        _capi.check(lib, h, lib.fd_conv7_forward(h, int(which), x.data_ptr(), weight.data_ptr(), bias.data_ptr(), B, L, y.data_ptr(), _stream(x.device)),
                    "fd_conv7_forward")
        ctx.save_for_backward(x, weight)
        ctx.which = int(which)
        return y.to(ctx.in_dtypes[0])

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous().float()
        B, _, L = x.shape
        need_x, need_w, need_b = ctx.needs_input_grad[:3]
        dx = torch.empty_like(x) if need_x else None
        dw = torch.empty_like(weight) if need_w else None
        db = torch.empty(32 if ctx.which == 0 else 1, device=x.device, dtype=torch.float32) if need_b else None
        lib, h = _handle(x.device)
        _capi.check(lib, h, lib.fd_conv7_backward(h, ctx.which, x.data_ptr(), weight.data_ptr(), dy.data_ptr(), B, L, None if dx is None else dx.data_ptr(),
                                                  None if dw is None else dw.data_ptr(), None if db is None else db.data_ptr(), _stream(x.device)),
                    "fd_conv7_backward")
        tx, tw, tb = ctx.in_dtypes
        return (None if dx is None else dx.to(tx), None if dw is None else dw.to(tw), None if db is None else db.to(tb), None)


def conv7_supported(x, weight):
    """first_audio_conv ([32,1,7] on [B,1,L]) or final_conv ([1,32,7] on [B,32,L]) on a HIP tensor whose length is a multiple of 4."""
    return (x.is_cuda and x.dim() == 3 and x.shape[2] % 4 == 0 and 4 <= x.shape[2] < (1 << 25) and
            ((tuple(weight.shape) == (32, 1, 7) and x.shape[1] == 1) or (tuple(weight.shape) == (1, 32, 7) and x.shape[1] == 32)))


def conv7(x, weight, bias):
    """conv1d(x, weight, bias, padding=3) for the model's first_audio_conv / final_conv as a differentiable HIP operator."""
    return _Conv7.apply(x, weight, bias, 0 if weight.shape[0] == 32 else 1)


class _Upsample(torch.autograd.Function):
    """`upsample(F.leaky_relu(x, 0.2))` of an LVC block (modules.py:163-166,205-206): activation + ConvTranspose1d(32, 32, 2 r, stride r,
    padding r / 2) forward and backward on HIP kernels (fd_upsample_forward / fd_upsample_backward)."""

    @staticmethod
    def forward(ctx, x, weight, bias, ratio):
        ctx.in_dtypes = (x.dtype, weight.dtype, bias.dtype)
        x, weight, bias = x.contiguous().float(), weight.contiguous().float(), bias.contiguous().float()
        B, _, Lin = x.shape
        y = torch.empty((B, 32, Lin * int(ratio)), device=x.device, dtype=torch.float32)
        lib, h = _handle(x.device)
        _capi.check(lib, h, lib.fd_upsample_forward(h, x.data_ptr(), weight.data_ptr(), bias.data_ptr(), B, Lin, int(ratio), y.data_ptr(), _stream(x.device)),
                    "fd_upsample_forward")
        ctx.save_for_backward(x, weight)
        ctx.ratio = int(ratio)
        return y.to(ctx.in_dtypes[0])

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous().float()
        B, _, Lin = x.shape
        need_x, need_w, need_b = ctx.needs_input_grad[:3]
        dx = torch.empty_like(x) if need_x else None
        dw = torch.empty_like(weight) if need_w else None
        db = torch.empty(32, device=x.device, dtype=torch.float32) if need_b else None
        lib, h = _handle(x.device)
        _capi.check(lib, h, lib.fd_upsample_backward(h, x.data_ptr(), weight.data_ptr(), dy.data_ptr(), B, Lin, ctx.ratio, None if dx is None else dx.data_ptr(),
                                                     None if dw is None else dw.data_ptr(), None if db is None else db.data_ptr(), _stream(x.device)),
                    "fd_upsample_backward")
        tx, tw, tb = ctx.in_dtypes
        return (None if dx is None else dx.to(tx), None if dw is None else dw.to(tw), None if db is None else db.to(tb), None)


def upsample_supported(x, m):
    """An LVC block's up-sampler as the model builds it: ConvTranspose1d(32, 32, 2 r, stride r, padding r / 2), r = 4 or 8, on a HIP tensor."""
    r = m.stride[0]
    return (x.is_cuda and x.dim() == 3 and x.shape[1] == 32 and isinstance(m, torch.nn.ConvTranspose1d) and r in (4, 8) and
            tuple(m.weight.shape) == (32, 32, 2 * r) and m.padding == (r // 2,) and m.output_padding == (0,) and m.dilation == (1,))


def upsample(x, weight, bias, ratio):
    """conv_transpose1d(leaky_relu(x, 0.2), weight, bias, stride=ratio, padding=ratio // 2) for weight [32, 32, 2 * ratio] as a
    differentiable HIP operator."""
    return _Upsample.apply(x, weight, bias, ratio)


class _WeightNorm(torch.autograd.Function):
    """w = torch._weight_norm(v, g, 0) (what torch.nn.utils.weight_norm's hook evaluates on every forward, FastDiff_model.py:115-122)
    as one HIP launch forward (fd_weight_norm_forward) and one backward (fd_weight_norm_backward)."""

    @staticmethod
    def forward(ctx, v, g):
        v, g = v.contiguous(), g.contiguous()
        rows, cols = v.shape[0], v.numel() // v.shape[0]
        w = torch.empty_like(v)
        norm = torch.empty(rows, device=v.device, dtype=torch.float32)
        lib, h = _handle(v.device)
        _capi.check(lib, h, lib.fd_weight_norm_forward(h, v.data_ptr(), g.data_ptr(), rows, cols, w.data_ptr(), norm.data_ptr(), _stream(v.device)),
                    "fd_weight_norm_forward")
        ctx.save_for_backward(v, g, norm)
        return w

    @staticmethod
    def backward(ctx, dw):
        v, g, norm = ctx.saved_tensors
        dw = dw.contiguous().float()
        rows, cols = v.shape[0], v.numel() // v.shape[0]
        dv, dg = torch.empty_like(v), torch.empty_like(g)
        lib, h = _handle(v.device)
        _capi.check(lib, h, lib.fd_weight_norm_backward(h, v.data_ptr(), g.data_ptr(), norm.data_ptr(), dw.data_ptr(), rows, cols, dv.data_ptr(),
                                                        dg.data_ptr(), _stream(v.device)), "fd_weight_norm_backward")
        return dv, dg


def weight_norm(v, g):
    """torch._weight_norm(v, g, 0) for float32 HIP tensors (g of shape [out, 1, ...]) as a differentiable HIP operator; anything else
    goes to torch."""
    if v.is_cuda and v.dtype == torch.float32 and g.dtype == torch.float32 and g.numel() == v.shape[0]:
        return _WeightNorm.apply(v, g)
    return torch._weight_norm(v, g, 0)


class _WeightNormAll(torch.autograd.Function):
    """w_i = torch._weight_norm(v_i, g_i, 0) for ALL weight-normed convolutions of the module in one or two launches each way
    (fd_weight_norm_multi_forward / _backward; inputs v_1, g_1, v_2, g_2, ...).  The w_i are views of one flat buffer, and so are the
    gradients dv_i / dg_i (which become the parameters' .grad).  The host work per call is a few vectorised numpy statements and two
    torch.split calls: in eager mode this operator sits at the very end of the backward, where nothing hides it."""

    @staticmethod
    def forward(ctx, *vg):
        import numpy as np
        n = len(vg) // 2
        vs, gs = [t.contiguous() for t in vg[0::2]], [t.contiguous() for t in vg[1::2]]
        dev = vs[0].device
        rows = np.array([v.shape[0] for v in vs], dtype=np.int64)
        numel = np.array([v.numel() for v in vs], dtype=np.int64)
        ow, on = np.concatenate(([0], np.cumsum(numel)[:-1])), np.concatenate(([0], np.cumsum(rows)[:-1]))
        W = torch.empty(int(numel.sum()), device=dev, dtype=torch.float32)
        N = torch.empty(int(rows.sum()), device=dev, dtype=torch.float32)
        tab = np.zeros((n, 9), dtype=np.int64)      # fd_wn_item: v, g, w, norm, dw, dv, dg, rows, cols (+ reserved)
        tab[:, 0] = [v.data_ptr() for v in vs]
        tab[:, 1] = [g.data_ptr() for g in gs]
        tab[:, 2] = W.data_ptr() + 4 * ow
        tab[:, 3] = N.data_ptr() + 4 * on
        tab[:, 7] = rows
        tab[:, 8] = numel // rows
        lib, h = _handle(dev)
        _capi.check(lib, h, lib.fd_weight_norm_multi_forward(h, tab.ctypes.data, n, _stream(dev)), "fd_weight_norm_multi_forward")
        ctx.save_for_backward(N, *vs, *gs)
        ctx.tab, ctx.ow, ctx.on, ctx.sizes = tab, ow, on, (numel.tolist(), rows.tolist())
        return tuple(w.view(v.shape) for w, v in zip(W.split(ctx.sizes[0]), vs))

    @staticmethod
    def backward(ctx, *dws):
        n = len(dws)
        N, vs, gs = ctx.saved_tensors[0], ctx.saved_tensors[1:1 + n], ctx.saved_tensors[1 + n:]
        dev = N.device
        DV = torch.empty(sum(ctx.sizes[0]), device=dev, dtype=torch.float32)
        DG = torch.empty(sum(ctx.sizes[1]), device=dev, dtype=torch.float32)
        dws = [None if d is None else d.contiguous().float() for d in dws]      # (kept alive until the launch is enqueued)
        tab = ctx.tab.copy()
        tab[:, 4] = [0 if d is None else d.data_ptr() for d in dws]
        tab[:, 5] = DV.data_ptr() + 4 * ctx.ow
        tab[:, 6] = DG.data_ptr() + 4 * ctx.on
        lib, h = _handle(dev)
        _capi.check(lib, h, lib.fd_weight_norm_multi_backward(h, tab.ctypes.data, n, _stream(dev)), "fd_weight_norm_multi_backward")
        out = []
        for dv, dg, v, g in zip(DV.split(ctx.sizes[0]), DG.split(ctx.sizes[1]), vs, gs):
            out.append(dv.view(v.shape))
            out.append(dg.view(g.shape))
        return tuple(out)


def weight_norm_all(pairs):
    """[torch._weight_norm(v, g, 0) for (v, g) in pairs] for float32 HIP tensors, all in one differentiable operator (one or two
    launches each way whatever the number of tensors)."""
    flat = []
    for v, g in pairs:
        flat += [v, g]
    return list(_WeightNormAll.apply(*flat))


class _Conv32(torch.autograd.Function):
    """xs = x (+ skip); y = post(bias + conv1d(pre(xs), weight, dilation, padding = dilation)) -- one of the denoiser's 21 small
    convolutions with everything the reference wraps around it (modules.py:136-137 and 209-212), forward in one HIP pass
    (fd_conv32_forward) and backward in one pass plus a fixed-order reduction of the weight / bias gradients (fd_conv32_backward)."""

    @staticmethod
    def forward(ctx, x, skip, weight, bias, dilation, pre_slope, post_slope):
        ctx.in_dtypes = (x.dtype, None if skip is None else skip.dtype, weight.dtype, bias.dtype)
        x, weight, bias = x.contiguous().float(), weight.contiguous().float(), bias.contiguous().float()
        B, _, L = x.shape
        y = torch.empty_like(x)
        xs = x
        if skip is not None:
            skip = skip.contiguous().float()
            xs = torch.empty_like(x)
        lib, h = _handle(x.device)
        _capi.check(lib, h, lib.fd_conv32_forward(h, x.data_ptr(), None if skip is None else skip.data_ptr(), weight.data_ptr(), bias.data_ptr(), B, L,
                                                  int(dilation), float(pre_slope), float(post_slope), None if skip is None else xs.data_ptr(),
                                                  y.data_ptr(), _stream(x.device)), "fd_conv32_forward")
        ctx.save_for_backward(xs, y, weight)
        ctx.cfg = (int(dilation), float(pre_slope), float(post_slope), skip is not None)
        ctx.set_materialize_grads(False)                 # an output nobody differentiates arrives as None, not as a tensor of zeros
        if skip is None:
            return y.to(ctx.in_dtypes[0])
        return xs.to(ctx.in_dtypes[0]), y.to(ctx.in_dtypes[0])

    @staticmethod
    def backward(ctx, *grads):
        xs, y, weight = ctx.saved_tensors
        dilation, pre, post, has_skip = ctx.cfg
        gxs, gy = (grads if has_skip else (None, grads[0]))
        B, _, L = xs.shape
        need_x = ctx.needs_input_grad[0] or (has_skip and ctx.needs_input_grad[1])
        need_w, need_b = ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        lib, h = _handle(xs.device)
        if gy is None:                                   # y had no reader: only the pass-through of xs is left
            d = gxs
            dw = torch.zeros_like(weight) if need_w else None
            db = torch.zeros(32, device=xs.device) if need_b else None
        else:
            gy = gy.contiguous().float()
            gxs = None if gxs is None else gxs.contiguous().float()
            d = torch.empty_like(xs) if need_x else None
            dw = torch.empty_like(weight) if need_w else None
            db = torch.empty(32, device=xs.device, dtype=torch.float32) if need_b else None
            _capi.check(lib, h, lib.fd_conv32_backward(h, xs.data_ptr(), y.data_ptr(), weight.data_ptr(), gy.data_ptr(),
                                                       None if gxs is None else gxs.data_ptr(), B, L, dilation, pre, post,
                                                       None if d is None else d.data_ptr(), None if dw is None else dw.data_ptr(),
                                                       None if db is None else db.data_ptr(), _stream(xs.device)), "fd_conv32_backward")
        tx, ts, tw, tb = ctx.in_dtypes
        dx = d.to(tx) if (d is not None and ctx.needs_input_grad[0]) else None
        ds = d.to(ts) if (d is not None and has_skip and ctx.needs_input_grad[1]) else None
        return dx, ds, (None if dw is None else dw.to(tw)), (None if db is None else db.to(tb)), None, None, None


def conv32_supported(x, weight, dilation):
    """The shapes the HIP operator covers: 32 -> 32 channels, kernel 3, dilation 1 / 2 / 3 / 4 / 9 / 27 (every small convolution of
    the model), a length that is a multiple of 4."""
    return (x.is_cuda and x.dim() == 3 and x.shape[1] == 32 and tuple(weight.shape) == (32, 32, 3) and int(dilation) in (1, 2, 3, 4, 9, 27)
            and x.shape[2] % 4 == 0 and 4 <= x.shape[2] < (1 << 25))


def conv32(x, weight, bias, dilation, skip=None, pre_slope=0.2, post_slope=1.0):
    """One small convolution of the denoiser as the reference applies it, as a differentiable HIP operator:
        xs = x + skip (when a skip is given);   y = leaky_relu_post(bias + conv1d(leaky_relu_pre(xs), weight, dilation=d, padding=d))
    with slope 1.0 = no activation.  DiffusionDBlock: conv32(x, w, b, d) (modules.py:136-137); TimeAware_LVCBlock layer i:
    xs, y = conv32(x, w, b, 3 ** i, skip=audio_down, post_slope=0.2) (modules.py:209-212).  Returns y, or (xs, y) with a skip.
    weight is the folded weight (torch._weight_norm of the module's weight_v / weight_g under weight-norm)."""
    return _Conv32.apply(x, skip, weight, bias, dilation, pre_slope, post_slope)


def kernel_conv_supported(x, weight):
    """The shapes the HIP kernels cover: 64 input channels, kernel 3, a multiple of 32 output channels (kernel_conv 24576, bias_conv
    256, the predictor's residual convolutions 64), at most 128 frames (the reference trains on crops of 100: base.yaml:50-51)."""
    return (x.is_cuda and x.dim() == 3 and weight.dim() == 3 and x.shape[1] == 64 and tuple(weight.shape[1:]) == (64, 3)
            and weight.shape[0] % 32 == 0 and 1 <= x.shape[2] <= 128)


def kernel_conv1d(x, weight, bias, post_slope=1.0):
    """leaky_relu(conv1d(x [B,64,T], weight [M,64,3], bias [M], padding=1), post_slope) -> [B,M,T] as a differentiable HIP operator
    (the predictor's kernel_conv, and with post_slope = 0.1 and M = 64 its residual convolutions with their activation); shapes outside
    kernel_conv_supported(), or an activation on M > 512, are refused by the library (FD_ERR_UNSUPPORTED)."""
    return _KConv.apply(x, weight, bias, post_slope)


def lvc_operator_supported(in_channels, out_channels, kernel_size):
    """The shapes fd_lvc_forward / fd_lvc_backward have kernels for (fd_api.cpp: check_lvc_op): the coefficient block of a frame fits
    the operator's staging (Cin * Cout * ks <= 8192, Cout <= 256) -- with ks = 3 that is inner_channels <= 36."""
    return in_channels * out_channels * kernel_size <= 8192 and out_channels <= 256 and kernel_size % 2 == 1


def _lvc_torch(x, kernel, bias, hop_size):
    """out[b, o, l * hop + s] = bias[b, o, l] + sum_{i, k} xpad[b, i, l * hop + s + k] * kernel[b, i, o, k, l] (modules.py:220-253 with
    dilation 1) on torch ops, differentiable by autograd: for the configurations the constructor accepts but the HIP operator has no
    kernel for (a frame's coefficient block beyond its staging: inner_channels > 36) -- a correctness path, like fd_generic.hip's."""
    B, Cin, L = x.shape
    _, _, Cout, ks, T = kernel.shape
    pad = (ks - 1) // 2
    win = torch.nn.functional.pad(x, (pad, pad)).unfold(2, hop_size + 2 * pad, hop_size).unfold(3, ks, 1)      # [B, Cin, T, hop, ks]
    out = torch.einsum("bilsk,biokl->bols", win, kernel) + bias.unsqueeze(-1)
    return out.reshape(B, Cout, L)


def location_variable_convolution(x, kernel, bias, dilation=1, hop_size=256, grad_slot=None):
    """(batch, in_channels, in_length), (batch, in_channels, out_channels, kernel_size, kernel_length), (batch, out_channels,
    kernel_length) -> (batch, out_channels, in_length); same assert as the reference (modules.py:236).  dilation must be 1: it is
    the only value the model ever passes (modules.py:216).  grad_slot (train.py only): see split_layers.
    Shapes outside the HIP operator's limits (lvc_operator_supported) run on torch ops (_lvc_torch) -- still on the device, still
    differentiable; a shared gradient slot is then not used (autograd owns the gradient of that layer's slice)."""
    batch, in_channels, in_length = x.shape
    batch, in_channels, out_channels, kernel_size, kernel_length = kernel.shape
    assert in_length == (kernel_length * hop_size), "length of (x, kernel) is not matched"
    if dilation != 1:
        raise NotImplementedError("location_variable_convolution: the HIP operator implements dilation = 1 (modules.py:216)")
    if not lvc_operator_supported(in_channels, out_channels, kernel_size):
        if not x.is_cuda:
            raise RuntimeError("fastdiff_amd.location_variable_convolution runs only on a HIP device (no CPU fallback)")
        return _lvc_torch(x, kernel, bias, int(hop_size))
    return _LVC.apply(x, kernel, bias, hop_size, grad_slot)


# ---- "frames": kernel_conv and the operator joined through frame-major tensors (include/fastdiff_hip.h: fd_kconv_*_frames, fd_lvc_*_frames) ----

FRAME = 32 * 64 * 3      # coefficients per frame of the model's operator


def frame_order(order):
    """LongTensor [6144]: position e of a frame -> row (i * 64 + o) * 3 + k of the reference's [32, 64, 3] coefficient block, for
    order = "forward" (the frames kernel_conv writes: the operator's forward operand order) or "grad" (the frames of the gradient: its
    dK accumulator order).  The Python statement of csrc/fd_frame_order.h: row_of, for tests and inspection."""
    e = torch.arange(FRAME)
    j, lane, grp = e & 3, (e >> 2) & 63, e >> 8
    l31, hi = lane & 31, lane >> 5
    if order == "forward":
        mt, sq = grp // 12, grp % 12
        kidx = 2 * (4 * sq + j) + hi
        o, k, i = 32 * mt + l31, kidx >> 5, kidx & 31
    elif order == "grad":
        g, t6 = grp & 3, grp >> 2
        mt = t6 // 3
        k, o, i = t6 - 3 * mt, 32 * mt + 8 * g + 4 * hi + j, l31
    else:
        raise ValueError("frame_order: order must be 'forward' or 'grad'")
    return (i * 64 + o) * 3 + k


def frames_to_reference(frames, order="forward"):
    """frames [B, layers, T, 6144] -> the reference's kernels [B, layers, 32, 64, 3, T] (modules.py:333-338)."""
    B, nl, T, _ = frames.shape
    out = torch.empty((B, nl, FRAME, T), device=frames.device, dtype=frames.dtype)
    out[:, :, frame_order(order).to(frames.device)] = frames.transpose(2, 3)
    return out.view(B, nl, 32, 64, 3, T)


def reference_to_frames(kernels, order="forward"):
    """the reference's kernels [B, layers, 32, 64, 3, T] -> frames [B, layers, T, 6144]."""
    B, nl = kernels.shape[:2]
    T = kernels.shape[-1]
    return kernels.reshape(B, nl, FRAME, T)[:, :, frame_order(order).to(kernels.device)].transpose(2, 3).contiguous()


class _KConvFrames(torch.autograd.Function):
    """kernel_conv writing the LVC operator's frames directly (fd_kconv_forward_frames) and reading the gradient as frames
    (fd_kconv_backward_frames): out [B, M / 6144, T, 6144]."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.in_dtypes = (x.dtype, weight.dtype, bias.dtype)
        x, weight, bias = x.contiguous().float(), weight.contiguous().float(), bias.contiguous().float()
        B, _, T = x.shape
        M = weight.shape[0]
        out = torch.empty((B, M // FRAME, T, FRAME), device=x.device, dtype=torch.float32)
        lib, h = _handle(x.device)
        _capi.check(lib, h, lib.fd_kconv_forward_frames(h, x.data_ptr(), weight.data_ptr(), bias.data_ptr(), B, M, T, out.data_ptr(), _stream(x.device)),
                    "fd_kconv_forward_frames")
        ctx.save_for_backward(x, weight)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight = ctx.saved_tensors
        dout = dout.contiguous().float()
        B, _, T = x.shape
        M = weight.shape[0]
        need_x, need_w, need_b = ctx.needs_input_grad
        dx = torch.empty_like(x) if need_x else None
        dw = torch.empty_like(weight) if need_w else None
        db = torch.empty((M,), device=x.device, dtype=torch.float32) if need_b else None
        lib, h = _handle(x.device)
        _capi.check(lib, h, lib.fd_kconv_backward_frames(h, x.data_ptr(), weight.data_ptr(), dout.data_ptr(), B, M, T,
                                                         None if dx is None else dx.data_ptr(), None if dw is None else dw.data_ptr(),
                                                         None if db is None else db.data_ptr(), _stream(x.device)), "fd_kconv_backward_frames")
        tx, tw, tb = ctx.in_dtypes
        return (None if dx is None else dx.to(tx), None if dw is None else dw.to(tw), None if db is None else db.to(tb))


def kernel_conv_frames_supported(x, weight):
    """kernel_conv_supported() and a whole number of the operator's layers in the output channels (kernel_conv: 4 x 6144)."""
    return kernel_conv_supported(x, weight) and weight.shape[0] % FRAME == 0 and x.dtype == weight.dtype == torch.float32


def kernel_conv1d_frames(x, weight, bias):
    """The predictor's kernel_conv with its result as frames [B, layers, T, 6144] (frames_to_reference() gives the reference's tensor)."""
    return _KConvFrames.apply(x, weight, bias)


class _LVCFrames(torch.autograd.Function):
    """The location-variable convolution on one layer's frames: kernel [B, T, 6144] = frames[:, i] of kernel_conv1d_frames and bias
    [B, 64, T] = bias_conv's output [:, i] (both batch-strided, used where they lie); their gradients are written into the layer's
    slices of the shared buffers of split_layers (grad_slot, bias_slot).
    INVARIANT of the frames pair: the gradient this node returns for `kernel` is in the operator's GRADIENT element order, not the
    forward tensor's, and _KConvFrames.backward (fd_kconv_backward_frames) is its only legal consumer -- the frames tensor is an
    internal edge between kernel_conv1d_frames and this node, created and consumed inside train._kernel_predictor / _lvc_block.  Do not
    attach hooks, retain_grad or a second reader to it, and do not ask torch.autograd.grad for it: take
    lvc_op.frames_to_reference(frames) (a real tensor in the reference's layout, with an ordinary gradient) for anything of that kind."""

    @staticmethod
    def forward(ctx, x, kernel, bias, hop_size, grad_slot, bias_slot):
        if not (x.is_cuda and kernel.is_cuda and bias.is_cuda):
            raise RuntimeError("fastdiff_amd.location_variable_convolution_frames runs only on a HIP device (no CPU fallback)")
        ctx.in_dtypes = (x.dtype, bias.dtype)
        ctx.grad_slot, ctx.bias_slot = grad_slot, bias_slot
        B, _, L = x.shape
        T = kernel.shape[1]
        # (the stride of a dimension of size 1 means nothing and torch reports what it likes there)
        assert kernel.dtype == torch.float32 and tuple(kernel.shape) == (B, T, FRAME) and kernel.stride(2) == 1 and \
            (T == 1 or kernel.stride(1) == FRAME) and (B == 1 or kernel.stride(0) >= T * FRAME), "frames [B, T, 6144]"
        assert L == T * int(hop_size), "length of (x, kernel) is not matched"
        ctx.kbs = kernel.stride(0) if B > 1 else T * FRAME
        x = x.contiguous().float()
        # a layer's slice of bias_conv's [B, layers, 64, T] output is read where it lies
        if not (bias.dtype == torch.float32 and tuple(bias.shape) == (B, 64, T) and (T == 1 or bias.stride(2) == 1) and bias.stride(1) == T and
                (B == 1 or bias.stride(0) >= 64 * T)):
            bias = bias.contiguous().float()
        bbs = bias.stride(0) if B > 1 else 64 * T
        out = torch.empty((B, 64, L), device=x.device, dtype=torch.float32)
        lib, h = _handle(x.device)
        _capi.check(lib, h, lib.fd_lvc_forward_frames(h, x.data_ptr(), kernel.data_ptr(), ctx.kbs, bias.data_ptr(), bbs, B, T, int(hop_size),
                                                      out.data_ptr(), _stream(x.device)), "fd_lvc_forward_frames")
        ctx.save_for_backward(x, kernel)
        ctx.hop = int(hop_size)
        return out.to(ctx.in_dtypes[0])

    @staticmethod
    def backward(ctx, dout):
        x, kernel = ctx.saved_tensors
        dout = dout.contiguous().float()
        B, _, L = x.shape
        T = kernel.shape[1]
        need_x, need_k, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        dx = torch.empty_like(x) if need_x else None

        def slot_or_new(slot, shape):      # this layer's slice of a shared gradient buffer (split_layers), or a tensor of its own
            if slot is None:
                return torch.empty(shape, device=x.device, dtype=torch.float32)
            holder, i, full = slot
            if holder.get("buf") is None:
                holder["buf"] = torch.empty(full, device=x.device, dtype=torch.float32)
            return holder["buf"][:, i]

        dk = slot_or_new(ctx.grad_slot, (B, T, FRAME)) if need_k else None
        db = slot_or_new(ctx.bias_slot if ctx.in_dtypes[1] == torch.float32 else None, (B, 64, T)) if need_b else None
        lib, h = _handle(x.device)
        _capi.check(lib, h, lib.fd_lvc_backward_frames(h, x.data_ptr(), kernel.data_ptr(), ctx.kbs, dout.data_ptr(), B, T, ctx.hop,
                                                       None if dx is None else dx.data_ptr(), None if dk is None else dk.data_ptr(),
                                                       0 if dk is None else (dk.stride(0) if B > 1 else T * FRAME),
                                                       None if db is None else db.data_ptr(),
                                                       0 if db is None else (db.stride(0) if B > 1 else 64 * T), _stream(x.device)),
                    "fd_lvc_backward_frames")
        tx, tb = ctx.in_dtypes
        return (None if dx is None else dx.to(tx), dk, None if db is None else db.to(tb), None, None, None)


def location_variable_convolution_frames(x, kernel_frames, bias, hop_size, grad_slot=None, bias_slot=None):
    """location_variable_convolution for x [B, 32, L], one layer's frames [B, T, 6144] (forward order) and bias [B, 64, T]; the gradient
    with respect to the frames comes back in the "grad" order (what kernel_conv1d_frames' backward reads).  grad_slot / bias_slot: the
    slots split_layers hands out with the slices of the frames / of bias_conv's [B, layers, 64, T] output."""
    return _LVCFrames.apply(x, kernel_frames, bias, hop_size, grad_slot, bias_slot)
