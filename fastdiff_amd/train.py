"""The denoiser under autograd: what `FastDiffTask._training_step` (FastDiff.py:44-49) differentiates through
`theta_timestep_loss` (util.py:291-325).

The inference path (fd_forward / fd_sample) is one hand-written pipeline with no saved activations; training needs them, and needs
gradients with respect to 175 parameter tensors.  Here the network is a graph of torch.autograd.Function nodes over the C ABI's training
operators (fastdiff_amd/lvc_op.py), forward and backward on HIP kernels:
  * the location-variable convolution, twelve per forward -- the operator the reference states as unfold + einsum over a
    [B, C, T, hop + 2, 3] view (modules.py:220-253) -- reading the predictor's kernel_conv output as frame-major operands
    (kernel_conv1d_frames -> location_variable_convolution_frames: no transposes between the two), and the gate behind it;
  * the 21 small 32-channel convolutions with their skip add, activations and bias (conv32), first_audio_conv / final_conv (conv7),
    the up-samplers (upsample), the skip tensors' fan-out (skip_fan);
  * the KernelPredictors: their front ends (input convolution + activation, residual stack, c + r) for all three blocks side by side
    in one node (predictor_fronts: a predictor never sees x, so the three latency-bound chains run as one), bias_conv likewise
    (kernel_conv1d_side_by_side), kernel_conv per block;
  * weight-norm of all 53 convolutions in one operator (weight_norm_all).
What stays on torch: the step embedding's five linear layers and swish, three broadcast adds, the loss (4 % of the step's kernel time).
The sub-modules of fastdiff_amd.FastDiff are real nn.Conv1d / nn.Linear holders with the reference's weight_g / weight_v
parametrisation, so the reference's optimizer, checkpointing and DDP wrapper see the module they expect.  FastDiff.forward takes this
path when autograd is recording and the module is in train() mode or an input requires a gradient; everything else stays on the
inference kernels.  module._train_frames / _train_fuse_act / _train_skip_fan / _train_stack / _train_wn_all / _train_fronts = False switch single
pieces back to their predecessor (A/B runs: tools/train_step_probe.py).

`lvc` (tests only): a replacement for the HIP operator with the same signature, so that the structure around it can be pinned on
the reference's gradients on a machine without a GPU; the product never passes it.
"""
import threading

import torch
import torch.nn.functional as F

from .sampler import calc_diffusion_step_embedding


def _swish(x):
    return x * torch.sigmoid(x)


class _ForwardWeights(threading.local):
    """conv module -> its effective weight for the forward being recorded (differentiable_forward: all of them from ONE operator).
    Per thread, and keyed on the module OBJECT (held alive by the key): forwards recorded concurrently by replicas in other threads
    neither see nor clear each other's entries, and a recycled id() can never hand out another module's weight."""

    def __init__(self):
        self.map = {}


_WN = _ForwardWeights()


def _conv_weight(m):
    """The effective weight of a conv module: g * v / ||v|| while weight-norm is attached (what its forward hook computes: on HIP
    tensors one operator for all convolutions of the module, lvc_op.weight_norm_all, else one launch of lvc_op.weight_norm each way per
    convolution), the plain weight after remove_weight_norm()."""
    w = _WN.map.get(m)
    if w is not None:
        return w
    if hasattr(m, "weight_g"):
        if m.weight_v.is_cuda:
            from .lvc_op import weight_norm
            return weight_norm(m.weight_v, m.weight_g)
        return torch._weight_norm(m.weight_v, m.weight_g, 0)
    return m.weight


def _conv(m, x):
    """m(x) for a Conv1d holder: on HIP tensors the weight-norm runs on this repo's operator (the module's own hook would run torch's), and
    the two 7-tap convolutions at the ends of the network on theirs (fastdiff_amd.lvc_op.conv7)."""
    if x.is_cuda and m.kernel_size == (7,):
        from .lvc_op import conv7, conv7_supported
        if conv7_supported(x, m.weight_v if hasattr(m, "weight_v") else m.weight):
            return conv7(x, _conv_weight(m), m.bias)
    if x.is_cuda and hasattr(m, "weight_g"):
        return F.conv1d(x, _conv_weight(m), m.bias, stride=m.stride, padding=m.padding, dilation=m.dilation)
    return m(x)


def _dblock(p, x, cconv=None, picked=False):
    """DiffusionDBlock.forward (modules.py:127-138); F.interpolate(size = L // factor) in its default nearest mode picks every
    factor-th sample.  The reference runs the 1 x 1 residual convolution at the full rate and then picks (modules.py:129-130); a
    1 x 1 convolution commutes with picking columns, so here it runs on the picked columns: 1 / factor of the work, the same values
    and the same gradients (only the picked columns ever receive one).  cconv: the HIP operator for `layer(leaky_relu(x, 0.2))`."""
    if not picked:                   # (picked: the caller's skip_fan has taken the columns already)
        size = x.shape[-1] // p.factor
        x = F.interpolate(x, size=size)
    rd = p.residual_dense
    if cconv is not None and rd.kernel_size == (1,) and rd.in_channels == rd.out_channels == 32 and cconv[1](x, torch.empty(32, 32, 3, device="meta"), 1):
        # the 1 x 1 residual convolution as the centre tap of a 3-tap one (zeros either side) on the same HIP operator, no activations
        residual = cconv[0](x, F.pad(_conv_weight(rd), (1, 1)), rd.bias, 1, pre_slope=1.0, post_slope=1.0)
    else:
        residual = _conv(rd, x)
    for layer in p.conv:
        if cconv is not None and cconv[1](x, layer.weight_v if hasattr(layer, "weight_v") else layer.weight, layer.dilation[0]):
            x = cconv[0](x, _conv_weight(layer), layer.bias, layer.dilation[0])
        else:
            x = _conv(layer, F.leaky_relu(x, 0.2))
    return x + residual


def _front_spec(p):
    """(input conv, [stack convs], slope) if the predictor's front end is the model's: Sequential(Conv1d(80, 64, 5, padding 2), LeakyReLU)
    and a residual Sequential of Conv1d(64, 64, 3, padding 1) + LeakyReLU pairs between Dropout(p = 0) modules, one slope throughout."""
    ic = list(p.input_conv)
    mods = [m for m in p.residual_conv if not (isinstance(m, torch.nn.Dropout) and (m.p == 0 or not m.training))]
    convs, acts = mods[0::2], mods[1::2]
    if not (len(ic) == 2 and isinstance(ic[0], torch.nn.Conv1d) and isinstance(ic[1], torch.nn.LeakyReLU) and ic[0].kernel_size == (5,) and
            ic[0].padding == (2,) and ic[0].in_channels == 80 and ic[0].out_channels == 64 and len(convs) == len(acts) >= 1 and
            all(isinstance(a, torch.nn.LeakyReLU) and a.negative_slope == ic[1].negative_slope for a in acts) and
            all(isinstance(m, torch.nn.Conv1d) and m.kernel_size == (3,) and m.padding == (1,) and m.dilation == (1,) and
                m.in_channels == 64 and m.out_channels == 64 for m in convs)):
        return None
    return ic[0], convs, ic[1].negative_slope


def _kernel_predictor(p, c, layers, cin, cout, ks, kconv=None, split=None, frames=None, fuse_act=True, front=None, bias_out=None, hop=None):
    """KernelPredictor.forward (modules.py:320-343).  kconv: the HIP operator for kernel_conv (64 -> 24576 channels: the largest
    matrix product of the step) where its shapes fit, else the module's own convolution.  frames (the product path): kernel_conv
    writes the LVC operator's frame-major operand order directly -- [B, layers, T, 6144] instead of the reference's
    [B, layers, 32, 64, 3, T] -- and reads the gradient that way (lvc_op: kernel_conv1d_frames); the third return value says so."""
    B, _, T = c.shape

    def fits(m, h):
        return kconv is not None and isinstance(m, torch.nn.Conv1d) and m.padding == (1,) and m.dilation == (1,) and \
            kconv[1](h, m.weight_v if hasattr(m, "weight_v") else m.weight)

    def conv(m, h):      # a 64 -> M, k3 convolution of the predictor: the HIP operator where its shapes fit, else the module itself
        if fits(m, h):
            return kconv[0](h, _conv_weight(m), m.bias)
        return _conv(m, h) if isinstance(m, torch.nn.Conv1d) else m(h)

    def run(seq, h):     # a Sequential of the predictor; "Conv1d, LeakyReLU" pairs run as ONE operator where the convolution fits
        mods, i = list(seq), 0
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if fuse_act and fits(m, h) and isinstance(nxt, torch.nn.LeakyReLU) and m.out_channels <= 512:
                h = kconv[0](h, _conv_weight(m), m.bias, nxt.negative_slope)
                i += 2
            elif fuse_act and kconv is not None and isinstance(m, torch.nn.Conv1d) and isinstance(nxt, torch.nn.LeakyReLU) and \
                    m.padding == (2,) and m.dilation == (1,) and m.stride == (1,) and \
                    kconv[2](h, m.weight_v if hasattr(m, "weight_v") else m.weight):      # input_conv: Conv1d(80, 64, 5), LeakyReLU
                h = kconv[3](h, _conv_weight(m), m.bias, nxt.negative_slope)
                i += 2
            else:
                h = conv(m, h)
                i += 1
        return h

    def stack(seq, h):   # a Sequential of nothing but "Conv1d(64, 64, 3), LeakyReLU(s)" pairs and Dropout(p = 0): ONE autograd node
        mods = [m for m in seq if not (isinstance(m, torch.nn.Dropout) and (m.p == 0 or not m.training))]
        convs, acts = mods[0::2], mods[1::2]
        if not (fuse_act and kconv is not None and len(kconv) > 4 and len(mods) >= 2 and len(convs) == len(acts) and
                all(isinstance(a, torch.nn.LeakyReLU) and a.negative_slope == acts[0].negative_slope for a in acts) and
                all(fits(m, h) and m.out_channels == 64 for m in convs)):
            return None
        return kconv[4](h, [_conv_weight(m) for m in convs], [m.bias for m in convs], acts[0].negative_slope)

    if front is not None:                        # (computed for all predictors at once: lvc_op.predictor_fronts)
        c = front
    else:
        c = run(p.input_conv, c)                 # Conv1d 80 -> 64 k5, LeakyReLU(0.1)
        r = stack(p.residual_conv, c)            # Dropout(p = 0), Conv1d 64 -> 64 k3, LeakyReLU(0.1), Conv1d, LeakyReLU, three times
        if r is None:
            r = run(p.residual_conv, c)
        c = c + r
    kc = p.kernel_conv
    # the frames pair exists for the model's own shape only (fd_lvc_*_frames: Cin 32, Cout 64, ks 3, hop 8 / 64 / 256); any other
    # configuration the constructor accepts takes the reference's kernel tensor through the generic operator below
    if frames is not None and split is not None and (cin, cout, ks) == (32, 64, 3) and hop in (8, 64, 256) and \
            frames[2](c, kc.weight_v if hasattr(kc, "weight_v") else kc.weight):
        kf = frames[0](c, _conv_weight(kc), kc.bias)                     # [B, layers, T, 6144]
        # bias_conv's output likewise: the operator reads a layer's [B, 64, T] slice where it lies and writes its gradient into one buffer
        bo = bias_out if bias_out is not None else conv(p.bias_conv, c)      # (bias_out: computed for all predictors at once)
        return split(kf), split(bo.contiguous().view(B, layers, cout, T)), True
    k = conv(kc, c)
    # the reference slices kernels[:, i] (modules.py:213-214), whose backward builds a zero tensor of all four layers per slice and
    # adds the four up; unbind hands autograd the same views and gets one stack back.  (One kernel_conv call per layer on that
    # layer's weight rows -- contiguous kernels, no stack -- measured slower: 18.3 vs 17.1 ms per step; the slices of the WEIGHT then
    # pay the same zero-fill-and-add in their backward.)
    k6 = k.contiguous().view(B, layers, cin, cout, ks, T)
    # on the product path the slices are used where they lie and their gradients land in one buffer (lvc_op.split_layers)
    bo = bias_out if bias_out is not None else conv(p.bias_conv, c)
    return (split(k6) if split is not None else (k6.unbind(1), None), bo.contiguous().view(B, layers, cout, T).unbind(1), False)


def _torch_gate(x, y):
    C = x.shape[1]
    return x + torch.sigmoid(y[:, :C]) * torch.tanh(y[:, C:])


def _lvc_block(p, x, audio_down, c, emb, cfg, lvc, gate=_torch_gate, kconv=None, cconv=None, split=None, frames=None, fuse_act=True, front=None):
    """TimeAware_LVCBlock.forward (modules.py:189-218); the in-place `x += audio_down` of the reference written out of place.
    front: (cond, the predictor's front-end output for it) when the caller computed the front ends of all blocks at once."""
    C = cfg["inner_channels"]
    cond = front[0] if front is not None else c + p.fc_t(emb).unsqueeze(-1)
    (kernels, slots), bias, as_frames = _kernel_predictor(p.kernel_predictor, cond, cfg["lvc_layers_each_block"], C, 2 * C, cfg["lvc_kernel_size"],
                                                          kconv, split if kconv is not None else None, frames if kconv is not None else None, fuse_act,
                                                          None if front is None else front[1], None if front is None else front[2],
                                                          hop=int(p.cond_hop_length))
    if cconv is not None and x.is_cuda:
        from .lvc_op import upsample, upsample_supported
    if cconv is not None and x.is_cuda and upsample_supported(x, p.upsample):
        x = upsample(x, p.upsample.weight, p.upsample.bias, p.upsample.stride[0])      # leaky_relu + ConvTranspose1d in one HIP pass each way
    else:
        x = p.upsample(F.leaky_relu(x, 0.2))
    # audio_down: the skip tensor, or one alias of it per layer (lvc_op.skip_fan: their gradients are then added up in one pass)
    skip_of = (lambda i: audio_down[i]) if isinstance(audio_down, (list, tuple)) else (lambda i: audio_down)
    for i, conv in enumerate(p.convs):
        if cconv is not None and cconv[1](x, conv.weight_v if hasattr(conv, "weight_v") else conv.weight, conv.dilation[0]):
            # x += audio_down; leaky_relu; conv; leaky_relu (modules.py:209-212) in one HIP pass each way
            x, y = cconv[0](x, _conv_weight(conv), conv.bias, conv.dilation[0], skip=skip_of(i), post_slope=0.2)
        else:
            x = x + skip_of(i)
            y = F.leaky_relu(_conv(conv, F.leaky_relu(x, 0.2)), 0.2)
        if as_frames:
            y = frames[1](y, kernels[i], bias[0][i], p.cond_hop_length, grad_slot=slots[i], bias_slot=bias[1][i])
        else:
            y = lvc(y, kernels[i], bias[i], 1, p.cond_hop_length) if slots is None else lvc(y, kernels[i], bias[i], 1, p.cond_hop_length, grad_slot=slots[i])
        x = gate(x, y)                                   # x + sigmoid(y[:, :C]) * tanh(y[:, C:])  (modules.py:217)
    return x


def differentiable_forward(module, data, lvc=None):
    """eps = net((audio, c, diffusion_steps)) as FastDiff.forward (FastDiff_model.py:74-102), recorded by autograd."""
    gate, kconv, cconv, split, frames = _torch_gate, None, None, None, None
    if lvc is None:                    # the product path: the layer's operators, its convolution and the predictor's kernel_conv on HIP kernels
        from .lvc_op import (location_variable_convolution as lvc, gated_residual as gate, kernel_conv1d, kernel_conv_supported, conv32,
                             conv32_supported, split_layers, kernel_conv1d_frames, location_variable_convolution_frames,
                             kernel_conv_frames_supported, input_conv, input_conv_supported, kernel_conv_stack)
        kconv = (kernel_conv1d, kernel_conv_supported, input_conv_supported, input_conv) + \
            ((kernel_conv_stack,) if getattr(module, "_train_stack", True) else ())      # (False: one node per pair, for A/B runs)
        cconv = (conv32, conv32_supported)
        split = split_layers
        if getattr(module, "_train_frames", True):      # (False: the reference's kernel tensor between the two operators, for A/B runs)
            frames = (kernel_conv1d_frames, location_variable_convolution_frames, kernel_conv_frames_supported)
    audio, c, diffusion_steps = data
    cfg = module._cfg
    _WN.map.clear()
    if cconv is not None and audio.is_cuda and getattr(module, "_train_wn_all", True):      # (False: one operator per convolution, for A/B runs)
        from .lvc_op import weight_norm_all
        n_mod = sum(1 for _ in module.modules())          # (the walk is cached while the module tree keeps its size: a convolution that
        cached = module.__dict__.get("_wn_candidates")    # is added or replaced later is picked up on the next step)
        if cached is None or cached[0] != n_mod or any(a is not b for a, b in zip(cached[2], module.modules())):
            mods_now = list(module.modules())
            cached = module.__dict__["_wn_candidates"] = (n_mod, [m for m in mods_now if isinstance(m, torch.nn.Conv1d)], mods_now)
        cands = cached[1]
        mods = [m for m in cands if hasattr(m, "weight_g") and m.weight_v.is_cuda and
                m.weight_v.dtype == torch.float32 and m.weight_g.dtype == torch.float32 and m.weight_g.numel() == m.weight_v.shape[0]]
        if mods:
            for m, w in zip(mods, weight_norm_all([(m.weight_v, m.weight_g) for m in mods])):
                _WN.map[m] = w
    try:
        return _forward_body(module, audio, c, diffusion_steps, cfg, lvc, gate, kconv, cconv, split, frames)
    finally:
        _WN.map.clear()


def _forward_body(module, audio, c, diffusion_steps, cfg, lvc, gate, kconv, cconv, split, frames):
    if c.dim() == 2:
        c = c.unsqueeze(0)
    emb = calc_diffusion_step_embedding(diffusion_steps.to(audio.dtype).view(audio.shape[0], 1), cfg["diffusion_step_embed_dim_in"])
    emb = _swish(module.fc_t2(_swish(module.fc_t1(emb))))
    x = _conv(module.first_audio_conv, audio)
    skips = []
    fan = None
    if cconv is not None and getattr(module, "_train_skip_fan", True):      # (False: autograd's own fan-out, for A/B runs)
        from .lvc_op import skip_fan, skip_fan_supported
        fan = (skip_fan, skip_fan_supported)
    for down in module.downsample:
        if fan is not None and len(module.lvc_blocks[0].convs) == 4 and fan[1](x, down.factor):
            picked, *aliases = fan[0](x, down.factor)      # the DBlock's nearest pick + one alias of x per LVC layer that adds it
            skips.append(aliases)
            x = _dblock(down, picked, cconv, picked=True)
        else:
            skips.append(x)
            x = _dblock(down, x, cconv)
    fuse_act = getattr(module, "_train_fuse_act", True)       # (False: the predictor's LeakyReLUs as torch nodes, for A/B runs)
    fronts = [None] * len(module.lvc_blocks)
    if kconv is not None and len(kconv) > 4 and fuse_act and c.is_cuda and c.dtype == torch.float32 and c.shape[1] == 80 and \
            1 <= c.shape[2] <= 128 and 2 <= len(module.lvc_blocks) <= 8 and getattr(module, "_train_fronts", True):
        # the KernelPredictors see only the mel and the step embedding, never x: their front ends -- input convolution + residual stack,
        # each a chain of latency-bound launches -- run side by side, one launch per chain step for all blocks (lvc_op.predictor_fronts)
        specs = [_front_spec(b.kernel_predictor) for b in module.lvc_blocks]
        if all(s is not None for s in specs) and len({(len(s[1]), s[2]) for s in specs}) == 1:
            from .lvc_op import predictor_fronts
            conds = [c + b.fc_t(emb).unsqueeze(-1) for b in module.lvc_blocks]
            outs = predictor_fronts(conds, [(_conv_weight(s[0]), s[0].bias) for s in specs],
                                    [[(_conv_weight(m), m.bias) for m in s[1]] for s in specs], specs[0][2])
            fronts = list(zip(conds, outs, [None] * len(outs)))
            # ... and their bias_conv (64 -> 256, k3) likewise
            bcs = [b.kernel_predictor.bias_conv for b in module.lvc_blocks]
            if all(isinstance(m, torch.nn.Conv1d) and m.kernel_size == (3,) and m.padding == (1,) and m.dilation == (1,) and m.in_channels == 64 and
                   m.out_channels == bcs[0].out_channels and m.out_channels % 32 == 0 and m.out_channels <= 512 for m in bcs):
                from .lvc_op import kernel_conv1d_side_by_side
                fronts = list(zip(conds, outs, kernel_conv1d_side_by_side(outs, [_conv_weight(m) for m in bcs], [m.bias for m in bcs])))
    for n, audio_down in enumerate(reversed(skips)):
        x = _lvc_block(module.lvc_blocks[n], x, audio_down, c, emb, cfg, lvc, gate, kconv, cconv, split, frames, fuse_act, fronts[n])
    return _conv(module.final_conv[0], x)
