"""The denoiser under autograd (fastdiff_amd/train.py) against tests/golden/theta_grad.npz: the reference's
training step -- theta_timestep_loss (util.py:291-325) on the reference module in train() mode, loss.backward() -- executed by
oracle/gen_golden.py in float64 and float32: the loss, d loss / d audio, and norm + 64 elements of every parameter's gradient.

CPU: the autograd graph around the LVC operator (float64, the operator replaced by the unfold + einsum restatement of
oracle/torch_eager.py) reproduces the float64 reference to rounding.  GPU: the product path -- FastDiff.forward in train() mode, the
twelve location-variable convolutions forward and backward on the HIP operator -- inside fastdiff_amd.theta_timestep_loss."""
import numpy as np
import pytest
import torch

from conftest import load_golden


def _module(dtype=torch.float32):
    import fastdiff_amd
    import synth
    m = fastdiff_amd.FastDiff()
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.synth_state_dict(1234).items()}, strict=True)
    return m.to(dtype)


def _compare(model, daudio, g, tol, tag="f64"):
    worst = (0.0, "")
    for name, p in model.named_parameters():
        assert p.grad is not None, name
        flat = p.grad.detach().double().reshape(-1).cpu()
        step = max(1, flat.numel() // 64)
        got, ref = flat[::step][:64].numpy(), g[f"{tag}_sample/{name}"]
        scale = max(float(np.abs(ref).max()), 1e-12)
        err = float(np.abs(got - ref).max()) / scale
        nerr = abs(float(flat.norm()) - float(g[f"{tag}_norm/{name}"])) / max(float(g[f"{tag}_norm/{name}"]), 1e-12)
        worst = max(worst, (max(err, nerr), name))
        assert err <= tol and nerr <= tol, (name, err, nerr)
    da = float(np.abs(daudio.detach().double().cpu().numpy() - g[f"daudio_{tag}"]).max()) / float(np.abs(g[f"daudio_{tag}"]).max())
    assert da <= tol, da
    return worst, da


def test_predictor_front_end_is_recognised_only_in_the_models_shape():
    """train._front_spec decides whether the three predictors' front ends may run side by side (lvc_op.predictor_fronts): the model's own
    KernelPredictor (modules.py:292-314) qualifies -- input convolution 80 -> 64 k5 + LeakyReLU(0.1), six 64 -> 64 k3 convolutions each
    followed by the same activation, Dropout(p = 0) in between --; a live dropout, another slope or another convolution does not (the
    forward then takes the predictors one by one)."""
    from fastdiff_amd import train
    m = _module().train()
    specs = [train._front_spec(b.kernel_predictor) for b in m.lvc_blocks]
    assert len(specs) == 3
    for s, b in zip(specs, m.lvc_blocks):
        ic, convs, slope = s
        assert ic is b.kernel_predictor.input_conv[0] and len(convs) == 6 and slope == pytest.approx(0.1)
        assert all(isinstance(c, torch.nn.Conv1d) and c.in_channels == c.out_channels == 64 for c in convs)
    kp = m.lvc_blocks[1].kernel_predictor
    drop = next(x for x in kp.residual_conv if isinstance(x, torch.nn.Dropout))
    drop.p = 0.1
    assert train._front_spec(kp) is None                      # a live dropout sits between the pairs: not a plain chain
    m.eval()
    assert train._front_spec(kp) is not None                  # ... which eval() turns off again
    m.train()
    drop.p = 0.0
    act = next(x for x in kp.residual_conv if isinstance(x, torch.nn.LeakyReLU))
    act.negative_slope = 0.2
    assert train._front_spec(kp) is None                      # two slopes in one chain
    act.negative_slope = kp.input_conv[1].negative_slope
    assert train._front_spec(kp) is not None


def test_gradient_fixture_lists_every_parameter_of_the_module():
    g = load_golden("theta_grad")
    m = _module()
    assert [n for n, _ in m.named_parameters()] == list(g["names"]) and len(g["names"]) == 175


def test_autograd_graph_around_the_operator_matches_the_reference_backward_in_float64():
    """Everything of fastdiff_amd/train.py except the HIP operator, on the CPU in float64."""
    from fastdiff_amd import train
    from torch_eager import EagerFastDiff
    g = load_golden("theta_grad")
    sched = load_golden("schedule")
    m = _module(torch.float64).train()
    alpha = torch.from_numpy(sched["train_alpha"]).double()
    ts = torch.from_numpy(g["ts"])
    z = torch.from_numpy(g["z"]).double()
    audio = torch.from_numpy(g["audio"]).double().requires_grad_(True)
    a_t = alpha[ts]
    x_t = a_t * audio + (1 - a_t ** 2.).sqrt() * z
    eps = train.differentiable_forward(m, (x_t, torch.from_numpy(g["mel"]).double(), ts.view(-1, 1)),
                                       lvc=lambda y, k, b, dil, hop: EagerFastDiff.lvc(y, k, b, hop))
    loss = torch.nn.functional.mse_loss(eps, z)
    loss.backward()
    assert abs(loss.item() - float(g["loss_f64"])) <= 1e-12 * float(g["loss_f64"])
    worst, da = _compare(m, audio.grad, g, 1e-9)
    print("float64 graph vs float64 reference: worst parameter", worst, "d audio", da)


def test_inference_entry_points_still_refuse_autograd_inputs_on_cpu():
    m = _module()
    x = torch.zeros(1, 1, 512, requires_grad=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m((x, torch.zeros(1, 80, 2), torch.zeros(1, 1)))


@pytest.mark.gpu
@pytest.mark.parametrize("frames", [True, False])
def test_training_step_on_the_hip_operator_matches_the_reference_backward(monkeypatch, frames):
    """frames: kernel_conv hands the LVC operator its frame-major operands directly (the product path); False: through the reference's
    [B, layers, 32, 64, 3, T] tensor -- same kernels plus the transposes, kept for A/B runs (module._train_frames)."""
    import fastdiff_amd
    from fastdiff_amd import sampler
    g = load_golden("theta_grad")
    sched = load_golden("schedule")
    m = _module().cuda().train()
    m._train_frames = frames
    monkeypatch.setattr(sampler, "std_normal", lambda size: torch.from_numpy(g["z"].copy()).view(*size).cuda())
    monkeypatch.setattr(torch, "randint", lambda *a, **k: torch.from_numpy(g["ts"].copy()))
    dh = {"T": 1000, "alpha": torch.from_numpy(sched["train_alpha"]).cuda()}
    audio = torch.from_numpy(g["audio"]).cuda().requires_grad_(True)
    calls = []
    real = fastdiff_amd.lvc_op.location_variable_convolution
    monkeypatch.setattr(fastdiff_amd.lvc_op, "location_variable_convolution", lambda x, k, b, d, h, **kw: (calls.append(h), real(x, k, b, d, h, **kw))[1])
    fcalls = []
    real_f = fastdiff_amd.lvc_op.location_variable_convolution_frames
    monkeypatch.setattr(fastdiff_amd.lvc_op, "location_variable_convolution_frames",
                        lambda x, k, b, h, **kw: (fcalls.append((h, tuple(k.shape[1:]))), real_f(x, k, b, h, **kw))[1])
    convs = []
    real_c = fastdiff_amd.lvc_op.conv32
    monkeypatch.setattr(fastdiff_amd.lvc_op, "conv32", lambda x, w, b, d, **k: (convs.append((x.shape[-1], d, "skip" in k)), real_c(x, w, b, d, **k))[1])
    loss = fastdiff_amd.theta_timestep_loss(m, (torch.from_numpy(g["mel"]).cuda(), audio), dh)
    loss.backward()
    hops = [8] * 4 + [64] * 4 + [256] * 4                                # the twelve LVC calls went through the HIP operator
    T = g["mel"].shape[-1]
    assert (calls, fcalls) == (([], [(h, (T, 6144)) for h in hops]) if frames else (hops, []))
    # ... and the small convolutions through theirs: the DBlocks at 384 and 48 columns (the third one runs on 6 columns, not a multiple
    # of 4: torch), the twelve LVC-block layers with their skip add
    # (each DBlock: its 1 x 1 residual convolution first, as the centre tap of a 3-tap one, then the three dilated ones)
    assert convs == [(384, d, False) for d in (1, 1, 2, 4)] + [(48, d, False) for d in (1, 1, 2, 4)] + \
                    [(L, 3 ** i, True) for L in (48, 384, 1536) for i in range(4)], convs
    gap = abs(float(g["loss_f32"]) - float(g["loss_f64"]))
    print("loss %.9f: |d| vs float64 reference %.2e (the float32 reference: %.2e)" % (loss.item(), abs(loss.item() - float(g["loss_f64"])), gap))
    assert abs(loss.item() - float(g["loss_f64"])) <= 2e-6 * float(g["loss_f64"])
    # the float32 reference itself sits up to 8e-6 (relative to the largest element of a tensor) from the float64 one
    worst, da = _compare(m, audio.grad, g, 1e-4)
    print("HIP training step vs float64 reference: worst parameter", worst, "d audio", da)


@pytest.mark.gpu
def test_eval_mode_keeps_the_inference_kernels_and_train_mode_agrees_with_them():
    m = _module().cuda()
    g = load_golden("theta_grad")
    x, mel = torch.from_numpy(g["audio"]).cuda(), torch.from_numpy(g["mel"]).cuda()
    steps = torch.tensor([[437.0], [12.0]]).cuda()
    m.eval()
    y_inf = m((x, mel, steps))
    assert y_inf.grad_fn is None                                         # no graph: the fused pipeline
    m.train()
    y_tr = m((x, mel, steps))
    assert y_tr.grad_fn is not None
    assert float((y_tr.detach() - y_inf).abs().max()) <= 2e-5            # the forward tolerance of the parity tests
    with torch.no_grad():
        assert m((x, mel, steps)).grad_fn is None


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(upsample_ratios=[4, 4, 4], inner_channels=16), dict(upsample_ratios=[2, 4, 4]), dict(lvc_layers_each_block=3),
                                 dict(inner_channels=64, upsample_ratios=[4, 4, 2])])      # 64 channels: beyond the LVC operator's staging -> torch LVC
def test_training_path_of_a_non_reference_configuration_falls_back_and_matches_torch(cfg):
    """the constructor accepts other `inner_channels` / `upsample_ratios` / `lvc_layers_each_block`; the frames pair and
    the shared gradient slot exist for the model's own operator shape only (Cin 32, Cout 64, ks 3, hop 8 / 64 / 256).  Such a module
    must train through the generic operators -- same loss and gradients as the same graph on torch ops in float64 on the CPU.
    inner_channels = 64: a frame's 64 x 128 x 3 coefficient block is beyond what fd_lvc_forward stages
    (lvc_op.lvc_operator_supported); the location-variable convolution of such a module runs on torch ops on the device."""
    import fastdiff_amd
    from fastdiff_amd import train
    from torch_eager import EagerFastDiff
    torch.manual_seed(7)
    m = fastdiff_amd.FastDiff(**cfg).train()
    hop = int(np.prod(m._cfg["upsample_ratios"]))
    B, T = 2, 6
    x = torch.randn(B, 1, T * hop)
    mel = torch.rand(B, 80, T) * 7.5 - 6.0
    ts = torch.tensor([[437.0], [12.0]])
    z = torch.randn(B, 1, T * hop)
    ref = fastdiff_amd.FastDiff(**cfg)
    ref.load_state_dict(m.state_dict(), strict=True)
    ref = ref.double().train()
    eps_ref = train.differentiable_forward(ref, (x.double(), mel.double(), ts.double()), lvc=lambda y, k, b, dil, h: EagerFastDiff.lvc(y, k, b, h))
    torch.nn.functional.mse_loss(eps_ref, z.double()).backward()
    g = m.cuda()
    eps = g((x.cuda(), mel.cuda(), ts.cuda()))
    assert eps.grad_fn is not None
    torch.nn.functional.mse_loss(eps, z.cuda()).backward()
    assert float((eps.detach().cpu().double() - eps_ref.detach()).abs().max()) <= 2e-5 * max(1.0, float(eps_ref.abs().max()))
    for (n, p), (_, q) in zip(g.named_parameters(), ref.named_parameters()):
        assert p.grad is not None and q.grad is not None, n
        scale = max(float(q.grad.abs().max()), 1e-9)
        assert float((p.grad.cpu().double() - q.grad).abs().max()) <= 2e-4 * scale, (n, cfg)
