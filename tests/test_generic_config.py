"""Configurations other than base.yaml's: the reference constructor takes any inner_channels / cond_channels /
upsample_ratios / lvc_layers_each_block / kernel sizes / embedding widths (FastDiff_model.py:13-26; FastDiffTask.build_model fills them
from hparams, FastDiff.py:17-29).  fd_create accepts them too; they run on the runtime-shaped kernels of fastdiff_amd/csrc/fd_generic.hip
(the tuned kernel set is base.yaml-only).  Fixtures: tests/golden/forward_cfg{A,B,C}.npz, produced by EXECUTING the reference for each
configuration (oracle/gen_golden.py gen_forward_cfg); the CPU test of the oracle against them is in tests/test_oracle_golden.py."""
import ctypes as ct
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden, ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import synth   # noqa: E402


def test_create_refuses_only_what_the_reference_itself_rejects():
    """No GPU needed to be refused: the configuration check comes before the device is touched."""
    from fastdiff_amd import _capi
    lib = _capi.load()
    for field, value, why in (("audio_channels", 2, b"first_audio_conv"), ("lvc_kernel_size", 4, b"even lvc_kernel_size"),
                              ("kpnet_conv_size", 2, b"even kpnet_conv_size"), ("diffusion_step_embed_dim_in", 127, b"must be even"),
                              ("n_upsample", 0, b"upsample stages"), ("lvc_layers_each_block", 0, b"LVC layers")):
        cfg = _capi.FdConfig()
        lib.fd_default_config(ct.byref(cfg))
        setattr(cfg, field, value)
        h = ct.c_void_p()
        rc = lib.fd_create(ct.byref(cfg), 0, ct.byref(h))
        assert rc == _capi.FD_ERR_UNSUPPORTED and why in lib.fd_last_error(None), (field, rc, lib.fd_last_error(None))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cfgA", "cfgB", "cfgC"])
def test_other_configurations_forward_and_reverse_loop_against_the_reference(name):
    import fastdiff_amd
    import gpu_common as gc
    g, sch = load_golden("forward_" + name), load_golden("schedule")
    cfg = json.loads(str(g["cfg"]))
    m = fastdiff_amd.FastDiff(**cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.synth_state_dict(int(g["seed"]), cfg).items()}, strict=True)
    m = m.cuda().eval()
    assert m.hop_length == int(g["hop"])
    with torch.no_grad():
        y = m((torch.from_numpy(g["audio"]).cuda(), torch.from_numpy(g["mel"]).cuda(), torch.from_numpy(g["steps"]).cuda()))
    assert y.grad_fn is None and tuple(y.shape) == g["y_f64"].shape
    d_fwd = float(np.abs(y.cpu().numpy().astype(np.float64) - g["y_f64"]).max())
    ref_gap = float(np.abs(g["y_f32"].astype(np.float64) - g["y_f64"]).max())
    N = 4
    rows, _ = gc.table_rows(sch, N)
    with torch.no_grad():
        seq = m.sample(torch.from_numpy(g["mel"]).cuda(), rows, x_T=torch.from_numpy(g["x_T"]).cuda(),
                       noise=torch.from_numpy(gc.exec_order_noise(g["z"])).cuda(), return_sequence=True)
    seq = np.stack([s.cpu().numpy() for s in seq]).astype(np.float64)
    d_loop = float(np.abs(seq - g["seq_f64"]).max())
    print(f"{name} {cfg}: forward max|d| {d_fwd:.2e} (the float32 reference: {ref_gap:.2e}), N=4 trajectory {d_loop:.2e}")
    assert d_fwd <= 2e-5 and d_loop <= 1e-4                                  # the parity bars of the default architecture
    assert np.array_equal(seq[0], g["x_T"].astype(np.float64))
    # device noise: reproducible, per-utterance streams independent of the batch position, the int16 epilogue works on any length
    with torch.no_grad():
        mel = torch.from_numpy(g["mel"]).cuda()
        a = m.sample(mel, rows, seed=5, stream_ids=[11, 12])
        b = m.sample(mel.flip(0).contiguous(), rows, seed=5, stream_ids=[12, 11])
        assert torch.isfinite(a).all() and torch.equal(a, b.flip(0))
        pcm = m.peak_normalize_int16(a)
        assert pcm.dtype == torch.int16 and int(pcm.abs().max()) == 32767
    with pytest.raises(NotImplementedError, match="tuned kernel set"):
        m.read_tap("a0")


@pytest.mark.gpu
def test_a_configuration_the_fixtures_do_not_hold_against_the_oracle():
    """The oracle takes the configuration as a parameter (pinned on the three reference-executed fixtures above), so any other shape can
    be checked against it: base.yaml's ratios and layers with 24 inner channels, B = 3, T = 9, steps spread over the schedule."""
    import fastdiff_amd
    from oracle import Oracle
    cfg = dict(upsample_ratios=[8, 8, 4], lvc_layers_each_block=4, kpnet_hidden_channels=64, inner_channels=24)
    m = fastdiff_amd.FastDiff(**cfg)
    sd = synth.synth_state_dict(99, cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    m = m.cuda().eval()
    B, T = 3, 9
    mel, audio = synth.synth_mel(5, B, T), synth.synth_audio(5, B, T)
    steps = np.array([0.5, 77.25, 999.0], np.float32)
    o = Oracle("f64", cfg)
    o.set_weights(sd)
    ref = o.forward(audio, mel, steps)
    with torch.no_grad():
        y = m((torch.from_numpy(audio).cuda(), torch.from_numpy(mel).cuda(), torch.from_numpy(steps).view(B, 1).cuda()))
    assert float(np.abs(y.cpu().numpy() - ref).max()) <= 2e-5


@pytest.mark.gpu
def test_ragged_batch_on_the_generic_kernels_equals_each_utterance_alone():
    """`lens` on a non-default architecture means what it means on the tuned path: every utterance of a zero-padded batch gets, inside
    its own length, the bits it gets when it runs alone (forward with injected x, the N=4 loop on per-utterance noise streams); without
    `lens` the padded batch is computed as the reference computes it and the short utterance's tail differs."""
    import fastdiff_amd
    import gpu_common as gc
    sch = load_golden("schedule")
    cfg = dict(inner_channels=8, cond_channels=40, upsample_ratios=[2, 5, 3], lvc_layers_each_block=3, lvc_kernel_size=5, kpnet_hidden_channels=32,
               kpnet_conv_size=5, diffusion_step_embed_dim_in=64, diffusion_step_embed_dim_mid=256, diffusion_step_embed_dim_out=128)
    m = fastdiff_amd.FastDiff(**cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.synth_state_dict(21, cfg).items()}, strict=True)
    m = m.cuda().eval()
    hop, T, lens = m.hop_length, 9, [9, 4, 1]
    B = len(lens)
    mel = synth.synth_mel(8, B, T, cond=40)
    x = synth.synth_audio(8, B, T, hop=hop)
    for b, t in enumerate(lens):
        mel[b, :, t:] = 0.0
    steps = torch.tensor([[3.0], [40.5], [700.0]]).cuda()
    rows, _ = gc.table_rows(sch, 4)
    with torch.no_grad():
        y = m.forward((torch.from_numpy(x).cuda(), torch.from_numpy(mel).cuda(), steps), lens=lens)
        w = m.sample(torch.from_numpy(mel).cuda(), rows, seed=3, lens=lens, stream_ids=[5, 6, 7])
        w_padded = m.sample(torch.from_numpy(mel).cuda(), rows, seed=3, stream_ids=[5, 6, 7])
        for b, t in enumerate(lens):
            one = m.sample(torch.from_numpy(np.ascontiguousarray(mel[b:b + 1, :, :t])).cuda(), rows, seed=3, stream_ids=[5 + b])
            assert torch.equal(w[b, :, : t * hop], one[0]), b
            y1 = m((torch.from_numpy(np.ascontiguousarray(x[b:b + 1, :, : t * hop])).cuda(), torch.from_numpy(np.ascontiguousarray(mel[b:b + 1, :, :t])).cuda(), steps[b:b + 1]))
            assert torch.equal(y[b, :, : t * hop], y1[0]), b
        assert torch.equal(w_padded[0], w[0])                                  # the full-length utterance needs no lens
        assert not torch.equal(w_padded[1, :, : lens[1] * hop], w[1, :, : lens[1] * hop])   # the padded computation is a different signal near the end
    with pytest.raises(AssertionError, match="lens"):
        m.sample(torch.from_numpy(mel).cuda(), rows, lens=[9, 4, 10])
