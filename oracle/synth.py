"""Deterministic synthetic weights / inputs for parity tests (TEST INFRASTRUCTURE).

The reference ships no checkpoint that is reachable offline, and a 61 MB random state_dict is too
large to commit.  So the golden fixtures are generated from weights that BOTH sides can rebuild
bit-exactly from a seed with integer arithmetic only (a counter-based splitmix64 hash -> 24-bit
uniform), scaled like torch's default Conv1d/Linear init (uniform +-1/sqrt(fan_in)) so the
network is as well conditioned as `torch.manual_seed(1234); FastDiff()`.

Parameter names / shapes are those of the reference state_dict
(modules/FastDiff/module/FastDiff_model.py:13-72, modules.py:116-125,141-187,257-318).
"""
import numpy as np

RATIOS = (8, 8, 4)
C, COND, HID, LAYERS, KS = 32, 80, 64, 4, 3
E_IN, E_MID, E_OUT = 128, 512, 512
L_W = C * 2 * C * KS * LAYERS   # 24576
L_B = 2 * C * LAYERS            # 256
KP_RES_IDX = (1, 3, 6, 8, 11, 13)   # Conv1d positions inside KernelPredictor.residual_conv (modules.py:298-313)


DEFAULT_CFG = dict(audio_channels=1, inner_channels=C, cond_channels=COND, upsample_ratios=list(RATIOS), lvc_layers_each_block=LAYERS,
                   lvc_kernel_size=KS, kpnet_hidden_channels=HID, kpnet_conv_size=3, diffusion_step_embed_dim_in=E_IN,
                   diffusion_step_embed_dim_mid=E_MID, diffusion_step_embed_dim_out=E_OUT)


def full_cfg(cfg=None) -> dict:
    """The reference constructor's keyword arguments (FastDiff_model.py:13-26) with base.yaml's values for what `cfg` leaves out."""
    out = dict(DEFAULT_CFG)
    out.update(cfg or {})
    out["upsample_ratios"] = [int(r) for r in out["upsample_ratios"]]
    return out


def param_spec(cfg=None):
    """[(name, shape, kind)] in reference state_dict naming; kind: 'wn' (weight-normed Conv1d -> _g/_v), 'plain'.
    cfg: constructor arguments that differ from base.yaml's (None = the default architecture, in the order the fixtures were made with)."""
    c = full_cfg(cfg)
    ch, cond, hid, layers, ks, kk = (c["inner_channels"], c["cond_channels"], c["kpnet_hidden_channels"], c["lvc_layers_each_block"],
                                     c["lvc_kernel_size"], c["kpnet_conv_size"])
    e_in, e_mid, e_out = c["diffusion_step_embed_dim_in"], c["diffusion_step_embed_dim_mid"], c["diffusion_step_embed_dim_out"]
    spec = [("first_audio_conv", (ch, 1, 7), "wn"),
            ("fc_t1", (e_mid, e_in), "plain"), ("fc_t2", (e_out, e_mid), "plain")]
    for n, r in enumerate(c["upsample_ratios"]):
        p = f"lvc_blocks.{n}"
        spec.append((f"{p}.upsample", (ch, ch, 2 * r), "plain"))
        spec.append((f"{p}.kernel_predictor.input_conv.0", (hid, cond, 5), "wn"))
        for j in KP_RES_IDX:
            spec.append((f"{p}.kernel_predictor.residual_conv.{j}", (hid, hid, kk), "wn"))
        spec.append((f"{p}.kernel_predictor.kernel_conv", (ch * 2 * ch * ks * layers, hid, kk), "wn"))
        spec.append((f"{p}.kernel_predictor.bias_conv", (2 * ch * layers, hid, kk), "wn"))
        spec.append((f"{p}.fc_t", (cond, e_out), "plain"))
        for i in range(layers):
            spec.append((f"{p}.convs.{i}", (ch, ch, ks), "wn"))
        d = f"downsample.{n}"
        spec.append((f"{d}.residual_dense", (ch, ch, 1), "wn"))
        for i in range(3):
            spec.append((f"{d}.conv.{i}", (ch, ch, 3), "wn"))
    spec.append(("final_conv.0", (c["audio_channels"], ch, 7), "wn"))
    return spec


_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def hash_uniform(seed: int, stream: int, n: int) -> np.ndarray:
    """n float32 values uniform in [-1, 1), reproducible bit-for-bit everywhere (integer ops only)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        base = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream))
        h = _splitmix64(idx ^ base)
    u24 = (h >> np.uint64(40)).astype(np.int64)            # 24 bits
    return ((u24 - (1 << 23)).astype(np.float32) / np.float32(1 << 23)).astype(np.float32)


def hash_normal(seed: int, stream: int, n: int) -> np.ndarray:
    """Approximately N(0,1) float32 (sum of 12 uniforms), reproducible; used for x_T and injected noise."""
    acc = np.zeros(n, dtype=np.float64)
    for k in range(12):
        acc += hash_uniform(seed, stream * 16 + k, n).astype(np.float64) * 0.5 + 0.5
    return (acc - 6.0).astype(np.float32)


def synth_state_dict(seed: int = 1234, cfg=None, contractive: bool = False) -> dict:
    """Reference-named float32 state_dict: weight_v/weight_g/bias for weight-normed convs, weight/bias otherwise.
    cfg: see param_spec (another architecture gets its own shapes from the same hash streams).
    contractive: see make_contractive (a stand-in for a trained denoiser: x stays O(1) over a 1000-step schedule)."""
    sd = {}
    for stream, (name, shape, kind) in enumerate(param_spec(cfg)):
        n = int(np.prod(shape))
        if name.endswith(".upsample"):
            fan_in = shape[1] * shape[2]     # ConvTranspose1d: torch computes fan_in from dim 1
        else:
            fan_in = int(np.prod(shape[1:]))
        bound = np.float32(1.0 / np.sqrt(fan_in))
        w = (hash_uniform(seed, 3 * stream, n) * bound).reshape(shape)
        nb = shape[1] if name.endswith(".upsample") else shape[0]
        b = hash_uniform(seed, 3 * stream + 1, nb) * bound
        if kind == "wn":
            norm = np.sqrt((w.astype(np.float64) ** 2).sum(axis=(1, 2), keepdims=True)).astype(np.float32)
            g = norm * (np.float32(1.0) + np.float32(0.1) * hash_uniform(seed, 3 * stream + 2, shape[0]).reshape(-1, 1, 1))
            sd[name + ".weight_v"] = w
            sd[name + ".weight_g"] = g.astype(np.float32)
        else:
            sd[name + ".weight"] = w
        sd[name + ".bias"] = b.astype(np.float32)
    return make_contractive(sd) if contractive else sd


def make_contractive(sd: dict, lam: float = 1.0, gain: float = 0.5) -> dict:
    """Turns hash weights into a denoiser whose output is positively correlated with its input, as a trained one's is
    (eps ~ (x - alpha x_0) / sqrt(1 - alpha^2)), so that a long reverse schedule contracts instead of diverging.

    The network has one linear path from x to eps: `x += audio_down` adds the skip a0 = first_audio_conv(x) to the residual stream in
    each of the last block's four layers (modules.py:208-209) and final_conv reads that stream (FastDiff_model.py:100), so
    eps contains 4 * (final_conv * first_audio_conv) * x.  Adding `lam` times the time-reversed first-conv taps to final_conv's direction
    makes the centre tap of that cascade 4 * lam * sum(w_first^2) > 0 (a flat, positive gain over all frequencies) on top of the random
    13-tap filter the hash weights give; weight_g = `gain` sets final_conv's norm.  With the plain hash weights |x| grows to ~630 over
    N = 1000 (tests/golden/sample_s6.npz); with these it stays <= ~1 after the first 125 steps and ends inside [-1, 1] like a waveform.
    Only final_conv.0.weight_v / weight_g change; every value is still a pure function of the seed."""
    out = dict(sd)
    w1v, w1g = sd["first_audio_conv.weight_v"], sd["first_audio_conv.weight_g"]
    n1 = np.sqrt((w1v.astype(np.float64) ** 2).sum(axis=(1, 2), keepdims=True)).astype(np.float32)
    w1 = (w1g * w1v / n1).astype(np.float32)                               # folded first conv [32,1,7]
    v = sd["final_conv.0.weight_v"] + np.float32(lam) * w1[:, 0, ::-1][None]
    out["final_conv.0.weight_v"] = np.ascontiguousarray(v, dtype=np.float32)
    out["final_conv.0.weight_g"] = np.full((1, 1, 1), gain, np.float32)
    return out


def synth_mel(seed: int, B: int, T: int, lo: float = -6.0, hi: float = 1.5, cond: int = COND) -> np.ndarray:
    """Uniform on [mel_vmin, mel_vmax] (base.yaml:15-16), shape [B,80,T] (cond: another cond_channels)."""
    u = hash_uniform(seed, 900001, B * cond * T) * np.float32(0.5) + np.float32(0.5)
    return (u * np.float32(hi - lo) + np.float32(lo)).reshape(B, cond, T).astype(np.float32)


def synth_audio(seed: int, B: int, T: int, stream: int = 900002, hop: int = 256) -> np.ndarray:
    return hash_normal(seed, stream, B * T * hop).reshape(B, 1, T * hop)


def stub_noise_pred_batch(x, cond):
    """The same stand-in for a batch (phi_loss, util.py:356): x [B, L]; cond = (beta_next [B,1], delta^2 [B,1]) -> beta [B,1,1] in
    (0, min(beta_next, delta^2)), a smooth function of each item's own x."""
    import torch
    beta_next, delta2 = cond
    ratio = 0.3 + 0.2 * torch.tanh(x.abs().mean(-1, keepdim=True))
    return torch.minimum(beta_next * ratio, delta2 * 0.9).view(-1, 1, 1)


def stub_noise_pred(x, cond):
    """A stand-in for the BDDM scheduling network the reference calls but does not ship (`net.noise_pred`, util.py:284-285;
    SURVEY.md 3.5): any deterministic, smooth function of (x, beta_next, 1 - alpha^2) exercises noise_scheduling's arithmetic.
    x [B, L]; cond = (beta_next [1,1], delta [1,1]) -> beta [1,1,1] in (0, min(beta_next, delta))."""
    import torch
    beta_next, delta = cond
    ratio = 0.3 + 0.2 * torch.tanh(x.abs().mean())
    return torch.minimum(beta_next * ratio, delta * 0.9).view(1, 1, 1)
