"""Plain PyTorch restatement of the denoiser and the DDPM sampling loop -- TEST / BENCH INFRASTRUCTURE, not shipped.

What it is for: the reference is PyTorch eager code and /root/reference does not exist on the GPU box, so "the reference on
MI355X" cannot be timed there.  This file states the same network with torch.nn.functional ops (conv1d, conv_transpose1d, unfold +
einsum for the location-variable convolution, as eager PyTorch-ROCm would run them through MIOpen / rocBLAS) from the same
state_dict, so that bench.py --torch-eager-baseline can put an eager-framework number next to the HIP path on the same GPU.  It is
pinned on the reference-generated golden forwards (tests/test_oracle_golden.py) and is never imported by the product.

Reference rows: a1-a2 util.py:407-432 + FastDiff_model.py:85-87; a3 FastDiff_model.py:34-36; a4 modules.py:116-138;
a5 modules.py:257-343; a6-a9 modules.py:163-253; a10 FastDiff_model.py:67-68; a15 util.py:216-229.
"""
import math

import torch
import torch.nn.functional as F

RATIOS, DOWN, LAYERS, C = (8, 8, 4), (4, 8, 8), 4, 32
KP_RES = (1, 3, 6, 8, 11, 13)


def fold(sd, name):
    """weight-norm: w = g * v / ||v|| over all dims but the first (FastDiff_model.py:115-122); plain weights pass through."""
    if name + ".weight" in sd:
        return sd[name + ".weight"]
    v, g = sd[name + ".weight_v"], sd[name + ".weight_g"]
    return v * (g / v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1))))


class EagerFastDiff:
    def __init__(self, state_dict, device="cpu", dtype=torch.float32, weight_norm_each_forward=False):
        """weight_norm_each_forward: keep (v, g) and evaluate w = g v / ||v|| inside every convolution call, as the reference's
        weight-norm hooks do on every forward (FastDiff_model.py:115-122; nobody calls remove_weight_norm() at inference) -- part of
        what running the reference's own code costs, left out of the lean restatement."""
        sd = {k: torch.as_tensor(v).to(device=device, dtype=dtype) for k, v in state_dict.items()}
        self.w = {}
        self.vg = {}
        names = sorted({k.rsplit(".", 1)[0] for k in sd})
        for n in names:
            self.w[n] = (fold(sd, n), sd.get(n + ".bias"))
            if weight_norm_each_forward and n + ".weight_v" in sd:
                self.vg[n] = (sd[n + ".weight_v"], sd[n + ".weight_g"])
        self.freq = torch.exp(torch.arange(64, dtype=torch.float32) * -(math.log(10000.0) / 63)).to(device=device, dtype=dtype)

    def conv(self, name, x, dilation=1):
        w, b = self.w[name]
        if name in self.vg:
            w = torch._weight_norm(self.vg[name][0], self.vg[name][1], 0)
        return F.conv1d(x, w, b, padding=dilation * (w.shape[-1] - 1) // 2, dilation=dilation)

    def lin(self, name, x):
        w, b = self.w[name]
        return F.linear(x, w, b)

    def predictor(self, n, cond):
        p = f"lvc_blocks.{n}.kernel_predictor."
        h = F.leaky_relu(self.conv(p + "input_conv.0", cond), 0.1)
        r = h
        for i in KP_RES:
            r = F.leaky_relu(self.conv(p + f"residual_conv.{i}", r), 0.1)
        h = h + r
        B, _, T = h.shape
        return self.conv(p + "kernel_conv", h).view(B, LAYERS, C, 2 * C, 3, T), self.conv(p + "bias_conv", h).view(B, LAYERS, 2 * C, T)

    @staticmethod
    def lvc(y, kernel, bias, hop):
        """out[b,o,t*hop+s] = bias[b,o,t] + sum_{i,k} ypad[b,i,t*hop+s+k] * kernel[b,i,o,k,t]   (one zero sample each side)"""
        B, _, L = y.shape
        win = F.pad(y, (1, 1)).unfold(2, hop + 2, hop).unfold(3, 3, 1)           # [B, in, T, hop, 3]
        out = torch.einsum("bithk,biokt->both", win, kernel) + bias.unsqueeze(-1)
        return out.reshape(B, -1, L)

    def forward(self, audio, mel, steps):
        emb = steps.reshape(-1, 1).to(self.freq.dtype) * self.freq
        emb = torch.cat((emb.sin(), emb.cos()), 1)
        emb = self.lin("fc_t1", emb); emb = emb * torch.sigmoid(emb)
        emb = self.lin("fc_t2", emb); emb = emb * torch.sigmoid(emb)
        x = self.conv("first_audio_conv", audio)
        skips = []
        for d, f in enumerate(DOWN):
            skips.append(x)
            xs = x[..., ::f]
            h = xs
            for j, dil in enumerate((1, 2, 4)):
                h = self.conv(f"downsample.{d}.conv.{j}", F.leaky_relu(h, 0.2), dil)
            x = h + self.conv(f"downsample.{d}.residual_dense", xs)
        hop = 1
        for n, r in enumerate(RATIOS):
            hop *= r
            skip = skips[-1 - n]
            cond = mel + self.lin(f"lvc_blocks.{n}.fc_t", emb).unsqueeze(-1)
            kernels, biases = self.predictor(n, cond)
            w, b = self.w[f"lvc_blocks.{n}.upsample"]
            x = F.conv_transpose1d(F.leaky_relu(x, 0.2), w, b, stride=r, padding=r // 2 + r % 2, output_padding=r % 2)
            for i in range(LAYERS):
                x = x + skip
                y = F.leaky_relu(self.conv(f"lvc_blocks.{n}.convs.{i}", F.leaky_relu(x, 0.2), 3 ** i), 0.2)
                z = self.lvc(y, kernels[:, i], biases[:, i], hop)
                x = x + torch.sigmoid(z[:, :C]) * torch.tanh(z[:, C:])
        return self.conv("final_conv.0", x)

    def sample(self, mel, rows, x_T, noise=None):
        """rows: fastdiff_amd.sampler.InferenceSchedule.rows() (execution order); DDPM update of util.py:226-229."""
        x = x_T.clone()
        B = x.shape[0]
        for k, row in enumerate(rows):
            eps = self.forward(x, mel, torch.full((B,), row["t"], device=x.device))
            x = (x - row["c_eps"] * eps) / row["c_div"]
            if row["add_noise"]:
                x = x + row["sigma"] * (torch.randn_like(x) if noise is None else noise[k])
        return x

    def sample_like_the_reference(self, mel, diffusion_hyperparams, inference_noise_schedule, map_step):
        """sampling_given_noise_schedule with the host work the reference does on EVERY call and step (util.py:158-235): the
        alpha / sigma recursion and the step mapping (map_step: its 1000-iteration Python loop per level, util.py:394-404), x_T and
        every z drawn on the CPU and copied to the device (std_normal, util.py:63-68), the step tensor built on the CPU per step
        (util.py:217), in-place updates.  A restatement for timing only (bench.py --torch-eager-baseline)."""
        dev = mel.device
        alpha = diffusion_hyperparams["alpha"]
        beta = inference_noise_schedule.clone()
        a2, s2 = 1 - beta, beta + 0
        for t in range(1, len(beta)):
            a2[t] *= a2[t - 1]
            s2[t] *= (1 - a2[t - 1]) / (1 - a2[t])
        alpha_hat, sigma_hat = torch.sqrt(a2), torch.sqrt(s2)
        steps = [s for s in (map_step(a, alpha) for a in alpha_hat) if s >= 0]
        B, L = mel.shape[0], mel.shape[-1] * 256
        x = torch.normal(0, 1, size=(B, 1, L)).to(dev)
        beta_d, ah_d, sh_d = beta.to(dev), alpha_hat.to(dev), sigma_hat.to(dev)
        for n in range(len(steps) - 1, -1, -1):
            t = (steps[n] * torch.ones((B, 1))).to(dev)
            eps = self.forward(x, mel, t.view(-1))
            x -= beta_d[n] / torch.sqrt(1 - ah_d[n] ** 2.) * eps
            x /= torch.sqrt(1 - beta_d[n])
            if n > 0:
                x = x + sh_d[n] * torch.normal(0, 1, size=(B, 1, L)).to(dev)
        return x
