"""numpy/ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE -- never imported by fastdiff_amd/).

Each method restates one reference function; see fastdiff_oracle.c for the file:line citations.
"""
import ctypes as ct
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)
import build as _build   # noqa: E402
import synth as _synth   # noqa: E402


class _Cfg(ct.Structure):
    _fields_ = [("inner_channels", ct.c_int), ("cond_channels", ct.c_int), ("n_blocks", ct.c_int),
                ("ratios", ct.c_int * 8), ("lvc_layers", ct.c_int), ("lvc_kernel_size", ct.c_int),
                ("kp_hidden", ct.c_int), ("kp_conv_size", ct.c_int), ("embed_in", ct.c_int),
                ("embed_mid", ct.c_int), ("embed_out", ct.c_int)]


def _cfg_struct(cfg=None):
    """fdo_config of the reference constructor's arguments (FastDiff_model.py:13-26); cfg: those that differ from base.yaml's."""
    d = _synth.full_cfg(cfg)
    assert d["audio_channels"] == 1 and 1 <= len(d["upsample_ratios"]) <= 8
    c = _Cfg()
    c.inner_channels, c.cond_channels, c.n_blocks = d["inner_channels"], d["cond_channels"], len(d["upsample_ratios"])
    for i, r in enumerate(d["upsample_ratios"]):
        c.ratios[i] = r
    c.lvc_layers, c.lvc_kernel_size, c.kp_hidden, c.kp_conv_size = (d["lvc_layers_each_block"], d["lvc_kernel_size"],
                                                                    d["kpnet_hidden_channels"], d["kpnet_conv_size"])
    c.embed_in, c.embed_mid, c.embed_out = d["diffusion_step_embed_dim_in"], d["diffusion_step_embed_dim_mid"], d["diffusion_step_embed_dim_out"]
    return c


def canonical_weights(sd: dict, dtype, cfg=None) -> list:
    """Reference-named state_dict (numpy) -> folded weights in fdo_forward's canonical order.

    The weight-norm fold w = v * (g/||v||) is evaluated in `dtype` (the reference folds in the model dtype)."""
    dtype = np.dtype(dtype)
    d = _synth.full_cfg(cfg)
    n_blocks, layers = len(d["upsample_ratios"]), d["lvc_layers_each_block"]

    def wb(name):
        if name + ".weight_v" in sd:
            v = np.asarray(sd[name + ".weight_v"]).astype(dtype)
            g = np.asarray(sd[name + ".weight_g"]).astype(dtype)
            norm = np.sqrt((v * v).sum(axis=tuple(range(1, v.ndim)), keepdims=True))
            w = v * (g / norm)
        else:
            w = np.asarray(sd[name + ".weight"]).astype(dtype)
        return [np.ascontiguousarray(w, dtype=dtype), np.ascontiguousarray(np.asarray(sd[name + ".bias"]), dtype=dtype)]

    out = wb("first_audio_conv") + wb("fc_t1") + wb("fc_t2")
    for dd in range(n_blocks):
        out += wb(f"downsample.{dd}.residual_dense")
        for i in range(3):
            out += wb(f"downsample.{dd}.conv.{i}")
    for n in range(n_blocks):
        p = f"lvc_blocks.{n}"
        out += wb(f"{p}.fc_t") + wb(f"{p}.upsample") + wb(f"{p}.kernel_predictor.input_conv.0")
        for j in _synth.KP_RES_IDX:
            out += wb(f"{p}.kernel_predictor.residual_conv.{j}")
        out += wb(f"{p}.kernel_predictor.kernel_conv") + wb(f"{p}.kernel_predictor.bias_conv")
        for i in range(layers):
            out += wb(f"{p}.convs.{i}")
    out += wb("final_conv.0")
    return out


class Oracle:
    """precision: 'f64' (truth) or 'f32' (rounds like the reference's fp32 path, no FMA contraction)."""

    def __init__(self, precision: str = "f64", cfg=None):
        """cfg: the reference constructor's arguments that differ from base.yaml's (FastDiff_model.py:13-26); None = the default model."""
        libs = _build.build()
        self.lib = ct.CDLL(libs[precision])
        self.dtype = np.dtype(np.float64 if precision == "f64" else np.float32)
        assert self.lib.fdo_real_bytes() == self.dtype.itemsize
        self.cfg_dict = _synth.full_cfg(cfg)
        self.is_default = self.cfg_dict == _synth.full_cfg(None)
        self.cfg = _cfg_struct(cfg)
        self.hop = int(np.prod(self.cfg_dict["upsample_ratios"]))
        self.lib.fdo_map_noise_scale_to_time_step.restype = ct.c_double
        self._w = None
        self._wptr = None
        self.table = self.embed_table(self.cfg_dict["diffusion_step_embed_dim_in"] // 2)

    def set_threads(self, n):
        """OpenMP threads used by the oracle; returns the count in effect."""
        return int(self.lib.fdo_set_threads(int(n)))

    # -- helpers -----------------------------------------------------------------------------
    def _a(self, x):
        return np.ascontiguousarray(x, dtype=self.dtype)

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(ct.c_void_p)

    def _ptr_array(self, arrays):
        arr = (ct.c_void_p * len(arrays))()
        for i, a in enumerate(arrays):
            arr[i] = None if a is None else a.ctypes.data
        return arr

    def set_weights(self, sd: dict):
        self._w = canonical_weights(sd, self.dtype, self.cfg_dict)
        assert len(self._w) == self.lib.fdo_num_weights(ct.byref(self.cfg))
        self._wptr = self._ptr_array(self._w)

    # -- elementary ops ----------------------------------------------------------------------
    def weight_norm_fold(self, v, g):
        v = self._a(v); g = self._a(g).reshape(-1)
        w = np.empty_like(v)
        self.lib.fdo_weight_norm_fold(self._p(v), self._p(g), ct.c_int(v.shape[0]), ct.c_int(int(np.prod(v.shape[1:]))), self._p(w))
        return w

    def conv1d(self, x, w, b, dil=1):
        x = self._a(x); w = self._a(w); b = self._a(b)
        B, cin, L = x.shape
        cout, _, ks = w.shape
        out = np.empty((B, cout, L), self.dtype)
        self.lib.fdo_conv1d(self._p(x), B, cin, ct.c_int64(L), self._p(w), self._p(b), cout, ks, dil, self._p(out))
        return out

    def conv_transpose1d(self, x, w, b, r):
        x = self._a(x); w = self._a(w); b = self._a(b)
        B, cin, L = x.shape
        cout = w.shape[1]
        out = np.empty((B, cout, L * r), self.dtype)
        self.lib.fdo_conv_transpose1d(self._p(x), B, cin, ct.c_int64(L), self._p(w), self._p(b), cout, r, self._p(out))
        return out

    def embed_table(self, half=64):
        t = np.empty(half, np.float32)
        self.lib.fdo_embed_table(half, t.ctypes.data_as(ct.c_void_p))
        return t

    def step_embedding(self, steps, table=None):
        steps = self._a(steps).reshape(-1)
        table = self.table if table is None else np.ascontiguousarray(table, np.float32)
        out = np.empty((steps.shape[0], 2 * table.shape[0]), self.dtype)
        self.lib.fdo_step_embedding(self._p(steps), steps.shape[0], self._p(table), table.shape[0], self._p(out))
        return out

    def dblock(self, x, factor, weights):
        """weights: [res_w, res_b, c0_w, c0_b, c1_w, c1_b, c2_w, c2_b]"""
        x = self._a(x)
        ws = [self._a(a) for a in weights]
        B, C, L = x.shape
        out = np.empty((B, C, L // factor), self.dtype)
        self.lib.fdo_dblock(self._p(x), B, C, ct.c_int64(L), factor, self._ptr_array(ws), self._p(out))
        return out

    def lvc(self, x, kernel, bias, hop):
        """x [B,cin,L], kernel [B,cin,cout,ks,T], bias [B,cout,T] -> [B,cout,L]  (modules.py:220-253)"""
        x = self._a(x); kernel = self._a(kernel); bias = self._a(bias)
        B, cin, L = x.shape
        _, _, cout, ks, T = kernel.shape
        assert L == T * hop
        out = np.empty((B, cout, L), self.dtype)
        self.lib.fdo_lvc(self._p(x), B, cin, T, hop, self._p(kernel), self._p(bias), cout, ks, self._p(out))
        return out

    # -- the denoiser ------------------------------------------------------------------------
    def forward(self, audio, mel, steps, taps: bool = False):
        """FastDiff.forward((audio[B,1,L], mel[B,80,T], steps[B,1])) -> eps [B,1,L]; taps -> dict of intermediates."""
        assert self._wptr is not None, "set_weights first"
        audio = self._a(audio); mel = self._a(mel); steps = self._a(steps).reshape(-1)
        B, _, T = mel.shape
        L = T * self.hop
        assert audio.shape == (B, 1, L) and mel.shape[1] == self.cfg_dict["cond_channels"]
        out = np.empty((B, 1, L), self.dtype)
        tap_arrays, tap_ptr = None, None
        if taps:
            assert self.is_default, "taps are laid out for the default architecture"
            C = _synth.C
            lens = [L, L // 4, L // 32, T]
            tap_arrays = [np.empty((B, _synth.E_OUT), self.dtype)]
            tap_arrays += [np.empty((B, C, n), self.dtype) for n in lens]
            hop = 1
            for n in range(3):
                hop *= _synth.RATIOS[n]
                tap_arrays += [np.empty((B, _synth.L_W, T), self.dtype), np.empty((B, _synth.L_B, T), self.dtype),
                               np.empty((B, C, T * hop), self.dtype)]
            tap_ptr = self._ptr_array(tap_arrays)
        self.lib.fdo_forward(ct.byref(self.cfg), self._wptr, self._p(self.table), self._p(audio), self._p(mel),
                             self._p(steps), B, T, self._p(out), tap_ptr)
        if not taps:
            return out
        names = ["embed", "a0", "a1", "a2", "a3"]
        for n in range(3):
            names += [f"kernels{n}", f"bias{n}", f"x{n}"]
        return out, dict(zip(names, tap_arrays))

    # -- schedule math (fp32, as the reference's torch ops) -------------------------------------
    def compute_hyperparams(self, beta):
        beta = np.ascontiguousarray(beta, np.float32)
        alpha = np.empty_like(beta); sigma = np.empty_like(beta)
        self.lib.fdo_compute_hyperparams(self._p(beta), beta.shape[0], self._p(alpha), self._p(sigma))
        return alpha, sigma

    def map_noise_scale_to_time_step(self, alpha_infer, alpha):
        alpha = np.ascontiguousarray(alpha, np.float32)
        return self.lib.fdo_map_noise_scale_to_time_step(ct.c_float(float(alpha_infer)), self._p(alpha), alpha.shape[0])

    def inference_coefficients(self, beta):
        beta = np.ascontiguousarray(beta, np.float32)
        outs = [np.empty_like(beta) for _ in range(7)]
        self.lib.fdo_inference_coefficients(self._p(beta), beta.shape[0], *[self._p(o) for o in outs])
        return dict(zip(["alpha_hat", "sigma_hat", "c_eps", "c_div", "c1", "c2", "c3"], outs))

    def step_table(self, beta, alpha_train):
        """Everything sampling_given_noise_schedule derives from the schedule (util.py:187-209)."""
        co = self.inference_coefficients(beta)
        steps = [self.map_noise_scale_to_time_step(a, alpha_train) for a in co["alpha_hat"]]
        keep = [i for i, s in enumerate(steps) if s >= 0]
        co["steps"] = np.array([steps[i] for i in keep], np.float64)
        return co

    # -- the reverse loop with injected noise --------------------------------------------------
    def sample(self, mel, table, x_T, z, ddim=False, return_sequence=False):
        """The reverse loop (util.py:211-235) with injected noise.

        table: dict with 'steps' (mapped time steps, length N) and per-step 'c_eps','c_div','sigma_hat','c1','c2','c3'
        (from step_table(), from the golden schedule fixture, or from the product's host code -- torch.sqrt is not
        correctly rounded, so tables derived on different hosts may differ by an ulp; callers choose which to pin).
        Steps are rounded to float32 first, as torch.FloatTensor(steps_infer) does (util.py:204).
        x_T [B,1,L]; z [N,B,1,L] (z[n] is added after step n for n>0)."""
        assert self._wptr is not None
        mel = self._a(mel); x_T = self._a(x_T); z = self._a(z)
        B, _, T = mel.shape
        N = len(table["steps"])
        f64 = lambda a: np.ascontiguousarray(a, np.float64)   # noqa: E731
        out = np.empty_like(x_T)
        seq = np.empty((N + 1,) + x_T.shape, self.dtype) if return_sequence else None
        steps32 = np.asarray(table["steps"], np.float64).astype(np.float32)
        args = [f64(steps32)] + [f64(table[k]) for k in ("c_eps", "c_div", "sigma_hat", "c1", "c2", "c3")]
        self.lib.fdo_sample(ct.byref(self.cfg), self._wptr, self._p(self.table), self._p(mel), B, T, N,
                            *[self._p(a) for a in args[:4]], *[self._p(a) for a in args[4:]], int(bool(ddim)),
                            self._p(x_T), self._p(z), self._p(out), None if seq is None else self._p(seq))
        return seq if return_sequence else out

    def peak_normalize_int16(self, wav):
        wav = self._a(wav)
        B = wav.shape[0]
        L = int(np.prod(wav.shape[1:]))
        pcm = np.empty((B, L), np.int16)
        self.lib.fdo_peak_normalize_int16(self._p(wav), B, ct.c_int64(L), self._p(pcm))
        return pcm.reshape(wav.shape)
