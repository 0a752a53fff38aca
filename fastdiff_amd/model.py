"""Host-side mirror of the reference denoiser module, backed by libfastdiff_hip.so.

`FastDiff` keeps the reference's constructor signature, parameter names / shapes / registration order
(so `state_dict()`, `load_state_dict(ckpt['state_dict']['model'])`, `.cuda()`, `.eval()` behave as for
modules/FastDiff/module/FastDiff_model.py:10-122) and the `forward(data)` contract
(FastDiff_model.py:74-102).  On the inference path the torch.nn sub-modules below are parameter holders only: no
PyTorch op runs there -- forward() hands raw device pointers to the C ABI (include/fastdiff_hip.h).  Under autograd
(train() mode, or an input that requires a gradient) forward() records the same network as autograd nodes with the
location-variable convolutions on the HIP operator, forward and backward (fastdiff_amd/train.py).
There is no CPU fallback: without the HIP library or a HIP device it raises.
"""
import ctypes as ct

import numpy as np
import torch
import torch.nn as nn

from . import _capi


class _DBlockParams(nn.Module):
    """Parameters of DiffusionDBlock (modules.py:116-125)."""

    def __init__(self, input_size, hidden_size, factor):
        super().__init__()
        self.factor = factor
        self.residual_dense = nn.Conv1d(input_size, hidden_size, 1)
        self.conv = nn.ModuleList([
            nn.Conv1d(input_size, hidden_size, 3, dilation=1, padding=1),
            nn.Conv1d(hidden_size, hidden_size, 3, dilation=2, padding=2),
            nn.Conv1d(hidden_size, hidden_size, 3, dilation=4, padding=4),
        ])


class _KernelPredictorParams(nn.Module):
    """Parameters of KernelPredictor (modules.py:257-318)."""

    def __init__(self, cond_channels, conv_in_channels, conv_out_channels, conv_layers, conv_kernel_size,
                 hidden, kp_conv_size, dropout):
        super().__init__()
        l_w = conv_in_channels * conv_out_channels * conv_kernel_size * conv_layers
        l_b = conv_out_channels * conv_layers
        pad = (kp_conv_size - 1) // 2
        act = lambda: nn.LeakyReLU(negative_slope=0.1)  # noqa: E731
        self.input_conv = nn.Sequential(nn.Conv1d(cond_channels, hidden, 5, padding=2, bias=True), act())
        layers = []
        for _ in range(3):
            layers += [nn.Dropout(dropout),
                       nn.Conv1d(hidden, hidden, kp_conv_size, padding=pad, bias=True), act(),
                       nn.Conv1d(hidden, hidden, kp_conv_size, padding=pad, bias=True), act()]
        self.residual_conv = nn.Sequential(*layers)
        self.kernel_conv = nn.Conv1d(hidden, l_w, kp_conv_size, padding=pad, bias=True)
        self.bias_conv = nn.Conv1d(hidden, l_b, kp_conv_size, padding=pad, bias=True)


class _LVCBlockParams(nn.Module):
    """Parameters of TimeAware_LVCBlock (modules.py:141-187), registered and constructed in the reference's order."""

    def __init__(self, in_channels, cond_channels, upsample_ratio, conv_layers, conv_kernel_size, cond_hop_length,
                 kpnet_hidden_channels, kpnet_conv_size, kpnet_dropout, noise_scale_embed_dim_out):
        super().__init__()
        self.cond_hop_length = cond_hop_length
        self.convs = nn.ModuleList()
        r = upsample_ratio
        self.upsample = nn.ConvTranspose1d(in_channels, in_channels, kernel_size=2 * r, stride=r,
                                           padding=r // 2 + r % 2, output_padding=r % 2)
        self.kernel_predictor = _KernelPredictorParams(cond_channels, in_channels, 2 * in_channels, conv_layers,
                                                       conv_kernel_size, kpnet_hidden_channels, kpnet_conv_size,
                                                       kpnet_dropout)
        self.fc_t = nn.Linear(noise_scale_embed_dim_out, cond_channels)
        for i in range(conv_layers):
            pad = (3 ** i) * int((conv_kernel_size - 1) / 2)
            self.convs.append(nn.Conv1d(in_channels, in_channels, kernel_size=conv_kernel_size, padding=pad,
                                        dilation=3 ** i))


class FastDiff(nn.Module):
    """Drop-in for `modules.FastDiff.module.FastDiff_model.FastDiff` on an MI355X."""

    def __init__(self, audio_channels=1, inner_channels=32, cond_channels=80, upsample_ratios=[8, 8, 4],
                 lvc_layers_each_block=4, lvc_kernel_size=3, kpnet_hidden_channels=64, kpnet_conv_size=3,
                 dropout=0.0, diffusion_step_embed_dim_in=128, diffusion_step_embed_dim_mid=512,
                 diffusion_step_embed_dim_out=512, use_weight_norm=True):
        super().__init__()
        if dropout != 0.0:
            raise NotImplementedError("fastdiff_amd is the inference path: dropout must be 0.0 (base.yaml:29)")
        self.diffusion_step_embed_dim_in = diffusion_step_embed_dim_in
        self.audio_channels = audio_channels
        self.cond_channels = cond_channels
        self.lvc_block_nums = len(upsample_ratios)
        self._cfg = dict(audio_channels=audio_channels, inner_channels=inner_channels, cond_channels=cond_channels,
                         upsample_ratios=list(upsample_ratios), lvc_layers_each_block=lvc_layers_each_block,
                         lvc_kernel_size=lvc_kernel_size, kpnet_hidden_channels=kpnet_hidden_channels,
                         kpnet_conv_size=kpnet_conv_size, diffusion_step_embed_dim_in=diffusion_step_embed_dim_in,
                         diffusion_step_embed_dim_mid=diffusion_step_embed_dim_mid,
                         diffusion_step_embed_dim_out=diffusion_step_embed_dim_out, use_weight_norm=use_weight_norm)
        self.hop_length = int(np.prod(upsample_ratios))

        self.first_audio_conv = nn.Conv1d(1, inner_channels, kernel_size=7, padding=3, dilation=1, bias=True)
        self.lvc_blocks = nn.ModuleList()
        self.downsample = nn.ModuleList()
        self.fc_t = nn.ModuleList()
        self.fc_t1 = nn.Linear(diffusion_step_embed_dim_in, diffusion_step_embed_dim_mid)
        self.fc_t2 = nn.Linear(diffusion_step_embed_dim_mid, diffusion_step_embed_dim_out)
        hop = 1
        for n in range(self.lvc_block_nums):
            hop *= upsample_ratios[n]
            self.lvc_blocks.append(_LVCBlockParams(inner_channels, cond_channels, upsample_ratios[n],
                                                   lvc_layers_each_block, lvc_kernel_size, hop, kpnet_hidden_channels,
                                                   kpnet_conv_size, dropout, diffusion_step_embed_dim_out))
            self.downsample.append(_DBlockParams(inner_channels, inner_channels,
                                                 upsample_ratios[self.lvc_block_nums - n - 1]))
        self.final_conv = nn.Sequential(nn.Conv1d(inner_channels, audio_channels, kernel_size=7, padding=3,
                                                  dilation=1, bias=True))
        if use_weight_norm:
            self.apply_weight_norm()
        # HIP side
        self._handle = None
        self._handle_device = None
        self._synced_state = None
        self._keepalive = None
        self.last_ticket = 0
        # Library options of this module's handle.  "fallback": "host" -- the range check of a sample() call is read on the host
        # instead of trailing every fp16x2 kernel with an early-exit fp32 launch (21 launches per reverse step less: -3 % at B=8,
        # -7 % at B=1).  "defer_check": "1" -- the library does not settle that check inside fd_sample (its default for a C caller,
        # include/fastdiff_hip.h): sample() / check() / settle() below do it, which lets sample(..., defer_check=True) pipeline calls.
        self._options = {"fallback": "host", "defer_check": "1"}
        self._param_list = None                # the parameter tensors _state_signature last walked (see there)
        self._sig_calls = 0

    # ---- reference API --------------------------------------------------------------------------------
    def apply_weight_norm(self):
        def _apply(m):
            if isinstance(m, (nn.Conv1d, nn.Conv2d)):
                torch.nn.utils.weight_norm(m)
        self.apply(_apply)
        self.__dict__["_param_list"] = None

    def remove_weight_norm(self):
        def _remove(m):
            try:
                torch.nn.utils.remove_weight_norm(m)
            except ValueError:
                return
        self.apply(_remove)
        self._invalidate_params()

    def forward(self, data, lens=None):
        """eps = net((audio [B,1,L], c [B,80,T] or [80,T], diffusion_steps [B,1])) -- FastDiff_model.py:74-102.
        lens (extension, optional): valid frames per utterance of a zero-padded batch, see sample()."""
        audio, c, diffusion_steps = data
        if torch.is_grad_enabled() and (self.training or audio.requires_grad or c.requires_grad):
            # training (FastDiff.py:44-49): the same network as autograd nodes, the location-variable convolutions forward and
            # backward on the HIP operator (fastdiff_amd/train.py)
            if lens is not None:
                raise NotImplementedError("lens is an extension of the inference path; under autograd pass whole utterances")
            self._require_device(audio, c)
            if not getattr(self, "_warned_autograd_path", False) and not (audio.requires_grad or c.requires_grad):
                # nn.Module starts in train() mode: a freshly built model called outside no_grad lands here, not on the fused kernels
                import warnings
                warnings.warn("fastdiff_amd.FastDiff.forward is recording an autograd graph (module in train() mode with gradients "
                              "enabled): the fused inference kernels run under torch.no_grad() or after .eval()", stacklevel=2)
                self._warned_autograd_path = True
            from .train import differentiable_forward
            return differentiable_forward(self, (audio.float(), self._prep_condition(c, audio.shape[0], audio.device), diffusion_steps))
        self._require_inference(audio, c)
        audio = audio.contiguous().float()
        B, ch, L = audio.shape
        c = self._prep_condition(c, B, audio.device)
        T = c.shape[-1]
        # the reference's check is `in_length == kernel_length * hop_size` (modules.py:236)
        assert ch == 1 and L == T * self.hop_length, "length of (x, kernel) is not matched"
        steps = diffusion_steps.to(device=audio.device, dtype=torch.float32).reshape(-1).contiguous()
        assert steps.numel() == B
        out = torch.empty_like(audio)
        lib, h = self._ready(audio.device)
        lens_arr = None if lens is None else (ct.c_int * B)(*[int(v) for v in lens])
        rc = lib.fd_forward(h, audio.data_ptr(), c.data_ptr(), steps.data_ptr(), B, T, lens_arr, out.data_ptr(),
                            self._stream(audio.device))
        _capi.check(lib, h, rc, "fd_forward")
        return out

    # ---- HIP-side entry used by fastdiff_amd.util.sampling_given_noise_schedule ---------------------------
    def sample(self, condition, table, ddim=False, x_T=None, noise=None, seed=0, return_sequence=False, lens=None, stream_ids=None,
               defer_check=False):
        """Run the N-step reverse loop on the device.

        table: list of dicts with keys t, c_eps, c_div, sigma, c1, c2, c3, add_noise (executed first -> last).
        x_T [B,1,L] / noise [N,B,1,L] optional device tensors (None -> on-device Philox keyed by `seed`).
        lens: optional valid frames per utterance of a zero-padded batch: utterance b is then computed as if it were alone
        and lens[b] frames long (its first lens[b]*256 samples are exactly that result; the rest of its row is unspecified).
        stream_ids: optional [B] integers (fd_set_noise_streams): utterance b draws its noise from Philox stream (seed, stream_ids[b])
        over its own samples, i.e. independently of its place in the batch.
        defer_check: this class runs the library with option fallback = "host" (self._options; also the C ABI's default): a
        result is final only after its range check has been looked at (fd_sample_check: one stream synchronisation and, rarely, a
        second pass on the fp32 kernels).  sample() does that before returning unless told to defer.  A deferred call is settled ONLY by
        check(), settle(ticket), forward(), the next sample() (which looks at it after enqueuing itself), set_option and a weight
        upload -- NOT by peak_normalize_int16 / mel_spectrogram, which run on a provisional waveform without waiting: read their output
        only after check() / settle(last_ticket) returned False, and compute it again when they returned True."""
        B = condition.shape[0]
        self._require_inference(condition, condition)
        condition = condition.contiguous().float()
        T = condition.shape[-1]
        L = T * self.hop_length
        N = len(table)
        steps = getattr(table, "fd_steps", None)      # sampler.StepRows: the ctypes table built once per schedule
        if steps is None or len(steps) != N:
            steps = _capi.step_table(table)
        dev = condition.device
        out = torch.empty((B, 1, L), device=dev, dtype=torch.float32)
        seq = torch.empty((N + 1, B, 1, L), device=dev, dtype=torch.float32) if return_sequence else None
        if x_T is not None:
            x_T = x_T.to(device=dev, dtype=torch.float32).contiguous()
            assert tuple(x_T.shape) == (B, 1, L)
        if noise is not None:
            noise = noise.to(device=dev, dtype=torch.float32).contiguous()
            assert tuple(noise.shape) == (N, B, 1, L)
        lib, h = self._ready(dev)
        lens_arr = None if lens is None else (ct.c_int * B)(*[int(v) for v in lens])
        if stream_ids is not None:
            assert len(stream_ids) == B
            ids = (ct.c_uint64 * B)(*[int(v) & 0xFFFFFFFFFFFFFFFF for v in stream_ids])
            _capi.check(lib, h, lib.fd_set_noise_streams(h, ids, B), "fd_set_noise_streams")
        rc = lib.fd_sample(h, condition.data_ptr(), B, T, lens_arr, steps, N, int(bool(ddim)),
                           None if x_T is None else x_T.data_ptr(), None if noise is None else noise.data_ptr(),
                           ct.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), out.data_ptr(),
                           None if seq is None else seq.data_ptr(), self._stream(dev))
        _capi.check(lib, h, rc, "fd_sample")
        # inputs (and outputs) of the calls a deferred check may still run again: this one and, under the pipelined host check,
        # the one before it (looked at only after this call was enqueued)
        self._keepalive = ((condition, x_T, noise, out, seq), self._keepalive[0] if self._keepalive else None)
        self.last_ticket = int(lib.fd_sample_ticket(h))
        if not defer_check:
            self.check()
        if return_sequence:
            return [seq[k] for k in range(N + 1)]
        return out

    def check(self):
        """Settle the last sample() under fallback = "host" (no-op otherwise).  Returns True if it had to be redone."""
        if self._handle is None:
            return False
        lib = _capi.load()
        rc = lib.fd_sample_check(self._handle)
        _capi.check(lib, self._handle, rc, "fd_sample_check")
        self._keepalive = None
        return rc == 1

    def settle(self, ticket):
        """Pipelined host check (option fallback = "host", sample(..., defer_check=True)): make the sample() call whose `last_ticket`
        was `ticket` final -- waiting for it only if nothing has looked at it yet; the next sample() on the module does so after
        enqueuing itself -- and return True if it had to be run again on the fp32 kernels: whatever the caller computed from its
        output in the meantime (epilogue, copies) must then be computed again.  With in-graph fallbacks (option fallback = "graph"): False."""
        if self._handle is None:
            return False
        lib = _capi.load()
        rc = lib.fd_sample_settle(self._handle, int(ticket))
        _capi.check(lib, self._handle, rc, "fd_sample_settle")
        return rc == 1

    def peak_normalize_int16(self, wav, valid=None):
        """wav [B,1,L] float32 -> int16 PCM [B,L]: wav/abs(wav).max() * 32767 (FastDiff.py:110; utils/audio.py:11-16).
        valid (optional, [B] sample counts of a zero-padded batch): each utterance's peak is searched over its own samples only
        and the PCM behind them is 0 -- one call and one device-to-host copy per micro-batch instead of one per utterance."""
        self._require_inference(wav, wav)
        wav = wav.contiguous().float()
        B = wav.shape[0]
        L = wav.numel() // B
        pcm = torch.empty((B, L), device=wav.device, dtype=torch.int16)
        lib, h = self._ready(wav.device)
        varr = None if valid is None else (ct.c_int64 * B)(*[int(v) for v in valid])
        rc = lib.fd_peak_normalize_int16_ragged(h, wav.data_ptr(), B, L, varr, pcm.data_ptr(), self._stream(wav.device))
        _capi.check(lib, h, rc, "fd_peak_normalize_int16")
        return pcm

    def mel_spectrogram(self, wav, n_frames=None, variant="pwg"):
        """wav [B, n] float32 in [-1, 1] (int16 PCM / 32768) -> log-mel [B, 80, T], T = 1 + n // 256 by default, on the device.
        variant "pwg": the reference's process_utterance(..., vocoder='pwg') (data_gen/tts/data_gen_utils.py:93-147): zero padding,
        filters 80-7600 Hz, log10(max(1e-6, .)).  variant "tacotron": TacotronSTFT.mel_spectrogram (data_gen/tts/tacotron/layers.py:
        42-80): reflect padding, filters 0-8000 Hz, ln(clamp(., 1e-5)); asserts the range [-1, 1] like the reference (layers.py:70-71)."""
        if variant not in ("pwg", "tacotron"):
            raise ValueError(f"mel_spectrogram: variant must be 'pwg' or 'tacotron', got {variant!r}")
        self._require_inference(wav, wav)
        wav = wav.contiguous().float()
        if wav.dim() == 1:
            wav = wav.unsqueeze(0)
        B, n = wav.shape
        if variant == "tacotron":
            assert torch.min(wav) >= -1
            assert torch.max(wav) <= 1
        T = 1 + n // self.hop_length if n_frames is None else int(n_frames)
        mel = torch.empty((B, 80, T), device=wav.device, dtype=torch.float32)
        lib, h = self._ready(wav.device)
        _capi.check(lib, h, lib.fd_set_option(h, b"mel", variant.encode()), "fd_set_option")
        _capi.check(lib, h, lib.fd_mel_spectrogram(h, wav.data_ptr(), B, n, mel.data_ptr(), T, self._stream(wav.device)), "fd_mel_spectrogram")
        return mel

    def set_mel_filterbank(self, fb, variant="pwg", device=None):
        """Use `fb` [80, 513] (numpy / tensor, float32; librosa.filters.mel's own layout) as the filter bank of front-end `variant`
        instead of the library's restated default -- what a deployment that has librosa passes
        (`librosa.filters.mel(sr=22050, n_fft=1024, n_mels=80, fmin=80, fmax=7600)`; Tacotron: fmin 0, fmax 8000;
        data_gen/tts/data_gen_utils.py:122-134, tacotron/layers.py:42-60).  The weights are used bit for bit.  fb None: back to the default."""
        import numpy as np
        if variant not in ("pwg", "tacotron"):
            raise ValueError(f"set_mel_filterbank: variant must be 'pwg' or 'tacotron', got {variant!r}")
        dev = device if device is not None else next(self.parameters()).device
        lib, h = self._ready(torch.device(dev))
        _capi.check(lib, h, lib.fd_set_option(h, b"mel", variant.encode()), "fd_set_option")
        banks = self.__dict__.setdefault("_mel_banks", {})
        if fb is None:
            banks.pop(variant, None)
            _capi.check(lib, h, lib.fd_set_mel_filterbank(h, None, 80, 513), "fd_set_mel_filterbank")
            return
        a = np.ascontiguousarray(fb.detach().cpu().numpy() if torch.is_tensor(fb) else fb, dtype=np.float32)
        if a.shape != (80, 513):
            raise ValueError(f"set_mel_filterbank: expected [80, 513] (80 mel filters over the bins of a 1024-point FFT), got {list(a.shape)}")
        _capi.check(lib, h, lib.fd_set_mel_filterbank(h, a.ctypes.data, 80, 513), "fd_set_mel_filterbank")
        banks[variant] = a.copy()

    def mel_filterbank(self, variant="pwg", device=None):
        """(bank [80, 513] float32 numpy, supplied_by_caller) of front-end `variant` as the library uses it."""
        import numpy as np
        dev = device if device is not None else next(self.parameters()).device
        lib, h = self._ready(torch.device(dev))
        _capi.check(lib, h, lib.fd_set_option(h, b"mel", variant.encode()), "fd_set_option")
        out = np.empty((80, 513), np.float32)
        rc = lib.fd_get_mel_filterbank(h, out.ctypes.data, 80, 513)
        if rc < 0:
            _capi.check(lib, h, rc, "fd_get_mel_filterbank")
        return out, bool(rc)

    # ---- options / introspection (tests, bench) ---------------------------------------------------------
    def set_option(self, key, value):
        self._options[key] = str(value)
        if self._handle is not None:
            lib = _capi.load()
            _capi.check(lib, self._handle, lib.fd_set_option(self._handle, key.encode(), str(value).encode()), "fd_set_option")

    def read_tap(self, name):
        lib = _capi.load()
        n = lib.fd_read_tap(self._handle, name.encode(), None, 0)
        _capi.check(lib, self._handle, n, "fd_read_tap")
        buf = np.empty(n, np.float32)
        _capi.check(lib, self._handle, lib.fd_read_tap(self._handle, name.encode(), buf.ctypes.data, n), "fd_read_tap")
        return buf

    def counter(self, name):
        """fd_get_counter: "pieces" / "pieces_redone" / "pieces_fp32" / "fp32_mask" of the last long sample() call, "calls_redone"."""
        lib = _capi.load()
        return int(_capi.check(lib, self._handle, lib.fd_get_counter(self._handle, name.encode()), "fd_get_counter"))

    def profile(self, reset=False):
        lib = _capi.load()
        stats = (_capi.FdKernelStat * 128)()
        n = lib.fd_get_profile(self._handle, stats, 128)
        res = {stats[i].name.decode(): (int(stats[i].launches), float(stats[i].total_ms)) for i in range(min(n, 128))}
        if reset:
            lib.fd_reset_profile(self._handle)
        return res

    @staticmethod
    def kernel_index(layer, in_ch, out_ch, tap):
        return _capi.load().fd_kernel_index(layer, in_ch, out_ch, tap)

    @staticmethod
    def bias_index(layer, out_ch):
        return _capi.load().fd_bias_index(layer, out_ch)

    # ---- internals ----------------------------------------------------------------------------------------
    @staticmethod
    def _require_device(*tensors):
        for t in tensors:
            if not t.is_cuda:
                raise RuntimeError("fastdiff_amd.FastDiff runs only on a HIP device (no CPU fallback): move the module "
                                   "and its inputs to cuda")

    def _require_inference(self, *tensors):
        self._require_device(*tensors)
        if torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
            raise NotImplementedError("this entry point is the inference pipeline (no saved activations): only FastDiff.forward "
                                      "records an autograd graph (fastdiff_amd/train.py)")

    def _prep_condition(self, c, B, device):
        c = c.to(device=device, dtype=torch.float32)
        if c.dim() == 2:                      # egs/demo.ipynb feeds [80,T]; the reference broadcasts it (modules.py:203)
            c = c.unsqueeze(0)
        if c.shape[0] != B:
            c = c.expand(B, -1, -1)
        assert c.shape[1] == self.cond_channels
        return c.contiguous()

    @staticmethod
    def _stream(device):
        return ct.c_void_p(torch.cuda.current_stream(device).cuda_stream)

    _SIG_WALK_EVERY = 64

    def _state_signature(self):
        # (storage, version) of every parameter: changes when a tensor is moved (.cuda()), loaded (load_state_dict copies in place) or
        # written in place.  The walk over the module tree that finds the tensors (named_parameters(): ~150-250 us for 175 parameters,
        # a sixth of what a one-utterance sample call takes on the GPU) is done when the set of parameters can have changed
        # (_invalidate_params: _apply, load_state_dict, apply / remove_weight_norm) and otherwise on every 64th call, which catches a
        # parameter object replaced behind the module's back (torch.nn.utils.remove_weight_norm on a sub-module, an assignment);
        # in between only the known tensors are looked at (~30 us).
        self._sig_calls += 1
        if self._param_list is None or self._sig_calls % self._SIG_WALK_EVERY == 0:
            named = list(self.named_parameters())
            self._param_names = tuple(k for k, _ in named)
            self._param_list = [v for _, v in named]
        return self._param_names, tuple((v.data_ptr(), v._version) for v in self._param_list)

    def _invalidate_params(self):
        self._param_list = None

    def _apply(self, fn, *args, **kwargs):
        self._invalidate_params()
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._invalidate_params()
        return super().load_state_dict(*args, **kwargs)

    def _ready(self, device):
        """Create the context on `device` if needed and (re)upload weights when any parameter changed."""
        lib = _capi.load()
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if self._handle is None or self._handle_device != idx:
            self._release()
            cfg = _capi.FdConfig()
            lib.fd_default_config(ct.byref(cfg))
            c = self._cfg
            if len(c["upsample_ratios"]) > 8:
                raise NotImplementedError("at most 8 upsample stages")
            cfg.audio_channels, cfg.inner_channels, cfg.cond_channels = c["audio_channels"], c["inner_channels"], c["cond_channels"]
            cfg.n_upsample = len(c["upsample_ratios"])
            for i in range(8):
                cfg.upsample_ratios[i] = c["upsample_ratios"][i] if i < cfg.n_upsample else 0
            cfg.lvc_layers_each_block, cfg.lvc_kernel_size = c["lvc_layers_each_block"], c["lvc_kernel_size"]
            cfg.kpnet_hidden_channels, cfg.kpnet_conv_size = c["kpnet_hidden_channels"], c["kpnet_conv_size"]
            cfg.diffusion_step_embed_dim_in = c["diffusion_step_embed_dim_in"]
            cfg.diffusion_step_embed_dim_mid = c["diffusion_step_embed_dim_mid"]
            cfg.diffusion_step_embed_dim_out = c["diffusion_step_embed_dim_out"]
            cfg.use_weight_norm = int(bool(c["use_weight_norm"]))
            h = ct.c_void_p()
            rc = lib.fd_create(ct.byref(cfg), idx, ct.byref(h))
            _capi.check(lib, None, rc, "fd_create")
            self._handle, self._handle_device, self._synced_state = h, idx, None
            for k, v in self._options.items():
                _capi.check(lib, h, lib.fd_set_option(h, k.encode(), v.encode()), "fd_set_option")
            for variant, a in self.__dict__.get("_mel_banks", {}).items():      # caller-supplied filter banks follow the module to a new device
                _capi.check(lib, h, lib.fd_set_option(h, b"mel", variant.encode()), "fd_set_option")
                _capi.check(lib, h, lib.fd_set_mel_filterbank(h, a.ctypes.data, 80, 513), "fd_set_mel_filterbank")
        sig = self._state_signature()
        if sig != self._synced_state:
            self._upload_weights(lib)
            self._synced_state = sig
        return lib, self._handle

    def _upload_weights(self, lib):
        h = self._handle
        for name, t in self.state_dict().items():
            a = np.ascontiguousarray(t.detach().to("cpu", torch.float32).numpy())
            dims = (ct.c_int64 * a.ndim)(*a.shape)
            rc = lib.fd_set_weight(h, name.encode(), a.ctypes.data, dims, a.ndim)
            _capi.check(lib, h, rc, f"fd_set_weight({name})")
        _capi.check(lib, h, lib.fd_commit_weights(h), "fd_commit_weights")

    def _release(self):
        if self._handle is not None:
            try:
                _capi.load().fd_destroy(self._handle)
            finally:
                self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass
