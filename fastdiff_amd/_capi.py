"""ctypes binding of libfastdiff_hip.so (include/fastdiff_hip.h, fastdiff_hip_ext.h, fastdiff_hip_train.h).  No torch types cross this boundary."""
import ctypes as ct
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfastdiff_hip.so")

FD_OK, FD_ERR_INVALID, FD_ERR_UNSUPPORTED, FD_ERR_HIP, FD_ERR_STATE, FD_ERR_MISSING = 0, -1, -2, -3, -4, -5


class FdConfig(ct.Structure):
    _fields_ = [("audio_channels", ct.c_int), ("inner_channels", ct.c_int), ("cond_channels", ct.c_int),
                ("n_upsample", ct.c_int), ("upsample_ratios", ct.c_int * 8), ("lvc_layers_each_block", ct.c_int),
                ("lvc_kernel_size", ct.c_int), ("kpnet_hidden_channels", ct.c_int), ("kpnet_conv_size", ct.c_int),
                ("diffusion_step_embed_dim_in", ct.c_int), ("diffusion_step_embed_dim_mid", ct.c_int),
                ("diffusion_step_embed_dim_out", ct.c_int), ("use_weight_norm", ct.c_int)]


class FdStep(ct.Structure):
    _fields_ = [("t", ct.c_float), ("c_eps", ct.c_float), ("c_div", ct.c_float), ("sigma", ct.c_float),
                ("c1", ct.c_float), ("c2", ct.c_float), ("c3", ct.c_float), ("add_noise", ct.c_int32)]


class FdKernelStat(ct.Structure):
    _fields_ = [("name", ct.c_char * 48), ("launches", ct.c_int64), ("total_ms", ct.c_double)]


def step_table(rows):
    """[{t, c_eps, c_div, sigma, c1, c2, c3, add_noise}] (executed first -> last) -> the fd_step array fd_sample takes."""
    steps = (FdStep * len(rows))()
    for k, row in enumerate(rows):
        steps[k] = FdStep(float(row["t"]), float(row["c_eps"]), float(row["c_div"]), float(row["sigma"]),
                          float(row["c1"]), float(row["c2"]), float(row["c3"]), int(row["add_noise"]))
    return steps


EXPORTS = ["fd_default_config", "fd_create", "fd_destroy", "fd_last_error", "fd_set_weight", "fd_commit_weights",
           "fd_forward", "fd_sample", "fd_sample_check", "fd_sample_ticket", "fd_sample_settle", "fd_set_noise_streams", "fd_peak_normalize_int16", "fd_peak_normalize_int16_ragged", "fd_mel_spectrogram", "fd_set_mel_filterbank", "fd_get_mel_filterbank", "fd_lvc_forward", "fd_lvc_backward", "fd_lvc_forward_strided", "fd_lvc_backward_strided", "fd_gate_forward", "fd_gate_backward", "fd_kconv_forward", "fd_kconv_backward", "fd_weight_norm_multi_forward", "fd_weight_norm_multi_backward", "fd_fan_forward", "fd_fan_backward", "fd_input_conv_forward", "fd_input_conv_backward", "fd_kconv_backward_w_multi", "fd_kconv_forward_act_multi", "fd_kconv_backward_x_multi", "fd_input_conv_forward_multi", "fd_input_conv_backward_multi", "fd_kconv_forward_act", "fd_kconv_backward_act", "fd_kconv_forward_frames", "fd_kconv_backward_frames", "fd_lvc_forward_frames", "fd_lvc_backward_frames", "fd_conv32_forward", "fd_conv32_backward", "fd_weight_norm_forward", "fd_weight_norm_backward", "fd_conv7_forward", "fd_conv7_backward", "fd_upsample_forward", "fd_upsample_backward", "fd_set_option", "fd_read_tap", "fd_kernel_index", "fd_bias_index",
           "fd_get_profile", "fd_reset_profile", "fd_get_counter", "fd_version", "fd_abi_revision"]

_lib = None


class FastDiffHipError(RuntimeError):
    pass


def load():
    """dlopen the HIP library.  Fails loudly: there is no Python/CPU fallback for the compute path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FastDiffHipError(
            f"{LIB_PATH} is missing: build it with `python -m fastdiff_amd.build` (hipcc, gfx950). "
            "fastdiff_amd has no CPU fallback.")
    lib = ct.CDLL(LIB_PATH)
    vp, ci, cf = ct.c_void_p, ct.c_int, ct.c_float  # noqa: F841
    lib.fd_version.restype = ct.c_char_p
    lib.fd_last_error.restype = ct.c_char_p
    lib.fd_last_error.argtypes = [vp]
    lib.fd_default_config.argtypes = [ct.POINTER(FdConfig)]
    lib.fd_create.argtypes = [ct.POINTER(FdConfig), ci, ct.POINTER(vp)]
    lib.fd_destroy.argtypes = [vp]
    lib.fd_set_weight.argtypes = [vp, ct.c_char_p, vp, ct.POINTER(ct.c_int64), ci]
    lib.fd_commit_weights.argtypes = [vp]
    lib.fd_forward.argtypes = [vp, vp, vp, vp, ci, ci, vp, vp, vp]
    lib.fd_sample.argtypes = [vp, vp, ci, ci, vp, ct.POINTER(FdStep), ci, ci, vp, vp, ct.c_uint64, vp, vp, vp]
    lib.fd_set_noise_streams.argtypes = [vp, vp, ci]
    lib.fd_sample_check.argtypes = [vp]
    lib.fd_sample_ticket.argtypes = [vp]
    lib.fd_sample_ticket.restype = ct.c_int64
    lib.fd_sample_settle.argtypes = [vp, ct.c_int64]
    lib.fd_peak_normalize_int16.argtypes = [vp, vp, ci, ct.c_int64, vp, vp]
    lib.fd_peak_normalize_int16_ragged.argtypes = [vp, vp, ci, ct.c_int64, vp, vp, vp]
    lib.fd_mel_spectrogram.argtypes = [vp, vp, ci, ct.c_int64, vp, ci, vp]
    lib.fd_set_mel_filterbank.argtypes = [vp, vp, ci, ci]
    lib.fd_get_mel_filterbank.argtypes = [vp, vp, ci, ci]
    lib.fd_lvc_forward.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, vp]
    lib.fd_lvc_backward.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp]
    lib.fd_lvc_forward_strided.argtypes = [vp, vp, vp, ct.c_int64, vp, ci, ci, ci, ci, ci, ci, vp, vp]
    lib.fd_lvc_backward_strided.argtypes = [vp, vp, vp, ct.c_int64, vp, ci, ci, ci, ci, ci, ci, vp, vp, ct.c_int64, vp, vp]
    lib.fd_kconv_forward.argtypes = [vp, vp, vp, vp, ci, ci, ci, vp, vp]
    lib.fd_kconv_backward.argtypes = [vp, vp, vp, vp, ci, ci, ci, vp, vp, vp, vp]
    lib.fd_weight_norm_multi_forward.argtypes = [vp, vp, ci, vp]
    lib.fd_weight_norm_multi_backward.argtypes = [vp, vp, ci, vp]
    lib.fd_fan_forward.argtypes = [vp, vp, ci, ct.c_int64, ci, vp, vp]
    lib.fd_fan_backward.argtypes = [vp, vp, vp, vp, vp, vp, ci, ct.c_int64, ci, vp, vp]
    lib.fd_input_conv_forward.argtypes = [vp, vp, vp, vp, ci, ci, ct.c_float, vp, vp]
    lib.fd_input_conv_backward.argtypes = [vp, vp, vp, vp, vp, ci, ci, ct.c_float, vp, vp, vp, vp]
    lib.fd_kconv_forward_act.argtypes = [vp, vp, vp, vp, ci, ci, ci, ct.c_float, vp, vp]
    lib.fd_kconv_backward_act.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ct.c_float, ct.c_float, vp, vp, vp, vp]
    lib.fd_kconv_forward_act_multi.argtypes = [vp, ci, vp, vp, vp, ci, ci, ci, ct.c_float, vp, vp]
    lib.fd_kconv_backward_x_multi.argtypes = [vp, ci, vp, vp, vp, vp, ci, ci, ci, ct.c_float, ct.c_float, vp, vp]
    lib.fd_input_conv_forward_multi.argtypes = [vp, ci, vp, vp, vp, ci, ci, ct.c_float, vp, vp]
    lib.fd_input_conv_backward_multi.argtypes = [vp, ci, vp, vp, vp, vp, ci, ci, ct.c_float, vp, vp, vp, vp]
    lib.fd_kconv_backward_w_multi.argtypes = [vp, ci, vp, vp, vp, ci, ci, ci, ct.c_float, vp, vp, vp]
    lib.fd_kconv_forward_frames.argtypes = [vp, vp, vp, vp, ci, ci, ci, vp, vp]
    lib.fd_kconv_backward_frames.argtypes = [vp, vp, vp, vp, ci, ci, ci, vp, vp, vp, vp]
    lib.fd_lvc_forward_frames.argtypes = [vp, vp, vp, ct.c_int64, vp, ct.c_int64, ci, ci, ci, vp, vp]
    lib.fd_lvc_backward_frames.argtypes = [vp, vp, vp, ct.c_int64, vp, ci, ci, ci, vp, vp, ct.c_int64, vp, ct.c_int64, vp]
    lib.fd_conv32_forward.argtypes = [vp, vp, vp, vp, vp, ci, ct.c_int64, ci, cf, cf, vp, vp, vp]
    lib.fd_conv32_backward.argtypes = [vp, vp, vp, vp, vp, vp, ci, ct.c_int64, ci, cf, cf, vp, vp, vp, vp]
    lib.fd_conv7_forward.argtypes = [vp, ci, vp, vp, vp, ci, ct.c_int64, vp, vp]
    lib.fd_conv7_backward.argtypes = [vp, ci, vp, vp, vp, ci, ct.c_int64, vp, vp, vp, vp]
    lib.fd_upsample_forward.argtypes = [vp, vp, vp, vp, ci, ct.c_int64, ci, vp, vp]
    lib.fd_upsample_backward.argtypes = [vp, vp, vp, vp, ci, ct.c_int64, ci, vp, vp, vp, vp]
    lib.fd_weight_norm_forward.argtypes = [vp, vp, vp, ct.c_int64, ci, vp, vp, vp]
    lib.fd_weight_norm_backward.argtypes = [vp, vp, vp, vp, vp, ct.c_int64, ci, vp, vp, vp]
    lib.fd_gate_forward.argtypes = [vp, vp, vp, ci, ci, ct.c_int64, vp, vp]
    lib.fd_gate_backward.argtypes = [vp, vp, vp, ci, ci, ct.c_int64, vp, vp]
    lib.fd_set_option.argtypes = [vp, ct.c_char_p, ct.c_char_p]
    lib.fd_read_tap.argtypes = [vp, ct.c_char_p, vp, ct.c_int64]
    lib.fd_read_tap.restype = ct.c_int64
    lib.fd_kernel_index.argtypes = [ci, ci, ci, ci]
    lib.fd_bias_index.argtypes = [ci, ci]
    lib.fd_get_profile.argtypes = [vp, ct.POINTER(FdKernelStat), ci]
    lib.fd_reset_profile.argtypes = [vp]
    lib.fd_get_counter.argtypes = [vp, ct.c_char_p]
    lib.fd_get_counter.restype = ct.c_int64
    _lib = lib
    return lib


def check(lib, handle, rc, what):
    """Map fd_status to the exception type the reference would raise (SURVEY 8b 'Errors')."""
    if rc >= 0:
        return rc
    msg = lib.fd_last_error(handle)
    msg = msg.decode() if msg else what
    if rc == FD_ERR_INVALID:
        raise AssertionError(msg)
    if rc == FD_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == FD_ERR_MISSING:
        raise KeyError(msg)
    raise FastDiffHipError(f"{what}: {msg} (status {rc})")
