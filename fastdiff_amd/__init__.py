"""fastdiff_amd -- MI355X-native FastDiff vocoder inference path (denoiser + N-step reverse sampler).

Drop-in names of the reference:
    fastdiff_amd.FastDiff                          <- modules.FastDiff.module.FastDiff_model.FastDiff
    fastdiff_amd.sampler (alias fastdiff_amd.util): sampling_given_noise_schedule, compute_hyperparams_given_schedule, noise_scheduling, theta_timestep_loss (under no_grad and under autograd), phi_loss, ...
                                                   <- modules.FastDiff.module.util
    fastdiff_amd.location_variable_convolution     <- TimeAware_LVCBlock.location_variable_convolution (modules.py:220-253), forward AND backward
FastDiff.forward in train() mode records an autograd graph (fastdiff_amd/train.py) whose LVC nodes are that operator.
"""
from .model import FastDiff  # noqa: F401
from . import sampler, schedules  # noqa: F401
from . import sampler as util  # noqa: F401  (the reference module is called util)
from .lvc_op import location_variable_convolution, gated_residual, kernel_conv1d, conv32  # noqa: F401
from .sampler import (compute_hyperparams_given_schedule, sampling_given_noise_schedule, noise_scheduling,  # noqa: F401
                   map_noise_scale_to_time_step, calc_diffusion_step_embedding, std_normal, theta_timestep_loss, phi_loss)

__all__ = ["FastDiff", "location_variable_convolution", "gated_residual", "kernel_conv1d", "conv32", "util", "schedules", "compute_hyperparams_given_schedule", "sampling_given_noise_schedule",
           "noise_scheduling", "map_noise_scale_to_time_step", "calc_diffusion_step_embedding", "std_normal", "theta_timestep_loss", "phi_loss"]
