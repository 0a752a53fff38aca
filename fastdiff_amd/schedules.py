"""Inference noise schedules selected by N in FastDiffTask.test_step (modules/FastDiff/task/FastDiff.py:65-96)."""
import torch

# literal values derived by the reference's noise predictor (FastDiff.py:82-91)
_LITERAL = {
    8: [6.689325005027058e-07, 1.0033881153503899e-05, 0.00015496854030061513, 0.002387222135439515,
        0.035597629845142365, 0.3681158423423767, 0.4735414385795593, 0.5],
    6: [1.7838445955931093e-06, 2.7984189728158526e-05, 0.00043231004383414984, 0.006634317338466644,
        0.09357017278671265, 0.6000000238418579],
    4: [3.2176e-04, 2.5743e-03, 2.5376e-02, 7.0414e-01],
    3: [9.0000e-05, 9.0000e-03, 6.0000e-01],
}


def noise_schedule_for(reverse_step, noise_schedule=''):
    """hparams['noise_schedule'] (a list) overrides; otherwise pick by N; unknown N -> NotImplementedError (FastDiff.py:92-93)."""
    if noise_schedule != '' and noise_schedule is not None:
        if isinstance(noise_schedule, list):
            return torch.FloatTensor(noise_schedule)
        return noise_schedule
    try:
        reverse_step = int(reverse_step)
    except (TypeError, ValueError):
        print('Please specify $N (the number of revere iterations) in config file. Now denoise with 4 iterations.')
        reverse_step = 4
    if reverse_step == 1000:
        return torch.linspace(0.000001, 0.01, 1000)
    if reverse_step == 200:
        return torch.linspace(0.0001, 0.02, 200)
    if reverse_step in _LITERAL:
        return torch.FloatTensor(_LITERAL[reverse_step])
    raise NotImplementedError


def training_hyperparams():
    """beta = linspace(beta_0, beta_T, T) of base.yaml:38-40, as FastDiffTask.build_model does (FastDiff.py:31-41)."""
    from .sampler import compute_hyperparams_given_schedule
    return compute_hyperparams_given_schedule(torch.linspace(0.000001, 0.01, 1000))
