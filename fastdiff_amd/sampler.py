"""Host side of the N-step reverse sampler: the reference's `modules/FastDiff/module/util.py` entry points
(same names, arguments, return values, prints, error behaviour) re-implemented around the C ABI.

What runs where
  * schedule arithmetic -- here, on CPU fp32 torch scalars, evaluated with the same
    operations in the same order as util.py:187-204,365-404 so the kernels are fed the reference's own numbers;
  * everything per sample (denoiser, update, Gaussian draws) -- on the MI355X behind fd_sample().
"""
import numpy as np
import torch

from .model import FastDiff


# ---------------------------------------------------------------------------------------------------------
# schedule arithmetic
# ---------------------------------------------------------------------------------------------------------
def _alpha_sigma_recursion(beta):
    """alpha_t = sqrt(prod_{s<=t} (1-beta_s)),  sigma_t = sqrt(beta_t (1-alpha_{t-1}^2)/(1-alpha_t^2)).

    Sequential fp32 recursion on views of two fresh tensors (in-place `*=` on 0-d views), which is what both
    compute_hyperparams_given_schedule (util.py:380-386) and the sampler preamble (util.py:189-195) do."""
    a2 = 1 - beta          # running product of (1 - beta)
    s2 = beta + 0
    for t in range(1, len(beta)):
        a2[t] *= a2[t - 1]
        s2[t] *= (1 - a2[t - 1]) / (1 - a2[t])
    return torch.sqrt(a2), torch.sqrt(s2)


def compute_hyperparams_given_schedule(beta):
    """beta [T] -> {"T", "beta", "alpha", "sigma"} (util.py:365-390)."""
    alpha, sigma = _alpha_sigma_recursion(beta)
    return {"T": len(beta), "beta": beta, "alpha": alpha, "sigma": sigma}


def map_noise_scale_to_time_step(alpha_infer, alpha):
    """Fractional training step whose noise level equals alpha_infer (util.py:394-404).

    Clamps to the ends of the table, returns -1 when no bracket [alpha[t+1], alpha[t]] contains the value.  The reference walks
    the table with a Python loop of 0-d tensor comparisons (10 ms per call at T = 1000); here the first bracket is found with one
    vectorised comparison and the interpolation is then evaluated with the reference's own scalar expressions on that bracket, so
    the result is the same float (tests/test_host_logic.py compares with the loop on every table of the reference)."""
    last = len(alpha) - 1
    if alpha_infer < alpha[last]:
        return last
    if alpha_infer > alpha[0]:
        return 0
    inside = (alpha[1:] <= alpha_infer) & (alpha_infer <= alpha[:-1])
    hit = torch.nonzero(inside)
    if hit.numel() == 0:
        return -1
    t = int(hit[0])
    hi, lo = alpha[t], alpha[t + 1]
    frac = hi - alpha_infer
    frac /= hi - lo            # fp32 quotient, then a Python-float sum, as the reference
    return t + frac.item()


def _map_noise_scale_to_time_step_loop(alpha_infer, alpha):
    """The reference's loop form of the function above (kept for the equality test)."""
    last = len(alpha) - 1
    if alpha_infer < alpha[last]:
        return last
    if alpha_infer > alpha[0]:
        return 0
    for t in range(last):
        hi, lo = alpha[t], alpha[t + 1]
        if lo <= alpha_infer <= hi:
            frac = hi - alpha_infer
            frac /= hi - lo
            return t + frac.item()
    return -1


_EMBED_FREQ = {}


def calc_diffusion_step_embedding(diffusion_steps, diffusion_step_embed_dim_in):
    """[B,1] -> [B,dim] = cat(sin(t*f), cos(t*f)), f_j = exp(-j ln(1e4)/(dim/2-1)) (util.py:407-432).
    API parity only: on the device path the embed kernel evaluates this expression itself."""
    assert diffusion_step_embed_dim_in % 2 == 0
    half = diffusion_step_embed_dim_in // 2
    key = (half, diffusion_steps.device)
    freq = _EMBED_FREQ.get(key)
    if freq is None:      # evaluated on the CPU, as the reference does (util.py:425-427), once per device: no host-to-device copy
        freq = torch.exp(torch.arange(half) * -(np.log(10000) / (half - 1))).to(diffusion_steps.device)      # per call, so the step can be captured in a hipGraph
        _EMBED_FREQ[key] = freq
    arg = diffusion_steps * freq
    return torch.cat((torch.sin(arg), torch.cos(arg)), 1)


def std_normal(size):
    """N(0, I) drawn like the reference: CPU generator, then copied to the GPU (util.py:63-68)."""
    return torch.normal(0, 1, size=size).cuda()


class InferenceSchedule:
    """What sampling_given_noise_schedule derives from (diffusion_hyperparams, inference_noise_schedule) before its
    loop (util.py:182-209): alpha_hat/sigma_hat, the mapped fractional steps, and the per-step scalars."""

    def __init__(self, diffusion_hyperparams, inference_noise_schedule, verbose=True):
        T, alpha = diffusion_hyperparams["T"], diffusion_hyperparams["alpha"]
        assert len(alpha) == T
        alpha = alpha.detach().to("cpu", torch.float32)
        self.beta = inference_noise_schedule.detach().to("cpu", torch.float32).clone()
        self.alpha_hat, self.sigma_hat = _alpha_sigma_recursion(self.beta)
        kept = [s for s in (map_noise_scale_to_time_step(a, alpha) for a in self.alpha_hat) if s >= 0]
        if verbose:
            print(kept, flush=True)
        self.steps = torch.FloatTensor(kept)      # float32, as the reference feeds them to the net (util.py:204,217)
        self.N = len(kept)                         # can be < len(beta) when a level is off the table (util.py:206-207)
        if verbose:
            print('begin sampling, total number of reverse steps = %s' % self.N)

    def rows(self):
        """fd_step rows in EXECUTION order (n = N-1 .. 0); each scalar is the reference's 0-d fp32 expression.  A StepRows: a list of
        read-only mappings that also carries the ctypes table FastDiff.sample hands to fd_sample, built once per schedule."""
        out = StepRows()
        for n in reversed(range(self.N)):
            b, a = self.beta[n], self.alpha_hat[n]
            a_next = a / (1 - b).sqrt()                      # util.py:220
            c1 = a_next / a                                  # :221
            c2 = -(1 - a ** 2.).sqrt() * c1                  # :222
            c3 = (1 - a_next ** 2.).sqrt()                   # :223
            out.append({"t": self.steps[n].item(),
                        "c_eps": (b / torch.sqrt(1 - a ** 2.)).item(),    # :226
                        "c_div": torch.sqrt(1 - b).item(),               # :227
                        "sigma": self.sigma_hat[n].item(),               # :229
                        "c1": c1.item(), "c2": c2.item(), "c3": c3.item(),
                        "add_noise": int(n > 0)})                        # :228
        return out.freeze()


class StepRows(list):
    """The step table of one schedule.  Its rows are read-only (copy them -- `[dict(r) for r in rows]`, a slice -- to edit: the copy
    is a plain list), so the fd_step array derived from them can be kept: one utterance per call (the reference CLI's mode) builds
    it once per schedule instead of once per call."""
    fd_steps = None

    def freeze(self):
        import types
        from . import _capi
        self[:] = [types.MappingProxyType(dict(r)) for r in self]
        self.fd_steps = _capi.step_table(self)
        return self


# ---------------------------------------------------------------------------------------------------------
# the reverse loop
# ---------------------------------------------------------------------------------------------------------
def _apply_row(x, eps, row, ddim, z):
    """One reverse update on torch tensors (used only for denoisers that are not the HIP module)."""
    if ddim:
        return row["c1"] * x + row["c2"] * eps + row["c3"] * eps
    x = (x - row["c_eps"] * eps) / row["c_div"]
    return x + row["sigma"] * z if row["add_noise"] else x


def sampling_given_noise_schedule(net, size, diffusion_hyperparams, inference_noise_schedule, condition=None,
                                  ddim=False, return_sequence=False, *, x_T=None, noise=None, seed=None,
                                  noise_source="device", verbose=True, lens=None, stream_ids=None):
    """x_0 ~ p(x_0|x_T) under a given inference schedule; same positional signature as util.py:158-165.

    `net` a fastdiff_amd.FastDiff: the whole loop runs on the MI355X (one captured denoiser step replayed N times).
    Keyword-only extras: x_T [B,1,L], noise [N,B,1,L] (noise[k] is added after the k-th executed step) inject the
    Gaussian draws; otherwise noise_source="device" draws them with on-device Philox (keyed by `seed`), and
    noise_source="reference" draws them with std_normal in the reference's order, so a seeded run reproduces the
    reference's random stream.  Any other callable `net` is driven step by step from the host with the same tables.
    lens (FastDiff nets only): valid frames per utterance of a zero-padded batch; every utterance is then computed as if alone.
    stream_ids (FastDiff nets only): per-utterance noise streams, see FastDiff.sample.
    Returns the tensor of shape `size`, or the list of N+1 intermediate tensors when return_sequence=True."""
    assert len(size) == 3
    sched = InferenceSchedule(diffusion_hyperparams, inference_noise_schedule, verbose)
    rows, N = sched.rows(), sched.N
    B = size[0]

    if noise_source == "reference" and x_T is None and noise is None:
        x_T = std_normal(size)
        if not ddim:
            noise = torch.stack([std_normal(size) if r["add_noise"] else torch.zeros(size, device=x_T.device) for r in rows])

    if isinstance(net, FastDiff):
        cond = condition if condition.dim() == 3 else condition.unsqueeze(0)
        if cond.shape[0] != B:
            cond = cond.expand(B, -1, -1)
        assert size[2] == cond.shape[-1] * net.hop_length, "length of (x, kernel) is not matched"
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        with torch.no_grad():
            return net.sample(cond.cuda(), rows, ddim=ddim, x_T=x_T, noise=noise, seed=seed,
                              return_sequence=return_sequence, lens=lens, stream_ids=stream_ids)

    # a denoiser this package does not own: drive it from the host
    x = std_normal(size) if x_T is None else x_T.clone()
    trajectory = [x.clone()]
    with torch.no_grad():
        for k, row in enumerate(rows):
            t = torch.full((B, 1), row["t"], dtype=torch.float32, device=x.device)
            eps = net((x, condition, t))
            z = None
            if row["add_noise"] and not ddim:
                z = std_normal(size) if noise is None else noise[k]
            x = _apply_row(x, eps, row, ddim, z)
            trajectory.append(x.clone())
    return trajectory if return_sequence else x


def theta_timestep_loss(net, X, diffusion_hyperparams, reverse=False):
    """MSE(eps_theta(x_t, mel, t), z) at a random training step per item; same signature as util.py:291-325.

    FastDiffTask.validation_step reports this quantity under no_grad (FastDiff.py:52-57): one denoiser evaluation on the inference
    kernels.  _training_step (FastDiff.py:44-49) calls the same function with the module in train() mode and autograd recording:
    FastDiff.forward then builds the graph of fastdiff_amd/train.py (location-variable convolutions forward and backward on the HIP
    operator), and loss.backward() fills the gradients of the module's weight_g / weight_v / bias parameters.
    Random draws follow the reference order: torch.randint for the steps, then std_normal for z (both on the CPU generator)."""
    assert type(X) == tuple and len(X) == 2
    mel_spectrogram, audio = X
    T_train, alpha = diffusion_hyperparams["T"], diffusion_hyperparams["alpha"]
    n_items = audio.shape[0]
    ts = torch.randint(T_train, size=(n_items, 1, 1)).cuda()
    z = std_normal(audio.shape)
    alpha_t = alpha.to(ts.device)[ts]
    delta = (1 - alpha_t ** 2.).sqrt()
    x_t = alpha_t * audio + delta * z                       # a draw of q(x_t | x_0)
    eps = net((x_t, mel_spectrogram, ts.view(n_items, 1)))
    loss = torch.nn.functional.mse_loss(eps, z)
    if reverse:
        return loss, (x_t - delta * eps) / alpha_t
    return loss


def phi_loss(net, X, diffusion_hyperparams):
    """The BDDM loss of the scheduling network (util.py:328-362); same signature.  Needs `net.noise_pred(x_t [B, L], (beta_next
    [B, 1], delta^2 [B, 1]))` -- the network the reference calls but never defines, so with the stock module it
    ends, there as here, in AttributeError after the denoiser evaluation.  The denoiser evaluation itself is FastDiff.forward:
    the inference kernels under no_grad, the autograd graph of fastdiff_amd/train.py otherwise.
    Random draws in the reference's order: torch.randint for the steps, then std_normal for z."""
    assert type(X) == tuple and len(X) == 2
    dh = diffusion_hyperparams
    T_train, alpha, tau = dh["T"], dh["alpha"], dh["tau"]
    mel_spectrogram, audio = X
    n_items = audio.shape[0]
    ts = torch.randint(tau, T_train - tau, size=(n_items,)).cuda()
    alpha = alpha.to(ts.device)
    alpha_cur = alpha.index_select(0, ts).view(n_items, 1, 1)
    alpha_nxt = alpha.index_select(0, ts + tau).view(n_items, 1, 1)
    beta_nxt = 1 - (alpha_nxt / alpha_cur) ** 2.
    delta = (1 - alpha_cur ** 2.).sqrt()
    z = std_normal(audio.shape)
    x_t = alpha_cur * audio + delta * z                     # a draw of q(x_t | x_0)
    eps = net((x_t, mel_spectrogram, ts.view(n_items, 1)))
    beta_est = net.noise_pred(x_t.squeeze(1), (beta_nxt.view(n_items, 1), delta.view(n_items, 1) ** 2.))
    loss = 1 / (2. * (delta ** 2. - beta_est)) * (delta * z - beta_est / delta * eps) ** 2.
    loss = loss + torch.log(1e-8 + delta ** 2. / (beta_est + 1e-8)) / 4.
    return (torch.mean(loss, -1, keepdim=True) + beta_est / delta ** 2 / 2.).mean()


def noise_scheduling(net, size, diffusion_hyperparams, condition=None, ddim=False):
    """Greedy search of an inference schedule with a learned noise predictor; same signature as util.py:237.

    Needs `net.noise_pred(x, (beta_next, 1 - alpha^2))`.  The reference's FastDiff class has no such method
    (SURVEY.md 3.5), so there -- as here -- the call ends in AttributeError right after the first denoiser
    evaluation; with a net that provides it the search proceeds, the denoiser itself running on the HIP path."""
    dh = diffusion_hyperparams
    N, rho, alpha = dh["N"], dh["rho"], dh["alpha"]
    print('begin noise scheduling, maximum number of reverse steps = %d' % (N))
    x = std_normal(size)
    beta_cur = torch.full((1, 1, 1), float(dh["betaN"]), device=x.device)
    alpha_cur = torch.full((1, 1, 1), float(dh["alphaN"]), device=x.device)
    found = []
    with torch.no_grad():
        for _ in range(N):
            step = map_noise_scale_to_time_step(alpha_cur.item(), alpha)
            if step >= 0:
                found.append(beta_cur.item())
            eps = net((x, condition, torch.full((size[0], 1), float(step), device=x.device)))
            if ddim:
                a_next = alpha_cur / (1 - beta_cur).sqrt()
                c1 = a_next / alpha_cur
                x = c1 * x + (-(1 - alpha_cur ** 2.).sqrt() * c1) * eps + (1 - a_next ** 2.).sqrt() * eps
            else:
                x = (x - beta_cur / torch.sqrt(1 - alpha_cur ** 2.) * eps) / torch.sqrt(1 - beta_cur)
            beta_prev = beta_cur
            alpha_cur = alpha_cur / (1 - beta_prev).sqrt()
            if alpha_cur > 1:
                break
            beta_cur = net.noise_pred(x.squeeze(1), (beta_prev.view(-1, 1), (1 - alpha_cur ** 2.).view(-1, 1)))
            if beta_cur.item() < rho:
                break
    return torch.FloatTensor(found[::-1]).cuda()
