"""Generate tests/golden/*.npz by EXECUTING THE REFERENCE ITSELF (TEST INFRASTRUCTURE).

Runs only in the build container, where /root/reference exists:

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

The reference has no tests / golden vectors of its own, so these fixtures are
the pins: every tensor below is an output of the reference's unmodified PyTorch modules
(modules/FastDiff/module/{FastDiff_model,modules,util}.py), fed with the seed-reproducible synthetic
weights and inputs of oracle/synth.py.  The only shim is `torch.Tensor.cuda = identity`, because the
reference hard-codes .cuda() (util.py:68,217,427) and this container has no GPU.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
REF = os.environ.get("FASTDIFF_REFERENCE", "/root/reference")
sys.path.insert(0, HERE)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

torch.Tensor.cuda = lambda self, *a, **k: self   # CPU shim, see docstring

import synth  # noqa: E402
from modules.FastDiff.module.FastDiff_model import FastDiff  # noqa: E402
from modules.FastDiff.module import modules as ref_modules  # noqa: E402
from modules.FastDiff.module import util as ref_util  # noqa: E402

torch.set_num_threads(8)
SEED = 1234

# Inference schedules: modules/FastDiff/task/FastDiff.py:76-91
SCHEDULES = {
    1000: ("linspace", 0.000001, 0.01, 1000),
    200: ("linspace", 0.0001, 0.02, 200),
    8: [6.689325005027058e-07, 1.0033881153503899e-05, 0.00015496854030061513, 0.002387222135439515,
        0.035597629845142365, 0.3681158423423767, 0.4735414385795593, 0.5],
    6: [1.7838445955931093e-06, 2.7984189728158526e-05, 0.00043231004383414984, 0.006634317338466644,
        0.09357017278671265, 0.6000000238418579],
    4: [3.2176e-04, 2.5743e-03, 2.5376e-02, 7.0414e-01],
    3: [9.0000e-05, 9.0000e-03, 6.0000e-01],
}


def schedule_tensor(N):
    s = SCHEDULES[N]
    if isinstance(s, tuple):
        return torch.linspace(s[1], s[2], s[3])
    return torch.FloatTensor(s)


def make_model(dtype=torch.float32, contractive=False):
    m = FastDiff().eval()
    sd = {k: torch.from_numpy(v.copy()) for k, v in synth.synth_state_dict(SEED, contractive=contractive).items()}
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return m.to(dtype)


def gen_schedule():
    out = {}
    dh = ref_util.compute_hyperparams_given_schedule(torch.linspace(1e-6, 0.01, 1000))
    out["train_alpha"] = dh["alpha"].numpy()
    out["train_sigma"] = dh["sigma"].numpy()
    out["train_beta"] = dh["beta"].numpy()
    for N in SCHEDULES:
        beta = schedule_tensor(N)
        out[f"N{N}_beta"] = beta.numpy().copy()
        # exactly the statements of sampling_given_noise_schedule (util.py:187-204)
        beta_infer = beta
        alpha_infer = 1 - beta_infer
        sigma_infer = beta_infer + 0
        for n in range(1, N):
            alpha_infer[n] *= alpha_infer[n - 1]
            sigma_infer[n] *= (1 - alpha_infer[n - 1]) / (1 - alpha_infer[n])
        alpha_infer = torch.sqrt(alpha_infer)
        sigma_infer = torch.sqrt(sigma_infer)
        steps = [ref_util.map_noise_scale_to_time_step(alpha_infer[n], dh["alpha"]) for n in range(N)]
        out[f"N{N}_steps"] = np.array(steps, np.float64)
        out[f"N{N}_alpha_hat"] = alpha_infer.numpy()
        out[f"N{N}_sigma_hat"] = sigma_infer.numpy()
        # per-step coefficients, the expressions of util.py:220-227
        ce, cd, c1s, c2s, c3s = [], [], [], [], []
        for n in range(N):
            ce.append((beta_infer[n] / torch.sqrt(1 - alpha_infer[n] ** 2.)).item())
            cd.append(torch.sqrt(1 - beta_infer[n]).item())
            alpha_next = alpha_infer[n] / (1 - beta_infer[n]).sqrt()
            c1 = alpha_next / alpha_infer[n]
            c2 = -(1 - alpha_infer[n] ** 2.).sqrt() * c1
            c3 = (1 - alpha_next ** 2.).sqrt()
            c1s.append(c1.item()); c2s.append(c2.item()); c3s.append(c3.item())
        for k, v in (("c_eps", ce), ("c_div", cd), ("c1", c1s), ("c2", c2s), ("c3", c3s)):
            out[f"N{N}_{k}"] = np.array(v, np.float32)
    np.savez_compressed(os.path.join(GOLD, "schedule.npz"), **out)
    return dh


def gen_embed():
    half = 64
    table = torch.exp(torch.arange(half) * -(np.log(10000) / (half - 1)))   # util.py:425-427
    steps = torch.tensor([[0.0], [1.0], [7.413235306739807], [74.99228298664093], [498.05368650332093], [999.0]])
    emb = ref_util.calc_diffusion_step_embedding(steps, 128)
    emb64 = ref_util.calc_diffusion_step_embedding(steps.double(), 128)
    np.savez_compressed(os.path.join(GOLD, "embed.npz"), table=table.numpy(), steps=steps.numpy(),
                        emb_f32=emb.numpy(), emb_f64=emb64.numpy())


def gen_ops():
    """Function-level pins: standalone reference modules / methods on small random tensors."""
    out = {}
    u = lambda stream, shape, scale=1.0: (synth.hash_uniform(77, stream, int(np.prod(shape))) * np.float32(scale)).reshape(shape)  # noqa: E731
    with torch.no_grad():
        # location_variable_convolution (modules.py:220-253) at the three hops of the model
        blk = ref_modules.TimeAware_LVCBlock(32, 80, 8)
        for i, (hop, T) in enumerate(((8, 5), (64, 3), (256, 2))):
            x = u(10 + i, (2, 32, T * hop)); k = u(20 + i, (2, 32, 64, 3, T), 0.2); b = u(30 + i, (2, 64, T), 0.5)
            y = blk.location_variable_convolution(torch.from_numpy(x), torch.from_numpy(k), torch.from_numpy(b), 1, hop)
            y64 = blk.location_variable_convolution(torch.from_numpy(x).double(), torch.from_numpy(k).double(),
                                                    torch.from_numpy(b).double(), 1, hop)
            out[f"lvc{hop}_x"], out[f"lvc{hop}_k"], out[f"lvc{hop}_b"] = x, k, b
            out[f"lvc{hop}_y"], out[f"lvc{hop}_y64"] = y.numpy(), y64.numpy()
        # DiffusionDBlock (modules.py:116-138), factors 4 and 8
        for f in (4, 8):
            db = ref_modules.DiffusionDBlock(32, 32, f).eval()
            ws = []
            with torch.no_grad():
                for j, p in enumerate(db.parameters()):
                    w = u(100 + 10 * f + j, tuple(p.shape), 0.12)
                    p.copy_(torch.from_numpy(w))
            named = dict(db.named_parameters())
            ws = [named["residual_dense.weight"], named["residual_dense.bias"]]
            for i in range(3):
                ws += [named[f"conv.{i}.weight"], named[f"conv.{i}.bias"]]
            x = u(200 + f, (2, 32, 40 * f))
            out[f"dblock{f}_x"] = x
            out[f"dblock{f}_y"] = db(torch.from_numpy(x)).numpy()
            out[f"dblock{f}_y64"] = db.double()(torch.from_numpy(x).double()).numpy()
            for j, w in enumerate(ws):
                out[f"dblock{f}_w{j}"] = w.detach().float().numpy()
        # ConvTranspose1d as TimeAware_LVCBlock builds it (modules.py:163-166)
        for r in (8, 4):
            ct = torch.nn.ConvTranspose1d(32, 32, 2 * r, stride=r, padding=r // 2 + r % 2, output_padding=r % 2)
            w = u(300 + r, (32, 32, 2 * r), 0.1); b = u(310 + r, (32,), 0.1)
            ct.weight.copy_(torch.from_numpy(w)); ct.bias.copy_(torch.from_numpy(b))
            x = u(320 + r, (2, 32, 11))
            out[f"convt{r}_x"], out[f"convt{r}_w"], out[f"convt{r}_b"] = x, w, b
            out[f"convt{r}_y"] = ct(torch.from_numpy(x)).numpy()
        # weight-norm fold (FastDiff_model.py:115-122): torch's own _weight_norm
        v = u(400, (64, 80, 5), 0.3); g = np.abs(u(401, (64, 1, 1))) + np.float32(0.5)
        out["wn_v"], out["wn_g"] = v, g
        out["wn_w"] = torch._weight_norm(torch.from_numpy(v), torch.from_numpy(g), 0).numpy()
    np.savez_compressed(os.path.join(GOLD, "ops.npz"), **out)


def gen_forward():
    m32 = make_model(torch.float32)
    m64 = make_model(torch.float64)
    cases = {
        # name: (B, T, steps, mel range, want taps)
        "f1": (1, 4, [7.413235306739807], (-6.0, 1.5), True),
        "f2": (2, 7, [23.467590302228928, 498.05368650332093], (-6.0, 1.5), False),
        "f3": (1, 33, [74.99228298664093], (-6.0, 1.5), False),
        "f4": (1, 5, [3.0], (-11.5, 2.0), False),     # Tacotron-style natural-log mel range (FastDiff_tacotron.yaml)
    }
    for ci, (name, (B, T, steps, (lo, hi), want_taps)) in enumerate(cases.items()):
        mel = synth.synth_mel(SEED + ci, B, T, lo, hi)
        audio = synth.synth_audio(SEED + ci, B, T)
        st = torch.tensor(steps, dtype=torch.float32).view(B, 1)
        out = {"mel": mel, "audio": audio, "steps": st.numpy()}
        taps = {}
        hooks = []
        if want_taps:
            def keep(key):
                def fn(mod, inp, res):
                    if isinstance(res, tuple):
                        taps[key + "_kernels"] = res[0].detach().numpy().copy()
                        taps[key + "_bias"] = res[1].detach().numpy().copy()
                    else:
                        taps[key] = res.detach().numpy().copy()
                return fn
            hooks.append(m32.first_audio_conv.register_forward_hook(keep("a0")))
            for d in range(3):
                hooks.append(m32.downsample[d].register_forward_hook(keep(f"a{d + 1}")))
            for n in range(3):
                hooks.append(m32.lvc_blocks[n].register_forward_hook(keep(f"x{n}")))
                hooks.append(m32.lvc_blocks[n].kernel_predictor.register_forward_hook(keep(f"kp{n}")))
            hooks.append(m32.fc_t2.register_forward_hook(keep("fc_t2_pre")))
        with torch.no_grad():
            y32 = m32((torch.from_numpy(audio), torch.from_numpy(mel), st))
            y64 = m64((torch.from_numpy(audio).double(), torch.from_numpy(mel).double(), st.double()))
        for h in hooks:
            h.remove()
        out["y_f32"] = y32.numpy()
        out["y_f64"] = y64.numpy()
        for k, v in taps.items():
            if k.endswith("_kernels"):
                # [B,4,32,64,3,T] view -> store as the raw conv output [B,24576,T]
                v = v.reshape(v.shape[0], -1, v.shape[-1])
            if k.endswith("_bias"):
                v = v.reshape(v.shape[0], -1, v.shape[-1])
            out["tap_" + k] = v
        np.savez_compressed(os.path.join(GOLD, f"forward_{name}.npz"), **out)
        print(name, "max|y|", float(np.abs(out["y_f32"]).max()), "f32-vs-f64", float(np.abs(out["y_f32"] - out["y_f64"]).max()))


def run_sampler(model, dtype, B, T, N, dh, noises, ddim, return_sequence):
    """Call the reference sampler with std_normal replaced by a replay of `noises` (x_T first, then z_{N-1}..z_1)."""
    it = iter(noises)
    orig = ref_util.std_normal
    # .clone(): the sampler updates x in place (util.py:226-227); never let it alias the recorded noise
    ref_util.std_normal = lambda size: torch.from_numpy(next(it).copy()).to(dtype).view(*size).clone()
    net = model if dtype == torch.float32 else (lambda data: model((data[0], data[1], data[2].double())))
    try:
        devnull = open(os.devnull, "w")
        stdout, sys.stdout = sys.stdout, devnull
        res = ref_util.sampling_given_noise_schedule(net, (B, 1, T * 256), dh, schedule_tensor(N),
                                                     condition=model._mel, ddim=ddim,
                                                     return_sequence=return_sequence)
    finally:
        sys.stdout = stdout
        ref_util.std_normal = orig
    return res


def gen_sample(dh, only=None):
    m32 = make_model(torch.float32)
    m64 = make_model(torch.float64)
    cases = {
        # name: (B, T, N, ddim, return_sequence, run_f64)
        "s1": (2, 6, 4, False, True, True),
        "s2": (1, 5, 4, True, True, True),
        "s3": (1, 5, 6, False, False, True),
        "s4": (1, 4, 1000, False, False, True),
        "s5": (1, 4, 8, False, False, True),
        # BASELINE configs[2] at a length with tile edges (four 256-column tiles per frame row at hop 256, 64 frames): the full
        # N = 1000 schedule; stored: x_0 and the state after every 125 steps (the whole sequence is 65 MB)
        "s6": (1, 64, 1000, False, "every125", True),
    }
    for ci, (name, (B, T, N, ddim, seq, run64)) in enumerate(cases.items()):
        if only and name not in only:
            continue
        mel = synth.synth_mel(SEED + 100 + ci, B, T)
        n_el = B * T * 256
        x_T = synth.hash_normal(SEED + 100 + ci, 1, n_el).reshape(B, 1, T * 256)
        # z[n] is the noise added after step n (n>0); stored [N,B,1,L], z[0] unused (zeros)
        z = np.zeros((N, B, 1, T * 256), np.float32)
        for n in range(1, N):
            z[n] = synth.hash_normal(SEED + 100 + ci, 2 + n, n_el).reshape(B, 1, T * 256)
        noises = [x_T] + [z[n] for n in range(N - 1, 0, -1)]
        out = {"mel": mel, "x_T": x_T, "N": np.int64(N), "ddim": np.int64(ddim), "seed": np.int64(SEED + 100 + ci)}
        for tag, model, dt in (("f32", m32, torch.float32), ("f64", m64, torch.float64)):
            if tag == "f64" and not run64:
                continue
            model._mel = torch.from_numpy(mel).to(dt)
            dh_t = {"T": dh["T"], "alpha": dh["alpha"], "beta": dh["beta"], "sigma": dh["sigma"]}
            res = run_sampler(model, dt, B, T, N, dh_t, noises, ddim, seq)
            if seq == "every125":
                out["ckpt_idx"] = np.arange(0, N + 1, 125)
                out[f"ckpt_{tag}"] = np.stack([res[k].numpy() for k in range(0, N + 1, 125)])
                out[f"y_{tag}"] = res[N].numpy()
            elif seq:
                out[f"seq_{tag}"] = np.stack([r.numpy() for r in res])
            else:
                out[f"y_{tag}"] = res.numpy()
        np.savez_compressed(os.path.join(GOLD, f"sample_{name}.npz"), **out)
        key = "seq_f32" if seq is True else "y_f32"
        k64 = "seq_f64" if seq is True else "y_f64"
        print(name, "max|y|", float(np.abs(out[key]).max()), "f32-vs-f64", float(np.abs(out[key] - out[k64]).max()) if k64 in out else None)


def gen_noise_scheduling(dh):
    """util.py:237-288 run on the reference module with synth.stub_noise_pred attached as `noise_pred` (the reference ships no such
    network) and std_normal replaying a recorded x_T: the schedule it finds, for the DDPM and the "ddim" update."""
    B, T, N = 1, 5, 8
    mel = synth.synth_mel(SEED + 300, B, T)
    x_T = synth.hash_normal(SEED + 300, 1, B * T * 256).reshape(B, 1, T * 256)
    out = {"mel": mel, "x_T": x_T, "N": np.int64(N), "betaN": np.float64(0.5), "alphaN": np.float64(0.2), "rho": np.float64(1e-3)}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        model = make_model(dt)
        model.noise_pred = synth.stub_noise_pred
        fwd = model.forward
        if dt == torch.float64:
            model.forward = lambda data: fwd((data[0], data[1].double(), data[2].double()))
        for ddim in (False, True):
            orig = ref_util.std_normal
            ref_util.std_normal = lambda size: torch.from_numpy(x_T.copy()).to(dt).view(*size).clone()
            stdout, sys.stdout = sys.stdout, open(os.devnull, "w")
            try:
                betas = ref_util.noise_scheduling(model, (B, 1, T * 256), {"N": N, "betaN": 0.5, "alphaN": 0.2, "rho": 1e-3, "alpha": dh["alpha"]},
                                                  condition=torch.from_numpy(mel).to(dt), ddim=ddim)
            finally:
                sys.stdout = stdout
                ref_util.std_normal = orig
            out[f"betas_{'ddim' if ddim else 'ddpm'}_{tag}"] = betas.double().numpy()
    np.savez_compressed(os.path.join(GOLD, "noise_scheduling.npz"), **out)
    for k in sorted(out):
        if k.startswith("betas"):
            print(k, out[k])


def gen_theta_loss(dh):
    """util.py:291-325 on the reference module (what validation_step reports, FastDiff.py:52-57): the random steps and z are
    recorded (torch.randint / std_normal replaced by replays), loss and the x_0 estimate of reverse=True stored."""
    B, T = 3, 6
    mel = synth.synth_mel(SEED + 400, B, T)
    audio = (0.3 * synth.hash_normal(SEED + 400, 1, B * T * 256)).reshape(B, 1, T * 256).astype(np.float32)
    z = synth.hash_normal(SEED + 400, 2, B * T * 256).reshape(B, 1, T * 256)
    ts = np.array([0, 437, 999], np.int64).reshape(B, 1, 1)
    out = {"mel": mel, "audio": audio, "z": z, "ts": ts}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        model = make_model(dt)
        net = model if dt == torch.float32 else (lambda data: model((data[0], data[1], data[2].double())))
        orig_n, orig_r = ref_util.std_normal, torch.randint
        ref_util.std_normal = lambda size: torch.from_numpy(z.copy()).to(dt).view(*size)
        torch.randint = lambda *a, **k: torch.from_numpy(ts.copy())
        try:
            with torch.no_grad():
                loss, x0 = ref_util.theta_timestep_loss(net, (torch.from_numpy(mel).to(dt), torch.from_numpy(audio).to(dt)),
                                                        {"T": dh["T"], "alpha": dh["alpha"].to(dt)}, reverse=True)
        finally:
            ref_util.std_normal, torch.randint = orig_n, orig_r
        out[f"loss_{tag}"] = np.float64(loss.item())
        out[f"x0_{tag}"] = x0.double().numpy()
        print("theta_loss", tag, loss.item(), float(x0.abs().max()))
    np.savez_compressed(os.path.join(GOLD, "theta_loss.npz"), **out)


def gen_phi_loss(dh):
    """util.py:328-362 on the reference module with synth.stub_noise_pred_batch attached as `noise_pred` (the reference ships no such
    network); random steps and z replayed; the loss for float32 and float64."""
    B, T, tau = 3, 6, 50
    mel = synth.synth_mel(SEED + 600, B, T)
    audio = (0.3 * synth.hash_normal(SEED + 600, 1, B * T * 256)).reshape(B, 1, T * 256).astype(np.float32)
    z = synth.hash_normal(SEED + 600, 2, B * T * 256).reshape(B, 1, T * 256)
    ts = np.array([50, 437, 949], np.int64)
    out = {"mel": mel, "audio": audio, "z": z, "ts": ts, "tau": np.int64(tau)}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        model = make_model(dt)
        model.noise_pred = synth.stub_noise_pred_batch
        fwd = model.forward
        if dt == torch.float64:
            model.forward = lambda data: fwd((data[0], data[1], data[2].double()))
        orig_n, orig_r = ref_util.std_normal, torch.randint
        ref_util.std_normal = lambda size: torch.from_numpy(z.copy()).to(dt).view(*size)
        torch.randint = lambda *a, **k: torch.from_numpy(ts.copy())
        try:
            with torch.no_grad():
                loss = ref_util.phi_loss(model, (torch.from_numpy(mel).to(dt), torch.from_numpy(audio).to(dt)),
                                         {"T": dh["T"], "alpha": dh["alpha"].to(dt), "tau": tau})
        finally:
            ref_util.std_normal, torch.randint = orig_n, orig_r
        out[f"loss_{tag}"] = np.float64(loss.item())
        print("phi_loss", tag, loss.item())
    np.savez_compressed(os.path.join(GOLD, "phi_loss.npz"), **out)


def grad_sample(t, n=64):
    """What the theta_grad fixture keeps of a gradient tensor: its L2 norm and n evenly spaced elements of the flattened tensor."""
    flat = t.detach().double().reshape(-1)
    step = max(1, flat.numel() // n)
    return float(flat.norm()), flat[::step][:n].numpy().copy()


def gen_theta_grad(dh):
    """The training step's backward (FastDiff.py:44-49): util.py:291-325 on the reference module in train() mode with autograd
    recording, loss.backward(), for float64 and float32.  Kept: the loss, d loss / d audio in full, and of every parameter's
    gradient (175 tensors, the reference's weight_g / weight_v parametrisation) the L2 norm and 64 evenly spaced elements."""
    B, T = 2, 6
    mel = synth.synth_mel(SEED + 500, B, T)
    audio = (0.3 * synth.hash_normal(SEED + 500, 1, B * T * 256)).reshape(B, 1, T * 256).astype(np.float32)
    z = synth.hash_normal(SEED + 500, 2, B * T * 256).reshape(B, 1, T * 256)
    ts = np.array([437, 12], np.int64).reshape(B, 1, 1)
    out = {"mel": mel, "audio": audio, "z": z, "ts": ts}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        model = make_model(dt).train()
        net = model if dt == torch.float32 else (lambda data: model((data[0], data[1], data[2].double())))
        orig_n, orig_r = ref_util.std_normal, torch.randint
        ref_util.std_normal = lambda size: torch.from_numpy(z.copy()).to(dt).view(*size)
        torch.randint = lambda *a, **k: torch.from_numpy(ts.copy())
        a = torch.from_numpy(audio).to(dt).requires_grad_(True)
        try:
            loss = ref_util.theta_timestep_loss(net, (torch.from_numpy(mel).to(dt), a), {"T": dh["T"], "alpha": dh["alpha"].to(dt)})
        finally:
            ref_util.std_normal, torch.randint = orig_n, orig_r
        loss.backward()
        out[f"loss_{tag}"] = np.float64(loss.item())
        out[f"daudio_{tag}"] = a.grad.double().numpy()
        names = []
        for name, p in model.named_parameters():
            assert p.grad is not None, name
            nrm, smp = grad_sample(p.grad)
            out[f"{tag}_norm/{name}"] = np.float64(nrm)
            out[f"{tag}_sample/{name}"] = smp
            names.append(name)
        out["names"] = np.array(names)
        print("theta_grad", tag, loss.item(), len(names), "parameters; |d audio| max", float(a.grad.abs().max()))
    np.savez_compressed(os.path.join(GOLD, "theta_grad.npz"), **out)


def gen_collate():
    """collate_2d (utils/__init__.py:136-150) cannot be imported (the package needs chardet): its definition is cut out of the
    reference file with ast and executed as is."""
    import ast
    src = open(os.path.join(REF, "utils", "__init__.py")).read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "collate_2d")
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "collate_2d", "exec"), ns)
    g = torch.Generator().manual_seed(77)
    lens = [5, 9, 3, 9, 1]
    mels = [torch.rand(t, 80, generator=g) * 13.5 - 11.5 for t in lens]          # Tacotron-range mels, [T, 80] as on disk
    out = ns["collate_2d"](mels, 0).transpose(2, 1)                              # what the collater hands to the model
    arrays = {f"mel{i}": m.numpy() for i, m in enumerate(mels)}
    np.savez_compressed(os.path.join(GOLD, "collate.npz"), batch=out.contiguous().numpy(), lens=np.array(lens), **arrays)
    print("collate", tuple(out.shape))


def gen_frontend():
    """Realistic-audio fixture for the mel front-end: the shortest of the reference's sample recordings (egs/audios, LJSpeech,
    public domain) as int16 PCM, with the log-mel the restated front-end (oracle/mel_frontend.py) gives for it."""
    from scipy.io import wavfile
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mel_frontend as mf
    sr, pcm = wavfile.read(os.path.join(REF, "egs", "audios", "LJ001-0002_gt.wav"))
    assert sr == mf.SR and pcm.dtype == np.int16
    wav = pcm.astype(np.float64) / 32768.0                   # librosa.core.load scaling
    mel = mf.log_mel(wav)
    np.savez_compressed(os.path.join(GOLD, "frontend_lj001_0002.npz"), pcm=pcm, mel_f64=mel)
    print("frontend", pcm.shape, mel.shape, float(mel.min()), float(mel.max()))


def gen_frontend_tacotron():
    """The same recording through the reference's OWN Tacotron front-end classes (data_gen/tts/tacotron/{layers,stft}.py), executed
    here.  librosa is absent, so a stand-in module supplies the three names those files import from it: pad_center (identity for
    win == n_fft), tiny (unused by transform) and filters.mel (the restated filter bank of oracle/mel_frontend.py)."""
    import types
    from scipy.io import wavfile
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mel_frontend as mf
    lib = types.ModuleType("librosa"); lib.util = types.ModuleType("librosa.util"); lib.filters = types.ModuleType("librosa.filters")
    lib.util.pad_center = lambda w, size, **kw: (w if len(w) == size else np.pad(w, ((size - len(w)) // 2, size - len(w) - (size - len(w)) // 2)))
    lib.util.tiny = lambda x: np.finfo(np.float32).tiny
    lib.util.normalize = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError)
    lib.filters.mel = lambda sr, n_fft, n_mels, fmin, fmax: mf.mel_basis(sr, n_fft, n_mels, fmin, fmax).astype(np.float32)
    for name, mod in (("librosa", lib), ("librosa.util", lib.util), ("librosa.filters", lib.filters)):
        sys.modules.setdefault(name, mod)
    from data_gen.tts.tacotron.layers import TacotronSTFT
    sr, pcm = wavfile.read(os.path.join(REF, "egs", "audios", "LJ001-0002_gt.wav"))
    stft = TacotronSTFT(1024, 256, 1024, 80, 22050, 0.0, 8000.0)
    audio_norm = torch.from_numpy(pcm.astype(np.float32)) / 32768.0              # vocoder_binarizer_tacotron.py:113
    with torch.no_grad():
        mel = stft.mel_spectrogram(audio_norm.unsqueeze(0))[0].numpy()
    ours = mf.tacotron_log_mel(pcm.astype(np.float64) / 32768.0)
    print("frontend_tacotron", mel.shape, float(mel.min()), float(mel.max()), "restatement max|diff|", float(np.abs(np.exp(mel) - np.exp(ours)).max()))
    np.savez_compressed(os.path.join(GOLD, "frontend_tacotron_lj001_0002.npz"), mel_ref_f32=mel)


def _ast_pick(path, *, functions=(), klass=None, methods=()):
    """Definitions cut out of a reference file that cannot be imported here (absent third-party modules at its top)."""
    import ast
    tree = ast.parse(open(path).read())
    out = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in functions]
    if klass:
        c = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == klass)
        body = [n for n in c.body if isinstance(n, ast.FunctionDef) and n.name in methods]
        out.append(ast.ClassDef(name=klass, bases=[], keywords=[], body=body, decorator_list=[]))
    return ast.fix_missing_locations(ast.Module(body=out, type_ignores=[]))


def gen_collater():
    """The reference's OWN test-time path from a mel directory to the batch the model sees, executed here:
    VocoderDataset.load_mel_inputs -> __getitem__ -> collater (tasks/vocoder/dataset_utils.py:186-204, 80-98, 100-160) with
    VocoderBinarizer.process_mel_item (data_gen/tts/vocoder_binarizer.py:115-122) and utils.collate_2d.  Neither file imports here
    (resemblyzer, chardet, tensorboard, librosa are absent), so the method bodies are cut out with ast and run unmodified against
    stand-ins for what they reach: `hparams` (use_wav False, test-time values), `utils.collate_2d` (the reference's own, also
    cut out), a registered module holding the binarizer class."""
    import glob as _glob, importlib, tempfile, types
    ns_u = {"torch": torch}
    exec(compile(_ast_pick(os.path.join(REF, "utils", "__init__.py"), functions=("collate_1d", "collate_2d")), "utils", "exec"), ns_u)
    utils_stub = types.SimpleNamespace(collate_2d=ns_u["collate_2d"], collate_1d=ns_u["collate_1d"])
    ns_b = {"np": np}
    exec(compile(_ast_pick(os.path.join(REF, "data_gen", "tts", "vocoder_binarizer.py"), klass="VocoderBinarizer", methods=("process_mel_item",)),
                 "vocoder_binarizer", "exec"), ns_b)
    mod = types.ModuleType("fd_ref_binarizer")
    mod.VocoderBinarizer = ns_b["VocoderBinarizer"]
    sys.modules["fd_ref_binarizer"] = mod
    hp = {"use_wav": False, "binarizer_cls": "fd_ref_binarizer.VocoderBinarizer", "binarization_args": {}, "use_spk_embed": False,
          "use_emo_embed": False}
    ns_d = {"np": np, "torch": torch, "glob": _glob, "importlib": importlib, "utils": utils_stub, "hparams": hp}
    exec(compile(_ast_pick(os.path.join(REF, "tasks", "vocoder", "dataset_utils.py"), klass="VocoderDataset",
                           methods=("load_mel_inputs", "collater", "__getitem__", "_get_item")), "dataset_utils", "exec"), ns_d)
    ds = ns_d["VocoderDataset"].__new__(ns_d["VocoderDataset"])
    # the attributes VocoderDataset.__init__ sets for prefix == 'test' (dataset_utils.py:50-58; base.yaml: aux_context_window 0)
    ds.hparams, ds.batch_max_frames, ds.aux_context_window, ds.hop_size = hp, 0, 0, 256
    g = torch.Generator().manual_seed(78)
    # sorted(glob('*.npy')) sees only the top level.  (A one-frame mel is not in the set: `.squeeze(0)` at dataset_utils.py:110
    # turns [1, 80] into [80] and collate_2d then fails its numel assert -- the reference cannot collate it.)
    files = {"b/x.npy": 7, "a.npy": 5, "c.npy": 12, "sub_dir.npy": 2}
    arrays = {}
    with tempfile.TemporaryDirectory() as d:
        for name, t in files.items():
            os.makedirs(os.path.dirname(os.path.join(d, name)), exist_ok=True)
            m = (torch.rand(t, 80, generator=g) * 13.5 - 11.5).numpy()
            np.save(os.path.join(d, name), m)
            arrays["in_" + name.replace("/", "__")] = m
        ds.indexed_ds, ds.sizes = ds.load_mel_inputs(d)
        ds.avail_idxs = list(range(len(ds.sizes)))
        samples = [ds[i] for i in range(len(ds.sizes))]
    np.random.seed(0)
    batch = ds.collater(samples)
    assert batch["wavs"] == [] and batch["z"] == []
    mels = batch["mels"].contiguous().numpy()
    lens = [s["mel"].shape[0] for s in samples]
    print("collater", mels.shape, batch["item_name"], "sizes", ds.sizes)
    np.savez_compressed(os.path.join(GOLD, "collater.npz"), mels=mels, item_names=np.array(batch["item_name"]), sizes=np.array(ds.sizes),
                        in_lens=np.array(lens), file_names=np.array(list(files)), **arrays)


def gen_frontend_pwg():
    """The reference's OWN process_utterance (data_gen/tts/data_gen_utils.py:93-147, vocoder='pwg'), cut out with ast (the file's
    imports need parselmouth, webrtcvad, skimage, pyloudnorm, librosa) and executed on the sample recording.  What it reaches is
    supplied by stand-ins: `audio.librosa_pad_lr` is the reference's own (utils/audio.py:67-76, cut out too); `librosa.stft` is
    torch.stft with the arguments librosa documents for this call (centered, zero padding, periodic Hann, onesided) -- an
    independent implementation of the same definition; `librosa.filters.mel` returns the restated filter bank of
    oracle/mel_frontend.py: THE FILTER VALUES REMAIN RESTATED, everything around them (magnitude, the matrix product, the log10
    clamp at 1e-6, the frame count, the padding and trimming of the returned wav) is the reference's code."""
    import types
    from scipy.io import wavfile
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mel_frontend as mf

    def stft(y, n_fft, hop_length, win_length, window, pad_mode):
        assert window == "hann" and pad_mode == "constant" and win_length == n_fft
        w = torch.hann_window(win_length, periodic=True, dtype=torch.float64).to(torch.from_numpy(np.asarray(y)).dtype)
        X = torch.stft(torch.from_numpy(np.asarray(y)), n_fft, hop_length, win_length, w, center=True, pad_mode="constant",
                       onesided=True, return_complex=True)
        return X.numpy()

    lib = types.SimpleNamespace(stft=stft, filters=types.SimpleNamespace(
        mel=lambda sr, n_fft, n_mels, fmin, fmax: mf.mel_basis(sr, n_fft, n_mels, fmin, fmax).astype(np.float32)))
    ns_a = {"np": np}
    exec(compile(_ast_pick(os.path.join(REF, "utils", "audio.py"), functions=("librosa_pad_lr",)), "audio", "exec"), ns_a)
    ns = {"np": np, "librosa": lib, "audio": types.SimpleNamespace(librosa_pad_lr=ns_a["librosa_pad_lr"])}
    exec(compile(_ast_pick(os.path.join(REF, "data_gen", "tts", "data_gen_utils.py"), functions=("process_utterance",)), "data_gen_utils", "exec"), ns)
    sr, pcm = wavfile.read(os.path.join(REF, "egs", "audios", "LJ001-0002_gt.wav"))
    wav32 = (pcm.astype(np.float32) / 32768.0)                  # what librosa.core.load returns: float32 in [-1, 1)
    out = {}
    for tag, wav in (("f32", wav32), ("f64", wav32.astype(np.float64))):
        # the keyword values of PWG.wav2spec (vocoders/pwg.py:107-123) under base.yaml:4-16
        wav_out, mel = ns["process_utterance"](wav, fft_size=1024, hop_size=256, win_length=1024, num_mels=80, fmin=80, fmax=7600,
                                               sample_rate=22050, loud_norm=False, min_level_db=-100, return_linear=False, vocoder="pwg")
        out["mel_ref_" + tag] = mel
        out["wav_len_" + tag] = np.int64(len(wav_out))
    ours = mf.log_mel(wav32.astype(np.float64))
    print("frontend_pwg", out["mel_ref_f64"].shape, "restatement max|d log10 mel| =", float(np.abs(out["mel_ref_f64"] - ours).max()),
          " f32 run:", float(np.abs(out["mel_ref_f32"] - ours).max()))
    np.savez_compressed(os.path.join(GOLD, "frontend_pwg_lj001_0002.npz"), **out)


def gen_lvc_grad():
    """TimeAware_LVCBlock.location_variable_convolution (modules.py:220-253) executed on the reference module in float64, forward and
    -- through torch.autograd -- backward, for the three hop sizes of the model on small tensors (Cin 32, Cout 64, ks 3)."""
    blk = ref_modules.TimeAware_LVCBlock(in_channels=32, cond_channels=80, upsample_ratio=8, conv_layers=4, conv_kernel_size=3,
                                         cond_hop_length=8, kpnet_hidden_channels=64, kpnet_conv_size=3, kpnet_dropout=0.0,
                                         noise_scale_embed_dim_out=512)
    out = {}
    for hop, T, B in ((8, 5, 2), (64, 3, 2), (256, 2, 1)):
        L = T * hop
        x = synth.hash_normal(SEED + 500 + hop, 1, B * 32 * L).reshape(B, 32, L).astype(np.float64)
        K = (0.1 * synth.hash_normal(SEED + 500 + hop, 2, B * 32 * 64 * 3 * T)).reshape(B, 32, 64, 3, T).astype(np.float64)
        bias = (0.1 * synth.hash_normal(SEED + 500 + hop, 3, B * 64 * T)).reshape(B, 64, T).astype(np.float64)
        dout = synth.hash_normal(SEED + 500 + hop, 4, B * 64 * L).reshape(B, 64, L).astype(np.float64)
        xt, Kt, bt = (torch.from_numpy(a.copy()).requires_grad_(True) for a in (x, K, bias))
        y = blk.location_variable_convolution(xt, Kt, bt, 1, hop)
        y.backward(torch.from_numpy(dout))
        tag = f"h{hop}_"
        out.update({tag + "x": x.astype(np.float32), tag + "K": K.astype(np.float32), tag + "bias": bias.astype(np.float32),
                    tag + "dout": dout.astype(np.float32)})
        # gradients of the float32-rounded inputs, in float64 (what the fp32 kernels are compared with)
        xt, Kt, bt = (torch.from_numpy(out[tag + n].astype(np.float64)).requires_grad_(True) for n in ("x", "K", "bias"))
        y = blk.location_variable_convolution(xt, Kt, bt, 1, hop)
        y.backward(torch.from_numpy(out[tag + "dout"].astype(np.float64)))
        out.update({tag + "out": y.detach().numpy().astype(np.float32), tag + "dx": xt.grad.numpy().astype(np.float32),
                    tag + "dK": Kt.grad.numpy().astype(np.float32), tag + "dbias": bt.grad.numpy().astype(np.float32)})      # (computed in float64)
        print("lvc_grad hop", hop, y.shape, float(np.abs(out[tag + "dK"]).max()))
    np.savez_compressed(os.path.join(GOLD, "lvc_grad.npz"), **out)


def gen_test_step(dh):
    """The reference's OWN caller of the hot path, executed here: FastDiffTask.test_step (modules/FastDiff/task/FastDiff.py:60-119)
    cut out with ast (the file's imports -- utils, tasks.vocoder.vocoder_base, utils.audio -- need chardet, tensorboard, resemblyzer,
    librosa, all absent) and run unmodified against stand-ins for what it reaches: `hparams` (N, noise_schedule '', hop_size,
    work_dir, gen_dir_name, audio_sample_rate), `audio.save_wav` = the reference's own save_wav, also cut out of utils/audio.py:11-16,
    `sampling_given_noise_schedule` = the reference's (util.py:158-235) with std_normal replaying recorded noise, `self` = an object
    with the reference model, the build_model hyper-parameters and trainer.global_step.  Stored: what save_wav wrote (read back from
    the RIFF file) for N = 4, N = 6 and an explicit hparams['noise_schedule'] list, plus the NotImplementedError of an unknown N."""
    import tempfile, types
    from scipy.io import wavfile
    ns_a = {"np": np, "wavfile": wavfile}
    exec(compile(_ast_pick(os.path.join(REF, "utils", "audio.py"), functions=("save_wav",)), "audio", "exec"), ns_a)
    hp = {}
    ns = {"os": os, "torch": torch, "hparams": hp, "audio": types.SimpleNamespace(save_wav=ns_a["save_wav"]),
          "sampling_given_noise_schedule": ref_util.sampling_given_noise_schedule}
    exec(compile(_ast_pick(os.path.join(REF, "modules", "FastDiff", "task", "FastDiff.py"), klass="FastDiffTask", methods=("test_step",)),
                 "FastDiffTask", "exec"), ns)
    task = ns["FastDiffTask"].__new__(ns["FastDiffTask"])
    task.model = make_model(torch.float32)
    task.diffusion_hyperparams = {"T": dh["T"], "alpha": dh["alpha"], "beta": dh["beta"], "sigma": dh["sigma"]}
    task.trainer = types.SimpleNamespace(global_step=160000)
    T, seed = 12, SEED + 300
    mel = synth.synth_mel(seed, 1, T)
    L = T * 256
    out = {"mel": mel, "seed": np.int64(seed), "item_name": np.array(["utt_a.npy"])}
    cases = {"N4": {"N": 4, "noise_schedule": ""}, "N6": {"N": 6, "noise_schedule": ""},
             "list3": {"N": 4, "noise_schedule": [9.0000e-05, 9.0000e-03, 6.0000e-01]}}      # a list overrides N (FastDiff.py:65-68)
    with tempfile.TemporaryDirectory() as d:
        for name, c in cases.items():
            hp.clear()
            hp.update({"N": c["N"], "noise_schedule": c["noise_schedule"], "hop_size": 256, "work_dir": d, "gen_dir_name": name,
                       "audio_sample_rate": 22050})
            n_draws = len(c["noise_schedule"]) if c["noise_schedule"] != "" else c["N"]
            noises = [synth.hash_normal(seed, 1, L).reshape(1, 1, L)] + [synth.hash_normal(seed, 2 + n, L).reshape(1, 1, L) for n in range(n_draws - 1, 0, -1)]
            it = iter(noises)
            orig = ref_util.std_normal
            ref_util.std_normal = lambda size: torch.from_numpy(next(it).copy()).view(*size).clone()
            try:
                devnull = open(os.devnull, "w")
                stdout, sys.stdout = sys.stdout, devnull
                ret = task.test_step({"mels": torch.from_numpy(mel), "wavs": [], "item_name": ["utt_a.npy"]}, 0)
            finally:
                sys.stdout = stdout
                ref_util.std_normal = orig
            assert ret == {}
            path = os.path.join(d, f"generated_160000_{name}", "utt_a.npy_pred.wav")
            sr, pcm = wavfile.read(path)
            assert sr == 22050 and pcm.dtype == np.int16 and pcm.shape == (L,)
            out["pcm_" + name] = pcm
            out["n_draws_" + name] = np.int64(n_draws)
            print("test_step", name, "peak", int(np.abs(pcm.astype(np.int32)).max()), "first", pcm[:4])
        hp.update({"N": 5, "noise_schedule": ""})
        try:
            task.test_step({"mels": torch.from_numpy(mel), "wavs": [], "item_name": ["utt_a.npy"]}, 0)
            raised = ""
        except NotImplementedError as e:
            raised = type(e).__name__
        out["unknown_N_raises"] = np.array([raised])
    np.savez_compressed(os.path.join(GOLD, "test_step.npz"), **out)


# Configurations other than base.yaml's that the reference constructor accepts (FastDiff_model.py:13-26): the product runs them on
# its runtime-shaped kernels (fastdiff_amd/csrc/fd_generic.hip); these fixtures are the reference's own outputs for them.
OTHER_CFGS = {
    "cfgA": dict(inner_channels=16, upsample_ratios=[4, 4, 4]),
    "cfgB": dict(inner_channels=8, cond_channels=40, upsample_ratios=[2, 5, 3], lvc_layers_each_block=3, lvc_kernel_size=5,
                 kpnet_hidden_channels=32, kpnet_conv_size=5, diffusion_step_embed_dim_in=64, diffusion_step_embed_dim_mid=256,
                 diffusion_step_embed_dim_out=128),
    "cfgC": dict(upsample_ratios=[16, 16], lvc_layers_each_block=2, kpnet_hidden_channels=48),
}


def gen_forward_cfg(dh):
    """Per configuration: one forward (B = 2, fractional steps) and one N = 4 reverse loop with replayed noise, float32 and float64,
    on FastDiff(**cfg) with the hash weights of synth.synth_state_dict(seed, cfg)."""
    import json
    for ci, (name, cfg) in enumerate(OTHER_CFGS.items()):
        full = synth.full_cfg(cfg)
        hop = int(np.prod(full["upsample_ratios"]))
        sd = {k: torch.from_numpy(v.copy()) for k, v in synth.synth_state_dict(SEED + 7, cfg).items()}
        models = {}
        for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
            m = FastDiff(**cfg).eval()
            missing, unexpected = m.load_state_dict(sd, strict=True)
            assert not missing and not unexpected
            models[tag] = m.to(dt)
        B, T, N = 2, 5, 4
        mel = synth.synth_mel(SEED + 300 + ci, B, T, cond=full["cond_channels"])
        audio = synth.synth_audio(SEED + 300 + ci, B, T, hop=hop)
        st = torch.tensor([7.413235306739807, 498.05368650332093], dtype=torch.float32).view(B, 1)
        out = {"cfg": np.array(json.dumps(cfg)), "seed": np.int64(SEED + 7), "mel": mel, "audio": audio, "steps": st.numpy(), "hop": np.int64(hop)}
        with torch.no_grad():
            out["y_f32"] = models["f32"]((torch.from_numpy(audio), torch.from_numpy(mel), st)).numpy()
            out["y_f64"] = models["f64"]((torch.from_numpy(audio).double(), torch.from_numpy(mel).double(), st.double())).numpy()
        n_el = B * T * hop
        x_T = synth.hash_normal(SEED + 300 + ci, 1, n_el).reshape(B, 1, T * hop)
        z = np.zeros((N, B, 1, T * hop), np.float32)
        for n in range(1, N):
            z[n] = synth.hash_normal(SEED + 300 + ci, 2 + n, n_el).reshape(B, 1, T * hop)
        noises = [x_T] + [z[n] for n in range(N - 1, 0, -1)]
        out["x_T"], out["z"] = x_T, z          # z[n] is added after step n (n > 0)
        for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
            model = models[tag]
            it = iter(noises)
            orig = ref_util.std_normal
            ref_util.std_normal = lambda size: torch.from_numpy(next(it).copy()).to(dt).view(*size).clone()
            net = model if dt == torch.float32 else (lambda data: model((data[0], data[1], data[2].double())))
            try:
                stdout, sys.stdout = sys.stdout, open(os.devnull, "w")
                res = ref_util.sampling_given_noise_schedule(net, (B, 1, T * hop), {k: dh[k] for k in ("T", "alpha", "beta", "sigma")}, schedule_tensor(N),
                                                             condition=torch.from_numpy(mel).to(dt), ddim=False, return_sequence=True)
            finally:
                sys.stdout = stdout
                ref_util.std_normal = orig
            out[f"seq_{tag}"] = np.stack([r.numpy() for r in res])
        np.savez_compressed(os.path.join(GOLD, f"forward_{name}.npz"), **out)
        print(name, cfg, "hop", hop, "max|y|", float(np.abs(out["y_f32"]).max()), "f32-vs-f64 forward", float(np.abs(out["y_f32"] - out["y_f64"]).max()),
              "loop", float(np.abs(out["seq_f32"] - out["seq_f64"]).max()))


def gen_statedict_manifest():
    """Key set + shapes of the reference module's state_dict, and a default-init digest, for the drop-in shim test."""
    torch.manual_seed(SEED)
    m = FastDiff()
    sd = m.state_dict()
    names = np.array(list(sd.keys()))
    shapes = np.array([",".join(map(str, v.shape)) for v in sd.values()])
    # a cheap digest of the default init: per-tensor sum and abs-sum
    sums = np.array([float(v.double().sum()) for v in sd.values()])
    asums = np.array([float(v.double().abs().sum()) for v in sd.values()])
    np.savez_compressed(os.path.join(GOLD, "state_dict_manifest.npz"), names=names, shapes=shapes, sums=sums, asums=asums)

def gen_mel_bank_third_party():
    """The mel filter banks of the two front-ends as a THIRD-PARTY implementation of librosa.filters.mel computes them: Hugging Face
    transformers' `audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney")` (present in the build image; librosa, which the reference
    calls at data_gen/tts/data_gen_utils.py:122-134 and tacotron/layers.py:42-60, is not).  transformers documents and tests that function as
    matching librosa's; it shares no code with this repo's restatement (oracle/mel_frontend.py, fd_api_ext.cpp: default_mel_bank).  Stored
    as float64 [80, 513], librosa's own layout."""
    import transformers
    from transformers.audio_utils import mel_filter_bank
    out = {"transformers_version": np.array(transformers.__version__)}
    for name, fmin, fmax in (("pwg", 80.0, 7600.0), ("tacotron", 0.0, 8000.0)):
        fb = mel_filter_bank(num_frequency_bins=513, num_mel_filters=80, min_frequency=fmin, max_frequency=fmax, sampling_rate=22050,
                             norm="slaney", mel_scale="slaney")
        out[name] = np.ascontiguousarray(np.asarray(fb, np.float64).T)
    np.savez_compressed(os.path.join(GOLD, "mel_bank_third_party.npz"), **out)
    print("mel banks by transformers", transformers.__version__, {k: v.shape for k, v in out.items() if k != "transformers_version"})


def gen_sample_long(dh):
    """BASELINE configs[2] at its own size (round 6): B = 1, T = 864, the full N = 1000 schedule (FastDiff.py:76-78) through the
    reference's sampler (util.py:158-235) in float32 and float64, on the CONTRACTIVE synthetic weights (synth.make_contractive: x stays
    O(1), as with a trained model).  The injected noise is synth.hash_normal(seed, 1, .) for x_T and (seed, 2 + n, .) after reverse
    index n, drawn step by step (the whole [1000, L] array is 885 MB; the GPU test rebuilds it with the torch twin of the hash).
    Stored: the mel, x_0 in full (float64 and float32), every 125th state at every 7th sample (float64 run, kept as float32) and the
    reference's own float32-vs-float64 distance at those states (over all samples).  ~25 minutes on 8 cores."""
    import time
    B, T, N, seed = 1, 864, 1000, SEED + 107
    n_el = B * T * 256
    mel = synth.synth_mel(seed, B, T)
    out = {"mel": mel, "N": np.int64(N), "seed": np.int64(seed), "ckpt_idx": np.arange(0, N + 1, 125), "sub": np.int64(7)}
    seqs = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        model = make_model(dt, contractive=True)
        model._mel = torch.from_numpy(mel).to(dt)

        def noises():
            yield synth.hash_normal(seed, 1, n_el).reshape(B, 1, T * 256)
            for n in range(N - 1, 0, -1):
                yield synth.hash_normal(seed, 2 + n, n_el).reshape(B, 1, T * 256)
        t0 = time.time()
        res = run_sampler(model, dt, B, T, N, {k: dh[k] for k in ("T", "alpha", "beta", "sigma")}, noises(), False, True)
        seqs[tag] = [res[k].numpy() for k in range(0, N + 1, 125)]
        print("s7", tag, "done in", round(time.time() - t0), "s; max|x| at the checkpoints", [round(float(np.abs(c).max()), 3) for c in seqs[tag]], flush=True)
        del res
    out["y_f64"], out["y_f32"] = seqs["f64"][-1], seqs["f32"][-1]
    out["ckpt_f64_sub"] = np.stack([c[..., ::7] for c in seqs["f64"]]).astype(np.float32)
    out["ckpt_peak"] = np.array([np.abs(c).max() for c in seqs["f64"]])
    out["ckpt_ref_drift"] = np.array([np.abs(a.astype(np.float64) - b).max() for a, b in zip(seqs["f32"], seqs["f64"])])
    np.savez_compressed(os.path.join(GOLD, "sample_s7.npz"), **out)
    print("s7 reference f32-vs-f64 at the checkpoints", out["ckpt_ref_drift"], "peaks", out["ckpt_peak"])


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    which = sys.argv[1:] or ["schedule", "embed", "ops", "forward", "sample", "manifest", "collate", "collater", "frontend", "frontend_tacotron",
                             "frontend_pwg", "noise_scheduling", "theta_loss", "phi_loss", "theta_grad", "lvc_grad", "test_step", "forward_cfg", "mel_bank"]
    dh = gen_schedule()
    if "embed" in which:
        gen_embed()
    if "ops" in which:
        gen_ops()
    if "forward" in which:
        gen_forward()
    if "sample" in which:
        gen_sample(dh)
    for w in which:                       # "sample:s6" regenerates one sampler case
        if w.startswith("sample:"):
            gen_sample(dh, only=w.split(":", 1)[1].split(","))
    if "mel_bank" in which:
        gen_mel_bank_third_party()
    if "sample_long" in which:            # not in the default list: ~25 minutes
        gen_sample_long(dh)
    if "manifest" in which:
        gen_statedict_manifest()
    if "collate" in which:
        gen_collate()
    if "lvc_grad" in which:
        gen_lvc_grad()
    if "collater" in which:
        gen_collater()
    if "frontend_pwg" in which:
        gen_frontend_pwg()
    if "frontend" in which:
        gen_frontend()
    if "theta_loss" in which:
        gen_theta_loss(dh)
    if "phi_loss" in which:
        gen_phi_loss(dh)
    if "theta_grad" in which:
        gen_theta_grad(dh)
    if "noise_scheduling" in which:
        gen_noise_scheduling(dh)
    if "test_step" in which:
        gen_test_step(dh)
    if "forward_cfg" in which:
        gen_forward_cfg(dh)
    if "frontend_tacotron" in which:
        gen_frontend_tacotron()
    print("golden fixtures written to", GOLD)
