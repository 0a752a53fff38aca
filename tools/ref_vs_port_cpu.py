"""The reference itself against its port, on the SAME CPU with the SAME threads (build container only: needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python tools/ref_vs_port_cpu.py [threads ...]        -> profiles/r06_ref_vs_port_cpu.json

bench.py's `cpu_baseline` is `oracle/torch_eager.py` (kind "port") because /root/reference cannot travel to the GPU box.  This
records what that substitution is worth: the reference's own `sampling_given_noise_schedule(FastDiff(), (1,1,221184), dh,
[N=4 schedule], condition=mel)` (`.cuda` shimmed to identity, model.eval(), stdout of its prints suppressed, one
warm-up, best of three) next to the port's `EagerFastDiff.sample` and its `sample_like_the_reference` (weight-norm per convolution
call and the per-call host work included) on identical weights, mel and thread count.  Outputs are compared too: same x_T and z.
"""
import io
import json
import os
import sys
import time
from contextlib import redirect_stdout

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("FASTDIFF_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True
torch.Tensor.cuda = lambda self, *a, **k: self

import synth  # noqa: E402
from torch_eager import EagerFastDiff  # noqa: E402
from modules.FastDiff.module.FastDiff_model import FastDiff as RefFastDiff  # noqa: E402
from modules.FastDiff.module import util as ref_util  # noqa: E402
from fastdiff_amd import sampler, schedules  # noqa: E402

T, N = 864, 4
L = T * 256


def best_of(fn, n=3):
    fn()
    best = float("inf")
    for _ in range(n):
        t0 = time.perf_counter()
        out = fn()
        best = min(best, time.perf_counter() - t0)
    return best, out


def main():
    threads = [int(a) for a in sys.argv[1:]] or [os.cpu_count() or 1, 1]
    sd = synth.synth_state_dict(1234)
    ref = RefFastDiff().eval()
    ref.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    mel = torch.from_numpy(synth.synth_mel(1, 1, T))
    dh = ref_util.compute_hyperparams_given_schedule(torch.linspace(0.000001, 0.01, 1000))
    sched = schedules.noise_schedule_for(N)
    rows = sampler.InferenceSchedule(schedules.training_hyperparams(), sched, verbose=False).rows()
    noises = [synth.hash_normal(1, 1, L).reshape(1, 1, L)] + [synth.hash_normal(1, 2 + n, L).reshape(1, 1, L) for n in range(N - 1, 0, -1)]
    lean = EagerFastDiff(sd)
    full = EagerFastDiff(sd, weight_norm_each_forward=True)
    res = {"host": {"cpu_count": os.cpu_count(), "torch": torch.__version__}, "workload": f"B=1 T={T} N={N}, synthetic weights seed 1234, 1 warm-up + best of 3",
           "runs": []}
    for nt in threads:
        torch.set_num_threads(nt)

        def run_ref():
            it = iter(noises)
            orig = ref_util.std_normal
            ref_util.std_normal = lambda size: torch.from_numpy(next(it).copy()).view(*size).clone()
            try:
                with redirect_stdout(io.StringIO()):
                    return ref_util.sampling_given_noise_schedule(ref, (1, 1, L), dh, sched.clone(), condition=mel, ddim=False, return_sequence=False)
            finally:
                ref_util.std_normal = orig

        def run_lean():
            with torch.no_grad():
                return lean.sample(mel, rows, torch.from_numpy(noises[0].copy()), [torch.from_numpy(z) for z in noises[1:]] + [None])

        def run_full():
            with torch.no_grad():
                return full.sample_like_the_reference(mel, dh, sched.clone(), sampler._map_noise_scale_to_time_step_loop)

        t_ref, y_ref = best_of(run_ref)
        t_lean, y_lean = best_of(run_lean)
        t_full, _ = best_of(run_full)
        d = float((y_ref - y_lean).abs().max())
        audio = L / 22050
        res["runs"].append({"threads": nt,
                            "reference_s": round(t_ref, 3), "reference_rtf": round(audio / t_ref, 3),
                            "port_lean_s": round(t_lean, 3), "port_lean_rtf": round(audio / t_lean, 3),
                            "port_like_the_reference_s": round(t_full, 3),
                            "port_lean_over_reference": round(t_lean / t_ref, 4), "port_like_the_reference_over_reference": round(t_full / t_ref, 4),
                            "max_abs_diff_port_vs_reference_x0": d, "max_abs_x0": float(y_ref.abs().max())})
        print(res["runs"][-1], flush=True)
    res["reading"] = ("cpu_baseline.value in bench.py is the lean port's RTF on the GPU box's host; the reference itself on the same CPU and threads "
                      "takes port_lean_over_reference^-1 x that time here")
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r06_ref_vs_port_cpu.json"), "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
