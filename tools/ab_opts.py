"""A/B of library options inside ONE process and ONE GPU session (box-to-box variation is +-4 %, in-session +-0.3 %):
    python tools/ab_opts.py [--batch 8] [--frames 864] [--nsteps 4] [--reps 3] [--steps 20] "" "fuse_up=0" "hoist=off" ...
Every argument is one configuration ("k=v,k=v"; "" = defaults).  The configurations are timed in turn, `reps` times round-robin, on the
same model and the same mel (HBM-resident in and out, like bench.py's `value`); printed: ms per sample call, per round and the mean.
Every configuration is its own model instance with its own workspace, and two instances of the SAME configuration can differ by ~1.4 %
(where their buffers landed): list a configuration twice, alternating with the other, before believing a difference of that size
(profiles/r03/s31_order.txt)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import fastdiff_amd
from fastdiff_amd import sampler, schedules

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--frames", type=int, default=864)
ap.add_argument("--nsteps", type=int, default=4)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("configs", nargs="+")
a = ap.parse_args()
torch.manual_seed(1234)
dev = torch.device("cuda", 0)
rows = sampler.InferenceSchedule(schedules.training_hyperparams(), schedules.noise_schedule_for(a.nsteps), verbose=False).rows()
mel = (torch.rand(a.batch, 80, a.frames) * 7.5 - 6.0).to(dev)
models = []
for cfg in a.configs:
    torch.manual_seed(1234)
    m = fastdiff_amd.FastDiff().to(dev).eval()
    for kv in filter(None, cfg.split(",")):
        m.set_option(*kv.split("=", 1))
    models.append(m)
ref = None
times = [[] for _ in a.configs]
with torch.no_grad():
    for i, m in enumerate(models):                     # warm-up + agreement of the results (same seed -> same noise)
        y = m.sample(mel, rows, seed=7)
        for _ in range(2):
            m.sample(mel, rows, seed=8)
        torch.cuda.synchronize()
        if ref is None:
            ref = y
        else:
            print(f"config {i} vs config 0: max |difference| {float((y - ref).abs().max()):.3e}, bit-equal {bool(torch.equal(y, ref))}")
    for r in range(a.reps):
        for i, m in enumerate(models):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(a.steps):
                m.sample(mel, rows, seed=100 + k, defer_check=True)      # fallback = host: each call is looked at by its successor,
            m.check()                                                    # the last one here
            torch.cuda.synchronize()
            times[i].append((time.perf_counter() - t0) / a.steps * 1e3)
for cfg, t in zip(a.configs, times):
    print(f"B={a.batch} T={a.frames} N={a.nsteps}  [{cfg or 'defaults':40s}]  " + "  ".join(f"{v:.3f}" for v in t) + f"   mean {sum(t) / len(t):.3f} ms")
