"""Stress: a frame-bucketed fd_sample call (option t_bucket = 32, 64, 100) must be bit-identical to the exact-T call (t_bucket = 0) whatever
the shape: random B, T, schedule length, `lens`, noise streams, under a graph cache of 3 (constant evictions) on the bucketed handles.
Usage: python tools/stress_bucketing.py [iterations]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import gpu_common as gc
from conftest import load_golden

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
sched = load_golden("schedule")
rows = {n: gc.table_rows(sched, n)[0] for n in (3, 4, 6, 8)}
rows[19] = [{"t": 190.0 - 9.5 * k, "c_eps": 0.02, "c_div": 0.99, "sigma": 0.05, "c1": 1.0, "c2": 0.0, "c3": 0.0, "add_noise": int(k < 18)} for k in range(19)]
exact = gc.make_model()
exact.set_option("t_bucket", "0")
others = {}
for g in (32, 64, 100):
    m = gc.make_model()
    m.set_option("t_bucket", str(g))
    m.set_option("graph_cache", "3")
    others[g] = m
rng = np.random.default_rng(6)
bad = 0
with torch.no_grad():
    for it in range(iters):
        B = int(rng.choice([1, 1, 2, 3, 8]))
        T = int(rng.integers(3, 900)) if it % 10 else int(rng.choice([32, 64, 96, 100, 128, 864]))
        if B * T > 3000:
            T = max(3, 3000 // B)
        N = int(rng.choice([3, 4, 4, 6, 8, 19]))
        use_lens = bool(rng.integers(0, 2)) and B > 1
        lens = sorted((int(v) for v in rng.integers(1, T + 1, B)), reverse=True) if use_lens else None
        if lens:
            lens[0] = T
        ids = [int(v) for v in rng.integers(0, 1 << 40, B)] if rng.integers(0, 2) else None
        seed = int(rng.integers(0, 1 << 30))
        mel = (torch.rand(B, 80, T) * 7.5 - 6.0).cuda()
        kw = dict(seed=seed, lens=lens, stream_ids=ids, ddim=bool(rng.integers(0, 4) == 0), defer_check=bool(rng.integers(0, 2)))
        want = exact.sample(mel, rows[N], **kw)
        for g, m in others.items():
            got = m.sample(mel, rows[N], **kw)
            for b in range(B):
                n = (lens[b] if lens else T) * 256
                if not torch.equal(got[b, :, :n], want[b, :, :n]):
                    d = (got[b, :, :n] - want[b, :, :n]).abs()
                    print(f"iteration {it}: t_bucket {g} B={B} T={T} N={N} lens={lens} utterance {b}: {int((d > 0).sum())} samples differ, max |d| {float(d.max()):.3e}")
                    bad += 1
        if not torch.isfinite(want).all() and lens is None:
            print(f"iteration {it}: non-finite output B={B} T={T} N={N}")
            bad += 1
    for m in [exact, *others.values()]:
        m.check()
print("iterations:", iters, "mismatches:", bad, "captures (exact / 32 / 64 / 100):", exact.counter("graph_captures"), [m.counter("graph_captures") for m in others.values()],
      "evictions:", [m.counter("graph_evictions") for m in others.values()])
sys.exit(1 if bad else 0)
