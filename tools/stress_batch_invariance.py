"""Stress: an utterance's PCM must not depend on the micro-batch it rides in (lens + per-utterance noise streams).
Runs the same 9 utterances through infer.synthesize with max_batch 1, 2, 3, 4, 9 several times and reports every mismatch."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import gpu_common
from fastdiff_amd import infer

g = torch.Generator().manual_seed(11)
lens = [40, 12, 33, 7, 25, 18, 40, 3, 29]
items = [{"item_name": f"utt{i:02d}.npy", "mel": torch.rand(t, 80, generator=g) * 7.5 - 6.0, "len": t} for i, t in enumerate(lens)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
bad = 0
ref = None
for rep in range(reps):
    model = gpu_common.make_model() if rep % 3 == 0 else model      # a fresh handle every third repetition (cold first call)
    for mb in (2, 4, 1, 9, 3):
        out = infer.synthesize(model, items, n_steps=4, max_batch=mb, seed=77)
        if ref is None:
            ref = out
            continue
        for name in ref:
            if not np.array_equal(out[name], ref[name]):
                d = np.abs(out[name].astype(np.int32) - ref[name].astype(np.int32))
                idx = np.nonzero(d)[0]
                print(f"rep {rep} max_batch {mb} {name}: {idx.size} of {d.size} samples differ, max |d| {d.max()}, first at {idx[0]} last at {idx[-1]}")
                bad += 1
print("mismatches:", bad)
