"""Where the wall time of the single-process config-4 job goes (64 ragged utterances, N=6): wall clock of infer.synthesize and
the GPU time between its first and last operation."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, fastdiff_amd
from fastdiff_amd import infer, shard
torch.manual_seed(1234)
model = fastdiff_amd.FastDiff().cuda().eval()
items = bench.config4_items()
lens = [it["len"] for it in items]
mbs = shard.micro_batches(range(64), lens, 8)
print("padded T per micro-batch:", [max(lens[i] for i in mb) for mb in mbs], "valid frames", sum(lens))
for i in range(2):
    infer.synthesize(model, items, n_steps=6, max_batch=8, seed=i, drop_last_frame=False)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    infer.synthesize(model, items, n_steps=6, max_batch=8, seed=5 + rep, drop_last_frame=False)
    e1.record()
    torch.cuda.synchronize()
    print("wall %.1f ms, first-to-last GPU event %.1f ms" % ((time.perf_counter() - t0) * 1e3, e0.elapsed_time(e1)))

# per-call GPU durations inside the job, and the idle gaps between them
evs = []
orig_sample, orig_pn = model.sample, model.peak_normalize_int16
def sample(*a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y = orig_sample(*a, **k); e1.record(); evs.append(("sample", e0, e1)); return y
def pn(*a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y = orig_pn(*a, **k); e1.record(); evs.append(("epilogue", e0, e1)); return y
model.sample, model.peak_normalize_int16 = sample, pn
infer.synthesize(model, items, n_steps=6, max_batch=8, seed=11, drop_last_frame=False)
torch.cuda.synchronize()
prev = None
for name, a, b in evs:
    gap = prev.elapsed_time(a) if prev is not None else 0.0
    print("%-9s %6.2f ms   (idle before it %6.2f ms)" % (name, a.elapsed_time(b), gap))
    prev = b
