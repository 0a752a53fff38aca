"""Where the host time of the single-process config-4 job goes (64 ragged utterances, N=6): cProfile of infer.synthesize."""
import cProfile, pstats, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, fastdiff_amd
from fastdiff_amd import infer
torch.manual_seed(1234)
model = fastdiff_amd.FastDiff().cuda().eval()
items = bench.config4_items()
for i in range(2):
    infer.synthesize(model, items, n_steps=6, max_batch=8, seed=i, drop_last_frame=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
infer.synthesize(model, items, n_steps=6, max_batch=8, seed=5, drop_last_frame=False)
torch.cuda.synchronize()
pr.disable()
print("wall %.1f ms" % ((time.perf_counter() - t0) * 1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
