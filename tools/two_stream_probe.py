"""Does running the batch as two half-batches on two streams (two handles) fill the partly empty last workgroup round of the
short kernels?  python tools/two_stream_probe.py   (prints ms per 8 utterances for 1 x B=8 and 2 x B=4 concurrently)"""
import time

import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import fastdiff_amd
from fastdiff_amd import sampler, schedules

torch.manual_seed(1234)
T, N = 864, 4
rows = sampler.InferenceSchedule(schedules.training_hyperparams(), schedules.noise_schedule_for(N), verbose=False).rows()
one = fastdiff_amd.FastDiff().cuda().eval()
models = [fastdiff_amd.FastDiff().cuda().eval() for _ in range(4)]
for h in models:
    h.load_state_dict(one.state_dict())
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mel = (torch.rand(B, 80, T) * 7.5 - 6.0).cuda()
streams = [torch.cuda.Stream() for _ in range(4)]


def run_split(parts):
    def fn(reps):
        n = B // parts
        for i in range(reps):
            for k in range(parts):
                with torch.cuda.stream(streams[k]):
                    models[k].sample(mel[n * k:n * k + n], rows, seed=i)
    return fn


def run_one(reps):
    for i in range(reps):
        one.sample(mel, rows, seed=i)


with torch.no_grad():
    for name, fn in ((f"1 x B={B}", run_one), (f"2 x B={B // 2}, two streams", run_split(2)), (f"4 x B={B // 4}, four streams", run_split(4))) * 3:
        fn(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(20)
        torch.cuda.synchronize()
        print(f"{name:28s} {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per {B} utterances")
