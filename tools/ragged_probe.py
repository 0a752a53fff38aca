"""Time of one padded micro-batch (B=8, N=6) with and without `lens`, full-length and ragged, graph cache warm."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fastdiff_amd
from fastdiff_amd import sampler, schedules
torch.manual_seed(1234)
m = fastdiff_amd.FastDiff().cuda().eval()
rows = sampler.InferenceSchedule(schedules.training_hyperparams(), schedules.noise_schedule_for(6), verbose=False).rows()

def t(B, T, lens, reps=10, **kw):
    mel = (torch.rand(B, 80, T) * 7.5 - 6.0).cuda()
    with torch.no_grad():
        for _ in range(2): m.sample(mel, rows, seed=1, lens=lens, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(reps): m.sample(mel, rows, seed=i, lens=lens, **kw)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

print("B=8 T=864 no lens      %.2f ms" % t(8, 864, None))
print("B=8 T=864 lens all 864 %.2f ms" % t(8, 864, [864] * 8))
print("B=8 T=864 lens 800-864 %.2f ms" % t(8, 864, [864, 860, 850, 840, 830, 820, 810, 800]))
print("B=8 T=600 no lens      %.2f ms" % t(8, 600, None))
print("B=8 T=600 lens 560-600 %.2f ms" % t(8, 600, [600, 595, 590, 585, 580, 575, 570, 560]))
print("B=8 T=600 lens+ids     %.2f ms" % t(8, 600, [600, 595, 590, 585, 580, 575, 570, 560], stream_ids=list(range(8))))
print("B=8 T=300 lens 260-300 %.2f ms" % t(8, 300, [300, 295, 290, 285, 280, 275, 270, 260]))

# alternating padded shapes, as a length-sorted job does: is a change of shape expensive?
shapes = [(8, 842), (8, 792), (8, 663), (8, 594), (8, 537), (8, 457), (8, 380), (8, 306)]
mels = {s: (torch.rand(s[0], 80, s[1]) * 7.5 - 6.0).cuda() for s in shapes}
lens = {s: [s[1] - 3 * i for i in range(8)] for s in shapes}
with torch.no_grad():
    for rnd in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        per = []
        for s in shapes:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); y = m.sample(mels[s], rows, seed=rnd, lens=lens[s], stream_ids=list(range(8))); e1.record()
            per.append((e0, e1))
        torch.cuda.synchronize()
        print("round %d: 8 shapes in %.1f ms wall; per call (GPU events) %s" % (rnd, (time.perf_counter() - t0) * 1e3, ["%.1f" % a.elapsed_time(b) for a, b in per]))
    # the same with the int16 epilogue and the copy to pinned memory after every call
    pin = torch.empty(8 * 864 * 256, dtype=torch.int16).pin_memory()
    for rnd in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        per = []
        for s in shapes:
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(); y = m.sample(mels[s], rows, seed=rnd, lens=lens[s], stream_ids=list(range(8))); e1.record()
            pcm = m.peak_normalize_int16(y, valid=[t * 256 for t in lens[s]])
            pin[: pcm.numel()].view(pcm.shape).copy_(pcm, non_blocking=True); e2.record()
            per.append((e0, e1, e2))
        torch.cuda.synchronize()
        print("with epilogue + D2H: %.1f ms wall; sample %s; epilogue+copy %s" % ((time.perf_counter() - t0) * 1e3, ["%.1f" % a.elapsed_time(b) for a, b, c in per], ["%.1f" % b.elapsed_time(c) for a, b, c in per]))
