import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
import fastdiff_amd
from fastdiff_amd import sampler, schedules
torch.manual_seed(1234)
m = fastdiff_amd.FastDiff().cuda().eval()
T = 864
mel = (torch.rand(1, 80, T) * 7.5 - 6.0).cuda()
rows = sampler.InferenceSchedule(schedules.training_hyperparams(), schedules.noise_schedule_for(1000), verbose=False).rows()
for mode in ("f16x2", "fp32"):
    m.set_option("lvc", mode)
    with torch.no_grad():
        out = m.sample(mel, rows, seed=1); torch.cuda.synchronize()
        t0 = time.time(); out = m.sample(mel, rows, seed=2); torch.cuda.synchronize(); dt = time.time() - t0
    print(mode, "N=1000: %.0f ms, |x|max %.3g" % (dt * 1e3, float(out.abs().max())))
seq = m.sample(mel[:, :, :64], rows, seed=1, return_sequence=True)
mx = [float(s.abs().max()) for s in seq[::100]]
print("max |x| every 100 steps (T=64):", ["%.3g" % v for v in mx])
