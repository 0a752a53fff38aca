import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle'); sys.path.insert(0, '/root/repo/tests')
import fastdiff_amd
import gpu_common as gc
from conftest import load_golden
from fastdiff_amd import sampler, schedules
m = gc.make_model()
g = load_golden("sample_s1")
sched = sampler.InferenceSchedule(schedules.training_hyperparams(), schedules.noise_schedule_for(4), verbose=False)
rows = sched.rows()
noise = torch.from_numpy(gc.exec_order_noise(gc.noise_from_seed(int(g["seed"]), 2, 6, 4))).cuda()
args = dict(x_T=torch.from_numpy(g["x_T"]).cuda(), noise=noise)
mel = torch.from_numpy(g["mel"]).cuda()
for lvc in ("f16x2", "fp32"):
    m.set_option("lvc", lvc)
    for graph in ("1", "1", "1", "0", "0"):
        m.set_option("graph", graph)
        with torch.no_grad():
            y = m.sample(mel, rows, **args)
        torch.cuda.synchronize()
        fl = m.read_tap("range_flags").view(np.int32)
        print(lvc, "graph", graph, "flags", fl[:13], "sum", repr(float(y.double().sum())))
