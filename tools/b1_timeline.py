"""One-utterance sample calls (B=1, T given, N=4) under rocprofv3 --kernel-trace --memory-copy-trace: what is on the GPU's timeline between
two calls besides the graph?  Usage (GPU box): rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -o b1 -- python tools/b1_timeline.py run [T]
then: python tools/b1_timeline.py report DIR"""
import glob
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def run(T):
    import torch
    import fastdiff_amd
    from fastdiff_amd import sampler, schedules
    torch.manual_seed(1234)
    m = fastdiff_amd.FastDiff().cuda().eval()
    rows = sampler.InferenceSchedule(schedules.training_hyperparams(), schedules.noise_schedule_for(4), verbose=False).rows()
    mel = (torch.rand(1, 80, T) * 7.5 - 6.0).cuda()
    with torch.no_grad():
        for i in range(30):
            m.sample(mel, rows, seed=i, defer_check=True)
        m.check()
    torch.cuda.synchronize()


def report(d):
    import csv
    ev = []
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][-60:]))
    for path in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "M " + r.get("Direction", r.get("Name", "copy")) + " " + r.get("Size", "")))
    ev.sort()
    # the last full call: from the last-but-one "init_noise" kernel to the last one
    idx = [i for i, e in enumerate(ev) if "k_init_noise" in e[2]]
    if len(idx) < 3:
        print("not enough calls in the trace", len(ev))
        return
    a, b = idx[-3], idx[-2]
    t0 = ev[a][0]
    busy = 0
    prev_end = None
    print(f"one call: {len(ev[a:b])} GPU operations, {(ev[b][0] - t0) / 1e3:.1f} us from its init_noise to the next call's")
    for s, e, name in ev[a:b]:
        gap = 0 if prev_end is None else (s - prev_end) / 1e3
        busy += (e - s)
        print(f"  +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  {name}")
        prev_end = max(prev_end or e, e)
    print(f"busy {busy / 1e3:.1f} us of {(ev[b][0] - t0) / 1e3:.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 864)
    else:
        report(sys.argv[2])
