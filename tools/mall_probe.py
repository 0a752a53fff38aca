"""Does a buffer written by one kernel stay in the 256 MB memory-side cache (MALL) for the next kernel that reads it,
and which traffic in between pushes it out?  (read+write of the same size as the probe: us, lower = served from cache)"""
import torch
def t(f):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); f(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3
big = torch.empty(1024 * 1024 * 1024 // 4, device='cuda'); big.fill_(0.0)
MB = 1024 * 1024 // 4
a = torch.empty(172 * MB, device='cuda'); b = torch.empty_like(a)
cases = [("nothing in between", None), ("read 113 MB", ("r", 113)), ("read 452 MB", ("r", 452)), ("read 1024 MB", ("r", 1024)),
         ("write 57 MB", ("w", 57)), ("write 226 MB", ("w", 226)), ("write 1024 MB", ("w", 1024))]
for name, op in cases:
    us = []
    for rep in range(5):
        big.fill_(0.0); torch.cuda.synchronize()
        a.fill_(1.0)
        if op:
            v = big[: op[1] * MB]
            if op[0] == "r": v.sum()
            else: v.fill_(2.0)
        torch.cuda.synchronize()
        us.append(t(lambda: torch.add(a, 1.0, out=b)))
    print(f"172 MB written, then {name:20s}: consumer (read 172 + write 172 MB) {min(us):6.1f} us")
