import torch, time
x = torch.empty(2*1024**3//4, device='cuda', dtype=torch.float32)
y = torch.empty_like(x)
def t(f, n=10):
    f(); torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e-3
tf = t(lambda: x.fill_(1.0)); print("fill 2GiB: %.1f us  %.2f TB/s write" % (tf*1e6, x.numel()*4/tf/1e12))
tc = t(lambda: y.copy_(x)); print("copy 2GiB: %.1f us  %.2f TB/s (r+w)" % (tc*1e6, 2*x.numel()*4/tc/1e12))
tr = t(lambda: x.sum()); print("sum 2GiB: %.1f us  %.2f TB/s read" % (tr*1e6, x.numel()*4/tr/1e12))
