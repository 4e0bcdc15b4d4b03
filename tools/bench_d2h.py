import time, torch
x = torch.randn(4600, device="cuda")
p = torch.empty(4600, pin_memory=True)
torch.cuda.synchronize()
def t(fn, n=200):
    for _ in range(20): fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e6
print("x.cpu()                         %.1f us" % t(lambda: x.cpu()))
def f2():
    p.copy_(x, non_blocking=True); torch.cuda.current_stream().synchronize()
print("pinned copy_ + stream sync      %.1f us" % t(f2))
ev = torch.cuda.Event()
def f3():
    p.copy_(x, non_blocking=True); ev.record(); ev.synchronize()
print("pinned copy_ + event sync       %.1f us" % t(f3))
def f4():
    x.add_(1.0); x.cpu()
print("kernel + x.cpu()                %.1f us" % t(f4))
def f5():
    x.add_(1.0); p.copy_(x, non_blocking=True); torch.cuda.current_stream().synchronize()
print("kernel + pinned copy + sync     %.1f us" % t(f5))
def f6():
    x.add_(1.0); torch.cuda.current_stream().synchronize()
print("kernel + sync                   %.1f us" % t(f6))
