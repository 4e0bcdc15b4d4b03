import sys, torch
sys.path.insert(0, "/root/repo")
from srfdet3d_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x0 = torch.randn(64 << 20, device=dev)
for _ in range(20): x0.mul_(1.0)
def timeit(fn, n=20):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
K, Cout = 1728, 768
w = (torch.randn(Cout, K, generator=g) / K ** 0.5).to(dev)
p = ops.pack_conv1x1_nhwc_weights(w)
for M in (32768, 33792, 34816, 34800, 36864, 49152):
    x = torch.randn(1, 1, M, K, generator=g).to(dev)
    y = ops.conv1x1_nhwc(x, p, Cout)
    t = timeit(lambda: ops.conv1x1_nhwc(x, p, Cout, out=y))
    tiles = ((M + 127) // 128) * 6
    print(f"M={M} tiles={tiles} rounds={tiles/768:.3f}: {t:8.1f} us {2.0*M*K*Cout/t/1e6:6.1f} TF  us per 1000 tiles {t/tiles*1000:.1f}", flush=True)
