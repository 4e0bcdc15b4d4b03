import sys, torch
sys.path.insert(0, "/root/repo")
from srfdet3d_amd import ops
dev = torch.device("cuda:0")
x = torch.randn(6, 3, 928, 1600, device=dev)
w = torch.randn(64, 3, 3, 3, device=dev) / 5
sc = torch.rand(64, device=dev) + 0.5
sh = torch.randn(64, device=dev)
for _ in range(5): y = ops.stem_conv_nchw(x, w, sc, sh, True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): y = ops.stem_conv_nchw(x, w, sc, sh, True)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20 * 1e3
print(f"stem_1 6x3x928x1600: {t:.1f} us, {(x.numel() + y.numel()) * 4 / t / 1e6:.2f} TB/s")
