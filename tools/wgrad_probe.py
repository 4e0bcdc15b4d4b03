"""Weight-gradient kernels MIOpen picks for the trainable 3x3 layers of config 4, channels-last vs contiguous operands
(developer probe: which layout to hand to aten.convolution_backward in train_conv.py)."""
import time

import torch

dev = torch.device("cuda:0")
shapes = [(12, 256, 256, 232, 400), (12, 256, 128, 232, 400), (12, 256, 256, 116, 200), (12, 192, 192, 58, 100), (12, 768, 192, 58, 100),
          (12, 224, 224, 29, 50), (12, 1024, 224, 29, 50)]
for (N, Cin, Cout, H, W) in shapes:
    x = torch.randn(N, Cin, H, W, device=dev)
    gy = torch.randn(N, Cout, H, W, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev)
    res = []
    for fmt in (torch.channels_last, torch.contiguous_format):
        xf, gf = x.contiguous(memory_format=fmt), gy.contiguous(memory_format=fmt)
        wf = w.contiguous(memory_format=fmt)
        f = lambda: torch.ops.aten.convolution_backward(gf, xf, wf, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (False, True, False))[1]
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 3 * 1e3)
    fl = 2.0 * 9 * Cin * Cout * N * H * W
    print(f"{N}x{Cin}->{Cout}@{H}x{W}: channels_last {res[0]:7.2f} ms ({fl / res[0] / 1e9:5.1f} TF)  contiguous {res[1]:7.2f} ms ({fl / res[1] / 1e9:5.1f} TF)", flush=True)
