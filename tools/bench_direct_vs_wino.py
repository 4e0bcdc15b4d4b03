"""3x3 / stride-1 layers of the BEV backbone: Winograd (srf_wino3x3) against the direct implicit-im2col GEMM
(srf_conv_gemm_nhwc) -- the maps are small enough for the Winograd kernel's 64-tile x 64-channel work items to quantise
badly (264-288 items on 256 CUs).  python tools/bench_direct_vs_wino.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srfdet3d_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
shapes = [(1, 184, 184, 128, 128), (1, 184, 184, 256, 128), (1, 92, 92, 256, 256), (1, 192, 192, 128, 128), (1, 176, 200, 128, 128),
          (1, 96, 96, 256, 256), (6, 29, 50, 224, 224), (6, 58, 100, 192, 192)]
x0 = torch.randn(64 << 20, device=dev)
for _ in range(20):
    x0.mul_(1.0)
for (N, H, W, Cin, Cout) in shapes:
    x = torch.randn(N, H, W, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(dev)
    pw = ops.pack_wino3x3_weights(w)
    pg = ops.pack_conv_gemm_weights(w)
    sh = torch.zeros(Cout, device=dev)
    yw = ops.wino3x3(x, pw, Cout, None, sh, True)
    yg = ops.conv_gemm_nhwc(x, pg, Cout, (3, 3), 1, 1, None, sh, True)
    err = (yw - yg).abs().max().item() / yg.abs().max().item()
    res = []
    for fn in (lambda: ops.wino3x3(x, pw, Cout, None, sh, True, out=yw), lambda: ops.conv_gemm_nhwc(x, pg, Cout, (3, 3), 1, 1, None, sh, True, out=yg)):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / 20)
    print(f"{Cin:4d}->{Cout:4d} @{N}x{H}x{W}: wino {res[0]:7.1f} us   direct gemm {res[1]:7.1f} us   rel diff {err:.2e}", flush=True)
