"""The 1x1 GEMM shapes of the LC camera branch (OSA concat convolutions with the fused eSE mean, FPN laterals), timed one by
one (developer tool; both with and without the mixed-tile launch).  python tools/stage_gemm_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srfdet3d_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x0 = torch.randn(64 << 20, device=dev)
for _ in range(20):
    x0.mul_(1.0)


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


N = 6
SHAPES = [("stage2 concat", 232 * 400, 768, 256, True, 1), ("stage3 concat 0", 116 * 200, 1056, 512, True, 1),
          ("stage3 concat", 116 * 200, 1312, 512, True, 2), ("stage4 concat 0", 58 * 100, 1472, 768, True, 1),
          ("stage4 concat", 58 * 100, 1728, 768, True, 8), ("stage5 concat 0", 29 * 50, 1888, 1024, True, 1),
          ("stage5 concat", 29 * 50, 2144, 1024, True, 2), ("fpn lateral 2", 232 * 400, 256, 256, False, 1),
          ("fpn lateral 3", 116 * 200, 512, 256, False, 1), ("fpn lateral 4", 58 * 100, 768, 256, False, 1)]
total = [0.0, 0.0]
for name, HW, K, Cout, pool, count in SHAPES:
    x = torch.randn(N, 1, HW, K, device=dev)
    w = (torch.randn(Cout, K, generator=g) / K ** 0.5).to(dev)
    p = ops.pack_conv1x1_nhwc_weights(w)
    pd = ops.pack_conv1x1_nhwc_direct_weights(w)
    sh = torch.zeros(Cout, device=dev)
    ts = {"0": [], "1": []}
    for rep in range(5):          # the two kernels interleaved: box and clock drift hit both alike
        for tail in ("0", "1"):   # "0": srf_conv1x1_nhwc (LDS-staged), "1": srf_conv1x1_nhwc_direct (round 2 compared SRF_GEMM_TAIL
            # here; that knob is read once per process now)
            os.environ["SRF_GEMM_DIRECT"] = tail
            ts[tail].append(timeit(lambda: ops.conv1x1_nhwc(x, p, Cout, None, sh, True, pool=pool, packed_direct=pd)))
    t0, t1 = min(ts["0"]), min(ts["1"])
    total[0] += t0 * count
    total[1] += t1 * count
    fl = 2.0 * N * HW * K * Cout
    print(f"{name:18s} HW={HW:6d} K={K:5d} Cout={Cout:5d}: one size {t0:8.1f} us {fl / t0 / 1e6:6.1f} TF | mixed {t1:8.1f} us {fl / t1 / 1e6:6.1f} TF  x{count}",
          flush=True)
    del x
print(f"per LC frame: one size {total[0] / 1e3:.3f} ms, mixed {total[1] / 1e3:.3f} ms")
