"""One 1x1 layer shape, a few launches of srf_conv1x1_nhwc only (+ the same GEMM on torch/rocBLAS): target of rocprofv3 --pmc.
python tools/prof_gemm.py M K Cout [reps]"""
import sys

import torch

sys.path.insert(0, ".")
from srfdet3d_amd import ops  # noqa: E402

M, K, Cout = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 8
g = torch.Generator().manual_seed(0)
x = torch.randn(1, 1, M, K, generator=g).cuda()
w = (torch.randn(Cout, K, generator=g) / K ** 0.5).cuda()
pk = ops.pack_conv1x1_nhwc_weights(w)
out = torch.empty(1, 1, M, Cout, device="cuda")
for _ in range(reps):
    ops.conv1x1_nhwc(x, pk, Cout, None, None, False, out=out)
x2 = x.view(M, K)
for _ in range(reps):
    torch.mm(x2, w.t())
torch.cuda.synchronize()
print("done")
