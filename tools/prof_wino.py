"""One 3x3 layer shape, a few launches of srf_wino3x3 only: the target of the `rocprofv3 --pmc` passes.
python tools/prof_wino.py N H W Cin Cout [reps]"""
import sys

import torch

sys.path.insert(0, ".")
from srfdet3d_amd import ops  # noqa: E402

N, H, W, Cin, Cout = (int(v) for v in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 8
g = torch.Generator().manual_seed(0)
x = torch.randn(N, H, W, Cin, generator=g).cuda()
w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).cuda()
pk = ops.pack_wino3x3_weights(w)
out = torch.empty(N, H, W, Cout, device="cuda")
shift = torch.zeros(Cout, device="cuda")
for _ in range(reps):
    ops.wino3x3(x, pk, Cout, None, shift, True, out=out)
torch.cuda.synchronize()
print("done", float(out.abs().max()))
