"""In-kernel phase times of srf_wino3x3_k (SRF_WINO_DBG=8): prologue / loop / epilogue cycles per workgroup.
Needs the developer build of the library (`python -m srfdet3d_amd.build --dev`: ablation kernels + stamp hook)."""
import ctypes
import os
import sys

os.environ["SRF_WINO_DBG"] = "8"
os.environ["SRF_DEV_LIB"] = "1"   # load libsrfdet3d_hip_dev.so (never the production library)
os.environ["SRF_WINO_W8"] = "0"
import torch

sys.path.insert(0, ".")
from srfdet3d_amd import _lib, ops  # noqa: E402

N, H, W, Cin, Cout = (int(v) for v in sys.argv[1:6]) if len(sys.argv) > 5 else (6, 232, 400, 128, 128)
g = torch.Generator().manual_seed(0)
x = torch.randn(N, H, W, Cin, generator=g).cuda()
w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).cuda()
pk = ops.pack_wino3x3_weights(w)
out = torch.empty(N, H, W, Cout, device="cuda")
nblk = 40000
stamps = torch.zeros(nblk * 4, dtype=torch.int64, device="cuda")
L = _lib.lib()
fn = ctypes.CDLL(_lib.DEV_LIB_PATH).srf_dev_set_stamp_buffer
fn.argtypes = [ctypes.c_void_p]
fn(ctypes.c_void_p(stamps.data_ptr()))
for _ in range(3):
    ops.wino3x3(x, pk, Cout, None, None, True, out=out)
torch.cuda.synchronize()
s = stamps.view(-1, 4).cpu()
s = s[s[:, 3] > 0]
pro = (s[:, 1] - s[:, 0]).double()
loop = (s[:, 2] - s[:, 1]).double()
epi = (s[:, 3] - s[:, 2]).double()
tot = (s[:, 3] - s[:, 0]).double()
span = (s[:, 3].max() - s[:, 0].min()).item()
print(f"workgroups {len(s)}  nchunk {Cin // 8}")
for name, v in (("prologue", pro), ("loop", loop), ("epilogue", epi), ("total", tot)):
    print(f"{name:9s} mean {v.mean():9.0f}  median {v.median():9.0f}  min {v.min():9.0f}  max {v.max():9.0f} cycles")
print(f"loop per chunk: {loop.mean() / (Cin // 8):.0f} cycles (4096 = 64 MFMAs back to back)")
print(f"kernel span {span} ticks (s_memtime, 100 MHz?)")
