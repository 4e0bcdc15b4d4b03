"""In-kernel s_memtime stamps of srf_spconv_gsp_k (developer library: `python -m srfdet3d_amd.build --dev`, then
`SRF_DEV_LIB=1 SRF_GSP_ABL=5 python tools/gsp_stamps.py [level]`): per-workgroup prologue / loop / epilogue cycles and the per-step spread -- the
measurement behind DESIGN.md section 4, "a wave issues no vector instruction while its partner streams MFMAs"."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srfdet3d_amd import _lib, ops, synthetic  # noqa: E402

lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
pts = torch.from_numpy(synthetic.nuscenes_sweep(2000, 30000)).to(dev)
_, c, _, _ = ops.hard_voxelize(pts, [0.075, 0.075, 0.2], list(synthetic.NUSC_RANGE), 10, 160000)
idx = torch.cat([torch.zeros((c.shape[0], 1), dtype=torch.int32, device=dev), c], 1).contiguous()
shape = [41, 1472, 1472]
idx = idx[ops.spatial_order(idx, shape, 1)].contiguous()
specs = [(16, [1, 1, 1]), (32, [1, 1, 1]), (64, [0, 1, 1]), (128, None)]
g = torch.Generator(device="cpu").manual_seed(0)
for l, (C, pad) in enumerate(specs, 1):
    if l == lvl:
        table = ops.coord_table_build(idx, shape, 1)
        nbr, cnt = ops.rulebook_subm(idx, shape, [3, 3, 3], table)
        A = idx.shape[0]
        f = torch.randn(A, C, generator=g).to(dev)
        W = (torch.randn(27, C, C, generator=g) * 0.05).to(dev)
        pk = ops.pack_spconv_weights(W)
        tiles = ops.spconv_tiles(nbr)
        for _ in range(5):
            ops.spconv_fwd(f, W, nbr, None, None, f, True, packed=pk, tiles=tiles)
        torch.cuda.synchronize()
        buf = (ctypes.c_longlong * (512 * 16))()
        L = _lib.lib()
        L.srf_dev_gsp_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
        assert L.srf_dev_gsp_stamps(buf, 512 * 16) == 0
        a = np.frombuffer(buf, dtype=np.int64).reshape(512, 16)
        np.save(os.path.join("gpurun_out", f"stamps_L{lvl}.npy"), a)
        tl = tiles.cpu().numpy()
        np.save(os.path.join("gpurun_out", f"tiles_L{lvl}.npy"), tl)
        np.save(os.path.join("gpurun_out", f"nbr_L{lvl}.npy"), (nbr.cpu().numpy() >= 0).astype(np.uint8))
        a = a[a[:, 4] > 0]
        nsub = a[:, 4] // 1000000
        a[:, 4] = a[:, 4] % 1000000
        if len(nsub) and nsub.max() > 0:
            for n in sorted(set(nsub.tolist())):
                sel = nsub == n
                print(f"  workgroups with {n} sub-tile(s): {sel.sum()}, whole kernel mean {a[sel, 8].mean():.0f} max {a[sel, 8].max():.0f}, steps mean {a[sel, 4].mean():.1f}, "
                      f"prologue {a[sel, 5].mean():.0f} loop {a[sel, 6].mean():.0f} epilogue {a[sel, 7].mean():.0f}")
        st = a[:, 4].astype(float)
        print(f"L{lvl} A={A} C={C}: teams {len(a)}, steps/team mean {st.mean():.1f} max {st.max():.0f}")
        for j, n in enumerate(["S phase", "wait after S", "M phase", "wait after M"]):
            print(f"  {n:14s}: {np.mean(a[:, j] / st):8.0f} cycles/step (min {np.min(a[:, j] / st):.0f}, max {np.max(a[:, j] / st):.0f})")
        print(f"  whole kernel  : mean {a[:, 8].mean():.0f} max {a[:, 8].max():.0f} cycles; in steps {np.mean(a[:, :4].sum(1)):.0f}; prologue {a[:, 5].mean():.0f} loop {a[:, 6].mean():.0f} (max {a[:, 6].max():.0f}) epilogue {a[:, 7].mean():.0f}")
        for t in (0, 1):
            b = a[t::2]
            sb = b[:, 4].astype(float)
            print(f"  team {t}: S {np.mean(b[:, 0] / sb):.0f}  w {np.mean(b[:, 1] / sb):.0f}  M {np.mean(b[:, 2] / sb):.0f}  w {np.mean(b[:, 3] / sb):.0f}")
        break
    idx, _, _, _, shape = ops.rulebook_strided(idx, shape, 1, [3, 3, 3], [2, 2, 2], pad)
