"""Functional layer over the C ABI: torch device tensors in, torch device tensors out.

torch is used here only to own device memory and to name the current HIP stream; every computation is a
kernel of libsrfdet3d_hip.so.  All functions raise on CPU tensors (there is no CPU fallback).
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import FeatMap, check, hf, hi


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """hipStream_t of torch's current stream.  torch.cuda.current_stream() costs ~9 us of Python per call (35 calls per
    frame); the raw accessor the graph-capture machinery itself uses is ~30x cheaper."""
    if _raw_stream is not None and _cur_device is not None:
        return ctypes.c_void_p(_raw_stream(_cur_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _dev(t, name, dtype=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"srfdet3d_amd: `{name}` must be a GPU tensor (no CPU fallback exists)")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"srfdet3d_amd: `{name}` must be {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _empty(shape, dtype, device):
    return torch.empty(shape, dtype=dtype, device=device)


def grid_size(voxel_size, pc_range):
    """mmcv Voxelization.__init__: round((range[3:] - range[:3]) / voxel_size) in float32 -> (gx, gy, gz)."""
    r = np.asarray(pc_range, np.float32)
    v = np.asarray(voxel_size, np.float32)
    return [int(x) for x in np.round((r[3:] - r[:3]) / v)]


# ---------------------------------------------------------------------------------------------- voxelization
def dynamic_voxelize(points, voxel_size, pc_range):
    points = _dev(points, "points", torch.float32)
    n, nf = points.shape
    coors = _empty((n, 3), torch.int32, points.device)
    check(_lib.lib().srf_dynamic_voxelize(_ptr(points), n, nf, hf(voxel_size), hf(pc_range),
                                          hi(grid_size(voxel_size, pc_range)), _ptr(coors), _stream()),
          "dynamic_voxelize")
    return coors


def points_filter(points, pc_range=None, close_radius=0.0, static=False, with_index=False):
    """PointsRangeFilter (strict inequalities on x, y, z against pc_range) and / or the multi-sweep `remove_close`
    (|x| < r and |y| < r dropped) as one order-preserving compaction -> kept points (M, nf) [, their source rows].
    One D2H sync for M; static=True returns all n rows (rows >= M undefined) and the device scalar M instead."""
    points = _dev(points, "points", torch.float32)
    n, nf = points.shape
    L = _lib.lib()
    dev = points.device
    out = _empty((max(n, 1), nf), torch.float32, dev)
    index = _empty((max(n, 1),), torch.int32, dev) if with_index else None
    num = _empty((1,), torch.int32, dev)
    ws = _empty((max(L.srf_points_filter_workspace_bytes(n), 4),), torch.uint8, dev)
    check(L.srf_points_filter(_ptr(points), n, nf, hf(pc_range) if pc_range is not None else None, float(close_radius),
                              _ptr(out), _ptr(index), _ptr(num), _ptr(ws), _stream()), "points_filter")
    if static:
        return (out[:n], index[:n], num) if with_index else (out[:n], num)
    M = int(num.item())
    return (out[:M], index[:M]) if with_index else out[:M]


def image_prepare(images_u8, mean, std, to_rgb=False, size_divisor=32, size=None):
    """(V, H, W, 3) uint8 decoded views -> (V, 3, Hp, Wp) float32: NormalizeMultiviewImage + PadMultiViewImage + the
    HWC -> CHW transpose of the format bundle in one pass.  Padding to `size` (Hp, Wp) or up to a multiple of
    `size_divisor`."""
    if not isinstance(images_u8, torch.Tensor) or not images_u8.is_cuda or images_u8.dtype != torch.uint8:
        raise RuntimeError("srfdet3d_amd: `images_u8` must be a GPU uint8 tensor (no CPU fallback exists)")
    images_u8 = images_u8.contiguous()
    V, H, W, C = images_u8.shape
    if C != 3:
        raise RuntimeError("srfdet3d_amd: image_prepare expects 3-channel views")
    if size is not None:
        Hp, Wp = int(size[0]), int(size[1])
    else:
        d = int(size_divisor)
        Hp, Wp = -(-H // d) * d, -(-W // d) * d
    out = _empty((V, 3, Hp, Wp), torch.float32, images_u8.device)
    check(_lib.lib().srf_image_prepare(_ptr(images_u8), V, H, W, hf(mean), hf(std), int(bool(to_rgb)), Hp, Wp, _ptr(out),
                                       _stream()), "image_prepare")
    return out


def hard_voxelize(points, voxel_size, pc_range, max_points, max_voxels, mean_features=0, static=False, batch_index=0):
    """-> voxels (M,max_points,nf), coors (M,3) zyx, num (M,), mean (M,mean_features) or None.  One D2H sync for M.
    static=True: no sync; all min(n, max_voxels) rows are returned, rows >= M being padding (coors -1, num 0, zeros), the coordinates
    as (rows, 4) (batch_index, z, y, x), followed by the device scalar M -- the fixed-shape form a hipGraph can replay
    (srf_hard_voxelize_static: the padding is written by the gather kernel itself, no fills and no batch-column ops around it)."""
    points = _dev(points, "points", torch.float32)
    n, nf = points.shape
    L = _lib.lib()
    dev = points.device
    rows = max(min(n, max_voxels), 1)
    voxels = _empty((rows, max_points, nf), torch.float32, dev)
    coors = _empty((rows, 4 if static else 3), torch.int32, dev)
    num = _empty((rows,), torch.int32, dev)
    mean = _empty((rows, mean_features), torch.float32, dev) if mean_features else None
    vnum = _empty((1,), torch.int32, dev)
    ws_bytes = L.srf_hard_voxelize_workspace_bytes(n, max_points)
    ws = _empty((max(ws_bytes, 1),), torch.uint8, dev)
    if static:
        if n < 1:
            raise ValueError("hard_voxelize(static=True) needs at least one point row")
        check(L.srf_hard_voxelize_static(_ptr(points), n, nf, hf(voxel_size), hf(pc_range), hi(grid_size(voxel_size, pc_range)),
                                         max_points, max_voxels, _ptr(voxels), _ptr(coors), _ptr(num), _ptr(vnum), _ptr(mean),
                                         mean_features, int(batch_index), _ptr(ws), ws_bytes, _stream()), "hard_voxelize_static")
        return voxels, coors, num, mean, vnum
    check(L.srf_hard_voxelize(_ptr(points), n, nf, hf(voxel_size), hf(pc_range), hi(grid_size(voxel_size, pc_range)),
                              max_points, max_voxels, _ptr(voxels), _ptr(coors), _ptr(num), _ptr(vnum), _ptr(mean),
                              mean_features, _ptr(ws), ws_bytes, _stream()), "hard_voxelize")
    M = int(vnum.item())
    return voxels[:M], coors[:M], num[:M], (mean[:M] if mean is not None else None)


# ---------------------------------------------------------------------------------------------- dynamic scatter
class VoxelMap:
    """Sorted unique voxels of a (n,4) coordinate list and the point lists of each voxel."""

    def __init__(self, coors, grid_zyx, batch, known_num_voxels=None, static_rows=None):
        """static_rows: fixed-shape mode for hipGraph replay -- nothing is read back; `coors` / `reduce` have static_rows
        rows, those past the real count (device scalar `num_dev`) being padding (coordinates -1, zero features)."""
        coors = _dev(coors, "coors", torch.int32)
        n = coors.shape[0]
        L = _lib.lib()
        dev = coors.device
        self.n = n
        rows = max(n, 1)
        self.static = static_rows is not None
        out_coors = torch.full((rows, 4), -1, dtype=torch.int32, device=dev) if self.static else _empty((rows, 4), torch.int32, dev)
        self.point2voxel = _empty((rows,), torch.int32, dev)
        self.counts = _empty((rows,), torch.int32, dev)
        self.offsets = _empty((rows,), torch.int32, dev)
        self.order = _empty((rows,), torch.int32, dev)
        self.num_dev = _empty((1,), torch.int32, dev)
        ws_bytes = L.srf_voxel_unique_workspace_bytes(n, hi(grid_zyx), batch)
        if ws_bytes == 0 and n > 0:
            raise RuntimeError("srfdet3d voxel_unique: grid too large for 32-bit keys")
        ws = _empty((max(ws_bytes, 1),), torch.uint8, dev)
        check(L.srf_voxel_unique(_ptr(coors), n, hi(grid_zyx), batch, _ptr(out_coors), _ptr(self.point2voxel),
                                 _ptr(self.counts), _ptr(self.offsets), _ptr(self.order), _ptr(self.num_dev), _ptr(ws),
                                 ws_bytes, _stream()), "voxel_unique")
        # the one device->host read of this op; a caller that knows the count (all rows distinct) skips it
        if self.static:
            self.M = min(int(static_rows), rows)
        else:
            self.M = int(self.num_dev.item()) if known_num_voxels is None else int(known_num_voxels)
        self.coors = out_coors[:self.M]
        self.point2voxel = self.point2voxel[:n]

    def reduce(self, feats, mode):
        if torch.is_grad_enabled() and feats.requires_grad:
            return _ScatterReduceFn.apply(feats, self, mode)
        return self._reduce(feats, mode)

    def _reduce(self, feats, mode):
        feats = _dev(feats, "feats", torch.float32)
        C = feats.shape[1]
        if self.static:
            out = torch.zeros((max(self.M, 1), C), dtype=torch.float32, device=feats.device)
        else:
            out = _empty((max(self.M, 1), C), torch.float32, feats.device)
        check(_lib.lib().srf_scatter_reduce(_ptr(feats), _ptr(self.order), _ptr(self.offsets), _ptr(self.counts),
                                            _ptr(self.num_dev), self.M, C, 0 if mode == "mean" else 1, _ptr(out),
                                            _stream()), "scatter_reduce")
        return out[:self.M]


def spatial_order(indices, spatial_shape, batch):
    """Permutation that sorts DISTINCT active sites (A,4) (b,z,y,x) by (b, y, x, z): rows that are close in space
    become close in index, so a tile of consecutive output rows gathers from a compact set of input rows (L2 reuse,
    see csrc/spconv.hip).  Uses the occupancy-bitmap rank of K3 -- no sort, no host sync."""
    indices = _dev(indices, "indices", torch.int32)
    D, H, W = spatial_shape
    byxz = indices[:, [0, 2, 3, 1]].contiguous()
    vm = VoxelMap(byxz, [H, W, D], batch, known_num_voxels=indices.shape[0])
    return vm.order[:indices.shape[0]].long()


# ---------------------------------------------------------------------------------------------- rulebooks
class CoordTable:
    def __init__(self, capacity, device):
        self.capacity = capacity
        self.buf = _empty((capacity * 2,), torch.int32, device)


def coord_table_build(indices, spatial_shape, batch):
    indices = _dev(indices, "indices", torch.int32)
    L = _lib.lib()
    A = indices.shape[0]
    t = CoordTable(L.srf_coord_table_capacity(A), indices.device)
    check(L.srf_coord_table_build(_ptr(indices), A, hi(spatial_shape), batch, _ptr(t.buf), t.capacity, _stream()),
          "coord_table_build")
    return t


def rulebook_subm(indices, spatial_shape, ksize, table):
    indices = _dev(indices, "indices", torch.int32)
    A = indices.shape[0]
    K = int(np.prod(ksize))
    nbr = _empty((K, max(A, 1)), torch.int32, indices.device)
    counts = _empty((K,), torch.int32, indices.device)
    check(_lib.lib().srf_rulebook_subm(_ptr(indices), A, hi(spatial_shape), hi(ksize), _ptr(table.buf), table.capacity,
                                       _ptr(nbr), _ptr(counts), _stream()), "rulebook_subm")
    return nbr[:, :A], counts


def out_spatial_shape(shape, ksize, stride, pad):
    return [int((shape[d] + 2 * pad[d] - ksize[d]) // stride[d] + 1) for d in range(3)]


def rulebook_strided(indices, spatial_shape, batch, ksize, stride, pad):
    """-> out_indices (A_out,4), nbr (K,A_out), pair_counts (K,), out_table, out_shape.  One D2H sync for A_out."""
    indices = _dev(indices, "indices", torch.int32)
    L = _lib.lib()
    dev = indices.device
    A = indices.shape[0]
    K = int(np.prod(ksize))
    bound = L.srf_strided_max_outputs(A, batch, hi(spatial_shape), hi(ksize), hi(stride), hi(pad))
    if bound < 0:
        check(bound, "strided_max_outputs")
    table = CoordTable(L.srf_coord_table_capacity(bound), dev)
    out_idx = _empty((max(bound, 1), 4), torch.int32, dev)
    num_out = _empty((1,), torch.int32, dev)
    ws_bytes = L.srf_rulebook_strided_workspace_bytes(A, hi(ksize), table.capacity)
    ws = _empty((max(ws_bytes, 1),), torch.uint8, dev)
    check(L.srf_rulebook_strided_outputs(_ptr(indices), A, hi(spatial_shape), batch, hi(ksize), hi(stride), hi(pad),
                                         _ptr(out_idx), _ptr(num_out), _ptr(table.buf), table.capacity, _ptr(ws),
                                         ws_bytes, _stream()), "rulebook_strided_outputs")
    A_out = int(num_out.item())
    nbr = _empty((K, max(A_out, 1)), torch.int32, dev)
    counts = _empty((K,), torch.int32, dev)
    check(L.srf_rulebook_strided_pairs(A, hi(ksize), _ptr(table.buf), table.capacity, _ptr(ws), A_out, _ptr(nbr),
                                       _ptr(counts), _stream()), "rulebook_strided_pairs")
    return out_idx[:A_out], nbr[:, :A_out], counts, table, out_spatial_shape(spatial_shape, ksize, stride, pad)


# ---------------------------------------------------------------------------------------------- bitmap-rank rulebooks
class BitmapLevel:
    """Occupancy bitmap + popcount prefix of one level's grid; valid for rows sorted by (b, y, x, z)."""

    def __init__(self, spatial_shape, batch, device):
        self.shape = [int(v) for v in spatial_shape]
        self.batch = int(batch)
        self.words = _lib.lib().srf_bitmap_words(hi(self.shape), self.batch)
        if self.words == 0:
            raise ValueError("bitmap level: grid too large or empty")
        self.bitmap = _empty((self.words,), torch.int32, device)
        self.prefix = _empty((self.words,), torch.int32, device)

    def workspace(self):
        nbytes = _lib.lib().srf_bitmap_workspace_bytes(self.words)
        return _empty((nbytes,), torch.uint8, self.bitmap.device), nbytes


def bitmap_build(indices, spatial_shape, batch, want_order=True, padded=False):
    """Distinct active sites (A,4) (b,z,y,x) -> (BitmapLevel, order, sorted_indices): row r of the sorted set is
    original row order[r].  One clear + mark + rank scan (three launches; one workgroup for <= 8192 bitmap words) + place; no host sync.  padded=True: rows with b < 0 are
    padding of a capacity-sized set; the valid rows come first in the sorted result, padding stays (-1,...)."""
    indices = _dev(indices, "indices", torch.int32)
    lvl = BitmapLevel(spatial_shape, batch, indices.device)
    A = indices.shape[0]
    order = sorted_idx = None
    if want_order:
        # padded: rows with b < 0 are padding: their sorted slots become (-1,-1,-1,-1) and point at row 0 (written by the call)
        order = _empty((max(A, 1),), torch.int32, indices.device)
        sorted_idx = _empty((max(A, 1), 4), torch.int32, indices.device)
    ws, nbytes = lvl.workspace()
    build = _lib.lib().srf_bitmap_build_padded if padded and want_order else _lib.lib().srf_bitmap_build
    check(build(_ptr(indices), A, hi(lvl.shape), lvl.batch, _ptr(lvl.bitmap), _ptr(lvl.prefix), _ptr(order), _ptr(sorted_idx), _ptr(ws),
                nbytes, _stream()), "bitmap_build")
    if not want_order:
        return lvl, None, None
    return lvl, order[:A], sorted_idx[:A]


def rulebook_subm_bitmap(sorted_indices, level, ksize, want_counts=True):
    sorted_indices = _dev(sorted_indices, "indices", torch.int32)
    A = sorted_indices.shape[0]
    K = int(np.prod(ksize))
    nbr = _empty((K, max(A, 1)), torch.int32, sorted_indices.device)
    counts = _empty((_lib.lib().srf_bitmap_pair_count_ints(),), torch.int32, sorted_indices.device) if want_counts else None
    check(_lib.lib().srf_bitmap_rulebook_subm(_ptr(sorted_indices), A, hi(level.shape), level.batch, hi(ksize), _ptr(level.bitmap),
                                              _ptr(level.prefix), _ptr(nbr), _ptr(counts), _stream()), "bitmap_rulebook_subm")
    return nbr[:, :A], (counts[:K] if want_counts else None)


def rulebook_strided_bitmap(indices, level, ksize, stride, pad, out_capacity=None):
    """-> out_indices (A_out,4) sorted by (b,y,x,z), nbr (K,A_out), pair_counts (K,), the output BitmapLevel, out_shape.
    One D2H sync for A_out.  With out_capacity the shapes are static: out_indices has out_capacity rows (padding rows are
    -1), nbr is (K, out_capacity) with -1 in the padding columns, nothing is read back, and the device scalar A_out is
    returned as a sixth value (the caller compares it with out_capacity once the frame is done)."""
    indices = _dev(indices, "indices", torch.int32)
    L = _lib.lib()
    dev = indices.device
    A = indices.shape[0]
    K = int(np.prod(ksize))
    static = out_capacity is not None
    if static:
        bound = int(out_capacity)
    else:
        bound = L.srf_strided_max_outputs(A, level.batch, hi(level.shape), hi(ksize), hi(stride), hi(pad))
        if bound < 0:
            check(bound, "strided_max_outputs")
    oshape = out_spatial_shape(level.shape, ksize, stride, pad)
    out_lvl = BitmapLevel(oshape, level.batch, dev)
    cap = max(bound, 1)
    out_idx = _empty((cap, 4), torch.int32, dev)   # static: the rows past the count are written as -1 by the call
    num_out = _empty((1,), torch.int32, dev)
    ws, nbytes = out_lvl.workspace()
    outputs = L.srf_bitmap_strided_outputs_static if static else L.srf_bitmap_strided_outputs
    check(outputs(_ptr(indices), A, hi(level.shape), level.batch, hi(ksize), hi(stride), hi(pad),
                                       _ptr(out_lvl.bitmap), _ptr(out_lvl.prefix), _ptr(out_idx), bound, _ptr(num_out), _ptr(ws),
                                       nbytes, _stream()), "bitmap_strided_outputs")
    # phase 2 reads the count on the device: it is enqueued before the host learns A_out, so the read-back below
    # overlaps it instead of leaving the GPU idle
    nbr = _empty((K, cap), torch.int32, dev)
    counts = None if static else _empty((L.srf_bitmap_pair_count_ints(),), torch.int32, dev)  # bookkeeping only
    check(L.srf_bitmap_strided_pairs(_ptr(out_idx), _ptr(num_out), bound, hi(level.shape), level.batch, hi(ksize), hi(stride),
                                     hi(pad), _ptr(level.bitmap), _ptr(level.prefix), A, _ptr(nbr), cap, int(static), _ptr(counts),
                                     _stream()), "bitmap_strided_pairs")
    if static:
        return out_idx, nbr, None, out_lvl, oshape, num_out
    A_out = int(num_out.item())
    return out_idx[:A_out], nbr[:, :A_out], counts[:K], out_lvl, oshape


# ---------------------------------------------------------------------------------------------- sparse conv
# bench.py sets this to {"spconv": []}: every sparse-conv launch is then bracketed by HIP events on the launch stream
KERNEL_TIMING = None


def pack_spconv_weights(weight):
    """(K,Cin,Cout) -> the LDS operand image srf_spconv_fwd_packed streams (once per layer; weights are constants)."""
    weight = _dev(weight, "weight", torch.float32)
    K, Cin, Cout = weight.shape
    L = _lib.lib()
    packed = _empty((L.srf_spconv_packed_weight_bytes(K, Cin, Cout) // 4,), torch.float32, weight.device)
    check(L.srf_spconv_pack_weights(_ptr(weight), K, Cin, Cout, _ptr(packed), _stream()), "spconv_pack_weights")
    return packed


def spconv_tiles_wanted(Cin, Cout):
    """True for the layer shapes whose packed kernel takes a row plan (`tiles=` of spconv_fwd): work-balanced row ranges
    (`spconv_tiles`) for the 64- / 128-channel kernels, a mask-sorted row order (`spconv_order`) for the 32-channel one."""
    return (Cout == 128 and Cin in (64, 128)) or (Cout == 64 and Cin in (32, 64)) or spconv_order_wanted(Cin, Cout)


SPCONV_ORDER_MIN_ROWS = 100000   # rows of a level from which the sorted order pays for its launch (tools/spconv_order_probe.py, MI355X:
# nuScenes level 2, 60k rows: 32 -> 32 40.2 -> 35.5 us and 16 -> 32 22.7 -> 18.7 for 15-17 us of plan build per rulebook -- a loss;
# Waymo level 2, 226k rows: 145 -> 120 us and 70 -> 45 for 20 us -- 85 us per frame gained)


def spconv_order_wanted(Cin, Cout, K=27, rows=None):
    """rows=None: is this a layer shape the plan exists for; with rows: also, is the level large enough for it to pay
    (SRF_SPCONV_ORDER=0 / 2: never / at any size)."""
    mode = os.environ.get("SRF_SPCONV_ORDER", "1")
    if not (Cout == 32 and Cin in (16, 32) and K == 27 and mode != "0"):
        return False
    return rows is None or mode == "2" or rows >= SPCONV_ORDER_MIN_ROWS


def spconv_order(nbr, rows_dev=None):
    """The plan of `srf_spconv_order_build` for the rulebook nbr (K, A_out): int32 [order | sorted rulebook].  Built once per
    rulebook and passed as `tiles=` to every 32-channel spconv_fwd(..., packed=) that uses it."""
    if not nbr.is_cuda or nbr.dtype != torch.int32 or nbr.stride(1) != 1:
        raise RuntimeError("srfdet3d_amd: `nbr` must be a GPU int32 tensor with unit inner stride")
    K, A_out = nbr.shape
    L = _lib.lib()
    plan = _empty((max(L.srf_spconv_order_ints(A_out, K), 1),), torch.int32, nbr.device)
    check(L.srf_spconv_order_build(_ptr(nbr), nbr.stride(0) if A_out > 0 else 0, K, A_out, _ptr(rows_dev), _ptr(plan), _stream()),
          "spconv_order_build")
    return plan


def spconv_tiles(nbr, rows_dev=None):
    """Row ranges of equal pair count for the rulebook nbr (K, A_out): int32 (srf_spconv_tiles_count(A_out) + 1,).
    Built once per rulebook and passed to every spconv_fwd(..., packed=, tiles=) that uses it."""
    if not nbr.is_cuda or nbr.dtype != torch.int32 or nbr.stride(1) != 1:
        raise RuntimeError("srfdet3d_amd: `nbr` must be a GPU int32 tensor with unit inner stride")
    K, A_out = nbr.shape
    L = _lib.lib()
    tiles = _empty((L.srf_spconv_tiles_count(A_out) + 1,), torch.int32, nbr.device)
    ws = _empty((max(L.srf_spconv_tiles_workspace_bytes(A_out), 4),), torch.uint8, nbr.device)
    check(L.srf_spconv_tiles_build(_ptr(nbr), nbr.stride(0) if A_out > 0 else 0, K, A_out, _ptr(rows_dev), _ptr(ws), _ptr(tiles),
                                   _stream()), "spconv_tiles_build")
    return tiles


def spconv_fwd(feats, weight, nbr, alpha=None, beta=None, residual=None, relu=False, pair_counts=None, packed=None,
               rows_dev=None, tiles=None, subm=False):
    """feats (A_in,Cin); weight (K,Cin,Cout); nbr (K,A_out) (row stride nbr.stride(0)) -> (A_out,Cout).
    `packed` = pack_spconv_weights(weight) selects the packed-weight kernel (Cout >= 32, Cin % 4 == 0);
    `tiles` = spconv_tiles(nbr) lets its 128-channel variant balance the workgroups by pair count."""
    if torch.is_grad_enabled() and (feats.requires_grad or weight.requires_grad or (residual is not None and residual.requires_grad)
                                    or (alpha is not None and alpha.requires_grad) or (beta is not None and beta.requires_grad)):
        # a gradient is wanted: plain convolution through the autograd Function (srf_spconv_bwd_data / _bwd_weight), the
        # epilogue as torch ops so that autograd differentiates it
        out = _SpconvFn.apply(feats, weight, nbr, bool(subm))
        if alpha is not None:
            out = out * alpha
        if beta is not None:
            out = out + beta
        if residual is not None:
            out = out + residual
        return torch.relu(out) if relu else out
    feats = _dev(feats, "feats", torch.float32)
    weight = _dev(weight, "weight", torch.float32)
    if not nbr.is_cuda or nbr.dtype != torch.int32 or nbr.stride(1) != 1:
        raise RuntimeError("srfdet3d_amd: `nbr` must be a GPU int32 tensor with unit inner stride")
    K, Cin, Cout = weight.shape
    A_out = nbr.shape[1]
    out = _empty((A_out, Cout), torch.float32, feats.device)
    if residual is not None:
        residual = _dev(residual, "residual", torch.float32)
    timing = KERNEL_TIMING
    if timing is not None and torch.cuda.is_current_stream_capturing():
        timing = None  # events recorded inside a graph capture are not timing events
    if timing is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    if packed is not None and Cout >= 32 and Cin % 4 == 0 and feats.shape[0] > 0:
        check(_lib.lib().srf_spconv_fwd_packed(_ptr(feats), feats.shape[0], Cin, _ptr(packed), K, _ptr(nbr),
                                               nbr.stride(0) if A_out > 0 else 0, A_out, Cout, _ptr(alpha), _ptr(beta),
                                               _ptr(residual), int(bool(relu)), _ptr(out), _ptr(rows_dev),
                                               _ptr(tiles) if spconv_tiles_wanted(Cin, Cout) else None, _stream()),
              "spconv_fwd_packed")
    else:
        check(_lib.lib().srf_spconv_fwd(_ptr(feats), feats.shape[0], Cin, _ptr(weight), K, _ptr(nbr),
                                        nbr.stride(0) if A_out > 0 else 0, A_out, Cout,
                                        _ptr(alpha), _ptr(beta), _ptr(residual), int(bool(relu)), _ptr(out), _ptr(rows_dev),
                                        _stream()), "spconv_fwd")
    if timing is not None:
        ev1.record()
        timing["spconv"].append(_SpconvRecord(ev0, ev1, Cin, Cout, K, feats.shape[0], A_out, pair_counts, residual is not None))
    return out


class _SpconvRecord:
    """(start, end, Cin, Cout, K, algorithmic flops, algorithmic bytes) of one launch; the pair count is read from the
    rulebook's device counters only when the record is unpacked, after the timed region."""

    def __init__(self, ev0, ev1, cin, cout, K, a_in, a_out, pair_counts, has_residual=False):
        self.v = (ev0, ev1, cin, cout, K)
        self.a_in, self.a_out, self.pair_counts, self.has_residual = a_in, a_out, pair_counts, has_residual

    def __iter__(self):
        ev0, ev1, cin, cout, K = self.v
        pairs = int(self.pair_counts.sum().item()) if self.pair_counts is not None else 0
        flops = 2 * pairs * cin * cout
        byts = 4 * (self.a_in * cin + self.a_out * cout) + 4 * K * cin * cout + 8 * pairs
        if self.has_residual:
            byts += 4 * self.a_out * cout  # the residual rows are read once
        return iter((ev0, ev1, cin, cout, K, flops, byts))


def densify(feats, indices, batch, spatial_shape):
    if torch.is_grad_enabled() and feats.requires_grad:
        return _DensifyFn.apply(feats, indices, batch, tuple(spatial_shape))
    return _densify(feats, indices, batch, spatial_shape)


def densify_bev(feats, level, batch, spatial_shape):
    """sorted rows (A, C) + their BitmapLevel -> the BEV map (batch, C * D, H, W) with CHANNELS-LAST strides: dense() + the
    (N, C, D, H, W) -> (N, C * D, H, W) view in one pass (no zero fill, no transpose; srf_densify_bev).  Inference only."""
    feats = _dev(feats, "feats", torch.float32)
    A, C = feats.shape
    D, H, W = [int(v) for v in spatial_shape]
    if [int(v) for v in level.shape] != [D, H, W] or level.batch != batch:
        raise ValueError("densify_bev: the bitmap level belongs to another grid")
    out = torch.empty((batch, C * D, H, W), dtype=torch.float32, device=feats.device, memory_format=torch.channels_last)
    check(_lib.lib().srf_densify_bev(_ptr(feats), A, C, _ptr(level.bitmap), _ptr(level.prefix), batch, D, H, W, _ptr(out), _stream()),
          "densify_bev")
    return out


def _densify(feats, indices, batch, spatial_shape):
    feats = _dev(feats, "feats", torch.float32)
    indices = _dev(indices, "indices", torch.int32)
    A, C = feats.shape
    D, H, W = spatial_shape
    out = _empty((batch, C, D, H, W), torch.float32, feats.device)
    check(_lib.lib().srf_densify(_ptr(feats), _ptr(indices), A, C, batch, D, H, W, _ptr(out), 1, _stream()), "densify")
    return out


# ---------------------------------------------------------------------------------------------- gradients of the sparse half
def spconv_transpose_rulebook(nbr, a_in):
    """(K, A_out) output-stationary table -> (K, A_in): nbrT[k][i] = o <=> nbr[k][o] = i."""
    K, A_out = nbr.shape
    nbrT = _empty((K, max(a_in, 1)), torch.int32, nbr.device)
    check(_lib.lib().srf_spconv_transpose_rulebook(_ptr(nbr), nbr.stride(0) if A_out > 0 else 0, K, A_out, _ptr(nbrT), a_in,
                                                   _stream()), "spconv_transpose_rulebook")
    return nbrT[:, :a_in]


def _spconv_plain(feats, weight, nbr):
    """out[o] = sum_k W[k]^T in[nbr[k][o]] without epilogue; output widths below 16 run zero-padded on the 16-wide kernel."""
    K, Cin, Cout = weight.shape
    pad = 0
    if Cout not in (16, 32, 64, 128):
        if Cout > 16:
            raise RuntimeError(f"srfdet3d spconv: no kernel for {Cout} output channels")
        pad = 16 - Cout
        weight = torch.nn.functional.pad(weight, (0, pad))
    with torch.no_grad():
        out = spconv_fwd(feats.detach().contiguous(), weight.detach().contiguous(), nbr)
    return out[:, :Cout].contiguous() if pad else out


class _SpconvFn(torch.autograd.Function):
    """Sparse convolution with gradients: data gradient = the forward kernel on the transposed rulebook with transposed
    weights, weight gradient = srf_spconv_bwd_weight (csrc/spconv_bwd.hip)."""

    @staticmethod
    def forward(ctx, feats, weight, nbr, subm):
        ctx.save_for_backward(feats, weight, nbr)
        ctx.subm = subm
        return _spconv_plain(feats, weight, nbr)

    @staticmethod
    def backward(ctx, g):
        feats, weight, nbr = ctx.saved_tensors
        g = g.contiguous()
        K, Cin, Cout = weight.shape
        gf = gw = None
        if ctx.needs_input_grad[0]:
            wt = weight.detach().transpose(1, 2)
            if ctx.subm:   # symmetric rulebook: offset k of the transpose is offset K - 1 - k of the table itself
                nbrT, wt = nbr, wt.flip(0)
            else:
                nbrT = spconv_transpose_rulebook(nbr, feats.shape[0])
            gf = _spconv_plain(g, wt.contiguous(), nbrT)
        if ctx.needs_input_grad[1]:
            gw = _empty((K, Cin, Cout), torch.float32, g.device)
            f = feats.detach().contiguous()
            check(_lib.lib().srf_spconv_bwd_weight(_ptr(f), f.shape[0], Cin, _ptr(g), g.shape[0], Cout, _ptr(nbr),
                                                   nbr.stride(0) if nbr.shape[1] > 0 else 0, K, _ptr(gw), _stream()),
                  "spconv_bwd_weight")
        return gf, gw, None, None


class _DensifyFn(torch.autograd.Function):
    """SparseConvTensor.dense(): scatter forward, gather backward."""

    @staticmethod
    def forward(ctx, feats, indices, batch, spatial_shape):
        ctx.save_for_backward(indices)
        return _densify(feats.detach(), indices, batch, spatial_shape)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        i = idx.long()
        ok = (i[:, 0] >= 0).unsqueeze(1)
        i = i.clamp(min=0)
        return g[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3]] * ok, None, None, None


class _ScatterReduceFn(torch.autograd.Function):
    """DynamicScatter mean / max over the points of a voxel.  mean: every point receives d_voxel / count; max: the gradient
    of a (voxel, channel) goes to the first point (in input order) that attains the maximum."""

    @staticmethod
    def forward(ctx, feats, vmap, mode):
        out = vmap._reduce(feats.detach(), mode)
        ctx.vmap, ctx.mode = vmap, mode
        ctx.save_for_backward(feats, out)
        return out

    @staticmethod
    def backward(ctx, g):
        feats, out = ctx.saved_tensors
        vm = ctx.vmap
        p2v = vm.point2voxel.long()
        ok = p2v >= 0
        pv = p2v.clamp(min=0)
        if ctx.mode == "mean":
            cnt = vm.counts[:vm.M].clamp(min=1).to(g.dtype).unsqueeze(1)
            return (g / cnt)[pv] * ok.unsqueeze(1), None, None
        n, C = feats.shape
        hit = (feats.detach() == out[pv]) & ok.unsqueeze(1)
        cand = torch.where(hit, torch.arange(n, device=g.device).unsqueeze(1).expand(n, C), torch.full((1, 1), n, device=g.device))
        first = torch.full((vm.M, C), n, dtype=torch.long, device=g.device).scatter_reduce(0, pv.unsqueeze(1).expand(n, C), cand, "amin")
        sel = hit & (first[pv] == torch.arange(n, device=g.device).unsqueeze(1))
        return g[pv] * sel, None, None


# ---------------------------------------------------------------------------------------------- RoI gather
def _featmap(t, scale):
    """4-D tensor in any dense layout (NCHW or channels-last) -> srf_featmap."""
    if t.dim() != 4 or not t.is_cuda or t.dtype != torch.float32:
        raise RuntimeError("srfdet3d_amd: feature maps must be 4-D float32 GPU tensors")
    sn, sc, sh, sw = t.stride()
    return FeatMap(t.data_ptr(), t.shape[0], t.shape[2], t.shape[3], sn, sc, sh, sw, float(scale))


def roi_extract(feats, rois, strides, out_size=7, sampling_ratio=2, finest_scale=56.0, out=None, accumulate=False,
                bin_major=False, return_levels=False, n_sum=1):
    """SingleRoIExtractor over len(strides) maps.  out: (R,C,out_size,out_size), or (R,out_size^2,C) if bin_major; a given `out` may
    be a strided view (e.g. a channel slice of a wider buffer).  n_sum > 1: `rois` holds n_sum groups of R RoIs (group-major) and
    row r of the result is the sum of the gathers of rois[s * R + r], s ascending (`srf_roi_extract_sum`: the camera sum of
    srfdet_head.py:2543-2562 in the gather itself)."""
    rois = _dev(rois, "rois", torch.float32)
    if rois.shape[0] % n_sum:
        raise ValueError("roi_extract: the number of RoIs must be a multiple of n_sum")
    R = rois.shape[0] // n_sum
    C = feats[0].shape[1]
    nl = len(strides)
    fm = (FeatMap * nl)(*[_featmap(f, 1.0 / s) for f, s in zip(feats, strides)])
    bins = out_size * out_size
    shape = (R, bins, C) if bin_major else (R, C, out_size, out_size)
    if out is None:
        out = _empty(shape, torch.float32, rois.device)
        accumulate = False
    elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_cuda:
        raise ValueError(f"roi_extract: out must be a float32 GPU tensor of shape {shape}")
    if bin_major:
        so_r, so_b, so_c = out.stride(0) if R > 1 else bins * C, out.stride(1), out.stride(2) if C > 1 else 1
    else:
        if out.stride(3) != 1 or out.stride(2) != out_size:
            raise ValueError("roi_extract: the bins of a channel must be contiguous in `out`")
        so_r, so_c, so_b = out.stride(0) if R > 1 else C * bins, out.stride(1) if C > 1 else bins, 1
    if n_sum > 1:
        if accumulate or return_levels:
            raise ValueError("roi_extract: n_sum excludes accumulate / return_levels")
        check(_lib.lib().srf_roi_extract_sum(fm, nl, C, _ptr(rois), R, n_sum, out_size, sampling_ratio, float(finest_scale), _ptr(out),
                                             so_r, so_c, so_b, _stream()), "roi_extract_sum")
        return out
    lv = _empty((max(R, 1),), torch.int32, rois.device) if return_levels else None
    check(_lib.lib().srf_roi_extract(fm, nl, C, _ptr(rois), R, out_size, sampling_ratio, float(finest_scale), _ptr(out),
                                     so_r, so_c, so_b, int(bool(accumulate)), _ptr(lv), _stream()), "roi_extract")
    return (out, lv[:R]) if return_levels else out


def box_rois(boxes, pc_range, voxel_size, mutate_centres=True, want_bev=True, lidar2img=None):
    """boxes (B,P,>=8) normalised centres -> rois_bev (B*P,5) and/or rois_img (n_cam*B*P,5)."""
    boxes = _dev(boxes, "boxes", torch.float32)
    B, P, D = boxes.shape
    dev = boxes.device
    rb = _empty((B * P, 5), torch.float32, dev) if want_bev else None
    ri, n_cam = None, 0
    if lidar2img is not None:
        lidar2img = _dev(lidar2img, "lidar2img", torch.float32)
        n_cam = lidar2img.shape[1]
        ri = _empty((n_cam * B * P, 5), torch.float32, dev)
    check(_lib.lib().srf_box_rois(_ptr(boxes), B, P, D, hf(pc_range), hf(voxel_size), int(bool(mutate_centres)), _ptr(rb),
                                  _ptr(lidar2img), n_cam, _ptr(ri), _stream()), "box_rois")
    return rb, ri


# ---------------------------------------------------------------------------------------------- NMS
def nms_rotated(boxes_xywhr, scores, iou_threshold, classes=None):
    """Greedy rotated NMS; returns indices (into the input) of the kept boxes, in descending score order.  classes (n,)
    int64: a box only suppresses boxes of its own class (per-class NMS in one pass)."""
    boxes_xywhr = _dev(boxes_xywhr, "boxes", torch.float32)
    n = boxes_xywhr.shape[0]
    if n == 0:
        return torch.zeros((0,), dtype=torch.long, device=boxes_xywhr.device)
    order = scores.sort(0, descending=True)[1]
    sorted_boxes = boxes_xywhr[order].contiguous()
    L = _lib.lib()
    keep = _empty((n,), torch.int32, boxes_xywhr.device)
    ws_bytes = L.srf_nms_rotated_workspace_bytes(n)
    ws = _empty((ws_bytes,), torch.uint8, boxes_xywhr.device)
    if classes is None:
        check(L.srf_nms_rotated(_ptr(sorted_boxes), n, float(iou_threshold), _ptr(keep), _ptr(ws), ws_bytes, _stream()),
              "nms_rotated")
    else:
        cls = _dev(classes, "classes", torch.int64)[order].contiguous()
        check(L.srf_nms_rotated_classes(_ptr(sorted_boxes), _ptr(cls), n, None, float(iou_threshold), _ptr(keep), _ptr(ws), ws_bytes,
                                        _stream()), "nms_rotated_classes")
    return order[keep.bool()]


def nms_rotated_counted(sorted_boxes_xywhr, n_live, iou_threshold, classes=None):
    """Greedy rotated NMS over the first n_live (device int32 scalar) rows of boxes already in descending score order;
    returns keep flags (n,) int32, zero past n_live.  Fixed shapes, nothing read back: graph-capturable.  classes (n,) int64:
    suppression only inside a class."""
    b = _dev(sorted_boxes_xywhr, "boxes", torch.float32)
    n = b.shape[0]
    keep = _empty((max(n, 1),), torch.int32, b.device)
    if n == 0:
        return keep[:0]
    L = _lib.lib()
    ws_bytes = L.srf_nms_rotated_workspace_bytes(n)
    ws = _empty((ws_bytes,), torch.uint8, b.device)
    if classes is None:
        check(L.srf_nms_rotated_counted(_ptr(b), n, _ptr(_dev(n_live, "n_live", torch.int32)), float(iou_threshold), _ptr(keep),
                                        _ptr(ws), ws_bytes, _stream()), "nms_rotated_counted")
    else:
        cls = _dev(classes, "classes", torch.int64)
        if cls.shape != (n,) or not cls.is_contiguous():
            raise ValueError("nms_rotated_counted: classes must be a contiguous (n,) int64 tensor")
        check(L.srf_nms_rotated_classes(_ptr(b), _ptr(cls), n, _ptr(_dev(n_live, "n_live", torch.int32)), float(iou_threshold),
                                        _ptr(keep), _ptr(ws), ws_bytes, _stream()), "nms_rotated_classes")
    return keep


def nms_select(boxes, scores, score_thr, capacity):
    """(n, D) boxes, (n, C) scores -> the L = min(n*C, capacity) best (box, class) pairs above score_thr in descending score:
    cand (L, D), top_s (L,), cls (L,) int64, bev (L, 5) for the rotated NMS, m (1,) int32 = pairs above the threshold.
    One single-workgroup launch (pairs above the threshold compacted, rank-sorted in LDS when <= 1024 of them, else a bitonic sort of
    all pairs), nothing read back: graph-capturable.  n*C <= 16384."""
    boxes = _dev(boxes, "boxes", torch.float32).contiguous()
    scores = _dev(scores, "scores", torch.float32).contiguous()
    n, D = boxes.shape
    C = scores.shape[1]
    L = min(n * C, int(capacity))
    dev = boxes.device
    cand, top_s = _empty((L, D), torch.float32, dev), _empty((L,), torch.float32, dev)
    cls, bev, m = _empty((L,), torch.int64, dev), _empty((L, 5), torch.float32, dev), _empty((1,), torch.int32, dev)
    check(_lib.lib().srf_nms_select(_ptr(boxes), _ptr(scores), n, C, D, float(score_thr), L, _ptr(cand), _ptr(top_s), _ptr(cls),
                                    _ptr(bev), _ptr(m), _stream()), "nms_select")
    return cand, top_s, cls, bev, m


def nms_finish(cand, top_s, cls, keep, m=None):
    """keep flags of the NMS over nms_select's candidates -> (boxes (L, D), scores (L,), labels (L,) int64, kept (1,) int32):
    survivors first, class-major, descending score inside a class.  With m (nms_select's count) also the packed form of the
    same rows, (L, D+2) [box, score, label], and counts (2,) int32 [kept, m]: one D2H copy per frame."""
    L, D = cand.shape
    dev = cand.device
    ob, os_, ol, kept = (_empty((L, D), torch.float32, dev), _empty((L,), torch.float32, dev), _empty((L,), torch.int64, dev),
                         _empty((1,), torch.int32, dev))
    packed = _empty((L, D + 2), torch.float32, dev) if m is not None else None
    counts = _empty((2,), torch.int32, dev) if m is not None else None
    check(_lib.lib().srf_nms_finish(_ptr(cand), _ptr(top_s), _ptr(cls), _ptr(_dev(keep, "keep", torch.int32)), L, D, _ptr(ob),
                                    _ptr(os_), _ptr(ol), _ptr(kept), _ptr(packed), _ptr(m), _ptr(counts), _stream()), "nms_finish")
    if m is not None:
        return ob, os_, ol, kept, packed, counts
    return ob, os_, ol, kept


# ---------------------------------------------------------------------------------------------- decoder stage
def _ln(ln):
    """nn.LayerNorm or (gamma, beta, eps) or None -> (ptr_g, ptr_b, eps)."""
    if ln is None:
        return None, None, 0.0
    if isinstance(ln, tuple):
        return _ptr(ln[0]), _ptr(ln[1]), float(ln[2])
    return _ptr(ln.weight), _ptr(ln.bias), float(ln.eps)


def linear(x, weight, bias=None, ln1=None, relu1=False, residual=None, ln2=None, relu2=False):
    """Y = [LN2]([residual +] relu1?([LN1](x @ weight.T + bias))), then relu2? -- one launch (two when K >= 2048 or N > 128
    with a row epilogue).  x (M,K), weight (N,K), both with unit inner stride."""
    x = _dev(x, "x", torch.float32)
    if not weight.is_cuda or weight.stride(1) != 1:
        raise RuntimeError("srfdet3d_amd: `weight` must be a GPU tensor with unit inner stride")
    M, K = x.shape
    N = weight.shape[0]
    L = _lib.lib()
    y = _empty((M, N), torch.float32, x.device)
    if residual is not None:
        residual = _dev(residual, "residual", torch.float32)
    need_ws = K >= 2048 or ((ln1 is not None or ln2 is not None or residual is not None or relu2) and N > 128)
    ws_bytes = L.srf_linear_workspace_bytes(M, N, K) if need_ws else 0
    ws = _empty((ws_bytes,), torch.uint8, x.device) if ws_bytes else None
    g1, b1, e1 = _ln(ln1)
    g2, b2, e2 = _ln(ln2)
    check(L.srf_linear(_ptr(x), M, K, x.stride(0), _ptr(weight), N, weight.stride(0), _ptr(bias), g1, b1, e1, int(bool(relu1)),
                       _ptr(residual), residual.stride(0) if residual is not None else 0, g2, b2, e2, int(bool(relu2)),
                       _ptr(y), N, _ptr(ws), ws_bytes, _stream()), "linear")
    return y


def self_attention(qkv, num_heads, out=None, batch=1):
    """qkv (batch * P, 3E) rows [q|k|v], sample-major -> (batch * P, E): softmax(q k^T / sqrt(d)) v per head among the P rows of
    each sample, one launch for the whole batch.  out: a contiguous (batch * P, E) tensor written in place."""
    qkv = _dev(qkv, "qkv", torch.float32)
    R, E3 = qkv.shape
    E = E3 // 3
    if batch < 1 or R % batch:
        raise ValueError("self_attention: rows must be batch * P")
    if out is None:
        out = _empty((R, E), torch.float32, qkv.device)
    elif tuple(out.shape) != (R, E) or not out.is_contiguous() or out.dtype != torch.float32 or out.device != qkv.device:
        raise ValueError("self_attention: out must be a contiguous float32 (rows, E) tensor on the input's device")
    check(_lib.lib().srf_self_attention_batched(_ptr(qkv), batch, R // batch, E, num_heads, _ptr(out), _stream()), "self_attention")
    return out


def dynconv_mid(feats, params, ln1, ln2):
    """feats (R,S,C), params (R, 2*C*D) -> relu(LN_C(relu(LN_D(feats @ W1)) @ W2)), (R,S,C)."""
    feats = _dev(feats, "feats", torch.float32)
    params = _dev(params, "params", torch.float32)
    R, S, C = feats.shape
    D = params.shape[1] // (2 * C)
    out = _empty((R, S, C), torch.float32, feats.device)
    g1, b1, e1 = _ln(ln1)
    g2, b2, e2 = _ln(ln2)
    check(_lib.lib().srf_dynconv_mid(_ptr(feats), _ptr(params), R, S, C, D, g1, b1, e1, g2, b2, e2, _ptr(out), _stream()),
          "dynconv_mid")
    return out


def apply_deltas(deltas, boxes, weights6, pc_range, scale_clamp):
    deltas = _dev(deltas, "deltas", torch.float32)
    boxes = _dev(boxes, "boxes", torch.float32)
    R, Dd = deltas.shape
    out = _empty((R, Dd), torch.float32, deltas.device)
    check(_lib.lib().srf_apply_deltas(_ptr(deltas), _ptr(boxes), R, Dd, hf(weights6), hf(pc_range), float(scale_clamp),
                                      _ptr(out), _stream()), "apply_deltas")
    return out


def host_pack(packed, counts, level_counts):
    """float32 vector [packed.flatten(), counts, level_counts] (the ints converted; all < 2^24) in one launch."""
    a = _dev(packed, "packed", torch.float32)
    b = _dev(counts, "counts", torch.int32)
    c = _dev(level_counts, "level_counts", torch.int32)
    out = _empty((a.numel() + b.numel() + c.numel(),), torch.float32, a.device)
    check(_lib.lib().srf_host_pack(_ptr(a), a.numel(), _ptr(b), b.numel(), _ptr(c), c.numel(), _ptr(out), _stream()), "host_pack")
    return out


def decode_boxes(logits, pred, pc_range):
    """last-stage logits (..., ncls) and boxes (..., Dd) with normalised centres -> (scores, boxes (..., Dd - 1) in metres with
    bottom-centre z): the end of `forward` + SRFDetHead.decode (srfdet_head.py:1002-1006, :1246-1271) as one launch."""
    logits = _dev(logits, "logits", torch.float32)
    pred = _dev(pred, "pred", torch.float32)
    lead, ncls, Dd = logits.shape[:-1], logits.shape[-1], pred.shape[-1]
    R = int(np.prod(lead)) if len(lead) else 1
    scores = _empty((*lead, ncls), torch.float32, logits.device)
    boxes = _empty((*lead, Dd - 1), torch.float32, logits.device)
    check(_lib.lib().srf_decode_boxes(_ptr(logits), _ptr(pred), R, ncls, Dd, hf(pc_range), _ptr(scores), _ptr(boxes), _stream()),
          "decode_boxes")
    return scores, boxes


def upsample_add_supported(lateral, top):
    return (lateral.is_cuda and lateral.dtype == torch.float32 and top.dtype == torch.float32 and lateral.dim() == 4
            and lateral.shape[:2] == top.shape[:2] and lateral.shape[3] % 4 == 0 and lateral.is_contiguous() and top.is_contiguous()
            and not (torch.is_grad_enabled() and (lateral.requires_grad or top.requires_grad)))


def upsample_add(lateral, top):
    """lateral + F.interpolate(top, size=lateral.shape[2:], mode='nearest') in one pass (the FPN top-down step)."""
    lateral = _dev(lateral, "lateral", torch.float32)
    top = _dev(top, "top", torch.float32)
    N, C, H, W = lateral.shape
    out = _empty((N, C, H, W), torch.float32, lateral.device)
    check(_lib.lib().srf_upsample_add(_ptr(lateral), _ptr(top), N * C, H, W, top.shape[2], top.shape[3], _ptr(out), _stream()),
          "upsample_add")
    return out


def dwconv3x3s2(x, weight, scale=None, shift=None, relu=False):
    """Depthwise 3x3 / stride 2 / padding 1 convolution (weight (C, 1, 3, 3)) + per-channel scale/shift + ReLU."""
    x = _dev(x, "x", torch.float32).contiguous()
    N, C, H, W = x.shape
    w = _dev(weight, "weight", torch.float32).reshape(C, 9).contiguous()
    y = _empty((N, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), torch.float32, x.device)
    check(_lib.lib().srf_dwconv3x3s2(_ptr(x), N, C, H, W, _ptr(w), _ptr(scale), _ptr(shift), int(bool(relu)), _ptr(y), _stream()),
          "dwconv3x3s2")
    return y


def to_channels_last(x):
    """x.contiguous(memory_format=torch.channels_last) for a contiguous NCHW f32 GPU tensor with H*W % 4 == 0 (anything
    else goes through torch): same values, same strides, a faster transposing copy."""
    if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous() and (x.shape[2] * x.shape[3]) % 4 == 0
            and x.shape[1] > 1 and not (torch.is_grad_enabled() and x.requires_grad)):
        N, C, H, W = x.shape
        y = torch.empty((N, C, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        check(_lib.lib().srf_nchw_to_nhwc(_ptr(x), N, C, H * W, _ptr(y), _stream()), "nchw_to_nhwc")
        return y
    return x.contiguous(memory_format=torch.channels_last)


def ese_gate(mean, weight, bias):
    """(N, C) global averages, (C, C[, 1, 1]) fc weight, (C,) bias -> hsigmoid(fc(mean)) (N, C): VoVNet's eSE gate."""
    mean = _dev(mean, "mean", torch.float32).contiguous()
    N, C = mean.shape
    w = _dev(weight, "weight", torch.float32).reshape(C, C).contiguous()
    gate = _empty((N, C), torch.float32, mean.device)
    for n0 in range(0, N, 8):   # the GEMV kernel takes up to 8 rows per launch (6 cameras; 12 images at batch size 2)
        n1 = min(N, n0 + 8)
        check(_lib.lib().srf_ese_gate(_ptr(mean[n0:n1]), n1 - n0, C, _ptr(w), _ptr(bias), _ptr(gate[n0:n1]), _stream()), "ese_gate")
    return gate


def ese_apply_supported(x, C):
    """Opt-in (SRF_ESE_FUSED=1): measured on the LC frame the fused launch LOSES 0.34 frames/s (29.27 -> 28.93, same box, three alternating
    runs) -- a workgroup's 64-channel slices stream worse than srf_nhwc_affine's linear pass and every workgroup starts with ~3 us of
    gate arithmetic; the separate gate GEMV (14 launches of ~14 us on the camera graph) stays the default."""
    return C % 64 == 0 and C <= 1024 and x.shape[0] <= 65535 and os.environ.get("SRF_ESE_FUSED", "0") == "1"


def ese_apply(x, mean, weight, bias, residual=None, out=None, want_gate=False):
    """VoVNet's eSE gate + its application in one launch: out = x * hsigmoid(fc(mean)) (+ residual) on NHWC slices (N, H, W, C);
    the same bits as `ese_gate` followed by `nhwc_affine(x, scale=gate, residual=...)`.  -> out (, gate (N, C))."""
    x_ld = nhwc_ld(x)
    N, H, W, C = x.shape
    mean = _dev(mean, "mean", torch.float32).contiguous()
    w = _dev(weight, "weight", torch.float32).reshape(C, C).contiguous()
    if out is None:
        out = _empty((N, H, W, C), torch.float32, x.device)
    gate = _empty((N, C), torch.float32, x.device) if want_gate else None
    check(_lib.lib().srf_ese_apply(_ptr(x), x_ld, N, H * W, C, _ptr(mean), _ptr(w), _ptr(bias), _ptr(residual),
                                   nhwc_ld(residual) if residual is not None else 0, _ptr(out), nhwc_ld(out), _ptr(gate), _stream()),
          "ese_apply")
    return (out, gate) if want_gate else out


def maxpool3s2_ceil(x):
    """nn.MaxPool2d(3, stride=2, ceil_mode=True) on a contiguous NCHW f32 tensor."""
    x = _dev(x, "x", torch.float32).contiguous()
    N, C, H, W = x.shape

    def osz(n):
        o = (n - 3 + 1) // 2 + 1 if n >= 3 else 1
        return o - 1 if (o - 1) * 2 >= n else o
    y = _empty((N, C, osz(H), osz(W)), torch.float32, x.device)
    check(_lib.lib().srf_maxpool3s2_ceil(_ptr(x), N * C, H, W, _ptr(y), _stream()), "maxpool3s2_ceil")
    return y


def channel_affine(x, scale, shift, relu, out=None, residual=None):
    """y = x * scale + shift (+ residual) (+ ReLU) on a contiguous NCHW tensor; scale / shift hold C values (per
    channel) or N*C values (per sample and channel); shift may be None.  `out` may be x itself or a channel slice of a
    wider contiguous NCHW tensor (same N, H, W)."""
    x = _dev(x, "x", torch.float32)
    N, C = x.shape[0], x.shape[1]
    HW = x[0, 0].numel()
    if out is None:
        out = torch.empty_like(x)
    if out.shape != x.shape or out.dtype != torch.float32 or out.device != x.device:
        raise ValueError("channel_affine: out must match x")
    if HW and (out.stride(1) != HW or not out[0, 0].is_contiguous()):
        raise ValueError("channel_affine: out must be NCHW with contiguous channel planes")
    y_sn = out.stride(0) if N > 1 else C * HW
    if scale.numel() not in (C, N * C) or (shift is not None and shift.numel() != scale.numel()):
        raise ValueError("channel_affine: scale/shift must hold C or N*C values")
    per_sample = int(scale.numel() != C)
    if residual is not None:
        residual = _dev(residual, "residual", torch.float32)
        if residual.shape != x.shape:
            raise ValueError("channel_affine: residual must match x")
    check(_lib.lib().srf_channel_affine(_ptr(x), N, C, HW, C * HW, _ptr(_dev(scale, "scale", torch.float32)),
                                        None if shift is None else _ptr(_dev(shift, "shift", torch.float32)), per_sample,
                                        None if residual is None else _ptr(residual), int(bool(relu)), _ptr(out),
                                        max(y_sn, C * HW), _stream()), "channel_affine")
    return out


def pack_conv1x1_weights(weight):
    """(Cout, K) or (Cout, K, 1, 1) -> the per-lane MFMA operand order srf_conv1x1 streams (once per layer)."""
    weight = _dev(weight.reshape(weight.shape[0], -1), "weight", torch.float32)
    Cout, K = weight.shape
    L = _lib.lib()
    nbytes = L.srf_conv1x1_packed_weight_bytes(Cout, K)
    if nbytes == 0:
        raise ValueError("conv1x1: Cout and K must be multiples of 32")
    packed = _empty((nbytes // 4,), torch.float32, weight.device)
    check(L.srf_conv1x1_pack_weights(_ptr(weight), Cout, K, _ptr(packed), _stream()), "conv1x1_pack_weights")
    return packed


def conv1x1_supported(xs, Cout):
    """Shapes srf_conv1x1 takes: contiguous NCHW f32 sources of equal N, H, W, channels % 32 == 0, H*W % 4 == 0."""
    x0 = xs[0]
    hw = x0.shape[2] * x0.shape[3]
    return (Cout % 128 == 0 and hw % 4 == 0 and hw >= 4 and 0 < len(xs) <= 8
            and all(x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous() and x.shape[1] % 32 == 0
                    and x.shape[0] == x0.shape[0] and x.shape[2:] == x0.shape[2:] for x in xs))


def conv1x1(xs, packed_weight, Cout, scale=None, shift=None, relu=False):
    """1x1 convolution of cat(xs, dim=1) (never materialised) + per-channel scale/shift + ReLU -> (N, Cout, H, W)."""
    import ctypes
    if not conv1x1_supported(xs, Cout):
        raise ValueError("conv1x1: unsupported shapes (see conv1x1_supported)")
    N, _, H, W = xs[0].shape
    out = _empty((N, Cout, H, W), torch.float32, xs[0].device)
    ptrs = (ctypes.c_void_p * len(xs))(*[x.data_ptr() for x in xs])
    chans = (ctypes.c_int * len(xs))(*[x.shape[1] for x in xs])
    K = sum(x.shape[1] for x in xs)
    if packed_weight.numel() != Cout * K:
        raise ValueError("conv1x1: packed weight does not match the sources")
    check(_lib.lib().srf_conv1x1(ptrs, chans, len(xs), N, H * W, _ptr(packed_weight), Cout,
                                 None if scale is None else _ptr(_dev(scale, "scale", torch.float32)),
                                 None if shift is None else _ptr(_dev(shift, "shift", torch.float32)), int(bool(relu)),
                                 _ptr(out), _stream()), "conv1x1")
    return out


# ---- channels-last (NHWC) dense convolutions (csrc/conv.hip) ---------------------------------------------------------
def _dense_timing(key):
    """(start event, record list) when bench.py asked for per-launch times of the dense kernels (KERNEL_TIMING[key]), else None."""
    t = KERNEL_TIMING
    if t is None or key not in t or torch.cuda.is_current_stream_capturing():
        return None
    ev0 = torch.cuda.Event(enable_timing=True)
    ev0.record()
    return ev0, t[key]


def nhwc_ld(x):
    """Floats per pixel of an (N, H, W, C) f32 view whose channels are a slice of a pixel-major buffer; raises if the
    view is not of that form."""
    if x.dim() != 4 or x.dtype != torch.float32 or not x.is_cuda:
        raise RuntimeError("srfdet3d: expected a 4-d f32 GPU tensor (N, H, W, C)")
    N, H, W, C = x.shape
    ld = x.stride(2) if W > 1 else (x.stride(1) if H > 1 else max(C, 1))
    ok = (C == 1 or x.stride(3) == 1) and ld >= C and (W == 1 or x.stride(2) == ld) and (H == 1 or x.stride(1) == W * ld) \
        and (N == 1 or x.stride(0) == H * W * ld)
    if not ok:
        raise RuntimeError(f"srfdet3d: tensor of shape {tuple(x.shape)} / strides {x.stride()} is not a channel slice of an NHWC buffer")
    return ld


def pack_wino3x3_weights(weight):
    """(Cout, Cin, 3, 3) -> G g G^T in the LDS operand order srf_wino3x3 copies (once per layer)."""
    weight = _dev(weight, "weight", torch.float32)
    Cout, Cin, kh, kw = weight.shape
    L = _lib.lib()
    nbytes = L.srf_wino3x3_packed_weight_bytes(Cout, Cin)
    if (kh, kw) != (3, 3) or nbytes == 0:
        raise ValueError("wino3x3: needs a (Cout, Cin, 3, 3) weight with Cin % 8 == 0")
    packed = _empty((nbytes // 4,), torch.float32, weight.device)
    check(L.srf_wino3x3_pack_weights(_ptr(weight), Cout, Cin, _ptr(packed), _stream()), "wino3x3_pack_weights")
    return packed


def wino3x3_supported(x):
    return (x.dim() == 4 and x.is_cuda and x.dtype == torch.float32 and x.shape[3] % 8 == 0 and x.data_ptr() % 16 == 0
            and x.stride(2) % 4 == 0 and 4 * x.shape[1] * x.shape[2] * x.stride(2) < (1 << 30))


def wino3x3(x, packed_weight, Cout, scale=None, shift=None, relu=False, out=None):
    """3x3 / stride 1 / padding 1 convolution of the NHWC slice x (N, H, W, Cin) + per-channel scale / shift + ReLU into
    `out` (an (N, H, W, Cout) slice of an NHWC buffer; a new contiguous tensor when None)."""
    x_ld = nhwc_ld(x)
    N, H, W, Cin = x.shape
    if out is None:
        out = _empty((N, H, W, Cout), torch.float32, x.device)
    elif tuple(out.shape) != (N, H, W, Cout):
        raise ValueError("wino3x3: out has the wrong shape")
    y_ld = nhwc_ld(out)
    L = _lib.lib()
    if packed_weight.numel() * 4 != L.srf_wino3x3_packed_weight_bytes(Cout, Cin):
        raise ValueError("wino3x3: packed weight does not match (Cout, Cin)")
    timing = _dense_timing("wino")
    check(L.srf_wino3x3(_ptr(x), N, H, W, Cin, x_ld, _ptr(packed_weight), Cout,
                        None if scale is None else _ptr(_dev(scale, "scale", torch.float32)),
                        None if shift is None else _ptr(_dev(shift, "shift", torch.float32)), int(bool(relu)),
                        _ptr(out), y_ld, _stream()), "wino3x3")
    if timing is not None:
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record()
        # direct-convolution FLOPs, FLOPs the Winograd kernel executes on the MFMA (16 of 36 products), bytes in + out + weights
        direct = 2.0 * 9 * Cin * Cout * N * H * W
        timing[1].append((timing[0], ev1, f"{Cin}->{Cout} @{N}x{H}x{W}", direct, direct / 2.25,
                          4.0 * N * H * W * (Cin + Cout) + 4.0 * 16 * Cin * Cout))
    return out


def pack_wino43_weights(weight):
    """(Cout, Cin, 3, 3) -> U = G g G^T of Winograd F(4x4, 3x3) in the operand order srf_wino43 streams (once per layer)."""
    weight = _dev(weight, "weight", torch.float32)
    Cout, Cin, kh, kw = weight.shape
    L = _lib.lib()
    nbytes = L.srf_wino43_packed_weight_bytes(Cout, Cin)
    if (kh, kw) != (3, 3) or nbytes == 0:
        raise ValueError("wino43: needs a (Cout, Cin, 3, 3) weight with Cin % 8 == 0")
    packed = _empty((nbytes // 4,), torch.float32, weight.device)
    check(L.srf_wino43_pack_weights(_ptr(weight), Cout, Cin, _ptr(packed), _stream()), "wino43_pack_weights")
    return packed


def wino43_supported(x, Cout, out=None):
    """Shape / layout limits of srf_wino43 (csrc/wino43.hip: w43_make_args, w43_slab_args)."""
    if not (x.dim() == 4 and x.is_cuda and x.dtype == torch.float32 and x.shape[3] % 8 == 0 and Cout % 4 == 0
            and x.data_ptr() % 16 == 0):
        return False
    try:
        ld = nhwc_ld(x)
        old = nhwc_ld(out) if out is not None else Cout
    except RuntimeError:
        return False
    if ld % 4 or old % 4 or (out is not None and out.data_ptr() % 16):
        return False
    N, H, W, _ = x.shape
    # one slab = the whole layer below 2 GB of V: its images must span less than 4 GB of x and of y
    return 4 * N * H * W * max(ld, old) < (1 << 32) - 16 and N * ((H + 3) // 4) * ((W + 3) // 4) < (1 << 31) - 64


def _aligned16(t):
    return t if t.data_ptr() % 16 == 0 else t.clone()


def wino43(x, packed_weight, Cout, scale=None, shift=None, relu=False, out=None):
    """3x3 / stride 1 / padding 1 convolution of the NHWC slice x (N, H, W, Cin) as Winograd F(4x4, 3x3) + per-channel scale /
    shift + ReLU into `out` (an (N, H, W, Cout) slice of an NHWC buffer; a new contiguous tensor when None)."""
    x_ld = nhwc_ld(x)
    N, H, W, Cin = x.shape
    if out is None:
        out = _empty((N, H, W, Cout), torch.float32, x.device)
    elif tuple(out.shape) != (N, H, W, Cout):
        raise ValueError("wino43: out has the wrong shape")
    y_ld = nhwc_ld(out)
    L = _lib.lib()
    if packed_weight.numel() * 4 != L.srf_wino43_packed_weight_bytes(Cout, Cin):
        raise ValueError("wino43: packed weight does not match (Cout, Cin)")
    if N == 0:
        return out
    ws_bytes = L.srf_wino43_workspace_bytes(N, H, W, Cin, Cout)
    ws = _empty((ws_bytes // 4,), torch.float32, x.device)
    sc = None if scale is None else _aligned16(_dev(scale, "scale", torch.float32))
    sh = None if shift is None else _aligned16(_dev(shift, "shift", torch.float32))
    scp, shp = None if sc is None else _ptr(sc), None if sh is None else _ptr(sh)
    timing = _dense_timing("w43m")
    if timing is None:
        check(L.srf_wino43(_ptr(x), N, H, W, Cin, x_ld, _ptr(packed_weight), Cout, scp, shp, int(bool(relu)), _ptr(out), y_ld, _ptr(ws),
                           ws_bytes, _stream()), "wino43")
        return out
    # bench.py asked for per-launch times: the transform (HBM-bound) and the multiply (MFMA-bound) as separate calls
    rc = L.srf_wino43_transform(_ptr(x), N, H, W, Cin, x_ld, Cout, _ptr(ws), ws_bytes, _stream())
    if rc != 0:   # a layer cut into slabs cannot be timed apart
        check(L.srf_wino43(_ptr(x), N, H, W, Cin, x_ld, _ptr(packed_weight), Cout, scp, shp, int(bool(relu)), _ptr(out), y_ld, _ptr(ws),
                           ws_bytes, _stream()), "wino43")
        return out
    ev1 = torch.cuda.Event(enable_timing=True)
    ev1.record()
    check(L.srf_wino43_multiply(_ptr(ws), ws_bytes, N, H, W, Cin, _ptr(packed_weight), Cout, scp, shp, int(bool(relu)), _ptr(out), y_ld,
                                _stream()), "wino43_multiply")
    ev2 = torch.cuda.Event(enable_timing=True)
    ev2.record()
    direct = 2.0 * 9 * Cin * Cout * N * H * W
    label = f"{Cin}->{Cout} @{N}x{H}x{W}"
    # transform: reads the input once, writes V (2.25x the input, padded to whole tile blocks)
    xrec = KERNEL_TIMING.get("w43x")
    if xrec is not None:
        xrec.append((timing[0], ev1, label, 0.0, 0.0, 4.0 * N * H * W * Cin + ws_bytes))
    # multiply: executes direct / 4 FLOPs on the MFMA; reads V and U once, writes the output
    timing[1].append((ev1, ev2, label, direct, direct / 4.0, float(ws_bytes) + 4.0 * 36 * Cin * Cout + 4.0 * N * H * W * Cout))
    return out


def pack_conv1x1_nhwc_weights(weight):
    """(Cout, K) or (Cout, K, 1, 1) -> the LDS operand order srf_conv1x1_nhwc copies (once per layer)."""
    weight = _dev(weight.reshape(weight.shape[0], -1), "weight", torch.float32)
    Cout, K = weight.shape
    L = _lib.lib()
    nbytes = L.srf_conv1x1_nhwc_packed_weight_bytes(Cout, K)
    if nbytes == 0:
        raise ValueError("conv1x1_nhwc: K must be a multiple of 32")
    packed = _empty((nbytes // 4,), torch.float32, weight.device)
    check(L.srf_conv1x1_nhwc_pack_weights(_ptr(weight), Cout, K, _ptr(packed), _stream()), "conv1x1_nhwc_pack_weights")
    return packed


def pack_conv1x1_nhwc_direct_weights(weight):
    """(Cout, K) or (Cout, K, 1, 1) -> the operand order srf_conv1x1_nhwc_direct streams from L2 (once per layer)."""
    weight = _dev(weight.reshape(weight.shape[0], -1), "weight", torch.float32)
    Cout, K = weight.shape
    L = _lib.lib()
    nbytes = L.srf_conv1x1_nhwc_direct_packed_weight_bytes(Cout, K)
    if nbytes == 0:
        raise ValueError("conv1x1_nhwc: K must be a multiple of 32")
    packed = _empty((nbytes // 4,), torch.float32, weight.device)
    check(L.srf_conv1x1_nhwc_direct_pack_weights(_ptr(weight), Cout, K, _ptr(packed), _stream()), "conv1x1_nhwc_direct_pack_weights")
    return packed


def pack_conv1x1_nhwc_split_weights(weight):
    """(Cout, K) or (Cout, K, 1, 1) -> the three bf16 planes of the weight (w = wh + wm + wl exactly) in the LDS operand order
    srf_conv1x1_nhwc_split copies (once per layer); an int16 tensor (6 bytes per weight)."""
    weight = _dev(weight.reshape(weight.shape[0], -1), "weight", torch.float32)
    Cout, K = weight.shape
    L = _lib.lib()
    nbytes = L.srf_conv1x1_nhwc_split_packed_weight_bytes(Cout, K)
    if nbytes == 0:
        raise ValueError("conv1x1_nhwc: K must be a multiple of 32")
    if not gemm_split_weight_in_domain(weight):
        return None     # callers keep such a layer on the f32-MFMA kernels
    packed = _empty((nbytes // 2,), torch.int16, weight.device)
    check(L.srf_conv1x1_nhwc_split_pack_weights(_ptr(weight), Cout, K, _ptr(packed), _stream()), "conv1x1_nhwc_split_pack_weights")
    return packed


GEMM_SPLIT_MAX = float.fromhex("0x1.FEp127")   # the largest bf16 (3.3895e38): above it the first plane rounds to infinity


def gemm_split_weight_in_domain(weight):
    """The three-way bf16 split is exact for finite values of magnitude <= GEMM_SPLIT_MAX (csrc/gemm_split.hip, "Domain"); a weight
    outside that range (larger, infinite or NaN) would turn whole output columns into NaN where the f32 fma chain stays finite or
    propagates an infinity, so such a layer is kept on the f32-MFMA kernels.  Checked once per packed weight (one reduction and one
    device -> host read; skipped -- the weight taken as in range -- while a stream capture is running, where a read-back is illegal:
    the graphs are captured after an eager warm-up that has packed, and checked, every layer)."""
    if weight.is_cuda and torch.cuda.is_current_stream_capturing():
        return True
    return bool((weight.abs().max() <= GEMM_SPLIT_MAX).item()) if weight.numel() else True   # NaN compares False


def gemm_split_enabled():
    """SRF_GEMM_SPLIT=0 keeps the 1x1 convolutions on the f32-MFMA kernels (`srf_conv1x1_nhwc` / `_direct`: one k-ordered fma chain per
    output); the default runs them on `srf_conv1x1_nhwc_split` -- the same f32 GEMM through an exact three-way bf16 split of both
    operands on the bf16 MFMA (csrc/gemm_split.hip: error against float64 equal to the f32 chain's, 1.4-1.5x its rate)."""
    import os
    return os.environ.get("SRF_GEMM_SPLIT", "1") != "0"


GEMM_SPLIT_MIN_TILES = 128   # 128 x 128 tiles of a launch from which the split kernel is taken (tools/small_gemm_ab.py)


def gemm_split_wanted(M, Cout):
    """The split GEMM has one tile form (128 x 128, three workgroups per CU); a launch of fewer than ~half a round of tiles (the BEV
    FPN's laterals and stride-2 extras, the coarse image laterals) fills the chip better on the 64 x 64 tiles of the f32-MFMA kernels:
    nusc_L, whose GEMMs are all of that size, ran 1.5-2 % slower with everything on the split kernel (same box, alternating runs:
    229.7 / 233.7 against 234.8 / 237.1 frames/s).  SRF_GEMM_SPLIT_MIN overrides the threshold (A/B switch)."""
    import os
    if not gemm_split_enabled():
        return False
    thr = int(os.environ.get("SRF_GEMM_SPLIT_MIN", GEMM_SPLIT_MIN_TILES))
    return ((M + 127) // 128) * ((Cout + 127) // 128) >= thr


GEMM_DIRECT_MIN_TILES = 1024   # 128 x 128 tiles of a launch from which the LDS-free kernel wins (tools/micro/gemm_direct_bench.hip)


def conv1x1_direct_wanted(M, Cout):
    """The LDS-free GEMM (`srf_conv1x1_nhwc_direct`) pays on launches of more than a round of 128 x 128 tiles at three workgroups
    per CU (VoVNet stages 2-4: 122-135 against 110-117 TFLOP/s); below that the 64 x 64 tiles of `srf_conv1x1_nhwc` fill the chip
    better (stage 5: 104 against 93).  SRF_GEMM_DIRECT=0 / 1 forces the choice (A/B switch for tests and benchmarks)."""
    import os
    force = os.environ.get("SRF_GEMM_DIRECT")
    if force is not None:
        return force != "0"
    return ((M + 127) // 128) * ((Cout + 127) // 128) >= GEMM_DIRECT_MIN_TILES


def conv1x1_nhwc(x, packed_weight, Cout, scale=None, shift=None, relu=False, out=None, pool=False, top=None, packed_direct=None,
                 packed_split=None):
    """1x1 convolution of the NHWC slice x (N, H, W, K) + per-channel scale / shift + ReLU into `out` ((N, H, W, Cout) slice
    of an NHWC buffer; new contiguous tensor when None).  pool=True: returns (out, mean (N, Cout) over the pixels of each
    image) from the same pass (`srf_conv1x1_nhwc_pooled`).  top: an (N, Ht, Wt, Cout) NHWC slice whose nearest-neighbour
    upsampling to (H, W) is added to the result in the epilogue (`srf_conv1x1_nhwc_topdown`: the FPN top-down step).
    packed_direct: the same weight packed by `pack_conv1x1_nhwc_direct_weights`, or a callable returning it; large launches then
    run on the LDS-free kernel (`srf_conv1x1_nhwc_direct*`: the same bits in `out`).  packed_split: the weight packed by
    `pack_conv1x1_nhwc_split_weights` (or a callable): unless SRF_GEMM_SPLIT=0 the layer runs on `srf_conv1x1_nhwc_split*` (f32 GEMM
    on the bf16 MFMA through an exact three-way split; f32-accurate, not the bits of the fma chain)."""
    x_ld = nhwc_ld(x)
    N, H, W, K = x.shape
    if out is None:
        out = _empty((N, H, W, Cout), torch.float32, x.device)
    elif tuple(out.shape) != (N, H, W, Cout):
        raise ValueError("conv1x1_nhwc: out has the wrong shape")
    y_ld = nhwc_ld(out)
    L = _lib.lib()
    split = packed_split is not None and gemm_split_wanted(N * H * W, Cout) and max(x_ld, y_ld) * 512 < (1 << 31)
    if split and callable(packed_split):
        packed_split = packed_split()
        split = packed_split is not None     # None: a weight outside the split's exact domain (gemm_split_weight_in_domain)
    direct = not split and packed_direct is not None and conv1x1_direct_wanted(N * H * W, Cout) and max(x_ld, y_ld) * 512 < (1 << 31)
    if split:
        if packed_split.numel() * 2 != L.srf_conv1x1_nhwc_split_packed_weight_bytes(Cout, K):
            raise ValueError("conv1x1_nhwc: split-packed weight does not match (Cout, K)")
        wp = _ptr(packed_split)
    elif direct:
        if callable(packed_direct):
            packed_direct = packed_direct()
        if packed_direct.numel() * 4 != L.srf_conv1x1_nhwc_direct_packed_weight_bytes(Cout, K):
            raise ValueError("conv1x1_nhwc: direct-packed weight does not match (Cout, K)")
        wp = _ptr(packed_direct)
    else:
        if callable(packed_weight):
            packed_weight = packed_weight()
        if packed_weight.numel() * 4 != L.srf_conv1x1_nhwc_packed_weight_bytes(Cout, K):
            raise ValueError("conv1x1_nhwc: packed weight does not match (Cout, K)")
        wp = _ptr(packed_weight)
    timing = _dense_timing("gsplit" if split else "gemm")
    sc = None if scale is None else _ptr(_dev(scale, "scale", torch.float32))
    sh = None if shift is None else _ptr(_dev(shift, "shift", torch.float32))
    mean = None
    if top is not None:
        if pool or top.dim() != 4 or top.shape[0] != N or top.shape[3] != Cout:
            raise ValueError("conv1x1_nhwc: top must be (N, Ht, Wt, Cout) and excludes pool")
        fn = L.srf_conv1x1_nhwc_split_topdown if split else (L.srf_conv1x1_nhwc_direct_topdown if direct else L.srf_conv1x1_nhwc_topdown)
        check(fn(_ptr(x), N, H, W, K, x_ld, wp, Cout, sc, sh, int(bool(relu)), _ptr(top), top.shape[1], top.shape[2], nhwc_ld(top), _ptr(out),
                 y_ld, _stream()), "conv1x1_nhwc_topdown")
    elif pool:
        mean = _empty((N, Cout), torch.float32, x.device)
        nbytes = L.srf_conv1x1_nhwc_pooled_workspace_bytes(N, H * W, Cout)
        ws = _empty((max(nbytes, 4) // 4,), torch.float32, x.device)
        fn = L.srf_conv1x1_nhwc_split_pooled if split else (L.srf_conv1x1_nhwc_direct_pooled if direct else L.srf_conv1x1_nhwc_pooled)
        check(fn(_ptr(x), N, H * W, K, x_ld, wp, Cout, sc, sh, int(bool(relu)), _ptr(out), y_ld, _ptr(mean), _ptr(ws), nbytes, _stream()),
              "conv1x1_nhwc_pooled")
    else:
        fn = L.srf_conv1x1_nhwc_split if split else (L.srf_conv1x1_nhwc_direct if direct else L.srf_conv1x1_nhwc)
        check(fn(_ptr(x), N * H * W, K, x_ld, wp, Cout, sc, sh, int(bool(relu)), _ptr(out), y_ld, _stream()), "conv1x1_nhwc")
    if timing is not None:
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record()
        fl = 2.0 * K * Cout * N * H * W
        # split: six bf16 products per f32 product are issued on the bf16 MFMA; the weights are 6 bytes each
        timing[1].append((timing[0], ev1, f"{K}->{Cout} @{N}x{H}x{W}" + (" split" if split else (" direct" if direct else "")), fl,
                          6.0 * fl if split else fl, 4.0 * N * H * W * (K + Cout) + (6.0 if split else 4.0) * K * Cout))
    return (out, mean) if pool else out


def _opt(t, name):
    return None if t is None else _ptr(_dev(t, name, torch.float32))


def nhwc_affine(x, scale=None, shift=None, relu=False, residual=None, out=None):
    """out = x * scale + shift (+ residual), optional ReLU, on NHWC slices.  scale: (C,) or per sample (N, C)."""
    x_ld = nhwc_ld(x)
    N, H, W, C = x.shape
    if out is None:
        out = _empty((N, H, W, C), torch.float32, x.device)
    per_sample = int(scale is not None and scale.dim() == 2) | (2 if (shift is not None and shift.dim() == 2) else 0)
    if scale is not None and scale.numel() != (N * C if per_sample & 1 else C):
        raise ValueError("nhwc_affine: scale has the wrong size")
    if shift is not None and shift.numel() != (N * C if per_sample & 2 else C):
        raise ValueError("nhwc_affine: shift has the wrong size")
    check(_lib.lib().srf_nhwc_affine(_ptr(x), x_ld, N, H * W, C, _opt(scale, "scale"), per_sample, _opt(shift, "shift"),
                                     None if residual is None else _ptr(residual),
                                     0 if residual is None else nhwc_ld(residual), int(bool(relu)), _ptr(out), nhwc_ld(out),
                                     _stream()), "nhwc_affine")
    return out


def nhwc_colmean(x):
    """(N, H, W, C) NHWC slice -> (N, C) mean over the pixels."""
    x_ld = nhwc_ld(x)
    N, H, W, C = x.shape
    L = _lib.lib()
    ws = _empty((max(L.srf_nhwc_colmean_workspace_bytes(N, C) // 4, 1),), torch.float32, x.device)
    mean = _empty((N, C), torch.float32, x.device)
    check(L.srf_nhwc_colmean(_ptr(x), x_ld, N, H * W, C, _ptr(mean), _ptr(ws), ws.numel() * 4, _stream()), "nhwc_colmean")
    return mean


def nhwc_colsum_prod(a, b):
    """(N, H, W, C) NHWC slices a, b -> (N, C): the sum over the pixels of a * b (deterministic two-level sum)."""
    a_ld, b_ld = nhwc_ld(a), nhwc_ld(b)
    N, H, W, C = a.shape
    if tuple(b.shape) != (N, H, W, C):
        raise ValueError("nhwc_colsum_prod: shapes differ")
    L = _lib.lib()
    ws = _empty((max(L.srf_nhwc_colmean_workspace_bytes(N, C) // 4, 1),), torch.float32, a.device)
    out = _empty((N, C), torch.float32, a.device)
    check(L.srf_nhwc_colsum_prod(_ptr(a), a_ld, _ptr(b), b_ld, N, H * W, C, _ptr(out), _ptr(ws), ws.numel() * 4, _stream()), "nhwc_colsum_prod")
    return out


def pool3s2_out(h):
    ho = 1 if h < 3 else (h - 3 + 1) // 2 + 1
    return ho - 1 if (ho - 1) * 2 >= h else ho


def nhwc_maxpool3s2_ceil(x, out=None):
    x_ld = nhwc_ld(x)
    N, H, W, C = x.shape
    Ho, Wo = pool3s2_out(H), pool3s2_out(W)
    if out is None:
        out = _empty((N, Ho, Wo, C), torch.float32, x.device)
    elif tuple(out.shape) != (N, Ho, Wo, C):
        raise ValueError("nhwc_maxpool3s2_ceil: out has the wrong shape")
    check(_lib.lib().srf_nhwc_maxpool3s2_ceil(_ptr(x), x_ld, N, H, W, C, _ptr(out), nhwc_ld(out), _stream()), "nhwc_maxpool3s2_ceil")
    return out


def nhwc_upsample_add(lat, top, out=None):
    """lat (N, H, W, C) + nearest-upsampled top (N, Ht, Wt, C); out defaults to lat (in place)."""
    N, H, W, C = lat.shape
    if out is None:
        out = lat
    check(_lib.lib().srf_nhwc_upsample_add(_ptr(lat), nhwc_ld(lat), _ptr(top), nhwc_ld(top), N, H, W, top.shape[1], top.shape[2], C,
                                           _ptr(out), nhwc_ld(out), _stream()), "nhwc_upsample_add")
    return out


def nhwc_dwconv3x3s2(x, weight, scale=None, shift=None, relu=False, out=None):
    """Depthwise 3x3 / stride 2 / padding 1 on an NHWC slice; weight (C, 1, 3, 3)."""
    x_ld = nhwc_ld(x)
    N, H, W, C = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if out is None:
        out = _empty((N, Ho, Wo, C), torch.float32, x.device)
    elif tuple(out.shape) != (N, Ho, Wo, C):
        raise ValueError("nhwc_dwconv3x3s2: out has the wrong shape")
    w = _dev(weight.reshape(C, 9), "weight", torch.float32)
    check(_lib.lib().srf_nhwc_dwconv3x3s2(_ptr(x), x_ld, N, H, W, C, _ptr(w), _opt(scale, "scale"), _opt(shift, "shift"),
                                          int(bool(relu)), _ptr(out), nhwc_ld(out), _stream()), "nhwc_dwconv3x3s2")
    return out


def nhwc_dwconv3x3s2_cat(x, weight, scale, shift, relu, side, out):
    """One step of the proposal generator's stair: out (N, Ho, Wo, Cs + C) = cat([side, dwconv(x)], -1) in one launch."""
    x_ld = nhwc_ld(x)
    N, H, W, C = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Cs = side.shape[3]
    if tuple(side.shape[:3]) != (N, Ho, Wo) or tuple(out.shape) != (N, Ho, Wo, Cs + C):
        raise ValueError("nhwc_dwconv3x3s2_cat: side / out have the wrong shape")
    w = _dev(weight.reshape(C, 9), "weight", torch.float32)
    check(_lib.lib().srf_nhwc_dwconv3x3s2_cat(_ptr(x), x_ld, N, H, W, C, _ptr(w), _opt(scale, "scale"), _opt(shift, "shift"),
                                              int(bool(relu)), _ptr(out[..., Cs:]), nhwc_ld(out), _ptr(side), nhwc_ld(side), Cs,
                                              _ptr(out), _stream()), "nhwc_dwconv3x3s2_cat")
    return out


def nhwc_affine_relu_bwd(gy, y, scale, relu, gy2=None):
    """Backward of y = relu(z * scale + shift) on NHWC tensors (N, H, W, C): -> gz (N, H, W, C) = masked gy * scale, and the column
    sums (2, C) [sum gu, sum gu * y] (gu = gy where y > 0); one streaming pass + a finish launch (srf_nhwc_affine_relu_bwd).
    gy2: a second gradient of the same output, added to gy on the way in."""
    N, H, W, C = gy.shape
    gy_ld, y_ld = nhwc_ld(gy), nhwc_ld(y)
    if gy2 is not None and tuple(gy2.shape) != tuple(gy.shape):
        raise ValueError("nhwc_affine_relu_bwd: gy2 has another shape")
    M = N * H * W
    L = _lib.lib()
    gz = _empty((N, H, W, C), torch.float32, gy.device)
    sums = _empty((2, C), torch.float32, gy.device)
    nbytes = L.srf_nhwc_affine_relu_bwd_workspace_bytes(M, C)
    ws = _empty((max(nbytes, 4) // 4,), torch.float32, gy.device)
    sc = None if scale is None else _aligned16(_dev(scale, "scale", torch.float32))
    check(L.srf_nhwc_affine_relu_bwd2(_ptr(gy), gy_ld, _ptr(gy2), 0 if gy2 is None else nhwc_ld(gy2), _ptr(y), y_ld, M, C, _ptr(sc),
                                      int(bool(relu)), _ptr(gz), C, _ptr(sums), _ptr(ws), nbytes, _stream()), "nhwc_affine_relu_bwd")
    return gz, sums


def bn_eval_fold(gamma, beta, mean, var, eps):
    """(3, C): s = gamma / sqrt(var + eps), t0 = beta - mean s, inv = 1 / sqrt(var + eps) of an eval-mode BatchNorm (one launch)."""
    C = gamma.numel()
    out = _empty((3, C), torch.float32, gamma.device)
    check(_lib.lib().srf_bn_eval_fold(_ptr(_dev(gamma, "gamma", torch.float32)), _ptr(_dev(beta, "beta", torch.float32)),
                                      _ptr(_dev(mean, "mean", torch.float32)), _ptr(_dev(var, "var", torch.float32)), float(eps), C, _ptr(out),
                                      _stream()), "bn_eval_fold")
    return out


def bn_eval_grads(sums, fold, mean):
    """(2, C): d gamma, d beta of an eval-mode BatchNorm from nhwc_affine_relu_bwd's column sums and bn_eval_fold's vectors (one launch)."""
    C = mean.numel()
    out = _empty((2, C), torch.float32, sums.device)
    check(_lib.lib().srf_bn_eval_grads(_ptr(sums), _ptr(fold), _ptr(_dev(mean, "mean", torch.float32)), C, _ptr(out), _stream()), "bn_eval_grads")
    return out


def nhwc_pool_sum(x, n_cam=1, size=None, pad_to=4):
    """x (B * n_cam, H, W, C) channels-last -> (B, pad(Ho * Wo)): per output pixel the sum over cameras and channels at its `nearest`
    source pixel (size = (Ho, Wo); None: the map itself); columns past Ho * Wo are zeros (row length rounded up to pad_to)."""
    x_ld = nhwc_ld(x)
    Nimg, H, W, C = x.shape
    B = Nimg // n_cam
    Ho, Wo = (H, W) if size is None else (int(size[0]), int(size[1]))
    out_ld = -(-(Ho * Wo) // pad_to) * pad_to
    out = _empty((B, out_ld), torch.float32, x.device)
    check(_lib.lib().srf_nhwc_pool_sum(_ptr(x), x_ld, B, n_cam, H, W, C, Ho, Wo, _ptr(out), out_ld, _stream()), "nhwc_pool_sum")
    return out


def dpg_mix(wl, wi, boxes_w, feats_w, E, P):
    """expert logits (B, E * P) (+ the camera half's) -> proposal boxes (B, P, D) with sigmoid centres, features (B, P, C)."""
    wl = _dev(wl, "wl", torch.float32)
    wi = _dev(wi, "wi", torch.float32) if wi is not None else None
    boxes_w = _dev(boxes_w, "boxes_w", torch.float32)
    feats_w = _dev(feats_w, "feats_w", torch.float32)
    B, D, C = wl.shape[0], boxes_w.shape[1], feats_w.shape[1]
    if wl.shape[1] != E * P or boxes_w.shape[0] != E * P or feats_w.shape[0] != E * P or (wi is not None and wi.shape != wl.shape):
        raise ValueError("dpg_mix: shapes disagree")
    boxes = _empty((B, P, D), torch.float32, wl.device)
    feats = _empty((B, P, C), torch.float32, wl.device)
    check(_lib.lib().srf_dpg_mix(_ptr(wl), _ptr(wi), B, E, P, _ptr(boxes_w), D, _ptr(feats_w), C, _ptr(boxes), _ptr(feats), _stream()),
          "dpg_mix")
    return boxes, feats


def pack_conv_gemm_weights(weight):
    """(Cout, Cin, kh, kw) -> packed operand of srf_conv_gemm_nhwc: k = (tap, input channel), tap slowest."""
    Cout = weight.shape[0]
    return pack_conv1x1_nhwc_weights(weight.detach().permute(0, 2, 3, 1).reshape(Cout, -1).contiguous())


def pack_conv_gemm_split_weights(weight):
    """(Cout, Cin, kh, kw) -> packed operand of srf_conv_gemm_nhwc_split (bf16 planes): k = (tap, input channel), tap slowest."""
    Cout = weight.shape[0]
    return pack_conv1x1_nhwc_split_weights(weight.detach().permute(0, 2, 3, 1).reshape(Cout, -1).contiguous())


def conv_gemm_nhwc(x, packed_weight, Cout, ksize, stride, pad, scale=None, shift=None, relu=False, out=None, packed_split=None):
    """Conv2d on an NHWC slice as an implicit-im2col GEMM (the strided 3x3 layers); -> (N, Ho, Wo, Cout).  On the f32 MFMA
    (`srf_conv_gemm_nhwc`, packed_weight) or, when packed_split (from `pack_conv_gemm_split_weights`, or a callable) is given and
    SRF_GEMM_SPLIT is not 0, on the split GEMM (`srf_conv_gemm_nhwc_split`: f32-accurate on the bf16 MFMA)."""
    x_ld = nhwc_ld(x)
    N, H, W, Cin = x.shape
    kh, kw = ksize
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    if out is None:
        out = _empty((N, Ho, Wo, Cout), torch.float32, x.device)
    elif tuple(out.shape) != (N, Ho, Wo, Cout):
        raise ValueError("conv_gemm_nhwc: out has the wrong shape")
    L = _lib.lib()
    split = packed_split is not None and gemm_split_wanted(N * Ho * Wo, Cout)
    if split and callable(packed_split):
        packed_split = packed_split()
        split = packed_split is not None     # None: a weight outside the split's exact domain stays on the f32 MFMA
    if split:
        if packed_split.numel() * 2 != L.srf_conv1x1_nhwc_split_packed_weight_bytes(Cout, kh * kw * Cin):
            raise ValueError("conv_gemm_nhwc: split-packed weight does not match the layer")
        packed_weight, fn = packed_split, L.srf_conv_gemm_nhwc_split
    else:
        if callable(packed_weight):
            packed_weight = packed_weight()
        if packed_weight.numel() * 4 != L.srf_conv1x1_nhwc_packed_weight_bytes(Cout, kh * kw * Cin):
            raise ValueError("conv_gemm_nhwc: packed weight does not match the layer")
        fn = L.srf_conv_gemm_nhwc
    # the kernel addresses its whole input through ONE 32-bit buffer descriptor (N H W x_ld 4 < 2^31 bytes): larger batches
    # run in groups of images (VoVNet stem_3 reads 464 x 800 x 64 per camera: 23 images reach the limit -- LC inference at
    # batch 4, the frozen prefix of config 4 at bs >= 4)
    per_img = 4 * H * W * x_ld
    group = N if N * per_img < (1 << 31) else max(1, ((1 << 31) - 1) // per_img)
    sc, sh = _opt(scale, "scale"), _opt(shift, "shift")
    timing = _dense_timing("cgemm")
    for n0 in range(0, max(N, 1), max(group, 1)):
        xs, os_ = x[n0:n0 + group], out[n0:n0 + group]
        check(fn(_ptr(xs), xs.shape[0], H, W, Cin, x_ld, _ptr(packed_weight), Cout, kh, kw, stride, pad, sc, sh,
                 int(bool(relu)), _ptr(os_), nhwc_ld(out), _stream()), "conv_gemm_nhwc")
    if timing is not None:
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record()
        fl = 2.0 * kh * kw * Cin * Cout * N * Ho * Wo
        timing[1].append((timing[0], ev1, f"{Cin}->{Cout} {kh}x{kw}/s{stride} @{N}x{H}x{W}" + (" split" if split else ""), fl,
                          6.0 * fl if split else fl, 4.0 * N * (H * W * Cin + Ho * Wo * Cout) + (6.0 if split else 4.0) * kh * kw * Cin * Cout))
    return out


def conv_wgrad_supported(g, x, ksize):
    """Shapes `conv_wgrad_nhwc` takes: (N, H, W, C) channel slices of channels-last f32 buffers, ksize 1 or 3, W >= 32, channel counts
    and pixel pitches multiples of 4, every tensor below 2 GB."""
    try:
        g_ld, x_ld = nhwc_ld(g), nhwc_ld(x)
    except RuntimeError:
        return False
    N, H, W, Cin = x.shape
    Cout = g.shape[3]
    return (ksize in (1, 3) and tuple(g.shape[:3]) == (N, H, W) and W >= 32 and Cin % 4 == 0 and Cout % 4 == 0 and g_ld % 4 == 0
            and x_ld % 4 == 0 and g.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 and N * H * W * max(g_ld, x_ld) * 4 < (1 << 31))


def conv_wgrad_nhwc(g, x, ksize):
    """Weight gradient of a stride-1 Conv2d (ksize 1 / padding 0, or ksize 3 / padding 1) from the channels-last output gradient
    g (N, H, W, Cout) and input x (N, H, W, Cin): -> (Cout, Cin, ksize, ksize), `srf_conv_wgrad_nhwc` (an f32 GEMM over the pixels on
    the bf16 MFMA, exact three-way split of both operands, fixed summation order: deterministic)."""
    if not conv_wgrad_supported(g, x, ksize):
        raise ValueError("conv_wgrad_nhwc: unsupported shapes / layout")
    N, H, W, Cin = x.shape
    Cout = g.shape[3]
    L = _lib.lib()
    nbytes = L.srf_conv_wgrad_workspace_bytes(N, H, W, Cin, Cout, ksize)
    ws = _empty((max(nbytes, 4) // 4,), torch.float32, x.device)
    dW = _empty((Cout, Cin, ksize, ksize), torch.float32, x.device)
    check(L.srf_conv_wgrad_nhwc(_ptr(g), nhwc_ld(g), _ptr(x), nhwc_ld(x), N, H, W, Cin, Cout, ksize, _ptr(ws), nbytes, _ptr(dW), _stream()),
          "conv_wgrad_nhwc")
    return dW


def conv_gemm_nhwc_supported(x):
    """Layout / size limits of srf_conv_gemm_nhwc as `conv_gemm_nhwc` drives it (batches are split, one image must fit)."""
    try:
        ld = nhwc_ld(x)
    except RuntimeError:
        return False
    return x.shape[3] % 32 == 0 and ld % 4 == 0 and x.data_ptr() % 16 == 0 and 4 * x.shape[1] * x.shape[2] * ld < (1 << 31)


def stem_conv_nchw(x, weight, scale=None, shift=None, relu=False, out=None):
    """Conv2d(Cin <= 4, 64, 3, stride 2, padding 1) + affine + ReLU from contiguous NCHW images to an NHWC tensor."""
    x = _dev(x, "x", torch.float32)
    N, Cin, H, W = x.shape
    Cout = weight.shape[0]
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if out is None:
        out = _empty((N, Ho, Wo, Cout), torch.float32, x.device)
    check(_lib.lib().srf_stem_conv_nchw(_ptr(x), N, Cin, H, W, _ptr(_dev(weight.detach(), "weight", torch.float32)), Cout,
                                        _opt(scale, "scale"), _opt(shift, "shift"), int(bool(relu)), _ptr(out), nhwc_ld(out),
                                        _stream()), "stem_conv_nchw")
    return out


def _ptr_array(tensors):
    import ctypes
    arr = (ctypes.c_void_p * max(len(tensors), 1))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def stage_tail(obj, ffn, norm3, cls_layers, reg_layers, logits_fc, deltas_fc, boxes, weights6, pc_range, scale_clamp, split_ffn=True):
    """FFN + residual + norm3, both towers, class_logits, bboxes_delta and apply_deltas: the FFN as one workgroup per (32 rows,
    128 hidden units), everything after it in one launch per 32 rows (split_ffn=False: the FFN in line, a single launch).
    ffn = (linear1, linear2); cls_layers / reg_layers = [(Linear(bias=False), LayerNorm), ...].
    Returns obj_out (R,C), logits (R,ncls), pred (R,Dd)."""
    obj = _dev(obj, "obj", torch.float32)
    boxes = _dev(boxes, "boxes", torch.float32)
    R, C = obj.shape
    lin1, lin2 = ffn
    F = lin1.weight.shape[0]
    ncls, Dd = logits_fc.weight.shape[0], deltas_fc.weight.shape[0]
    if boxes.shape != (R, Dd):
        raise ValueError("boxes must be (R, Dd)")
    obj_out = _empty((R, C), torch.float32, obj.device)
    logits = _empty((R, ncls), torch.float32, obj.device)
    pred = _empty((R, Dd), torch.float32, obj.device)
    cw, cg, cb = ([m.weight for m, _ in cls_layers], [n.weight for _, n in cls_layers], [n.bias for _, n in cls_layers])
    rw, rg, rb = ([m.weight for m, _ in reg_layers], [n.weight for _, n in reg_layers], [n.bias for _, n in reg_layers])
    for t in [lin1.weight, lin1.bias, lin2.weight, lin2.bias, norm3.weight, norm3.bias, logits_fc.weight, logits_fc.bias,
              deltas_fc.weight, deltas_fc.bias] + cw + cg + cb + rw + rg + rb:
        if t.dtype != torch.float32 or not t.is_contiguous() or t.device != obj.device:
            raise ValueError("stage_tail: parameters must be contiguous float32 on the input's device")
    L = _lib.lib()
    ws, ws_bytes = None, 0
    if split_ffn:
        ws_bytes = L.srf_stage_tail_workspace_bytes(R, C, F)
        ws = _empty((max(ws_bytes, 4) // 4,), torch.float32, obj.device)
    check(L.srf_stage_tail(
        _ptr(obj), R, C, F, _ptr(lin1.weight), _ptr(lin1.bias), _ptr(lin2.weight), _ptr(lin2.bias), _ptr(norm3.weight),
        _ptr(norm3.bias), float(norm3.eps), len(cls_layers), _ptr_array(cw), _ptr_array(cg), _ptr_array(cb),
        hf([n.eps for _, n in cls_layers] or [0.0]), len(reg_layers), _ptr_array(rw), _ptr_array(rg), _ptr_array(rb),
        hf([n.eps for _, n in reg_layers] or [0.0]), _ptr(logits_fc.weight), _ptr(logits_fc.bias), ncls, _ptr(deltas_fc.weight),
        _ptr(deltas_fc.bias), Dd, _ptr(boxes), hf(weights6), hf(pc_range), float(scale_clamp), _ptr(obj_out), _ptr(logits),
        _ptr(pred), None if ws is None else _ptr(ws), ws_bytes, _stream()), "stage_tail")
    return obj_out, logits, pred


# ---------------------------------------------------------------------------------------------- training support
class _RoIExtractFn(torch.autograd.Function):
    """roi_extract with a gradient for the feature maps (RoIs carry none, as in mmcv)."""

    @staticmethod
    def forward(ctx, rois, strides, out_size, sampling_ratio, finest_scale, bin_major, *feats):
        out = roi_extract(list(feats), rois, strides, out_size, sampling_ratio, finest_scale, bin_major=bin_major)
        ctx.save_for_backward(rois, *feats)
        ctx.cfg = (strides, out_size, sampling_ratio, finest_scale, bin_major)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        rois, *feats = ctx.saved_tensors
        strides, out_size, sampling_ratio, finest_scale, bin_major = ctx.cfg
        grad_out = grad_out.contiguous()
        R, C, nl = rois.shape[0], feats[0].shape[1], len(strides)
        grads = [torch.zeros_like(f) for f in feats]
        fm = (FeatMap * nl)(*[_featmap(f, 1.0 / s) for f, s in zip(feats, strides)])
        gp = (ctypes.c_void_p * nl)(*[g.data_ptr() for g in grads])
        bins = out_size * out_size
        so_c, so_b = (1, C) if bin_major else (bins, 1)
        check(_lib.lib().srf_roi_extract_bwd(fm, gp, nl, C, _ptr(rois), R, out_size, sampling_ratio, float(finest_scale),
                                             _ptr(grad_out), C * bins, so_c, so_b, _stream()), "roi_extract_bwd")
        return (None, None, None, None, None, None, *grads)


def roi_extract_autograd(feats, rois, strides, out_size=7, sampling_ratio=2, finest_scale=56.0, bin_major=False):
    feats = [f if f.is_contiguous() or f.is_contiguous(memory_format=torch.channels_last) else f.contiguous() for f in feats]
    return _RoIExtractFn.apply(rois.detach(), list(strides), out_size, sampling_ratio, finest_scale, bin_major, *feats)


def box_iou_rotated(a, b):
    """pairwise rotated BEV IoU: a (n,5), b (m,5) as (cx, cy, w, h, angle) -> (n, m)."""
    a = _dev(a, "a", torch.float32)
    b = _dev(b, "b", torch.float32)
    out = _empty((a.shape[0], b.shape[0]), torch.float32, a.device)
    check(_lib.lib().srf_box_iou_rotated(_ptr(a), a.shape[0], _ptr(b), b.shape[0], _ptr(out), _stream()), "box_iou_rotated")
    return out
