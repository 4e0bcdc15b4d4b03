"""ctypes front end of oracle/libsrf_oracle.so (numpy in, numpy out).

TEST INFRASTRUCTURE ONLY: imported by tests/, `__graft_entry__.smoke()` and the `cpu_baseline` leg
of bench.py.  Nothing under srfdet3d_amd/ imports this module.  Parity status of every function
here: "parity unpinned" (see the header of srf_oracle.c).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsrf_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "srf_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _p(a, ct=ctypes.c_void_p):
    return a.ctypes.data_as(ct) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _check(rc, name):
    if rc != 0:
        raise RuntimeError(f"oracle {name} failed with code {rc}")


def grid_size(voxel_size, pc_range):
    """mmcv Voxelization.__init__: round((range[3:] - range[:3]) / voxel_size) in float32 (x,y,z)."""
    r = np.asarray(pc_range, np.float32)
    v = np.asarray(voxel_size, np.float32)
    return np.round((r[3:] - r[:3]) / v).astype(np.int64).astype(np.int32)


def dynamic_voxelize(points, voxel_size, pc_range):
    points = _f32(points)
    n, nf = points.shape
    grid = grid_size(voxel_size, pc_range)
    coors = np.empty((n, 3), np.int32)
    _check(lib().orc_dynamic_voxelize(_p(points), n, nf, _p(_f32(voxel_size)), _p(_f32(pc_range)), _p(grid),
                                      _p(coors)), "dynamic_voxelize")
    return coors


def hard_voxelize(points, voxel_size, pc_range, max_points, max_voxels):
    points = _f32(points)
    n, nf = points.shape
    grid = grid_size(voxel_size, pc_range)
    voxels = np.empty((max_voxels, max_points, nf), np.float32)
    coors = np.zeros((max_voxels, 3), np.int32)
    num = np.empty((max_voxels,), np.int32)
    m = ctypes.c_int(0)
    _check(lib().orc_hard_voxelize(_p(points), n, nf, _p(_f32(voxel_size)), _p(_f32(pc_range)), _p(grid),
                                   int(max_points), int(max_voxels), _p(voxels), _p(coors), _p(num),
                                   ctypes.byref(m)), "hard_voxelize")
    M = m.value
    return voxels[:M].copy(), coors[:M].copy(), num[:M].copy()


def vfe_mean(voxels, num, num_features=None):
    voxels = _f32(voxels)
    M, mp, nf = voxels.shape
    F = nf if num_features is None else num_features
    out = np.empty((M, F), np.float32)
    _check(lib().orc_vfe_mean(_p(voxels), _p(_i32(num)), M, mp, nf, F, _p(out)), "vfe_mean")
    return out


def dynamic_scatter(feats, coors, grid_zyx, mode):
    feats = _f32(feats)
    coors = _i32(coors)
    n, C = feats.shape
    out_f = np.empty((max(n, 1), C), np.float32)
    out_c = np.empty((max(n, 1), 4), np.int32)
    p2v = np.empty((max(n, 1),), np.int32)
    m = ctypes.c_int(0)
    _check(lib().orc_dynamic_scatter(_p(feats), _p(coors), n, C, _p(_i32(grid_zyx)), 0 if mode == "mean" else 1,
                                     _p(out_f), _p(out_c), _p(p2v), ctypes.byref(m)), "dynamic_scatter")
    M = m.value
    return out_f[:M].copy(), out_c[:M].copy(), p2v[:n].copy()


def rulebook_subm(indices, shape, ksize):
    indices = _i32(indices)
    A = indices.shape[0]
    K = int(np.prod(ksize))
    nbr = np.empty((K, max(A, 1)), np.int32)
    counts = np.empty((K,), np.int32)
    _check(lib().orc_rulebook_subm(_p(indices), A, _p(_i32(shape)), _p(_i32(ksize)), _p(nbr), _p(counts)),
           "rulebook_subm")
    return nbr[:, :A].copy(), counts


def out_shape(shape, ksize, stride, pad):
    return [int((shape[d] + 2 * pad[d] - ksize[d]) // stride[d] + 1) for d in range(3)]


def rulebook_strided(indices, shape, ksize, stride, pad):
    indices = _i32(indices)
    A = indices.shape[0]
    K = int(np.prod(ksize))
    per_in = int(np.prod([-(-ksize[d] // stride[d]) for d in range(3)]))
    cap = max(A * per_in, 1)
    out_idx = np.empty((cap, 4), np.int32)
    nbr = np.empty((K, cap), np.int32)
    counts = np.empty((K,), np.int32)
    a_out = ctypes.c_int(0)
    _check(lib().orc_rulebook_strided(_p(indices), A, _p(_i32(shape)), _p(_i32(ksize)), _p(_i32(stride)),
                                      _p(_i32(pad)), cap, _p(out_idx), ctypes.byref(a_out), _p(nbr), _p(counts)),
           "rulebook_strided")
    Ao = a_out.value
    return out_idx[:Ao].copy(), nbr[:, :Ao].copy(), counts, out_shape(shape, ksize, stride, pad)


def spconv_fwd(feats, weight, nbr, alpha=None, beta=None, residual=None, relu=False):
    """weight: (K, Cin, Cout); nbr: (K, A_out)."""
    feats = _f32(feats)
    weight = _f32(weight)
    nbr = _i32(nbr)
    K, Cin, Cout = weight.shape
    A_out = nbr.shape[1]
    out = np.empty((A_out, Cout), np.float32)
    a = _f32(alpha) if alpha is not None else None
    b = _f32(beta) if beta is not None else None
    r = _f32(residual) if residual is not None else None
    _check(lib().orc_spconv_fwd(_p(feats), feats.shape[0], Cin, _p(weight), K, _p(nbr), A_out, A_out, Cout,
                                _p(a), _p(b), _p(r), int(bool(relu)), _p(out)), "spconv_fwd")
    return out


def densify(feats, indices, batch, shape):
    feats = _f32(feats)
    indices = _i32(indices)
    A, C = feats.shape
    D, H, W = shape
    out = np.empty((batch, C, D, H, W), np.float32)
    _check(lib().orc_densify(_p(feats), _p(indices), A, C, batch, D, H, W, _p(out)), "densify")
    return out


def roi_levels(rois, num_levels, finest_scale=56.0):
    rois = _f32(rois)
    lvl = np.empty((rois.shape[0],), np.int32)
    _check(lib().orc_roi_level(_p(rois), rois.shape[0], num_levels, ctypes.c_float(finest_scale), _p(lvl)),
           "roi_level")
    return lvl


def roi_align(feat, rois, spatial_scale, out_size=7, sampling_ratio=2, aligned=True):
    """Plain mmcv RoIAlign(avg) on one NCHW map."""
    feat = _f32(feat)
    rois = _f32(rois)
    N, C, H, W = feat.shape
    R = rois.shape[0]
    out = np.zeros((R, C, out_size, out_size), np.float32)
    _check(lib().orc_roi_align_level(_p(feat), N, C, H, W, _p(rois), R, None, 0, ctypes.c_float(spatial_scale),
                                     out_size, out_size, sampling_ratio, int(aligned), _p(out)), "roi_align")
    return out


def roi_extract(feats, rois, strides, out_size=7, sampling_ratio=2, finest_scale=56.0):
    """mmdet SingleRoIExtractor over `len(strides)` NCHW maps."""
    rois = _f32(rois)
    R = rois.shape[0]
    C = feats[0].shape[1]
    lvl = roi_levels(rois, len(strides), finest_scale)
    out = np.zeros((R, C, out_size, out_size), np.float32)
    for i, (f, s) in enumerate(zip(feats, strides)):
        f = _f32(f)
        N, _, H, W = f.shape
        _check(lib().orc_roi_align_level(_p(f), N, C, H, W, _p(rois), R, _p(lvl), i, ctypes.c_float(1.0 / s),
                                         out_size, out_size, sampling_ratio, 1, _p(out)), "roi_align_level")
    return out, lvl


# ---- layer-by-layer sparse encoder on top of the primitives (reference: sparse_encoder_custom.py:109-140) ----

def bn_fold(gamma, beta, mean, var, eps):
    """alpha = gamma / sqrt(var + eps); beta' = beta - mean * alpha, in float32 like torch's CPU batch_norm."""
    gamma, beta, mean, var = (np.asarray(t, np.float32) for t in (gamma, beta, mean, var))
    alpha = (gamma / np.sqrt(var + np.float32(eps))).astype(np.float32)
    return alpha, (beta - mean * alpha).astype(np.float32)


# ---- the step before the path: CPU transforms of the reference's test pipeline (configs/nus/srfdet_voxel_nusc_LC.py:253-283) ----

def points_filter(points, pc_range=None, close_radius=0.0):
    """PointsRangeFilter = mmdet3d 1.0.0rc6 `BasePoints.in_range_3d` (strict: x > x_min & y > y_min & z > z_min & x < x_max &
    y < y_max & z < z_max; third party, parity unpinned) and, for close_radius > 0, `LoadPointsFromMultiSweeps._remove_close`
    (not (|x| < r and |y| < r)).  Order preserved.  -> (kept points, their source rows)."""
    p = np.asarray(points, np.float32)
    keep = np.ones(len(p), bool)
    if pc_range is not None:
        r = np.asarray(pc_range, np.float32)
        keep &= (p[:, 0] > r[0]) & (p[:, 1] > r[1]) & (p[:, 2] > r[2]) & (p[:, 0] < r[3]) & (p[:, 1] < r[4]) & (p[:, 2] < r[5])
    if close_radius > 0:
        rr = np.float32(close_radius)
        keep &= ~((np.abs(p[:, 0]) < rr) & (np.abs(p[:, 1]) < rr))
    idx = np.nonzero(keep)[0].astype(np.int32)
    return p[idx], idx


def image_prepare(images_u8, mean, std, to_rgb=False, size_divisor=32, size=None):
    """NormalizeMultiviewImage + PadMultiViewImage (mmdet3d_plugin/datasets/pipelines/transform_3d.py:7-93) + the HWC -> CHW
    transpose / stack of DefaultFormatBundle3D: float32 pixels, (x - mean) * (1 / float64(std) as float32) per channel after an
    optional BGR -> RGB swap (mmcv.imnormalize; third party, parity unpinned), zeros appended below / right up to a multiple
    of size_divisor.  (V, H, W, 3) uint8 -> (V, 3, Hp, Wp) float32."""
    img = np.asarray(images_u8)
    assert img.dtype == np.uint8 and img.ndim == 4 and img.shape[3] == 3
    V, H, W, _ = img.shape
    x = img.astype(np.float32)
    if to_rgb:
        x = x[..., ::-1]
    m = np.asarray(mean, np.float32).reshape(1, 1, 1, 3)
    inv = (1.0 / np.asarray(std, np.float32).astype(np.float64)).astype(np.float32).reshape(1, 1, 1, 3)
    y = ((x - m).astype(np.float32) * inv).astype(np.float32)
    if size is not None:
        Hp, Wp = int(size[0]), int(size[1])
    else:
        Hp, Wp = -(-H // size_divisor) * size_divisor, -(-W // size_divisor) * size_divisor
    out = np.zeros((V, 3, Hp, Wp), np.float32)
    out[:, :, :H, :W] = y.transpose(0, 3, 1, 2)
    return out
