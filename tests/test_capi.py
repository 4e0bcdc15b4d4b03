"""The C-ABI library: loads without a GPU, exports every symbol include/srfdet3d.h declares, host-side helpers
behave, and the Python layer refuses CPU tensors (there is no CPU fallback).  No compute call is made here."""
import ctypes
import os
import re

import pytest
import torch

from srfdet3d_amd import _lib, ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "srfdet3d.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(srf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 20
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/srfdet3d.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes signature table and header disagree"


def test_host_side_helpers():
    L = _lib.lib()
    assert L.srf_abi_version() == 1
    assert L.srf_error_string(0) == b"ok" and L.srf_error_string(-2) == b"workspace too small"
    cap = L.srf_coord_table_capacity(30000)
    assert cap >= 60000 and cap & (cap - 1) == 0
    assert L.srf_coord_table_bytes(cap) == cap * 8
    assert L.srf_hard_voxelize_workspace_bytes(30000, 10) > 65536 * 12 * 4
    hi = _lib.hi
    assert L.srf_strided_max_outputs(1000, 1, hi([41, 1472, 1472]), hi([3, 3, 3]), hi([2, 2, 2]), hi([1, 1, 1])) == 8000
    assert L.srf_strided_max_outputs(1000, 1, hi([5, 184, 184]), hi([3, 1, 1]), hi([2, 1, 1]), hi([0, 0, 0])) == 2000
    assert L.srf_strided_max_outputs(10 ** 6, 1, hi([5, 8, 8]), hi([3, 1, 1]), hi([2, 1, 1]), hi([0, 0, 0])) == 2 * 8 * 8
    assert L.srf_nms_rotated_workspace_bytes(900) == 900 * 15 * 8
    assert L.srf_voxel_unique_workspace_bytes(1000, hi([40, 1600, 1408]), 1) > 40 * 1600 * 1408 // 8
    assert L.srf_voxel_unique_workspace_bytes(1000, hi([41, 1472, 1472]), 64) == 0  # exceeds 32-bit keys
    # argument validation happens before any HIP call
    assert L.srf_spconv_fwd(None, 0, 16, None, 27, None, 0, 10, 24, None, None, None, 0, None, None, None) == -1
    assert L.srf_roi_extract(None, 0, 128, None, 0, 7, 2, 56.0, None, 0, 0, 0, 0, None, None) == -1
    assert L.srf_spconv_fwd_packed(None, 0, 128, None, 27, None, 0, 10, 128, None, None, None, 0, None, None, None, None) == -1
    assert L.srf_spconv_tiles_count(35000) == 512 and L.srf_spconv_tiles_count(100) == 3 and L.srf_spconv_tiles_count(0) == 1
    assert L.srf_spconv_tiles_workspace_bytes(35000) >= 35000 * 4
    assert L.srf_spconv_tiles_build(None, 0, 27, 10, None, None, None, None) == -1
    # round 5: the row plan of the 32-channel sparse convolutions, the weight-gradient kernel, the training helpers
    assert L.srf_spconv_tiles_row_cost() >= 0
    assert L.srf_spconv_order_ints(60451, 27) == ((60451 + 1023) // 1024 * 1024) * 28   # [order | 27 sorted rulebook rows], padded to 1024
    assert L.srf_spconv_order_ints(0, 27) == 0 and L.srf_spconv_order_ints(100, 28) == 0
    assert L.srf_spconv_order_build(None, 5, 27, 10, None, None, None) == -1          # row stride below the row count / no tables
    assert L.srf_spconv_order_build(None, 0, 27, 0, None, None, None) == 0            # an empty level is nothing to do
    assert L.srf_conv_wgrad_workspace_bytes(12, 58, 100, 192, 192, 3) >= 192 * 192 * 9 * 4
    assert L.srf_conv_wgrad_workspace_bytes(12, 58, 100, 192, 192, 2) == 0            # 1x1 and 3x3 only
    assert L.srf_conv_wgrad_nhwc(None, 192, None, 192, 12, 58, 100, 192, 192, 3, None, 0, None, None) == -1
    assert L.srf_nhwc_affine_relu_bwd2(None, 8, None, 0, None, 8, 10, 7, None, 1, None, 8, None, None, 0, None) == -3   # C % 4
    assert L.srf_nhwc_colsum_prod(None, 8, None, 8, 2, 10, 8, None, None, 0, None) == -1
    assert L.srf_bn_eval_fold(None, None, None, None, 1e-5, 8, None, None) == -1
    assert L.srf_bn_eval_grads(None, None, None, 8, None, None) == -1
    assert L.srf_points_filter_workspace_bytes(30000) >= 8 and L.srf_points_filter_workspace_bytes(-1) == 0
    assert L.srf_points_filter(None, 10, 2, None, 0.0, None, None, None, None, None) == -1      # nf < 3 / no counter
    assert L.srf_image_prepare(None, 6, 900, 1600, _lib.hf([0, 0, 0]), _lib.hf([1, 1, 1]), 0, 928, 1602, None, None) == -1  # Wp % 4
    assert L.srf_image_prepare(None, 6, 900, 1600, _lib.hf([0, 0, 0]), _lib.hf([1, 0, 1]), 0, 928, 1600, None, None) == -1  # std 0
    # bitmap-rank rulebooks / dense-side helpers: sizes and argument checks, still no GPU work
    assert L.srf_bitmap_words(hi([41, 1472, 1472]), 1) == (41 * 1472 * 1472 + 31) // 32
    assert L.srf_bitmap_words(hi([41, 1472, 1472]), 64) == 0          # cell index must fit 32 bits
    assert L.srf_bitmap_workspace_bytes(2776000) >= (2776000 // 2048 + 1) * 4
    assert L.srf_bitmap_pair_count_ints() >= 27
    assert L.srf_bitmap_build(None, -1, hi([41, 1472, 1472]), 1, None, None, None, None, None, 0, None) == -1
    assert L.srf_conv1x1_packed_weight_bytes(256, 768) == 256 * 768 * 4
    assert L.srf_conv1x1_packed_weight_bytes(250, 768) == 0           # Cout and K must be multiples of 32
    assert L.srf_conv1x1(None, None, 0, 1, 16, None, 128, None, None, 0, None, None) == -1
    assert L.srf_channel_affine(None, -1, 4, 16, 64, None, None, 0, None, 0, None, 64, None) == -1
    assert L.srf_nms_rotated_counted(None, 10, None, 0.4, None, None, 0, None) == -1
    assert L.srf_stage_tail(None, 10, 64, 512, *([None] * 6), 1e-5, 2, None, None, None, None, 3, None, None, None, None,
                            None, None, 10, None, None, 10, None, None, None, 5.0, None, None, None, None, 0, None) == -3  # C != 128


def test_ops_refuse_cpu_tensors():
    pts = torch.zeros(10, 5)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        ops.hard_voxelize(pts, [0.1, 0.1, 0.1], [0, 0, 0, 1, 1, 1], 5, 10)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        ops.coord_table_build(torch.zeros(4, 4, dtype=torch.int32), [4, 4, 4], 1)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        ops.box_rois(torch.zeros(1, 4, 10), [0, 0, 0, 1, 1, 1], [0.1, 0.1, 0.1])


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_developer_build_is_refused_as_the_product_library(monkeypatch, tmp_path):
    """`build --dev` (ablation kernels, wrong outputs by design) writes libsrfdet3d_hip_dev.so; a library
    that answers srf_build_flavour() == 1 under the production name is refused (and the reverse under SRF_DEV_LIB=1)."""
    import shutil
    import subprocess
    from srfdet3d_amd import build as B
    assert os.path.basename(B.DEV_LIB) != os.path.basename(B.LIB)
    assert _lib.lib().srf_build_flavour() == 0
    src = tmp_path / "fake.c"
    src.write_text("int srf_build_flavour(void) { return 1; }\n")
    fake = tmp_path / "libsrfdet3d_hip.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", str(fake), str(src)])
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(fake))
    with pytest.raises(RuntimeError, match="DEVELOPER"):
        _lib.lib()
    # and the production library is not accepted where the developer library was asked for
    monkeypatch.setenv("SRF_DEV_LIB", "1")
    monkeypatch.setattr(_lib, "DEV_LIB_PATH", str(tmp_path / "prod_as_dev.so"))
    shutil.copy(B.LIB, str(tmp_path / "prod_as_dev.so"))
    with pytest.raises(RuntimeError, match="production"):
        _lib.lib()
