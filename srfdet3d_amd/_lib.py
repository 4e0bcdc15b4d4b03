"""ctypes binding of libsrfdet3d_hip.so (the C ABI declared in include/srfdet3d.h).

There is no CPU fallback: if the library is missing or an op is handed a non-GPU tensor this module raises.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_longlong, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsrfdet3d_hip.so")
# the developer build (`python -m srfdet3d_amd.build --dev`: ablation kernels with wrong outputs by design) has its own file
# name and is loaded only when SRF_DEV_LIB=1 asks for it (tools/stamp_wino.py); it can never sit at LIB_PATH
DEV_LIB_PATH = os.path.join(_HERE, "libsrfdet3d_hip_dev.so")

_lib = None


class FeatMap(ctypes.Structure):
    """srf_featmap of include/srfdet3d.h."""
    _fields_ = [("data", c_void_p), ("N", c_int), ("H", c_int), ("W", c_int),
                ("stride_n", c_int64), ("stride_c", c_int64), ("stride_h", c_int64), ("stride_w", c_int64),
                ("spatial_scale", c_float)]


_P = c_void_p
_HF = POINTER(c_float)  # host float array
_HI = POINTER(c_int)    # host int array

# name -> (restype, argtypes); must list every symbol include/srfdet3d.h declares
SIGNATURES = {
    "srf_abi_version": (c_int, []),
    "srf_build_flavour": (c_int, []),
    "srf_error_string": (c_char_p, [c_int]),
    "srf_device_count": (c_int, []),
    "srf_points_filter_workspace_bytes": (c_size_t, [c_int]),
    "srf_points_filter": (c_int, [_P, c_int, c_int, _HF, c_float, _P, _P, _P, _P, _P]),
    "srf_image_prepare": (c_int, [_P, c_int, c_int, c_int, _HF, _HF, c_int, c_int, c_int, _P, _P]),
    "srf_dynamic_voxelize": (c_int, [_P, c_int, c_int, _HF, _HF, _HI, _P, _P]),
    "srf_hard_voxelize_workspace_bytes": (c_size_t, [c_int, c_int]),
    "srf_hard_voxelize": (c_int, [_P, c_int, c_int, _HF, _HF, _HI, c_int, c_int, _P, _P, _P, _P, _P, c_int, _P,
                                  c_size_t, _P]),
    "srf_hard_voxelize_static": (c_int, [_P, c_int, c_int, _HF, _HF, _HI, c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int, _P,
                                         c_size_t, _P]),
    "srf_voxel_unique_workspace_bytes": (c_size_t, [c_int, _HI, c_int]),
    "srf_voxel_unique": (c_int, [_P, c_int, _HI, c_int, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "srf_scatter_reduce": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P]),
    "srf_coord_table_capacity": (c_int, [c_int]),
    "srf_coord_table_bytes": (c_size_t, [c_int]),
    "srf_coord_table_build": (c_int, [_P, c_int, _HI, c_int, _P, c_int, _P]),
    "srf_rulebook_subm": (c_int, [_P, c_int, _HI, _HI, _P, c_int, _P, _P, _P]),
    "srf_strided_max_outputs": (c_int, [c_int, c_int, _HI, _HI, _HI, _HI]),
    "srf_rulebook_strided_workspace_bytes": (c_size_t, [c_int, _HI, c_int]),
    "srf_rulebook_strided_outputs": (c_int, [_P, c_int, _HI, c_int, _HI, _HI, _HI, _P, _P, _P, c_int, _P, c_size_t,
                                             _P]),
    "srf_rulebook_strided_pairs": (c_int, [c_int, _HI, _P, c_int, _P, c_int, _P, _P, _P]),
    "srf_spconv_fwd": (c_int, [_P, c_int, c_int, _P, c_int, _P, c_int, c_int, c_int, _P, _P, _P, c_int, _P, _P, _P]),
    "srf_spconv_transpose_rulebook": (c_int, [_P, c_int, c_int, c_int, _P, c_int, _P]),
    "srf_spconv_bwd_data": (c_int, [_P, c_int, c_int, _P, c_int, _P, c_int, c_int, c_int, _P, _P]),
    "srf_spconv_bwd_weight": (c_int, [_P, c_int, c_int, _P, c_int, c_int, _P, c_int, c_int, _P, _P]),
    "srf_spconv_packed_weight_bytes": (c_size_t, [c_int, c_int, c_int]),
    "srf_spconv_pack_weights": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "srf_spconv_fwd_packed": (c_int, [_P, c_int, c_int, _P, c_int, _P, c_int, c_int, c_int, _P, _P, _P, c_int, _P, _P, _P, _P]),
    "srf_conv_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "srf_conv_wgrad_nhwc": (c_int, [_P, c_int64, _P, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_size_t, _P, _P]),
    "srf_spconv_tiles_count": (c_int, [c_int]),
    "srf_spconv_tiles_row_cost": (c_int, []),
    "srf_spconv_tiles_workspace_bytes": (c_size_t, [c_int]),
    "srf_spconv_tiles_build": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P, _P]),
    "srf_spconv_order_ints": (c_size_t, [c_int, c_int]),
    "srf_spconv_order_build": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P]),
    "srf_densify": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    "srf_roi_extract": (c_int, [POINTER(FeatMap), c_int, c_int, _P, c_int, c_int, c_int, c_float, _P, c_int64,
                                c_int64, c_int64, c_int, _P, _P]),
    "srf_roi_extract_sum": (c_int, [POINTER(FeatMap), c_int, c_int, _P, c_int, c_int, c_int, c_int, c_float, _P, c_int64, c_int64, c_int64, _P]),
    "srf_linear_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "srf_linear": (c_int, [_P, c_int, c_int, c_int, _P, c_int, c_int, _P, _P, _P, c_float, c_int, _P, c_int, _P, _P, c_float,
                           c_int, _P, c_int, _P, c_size_t, _P]),
    "srf_self_attention": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "srf_self_attention_batched": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    "srf_dynconv_mid": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, c_float, _P, _P, c_float, _P, _P]),
    "srf_bitmap_words": (c_size_t, [_HI, c_int]),
    "srf_bitmap_pair_count_ints": (c_size_t, []),
    "srf_bitmap_workspace_bytes": (c_size_t, [c_size_t]),
    "srf_bitmap_build": (c_int, [_P, c_int, _HI, c_int, _P, _P, _P, _P, _P, c_size_t, _P]),
    "srf_densify_bev": (c_int, [_P, c_int, c_int, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "srf_bitmap_build_padded": (c_int, [_P, c_int, _HI, c_int, _P, _P, _P, _P, _P, c_size_t, _P]),
    "srf_bitmap_rulebook_subm": (c_int, [_P, c_int, _HI, c_int, _HI, _P, _P, _P, _P, _P]),
    "srf_bitmap_strided_outputs": (c_int, [_P, c_int, _HI, c_int, _HI, _HI, _HI, _P, _P, _P, c_int, _P, _P, c_size_t, _P]),
    "srf_bitmap_strided_outputs_static": (c_int, [_P, c_int, _HI, c_int, _HI, _HI, _HI, _P, _P, _P, c_int, _P, _P, c_size_t, _P]),
    "srf_bitmap_strided_pairs": (c_int, [_P, _P, c_int, _HI, c_int, _HI, _HI, _HI, _P, _P, c_int, _P, c_int, c_int, _P, _P]),
    "srf_upsample_add": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "srf_maxpool3s2_ceil": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "srf_ese_gate": (c_int, [_P, c_int, c_int, _P, _P, _P, _P]),
    "srf_nchw_to_nhwc": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "srf_dwconv3x3s2": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, _P, _P]),
    "srf_channel_affine": (c_int, [_P, c_int, c_int, c_int, c_longlong, _P, _P, c_int, _P, c_int, _P, c_longlong, _P]),
    "srf_conv1x1_packed_weight_bytes": (c_size_t, [c_int, c_int]),
    "srf_conv1x1_pack_weights": (c_int, [_P, c_int, c_int, _P, _P]),
    "srf_conv1x1": (c_int, [POINTER(c_void_p), _HI, c_int, c_int, c_int, _P, c_int, _P, _P, c_int, _P, _P]),
    "srf_wino43_packed_weight_bytes": (c_size_t, [c_int, c_int]),
    "srf_wino43_pack_weights": (c_int, [_P, c_int, c_int, _P, _P]),
    "srf_wino43_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "srf_wino43": (c_int, [_P, c_int, c_int, c_int, c_int, c_longlong, _P, c_int, _P, _P, c_int, _P, c_longlong, _P, c_size_t, _P]),
    "srf_wino43_transform": (c_int, [_P, c_int, c_int, c_int, c_int, c_longlong, c_int, _P, c_size_t, _P]),
    "srf_wino43_multiply": (c_int, [_P, c_size_t, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, c_int, _P, c_longlong, _P]),
    "srf_wino3x3_packed_weight_bytes": (c_size_t, [c_int, c_int]),
    "srf_wino3x3_pack_weights": (c_int, [_P, c_int, c_int, _P, _P]),
    "srf_wino3x3": (c_int, [_P, c_int, c_int, c_int, c_int, c_longlong, _P, c_int, _P, _P, c_int, _P, c_longlong, _P]),
    "srf_conv1x1_nhwc_packed_weight_bytes": (c_size_t, [c_int, c_int]),
    "srf_conv1x1_nhwc_pack_weights": (c_int, [_P, c_int, c_int, _P, _P]),
    "srf_conv1x1_nhwc_direct_packed_weight_bytes": (c_size_t, [c_int, c_int]),
    "srf_conv1x1_nhwc_direct_pack_weights": (c_int, [_P, c_int, c_int, _P, _P]),
    "srf_conv1x1_nhwc_direct": (c_int, [_P, c_longlong, c_int, c_longlong, _P, c_int, _P, _P, c_int, _P, c_longlong, _P]),
    "srf_conv1x1_nhwc_direct_topdown": (c_int, [_P, c_int, c_int, c_int, c_int, c_longlong, _P, c_int, _P, _P, c_int, _P, c_int, c_int, c_longlong,
                                                _P, c_longlong, _P]),
    "srf_conv1x1_nhwc_direct_pooled": (c_int, [_P, c_int, c_longlong, c_int, c_longlong, _P, c_int, _P, _P, c_int, _P, c_longlong, _P, _P,
                                               c_size_t, _P]),
    "srf_conv1x1_nhwc": (c_int, [_P, c_longlong, c_int, c_longlong, _P, c_int, _P, _P, c_int, _P, c_longlong, _P]),
    "srf_conv1x1_nhwc_split_packed_weight_bytes": (c_size_t, [c_int, c_int]),
    "srf_conv1x1_nhwc_split_pack_weights": (c_int, [_P, c_int, c_int, _P, _P]),
    "srf_conv1x1_nhwc_split": (c_int, [_P, c_longlong, c_int, c_longlong, _P, c_int, _P, _P, c_int, _P, c_longlong, _P]),
    "srf_conv1x1_nhwc_split_topdown": (c_int, [_P, c_int, c_int, c_int, c_int, c_longlong, _P, c_int, _P, _P, c_int, _P, c_int, c_int, c_longlong,
                                               _P, c_longlong, _P]),
    "srf_conv_gemm_nhwc_split": (c_int, [_P, c_int, c_int, c_int, c_int, c_longlong, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, _P,
                                         c_longlong, _P]),
    "srf_conv1x1_nhwc_split_pooled": (c_int, [_P, c_int, c_longlong, c_int, c_longlong, _P, c_int, _P, _P, c_int, _P, c_longlong, _P, _P,
                                              c_size_t, _P]),
    "srf_conv1x1_nhwc_topdown": (c_int, [_P, c_int, c_int, c_int, c_int, c_longlong, _P, c_int, _P, _P, c_int, _P, c_int, c_int, c_longlong,
                                         _P, c_longlong, _P]),
    "srf_conv1x1_nhwc_pooled_workspace_bytes": (c_size_t, [c_int, c_longlong, c_int]),
    "srf_conv1x1_nhwc_pooled": (c_int, [_P, c_int, c_longlong, c_int, c_longlong, _P, c_int, _P, _P, c_int, _P, c_longlong, _P, _P,
                                        c_size_t, _P]),
    "srf_conv_gemm_nhwc": (c_int, [_P, c_int, c_int, c_int, c_int, c_longlong, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, _P,
                                   c_longlong, _P]),
    "srf_stem_conv_nchw": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, c_int, _P, c_longlong, _P]),
    "srf_nhwc_affine": (c_int, [_P, c_longlong, c_int, c_longlong, c_int, _P, c_int, _P, _P, c_longlong, c_int, _P, c_longlong, _P]),
    "srf_nhwc_colmean_workspace_bytes": (c_size_t, [c_int, c_int]),
    "srf_nhwc_colmean": (c_int, [_P, c_longlong, c_int, c_longlong, c_int, _P, _P, c_size_t, _P]),
    "srf_nhwc_colsum_prod": (c_int, [_P, c_longlong, _P, c_longlong, c_int, c_longlong, c_int, _P, _P, c_size_t, _P]),
    "srf_nhwc_maxpool3s2_ceil": (c_int, [_P, c_longlong, c_int, c_int, c_int, c_int, _P, c_longlong, _P]),
    "srf_nhwc_upsample_add": (c_int, [_P, c_longlong, _P, c_longlong, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_longlong, _P]),
    "srf_nhwc_dwconv3x3s2": (c_int, [_P, c_longlong, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, _P, c_longlong, _P]),
    "srf_nhwc_dwconv3x3s2_cat": (c_int, [_P, c_longlong, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, _P, c_longlong, _P, c_longlong,
                                         c_int, _P, _P]),
    "srf_nhwc_affine_relu_bwd_workspace_bytes": (c_size_t, [c_longlong, c_int]),
    "srf_nhwc_affine_relu_bwd": (c_int, [_P, c_longlong, _P, c_longlong, c_longlong, c_int, _P, c_int, _P, c_longlong, _P, _P, c_size_t, _P]),
    "srf_bn_eval_fold": (c_int, [_P, _P, _P, _P, c_float, c_int, _P, _P]),
    "srf_bn_eval_grads": (c_int, [_P, _P, _P, c_int, _P, _P]),
    "srf_nhwc_affine_relu_bwd2": (c_int, [_P, c_longlong, _P, c_longlong, _P, c_longlong, c_longlong, c_int, _P, c_int, _P, c_longlong, _P, _P,
                                          c_size_t, _P]),
    "srf_ese_apply": (c_int, [_P, c_longlong, c_int, c_longlong, c_int, _P, _P, _P, _P, c_longlong, _P, c_longlong, _P, _P]),
    "srf_nhwc_pool_sum": (c_int, [_P, c_longlong, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    "srf_dpg_mix": (c_int, [_P, _P, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, _P, _P]),
    "srf_stage_tail": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, c_float, c_int, POINTER(c_void_p), POINTER(c_void_p),
                               POINTER(c_void_p), _HF, c_int, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), _HF, _P, _P,
                               c_int, _P, _P, c_int, _P, _HF, _HF, c_float, _P, _P, _P, _P, c_size_t, _P]),
    "srf_stage_tail_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "srf_apply_deltas": (c_int, [_P, _P, c_int, c_int, _HF, _HF, c_float, _P, _P]),
    "srf_host_pack": (c_int, [_P, c_int, _P, c_int, _P, c_int, _P, _P]),
    "srf_decode_boxes": (c_int, [_P, _P, c_int, c_int, c_int, _HF, _P, _P, _P]),
    "srf_nms_rotated_workspace_bytes": (c_size_t, [c_int]),
    "srf_nms_rotated": (c_int, [_P, c_int, c_float, _P, _P, c_size_t, _P]),
    "srf_nms_rotated_counted": (c_int, [_P, c_int, _P, c_float, _P, _P, c_size_t, _P]),
    "srf_nms_rotated_classes": (c_int, [_P, _P, c_int, _P, c_float, _P, _P, c_size_t, _P]),
    "srf_nms_select": (c_int, [_P, _P, c_int, c_int, c_int, c_float, c_int, _P, _P, _P, _P, _P, _P]),
    "srf_nms_finish": (c_int, [_P, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "srf_roi_extract_bwd": (c_int, [POINTER(FeatMap), POINTER(c_void_p), c_int, c_int, _P, c_int, c_int, c_int, c_float, _P,
                                    c_int64, c_int64, c_int64, _P]),
    "srf_box_iou_rotated": (c_int, [_P, c_int, _P, c_int, _P, _P]),
    "srf_box_rois": (c_int, [_P, c_int, c_int, c_int, _HF, _HF, c_int, _P, _P, c_int, _P, _P]),
}


def lib():
    """The loaded library; raises RuntimeError if it has not been built."""
    global _lib
    if _lib is None:
        want_dev = os.environ.get("SRF_DEV_LIB", "0") == "1"
        path = DEV_LIB_PATH if want_dev else LIB_PATH
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: the HIP extension has not been built. Run `python -m srfdet3d_amd.build"
                f"{' --dev' if want_dev else ''}` (or __graft_entry__.build()). srfdet3d_amd has no CPU fallback.")
        handle = ctypes.CDLL(path)
        flavour = getattr(handle, "srf_build_flavour", None)
        if flavour is None:
            raise RuntimeError(f"{path} does not export srf_build_flavour; rebuild it")
        flavour.restype = c_int
        if flavour() != (1 if want_dev else 0):
            raise RuntimeError(f"{path} is a {'production' if want_dev else 'DEVELOPER (-DSRF_DEV, wrong-by-design ablation kernels)'} "
                               "build under the other's name; rebuild with `python -m srfdet3d_amd.build --force`")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name, None)
            if fn is None:
                raise RuntimeError(f"{path} does not export {name}; rebuild it")
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, op):
    if rc != 0:
        msg = lib().srf_error_string(rc)
        raise RuntimeError(f"srfdet3d {op} failed ({rc}): {msg.decode() if msg else '?'}")


def hf(values):
    return (c_float * len(values))(*[float(v) for v in values])


def hi(values):
    return (c_int * len(values))(*[int(v) for v in values])
