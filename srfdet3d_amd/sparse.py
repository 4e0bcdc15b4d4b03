"""spconv-shaped Python surface over the HIP rulebook / sparse-conv kernels.

Mirrors the operator boundary the reference uses: `SparseConvTensor(features, indices, spatial_shape,
batch_size)` with `.features/.indices/.dense()`, `SubMConv3d` / `SparseConv3d(in, out, kernel_size, stride, padding,
bias=False, indice_key)`, `SparseSequential`, and mmdet3d's `SparseBasicBlock` / `make_sparse_convmodule`
(reference call sites: mmdet3d_plugin/models/middle_encoders/sparse_encoder_custom.py:7-15, :73-107, :123-138,
:182-211).  Parameter names and shapes follow spconv 1.x / mmcv (`weight`: kD,kH,kW,Cin,Cout) so reference
checkpoints load by key; spconv 2.x layout (Cout,kD,kH,kW,Cin) is converted on load.

What is different from spconv, by design:
  * rulebooks are output-stationary neighbour tables (K, A_out) and are cached per coordinate set, so the 16
    un-keyed SubM convs of the basic blocks (which the reference rebuilds from scratch) share one table per level;
  * in eval mode conv -> BatchNorm1d -> ReLU (and the residual add of SparseBasicBlock) run as ONE kernel.
"""
import math

import torch
from torch import nn

from . import ops
from .compat.cnn import build_norm_layer
from .compat.registry import CONV_LAYERS


def _triple(v):
    if isinstance(v, (list, tuple)):
        assert len(v) == 3
        return [int(x) for x in v]
    return [int(v)] * 3


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, indice_dict=None, num_rows=None):
        # num_rows: device int with the number of real rows when the arrays are padded to a capacity (static shapes)
        self.num_rows = num_rows
        self.features = features
        self.indices = indices if indices.dtype == torch.int32 else indices.int()
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = indice_dict if indice_dict is not None else {}

    def replace_feature(self, features):
        return SparseConvTensor(features, self.indices, self.spatial_shape, self.batch_size, self.indice_dict, self.num_rows)

    # coordinate table / SubM rulebook of this tensor's active set, built once and shared along the layer chain
    def _level_key(self):
        return (self.indices.data_ptr(), self.indices.shape[0], tuple(self.spatial_shape))

    def coord_table(self):
        key = ("table",) + self._level_key()
        if key not in self.indice_dict:
            self.indice_dict[key] = ops.coord_table_build(self.indices, self.spatial_shape, self.batch_size)
        return self.indice_dict[key]

    def bitmap_level(self):
        """The level's occupancy bitmap if the rows are known to be sorted by (b, y, x, z) (see `sorted_by_bitmap`)."""
        return self.indice_dict.get(("bitmap",) + self._level_key())

    @staticmethod
    def sorted_by_bitmap(features, indices, spatial_shape, batch_size, static_caps=None):
        """Reorders distinct active sites into (b, y, x, z) order and attaches the level's bitmap: every rulebook down the
        layer chain is then built by bitmap rank (ops.rulebook_*_bitmap) instead of hash tables, and every active
        set stays sorted.  The dense result of an encoder does not depend on the row order."""
        lvl, order, sorted_idx = ops.bitmap_build(indices if indices.dtype == torch.int32 else indices.int(), spatial_shape,
                                                  batch_size, padded=static_caps is not None)
        t = SparseConvTensor(torch.index_select(features, 0, order), sorted_idx, spatial_shape, batch_size)  # int32 index: one launch
        t.indice_dict[("bitmap",) + t._level_key()] = lvl
        if static_caps is not None:
            # static-shape mode (whole-frame hipGraph): `indices` may carry padding rows (b < 0); every strided conv
            # below produces static_caps[indice_key] rows, the surplus being padding, and appends its device-side
            # output count to "counts" for the caller's overflow check
            t.indice_dict["static"] = dict(caps={k: v for k, v in static_caps.items() if k != "__rows__"}, counts=[])
            t.num_rows = static_caps.get("__rows__")
        return t

    def subm_rulebook(self, ksize):
        key = ("subm", tuple(ksize)) + self._level_key()
        if key not in self.indice_dict:
            lvl = self.bitmap_level()
            if lvl is not None:
                self.indice_dict[key] = ops.rulebook_subm_bitmap(self.indices, lvl, ksize,
                                                                 want_counts="static" not in self.indice_dict)
            else:
                self.indice_dict[key] = ops.rulebook_subm(self.indices, self.spatial_shape, ksize, self.coord_table())
        return self.indice_dict[key]

    def dense(self, channels_first=True):
        out = ops.densify(self.features, self.indices, self.batch_size, self.spatial_shape)
        return out if channels_first else out.permute(0, 2, 3, 4, 1)

    def dense_bev(self):
        """dense() viewed as (N, C * D, H, W) (sparse_encoder_custom.py:144-147).  GPU inference on a bitmap-ranked level: one pass that
        writes the map channels-last for the dense backbone (ops.densify_bev); same values, other strides."""
        lvl = self.bitmap_level()
        C = self.features.shape[1]
        if (lvl is not None and self.features.is_cuda and not (torch.is_grad_enabled() and self.features.requires_grad)
                and (C * self.spatial_shape[0]) % 4 == 0 and self.features.dtype == torch.float32):
            return ops.densify_bev(self.features, lvl, self.batch_size, self.spatial_shape)
        dense = self.dense()
        N, C, D, H, W = dense.shape
        return dense.view(N, C * D, H, W)

    @property
    def spatial_size(self):
        return int(torch.tensor(self.spatial_shape).prod())


def _fold_bn(bn):
    """alpha = gamma / sqrt(var + eps), beta = bias - mean * alpha, cached until any BN tensor changes."""
    vers = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
            bn.weight.data_ptr(), bn.running_var.data_ptr())
    cache = getattr(bn, "_srf_fold", None)
    if cache is None or cache[0] != vers:
        with torch.no_grad():
            alpha = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            beta = bn.bias - bn.running_mean * alpha
        cache = (vers, alpha.contiguous(), beta.contiguous())
        bn._srf_fold = cache
    return cache[1], cache[2]


def _bn_foldable(bn):
    return isinstance(bn, nn.BatchNorm1d) and not bn.training and bn.track_running_stats and bn.affine


class SparseModule(nn.Module):
    pass


class _SparseConv(SparseModule):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=False,
                 indice_key=None, subm=False):
        super().__init__()
        assert groups == 1 and _triple(dilation) == [1, 1, 1], "only what the reference encoders use"
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = _triple(kernel_size), _triple(stride), _triple(padding)
        self.subm = subm
        self.indice_key = indice_key
        self.weight = nn.Parameter(torch.empty(*self.kernel_size, in_channels, out_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        w = state_dict.get(prefix + "weight")
        if w is not None and tuple(w.shape) != tuple(self.weight.shape) and w.dim() == 5 and \
                tuple(w.shape) == (self.out_channels, *self.kernel_size, self.in_channels):
            state_dict[prefix + "weight"] = w.permute(1, 2, 3, 4, 0).contiguous()  # spconv 2.x -> 1.x layout
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _rulebook(self, x):
        if self.subm:
            nbr, counts = x.subm_rulebook(self.kernel_size)
            return nbr, counts, x.indices, x.spatial_shape, x.num_rows
        key = ("strided", self.indice_key, tuple(self.kernel_size), tuple(self.stride), tuple(self.padding)) + \
            x._level_key()
        if key not in x.indice_dict:
            lvl = x.bitmap_level()
            static = x.indice_dict.get("static")
            if lvl is not None and static is not None:
                cap = static["caps"][self.indice_key]
                out_idx, nbr, counts, out_lvl, oshape, num_out = ops.rulebook_strided_bitmap(
                    x.indices, lvl, self.kernel_size, self.stride, self.padding, out_capacity=cap)
                static["counts"].append((self.indice_key, num_out, cap))
                x.indice_dict[("rows", out_idx.data_ptr())] = num_out
                x.indice_dict[("bitmap", out_idx.data_ptr(), out_idx.shape[0], tuple(oshape))] = out_lvl
            elif lvl is not None:
                out_idx, nbr, counts, out_lvl, oshape = ops.rulebook_strided_bitmap(x.indices, lvl, self.kernel_size, self.stride,
                                                                                   self.padding)
                x.indice_dict[("bitmap", out_idx.data_ptr(), out_idx.shape[0], tuple(oshape))] = out_lvl
            else:
                out_idx, nbr, counts, table, oshape = ops.rulebook_strided(x.indices, x.spatial_shape, x.batch_size,
                                                                           self.kernel_size, self.stride, self.padding)
                x.indice_dict[("table", out_idx.data_ptr(), out_idx.shape[0], tuple(oshape))] = table
            x.indice_dict[key] = (out_idx, nbr, counts, oshape)
        out_idx, nbr, counts, oshape = x.indice_dict[key]
        return nbr, counts, out_idx, oshape, x.indice_dict.get(("rows", out_idx.data_ptr()))

    def forward(self, x, bn=None, relu=False, residual=None):
        """conv, optionally with an eval-mode BatchNorm1d, residual rows and ReLU fused into the same kernel.
        An index-only tensor (`x.features is None`) gets the layer's rulebook (and row ranges) built and cached and comes back
        index-only: the encoder runs that dry pass on a second stream, ahead of the convolutions (middle_encoders.py)."""
        nbr, counts, out_idx, oshape, rows_dev = self._rulebook(x)
        use_packed = not self.training and self.out_channels >= 32 and self.in_channels % 4 == 0
        tiles = None
        if (use_packed and ops.spconv_tiles_wanted(self.in_channels, self.out_channels) and nbr.shape[1] > 0
                and (self.subm or self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2] == 27)):
            # balanced row ranges, one set per rulebook: shared by the SubM layers of a level; a 27-offset strided conv gains more
            # from them than they cost (64 -> 128 on the nuScenes encoder: 92 -> 76 us, its steps 36 +- 37 -> 35 +- 19 per workgroup;
            # the three small launches of the cut run on the index stream of the graph, ahead of the convolutions); conv_out (3
            # offsets) runs on equal-height tiles
            if self.out_channels == 32:
                # 32-channel layers: a mask-sorted row order instead of ranges, from ~100k rows up (Waymo's levels)
                if ops.spconv_order_wanted(self.in_channels, self.out_channels, rows=nbr.shape[1]):
                    tkey = ("order", nbr.data_ptr(), nbr.shape[1])
                    tiles = x.indice_dict.get(tkey)
                    if tiles is None:
                        tiles = x.indice_dict[tkey] = ops.spconv_order(nbr, rows_dev)
            else:
                tkey = ("tiles", nbr.data_ptr(), nbr.shape[1])
                tiles = x.indice_dict.get(tkey)
                if tiles is None:
                    tiles = x.indice_dict[tkey] = ops.spconv_tiles(nbr, rows_dev)
        if x.features is None:
            return SparseConvTensor(None, out_idx, oshape, x.batch_size, x.indice_dict, rows_dev)
        K = self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
        w = self.weight.view(K, self.in_channels, self.out_channels)
        alpha = beta = None
        if bn is not None:
            alpha, beta = _fold_bn(bn)
            if self.bias is not None:
                beta = beta + self.bias * alpha
        elif self.bias is not None:
            alpha, beta = torch.ones_like(self.bias), self.bias
        packed = None
        if use_packed:
            vers = (self.weight._version, self.weight.data_ptr())
            cache = getattr(self, "_srf_packed", None)
            if cache is None or cache[0] != vers:
                with torch.no_grad():
                    cache = (vers, ops.pack_spconv_weights(w.detach()))
                self._srf_packed = cache
            packed = cache[1]
        feats = ops.spconv_fwd(x.features, w, nbr, alpha, beta, residual, relu, pair_counts=counts, packed=packed,
                               rows_dev=rows_dev, tiles=tiles, subm=self.subm)
        return SparseConvTensor(feats, out_idx, oshape, x.batch_size, x.indice_dict, rows_dev)


@CONV_LAYERS.register_module("SubMConv3d")
class SubMConv3d(_SparseConv):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, indice_key,
                         subm=True)


@CONV_LAYERS.register_module("SparseConv3d")
class SparseConv3d(_SparseConv):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, indice_key,
                         subm=False)


class SparseSequential(SparseModule):
    """spconv SparseSequential: sparse modules see the SparseConvTensor, dense modules see `.features`.
    A `conv, BatchNorm1d(eval), ReLU` run is issued as one fused kernel."""

    def __init__(self, *mods, **named):
        super().__init__()
        for i, m in enumerate(mods):
            self.add_module(str(i), m)
        for n, m in named.items():
            self.add_module(n, m)

    def add(self, module, name=None):
        self.add_module(name if name is not None else str(len(self._modules)), module)

    def __len__(self):
        return len(self._modules)

    def __getitem__(self, i):
        return list(self._modules.values())[i]

    def forward(self, x):
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, _SparseConv):
                bn = mods[i + 1] if i + 1 < len(mods) and _bn_foldable(mods[i + 1]) else None
                relu = bn is not None and i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
                x = m(x, bn=bn, relu=relu)
                i += 1 + (bn is not None) + relu
            elif isinstance(m, SparseModule):
                x = m(x)
                i += 1
            else:
                if not (isinstance(x, SparseConvTensor) and x.features is None):   # (an index-only dry pass has nothing for it)
                    x = x.replace_feature(m(x.features)) if isinstance(x, SparseConvTensor) else m(x)
                i += 1
        return x


class SparseBasicBlock(SparseModule):
    """mmdet3d SparseBasicBlock: conv1-bn1-relu-conv2-bn2-(+identity)-relu; attribute names conv1/bn1/conv2/bn2."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, conv_cfg=None, norm_cfg=None):
        super().__init__()
        assert downsample is None and stride == 1
        conv_type = (conv_cfg or dict(type="SubMConv3d"))["type"]
        cls = CONV_LAYERS.get(conv_type)
        self.conv1 = cls(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = build_norm_layer(norm_cfg, planes)[1]
        self.conv2 = cls(planes, planes, 3, padding=1, bias=False)
        self.bn2 = build_norm_layer(norm_cfg, planes)[1]
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        identity = x.features
        if identity is None:   # index-only dry pass: both layers share the level's rulebook
            return self.conv2(self.conv1(x))
        if _bn_foldable(self.bn1) and _bn_foldable(self.bn2):
            out = self.conv1(x, bn=self.bn1, relu=True)
            return self.conv2(out, bn=self.bn2, relu=True, residual=identity)
        out = self.conv1(x)
        out = out.replace_feature(self.relu(self.bn1(out.features)))
        out = self.conv2(out)
        return out.replace_feature(self.relu(self.bn2(out.features) + identity))


def make_sparse_convmodule(in_channels, out_channels, kernel_size, indice_key, stride=1, padding=0,
                           conv_type="SubMConv3d", norm_cfg=None, order=("conv", "norm", "act")):
    """mmdet3d.ops.make_sparse_convmodule: SparseSequential(conv(bias=False), BN1d, ReLU) in `order`."""
    assert isinstance(order, tuple) and len(order) <= 3 and set(order) | {"conv", "norm", "act"} == {"conv", "norm", "act"}
    layers = []
    for layer in order:
        if layer == "conv":
            cls = CONV_LAYERS.get(conv_type)
            if conv_type in ("SparseInverseConv3d", "SparseInverseConv2d", "SparseInverseConv1d"):
                raise NotImplementedError("inverse sparse convs are not on the SRFDet3D path")
            layers.append(cls(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False,
                              indice_key=indice_key))
        elif layer == "norm":
            layers.append(build_norm_layer(norm_cfg, out_channels)[1])
        elif layer == "act":
            layers.append(nn.ReLU(inplace=True))
    return SparseSequential(*layers)
