"""`SparseEncoderCustom` (mmdet3d_plugin/models/middle_encoders/sparse_encoder_custom.py:19-216): the 3-D sparse
conv encoder that turns voxel features into the dense BEV map.  Same constructor arguments, same module names
(conv_input / encoder_layers.encoder_layer{i} / conv_out), so checkpoints load by key."""
import torch

from ..compat.cnn import BaseModule
from ..compat.registry import MIDDLE_ENCODERS
from ..sparse import SparseBasicBlock, SparseConvTensor, SparseSequential, make_sparse_convmodule


_INDEX_STREAMS = {}   # device index -> the stream the index-only dry pass of `_layers` runs on


@MIDDLE_ENCODERS.register_module()
class SparseEncoderCustom(BaseModule):
    def __init__(self, in_channels, sparse_shape, order=("conv", "norm", "act"),
                 norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01), base_channels=16, output_channels=128,
                 encoder_channels=((16,), (32, 32, 32), (64, 64, 64), (64, 64, 64)),
                 encoder_paddings=((1,), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1)), block_type="conv_module",
                 init_cfg=None):
        super().__init__(init_cfg=init_cfg)
        assert block_type in ("conv_module", "basicblock")
        assert isinstance(order, tuple) and set(order) == {"conv", "norm", "act"}
        self.sparse_shape = list(sparse_shape)
        self.in_channels = in_channels
        self.order = order
        self.base_channels = base_channels
        self.output_channels = output_channels
        self.encoder_channels = encoder_channels
        self.encoder_paddings = encoder_paddings
        self.stage_num = len(encoder_channels)
        self.fp16_enabled = False
        # reorder the active sites spatially before the first conv (results per site are unchanged: every output is
        # an fma chain ordered by kernel offset and channel, never by row); see ops.spatial_order
        self.spatial_sort = True

        pre_act = order[0] != "conv"
        self.conv_input = make_sparse_convmodule(in_channels, base_channels, 3, norm_cfg=norm_cfg, padding=1,
                                                 indice_key="subm1", conv_type="SubMConv3d",
                                                 order=("conv",) if pre_act else ("conv", "norm", "act"))
        last = self._make_stages(norm_cfg, base_channels, block_type)
        self.conv_out = make_sparse_convmodule(last, output_channels, kernel_size=(3, 1, 1), stride=(2, 1, 1),
                                               norm_cfg=norm_cfg, padding=0, indice_key="spconv_down2",
                                               conv_type="SparseConv3d")

    def _make_stages(self, norm_cfg, in_channels, block_type):
        self.encoder_layers = SparseSequential()
        n_stage = len(self.encoder_channels)
        out_channels = in_channels
        for i, widths in enumerate(self.encoder_channels):
            widths = tuple(widths)
            blocks = []
            for j, out_channels in enumerate(widths):
                padding = tuple(self.encoder_paddings[i])[j]
                down = dict(stride=2, padding=padding, indice_key=f"spconv{i + 1}", conv_type="SparseConv3d")
                if block_type == "conv_module" and i != 0 and j == 0:
                    blocks.append(make_sparse_convmodule(in_channels, out_channels, 3, norm_cfg=norm_cfg, **down))
                elif block_type == "basicblock":
                    if j == len(widths) - 1 and i != n_stage - 1:
                        blocks.append(make_sparse_convmodule(in_channels, out_channels, 3, norm_cfg=norm_cfg, **down))
                    else:
                        blocks.append(SparseBasicBlock(out_channels, out_channels, norm_cfg=norm_cfg,
                                                       conv_cfg=dict(type="SubMConv3d")))
                else:
                    blocks.append(make_sparse_convmodule(in_channels, out_channels, 3, norm_cfg=norm_cfg,
                                                         padding=padding, indice_key=f"subm{i + 1}",
                                                         conv_type="SubMConv3d"))
                in_channels = out_channels
            self.encoder_layers.add_module(f"encoder_layer{i + 1}", SparseSequential(*blocks))
        return out_channels

    def forward(self, voxel_features, coors, batch_size, static_caps=None):
        """(M,C) voxel features + (M,4) int (b,z,y,x) -> (B, C*D, H, W) BEV map.
        static_caps ({indice_key of a strided conv: rows}): fixed-shape execution for hipGraph replay -- `coors` may hold
        padding rows (b < 0), every level is padded to its capacity, nothing is read back to the host; returns
        (bev, [(indice_key, device count, capacity), ...]) and the caller checks the counts afterwards."""
        coors = coors.int()
        if static_caps is not None:
            x = SparseConvTensor.sorted_by_bitmap(voxel_features, coors, self.sparse_shape, int(batch_size), static_caps)
            static = x.indice_dict["static"]
            return self._layers(x).dense_bev(), static["counts"]
        if self.spatial_sort and coors.is_cuda and coors.shape[0] > 0:
            # rows into (b, y, x, z) order + the level's occupancy bitmap: all rulebooks below are built by bitmap rank
            x = SparseConvTensor.sorted_by_bitmap(voxel_features, coors, self.sparse_shape, int(batch_size))
        else:
            x = SparseConvTensor(voxel_features, coors, self.sparse_shape, int(batch_size))
        return self._layers(x).dense_bev()

    def _layers(self, x):
        """conv_input -> encoder stages -> conv_out.  Static-shape inference on the GPU: the rulebooks of ALL levels depend on the
        coordinates alone, so an index-only dry pass builds them on a second stream (~40 launches of ~5 us: marks, rank scans,
        emits, pair tables, row ranges) while this stream runs the convolutions of the levels already indexed; each module waits
        for the event behind its own rulebooks.  (Eager levels read their sizes back to the host inside the rulebook calls: no
        overlap to gain there.)  Measured: nusc_L 252.3 -> 256.0 frames/s (same box, alternating); on LC, where the BEV half replays beside
        the camera graph, the forked graph LOST 3 % (29.9 -> 29.0), so graphs.GraphedFrame switches it off there
        (`self.index_stream = False`).  SRF_SPARSE_INDEX_STREAM=0 keeps everything on one stream."""
        import os
        mods = [self.conv_input] + list(self.encoder_layers._modules.values()) + [self.conv_out]
        if not (x.features.is_cuda and "static" in x.indice_dict and not torch.is_grad_enabled()
                and getattr(self, "index_stream", True) and os.environ.get("SRF_SPARSE_INDEX_STREAM", "1") != "0"):
            for m in mods:
                x = m(x)
            return x
        cur = torch.cuda.current_stream()
        side = _INDEX_STREAMS.get(x.features.device.index)   # (not a module attribute: modules are deep-copied and pickled)
        if side is None:
            side = _INDEX_STREAMS[x.features.device.index] = torch.cuda.Stream(device=x.features.device)
        side.wait_stream(cur)
        events = []
        with torch.cuda.stream(side):
            xi = SparseConvTensor(None, x.indices, x.spatial_shape, x.batch_size, x.indice_dict, x.num_rows)
            for m in mods:
                xi = m(xi)
                ev = torch.cuda.Event()
                ev.record(side)
                events.append(ev)
        for m, ev in zip(mods, events):
            cur.wait_event(ev)
            x = m(x)
        cur.wait_stream(side)
        return x
