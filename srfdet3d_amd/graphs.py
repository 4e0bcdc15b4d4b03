"""HIP-graph replay of the static-shape tail of the path.

Everything after `SparseConvTensor.dense()` has shapes fixed by the config (BEV map -> SECOND -> FPN -> 5 decoder
stages -> decode), and at 30k points it is launch-bound: ~500 small launches per frame.  `GraphedTail` captures that
tail once per input shape into a hipGraph (through torch.cuda.CUDAGraph, which is hipGraph on ROCm) and replays it;
the data-dependent head of the path (voxelization, rulebooks, sparse convs) and the NMS stay eager.

Capture is legal because nothing in the tail allocates through hipMalloc, synchronises or reads back to the host:
the C-ABI entry points only enqueue on the current stream, and workspaces come from torch's graph-private pool.
"""
import torch


class GraphedTail:
    def __init__(self, model, warmup=3):
        self.model = model
        self.warmup = warmup
        self.entries = {}

    def _run(self, bev, img_feats, img_metas):
        m = self.model
        x = m.pts_backbone(bev)
        if m.pts_neck is not None:
            x = m.pts_neck(x)
        logits, boxes = m.bbox_head(img_feats, x, img_metas)
        return m.bbox_head.decode(logits, boxes)

    def __call__(self, bev, img_feats, img_metas, img_static=False):
        """img_static: img_feats are the persistent output buffers of a GraphedImageBranch -- the tail is captured
        reading them in place, so nothing is copied for them."""
        key = (tuple(bev.shape), None if img_feats is None else tuple(tuple(f.shape) for f in img_feats),
               None if not img_static else tuple(f.data_ptr() for f in img_feats))
        e = self.entries.get(key)
        if e is None:
            e = self._capture(key, bev, img_feats, img_metas, img_static)
        e["bev"].copy_(bev)
        if img_feats is not None and not img_static:
            for dst, src in zip(e["img"], img_feats):
                dst.copy_(src)
        e["graph"].replay()
        return e["scores"], e["boxes"]

    def _capture(self, key, bev, img_feats, img_metas, img_static=False):
        static_bev = bev.clone()
        static_img = None
        if img_feats is not None:
            static_img = list(img_feats) if img_static else [f.clone() for f in img_feats]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.warmup):  # MIOpen / rocBLAS pick their kernels here, outside the capture
                self._run(static_bev, static_img, img_metas)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph):
            scores, boxes = self._run(static_bev, static_img, img_metas)
        e = dict(graph=graph, bev=static_bev, img=static_img, scores=scores, boxes=boxes)
        self.entries[key] = e
        return e


class GraphedImageBranch:
    """The image branch of the LC configs (VoVNet -> FPN, srfdet.py:175-202) as a hipGraph replayed on a SIDE stream.

    It depends only on the camera images, so it is enqueued first and runs beside the eager, data-dependent LiDAR half
    (voxelization, rulebooks, sparse convs), whose many short launches and host read-backs leave most of the chip
    idle.  `__call__` returns the persistent feature buffers and an event the consumer stream must wait for."""

    def __init__(self, model, warmup=2, overlap=True):
        self.model = model
        self.warmup = warmup
        self.entries = {}
        self.overlap = overlap
        self.stream = torch.cuda.Stream()

    def __call__(self, img, img_metas):
        key = (tuple(img.shape), img.dtype)
        e = self.entries.get(key)
        main = torch.cuda.current_stream()
        if e is None:
            e = self._capture(key, img, img_metas)
        for meta in img_metas:
            meta.update(input_shape=img.shape[-2:])
        run_on = self.stream if self.overlap else main
        run_on.wait_stream(main)  # img is ready, and the previous frame's consumers of the buffers are done
        with torch.cuda.stream(run_on):
            e["img"].copy_(img)
            e["graph"].replay()
            e["done"].record(run_on)
        return e["feats"], e["done"]

    def _capture(self, key, img, img_metas):
        m = self.model
        static_img = img.clone()
        side = self.stream
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.warmup):
                m.extract_img_feat(static_img, img_metas)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph, stream=side):
            feats = m.extract_img_feat(static_img, img_metas)
        torch.cuda.synchronize()
        e = dict(graph=graph, img=static_img, feats=feats, done=torch.cuda.Event())
        self.entries[key] = e
        return e
