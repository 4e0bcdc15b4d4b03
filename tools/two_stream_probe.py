"""Do two half-batch chains of Winograd layers on two streams fill each other's idle CUs?  20 stage-4 layers (192 -> 192 @ 58 x 100)
on 6 images in one stream against 2 x 3 images on two streams, and the same for stage 5 (224 -> 224 @ 29 x 50) (developer tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srfdet3d_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x0 = torch.randn(64 << 20, device=dev)
for _ in range(20):
    x0.mul_(1.0)


def chain(x, y, pk, C, layers):
    a, b = x, y
    for _ in range(layers):
        ops.wino3x3(a, pk, C, None, None, True, out=b)
        a, b = b, a


for name, H, W, C in (("stage 4", 58, 100, 192), ("stage 5", 29, 50, 224)):
    w = (torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(dev)
    pk = ops.pack_wino3x3_weights(w)
    x = torch.randn(6, H, W, C, device=dev)
    y = torch.empty_like(x)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    L = 20

    def one():
        chain(x, y, pk, C, L)

    def two():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            chain(x[:3], y[:3], pk, C, L)
        with torch.cuda.stream(s2):
            chain(x[3:], y[3:], pk, C, L)
        cur.wait_stream(s1)
        cur.wait_stream(s2)

    def graphed(fn):
        gr = torch.cuda.CUDAGraph()
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(gr):
            fn()
        return gr.replay

    for label, fn in (("one stream, 6 images", one), ("two streams, 3 + 3 images", two), ("graph of the one-stream form", graphed(one)),
                      ("graph of the two-stream form", graphed(two))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name}: {label:32s} {e0.elapsed_time(e1) / 5 / L * 1e3:8.1f} us per layer", flush=True)


# the OSA pattern of stage 4: per block one 768 -> 192 layer + four 192 -> 192 layers, fork / join around every block
H, W, C = 58, 100, 192
w1 = (torch.randn(C, 768, 3, 3, generator=g) / (3 * 768 ** 0.5)).to(dev)
w2 = (torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(dev)
pk1, pk2 = ops.pack_wino3x3_weights(w1), ops.pack_wino3x3_weights(w2)
buf = torch.randn(6, H, W, 768 + 5 * C, device=dev)
s1 = torch.cuda.Stream()


def block(b):
    src = b[..., :768]
    off = 768
    for i in range(5):
        out = b[..., off:off + C]
        ops.wino3x3(src, pk1 if i == 0 else pk2, C, None, None, True, out=out)
        src, off = out, off + C


def blocks_one():
    for _ in range(9):
        block(buf)


def blocks_two():
    cur = torch.cuda.current_stream()
    for _ in range(9):
        s1.wait_stream(cur)
        block(buf[:3])
        with torch.cuda.stream(s1):
            block(buf[3:])
        cur.wait_stream(s1)


for label, fn in (("one stream", blocks_one), ("fork / join per block", blocks_two), ("graph, one stream", graphed(blocks_one)),
                  ("graph, fork / join per block", graphed(blocks_two))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"stage-4 OSA pattern, 9 blocks: {label:30s} {e0.elapsed_time(e1) / 5 * 1e3:9.1f} us", flush=True)
