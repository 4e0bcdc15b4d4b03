"""f32-MFMA kernels against the split GEMM on the SMALL GEMMs of a frame (below ops.GEMM_SPLIT_MIN_TILES), shape by shape:
the stride-2 3x3 layers of SECOND / the BEV FPN (implicit im2col, K = 9 Cin) and the 1x1 laterals.  Interleaved rounds, median.
    python tools/small_gemm_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srfdet3d_amd import ops  # noqa: E402


def timed(fn, reps=20):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2]


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    rows = []
    # (name, N, H, W, Cin, Cout, k, stride)
    shapes = [("SECOND 128->256 s2 @184", 1, 184, 184, 128, 256, 3, 2), ("FPN extra 128->128 s2 @46", 1, 46, 46, 128, 128, 3, 2),
              ("FPN extra 128->128 s2 @23", 1, 23, 23, 128, 128, 3, 2), ("waymo SECOND 128->256 s2 @188x188", 1, 188, 188, 128, 256, 3, 2),
              ("lateral 128->128 @184", 1, 184, 184, 128, 128, 1, 1), ("lateral 256->128 @92", 1, 92, 92, 256, 128, 1, 1),
              ("img lateral 512->256 @116x200 x6", 6, 116, 200, 512, 256, 1, 1), ("img lateral 768->256 @58x100 x6", 6, 58, 100, 768, 256, 1, 1),
              ("img lateral 1024->256 @29x50 x6", 6, 29, 50, 1024, 256, 1, 1), ("stage5 concat 2144->1024 @29x50 x6", 6, 29, 50, 2144, 1024, 1, 1)]
    for name, N, H, W, Cin, Cout, k, s in shapes:
        x = torch.randn(N, H, W, Cin, generator=g).to(dev)
        w = torch.randn(Cout, Cin, k, k, generator=g).to(dev) * 0.05
        if k == 3:
            pf, ps = ops.pack_conv_gemm_weights(w), ops.pack_conv_gemm_split_weights(w)
            f32 = lambda: ops.conv_gemm_nhwc(x, pf, Cout, (3, 3), s, 1, relu=True)
            os.environ["SRF_GEMM_SPLIT_MIN"] = "0"
            spl = lambda: ops.conv_gemm_nhwc(x, pf, Cout, (3, 3), s, 1, relu=True, packed_split=ps)
        else:
            w2 = w.reshape(Cout, Cin)
            pf, ps = ops.pack_conv1x1_nhwc_weights(w2), ops.pack_conv1x1_nhwc_split_weights(w2)
            f32 = lambda: ops.conv1x1_nhwc(x, pf, Cout, relu=True)
            spl = lambda: ops.conv1x1_nhwc(x, pf, Cout, relu=True, packed_split=ps)
        os.environ["SRF_GEMM_SPLIT_MIN"] = "0"
        a, b = f32(), spl()
        err = float((a - b).abs().max() / a.abs().max())
        tf, ts = [], []
        for _ in range(3):
            tf.append(timed(f32))
            ts.append(timed(spl))
        Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
        tiles = ((N * Ho * Wo + 127) // 128) * ((Cout + 127) // 128)
        rows.append((name, tiles, min(tf), min(ts), err))
        print(f"{name:40s} tiles {tiles:5d}  f32 {min(tf):7.1f} us   split {min(ts):7.1f} us   rel diff {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
