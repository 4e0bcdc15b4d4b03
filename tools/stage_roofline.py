#!/usr/bin/env python3
"""Per-stage roofline table of one frame (achieved HBM GB/s for the scatter/gather stages, MFMA TFLOP/s
for the GEMM stages, each against the MI355X peak).

Every operator of srfdet3d_amd.ops is wrapped with a pair of HIP events on the launch stream (eager execution, no
hipGraph), and its ALGORITHMIC bytes / FLOPs are computed from the shapes it was called with, using the formulas of
SURVEY.md 8d / DESIGN.md section 4.  A stage's figure covers everything the wrapper launches (its kernels, the memsets
of its tables and counters, and for the size-returning ops the device->host read), i.e. what a caller pays.

usage: python tools/stage_roofline.py [--workload nusc_L|nusc_LC|waymo_L|kitti_L] [--frames 10] [--md out.md]
"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (workload table, BN randomisation)
from srfdet3d_amd import ops, synthetic, workloads  # noqa: E402
from srfdet3d_amd.compat.boxes import LiDARInstance3DBoxes  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec (about 6.3 TB/s achievable)
MFMA_PEAK_TF = 157.3      # f32 MFMA

REC = collections.OrderedDict()


def _record(name, bound, ev, byts, flops, note=""):
    r = REC.setdefault(name, dict(bound=bound, ev=[], bytes=0.0, flops=0.0, calls=0, note=note))
    r["ev"].append(ev)
    r["bytes"] += byts
    r["flops"] += flops
    r["calls"] += 1


def _wrap(name, bound, work):
    """work(args, kwargs, result) -> (algorithmic bytes, flops, row name suffix or None)"""
    orig = getattr(ops, name)

    def f(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(*a, **k)
        e1.record()
        b, fl, suffix = work(a, k, out)
        _record(name + (f" [{suffix}]" if suffix else ""), bound, (e0, e1), b, fl)
        return out

    setattr(ops, name, f)


def n4(t):
    return 4 * t.numel()


def install():
    def w_hard(a, k, out):
        pts, max_points = a[0], a[3]
        N, nf = pts.shape
        M = out[1].shape[0]
        return 4 * N * nf + M * (4 * max_points * nf + 12 + 4) + (4 * M * nf if k.get("mean_features") else 0), 0, None
    _wrap("hard_voxelize", "hbm", w_hard)
    _wrap("dynamic_voxelize", "hbm", lambda a, k, o: (n4(a[0]) + 12 * a[0].shape[0], 0, None))

    def w_order(a, k, out):
        idx, shape, batch = a[0], a[1], a[2]
        cells = batch * shape[0] * shape[1] * shape[2]
        return 2 * cells // 8 + 16 * idx.shape[0] + 4 * idx.shape[0], 0, None
    _wrap("spatial_order", "hbm", w_order)
    _wrap("coord_table_build", "hbm", lambda a, k, o: (16 * a[0].shape[0] + 8 * a[0].shape[0], 0, None))

    def w_subm(a, k, out):
        A = a[0].shape[0]
        K = a[2][0] * a[2][1] * a[2][2]
        return 16 * A + 4 * K * A, 0, None
    _wrap("rulebook_subm", "hbm", w_subm)

    def w_strided(a, k, out):
        A_in = a[0].shape[0]
        K = a[3][0] * a[3][1] * a[3][2]
        A_out = out[0].shape[0]
        return 16 * A_in + 4 * K * A_in + 16 * A_out + 4 * K * A_out, 0, None
    _wrap("rulebook_strided", "hbm", w_strided)

    def w_bm(a, k, out):
        lvl = out[0]
        return 2 * 4 * lvl.words * 2 + 16 * a[0].shape[0] * 2 + 4 * a[0].shape[0], 0, None
    _wrap("bitmap_build", "hbm", w_bm)

    def w_subm_bm(a, k, out):
        A = a[0].shape[0]
        K = a[2][0] * a[2][1] * a[2][2]
        return 16 * A + 4 * K * A, 0, None
    _wrap("rulebook_subm_bitmap", "hbm", w_subm_bm)

    def w_strided_bm(a, k, out):
        A_in = a[0].shape[0]
        K = a[2][0] * a[2][1] * a[2][2]
        A_out = out[0].shape[0]
        return 16 * A_in + 2 * 4 * out[3].words * 2 + 16 * A_out + 4 * K * A_out, 0, None
    _wrap("rulebook_strided_bitmap", "hbm", w_strided_bm)

    def w_spconv(a, k, out):
        feats, weight, nbr = a[0], a[1], a[2]
        K, Cin, Cout = weight.shape
        cnt = k.get("pair_counts")
        P = int(cnt.sum().item()) if cnt is not None else int((nbr >= 0).sum().item())
        byts = 4 * (feats.shape[0] * Cin + out.shape[0] * Cout) + 4 * K * Cin * Cout + 4 * K * out.shape[0]
        if k.get("residual") is not None or (len(a) > 5 and a[5] is not None):
            byts += 4 * out.shape[0] * Cout
        return byts, 2.0 * P * Cin * Cout, f"{Cin}->{Cout}, K={K}"
    _wrap("spconv_fwd", "mfma", w_spconv)
    _wrap("spconv_tiles", "hbm", lambda a, k, o: (n4(a[0]) + 4 * a[0].shape[1] * 2, 0, None))
    _wrap("densify", "hbm", lambda a, k, o: (n4(a[0]) + n4(o), 0, None))
    _wrap("box_rois", "latency", lambda a, k, o: (n4(a[0]) * 2, 0, None))

    def w_roi(a, k, out):
        rois = a[1]
        C = a[0][0].shape[1]
        R = rois.shape[0]
        return 4 * R * C * 49 * (1 + 16), 0, None
    _wrap("roi_extract", "hbm", w_roi)

    def w_lin(a, k, out):
        x, w = a[0], a[1]
        M, K = x.shape
        N = w.shape[0]
        return 4 * (M * K + N * K + M * N), 2.0 * M * N * K, f"{K}->{N}"
    _wrap("linear", "mfma", w_lin)

    def w_att(a, k, out):
        P, E3 = a[0].shape
        E = E3 // 3
        return 4 * P * E * 4, 4.0 * P * P * E, None
    _wrap("self_attention", "mfma", w_att)

    def w_dyn(a, k, out):
        R, S, C = a[0].shape
        D = a[1].shape[1] // (2 * C)
        return n4(a[0]) + n4(a[1]) + n4(out), 4.0 * R * S * C * D, None
    _wrap("dynconv_mid", "mfma", w_dyn)

    def w_tail(a, k, out):
        obj, ffn, _, cls_layers, reg_layers, lfc, dfc = a[:7]
        R, C = obj.shape
        F = ffn[0].weight.shape[0]
        fl = 2.0 * R * (2 * C * F + (len(cls_layers) + len(reg_layers)) * C * C + C * (lfc.weight.shape[0] + dfc.weight.shape[0]))
        return 4 * (2 * R * C + 2 * C * F + (len(cls_layers) + len(reg_layers)) * C * C), fl, None
    _wrap("stage_tail", "mfma", w_tail)   # srf_stage_ffn_k + srf_stage_tail_k
    _wrap("apply_deltas", "latency", lambda a, k, o: (3 * n4(a[0]), 0, None))
    _wrap("channel_affine", "hbm", lambda a, k, o: (2 * n4(a[0]) + (n4(a[0]) if k.get("residual") is not None else 0), 0, None))
    _wrap("nms_rotated", "latency", lambda a, k, o: (n4(a[0]), 0, None))

    # channels-last dense layers (csrc/conv.hip, csrc/nhwc.hip): x is an (N, H, W, C) slice
    def w_wino(a, k, out):
        N, H, W, Cin = a[0].shape
        Cout = a[2]
        direct = 2.0 * N * H * W * Cin * Cout * 9
        return 4.0 * N * H * W * (Cin + Cout) + 4.0 * 16 * Cin * Cout, direct / 2.25, f"{Cin}->{Cout} @{H}x{W} (executed FLOPs = direct / 2.25)"
    _wrap("wino3x3", "mfma", w_wino)

    def w_wino43(a, k, o):  # (x, packed, Cout, ...): transform + multiply of one F(4x4, 3x3) layer (executed FLOPs = direct / 4)
        x, Cout = a[0], a[2]
        N, H, W, Cin = x.shape
        v = 2.25 * 4.0 * N * H * W * Cin
        return (4.0 * N * H * W * (Cin + Cout) + 2 * v + 4.0 * 36 * Cin * Cout, 2.0 * 9 * Cin * Cout * N * H * W / 4.0,
                f"{Cin}->{Cout} @{H}x{W} F(4,3): transform + multiply (executed FLOPs = direct / 4)")
    _wrap("wino43", "mfma", w_wino43)

    def w_g1(a, k, out):
        N, H, W, K = a[0].shape
        Cout = a[2]
        return 4.0 * N * H * W * (K + Cout) + 4.0 * K * Cout, 2.0 * N * H * W * K * Cout, f"{K}->{Cout} @{H}x{W}"
    _wrap("conv1x1_nhwc", "mfma", w_g1)

    def w_gs(a, k, out):
        N, H, W, Cin = a[0].shape
        Cout, (kh, kw) = a[2], a[3]
        No, Ho, Wo, _ = out.shape
        return 4.0 * (N * H * W * Cin + No * Ho * Wo * Cout) + 4.0 * kh * kw * Cin * Cout, 2.0 * No * Ho * Wo * Cout * Cin * kh * kw, \
            f"{Cin}->{Cout} k{kh} s{a[4]} @{H}x{W}"
    _wrap("conv_gemm_nhwc", "mfma", w_gs)
    _wrap("stem_conv_nchw", "hbm", lambda a, k, o: (n4(a[0]) + n4(o), 0, None))
    _wrap("nhwc_affine", "hbm", lambda a, k, o: (n4(a[0]) + n4(o) + (n4(k["residual"]) if k.get("residual") is not None else 0), 0, None))
    _wrap("nhwc_maxpool3s2_ceil", "hbm", lambda a, k, o: (n4(a[0]) + n4(o), 0, None))
    _wrap("nhwc_upsample_add", "hbm", lambda a, k, o: (n4(a[0]) + n4(a[1]) + n4(o), 0, None))
    _wrap("nhwc_dwconv3x3s2", "hbm", lambda a, k, o: (n4(a[0]) + n4(o), 0, None))
    _wrap("ese_gate", "latency", lambda a, k, o: (n4(a[0]) * 2, 0, None))


def dense_flops(module, x_shapes):
    """2 * out elements * Cin/groups * kh * kw summed over the Conv2d children, from forward hooks."""
    tot = [0.0]
    hooks = []

    def hook(m, inp, out):
        tot[0] += 2.0 * out.numel() * (m.in_channels // m.groups) * m.kernel_size[0] * m.kernel_size[1]
    for m in module.modules():
        if isinstance(m, torch.nn.Conv2d):
            hooks.append(m.register_forward_hook(hook))
    return tot, hooks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="nusc_L", choices=sorted(bench.WORKLOADS))
    ap.add_argument("--np", type=int, default=200)
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--md", default=None)
    a = ap.parse_args()
    wl = bench.WORKLOADS[a.workload]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = workloads.build(wl["cfg"], a.np).eval()
    bench.randomize_bn(model)
    model = model.to(dev)
    sweep = getattr(synthetic, wl.get("sweep", "nuscenes_sweep"))
    n_points = wl.get("points", 30000)
    frames = [torch.from_numpy(sweep(wl.get("seed", 2000) + i, n_points)).to(dev) for i in range(4)]
    metas = [dict(box_type_3d=LiDARInstance3DBoxes)]
    img = None
    if model.use_img:
        img = torch.from_numpy(synthetic.camera_images(3000)).to(dev)
        metas[0]["lidar2img"] = [m for m in synthetic.camera_rig()]
    with torch.no_grad():
        for i in range(3):
            model.simple_test(img, [frames[i % 4]], metas)
    torch.cuda.synchronize()
    install()
    # whole dense modules: module-level events + direct conv FLOPs from hooks.  Their kernels are the wino3x3 / conv1x1_nhwc /
    # conv_gemm_nhwc / nhwc_* rows above, so these rows are totals and stay out of the sum
    dense = [("SECONDCustom (whole module)", model.pts_backbone), ("BEV FPN (whole module)", model.pts_neck)]
    if model.use_img:
        dense += [("VoVNet-99 (whole module)", model.img_backbone), ("image FPN (whole module)", model.img_neck)]
    for name, mod in dense:
        if mod is None:
            continue
        tot, _ = dense_flops(mod, None)
        orig = mod.forward

        def fwd(*x, _orig=orig, _name=name, _tot=tot, **k):
            f0 = _tot[0]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = _orig(*x, **k)
            e1.record()
            _record(_name, "module", (e0, e1), 0.0, _tot[0] - f0)
            return out
        mod.forward = fwd
    e_all = []
    with torch.no_grad():
        for i in range(a.frames):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            model.simple_test(img, [frames[i % 4]], metas)
            e1.record()
            e_all.append((e0, e1))
    torch.cuda.synchronize()
    frame_ms = sum(s.elapsed_time(e) for s, e in e_all) / a.frames
    lines = [f"Per-stage roofline, `{wl['cfg']}`, np = {a.np}, {n_points} points, eager (no hipGraph), mean of {a.frames} frames; "
             f"frame = {frame_ms:.2f} ms eager.", "",
             "Peaks: HBM 8000 GB/s (spec), f32 MFMA 157.3 TFLOP/s.  `ms/frame` is HIP-event time around the operator wrapper "
             "(kernels + its memsets + size read-backs); algorithmic work per SURVEY.md 8d.  `(whole module)` rows time a dense "
             "module end to end and count its DIRECT convolution FLOPs (a Winograd layer executes 1 / 2.25 of them): they are "
             "totals of the wino3x3 / conv1x1_nhwc / conv_gemm_nhwc / nhwc_* rows and are left out of the sum.", "",
             "| stage (ops.* wrapper) | calls/frame | ms/frame | us/call | algorithmic MB/call | GFLOP/call | achieved | % of peak | bound |",
             "|---|---|---|---|---|---|---|---|---|"]
    tot_ms = 0.0
    for name, r in REC.items():
        ms = sum(s.elapsed_time(e) for s, e in r["ev"])
        calls = r["calls"] / a.frames
        per_call_us = ms / r["calls"] * 1e3
        mb = r["bytes"] / r["calls"] / 1e6
        gf = r["flops"] / r["calls"] / 1e9
        if r["bound"] != "module":
            tot_ms += ms / a.frames
        if r["bound"] in ("mfma", "module") and gf > 0:
            ach = gf / (per_call_us * 1e-6) / 1e3
            cell, pct = f"{ach:.1f} TFLOP/s", 100 * ach / MFMA_PEAK_TF
        elif r["bound"] == "hbm":
            ach = mb / 1e3 / (per_call_us * 1e-6)
            cell, pct = f"{ach:.0f} GB/s", 100 * ach / HBM_PEAK_GBS
        else:
            cell, pct = "--", float("nan")
        lines.append(f"| {name} | {calls:.1f} | {ms / a.frames:.3f} | {per_call_us:.1f} | {mb:.2f} | {gf:.3f} | {cell} | "
                     f"{'--' if pct != pct else f'{pct:.1f}'} | {r['bound']} |")
    lines.append(f"| **sum of the rows** | | **{tot_ms:.2f}** | | | | | | (the rest of the frame is torch glue between the operators) |")
    text = "\n".join(lines)
    print(text)
    if a.md:
        with open(a.md, "w") as fh:
            fh.write(text + "\n")


if __name__ == "__main__":
    main()
