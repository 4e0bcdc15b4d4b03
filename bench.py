#!/usr/bin/env python3
"""bench.py -- frames/s of the SRFDet3D hot path on synthetic nuScenes-shaped sweeps (BASELINE.json metric).

A step = one frame through the whole path (voxelize -> VFE -> sparse encoder -> SECOND -> FPN -> 5-stage decoder ->
decode + rotated NMS -> results on the host), inputs already resident in HBM.  One process per GPU; for N > 1 the
driver launches this file under torch.distributed.run and every rank runs K frames of its own (weak scaling, frames
are independent: no data-path collective); the job's time is the max over ranks.

Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline     -- the dominant hand-written kernel of the workload against the 157.3 TFLOP/s f32 MFMA peak, from HIP events
                  on the launch stream around every launch: LC = srf_wino43_mm_k (the multiply kernel of the Winograd
                  F(4x4, 3x3) layers of the camera branch: FLOPs it executes on the MFMA = direct FLOPs / 4, direct-equivalent
                  rate beside it), with roofline.xform = its HBM-bound input-transform kernel against the 8 TB/s peak,
                  roofline.wino23 = the layers left on F(2x2, 3x3) and roofline.gemm = srf_conv1x1_nhwc_k;
                  LiDAR-only = the 128->128 SubM sparse conv (2 * pairs * Cin * Cout per launch); plus
                  roofline.stage = the WHOLE sparse-conv stage (all 21 launches: sum of algorithmic FLOPs and bytes over the sum
                  of their event times, against the MFMA and the HBM peak);
  cpu_baseline -- oracle/pipeline.py (the CPU port of the same path: C/OpenMP operators + torch-CPU dense layers)
                  timed on this host, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide, "Peak BF16/FP16 MFMA": ~2.5 PF dense (the 5 PF headline includes 2:1 sparsity)
HBM_PEAK_GBS = 8000.0         # same guide, "HBM3E peak BW" (spec)
ROUND = "r05"                 # prefix of the profiles/ files this build's counter numbers live in

WORKLOADS = {
    "nusc_L": dict(cfg="srfdet_voxel_nusc_L", desc="srfdet_voxel_nusc_L inference (LiDAR-only), synthetic 30k-pt sweep, "
                   "grid 1472x1472x41, end-to-end incl. NMS"),
    "nusc_LC": dict(cfg="srfdet_voxel_nusc_LC", desc="srfdet_voxel_nusc_LC inference (LiDAR + 6 cameras), synthetic 30k-pt "
                    "sweep + 6x928x1600 images, end-to-end incl. NMS"),
    "waymo_L": dict(cfg="srfdet_dvoxel_waymo_L", desc="srfdet_dvoxel_waymo_L inference, synthetic 180k-pt sweep, dynamic "
                    "voxelization, grid 1536x1536x41, end-to-end incl. NMS", sweep="waymo_sweep", seed=5000, points=180000),
    "kitti_L": dict(cfg="srfdet_voxel_kitti_L", desc="srfdet_voxel_kitti_L inference, synthetic 17k-pt front-view sweep, "
                    "dynamic voxelization, grid 1408x1600x41, end-to-end incl. NMS", sweep="kitti_sweep", seed=1000,
                    points=17000),
}


def randomize_bn(model, seed=0):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="nusc_LC", choices=sorted(WORKLOADS),
                    help="default: the configuration BASELINE.json's metric is quoted on (srfdet_voxel_nusc_LC)")
    ap.add_argument("--np", type=int, default=200, help="num_proposals override (BASELINE.json: np~200)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f32-mfma-line", action="store_true",
                    help="LC only: skip the second timed loop that runs the same frames with SRF_GEMM_SPLIT=0 (f32_mfma_only in the line)")
    ap.add_argument("--eager", action="store_true", help="do not replay the static tail as a hipGraph")
    ap.add_argument("--whole-frame", default="auto", choices=["auto", "on", "off"],
                    help="replay voxelization + sparse encoder + tail as hipGraphs with capacity-padded static shapes (one graph "
                         "for the LiDAR-only workloads; with cameras two, the BEV half and the decoder half, so that the camera "
                         "graph runs beside the first).  auto = on wherever the configuration is eligible.  The per-launch HIP "
                         "events of `roofline` are then taken on eager frames right after the timed region")
    ap.add_argument("--img-overlap", default="auto", choices=["auto", "on", "off"],
                    help="LC only: replay the image-branch graph on a side stream beside the LiDAR half.  auto = on when the "
                         "frame replays as graphs (the kernels of `roofline` are then timed on serial eager frames after the "
                         "timed region, so sharing the chip does not distort them), off with --whole-frame off")
    ap.add_argument("--img-precomputed", action="store_true",
                    help="LC only: the camera features (VoVNet -> FPN) are computed once before the timed region and reused: the "
                         "'image features pre-computed' line of SURVEY.md 8d / BASELINE.md C3 (decoder + LiDAR path with the fusion "
                         "RoI gather); never the headline number")
    ap.add_argument("--img-dtype", default="fp32", choices=["fp32", "fp16", "bf16"],
                    help="LC only: run the image backbone+neck under autocast (the reference's auto_fp16 mode); "
                         "fp32 is the default and the only setting the headline number may use")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # one rank per GPU; the modulo only matters for rehearsing the N > 1 path on a box with fewer GPUs than ranks
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = os.environ.get("SRF_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm; "gloo" only for rehearsals
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from srfdet3d_amd import graphs, nhwc, ops, synthetic, workloads
    nhwc_on = nhwc.enabled()
    from srfdet3d_amd.compat.boxes import LiDARInstance3DBoxes

    wl = WORKLOADS[args.workload]
    torch.manual_seed(0)
    model_cpu = workloads.build(wl["cfg"], args.np).eval()
    randomize_bn(model_cpu)
    import copy
    model = copy.deepcopy(model_cpu).to(dev)
    if not args.eager:
        whole = args.whole_frame != "off"
        args.img_overlap = model.use_img and (args.img_overlap == "on" or (args.img_overlap == "auto" and whole))
        model.enable_hip_graphs(img_overlap=args.img_overlap, whole_frame=whole)
    else:
        args.img_overlap = False
    if args.img_dtype != "fp32":
        model.img_autocast_dtype = dict(fp16=torch.float16, bf16=torch.bfloat16)[args.img_dtype]
        model.img_backbone.to(memory_format=torch.channels_last)

    # a small pool of distinct frames, resident in HBM before the timed region; rank r starts at frame r
    n_pool = 8
    sweep = getattr(synthetic, wl.get("sweep", "nuscenes_sweep"))
    n_points = wl.get("points", 30000)
    frames = [torch.from_numpy(sweep(wl.get("seed", 2000) + i, n_points)).to(dev) for i in range(n_pool)]
    metas = [dict(box_type_3d=LiDARInstance3DBoxes)]
    img = None
    if model.use_img:
        img = torch.from_numpy(synthetic.camera_images(3000)).to(dev)
        metas[0]["lidar2img"] = [m for m in synthetic.camera_rig()]

    if args.img_precomputed and model.use_img:
        # the camera branch runs once, here; the timed frames reuse its features (the fusion gather, img_convs included in
        # the head, and everything on the LiDAR side still run per frame)
        with torch.no_grad():
            cached_feats = [f.clone() for f in model.extract_img_feat(img, metas)]
        model._graphed_img = None
        model.extract_img_feat = lambda *_a, **_k: cached_feats

    def step(i):
        with torch.no_grad():
            return model.simple_test(img, [frames[(rank + i) % n_pool]], metas)

    for i in range(args.warmup):
        step(i)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ops.KERNEL_TIMING = {"spconv": [], "wino": [], "gemm": [], "gsplit": [], "cgemm": [], "w43x": [], "w43m": []}  # HIP-event pairs around every launch of the timed region
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    records = ops.KERNEL_TIMING["spconv"]
    roofline_source = "HIP events on the launch stream around every launch of the timed region"
    gf = getattr(model, "_graphed_frame", None)
    if gf is not None and not records:
        # the timed frames replayed as ONE hipGraph (no per-launch Python to put events in): time the same launches
        # on a few eager frames right after the timed region
        roofline_source = ("HIP events around the launches of 5 eager frames run right after the timed region (the timed frames "
                           "replay as one hipGraph; per-launch times inside it: profiles/)")
        with torch.no_grad():
            for i in range(5):
                if model.use_img:
                    model.extract_bev([frames[(rank + i) % n_pool]])   # (the camera branch's launches are timed below: SECOND's stay out of its sums)
                else:   # + SECOND and the BEV FPN: their GEMM-shaped layers report which arithmetic route they take (config.gemm_route)
                    model.extract_point_features([frames[(rank + i) % n_pool]])
        torch.cuda.synchronize()
        records = ops.KERNEL_TIMING["spconv"]
    dense_source = roofline_source
    if model.use_img and not (ops.KERNEL_TIMING["wino"] or ops.KERNEL_TIMING["w43m"]) and not args.img_precomputed:
        # the camera branch replayed as a hipGraph: time its launches on 3 eager passes right after the timed region
        dense_source = ("HIP events around the launches of 3 eager passes of the camera branch run right after the timed region "
                        "(the timed frames replay it as a hipGraph; per-launch times inside it: profiles/)")
        gi, model._graphed_img = model._graphed_img, None
        with torch.no_grad():
            for _ in range(3):
                feats_ = model.extract_img_feat(img, metas)
                if model.bbox_head.hidden_dim != model.bbox_head.feat_channels_img:
                    model.bbox_head._img_convs_only(feats_)  # the head's 3x3 convolutions on the camera levels: same kernel
        torch.cuda.synchronize()
        model._graphed_img = gi
    wino_rec, gemm_rec, gsplit_rec = ops.KERNEL_TIMING["wino"], ops.KERNEL_TIMING["gemm"], ops.KERNEL_TIMING["gsplit"]
    w43x_rec, w43m_rec = ops.KERNEL_TIMING["w43x"], ops.KERNEL_TIMING["w43m"]
    cgemm_rec = ops.KERNEL_TIMING["cgemm"]
    ops.KERNEL_TIMING = None
    if world > 1:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    if rank == 0:
        def _source_id():
            """sha of the kernel sources: a counter file is only quoted for the build it was measured on."""
            import glob
            import hashlib
            h = hashlib.sha256()
            for f in sorted(glob.glob(os.path.join(ROOT, "srfdet3d_amd", "csrc", "*.h*"))):
                with open(f, "rb") as fh:
                    h.update(fh.read())
            return h.hexdigest()[:16]

        def _traffic(name):
            """HBM bytes per launch from the rocprofv3 --pmc passes of THIS build (tools/measure_traffic.py writes the file with
            the sha of the kernel sources it ran; a file measured on other sources is not quoted: null)."""
            path = os.path.join(ROOT, "profiles", f"{ROUND}_pmc_{name}_traffic.json")
            if not os.path.exists(path):
                return None
            with open(path) as fh:
                d = json.load(fh)
            if d.get("kernel_source_sha16") not in (None, _source_id()) and not os.environ.get("SRF_BENCH_ANY_TRAFFIC"):
                return None
            return d.get("traffic_bytes_per_launch")

        def _in_graph(name):
            """ms per frame of a kernel family inside the timed hipGraphs, QUOTED from the tracked rocprofv3 kernel trace of the same
            command (profiles/<round>_in_graph_summary.json, written by tools/in_graph_summary.py with the sha of the kernel sources it
            ran): not measured in this run, labelled `quoted_from`, and dropped when the sources have changed since (like `traffic`)."""
            path = os.path.join(ROOT, "profiles", f"{ROUND}_in_graph_summary.json")
            if not os.path.exists(path):
                return None
            with open(path) as fh:
                d = json.load(fh)
            if d.get("kernel_source_sha16") != _source_id() and not os.environ.get("SRF_BENCH_ANY_TRAFFIC"):
                return None
            return d.get(args.workload, {}).get(name)

        IN_GRAPH_FILE = f"profiles/{ROUND}_in_graph_summary.json"

        # the dominant sparse-conv shape = the (Cin, Cout, K) group with the most time: the 128 -> 128, 27-offset SubM conv on
        # nuScenes / Waymo (4 launches per frame on the 5x184x184 level), 64 -> 64 on KITTI's narrower encoder
        groups = {}
        for (s, e, cin, cout, K, flops, byts) in records:
            groups.setdefault((cin, cout, K), []).append((s.elapsed_time(e), flops, byts))
        dom_key = max(groups, key=lambda k: sum(d[0] for d in groups[k])) if groups else None
        if (128, 128, 27) in groups:
            dom_key = (128, 128, 27)
        dom = groups.get(dom_key, [])
        per_frame = {}
        for (s, e, cin, cout, K, flops, byts) in records:
            per_frame[(cin, cout, K)] = per_frame.get((cin, cout, K), 0) + 1
        spconv128 = None
        if dom:
            ms = sum(d[0] for d in dom) / len(dom)
            flops = sum(d[1] for d in dom) / len(dom)
            achieved = flops / (ms * 1e-3) / 1e12
            kname = ("srf_spconv_gs_k<4, 128> (SubM 3x3x3, 128->128, last level of the sparse encoder)" if dom_key == (128, 128, 27)
                     else f"sparse conv {dom_key[0]}->{dom_key[1]}, {dom_key[2]} offsets (the shape with the most time in this encoder)")
            spconv128 = dict(kernel=kname, bound="mfma",
                             achieved=round(achieved, 3), peak=F32_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                             frac=round(achieved / F32_MFMA_PEAK_TFLOPS, 4),
                             traffic=_traffic("spconv128") if args.workload in ("nusc_L", "nusc_LC") else None,
                             launches=len(dom), avg_us=round(ms * 1e3, 2), measured=roofline_source,
                             algorithmic_flops_per_launch=int(flops),
                             algorithmic_bytes_per_launch=int(sum(d[2] for d in dom) / len(dom)))
        stage = None
        if records:
            rows = [(s.elapsed_time(e), flops, byts) for (s, e, cin, cout, K, flops, byts) in records]
            # frames covered by the records: conv_input (first layer of the encoder) runs once per frame
            first = next(iter(per_frame))
            nfr = max(1, per_frame[first]) if per_frame else 1
            ms, fl, by = sum(r[0] for r in rows), sum(r[1] for r in rows), sum(r[2] for r in rows)
            stage = dict(name="sparse-conv stage: every srf_spconv_* launch of the encoder (21 per nuScenes frame)", launches=len(rows),
                         ms_per_frame=round(ms / nfr, 4), gflop_per_frame=round(fl / nfr / 1e9, 3), mbytes_per_frame=round(by / nfr / 1e6, 2),
                         tflops=round(fl / (ms * 1e-3) / 1e12, 3), frac_mfma=round(fl / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                         gbs=round(by / (ms * 1e-3) / 1e9, 1), frac_hbm=round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         measured=roofline_source)

            # the two regimes of the stage apart: layers of at most 32 channels move 64-128 B per gathered row for 1-4 kFLOP --
            # HBM / L2 bound, read against the 8 TB/s peak; layers from 64 channels up carry 22-33 FLOP per byte -- MFMA bound, read
            # against the f32 MFMA peak
            def _part(sel, label):
                rr = [(s_.elapsed_time(e_), fl_, by_) for (s_, e_, cin, cout, K, fl_, by_) in records if sel(cin, cout)]
                if not rr:
                    return None
                ms_, fl_, by_ = sum(r[0] for r in rr), sum(r[1] for r in rr), sum(r[2] for r in rr)
                return dict(layers=label, launches_per_frame=round(len(rr) / nfr, 1), ms_per_frame=round(ms_ / nfr, 4),
                            gflop_per_frame=round(fl_ / nfr / 1e9, 3), mbytes_per_frame=round(by_ / nfr / 1e6, 2),
                            tflops=round(fl_ / (ms_ * 1e-3) / 1e12, 3), frac_mfma=round(fl_ / (ms_ * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                            gbs=round(by_ / (ms_ * 1e-3) / 1e9, 1), frac_hbm=round(by_ / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
            stage["c_le_32"] = _part(lambda cin, cout: max(cin, cout) <= 32, "Cin, Cout <= 32 (bound: HBM; frac_hbm is the figure to read)")
            stage["c_ge_64"] = _part(lambda cin, cout: max(cin, cout) >= 64, "Cout >= 64 (bound: f32 MFMA; frac_mfma is the figure to read)")

        def _dense(recs, kernel, name, passes=1, peak=F32_MFMA_PEAK_TFLOPS):
            if not recs:
                return None
            t_ms = sum(r[0].elapsed_time(r[1]) for r in recs)
            direct, executed, byts = sum(r[3] for r in recs), sum(r[4] for r in recs), sum(r[5] for r in recs)
            worst = max(recs, key=lambda r: r[0].elapsed_time(r[1]))
            ach = executed / (t_ms * 1e-3) / 1e12
            d = dict(kernel=kernel, bound="mfma", achieved=round(ach, 3), peak=peak, unit="TFLOP/s",
                     frac=round(ach / peak, 4), traffic=_traffic(name), launches=len(recs),
                     avg_us=round(t_ms * 1e3 / len(recs), 2), serial_ms_per_frame=round(t_ms / passes, 3), measured=dense_source,
                     algorithmic_flops_per_launch=int(executed / len(recs)),
                     direct_equivalent_tflops=round(direct / (t_ms * 1e-3) / 1e12, 3),
                     algorithmic_bytes_per_launch=int(byts / len(recs)),
                     longest_launch=dict(layer=worst[2], us=round(worst[0].elapsed_time(worst[1]) * 1e3, 1)),
                     note="aggregate over all launches of the kernel in a frame: sum of FLOPs over sum of event times; traffic and "
                          "algorithmic bytes are averages over the same launches")
            if d["traffic"] is not None:
                d["traffic_quoted_from"] = f"profiles/{ROUND}_pmc_{name}_traffic.json"
            ig = _in_graph(name)
            if ig is not None:
                d["in_graph_ms_per_frame"] = ig
                d["in_graph_frac"] = round(executed / passes / (ig * 1e-3) / 1e12 / peak, 4)
                d["in_graph_quoted_from"] = IN_GRAPH_FILE
            return d

        def _stream(recs, kernel, name, passes=1):
            """an HBM-bound kernel: algorithmic bytes over event time against the 8 TB/s peak"""
            if not recs:
                return None
            t_ms = sum(r[0].elapsed_time(r[1]) for r in recs)
            byts = sum(r[5] for r in recs)
            ach = byts / (t_ms * 1e-3) / 1e9
            d = dict(kernel=kernel, bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4),
                     traffic=_traffic(name), launches=len(recs), avg_us=round(t_ms * 1e3 / len(recs), 2),
                     serial_ms_per_frame=round(t_ms / passes, 3), measured=dense_source, algorithmic_bytes_per_launch=int(byts / len(recs)))
            if d["traffic"] is not None:
                d["traffic_quoted_from"] = f"profiles/{ROUND}_pmc_{name}_traffic.json"
            ig = _in_graph(name)
            if ig is not None:
                d["in_graph_ms_per_frame"] = ig
                d["in_graph_quoted_from"] = IN_GRAPH_FILE
            return d

        cam_passes = 3 if dense_source != roofline_source else max(1, args.steps)
        w43m = _dense(w43m_rec, "srf_wino43_mm_k (Winograd F(4x4,3x3) multiply + output transform: every 3x3 / stride 1 convolution of "
                                "VoVNet-99 from 96 input channels up, the image FPN and img_convs; `achieved` counts the FLOPs it executes "
                                "on the MFMA = direct FLOPs / 4)", "wino43mm", cam_passes)
        xform = _stream(w43x_rec, "srf_wino43_xform_k (input transform V = B^T d B of the same layers: reads the input once, writes 2.25x)",
                        "wino43xf", cam_passes)
        wino = _dense(wino_rec, "srf_wino3x3_k (Winograd F(2x2,3x3): the 3x3 layers below 96 input channels -- VoVNet stem_2; `achieved` "
                                "counts the FLOPs it executes on the MFMA = direct FLOPs / 2.25)", "wino3x3", cam_passes)
        gemm = _dense(gemm_rec, "srf_conv1x1_nhwc_k / srf_gemm_direct_k (the OSA concat 1x1 convolutions and the FPN laterals as one GEMM "
                                "each, on the f32 MFMA)", "conv1x1", cam_passes)
        gsplit = _dense(gsplit_rec, "srf_gemm_split_k (the same 1x1 convolutions as f32 GEMMs on the bf16 MFMA: exact three-way bf16 split of "
                                    "both operands, six of the nine partial products accumulated in f32; `achieved` = the bf16 FLOPs it issues = "
                                    "6 x the f32 GEMM's, against the dense bf16 peak; direct_equivalent_tflops = the f32 GEMM's own rate)",
                        "gemmsplit", cam_passes, peak=BF16_MFMA_PEAK_TFLOPS)
        if gsplit is not None and gemm is None:
            gemm = gsplit
        elif gsplit is not None:
            gemm["split"] = gsplit
        if model.use_img and (w43m is not None or wino is not None):   # LC: the camera branch dominates the frame
            roofline = w43m if w43m is not None else wino
            if w43m is not None:
                roofline["xform"] = xform
                roofline["wino23"] = wino
            roofline["gemm"] = gemm
            roofline["spconv128"] = spconv128
        else:
            roofline = spconv128
        if roofline is not None:
            roofline["stage"] = stage
        # which GEMM-shaped layers of THIS workload run on which arithmetic (the LiDAR-only lines route SECOND's
        # stride-2 layer and the finest BEV FPN lateral to the split kernel too): the labels of the launches seen by the event records
        def _labels(recs, want_split):
            seen = []
            for r in recs:
                if r[2].endswith(" split") == want_split and r[2] not in seen:
                    seen.append(r[2])
            return seen
        gemm_route = dict(rule=f"a 1x1 / strided-3x3 layer runs on srf_gemm_split_k (f32 GEMM on the bf16 MFMA, exact three-way split) when its "
                               f"launch has >= {os.environ.get('SRF_GEMM_SPLIT_MIN', ops.GEMM_SPLIT_MIN_TILES)} tiles of 128 x 128 and SRF_GEMM_SPLIT != 0, "
                               "else on the f32 MFMA (k-ordered fma chain)",
                          split_kernel=[l[:-6] for l in _labels(gsplit_rec + cgemm_rec, True)],
                          f32_mfma=_labels(gemm_rec + cgemm_rec, False),
                          source=dense_source)
        # the same frames with the 1x1 convolutions on the f32-MFMA kernels (SRF_GEMM_SPLIT=0): a second model instance with its own
        # graphs, timed like the headline right after it -- so that the line carries both arithmetic routes measured in one run
        f32_mfma_only = None
        split_on = ops.gemm_split_enabled() and nhwc_on
        if (world == 1 and model.use_img and split_on and not args.img_precomputed and not args.eager and args.img_dtype == "fp32"
                and not args.no_f32_mfma_line):
            os.environ["SRF_GEMM_SPLIT"] = "0"
            try:
                m2 = copy.deepcopy(model_cpu).to(dev)
                m2.enable_hip_graphs(img_overlap=args.img_overlap, whole_frame=args.whole_frame != "off")
                with torch.no_grad():
                    for i in range(max(4, args.warmup)):
                        m2.simple_test(img, [frames[i % n_pool]], metas)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for i in range(args.steps):
                        m2.simple_test(img, [frames[(args.warmup + i) % n_pool]], metas)
                    torch.cuda.synchronize()
                    el2 = time.perf_counter() - t1
                f32_mfma_only = dict(value=round(args.steps / el2, 3), unit="frames/s", ms_per_step=round(el2 / args.steps * 1e3, 3),
                                     steps=args.steps, note="same frames, same process, SRF_GEMM_SPLIT=0: every GEMM of the frame on "
                                     "v_mfma_f32_32x32x2_f32 (one k-ordered fma chain per output)")
                m2._graphed_frame = m2._graphed_img = m2._graphed_tail = None
                del m2
                torch.cuda.empty_cache()
            finally:
                os.environ.pop("SRF_GEMM_SPLIT", None)
        cpu_baseline = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import pipeline
            # the box gives one GPU's share of the host (16 cores); more threads than that only oversubscribes
            cores = min(16, len(os.sched_getaffinity(0)))
            torch.set_num_threads(cores)
            O_lib = __import__("oracle.oracle", fromlist=["lib"]).lib()
            import ctypes
            ctypes.CDLL("libgomp.so.1").omp_set_num_threads(cores)
            pts = frames[0].cpu().numpy()
            img_cpu = img.cpu() if img is not None else None
            # warm-up, then the median of several frames.  One warm-up frame (page-in, torch-CPU / OpenMP thread pools),
            # then 3 timed frames (LC: ~15 s each), bounded to about a minute
            pipeline.forward_to_decode(model_cpu, [pts], metas, img_cpu)
            times = []
            t_all = time.perf_counter()
            while len(times) < 3 and (not times or time.perf_counter() - t_all < 60.0):
                tc = time.perf_counter()
                pipeline.forward_to_decode(model_cpu, [pts], metas, img_cpu)
                times.append(time.perf_counter() - tc)
            med = sorted(times)[len(times) // 2]
            cpu_baseline = dict(value=round(1.0 / med, 4), unit="frames/s", cores=cores, kind="port",
                                sample=f"median of {len(times)} frame(s) after 1 warm-up frame of the same workload through "
                                       f"oracle/pipeline.py (C/OpenMP operators + torch-CPU dense layers); frames took "
                                       + ", ".join(f"{t:.2f}" for t in times) + " s")
        total_frames = args.steps * world
        out = dict(metric=f"frames/sec, {wl['cfg']} synthetic {n_points // 1000}k-pt sweeps", value=round(total_frames / elapsed, 3),
                   unit="frames/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(elapsed / args.steps * 1e3, 3), higher_is_better=True, scaling="weak",
                   vs_baseline=None, dtype="f32" if args.img_dtype == "fp32" else f"f32 (image branch {args.img_dtype})",
                   data="synthetic",
                   config=dict(workload=wl["desc"], num_proposals=args.np, points_per_frame=n_points,
                               frames_per_rank=args.steps, hip_graph_tail=not args.eager,
                               whole_frame_graph=bool(getattr(model, "_graphed_frame", None) is not None), img_branch_overlap=bool(args.img_overlap and model.use_img),
                               img_features_precomputed=bool(args.img_precomputed and model.use_img),
                               img_branch=("channels-last on srf_wino43 / srf_wino3x3 / srf_conv1x1_nhwc" if nhwc.wino43_enabled() else "channels-last on srf_wino3x3 / srf_conv1x1_nhwc") if (model.use_img and nhwc_on) else ("MIOpen" if model.use_img else None),
                               gemm_1x1=(("f32 GEMM on the bf16 MFMA (srf_gemm_split_k): every f32 operand split EXACTLY into three bf16 values, six of "
                                          "the nine exact partial products accumulated in f32, dropped terms < 2^-23 of a product = one f32 "
                                          "rounding; error vs float64 equal to the f32 fma chain's (tests/test_gpu_gemm_split.py); exact for "
                                          "finite operands with 2^-110 <= |x| <= 3.3895e38 (the bf16 maximum) and zero: smaller ones add an "
                                          "absolute error <= 2^-126 each, larger activations / inf give NaN rows (weights of that size keep the "
                                          "layer on the f32 MFMA); SRF_GEMM_SPLIT=0 -> f32_mfma_only") if split_on else "f32 MFMA (k-ordered fma chain)")
                               if (model.use_img and nhwc_on) else None,
                               gemm_route=gemm_route,
                               graph_validation_failures=len(graphs.VALIDATION_LOG),
                               weights="seeded random init, randomised BN statistics",
                               parallelism=f"replica per GPU x{world}, frames sharded, no data-path collective"),
                   roofline=roofline, cpu_baseline=cpu_baseline, f32_mfma_only=f32_mfma_only)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
