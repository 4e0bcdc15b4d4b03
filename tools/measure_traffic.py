#!/usr/bin/env python3
"""HBM traffic per launch of the kernels bench.py's `roofline` names, from rocprofv3 PMC counters, collected as
/opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE, WRITE_SIZE and the SQ counters in SEPARATE
`--pmc` passes (each with --kernel-trace only), read bytes = 2 x FETCH_SIZE x 1024 on gfx950 (128-B requests are tallied
as 64 B), WRITE_SIZE x 1024 exact.  Run on the GPU box from the repo root:

    python tools/measure_traffic.py            # writes gpurun_out/traffic/r05_pmc_<name>_traffic.json

and copy the files into profiles/ (bench.py reads profiles/r05_pmc_<name>_traffic.json -> roofline.traffic).
This process never touches the GPU itself: every pass is `rocprofv3 ... -- python3 <tool>` started as a child.
"""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "traffic")
SQ = ("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS "
      "SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY").split()

# name -> (program after `--`, kernel-name substring, description, algorithmic bytes note)
ROUND = "r05"
TARGETS = {
    "wino43mm": (["tools/prof_img_branch.py", "3"], ("srf_wino43_mm_k<",),
                 "every srf_wino43_mm_k launch of 3 eager passes of the LC camera branch (88 per frame: VoVNet-99 from stage 2 on, image FPN, img_convs)"),
    "wino43xf": (["tools/prof_img_branch.py", "3"], ("srf_wino43_xform_k",),
                 "every srf_wino43_xform_k launch of the same passes (one per F(4x4,3x3) layer)"),
    "wino3x3": (["tools/prof_img_branch.py", "3"], ("srf_wino3x3_k<", "srf_wino3x3_mixed_k<"),
                "every srf_wino3x3_k launch of the same passes (the layers left on F(2x2,3x3): VoVNet stem_2)"),
    "conv1x1": (["tools/prof_img_branch.py", "3"], ("srf_conv1x1_nhwc_k<1, 1, 4, false>", "srf_conv1x1_nhwc_k<2, 2, 3, false>",
                                                     "srf_conv1x1_nhwc_k<4, 4, 1, false>", "srf_conv1x1_nhwc_mixed_k", "srf_gemm_direct_k<"),
                "EVERY srf_conv1x1_nhwc launch of the same passes, whatever its tile form (20 per frame: OSA concat convolutions + FPN "
                "laterals) -- the launch set bench.py's roofline.gemm aggregates"),
    "gemmsplit": (["tools/prof_img_branch.py", "3"], ("srf_gemm_split_k<",),
                  "every srf_gemm_split_k launch of the same passes (21 per frame: the OSA concat convolutions, the FPN laterals and stem_3 as "
                  "f32 GEMMs on the bf16 MFMA) -- the launch set bench.py's roofline.gemm aggregates"),
    # the scatter / gather stages north_star asks HBM GB/s for: 3 eager LiDAR frames (30k points, np = 200), traffic PER FRAME
    # (the single-workgroup scans of the coarse levels, the middle pass of the three-launch scans and the region fills --
    # table / bitmap clears -- belong to these families too; srf_fill_regions_k is shared by both and is counted with the rulebooks,
    # whose bitmap clears are its large launches)
    "voxelize": (["tools/prof_lidar_frame.py", "3"], ("srf_hv_insert_k", "srf_hv_gather_k", "srf_scan_reduce_k<HvFlag", "srf_scan_apply_k<HvFlag",
                                                       "srf_scan_single_k<HvFlag"),
                 "hard voxelization + per-voxel mean (K1 + a3): insert, first-seen numbering scan, gather; per frame", 3),
    "rulebook": (["tools/prof_lidar_frame.py", "3"], ("srf_bm_", "srf_scan_reduce_k<BmPop", "srf_scan_apply_k<BmPop", "srf_scan_single_k<BmPop",
                                                       "srf_scan_partials_k", "srf_fill_regions_k"),
                 "bitmap-rank rulebooks of the whole encoder (K4): clears, mark / rank scans / place / subm / strided mark, emit, pairs; per frame "
                 "(srf_scan_partials_k and srf_fill_regions_k of the voxelization included here: ~3 small launches)", 3),
    # the same stages on the 180k-point Waymo frame (C5, configs/waymo/srfdet_dvoxel_waymo_L.py:6-35: dynamic voxelization + DynamicScatter)
    "voxelize_waymo": (["tools/prof_lidar_frame.py", "3", "waymo_L"], ("srf_dynamic_voxelize_k",),
                       "dynamic voxelization (K2) of the 180k-point Waymo frame; per frame", 3),
    "scatter_waymo": (["tools/prof_lidar_frame.py", "3", "waymo_L"], ("srf_vu_", "srf_scatter_reduce_k"),
                      "DynamicScatter (K3): occupancy bitmap, rank, per-voxel point lists, mean / max reductions of DynamicVFECustom on the Waymo frame; per frame", 3),
    "rulebook_waymo": (["tools/prof_lidar_frame.py", "3", "waymo_L"], ("srf_bm_", "srf_scan_reduce_k<BmPop", "srf_scan_apply_k<BmPop", "srf_scan_single_k<BmPop",
                                                                        "srf_scan_partials_k", "srf_fill_regions_k"),
                       "bitmap-rank rulebooks of the whole encoder (K4) on the Waymo frame (input level 1536 x 1536 x 41); per frame", 3),
    "densify_waymo": (["tools/prof_lidar_frame.py", "3", "waymo_L"], ("srf_densify_bev_k", "srf_densify_k"),
                      "dense BEV map (K6) of the Waymo frame; per frame", 3),
    "roi_waymo": (["tools/prof_lidar_frame.py", "3", "waymo_L"], ("srf_roi_extract_k",),
                  "multi-level RoIAlign gather (K7) on the Waymo frame; per frame", 3),
    "densify": (["tools/prof_lidar_frame.py", "3"], ("srf_densify_bev_k",),
                "dense BEV map (K6: dense() + the (N, C D, H, W) view) written channels-last in one pass from the last level's bitmap, zero cells "
                "included (srf_densify_bev); per frame", 3),
    "roi": (["tools/prof_lidar_frame.py", "3"], ("srf_roi_extract_k",),
            "multi-level RoIAlign gather (K7), 5 stages x 200 RoIs on the channels-last BEV pyramid; per frame", 3),
    "spconv128": (["tools/bench_spconv.py", "--levels", "4", "--reps", "8"], ("srf_spconv_gs_k<4, 128>", "srf_spconv_gsp_k<4, 128"),
                  "SubM 128->128 on the 5x184x184 level of frame 2000 (A=34992), BN + residual + ReLU epilogue"),
    "spconv64": (["tools/bench_spconv.py", "--levels", "3", "--reps", "8"], ("srf_spconv_gs_k<2, 64>", "srf_spconv_gsp_k<2, 64"),
                 "SubM 64->64 on the 11x368x368 level of frame 2000, BN + residual + ReLU epilogue"),
}


def run_pass(tag, counters, prog):
    d = os.path.join(OUT, tag)
    cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", "python3", *prog]
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout[-3000:])
        raise SystemExit(f"pass {tag} failed")
    return d, " ".join(cmd[:-len(prog) - 1]).replace(ROOT + "/", "") + " python3 " + " ".join(prog)


def _match(sub, name):
    return any(x in name for x in sub) if isinstance(sub, tuple) else sub in name


def counters_of(d, sub):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if _match(sub, r["Kernel_Name"]):
                acc[r["Counter_Name"]].append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
    return acc


def durations_of(d, sub):
    ts = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        ts += [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f)) if _match(sub, r["Kernel_Name"])]
    return ts


def source_id():
    """sha of the kernel sources (bench.py quotes a counter file only for the build it was measured on)"""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "srfdet3d_amd", "csrc", "*.h*"))):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def main():
    want = sys.argv[1:] or list(TARGETS)
    os.makedirs(OUT, exist_ok=True)
    cache = {}
    for name in want:
        prog, sub, desc = TARGETS[name][:3]
        per_frames = TARGETS[name][3] if len(TARGETS[name]) > 3 else 0
        key = " ".join(prog)
        if key not in cache:
            tag = prog[0].split("/")[-1].replace(".py", "") + "_" + "_".join(p.strip("-") for p in prog[1:])
            cache[key] = [run_pass(tag + "_fetch", ["FETCH_SIZE"], prog), run_pass(tag + "_write", ["WRITE_SIZE"], prog),
                          run_pass(tag + "_sq", SQ, prog)]
            print("collected", key, flush=True)
        (df, cf), (dw, cw), (ds, cs) = cache[key]
        fetch = [v for _, v in sorted(counters_of(df, sub)["FETCH_SIZE"])]
        write = [v for _, v in sorted(counters_of(dw, sub)["WRITE_SIZE"])]
        sq = {k: [v for _, v in sorted(vs)] for k, vs in counters_of(ds, sub).items()}
        if not fetch or not write:
            print(f"{name}: no dispatch of {sub!r} found", flush=True)
            continue
        # the first pass of a multi-pass target is cold (weight packing, first-touch): drop the first third when there are >= 3 passes
        skip = len(fetch) // 3 if len(fetch) >= 3 else 0
        n = len(fetch) - skip
        f_kb, w_kb = sum(fetch[skip:]) / n, sum(write[skip:]) / n
        dur = durations_of(ds, sub)
        out = {
            "kernel": " | ".join(sub) if isinstance(sub, tuple) else sub, "workload": desc, "launches_averaged": n,
            "kernel_source_sha16": source_id(),
            "FETCH_SIZE_KB": f_kb, "WRITE_SIZE_KB": w_kb,
            "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B -> read bytes = 2 * FETCH_SIZE * 1024 "
                          "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
            "read_bytes_per_launch": int(2 * f_kb * 1024), "write_bytes_per_launch": int(w_kb * 1024),
            "traffic_bytes_per_launch": int(2 * f_kb * 1024 + w_kb * 1024),
            "sq_counters": {k: sum(v[skip:]) / max(1, len(v) - skip) for k, v in sq.items()},
            "avg_duration_us_under_pmc": (sum(dur[skip:]) / max(1, len(dur) - skip)) / 1e3 if dur else None,
            "commands": [cf, cw, cs],
        }
        s = out["sq_counters"]
        if s.get("SQ_VALU_MFMA_BUSY_CYCLES") and s.get("SQ_BUSY_CYCLES"):
            # SQ_BUSY_CYCLES is summed over the 32 shader engines, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs
            out["derived"] = {"mfma_busy_fraction_per_simd": round(s["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (s["SQ_BUSY_CYCLES"] / 32), 4)}
        if per_frames:
            # a family of different kernels: totals per frame instead of averages per launch (no cold pass dropped: nothing is packed here)
            dur_all = durations_of(ds, sub)
            rd, wr = 2 * sum(fetch) * 1024 / per_frames, sum(write) * 1024 / per_frames
            ms = sum(dur_all) / 1e6 / per_frames
            out.update(launches_averaged=len(fetch), frames=per_frames, launches_per_frame=len(fetch) / per_frames,
                       read_bytes_per_frame=int(rd), write_bytes_per_frame=int(wr), traffic_bytes_per_frame=int(rd + wr),
                       kernel_ms_per_frame_under_pmc=round(ms, 4), hbm_gbs_of_counted_traffic=round((rd + wr) / (ms * 1e-3) / 1e9, 1) if ms else None,
                       frac_of_8000_gbs_peak=round((rd + wr) / (ms * 1e-3) / 1e9 / 8000.0, 4) if ms else None)
            for k in ("FETCH_SIZE_KB", "WRITE_SIZE_KB", "read_bytes_per_launch", "write_bytes_per_launch", "traffic_bytes_per_launch", "avg_duration_us_under_pmc"):
                out.pop(k, None)
        path = os.path.join(OUT, f"{ROUND}_pmc_{name}_traffic.json")
        with open(path, "w") as fh:
            json.dump(out, fh, indent=1)
        print(name, json.dumps({k: out[k] for k in ("launches_averaged", "traffic_bytes_per_launch", "avg_duration_us_under_pmc",
                                                     "traffic_bytes_per_frame", "kernel_ms_per_frame_under_pmc", "hbm_gbs_of_counted_traffic") if k in out}), flush=True)


if __name__ == "__main__":
    main()
