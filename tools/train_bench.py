#!/usr/bin/env python3
"""Config 4 of BASELINE.json: srfdet_voxel_nusc_LC training, bs frames per GPU, DDP over RCCL (one process per GPU).

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/train_bench.py --iters 20

Synthetic frames + random ground truth; the LiDAR branch is frozen as tools/train.py:221-276
does, so gradients (and the all-reduce) cover VoVNet stages 3-5, the image FPN and the head.  Prints iterations/s and
the share of a step spent in backward+all-reduce on rank 0."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srfdet3d_amd import synthetic, workloads  # noqa: E402
from srfdet3d_amd.compat.boxes import LiDARInstance3DBoxes  # noqa: E402
from srfdet3d_amd.plugin import training  # noqa: E402


def random_gt(dev, n, rng):
    xy = rng.uniform(-45, 45, (n, 2))
    z = rng.uniform(-2.5, -0.5, (n, 1))
    size = rng.uniform([1.5, 3.5, 1.4], [2.2, 5.0, 2.0], (n, 3))
    yaw = rng.uniform(-np.pi, np.pi, (n, 1))
    vel = rng.normal(0, 1, (n, 2))
    t = torch.tensor(np.concatenate([xy, z, size, yaw, vel], 1), dtype=torch.float32, device=dev)
    return LiDARInstance3DBoxes(t, box_dim=9), torch.from_numpy(rng.integers(0, 10, n)).to(dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bs", type=int, default=2)
    ap.add_argument("--np", type=int, default=900)
    ap.add_argument("--img-hw", default="928x1600")
    ap.add_argument("--kernel-table", default=None, metavar="FILE",
                    help="after the timed iterations, trace 2 more with torch.profiler and write the per-kernel table of the steady "
                         "state to FILE (markdown); rank 0 only")
    a = ap.parse_args()
    rank, local, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    h, w = (int(v) for v in a.img_hw.split("x"))
    torch.manual_seed(0)
    model = workloads.build("srfdet_voxel_nusc_LC", a.np, train=True)
    training.freeze_lidar_components(model)
    model = model.to(dev).train()
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local], find_unused_parameters=True) if world > 1 else model
    # gradient all-reduce time (collective C1): a DDP communication hook that brackets every bucket's all-reduce with
    # events on the stream it runs on; the buckets overlap with the rest of backward, so this is time ON the communication stream, not
    # time added to the step
    comm_events = []
    if world > 1:
        def timed_allreduce(state, bucket):
            t = bucket.buffer()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fut = dist.all_reduce(t.div_(world), async_op=True).get_future()

            def done(f):
                e1.record()
                comm_events.append((e0, e1, t.numel() * 4))
                return f.value()[0]
            return fut.then(done)
        net.register_comm_hook(None, timed_allreduce)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=2e-4, weight_decay=0.01)
    rng = np.random.default_rng(rank)
    scale = w / 1600.0
    rig = synthetic.camera_rig(f=1266.0 * scale, cx=816.0 * scale, cy=491.0 * h / 928.0)
    pts = [torch.from_numpy(synthetic.nuscenes_sweep(2000 + rank * a.bs + i)).to(dev) for i in range(a.bs)]
    img = torch.cat([torch.from_numpy(synthetic.camera_images(3000 + rank * a.bs + i, h=h, w=w)) for i in range(a.bs)], 0).to(dev)
    gts = [random_gt(dev, 20, rng) for _ in range(a.bs)]
    metas = [dict(box_type_3d=LiDARInstance3DBoxes, lidar2img=[m for m in rig]) for _ in range(a.bs)]

    phases = []   # (forward, backward, optimiser) event triples of the timed iterations

    def step(record=False):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if record else None
        if record:
            ev[0].record()
        losses = net(return_loss=True, img=img, points=pts, img_metas=metas, gt_bboxes_3d=[g[0] for g in gts],
                     gt_labels_3d=[g[1] for g in gts])
        total = sum(losses.values())
        opt.zero_grad(set_to_none=True)
        if record:
            ev[1].record()
        total.backward()           # under DDP the bucketed gradient all-reduce runs inside (overlapped with) this call
        if record:
            ev[2].record()
        torch.nn.utils.clip_grad_norm_(params, 35.0)
        opt.step()
        if record:
            ev[3].record()
            phases.append(ev)
        return total.item()

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if os.environ.get("SRF_TRAIN_CONV_DEBUG"):
        from srfdet3d_amd import train_conv
        train_conv._DEBUG.clear()   # the warm-up iterations contain MIOpen's solver search
    if world > 1:
        dist.barrier()
    comm_events.clear()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        loss = step(record=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        ntrain = sum(p.numel() for p in params)
        fwd = sum(e[0].elapsed_time(e[1]) for e in phases) / len(phases)
        bwd = sum(e[1].elapsed_time(e[2]) for e in phases) / len(phases)
        optm = sum(e[2].elapsed_time(e[3]) for e in phases) / len(phases)
        step_ms = dt / a.iters * 1e3
        comm_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in comm_events) / a.iters if comm_events else 0.0
        # the per-bucket intervals start when the hook is CALLED (an event on the compute stream), so each contains the queueing behind
        # earlier buckets and they overlap one another: their sum is an upper bound; the span from the first bucket's start to the last
        # bucket's end (per step) is the time the communication stream was engaged at most
        span_ms = 0.0
        if comm_events:
            per_step = max(1, len(comm_events) // a.iters)
            for i in range(0, len(comm_events) - per_step + 1, per_step):
                span_ms += comm_events[i][0].elapsed_time(comm_events[i + per_step - 1][1])
            span_ms /= a.iters
        comm_bytes = sum(n for _, _, n in comm_events) / a.iters if comm_events else 0
        print(json.dumps(dict(metric="training iterations/s, srfdet_voxel_nusc_LC", value=round(a.iters / dt, 3), n_gpus=world,
                              frames_per_s=round(a.iters * a.bs * world / dt, 3), bs_per_gpu=a.bs, num_proposals=a.np,
                              image=f"{h}x{w}", trainable_params=ntrain, grad_bytes_per_step=4 * ntrain, last_loss=round(loss, 4),
                              ms_per_step=round(step_ms, 2),
                              phases_ms=dict(forward_and_loss=round(fwd, 2), backward_incl_overlapped_allreduce=round(bwd, 2),
                                             clip_and_optimizer=round(optm, 2)),
                              backward_share_of_step=round(bwd / step_ms, 4),
                              grad_allreduce=dict(ms_on_comm_stream_per_step=round(comm_ms, 3), first_start_to_last_end_ms_per_step=round(span_ms, 3),
                                                  upper_bound=True, verified_on_gpus=None if world == 1 else world, bytes_per_step=int(comm_bytes),
                                                  buckets_per_step=round(len(comm_events) / a.iters, 1),
                                                  share_of_step=round(comm_ms / step_ms, 4),
                                                  note="RCCL all-reduce of the DDP buckets; per-bucket intervals from the hook call to the future's callback "
                                                       "(they include queueing behind earlier buckets and overlap one another: an UPPER bound, as is the "
                                                       "first-start-to-last-end span); the buckets overlap with backward (world == 1: no collective)"),
                              loss_scalar_allreduces_per_step=1 if world > 1 else 0)))
    if rank == 0 and a.kernel_table:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(2):
                step()
            torch.cuda.synchronize()
        rows = {}
        for ev in prof.events():
            if ev.device_type.name != "CUDA":
                continue
            r = rows.setdefault(ev.name, [0, 0.0])
            r[0] += 1
            r[1] += ev.device_time_total if hasattr(ev, "device_time_total") else ev.cuda_time_total
        tot = sum(r[1] for r in rows.values())
        with open(a.kernel_table, "w") as fh:
            fh.write(f"Steady-state kernels of one training iteration (`tools/train_bench.py --bs {a.bs} --np {a.np} --img-hw {a.img_hw}`, "
                     f"torch.profiler over 2 iterations after the timed region; {a.iters / dt:.2f} iterations/s untraced).\n\n")
            fh.write(f"GPU kernel time per iteration: {tot / 2e3:.1f} ms\n\n| kernel | calls / iteration | ms / iteration | % |\n|---|---|---|---|\n")
            for name, (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:60]:
                fh.write(f"| `{name[:150]}` | {n / 2:.1f} | {us / 2e3:.3f} | {100 * us / tot:.1f} |\n")
    if rank == 0 and os.environ.get("SRF_TRAIN_OP_TABLE"):
        # developer: device time per (aten op, input shapes) of one iteration -- where the element-wise launches come from
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
            step()
            torch.cuda.synchronize()
        rows = []
        for ev in prof.key_averages(group_by_input_shape=True):
            t = getattr(ev, "self_device_time_total", None)
            if t is None:
                t = ev.self_cuda_time_total
            if t > 0:
                rows.append((t / 1e3, ev.count, ev.key, str(ev.input_shapes)[:150]))
        rows.sort(reverse=True)
        with open(os.environ["SRF_TRAIN_OP_TABLE"], "w") as fh:
            for t, n, key, shp in rows[:120]:
                fh.write(f"{t:8.3f} ms {n:5d} {key:40s} {shp}\n")
    if rank == 0 and os.environ.get("SRF_TRAIN_CONV_DEBUG"):
        from srfdet3d_amd import train_conv
        for k, n, ms in train_conv.debug_report()[:12]:
            print("wgrad", k, "calls", n, f"{ms:.2f} ms each")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
