"""world_size-2 gloo tests of the multi-process pieces: the synchronised BN statistics
(collectives C2/C3) and the frame sharding / max-over-ranks timing reduction that bench.py uses."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from srfdet3d_amd.plugin.norm import NaiveSyncBatchNorm1dCustom
    torch.manual_seed(0)
    full = torch.randn(64, 5)
    x = full[rank::world].clone().requires_grad_(True)
    bn = NaiveSyncBatchNorm1dCustom(5, eps=1e-3, momentum=0.1).train()
    y = bn(x)
    y.square().sum().backward()
    # frame sharding: rank r takes frames r, r+world, ...; job time = max over ranks
    frames = list(range(rank, 10, world))
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    # numpy (pickled by value): torch tensors travel through /dev/shm files that disappear if this process exits first
    q.put((rank, y.detach().numpy(), bn.running_mean.numpy().copy(), bn.running_var.numpy().copy(), x.grad.numpy().copy(), frames,
           t.item()))
    dist.barrier()
    dist.destroy_process_group()


def test_sync_bn_and_sharding_two_ranks():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    res = [tuple(torch.from_numpy(v) if isinstance(v, np.ndarray) else v for v in r) for r in res]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    full = torch.randn(64, 5)
    # equal shards: mean of per-rank means == global mean; the reference uses biased var (E[x^2]-E[x]^2) for
    # both the normalisation and the running estimate (norm.py:72-75)
    mean = full.mean(0)
    var = (full * full).mean(0) - mean * mean
    want = (full - mean) / torch.sqrt(var + 1e-3)
    for rank, y, rm, rv, grad, frames, tmax in res:
        torch.testing.assert_close(y, want[rank::world], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(rm, 0.1 * mean, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(rv, 0.9 * torch.ones(5) + 0.1 * var, rtol=1e-5, atol=1e-6)
        assert frames == list(range(rank, 10, world))
        assert abs(tmax - 0.2) < 1e-12
    # gradient through the synchronised statistics equals single-process autograd on the concatenated batch
    xf = full.clone().requires_grad_(True)
    m = xf.mean(0)
    v = (xf * xf).mean(0) - m * m
    (((xf - m) / torch.sqrt(v + 1e-3)).square().sum()).backward()
    for rank, *_rest in res:
        torch.testing.assert_close(res[rank][4], xf.grad[rank::world], rtol=1e-4, atol=1e-5)
    assert sorted(res[0][5] + res[1][5]) == list(range(10))


def _gather_worker(rank, world, port, q, n_frames):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from srfdet3d_amd.sharding import frame_indices, gather_detections
    mine = frame_indices(n_frames, rank, world)
    part = [dict(frame=i, boxes=np.full((2, 9), float(i), np.float32), scores=np.array([0.5, 0.25], np.float32)) for i in mine]
    out = gather_detections(part, n_frames)
    q.put((rank, mine, None if out is None else [(d["frame"], float(d["boxes"][0, 0])) for d in out]))
    dist.barrier()
    dist.destroy_process_group()


def test_frame_sharding_and_end_of_run_gather_two_ranks():
    """frames i -> rank i mod W (DistributedSampler(shuffle=False), tools/test.py:195-200), no collective on
    the data path, one gather of the detections at the end that restores dataset order and cuts the padding."""
    from srfdet3d_amd.sharding import frame_indices
    assert frame_indices(7, 0, 1) == list(range(7))
    assert frame_indices(7, 0, 2) == [0, 2, 4, 6] and frame_indices(7, 1, 2) == [1, 3, 5, 0]      # padded with the head
    assert frame_indices(2, 3, 4) == [1] and frame_indices(1, 2, 4) == [0]
    assert sorted(sum((frame_indices(10, r, 4) for r in range(4)), []))[:10] != []               # every rank has ceil(n/W)
    assert all(len(frame_indices(10, r, 4)) == 3 for r in range(4))
    world, n = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q, n)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5, 0]
    assert res[1][2] is None
    assert res[0][2] == [(i, float(i)) for i in range(n)]


def _loss_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from srfdet3d_amd.plugin import heads, training

    class _Gt:
        def __init__(self, t):
            self.tensor, self.gravity_center = t, t[:, :3]

    g = torch.Generator().manual_seed(100 + rank)
    n_p, n_cls, layers = 16, 10, 3
    n_match = [[3 + rank, 2], [1 + 2 * rank, 4], [5, 1 + rank]]          # pairs per (layer, sample): different on the two ranks
    outs = [dict(pred_logits=torch.randn(2, n_p, n_cls, generator=g), pred_boxes=torch.randn(2, n_p, 10, generator=g)) for _ in range(layers)]
    gts = [torch.cat([torch.rand(8, 3, generator=g) * 40 - 20, torch.rand(8, 3, generator=g) + 0.5, torch.randn(8, 3, generator=g)], 1)
           for _ in range(2)]
    labels = [torch.randint(0, n_cls, (8,), generator=g) for _ in range(2)]
    calls = {"assign": 0}

    def assigner(out, gts_, labels_, head_idx):
        li = calls["assign"] % layers
        calls["assign"] += 1
        return [(torch.arange(n_match[li][b]), torch.arange(n_match[li][b])) for b in range(2)]

    hd = object.__new__(heads.SRFDetHead)
    torch.nn.Module.__init__(hd)
    hd.assigner, hd.num_heads, hd.deep_supervision, hd.num_classes, hd.sync_cls_avg_factor = assigner, layers, True, n_cls, True
    hd.pc_range = [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0]
    hd.code_weights = torch.nn.Parameter(torch.tensor([1.0] * 8 + [0.2, 0.2]), requires_grad=False)
    hd.loss_cls = training.FocalLoss(use_sigmoid=True, gamma=2.0, alpha=0.25, reduction="sum", loss_weight=2.0)
    hd.loss_bbox = training.L1Loss(reduction="sum", loss_weight=0.25)
    n_allreduce = {"n": 0}
    real = dist.all_reduce

    def counting(*a, **k):
        n_allreduce["n"] += 1
        return real(*a, **k)

    dist.all_reduce = counting
    outputs = dict(outs[0], aux_outputs=outs[1:])
    batched = hd.loss_ota(outputs, [_Gt(t) for t in gts], labels)
    n_batched = n_allreduce["n"]
    # the reference's own sequence: one reduce_mean per loss and layer
    n_allreduce["n"] = 0
    seq = {}
    order = [(outs[0], ""), (outs[1], "s.0."), (outs[2], "s.1.")]
    for li, (o, prefix) in enumerate(order):
        idx = [(torch.arange(n_match[li][b]), torch.arange(n_match[li][b])) for b in range(2)]
        gl = [torch.cat((t[:, :3], t[:, 3:]), 1) for t in gts]
        seq[prefix + "loss_cls"] = hd.loss_classification(o, labels, idx)
        seq[prefix + "loss_bbox"] = hd.loss_boxes(o, gl, idx)
    dist.all_reduce = real
    q.put((rank, n_batched, n_allreduce["n"], {k: float(v) for k, v in batched.items()}, {k: float(v) for k, v in seq.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_loss_counts_travel_in_one_all_reduce_two_ranks():
    """SURVEY 8e / collective C4: the matched-pair counts of all decoder layers in ONE all-reduce (the reference: two
    `reduce_mean` + `.item()` per layer, srfdet_head.py:1134, :1178) -- same losses as the per-layer sequence, on ranks whose
    counts differ."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loss_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, n_batched, n_seq, batched, seq in res:
        assert n_batched == 1 and n_seq == 6
        assert set(batched) == set(seq)
        for k in seq:
            assert abs(batched[k] - seq[k]) <= 1e-6 * max(1.0, abs(seq[k])), (k, batched[k], seq[k])
