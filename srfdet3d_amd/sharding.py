"""Per-frame sharding of inference over the ranks of one node and the end-of-run gather of the detections (SURVEY 8e,
collective C5).

Reference: tools/test.py:195-200 builds the test loader with a `DistributedSampler(shuffle=False)` and hands the model to
mmdet's `multi_gpu_test`, whose `collect_results_*` interleaves the per-rank result lists back into dataset order and cuts
the padding (mmdet 2.28.2, third party).  Frames are independent: there is no collective on the data path, only this one
gather when the run is over.
"""
import math

import torch.distributed as dist


def frame_indices(n_frames, rank, world):
    """Dataset indices of `rank`: `DistributedSampler(shuffle=False)` semantics -- every rank gets ceil(n / world) frames,
    the list 0..n-1 being extended by its own head when n is not a multiple of world, rank r taking r, r + world, ..."""
    if world <= 1:
        return list(range(n_frames))
    per = math.ceil(n_frames / world)
    total = per * world
    idx = list(range(n_frames))
    while len(idx) < total:  # the sampler repeats the list when the padding is longer than it
        idx += idx[: total - len(idx)]
    return idx[rank:total:world]


def gather_detections(part, n_frames):
    """All ranks call it with their own result list (in the order of frame_indices); rank 0 returns the n_frames results in
    dataset order (interleaved, padding cut), the other ranks None -- `collect_results` of mmdet's `multi_gpu_test`.
    Results must be picklable host objects (numpy / python): detections have left the GPU by then."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(part)[:n_frames]
    world = dist.get_world_size()
    parts = [None] * world
    dist.all_gather_object(parts, list(part))
    if dist.get_rank() != 0:
        return None
    ordered = []
    for group in zip(*parts):
        ordered.extend(group)
    return ordered[:n_frames]
