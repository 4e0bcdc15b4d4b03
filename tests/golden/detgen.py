"""Deterministic tensors keyed by name, shared by make_fixtures.py (which feeds them to the reference's
own Python) and by the tests (which feed them to this repo's implementation).

Storing only a name -> seed rule keeps the committed fixtures down to the expected OUTPUTS: every input and
every weight is regenerated on both sides from `det(name, shape)`.
"""
import zlib

import numpy as np


def _rng(name):
    return np.random.default_rng(zlib.crc32(name.encode("utf-8")))


def det(name, shape, scale=1.0, shift=0.0):
    """float32 array ~ N(shift, scale^2), a pure function of (name, shape)."""
    return (_rng(name).standard_normal(tuple(shape)).astype(np.float32) * np.float32(scale) + np.float32(shift))


def det_uniform(name, shape, lo=0.0, hi=1.0):
    return _rng(name).uniform(lo, hi, size=tuple(shape)).astype(np.float32)


def det_param(name, shape):
    """Weight rule used for every module parameter / buffer in the fixtures.

    matrices / conv kernels: N(0, 2/(fan_in+fan_out)); LayerNorm/BatchNorm weight: 1 + 0.1 N(0,1);
    biases and running_mean: 0.1 N(0,1); running_var: U(0.5, 1.5).
    """
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "running_var":
        return det_uniform(name, shape, 0.5, 1.5)
    if leaf == "num_batches_tracked":
        return np.zeros(shape, np.int64)
    if len(shape) >= 2:
        recept = int(np.prod(shape[2:])) if len(shape) > 2 else 1
        fan_out, fan_in = shape[0] * recept, shape[1] * recept
        return det(name, shape, scale=float(np.sqrt(2.0 / (fan_in + fan_out))))
    if leaf == "weight":
        return det(name, shape, scale=0.1, shift=1.0)
    return det(name, shape, scale=0.1)


def load_det_params(module, prefix=""):
    """Overwrite every parameter and buffer of a torch module with det_param(prefix + its name)."""
    import torch
    with torch.no_grad():
        for n, p in list(module.named_parameters()) + list(module.named_buffers()):
            v = det_param(prefix + n, p.shape)
            p.copy_(torch.from_numpy(v).to(p.dtype))
    return module
