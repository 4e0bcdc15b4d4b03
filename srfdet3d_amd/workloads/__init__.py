"""Model sections of the reference configs as data (see tools/dump_model_cfgs.py) + the measurement overrides of
BASELINE.md section 4."""
import json
import os

from ..compat.config import ConfigDict, _wrap

_HERE = os.path.dirname(os.path.abspath(__file__))


def _untuple(o):
    if isinstance(o, dict):
        if set(o.keys()) == {"__tuple__"}:
            return tuple(_untuple(v) for v in o["__tuple__"])
        return {k: _untuple(v) for k, v in o.items()}
    if isinstance(o, list):
        return [_untuple(v) for v in o]
    return o


def model_cfg(name, **overrides):
    """`model` dict of reference config `name` (e.g. 'srfdet_voxel_nusc_L'); dotted-key overrides as --cfg-options."""
    with open(os.path.join(_HERE, name + ".json")) as f:
        cfg = _wrap(_untuple(json.load(f)["model"]))
    from ..compat.config import Config
    wrapper = Config(model=cfg)
    wrapper.merge_from_dict({"model." + k: v for k, v in overrides.items()})
    return wrapper.model


def build(name, num_proposals=None, train=False):
    from ..compat.registry import build_model
    ov = {}
    if num_proposals is not None:
        ov["bbox_head.num_proposals"] = num_proposals
    m = model_cfg(name, **ov)
    if not train:
        m["train_cfg"] = None
    return build_model(m)
