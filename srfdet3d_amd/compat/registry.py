"""`Registry` with `register_module` / `build(cfg)` and the builder functions the reference imports
(mmdet3d.models.builder; srfdet.py:5-8, srfdet_head.py:17)."""
import copy
import inspect


class Registry:
    def __init__(self, name):
        self.name = name
        self._modules = {}

    def register_module(self, name=None, force=False, module=None):
        if isinstance(name, type):  # @REG.register_module without parentheses
            return self.register_module()(name)

        def deco(cls):
            key = name if isinstance(name, str) else cls.__name__
            if key in self._modules and not force and self._modules[key] is not cls:
                raise KeyError(f"{key} is already registered in {self.name}")
            self._modules[key] = cls
            return cls

        if module is not None:
            return deco(module)
        return deco

    def get(self, key):
        return self._modules.get(key)

    def __contains__(self, key):
        return key in self._modules

    def build(self, cfg, default_args=None):
        if cfg is None:
            return None
        cfg = dict(copy.deepcopy(cfg))
        if default_args:
            for k, v in default_args.items():
                cfg.setdefault(k, v)
        t = cfg.pop("type")
        cls = self.get(t) if isinstance(t, str) else t
        if cls is None:
            raise KeyError(f"{t} is not in the {self.name} registry")
        return cls(**cfg)


DETECTORS = Registry("detector")
VOXEL_ENCODERS = Registry("voxel_encoder")
MIDDLE_ENCODERS = Registry("middle_encoder")
BACKBONES = Registry("backbone")
NECKS = Registry("neck")
HEADS = Registry("head")
ROI_EXTRACTORS = Registry("roi_extractor")
ROI_LAYERS = Registry("roi_layer")
LOSSES = Registry("loss")
BBOX_ASSIGNERS = Registry("bbox_assigner")
MATCH_COST = Registry("match_cost")
NORM_LAYERS = Registry("norm_layer")
CONV_LAYERS = Registry("conv_layer")
ACTIVATION_LAYERS = Registry("activation_layer")
PIPELINES = Registry("pipeline")


def build_voxel_encoder(cfg):
    return VOXEL_ENCODERS.build(cfg)


def build_middle_encoder(cfg):
    return MIDDLE_ENCODERS.build(cfg)


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_roi_extractor(cfg):
    return ROI_EXTRACTORS.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)


def build_assigner(cfg):
    return BBOX_ASSIGNERS.build(cfg)


def build_match_cost(cfg):
    return MATCH_COST.build(cfg)


def build_model(cfg, train_cfg=None, test_cfg=None):
    """mmdet3d.models.build_model as called at tools/test.py:203-204."""
    args = {}
    if train_cfg is not None:
        args["train_cfg"] = train_cfg
    if test_cfg is not None:
        args["test_cfg"] = test_cfg
    return DETECTORS.build(cfg, default_args=args)
