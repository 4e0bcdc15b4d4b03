"""Minimal in-repo stand-ins for the slice of mmcv / mmdet / mmdet3d that the reference configs and model
classes touch.  None of those packages is installable here; when they are present a
maintainer registers the srfdet3d_amd classes into THEIR registries instead."""
from .config import Config, ConfigDict  # noqa: F401
from .registry import (BACKBONES, BBOX_ASSIGNERS, DETECTORS, HEADS, LOSSES, MATCH_COST, MIDDLE_ENCODERS, NECKS,  # noqa
                       NORM_LAYERS, ROI_EXTRACTORS, VOXEL_ENCODERS, Registry, build_backbone, build_head, build_loss,
                       build_middle_encoder, build_model, build_neck, build_roi_extractor, build_voxel_encoder)
