"""srfdet3d_amd -- MI355X-native implementation of the SRFDet3D hot path (voxelize -> sparse-conv SECOND ->
sparse-region-fusion decoder) behind the reference's registry / operator API.

Importing the package registers every class the reference configs name.  The compute kernels live in
libsrfdet3d_hip.so (csrc/, C ABI in include/srfdet3d.h); there is no CPU fallback for them.
"""
from .compat import Config, build_model  # noqa: F401
from .compat import necks as _necks  # noqa: F401  (registers FPN)
from .compat import resnet as _resnet  # noqa: F401  (registers ResNet for the r50 image-backbone configs)
from .plugin import training  # noqa: F401  (registers losses, match costs and the OTA assigner before the heads build)
from .plugin import backbones, detectors, heads, middle_encoders, norm, pillar, pipelines, voxel_encoders, vovnet  # noqa: F401
from .roi import RoIAlign, SingleRoIExtractor, bbox2roi  # noqa: F401
from .sparse import (SparseBasicBlock, SparseConv3d, SparseConvTensor, SparseSequential, SubMConv3d,  # noqa: F401
                     make_sparse_convmodule)
from .voxel_layer import DynamicScatter, Voxelization  # noqa: F401

__version__ = "0.1.0"
