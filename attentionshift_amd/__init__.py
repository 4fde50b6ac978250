"""attentionshift_amd -- MI355X (gfx950) native hot path of AttentionShift.

Exposes the reference's plugin surface (mmdet-style registries, same class names / kwargs):
    BACKBONES: VisionTransformerDet, SwinTransformer (the detection-style Swin backbone of BASELINE config 5)
    HEADS:     AttnShiftRoIHead (also registered as StandardRoIHeadMaskPointSampleDeformAttnReppoints)
Compute goes through libattnshift_hip.so (include/attnshift.h); there is no CPU fallback.
"""
from .registry import BACKBONES, HEADS, Registry, build_backbone, build_from_cfg, build_head, register_into_mmdet  # noqa: F401
from .config import Config, ConfigDict  # noqa: F401
from .backbone import VisionTransformerDet  # noqa: F401
from .swin_det import SwinTransformerDet  # noqa: F401  (BACKBONES: SwinTransformer)
from .roi_head import AttnShiftRoIHead  # noqa: F401
from .checkpoint import load_checkpoint, save_checkpoint  # noqa: F401
from .mil_head import MAEBoxHeadMIL  # noqa: F401  (registers into HEADS)
from .mae_heads import MAEBoxHeadRec, MAEMaskHeadPointSup  # noqa: F401
from .annotations import PointAnnotations, parse_ann_info  # noqa: F401

__version__ = "0.1.0"
