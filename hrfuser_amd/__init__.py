"""hrfuser_amd — MI355X (gfx950) native HRFuser backbone forward/backward.

Only the hot path of timbroed/HRFuser is here: the `HRFuserHRFormerBased`
backbone behind the reference's `BACKBONES.register_module()` surface, executed by hand-written
HIP kernels through the C ABI in `include/hrfuser_hip.h`.  No CPU fallback exists.
"""
from .registry import BACKBONES, build_backbone          # noqa: F401
from .backbone import (HRFuserHRFormerBased, HRFuserFusionBlock, HRFormerBlock, HRFomerModule,  # noqa: F401
                       Bottleneck, CrossFFN, LocalWindowSelfAttention, MultiWindowCrossAttention,
                       WindowMSA, WindowMCA)

__all__ = ['BACKBONES', 'build_backbone', 'HRFuserHRFormerBased']
