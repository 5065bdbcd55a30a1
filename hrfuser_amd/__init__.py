"""hrfuser_amd — MI355X (gfx950) native HRFuser backbone forward/backward.

Only the hot path of timbroed/HRFuser is here: the `HRFuserHRFormerBased`
backbone behind the reference's `BACKBONES.register_module()` surface, executed by hand-written
HIP kernels through the C ABI in `include/hrfuser_hip.h`.  No CPU fallback exists.
"""
from .registry import BACKBONES, build_backbone          # noqa: F401
from .backbone import (HRFuserHRFormerBased, HRFuserHRNetBased, HRFormer, HRFuserFusionBlock, HRFormerBlock, HRFomerModule,  # noqa: F401
                       Bottleneck, CrossFFN, LocalWindowSelfAttention, MultiWindowCrossAttention,
                       WindowMSA, WindowMCA)

from .registry import NECKS                                # noqa: F401,E402
from .neck import HRFPN, build_neck                        # noqa: F401,E402
from .pipeline import DeviceInputPipeline                  # noqa: F401,E402

__all__ = ['BACKBONES', 'NECKS', 'build_backbone', 'build_neck', 'HRFuserHRFormerBased', 'HRFuserHRNetBased', 'HRFormer', 'HRFPN']
