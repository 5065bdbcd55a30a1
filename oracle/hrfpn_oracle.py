"""CPU restatement (plain PyTorch) of the reference HRFPN neck - TEST INFRASTRUCTURE ONLY.

Follows mmdet/models/necks/hrfpn.py:77-100 (forward) and :49-70 (layers): the four backbone maps are
bilinearly up-sampled to the finest grid (F.interpolate(scale_factor=2**i, mode='bilinear'), i.e.
align_corners=False), concatenated on channels, reduced by a 1x1 conv (mmcv ConvModule with norm_cfg=None and
act_cfg=None = nn.Conv2d with bias), average-pooled into a 5-level pyramid (kernel = stride = 2**i) and each
level passes through its own 3x3 conv (bias, stride 1, padding 1).  State-dict keys equal the reference's
(`reduction_conv.conv.*`, `fpn_convs.{i}.conv.*`).  Only tests/ and __graft_entry__.smoke() may import this.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _Conv(nn.Module):
    """mmcv ConvModule(norm_cfg=None, act_cfg=None): a biased conv under the attribute name `conv`."""

    def __init__(self, cin, cout, k, stride=1, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=padding, bias=True)

    def forward(self, x):
        return self.conv(x)


class HRFPNOracle(nn.Module):
    def __init__(self, in_channels, out_channels, num_outs=5, pooling_type='AVG', stride=1, **_):
        super().__init__()
        assert isinstance(in_channels, list)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_ins, self.num_outs = len(in_channels), num_outs
        self.reduction_conv = _Conv(sum(in_channels), out_channels, 1)                     # hrfpn.py:53-58
        self.fpn_convs = nn.ModuleList(_Conv(out_channels, out_channels, 3, stride, 1)     # hrfpn.py:60-70
                                       for _ in range(num_outs))
        self.pooling = F.max_pool2d if pooling_type == 'MAX' else F.avg_pool2d             # hrfpn.py:72-75

    def forward(self, inputs):
        assert len(inputs) == self.num_ins                                                 # hrfpn.py:79
        outs = [inputs[0]]
        for i in range(1, self.num_ins):
            outs.append(F.interpolate(inputs[i], scale_factor=2 ** i, mode='bilinear'))    # hrfpn.py:81-83
        out = self.reduction_conv(torch.cat(outs, dim=1))                                  # hrfpn.py:84-88
        outs = [out]
        for i in range(1, self.num_outs):
            outs.append(self.pooling(out, kernel_size=2 ** i, stride=2 ** i))              # hrfpn.py:90-91
        return tuple(self.fpn_convs[i](outs[i]) for i in range(self.num_outs))             # hrfpn.py:94-100
