"""HRFPN neck on the HIP engine - the immediate consumer of the backbone's four maps.

Drop-in for mmdet's `HRFPN` (mmdet/models/necks/hrfpn.py:12-100): same registry name, constructor
keywords, `forward(inputs) -> tuple[num_outs]` contract and state-dict keys (`reduction_conv.conv.*`,
`fpn_convs.{i}.conv.*`).  Data path on the device:

  up-sample + concat  : hrf_bilinear_up_into  - each branch written straight into its channel slice of ONE
                        NHWC row buffer [B*H*W][sum(in_channels)] (no separate interpolate outputs, no cat)
  reduction 1x1 conv  : hrf_conv_fwd          - row GEMM over the concatenated rows, bias in the accumulators
  average pyramid     : hrf_avg_pool          - k = 2, 4, 8, ... on the reduced rows
  3x3 output convs    : hrf_conv_fwd          - halo-tiled MFMA kernel, one lane per pyramid level

and the reverse tape mirrors it (weight gradients on the deferred side phase, like the backbone).
"""
import os

import torch
import torch.nn as nn

from . import runtime as R
from .backbone import HipModule, _BackboneFn
from .registry import NECKS


_WIDE = os.environ.get('HRF_NECK_WIDE', '1') != '0'       # 0: route the 3x3 convolutions through hrf_conv_fwd instead


class _ConvModule(nn.Module):
    """mmcv ConvModule(norm_cfg=None, act_cfg=None): a biased nn.Conv2d under the attribute `conv`
    (hrfpn.py:53-70) - parameter container only, the launches are issued by HRFPN._run."""

    def __init__(self, cin, cout, k, stride=1, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=padding, bias=True)


def _conv3_wide(ctx, x, conv):
    """3x3 / stride-1 convolution with Cin % 32 == 0 and Cout % 64 == 0 on the packed-weight MFMA engine
    (conv3w_engine.hip): forward and backward-data are the same kernel on the two packings of the weights."""
    L, s = ctx.L, ctx.stream
    B, H, W, C = x.t.shape
    w, b = conv.weight, conv.bias
    Cout = w.shape[0]
    out = R.Act(R._new((B, H, W, Cout), x.t.device))
    wp = R._new((9 * Cout * C,), x.t.device)
    L.hrf_conv3_pack(w, Cout, C, 0, wp, s)
    L.hrf_conv3_packed(x.t, C, wp, b, out.t, Cout, 0, B, H, W, C, Cout, s)

    def bwd():
        if out.grad is None:
            return
        L, s = ctx.L, ctx.stream
        if w.requires_grad:
            strides = R._nhwc_strides(B, H, W, C)
            bgrad = b.grad if (b is not None and b.requires_grad) else None
            if Cout % 128 == 0:
                scr = R._new((L.hrf_conv3_wgrad_wide_scratch(B, H, W, C, Cout),), x.t.device)
                ctx.side_launch(lambda: L.hrf_conv3_wgrad_wide(out.grad, Cout, x.t, C, B, H, W, C, Cout, w.grad, bgrad,
                                                               scr, ctx.stream), cost=float(B * H * W) * C * Cout)
            else:
                ctx.side_launch(lambda: L.hrf_conv_bwd_weight(
                    out.grad, Cout, 0, None, None, None, None, x.t, *strides, B, H, W, C, 3, 1, Cout,
                    R.TF_NONE, None, None, None, w.grad, bgrad, ctx.stream))
        if x.needs_grad:
            L.hrf_conv3_pack(w, Cout, C, 1, wp, s)                 # the forward pack is dead by now: reuse its buffer
            g, acc = x.grad_target()
            L.hrf_conv3_packed(out.grad, Cout, wp, None, g, C, acc, B, H, W, Cout, C, s)
    ctx.push(bwd)
    return out


def _conv_bias(ctx, x, conv):
    """Act -> Act: dense k=1|3 convolution with bias, no normalisation, no activation."""
    L, s = ctx.L, ctx.stream
    B, H, W, C = x.t.shape
    w, b = conv.weight, conv.bias
    Cout, KH, stride = w.shape[0], w.shape[2], conv.stride[0]
    if KH == 3 and stride == 1 and C % 32 == 0 and Cout % 64 == 0 and C % 64 == 0 and _WIDE:
        return _conv3_wide(ctx, x, conv)
    Ho, Wo = R._conv_out_hw(H, W, KH, stride)
    out = R.Act(R._new((B, Ho, Wo, Cout), x.t.device))
    strides = R._nhwc_strides(B, H, W, C)
    L.hrf_conv_fwd(x.t, *strides, B, H, W, C, w, b, KH, stride, Cout, out.t, Cout, 0, None, None, 0,
                   R.TF_NONE, None, None, None, None, None, None, 0.0, s)

    def bwd():
        if out.grad is not None:
            R._conv_backward(ctx, x, w, b, KH, stride, Cout, out.grad, Cout, 0, None, None)
    ctx.push(bwd)
    return out


@NECKS.register_module()
class HRFPN(HipModule):
    """HRFPN (hrfpn.py:12-100).  `inputs`: the backbone's list of `num_ins` logical-NCHW maps, branch i at
    1/2**i of branch 0's resolution; returns a tuple of `num_outs` logical-NCHW maps (channels-last memory)."""

    def __init__(self, in_channels, out_channels, num_outs=5, pooling_type='AVG', conv_cfg=None, norm_cfg=None,
                 with_cp=False, stride=1, init_cfg=dict(type='Caffe2Xavier', layer='Conv2d')):
        super().__init__()
        assert isinstance(in_channels, list)                                  # hrfpn.py:43
        if conv_cfg is not None or norm_cfg is not None:
            raise NotImplementedError('HRFPN on the HIP engine: conv_cfg / norm_cfg must be None '
                                      '(every HRFuser config leaves them unset, hrfpn.py:38-39)')
        if pooling_type == 'MAX':
            raise NotImplementedError('HRFPN on the HIP engine implements pooling_type="AVG" (the default and '
                                      'what every HRFuser config uses); MAX pooling is not built')
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.num_ins = len(in_channels)
        self.num_outs = num_outs
        self.with_cp = with_cp                    # activation checkpointing: accepted, a no-op here
        self.conv_cfg, self.norm_cfg = conv_cfg, norm_cfg
        self.init_cfg = init_cfg
        self.reduction_conv = _ConvModule(sum(in_channels), out_channels, 1)
        self.fpn_convs = nn.ModuleList(_ConvModule(out_channels, out_channels, 3, stride, 1)
                                       for _ in range(num_outs))

    def init_weights(self):
        """mmcv caffe2_xavier_init = kaiming_uniform(a=1, fan_in, leaky_relu), bias 0 (init_cfg, hrfpn.py:41)."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, a=1, mode='fan_in', nonlinearity='leaky_relu')
                nn.init.constant_(m.bias, 0)
        self.params_updated()

    # -- execution ----------------------------------------------------------------------------------
    def forward(self, inputs):
        assert len(inputs) == self.num_ins                                    # hrfpn.py:79
        inputs = tuple(t.float() for t in inputs)
        B, _, H, W = inputs[0].shape
        for i, t in enumerate(inputs):
            if t.shape[0] != B or t.shape[1] != self.in_channels[i]:
                raise RuntimeError(f'HRFPN input {i}: expected ({B}, {self.in_channels[i]}, H, W), got {tuple(t.shape)}')
            if (t.shape[2] << i, t.shape[3] << i) != (H, W):
                # torch.cat in the reference raises for the same inputs (hrfpn.py:84)
                raise RuntimeError(f'Sizes of tensors must match except in dimension 1: input {i} up-sampled by '
                                   f'{1 << i} is {(t.shape[2] << i, t.shape[3] << i)}, expected {(H, W)}')
        anchor = None
        if torch.is_grad_enabled():
            anchor = self.__dict__.get('_hrf_anchor')
            if anchor is None or anchor.device != inputs[0].device:
                anchor = torch.zeros(1, device=inputs[0].device, requires_grad=True)
                self.__dict__['_hrf_anchor'] = anchor
        return tuple(_BackboneFn.apply(self, anchor, *inputs))

    def _wrap_inputs(self, inputs):
        # the backbone hands over channels-last memory: this permute+contiguous is then a view, not a copy
        return [R.Act(t.permute(0, 2, 3, 1).contiguous(), bool(t.requires_grad)) for t in inputs]

    def _zeros(self, n, B, dev):
        z = self.__dict__.get('_zero_pad')
        if z is None or z.numel() < B * n or z.device != dev:
            z = self.__dict__['_zero_pad'] = torch.zeros(B * 16, device=dev, dtype=torch.float32)
        return z

    def _reduce_gemm(self, ctx, cat, Csum):
        """reduction_conv (hrfpn.py:53-58,85) on the row-GEMM engine: `cat` rows are padded to K % 16 == 0."""
        L, s = ctx.L, ctx.stream
        conv = self.reduction_conv.conv
        w, b = conv.weight, conv.bias
        B, H, W, Kp = cat.t.shape
        Cout, rows, dev = w.shape[0], B * H * W, cat.t.device
        red = R.Act(R._new((B, H, W, Cout), dev))
        wp = R._new((Cout * Kp,), dev)
        L.hrf_rowgemm_pack(w, Cout, Csum, 0, Cout, Kp, wp, s)
        L.hrf_rowgemm(cat.t, Kp, wp, b, red.t, Cout, 0, rows, Kp, Cout, s)

        def bwd():
            if red.grad is None:
                return
            L, s = ctx.L, ctx.stream
            if w.requires_grad:
                bgrad = b.grad if (b is not None and b.requires_grad) else None
                strides = (H * W * Kp, W * Kp, Kp, 1)
                ctx.side_launch(lambda: L.hrf_conv_bwd_weight(
                    red.grad, Cout, 0, None, None, None, None, cat.t, *strides, B, H, W, Csum, 1, 1, Cout,
                    R.TF_NONE, None, None, None, w.grad, bgrad, ctx.stream))
            L.hrf_rowgemm_pack(w, Cout, Csum, 1, Kp, Cout, wp, s)             # [Kp][Cout], zero rows past Csum
            cat.grad = R._new((B, H, W, Kp), dev)
            L.hrf_rowgemm(red.grad, Cout, wp, None, cat.grad, Kp, 0, rows, Cout, Kp, s)
        ctx.push(bwd)
        return red

    def _run(self, ctx, srcs):
        L, s = ctx.L, ctx.stream
        B, H, W, _ = srcs[0].t.shape
        dev = srcs[0].t.device
        Csum = sum(self.in_channels)
        gemm = _WIDE and self.out_channels % 16 == 0          # LDS-tiled MFMA row GEMM for the reduction convolution
        ld = (Csum + 15) // 16 * 16 if gemm else Csum         # its K is padded to 16 with zero columns
        cat = R.Act(R._new((B, H, W, ld), dev))
        off, spans = 0, []
        for a in srcs:                                                        # hrfpn.py:80-84
            _, Hs, Ws, C = a.t.shape
            L.hrf_bilinear_up_into(a.t, Hs, Ws, C, cat.t, ld, off, B, H, W, s)
            spans.append((a, off, Hs, Ws, C))
            off += C
        if ld > Csum:
            # zero the pad columns: the "up-sampling" of a 1x1 map of zeros into them (the buffer is fresh memory)
            L.hrf_bilinear_up_into(self._zeros(ld - Csum, B, dev), 1, 1, ld - Csum, cat.t, ld, Csum, B, H, W, s)

        def bwd_cat():
            if cat.grad is None:
                return
            for a, o, Hs, Ws, C in spans:
                if not a.needs_grad:
                    continue
                if (Hs, Ws) == (H, W):                    # branch 0: the concat only copied it
                    g, acc = a.grad_target()
                    ctx.L.hrf_slice_cols(cat.grad, ld, o, B * H * W, C, g, acc, ctx.stream)
                    continue
                du = R._new((B, Hs, Ws, C), dev)
                ctx.L.hrf_bilinear_up_bwd(cat.grad, ld, o, B, H, W, C, None, Hs, Ws, du, None, ctx.stream)
                a.add_grad(du)
        ctx.push(bwd_cat)

        red = self._reduce_gemm(ctx, cat, Csum) if gemm else _conv_bias(ctx, cat, self.reduction_conv.conv)   # hrfpn.py:85-88
        # hrfpn.py:89-91 pools `red` with kernel 2**i for every level; floor-mode k x k windows nest exactly
        # (rows/cols dropped at level i are dropped at every coarser level too), so level i is the 2 x 2 average
        # of level i-1: the full-resolution map is read once instead of num_outs-1 times, in both directions
        levels = [red]
        for i in range(1, self.num_outs):
            par = levels[-1]
            _, Hp, Wp, _ = par.t.shape
            if Hp // 2 < 1 or Wp // 2 < 1:
                raise RuntimeError(f'HRFPN level {i}: avg_pool2d kernel {1 << i} is larger than the {H}x{W} map')
            p = R.Act(R._new((B, Hp // 2, Wp // 2, self.out_channels), dev))
            L.hrf_avg_pool(par.t, B, Hp, Wp, self.out_channels, 2, p.t, s)

            def bwd_pool(p=p, par=par, Hp=Hp, Wp=Wp):
                if p.grad is None:
                    return
                g, acc = par.grad_target()
                ctx.L.hrf_avg_pool_bwd(p.grad, B, Hp, Wp, self.out_channels, 2, g, acc, ctx.stream)
            ctx.push(bwd_pool)
            levels.append(p)
        # hrfpn.py:92-100 - the pyramid levels are independent: one lane each (level 0 is 3/4 of the work,
        # the small levels fill the CUs its tail leaves idle)
        outs = [None] * self.num_outs
        lanes = ctx.fork(self.num_outs)
        for i in range(self.num_outs):
            with ctx.on(lanes[i]):
                outs[i] = _conv_bias(ctx, levels[i], self.fpn_convs[i].conv)
        ctx.join(lanes)
        return outs


def build_neck(cfg):
    """mmdet.models.builder.build_neck (builder.py:23-25)."""
    return NECKS.build(cfg)
