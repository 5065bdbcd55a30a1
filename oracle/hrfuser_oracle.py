"""ORACLE — plain PyTorch (ATen, fp32/fp64, CPU) restatement of the HRFuser backbone.

TEST INFRASTRUCTURE ONLY.  Importers allowed: `tests/`, `__graft_entry__.smoke()`, the
`cpu_baseline` leg of `bench.py`, and `oracle/tools/*`.  The product package `hrfuser_amd`
never imports this file and has no CPU fallback.

What it restates (all citations relative to /root/reference):
  * HRFuserHRFormerBased.forward            mmdet/models/backbones/hrfuser_hrformer_based.py:522-628
  * HRFuserFusionBlock._inner_forward       hrfuser_hrformer_based.py:305-317
  * MultiWindowCrossAttention / WindowMCA   hrfuser_hrformer_based.py:189-248 / 106-151
  * HRFormerBlock._inner_forward            mmdet/models/backbones/hrformer.py:365-373
  * LocalWindowSelfAttention / WindowMSA    hrformer.py:184-236 / 96-131
  * CrossFFN                                hrformer.py:267-295
  * HRFomerModule fuse layers + HRModule.forward   hrformer.py:498-561, hrnet.py:184-207
  * HRNet stem / _make_layer / _make_transition_layer   hrnet.py:337-371,419-463,465-510
  * Bottleneck.forward                      mmdet/models/backbones/resnet.py:263-302

Parity pin: the reference repo holds NO tests or golden vectors for this path,
so this oracle is pinned against *outputs of the reference itself run in the build
container* (`oracle/tools/make_golden.py` imports the unmodified reference files behind an
mmcv factory stand-in and writes `tests/golden/*.npz`); `tests/test_oracle_golden.py`
re-checks the oracle against those vectors on every CPU run.

State-dict keys are identical to the reference's so the same seeded
parameter fill (`seeded_fill_`) produces identical weights in reference, oracle and product.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

WIN = 7


# ----------------------------------------------------------------------------- helpers
def make_bn(norm_cfg, c):
    """mmcv.cnn.build_norm_layer for the norm types the path accepts: BN / SyncBN -> nn.BatchNorm2d, GN ->
    nn.GroupNorm(num_groups, c) (hrnet.py:338-339, resnet.py:161-164, hrformer.py:269); eps defaults to 1e-5."""
    cfg = dict(norm_cfg or dict(type='BN'))
    kind = cfg.pop('type')
    assert kind in ('BN', 'SyncBN', 'GN'), kind
    requires_grad = cfg.pop('requires_grad', True)
    if kind == 'GN':
        bn = nn.GroupNorm(cfg['num_groups'], c, eps=cfg.get('eps', 1e-5))
    else:
        bn = nn.BatchNorm2d(c, eps=cfg.get('eps', 1e-5), momentum=cfg.get('momentum', 0.1))
    for p in bn.parameters():
        p.requires_grad_(requires_grad)
    return bn


def norm_name(norm_cfg, postfix):
    """attribute name of a postfixed norm layer: mmcv's abbreviation of the norm type + postfix ('bn1' / 'gn1')"""
    return ('gn' if dict(norm_cfg or dict(type='BN')).get('type') == 'GN' else 'bn') + str(postfix)


def make_ln(ln_cfg, c):
    cfg = dict(ln_cfg or dict(type='LN', eps=1e-6))
    assert cfg.pop('type') == 'LN'
    return nn.LayerNorm(c, eps=cfg.get('eps', 1e-5))


def conv_bn(cin, cout, k, stride, norm_cfg, relu, groups=1):
    layers = [nn.Conv2d(cin, cout, k, stride, k // 2, groups=groups, bias=False),
              make_bn(norm_cfg, cout)]
    if relu:
        layers.append(nn.ReLU(inplace=False))
    return nn.Sequential(*layers)


def rel_pos_index(wh=WIN, ww=WIN):
    """idx[i,j] = (yi-yj+wh-1)*(2ww-1) + (xi-xj+ww-1)   (hrformer.py:64-80)."""
    ys, xs = torch.meshgrid(torch.arange(wh), torch.arange(ww), indexing='ij')
    ys, xs = ys.reshape(-1), xs.reshape(-1)
    dy = ys[:, None] - ys[None, :] + wh - 1
    dx = xs[:, None] - xs[None, :] + ww - 1
    return dy * (2 * ww - 1) + dx


def window_pads(H, W, wh=WIN, ww=WIN):
    ph = math.ceil(H / wh) * wh - H
    pw = math.ceil(W / ww) * ww - W
    return ph // 2, ph - ph // 2, pw // 2, pw - pw // 2


def window_partition(x, H, W, wh=WIN, ww=WIN):
    """(B, H*W, C) -> (B*nW, wh*ww, C) with centred zero padding (hrformer.py:196-209)."""
    B, _, C = x.shape
    t, b, l, r = window_pads(H, W, wh, ww)
    x = F.pad(x.view(B, H, W, C), (0, 0, l, r, t, b))
    Hp, Wp = H + t + b, W + l + r
    x = x.view(B, Hp // wh, wh, Wp // ww, ww, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, wh * ww, C)


def window_merge(w, B, H, W, wh=WIN, ww=WIN):
    """inverse of window_partition incl. de-pad (hrformer.py:229-236)."""
    C = w.shape[-1]
    t, b, l, r = window_pads(H, W, wh, ww)
    Hp, Wp = H + t + b, W + l + r
    x = w.reshape(B, Hp // wh, Wp // ww, wh, ww, C).permute(0, 1, 3, 2, 4, 5)
    x = x.reshape(B, Hp, Wp, C)[:, t:t + H, l:l + W]
    return x.reshape(B, H * W, C)


def nchw_to_nlc(x):
    return x.flatten(2).transpose(1, 2).contiguous()


def nlc_to_nchw(x, H, W):
    B, _, C = x.shape
    return x.transpose(1, 2).reshape(B, C, H, W).contiguous()


class DropPath(nn.Module):
    """mmcv DropPath: x/keep * floor(keep + U[0,1)) per sample, train only."""

    def __init__(self, p):
        super().__init__()
        self.p = p

    def forward(self, x):
        if self.p == 0. or not self.training:
            return x
        keep = 1 - self.p
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        return x.div(keep) * (keep + torch.rand(shape, dtype=x.dtype, device=x.device)).floor()


# ----------------------------------------------------------------------------- attention
def _window_attention_core(q, k, v, heads, bias_table, index):
    """softmax(q*d^-1/2 k^T + RPB) v over 49-token windows (hrformer.py:103-128)."""
    Bw, N, C = q.shape
    d = C // heads
    split = lambda t: t.reshape(Bw, N, heads, d).permute(0, 2, 1, 3)
    q, k, v = split(q) * d ** -0.5, split(k), split(v)
    logits = q @ k.transpose(-2, -1)
    bias = bias_table[index.view(-1)].view(N, N, heads).permute(2, 0, 1)
    attn = (logits + bias.unsqueeze(0)).softmax(dim=-1)
    return (attn @ v).transpose(1, 2).reshape(Bw, N, C)


class WindowMSA(nn.Module):
    def __init__(self, c, heads):
        super().__init__()
        self.heads = heads
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * WIN - 1) ** 2, heads))
        self.register_buffer('relative_position_index', rel_pos_index())
        self.qkv = nn.Linear(c, 3 * c)
        self.out_proj = nn.Linear(c, c)

    def forward(self, x):
        q, k, v = self.qkv(x).chunk(3, dim=-1)
        o = _window_attention_core(q, k, v, self.heads, self.relative_position_bias_table,
                                   self.relative_position_index)
        return self.out_proj(o)


class LocalWindowSelfAttention(nn.Module):
    def __init__(self, c, heads):
        super().__init__()
        self.attn = WindowMSA(c, heads)

    def forward(self, x, H, W):
        B = x.shape[0]
        return window_merge(self.attn(window_partition(x, H, W)), B, H, W)


class WindowMCA(nn.Module):
    def __init__(self, c, heads, proj_drop_rate=0.):
        super().__init__()
        self.heads = heads
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * WIN - 1) ** 2, heads))
        self.register_buffer('relative_position_index', rel_pos_index())
        self.k_proj = nn.Linear(c, c)
        self.v_proj = nn.Linear(c, c)
        self.q_proj = nn.Linear(c, c)
        self.out_proj = nn.Linear(c, c)
        self.proj_drop = nn.Dropout(proj_drop_rate)

    def forward(self, xq, xkv):
        o = _window_attention_core(self.q_proj(xq), self.k_proj(xkv), self.v_proj(xkv), self.heads,
                                   self.relative_position_bias_table, self.relative_position_index)
        return self.proj_drop(self.out_proj(o))


class MultiWindowCrossAttention(nn.Module):
    def __init__(self, c, heads, proj_drop_rate=0.):
        super().__init__()
        self.attn = WindowMCA(c, heads, proj_drop_rate)

    def forward(self, x, y, H, W):
        B = x.shape[0]
        out = self.attn(window_partition(x, H, W), window_partition(y, H, W))
        return window_merge(out, B, H, W)


# ----------------------------------------------------------------------------- FFN / blocks
class CrossFFN(nn.Module):
    """1x1(+b) BN GELU -> dw3x3(+b) BN GELU -> 1x1(+b) BN GELU (hrformer.py:267-282)."""

    def __init__(self, c, hidden, norm_cfg):
        super().__init__()
        self.layers = nn.Sequential(
            nn.Conv2d(c, hidden, 1), make_bn(norm_cfg, hidden), nn.GELU(),
            nn.Conv2d(hidden, hidden, 3, 1, 1, groups=hidden), make_bn(norm_cfg, hidden), nn.GELU(),
            nn.Conv2d(hidden, c, 1), make_bn(norm_cfg, c), nn.GELU())

    def forward(self, x, H, W):
        B, _, C = x.shape
        y = self.layers(x.transpose(1, 2).reshape(B, C, H, W))
        return y.flatten(2).transpose(1, 2)


class HRFormerBlock(nn.Module):
    expansion = 1

    def __init__(self, c, heads, mlp_ratio, norm_cfg, ln_cfg, drop_path=0.):
        super().__init__()
        self.norm1 = make_ln(ln_cfg, c)
        self.attn = LocalWindowSelfAttention(c, heads)
        self.norm2 = make_ln(ln_cfg, c)
        self.ffn = CrossFFN(c, int(c * mlp_ratio), norm_cfg)
        self.drop_path = DropPath(drop_path) if drop_path > 0 else nn.Identity()

    def forward(self, x):
        B, C, H, W = x.shape
        t = nchw_to_nlc(x)
        t = t + self.drop_path(self.attn(self.norm1(t), H, W))
        t = t + self.drop_path(self.ffn(self.norm2(t), H, W))
        return nlc_to_nchw(t, H, W)


class HRFuserFusionBlock(nn.Module):
    expansion = 1

    def __init__(self, c, heads, mlp_ratio, norm_cfg, ln_cfg, drop_path, num_mod, proj_drop_rate):
        super().__init__()
        self.num_mod = num_mod
        self.norm1 = nn.ModuleList(make_ln(ln_cfg, c) for _ in range(num_mod))
        self.norm2 = nn.ModuleList(make_ln(ln_cfg, c) for _ in range(num_mod))
        self.attn = nn.ModuleList(MultiWindowCrossAttention(c, heads, proj_drop_rate)
                                  for _ in range(num_mod))
        self.norm3 = make_ln(ln_cfg, c)
        self.ffn = CrossFFN(c, int(c * mlp_ratio), norm_cfg)
        self.drop_path = DropPath(drop_path) if drop_path > 0 else nn.Identity()

    def forward(self, x, mods):
        B, C, H, W = x.shape
        t = nchw_to_nlc(x)
        q_src = t                                   # every modality queries the PRE-fusion camera
        for k in range(self.num_mod):
            z = nchw_to_nlc(mods[k])
            t = t + z + self.drop_path(self.attn[k](self.norm1[k](q_src), self.norm2[k](z), H, W))
        t = t + self.drop_path(self.ffn(self.norm3(t), H, W))
        return nlc_to_nchw(t, H, W)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, norm_cfg, downsample=None):
        super().__init__()
        self._nn = [norm_name(norm_cfg, k) for k in (1, 2, 3)]
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.add_module(self._nn[0], make_bn(norm_cfg, planes))
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.add_module(self._nn[1], make_bn(norm_cfg, planes))
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.add_module(self._nn[2], make_bn(norm_cfg, planes * 4))
        self.downsample = downsample

    def forward(self, x):
        n1, n2, n3 = (getattr(self, k) for k in self._nn)
        idt = x if self.downsample is None else self.downsample(x)
        y = F.relu(n1(self.conv1(x)))
        y = F.relu(n2(self.conv2(y)))
        y = n3(self.conv3(y))
        return F.relu(y + idt)


def make_bottleneck_layer(cin, planes, blocks, norm_cfg):
    ds = None
    if cin != planes * 4:
        ds = nn.Sequential(nn.Conv2d(cin, planes * 4, 1, bias=False), make_bn(norm_cfg, planes * 4))
    layers = [Bottleneck(cin, planes, norm_cfg, ds)]
    layers += [Bottleneck(planes * 4, planes, norm_cfg) for _ in range(1, blocks)]
    return nn.Sequential(*layers)


# ----------------------------------------------------------------------------- HR module
class HRFormerModule(nn.Module):
    """Parallel branches of HRFormerBlocks + cross-resolution exchange (hrnet.py:184-207,
    fuse layers hrformer.py:498-561)."""

    def __init__(self, channels, num_blocks, heads, mlp_ratios, norm_cfg, ln_cfg,
                 multiscale_output=True, drop_paths=None):
        super().__init__()
        nb = len(channels)
        self.nb = nb
        # hrformer.py:453-497: block b of EVERY branch takes drop_paths[b]
        dp = (lambda b: drop_paths[b]) if drop_paths is not None else (lambda b: 0.)
        self.branches = nn.ModuleList(
            nn.Sequential(*[HRFormerBlock(channels[i], heads[i], mlp_ratios[i], norm_cfg, ln_cfg, drop_path=dp(b))
                            for b in range(num_blocks[i])]) for i in range(nb))
        self.fuse_layers = None
        if nb > 1:
            rows = []
            for i in range(nb if multiscale_output else 1):
                row = []
                for j in range(nb):
                    if j > i:       # up: 1x1 + BN (bilinear resize applied in forward)
                        row.append(conv_bn(channels[j], channels[i], 1, 1, norm_cfg, relu=False))
                    elif j == i:
                        row.append(None)
                    else:           # down: (dw3x3 s2 + BN + 1x1 + BN [+ReLU]) x (i-j)
                        steps = []
                        for s in range(i - j):
                            last = s == i - j - 1
                            cout = channels[i] if last else channels[j]
                            mods = [nn.Conv2d(channels[j], channels[j], 3, 2, 1, groups=channels[j], bias=False),
                                    make_bn(norm_cfg, channels[j]),
                                    nn.Conv2d(channels[j], cout, 1, bias=False),
                                    make_bn(norm_cfg, cout)]
                            if not last:
                                mods.append(nn.ReLU(False))
                            steps.append(nn.Sequential(*mods))
                        row.append(nn.Sequential(*steps))
                rows.append(nn.ModuleList(row))
            self.fuse_layers = nn.ModuleList(rows)

    def forward(self, xs):
        if self.nb == 1:
            return [self.branches[0](xs[0])]
        xs = [self.branches[i](xs[i]) for i in range(self.nb)]
        outs = []
        for i, row in enumerate(self.fuse_layers):
            acc = None
            for j in range(self.nb):
                if j == i:
                    term = xs[j]
                elif j > i:
                    term = F.interpolate(row[j](xs[j]), size=xs[i].shape[2:], mode='bilinear',
                                         align_corners=False)
                else:
                    term = row[j](xs[j])
                acc = term if acc is None else acc + term
            outs.append(F.relu(acc))
        return outs


def make_transition(pre, cur, norm_cfg):
    """hrnet.py:419-463: same-index channel change -> 3x3 s1; new branches -> chain of 3x3 s2."""
    layers = []
    for i, c in enumerate(cur):
        if i < len(pre):
            layers.append(conv_bn(pre[i], c, 3, 1, norm_cfg, relu=True) if c != pre[i] else None)
        else:
            steps = []
            for j in range(i + 1 - len(pre)):
                cout = c if j == i - len(pre) else pre[-1]
                steps.append(conv_bn(pre[-1], cout, 3, 2, norm_cfg, relu=True))
            layers.append(nn.Sequential(*steps))
    return nn.ModuleList(layers)


def _inplace_relu_fix(seq):
    return seq


# ----------------------------------------------------------------------------- backbone
class HRFuserOracle(nn.Module):
    """Restatement of HRFuserHRFormerBased (same ctor kwargs, same state-dict keys)."""
    stage_block = 'HRFORMER'

    def __init__(self, extra, in_channels=3, conv_cfg=None,
                 norm_cfg=dict(type='SyncBN', requires_grad=True),
                 transformer_norm_cfg=dict(type='LN', eps=1e-6), norm_eval=False, with_cp=False,
                 drop_path_rate=0., zero_init_residual=False, multiscale_output=True,
                 pretrained=None, init_cfg=None, num_fused_modalities=2, mod_in_channels=(3, 3)):
        super().__init__()
        assert all(f'stage{i}' in extra for i in (1, 2, 3, 4))
        self.norm_eval = norm_eval
        self.M = M = num_fused_modalities
        self.extra = extra
        ncfg, lcfg = norm_cfg, transformer_norm_cfg
        self.pre_neck_fusion = bool(extra.get('LidarStageD'))       # hrfuser_hrformer_based.py:364-366
        # camera stem + stage1 (hrnet.py:337-371)
        self.conv1 = nn.Conv2d(in_channels, 64, 3, 2, 1, bias=False)
        self._stem_nn = [norm_name(ncfg, 1), norm_name(ncfg, 2)]
        self.add_module(self._stem_nn[0], make_bn(ncfg, 64))
        self.conv2 = nn.Conv2d(64, 64, 3, 2, 1, bias=False)
        self.add_module(self._stem_nn[1], make_bn(ncfg, 64))
        s1 = extra['stage1']
        assert s1['block'] == 'BOTTLENECK'
        self.layer1 = make_bottleneck_layer(64, s1['num_channels'][0], s1['num_blocks'][0], ncfg)
        pre = [s1['num_channels'][0] * 4]
        for si in (2, 3, 4):
            cfg = extra[f'stage{si}']
            assert cfg['block'] == self.stage_block
            ch = list(cfg['num_channels'])
            setattr(self, f'transition{si - 1}', make_transition(pre, ch, ncfg))
            ms = multiscale_output if si == 4 else True
            setattr(self, f'stage{si}', self._make_stage(cfg, ncfg, lcfg, ms))
            pre = ch
        # modality stems + stage A (hrfuser_hrformer_based.py:375-412)
        self.conv_a = nn.ModuleList(nn.Conv2d(mod_in_channels[k], 64, 3, 2, 1, bias=False) for k in range(M))
        self.norm_a = nn.ModuleList(make_bn(ncfg, 64) for _ in range(M))
        self.conv_b = nn.ModuleList(nn.Conv2d(64, 64, 3, 2, 1, bias=False) for _ in range(M))
        self.norm_b = nn.ModuleList(make_bn(ncfg, 64) for _ in range(M))
        sa = extra['LidarStageA']
        self.layer_a = nn.ModuleList(make_bottleneck_layer(64, sa['num_channels'][0], sa['num_blocks'][0], ncfg)
                                     for _ in range(M))
        pre_m = [sa['num_channels'][0] * 4]
        for tag, nxt in (('a', 'B'), ('b', 'C'), ('c', None)):
            fcfg = extra[f'ModFusion{tag.upper()}']
            if fcfg['block'] not in ('CA', 'MWCA'):
                raise Exception('Not valid fusion block')
            ch = list(fcfg['num_channels'])
            setattr(self, f'transition_{tag}', nn.ModuleList(make_transition(pre_m, ch, ncfg) for _ in range(M)))
            setattr(self, f'fusion_{tag}', nn.ModuleList(
                HRFuserFusionBlock(ch[i], fcfg['num_heads'][i], fcfg['mlp_ratios'][i], ncfg, lcfg,
                                   fcfg['drop_path'], M, fcfg['proj_drop_rate'])
                for i in range(fcfg['num_branches'])))
            if nxt is not None:
                scfg = extra[f'LidarStage{nxt}']
                setattr(self, f'stage_{nxt.lower()}', nn.ModuleList(
                    self._make_stage(scfg, ncfg, lcfg, True) for _ in range(M)))
                pre_m = list(scfg['num_channels'])
        if self.pre_neck_fusion:                                    # hrfuser_hrformer_based.py:454-468
            scfg = extra['LidarStageD']
            self.stage_d = nn.ModuleList(self._make_stage(scfg, ncfg, lcfg, True) for _ in range(M))
            pre_m = list(scfg['num_channels'])
            fcfg = extra['ModFusionD']
            if fcfg['block'] not in ('CA', 'MWCA'):
                raise Exception('Not valid fusion block')
            ch = list(fcfg['num_channels'])
            self.transition_d = nn.ModuleList(make_transition(pre_m, ch, ncfg) for _ in range(M))
            self.fusion_d = nn.ModuleList(
                HRFuserFusionBlock(ch[i], fcfg['num_heads'][i], fcfg['mlp_ratios'][i], ncfg, lcfg,
                                   fcfg['drop_path'], M, fcfg['proj_drop_rate'])
                for i in range(fcfg['num_branches']))

    @staticmethod
    def _make_stage(cfg, ncfg, lcfg, multiscale_output):
        n = cfg['num_modules']
        return nn.Sequential(*[
            HRFormerModule(list(cfg['num_channels']), cfg['num_blocks'], cfg['num_heads'],
                           cfg['mlp_ratios'], ncfg, lcfg,
                           multiscale_output or m != n - 1) for m in range(n)])

    def train(self, mode=True):
        super().train(mode)
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()
        return self

    def forward(self, x, x_mod):
        if self.M != len(x_mod):
            raise Exception('num_fused_modalities does not fit the given input length')
        x = F.relu(getattr(self, self._stem_nn[0])(self.conv1(x)))
        x = self.layer1(F.relu(getattr(self, self._stem_nn[1])(self.conv2(x))))
        mods = []
        for k in range(self.M):
            m = F.relu(self.norm_a[k](self.conv_a[k](x_mod[k])))
            mods.append(self.layer_a[k](F.relu(self.norm_b[k](self.conv_b[k](m)))))

        def fuse_stage(cam_in, trans_cam, trans_mod, fusion, nb, first):
            xs, m0 = [], None
            for i in range(nb):
                if first:
                    # quirk (hrfuser_hrformer_based.py:550-551): transition1[i][0] takes only the
                    # FIRST child: branch 0 -> bare conv (no BN/ReLU); branch 1 -> conv+BN+ReLU.
                    cam = trans_cam[i][0](cam_in)
                elif trans_cam[i] is not None:
                    cam = trans_cam[i](cam_in[-1])
                else:
                    cam = cam_in[i]
                ms = [trans_mod[k][i](mods[k]) if trans_mod[k][i] is not None else mods[k]
                      for k in range(self.M)]
                if i == 0:
                    m0 = ms
                xs.append(fusion[i](cam, ms))
            return xs, m0

        xs, m0 = fuse_stage(x, self.transition1, self.transition_a, self.fusion_a,
                            self.extra['stage2']['num_branches'], True)
        ys = self._run(self.stage2, xs)
        mods = [self._run(self.stage_b[k], [m0[k]])[0] for k in range(self.M)]
        xs, m0 = fuse_stage(ys, self.transition2, self.transition_b, self.fusion_b,
                            self.extra['stage3']['num_branches'], False)
        ys = self._run(self.stage3, xs)
        mods = [self._run(self.stage_c[k], [m0[k]])[0] for k in range(self.M)]
        xs, m0 = fuse_stage(ys, self.transition3, self.transition_c, self.fusion_c,
                            self.extra['stage4']['num_branches'], False)
        ys = self._run(self.stage4, xs)
        if self.pre_neck_fusion:                                    # hrfuser_hrformer_based.py:609-625
            mods = [self._run(self.stage_d[k], [m0[k]])[0] for k in range(self.M)]
            outs = []
            for i in range(self.extra['stage4']['num_branches']):
                ms = [self.transition_d[k][i](mods[k]) if self.transition_d[k][i] is not None else mods[k]
                      for k in range(self.M)]
                outs.append(F.relu(self.fusion_d[i](ys[i], ms)))
            ys = outs
        return ys

    @staticmethod
    def _run(stage, xs):
        for mod in stage:
            xs = mod(xs)
        return xs


class BasicBlock(nn.Module):
    """resnet.py:14-97 (stride 1, no downsample inside an HRModule branch): conv3x3-BN-ReLU-conv3x3-BN, + x, ReLU."""
    expansion = 1

    def __init__(self, cin, planes, norm_cfg, downsample=None):
        super().__init__()
        self._nn = [norm_name(norm_cfg, k) for k in (1, 2)]
        self.conv1 = nn.Conv2d(cin, planes, 3, 1, 1, bias=False)
        self.add_module(self._nn[0], make_bn(norm_cfg, planes))
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.add_module(self._nn[1], make_bn(norm_cfg, planes))
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = F.relu(getattr(self, self._nn[0])(self.conv1(x)))
        return F.relu(getattr(self, self._nn[1])(self.conv2(y)) + idt)


class HRNetModule(nn.Module):
    """The convolutional HRModule (hrnet.py:14-207): parallel branches of BasicBlocks, cross-resolution exchange with
    1x1 conv + BN + NEAREST up-sampling (then F.interpolate(bilinear) to the target size, the identity when the sizes
    nest) and chains of 3x3 stride-2 conv + BN (+ReLU except last)."""

    def __init__(self, channels, num_blocks, norm_cfg, multiscale_output=True):
        super().__init__()
        nb = len(channels)
        self.nb = nb
        self.branches = nn.ModuleList(
            nn.Sequential(*[BasicBlock(channels[i], channels[i], norm_cfg) for _ in range(num_blocks[i])]) for i in range(nb))
        self.fuse_layers = None
        if nb > 1:
            rows = []
            for i in range(nb if multiscale_output else 1):
                row = []
                for j in range(nb):
                    if j > i:
                        row.append(nn.Sequential(nn.Conv2d(channels[j], channels[i], 1, bias=False), make_bn(norm_cfg, channels[i]),
                                                 nn.Upsample(scale_factor=2 ** (j - i), mode='nearest')))
                    elif j == i:
                        row.append(None)
                    else:
                        steps = []
                        for k in range(i - j):
                            last = k == i - j - 1
                            cout = channels[i] if last else channels[j]
                            mods = [nn.Conv2d(channels[j], cout, 3, 2, 1, bias=False), make_bn(norm_cfg, cout)]
                            if not last:
                                mods.append(nn.ReLU(inplace=False))
                            steps.append(nn.Sequential(*mods))
                        row.append(nn.Sequential(*steps))
                rows.append(nn.ModuleList(row))
            self.fuse_layers = nn.ModuleList(rows)

    def forward(self, xs):
        if self.nb == 1:
            return [self.branches[0](xs[0])]
        xs = [self.branches[i](xs[i]) for i in range(self.nb)]
        outs = []
        for i, row in enumerate(self.fuse_layers):
            acc = 0
            for j in range(self.nb):
                if j == i:
                    acc = acc + xs[j]
                elif j > i:
                    acc = acc + F.interpolate(row[j](xs[j]), size=xs[i].shape[2:], mode='bilinear', align_corners=False)
                else:
                    acc = acc + row[j](xs[j])
            outs.append(F.relu(acc))
        return outs


class HRFuserHRNetOracle(HRFuserOracle):
    """Restatement of HRFuserHRNetBased (hrfuser_hrnet_based.py:23-315): the fusion dataflow of HRFuserOracle over a purely
    convolutional HRNet trunk (stage blocks 'BASIC': hrnet.py:512-550 builds HRModules of BasicBlocks) - camera stages
    and modality stages alike.  Same ctor kwargs and state-dict keys as the reference class."""
    stage_block = 'BASIC'

    @staticmethod
    def _make_stage(cfg, ncfg, lcfg, multiscale_output):
        n = cfg['num_modules']
        return nn.Sequential(*[HRNetModule(list(cfg['num_channels']), cfg['num_blocks'], ncfg, multiscale_output or m != n - 1)
                               for m in range(n)])


class HRFormerOracle(nn.Module):
    """Restatement of the plain camera-only HRFormer (hrformer.py:565-740 over HRNet, hrnet.py:211-596): same ctor
    kwargs and state-dict keys.  Differences from the HRFuser camera stream: block key 'HRFORMERBLOCK', the
    stochastic-depth schedule IS applied (linspace over the blocks of stages 2-4, hrformer.py:666-678) and
    transition1[i] is applied whole (hrnet.py:563-566; the `[0]` quirk is HRFuser's)."""

    def __init__(self, extra, in_channels=3, conv_cfg=None, norm_cfg=dict(type='BN', requires_grad=True),
                 transformer_norm_cfg=dict(type='LN', eps=1e-6), norm_eval=False, with_cp=False,
                 multiscale_output=True, drop_path_rate=0., zero_init_residual=False, pretrained=None, init_cfg=None):
        super().__init__()
        self.norm_eval = norm_eval
        self.extra = extra
        ncfg, lcfg = norm_cfg, transformer_norm_cfg
        depths = [extra[s]['num_blocks'][0] * extra[s]['num_modules'] for s in ('stage2', 'stage3', 'stage4')]
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        cuts = [0, depths[0], depths[0] + depths[1], sum(depths)]
        self.conv1 = nn.Conv2d(in_channels, 64, 3, 2, 1, bias=False)
        self._stem_nn = [norm_name(ncfg, 1), norm_name(ncfg, 2)]
        self.add_module(self._stem_nn[0], make_bn(ncfg, 64))
        self.conv2 = nn.Conv2d(64, 64, 3, 2, 1, bias=False)
        self.add_module(self._stem_nn[1], make_bn(ncfg, 64))
        s1 = extra['stage1']
        assert s1['block'] == 'BOTTLENECK'
        self.layer1 = make_bottleneck_layer(64, s1['num_channels'][0], s1['num_blocks'][0], ncfg)
        pre = [s1['num_channels'][0] * 4]
        for k, si in enumerate((2, 3, 4)):
            cfg = extra[f'stage{si}']
            assert cfg['block'] == 'HRFORMERBLOCK'
            ch = list(cfg['num_channels'])
            setattr(self, f'transition{si - 1}', make_transition(pre, ch, ncfg))
            rates = dpr[cuts[k]:cuts[k + 1]]
            n, nb0 = cfg['num_modules'], cfg['num_blocks'][0]
            ms = multiscale_output if si == 4 else True
            setattr(self, f'stage{si}', nn.Sequential(*[
                HRFormerModule(ch, cfg['num_blocks'], cfg['num_heads'], cfg['mlp_ratios'], ncfg, lcfg,
                               ms or m != n - 1, drop_paths=rates[nb0 * m:nb0 * (m + 1)]) for m in range(n)]))
            pre = ch

    def train(self, mode=True):
        super().train(mode)
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()
        return self

    def forward(self, x):                                                   # hrnet.py:552-586
        x = F.relu(getattr(self, self._stem_nn[0])(self.conv1(x)))
        x = self.layer1(F.relu(getattr(self, self._stem_nn[1])(self.conv2(x))))
        xs = [t(x) if t is not None else x for t in self.transition1]
        ys = HRFuserOracle._run(self.stage2, xs)
        xs = [t(ys[-1]) if t is not None else ys[i] for i, t in enumerate(self.transition2)]
        ys = HRFuserOracle._run(self.stage3, xs)
        xs = [t(ys[-1]) if t is not None else ys[i] for i, t in enumerate(self.transition3)]
        return HRFuserOracle._run(self.stage4, xs)


# ----------------------------------------------------------------------------- shared test utils
def seeded_fill_(module, seed=0):
    """Deterministically randomise EVERY parameter and BN running stat from a CPU generator.

    Iterates the state dict in sorted-key order so reference, oracle and product (identical
    key sets) receive bit-identical values regardless of construction order.  RPB tables and
    running stats are randomised too, otherwise those paths would be untested.
    """
    g = torch.Generator(device='cpu').manual_seed(seed)
    sd = module.state_dict()
    with torch.no_grad():
        for key in sorted(sd.keys()):
            t = sd[key]
            if key.endswith('num_batches_tracked') or key.endswith('relative_position_index'):
                continue
            r = torch.randn(t.shape, generator=g, dtype=torch.float32)
            if key.endswith('running_var'):
                v = 0.5 + r.abs()
            elif key.endswith('running_mean'):
                v = 0.2 * r
            elif key.endswith('relative_position_bias_table'):
                v = 0.3 * r
            elif key.endswith('weight') and t.ndim == 1:          # BN / LN gamma
                v = 1.0 + 0.2 * r
            elif key.endswith('bias'):
                v = 0.2 * r
            elif t.ndim >= 2:                                      # conv / linear weight
                fan_in = t[0].numel()
                v = r * (1.5 / math.sqrt(fan_in))
            else:
                raise KeyError(key)
            t.copy_(v.to(t.dtype))
    return module


def seeded_inputs(B, H, W, mod_channels=(3, 3), seed=1, dtype=torch.float32):
    g = torch.Generator(device='cpu').manual_seed(seed)
    x = torch.randn(B, 3, H, W, generator=g).to(dtype)
    mods = [torch.randn(B, c, H, W, generator=g).to(dtype) for c in mod_channels]
    return x, mods
