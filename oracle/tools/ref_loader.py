"""Container-only tooling: import the reference HRFuser backbone from /root/reference.

TEST INFRASTRUCTURE ONLY.  Never imported by the product package (`hrfuser_amd`), by
`bench.py`'s timed path, or on the GPU box (`/root/reference` does not exist there).  It is
used by `oracle/tools/make_golden.py` to (1) validate the torch restatement in
`oracle/hrfuser_oracle.py` against the real reference and (2) produce the committed golden
vectors under `tests/golden/`.

The reference is a fork of MMDetection and needs `mmcv-full==1.3.17` (reference
README.md:41), which is absent from this image.  mmcv contributes only *factories*
(`build_conv_layer -> nn.Conv2d`, `build_norm_layer -> nn.BatchNorm2d / nn.LayerNorm`,
`build_activation_layer -> nn.GELU`, `BaseModule`, `Registry`) plus `DropPath`; all
arithmetic lives in PyTorch ATen.  The stand-in below provides exactly those
factory entry points so the four reference backbone files import *unmodified*; DropPath and
Dropout are disabled (eval / p=0) in every fixture so no stand-in arithmetic is ever on the
measured path.
"""
import ast
import copy
import importlib
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get('HRF_REFERENCE_ROOT', '/root/reference')


# ----------------------------------------------------------------------------- mmcv stand-in
class _Registry:
    def __init__(self, name, parent=None, **kw):
        self.name, self._mods = name, {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self._mods[name or cls.__name__] = cls
            return cls
        return deco(module) if module is not None else deco

    def get(self, key):
        return self._mods.get(key)

    def build(self, cfg, default_args=None):
        cfg = dict(cfg)
        for k, v in (default_args or {}).items():
            cfg.setdefault(k, v)
        return self._mods[cfg.pop('type')](**cfg)


class _BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = copy.deepcopy(init_cfg)

    def init_weights(self):
        for m in self.children():
            if hasattr(m, 'init_weights'):
                m.init_weights()


class _Sequential(_BaseModule, nn.Sequential):
    def __init__(self, *args, init_cfg=None):
        _BaseModule.__init__(self, init_cfg)
        nn.Sequential.__init__(self, *args)


class _ModuleList(_BaseModule, nn.ModuleList):
    def __init__(self, modules=None, init_cfg=None):
        _BaseModule.__init__(self, init_cfg)
        nn.ModuleList.__init__(self, modules)


def _build_conv_layer(cfg, *args, **kwargs):
    assert cfg is None or cfg.get('type') in ('Conv2d', 'Conv')
    return nn.Conv2d(*args, **kwargs)


_NORMS = {'BN': (nn.BatchNorm2d, 'bn'), 'SyncBN': (nn.BatchNorm2d, 'bn'),
          'LN': (nn.LayerNorm, 'ln'), 'GN': (nn.GroupNorm, 'gn')}


def _build_norm_layer(cfg, num_features, postfix=''):
    c = dict(cfg)
    kind = c.pop('type')
    requires_grad = c.pop('requires_grad', True)
    c.setdefault('eps', 1e-5)
    cls, abbr = _NORMS[kind]
    if kind == 'GN':
        assert 'num_groups' in c                          # mmcv: layer = norm_layer(num_channels=num_features, **cfg_)
        layer = cls(num_channels=num_features, **c)
    else:
        layer = cls(num_features, **c)
    for p in layer.parameters():
        p.requires_grad_(requires_grad)
    return abbr + str(postfix), layer


def _build_activation_layer(cfg):
    c = dict(cfg)
    kind = c.pop('type')
    return {'GELU': nn.GELU, 'ReLU': nn.ReLU}[kind](**c)


class _DropPath(nn.Module):
    """mmcv.cnn.bricks.drop.DropPath semantics (per-sample stochastic depth)."""

    def __init__(self, drop_prob=0.1):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep = 1 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        mask = (keep + torch.rand(shape, dtype=x.dtype, device=x.device)).floor()
        return x.div(keep) * mask


def _build_dropout(cfg):
    c = dict(cfg)
    assert c.pop('type') == 'DropPath'
    return _DropPath(**c)


def _trunc_normal_init(module, mean=0., std=1., a=-2., b=2., bias=0.):
    # mmcv's helper only touches `.weight` / `.bias`; on a bare nn.Parameter it is a no-op,
    # which is why RPB tables stay zero at init in the reference.
    if getattr(module, 'weight', None) is not None:
        nn.init.trunc_normal_(module.weight, mean, std, a, b)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def _constant_init(module, val, bias=0.):
    if getattr(module, 'weight', None) is not None:
        nn.init.constant_(module.weight, val)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


class _ConvModule(nn.Module):
    """mmcv.cnn.ConvModule reduced to what the reference HRFPN uses (norm_cfg=None, act_cfg=None):
    a biased nn.Conv2d under the attribute name `conv` (mmcv: bias='auto' -> True without a norm)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, conv_cfg=None, norm_cfg=None,
                 act_cfg=dict(type='ReLU'), **kw):
        super().__init__()
        assert norm_cfg is None and act_cfg is None and not kw, 'stand-in covers the HRFPN use only'
        self.conv = _build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size, stride=stride,
                                      padding=padding, bias=True)

    def forward(self, x):
        return self.conv(x)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_INSTALLED = False


def install():
    """Install the mmcv stand-in + empty mmdet package skeletons, then import the backbone."""
    global _INSTALLED
    if _INSTALLED:
        return sys.modules['mmdet.models.backbones.hrfuser_hrformer_based']
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError(f'reference tree {REF_ROOT} not present (container-only tool)')
    from torch.nn.modules.batchnorm import _BatchNorm

    _mod('mmcv', __version__='1.3.17')
    _mod('mmcv.cnn', build_conv_layer=_build_conv_layer, build_norm_layer=_build_norm_layer,
         build_activation_layer=_build_activation_layer, build_plugin_layer=None,
         constant_init=_constant_init, trunc_normal_init=_trunc_normal_init,
         kaiming_init=None, MODELS=_Registry('model'), ConvModule=_ConvModule)
    _mod('mmcv.cnn.bricks')
    _mod('mmcv.cnn.bricks.transformer', build_dropout=_build_dropout)
    _mod('mmcv.runner', BaseModule=_BaseModule, Sequential=_Sequential,
         ModuleList=_ModuleList, _load_checkpoint=None)
    _mod('mmcv.utils', Registry=_Registry, get_logger=lambda *a, **k: None)
    _mod('mmcv.utils.parrots_wrapper', _BatchNorm=_BatchNorm)

    # empty package skeletons whose __path__ points at the reference dirs, so the real
    # __init__.py files (which assert the mmcv version / import every head) never run
    for name, rel in (('mmdet', 'mmdet'), ('mmdet.models', 'mmdet/models'),
                      ('mmdet.models.backbones', 'mmdet/models/backbones')):
        pkg = _mod(name)
        pkg.__path__ = [os.path.join(REF_ROOT, rel)]
    _mod('mmdet.utils', get_root_logger=lambda *a, **k: None)
    importlib.import_module('mmdet.models.builder')

    # mmdet.models.utils cannot be imported (needs mmcv.ops); lift the three layout helpers
    # straight out of the reference source with `ast` and import the real res_layer.
    src_path = os.path.join(REF_ROOT, 'mmdet/models/utils/transformer.py')
    with open(src_path) as fh:
        tree = ast.parse(fh.read())
    wanted = {'nlc_to_nchw', 'nchw_to_nlc', 'nlc2nchw2nlc'}
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    assert {n.name for n in body} == wanted
    ns = {'torch': torch}
    exec(compile(ast.Module(body=body, type_ignores=[]), src_path, 'exec'), ns)
    utils = _mod('mmdet.models.utils', **{k: ns[k] for k in wanted})
    utils.__path__ = [os.path.join(REF_ROOT, 'mmdet/models/utils')]
    res_layer = importlib.import_module('mmdet.models.utils.res_layer')
    utils.ResLayer = res_layer.ResLayer

    mod = importlib.import_module('mmdet.models.backbones.hrfuser_hrformer_based')
    _INSTALLED = True
    return mod


# ----------------------------------------------------------------------------- config loading
def _merge(base, upd):
    """mmcv.Config dict merge: recursive, `_delete_=True` replaces the base value."""
    out = dict(base)
    for k, v in upd.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get('_delete_', False):
            out[k] = _merge(out[k], v)
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != '_delete_'}
            out[k] = v
    return out


def load_cfg(path):
    """Subset of mmcv.Config.fromfile sufficient for configs/hrfuser/*.py."""
    ns = {}
    with open(path) as fh:
        exec(compile(fh.read(), path, 'exec'), ns)
    cfg = {k: v for k, v in ns.items() if not k.startswith('__') and not isinstance(v, types.ModuleType)}
    bases = cfg.pop('_base_', [])
    if isinstance(bases, str):
        bases = [bases]
    merged = {}
    for b in bases:
        merged = _merge(merged, load_cfg(os.path.join(os.path.dirname(path), b)))
    return _merge(merged, cfg)


def backbone_cfg(cfg_name):
    """Resolved `model.backbone` dict (with `type`) of configs/hrfuser/<cfg_name>.py."""
    path = os.path.join(REF_ROOT, 'configs/hrfuser', cfg_name + '.py')
    return copy.deepcopy(load_cfg(path)['model']['backbone'])


def build_reference(cfg_name_or_dict):
    """Instantiate the reference HRFuserHRFormerBased from a config name or backbone dict."""
    install()
    from mmdet.models.builder import BACKBONES
    cfg = backbone_cfg(cfg_name_or_dict) if isinstance(cfg_name_or_dict, str) \
        else copy.deepcopy(cfg_name_or_dict)
    return BACKBONES.build(cfg)


def build_reference_hrfpn(**kw):
    """The unmodified reference neck (mmdet/models/necks/hrfpn.py) on the mmcv stand-in."""
    install()
    if 'mmdet.models.necks' not in sys.modules:
        pkg = _mod('mmdet.models.necks')
        pkg.__path__ = [os.path.join(REF_ROOT, 'mmdet/models/necks')]
    mod = importlib.import_module('mmdet.models.necks.hrfpn')
    return mod.HRFPN(**kw)
