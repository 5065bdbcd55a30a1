"""`BACKBONES.register_module()` surface (mmdet/models/builder.py:7-20).

If a real mmdet/mmcv installation is importable the class registers into ITS `BACKBONES`
registry (force-replacing the PyTorch implementation of the same name), so the reference's
`configs/hrfuser/*.py` build the HIP backbone unmodified through `build_backbone(cfg)`.
Otherwise (this image has no mmcv) an API-compatible minimal registry is used.
"""


class Registry:
    """mmcv.utils.Registry subset: register_module() decorator, get(), build(cfg)."""

    def __init__(self, name):
        self.name = name
        self._module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def _register(cls):
            key = name or cls.__name__
            if key in self._module_dict and not force:
                raise KeyError(f'{key} is already registered in {self.name}')
            self._module_dict[key] = cls
            return cls
        return _register(module) if module is not None else _register

    def get(self, key):
        return self._module_dict.get(key)

    def build(self, cfg, default_args=None):
        if not isinstance(cfg, dict) or 'type' not in cfg:
            raise KeyError('cfg must be a dict containing the key "type"')
        args = dict(cfg)
        for k, v in (default_args or {}).items():
            args.setdefault(k, v)
        kind = args.pop('type')
        cls = self.get(kind) if isinstance(kind, str) else kind
        if cls is None:
            raise KeyError(f'{kind} is not in the {self.name} registry')
        return cls(**args)


class _ForceRegistry:
    """Adapter over a real mmcv registry: always registers with force=True (drop-in replacement)."""

    def __init__(self, reg):
        self._reg = reg

    def register_module(self, name=None, force=True, module=None):
        return self._reg.register_module(name=name, force=True, module=module)

    def __getattr__(self, k):
        return getattr(self._reg, k)


def _resolve():
    try:
        from mmdet.models.builder import BACKBONES as real      # noqa: F401
        return _ForceRegistry(real), True
    except Exception:
        return Registry('models'), False


BACKBONES, USING_MMDET = _resolve()
MODELS = BACKBONES
NECKS = MODELS            # mmdet 2.19: BACKBONES = NECKS = ... = MODELS (mmdet/models/builder.py:7-15)


def build_backbone(cfg):
    """mmdet.models.builder.build_backbone (builder.py:18-20)."""
    return BACKBONES.build(cfg)
