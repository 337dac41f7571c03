"""Registries with the reference's names (mmdet registries re-exported by
mmdet3d/models/builder.py:1-52): configs written for the reference (`dict(type='ImVoxelNet', ...)`,
`type='KittiImVoxelNeck'`, ...) build the MI355X modules unchanged.  When mmdet is importable its own
registries could be aliased here; it is absent from this image, so a minimal registry is built in.
"""


class Registry:
    def __init__(self, name):
        self.name = name
        self._modules = {}

    def register_module(self, name=None, force=False, module=None):
        def _reg(cls):
            key = name or cls.__name__
            if key in self._modules and not force:
                raise KeyError(f'{key} is already registered in {self.name}')
            self._modules[key] = cls
            return cls
        if module is not None:
            return _reg(module)
        return _reg

    def get(self, key):
        return self._modules.get(key)

    def __contains__(self, key):
        return key in self._modules

    def build(self, cfg, default_args=None):
        if cfg is None:
            return None
        if not isinstance(cfg, dict) or 'type' not in cfg:
            raise TypeError(f'cfg must be a dict with a "type" key, got {cfg!r}')
        args = dict(cfg)
        typ = args.pop('type')
        cls = self._modules.get(typ) if isinstance(typ, str) else typ
        if cls is None:
            raise KeyError(f'{typ} is not in the {self.name} registry')
        for k, v in (default_args or {}).items():
            args.setdefault(k, v)
        return cls(**args)


BACKBONES = Registry('backbone')
NECKS = Registry('neck')
HEADS = Registry('head')
DETECTORS = Registry('detector')
ANCHOR_GENERATORS = Registry('anchor generator')
BBOX_CODERS = Registry('bbox coder')


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    return DETECTORS.build(cfg, dict(train_cfg=train_cfg, test_cfg=test_cfg))


def build_anchor_generator(cfg):
    return ANCHOR_GENERATORS.build(cfg)


def build_bbox_coder(cfg):
    return BBOX_CODERS.build(cfg)


class ConfigDict(dict):
    """dict with attribute access (the slice of mmcv.ConfigDict the path relies on: cfg.max_num, cfg.get)."""
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v
