"""Registries with the reference's names (mmdet registries re-exported by
mmdet3d/models/builder.py:1-52): configs written for the reference (`dict(type='ImVoxelNet', ...)`,
`type='KittiImVoxelNeck'`, ...) build the MI355X modules unchanged.  mmdet is absent from this image, so a minimal
registry is built in; where mmdet IS importable, register_into_mmdet() -- an explicit call, or IVX_REGISTER_MMDET=1 to have
the package import do it -- puts the same classes into mmdet's own DETECTORS / NECKS / HEADS / BACKBONES /
ANCHOR_GENERATORS / BBOX_CODERS under the same names (force=True, replacing the reference's CUDA-path classes AND mmdet's
generic ResNet / FPN for the process), so `build_detector(cfg.model)` of the reference's tools/test.py builds the MI355X
modules.  Call it after mmdet3d has been imported: mmdet3d's own non-forced registrations would otherwise collide.
"""
import os


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


# our registry -> (module path inside mmdet, attribute) of the registry the reference builds from
_MMDET_REGISTRIES = (
    (DETECTORS, 'mmdet.models', 'DETECTORS'), (NECKS, 'mmdet.models', 'NECKS'), (HEADS, 'mmdet.models', 'HEADS'),
    (BACKBONES, 'mmdet.models', 'BACKBONES'), (ANCHOR_GENERATORS, 'mmdet.core.anchor', 'ANCHOR_GENERATORS'),
    (BBOX_CODERS, 'mmdet.core.bbox.builder', 'BBOX_CODERS'),
)


def register_into_mmdet(force=True):
    """Alias every class registered here into the mmdet registry of the same role (mmdet3d/models/builder.py:1-52 builds
    from those).  Returns {registry name: [class names]} of what was registered; {} when mmdet is not importable."""
    import importlib
    done = {}
    try:
        importlib.import_module('mmdet')
    except Exception:
        return done
    for ours, modname, attr in _MMDET_REGISTRIES:
        try:
            theirs = getattr(importlib.import_module(modname), attr)
        except Exception:
            continue
        for name, cls in ours._modules.items():
            try:
                theirs.register_module(name=name, force=force, module=cls)
            except TypeError:                       # older mmcv: register_module(cls) / _register_module(cls, name, force)
                theirs._register_module(cls, module_name=name, force=force)
            done.setdefault(attr, []).append(name)
    return done


def maybe_register_into_mmdet():
    """Opt-in at import time: IVX_REGISTER_MMDET=1 aliases the classes into mmdet's registries while the package is imported
    (replacing mmdet's own 'ResNet' / 'FPN' / 'Anchor3DHead' for the whole process, and -- if mmdet3d is imported afterwards --
    colliding with its non-forced registration of 'ImVoxelNet').  Without it nothing outside this package is touched; a host
    that wants the aliases calls register_into_mmdet() itself, after importing mmdet3d."""
    if os.environ.get('IVX_REGISTER_MMDET', '0') == '1':
        return register_into_mmdet()
    return {}
