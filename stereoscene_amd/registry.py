"""Minimal registry + python-config loader so that the reference's config file
(projects/configs/occupancy/semantickitti/stereoscene.py) loads unchanged without mmcv.

Mirrors the parts of mmcv.utils.Registry / mmcv.Config the hot path relies on
(SURVEY.md section 8(b) "Registry/plugin API"): ``@X.register_module()``, ``build(cfg)``
with ``dict(type=...)``, ``_base_`` inheritance, ``plugin`` / ``plugin_dir``.
"""
import importlib
import os
import runpy


class Registry:
    def __init__(self, name):
        self.name = name
        self._modules = {}

    def register_module(self, name=None, force=False, module=None):
        def _register(cls):
            key = name or cls.__name__
            if key in self._modules and not force:
                raise KeyError(f"{key} is already registered in {self.name}")
            self._modules[key] = cls
            return cls
        if module is not None:
            return _register(module)
        return _register

    def get(self, key):
        return self._modules.get(key)

    def __contains__(self, key):
        return key in self._modules

    def build(self, cfg, **default_args):
        if not isinstance(cfg, dict) or "type" not in cfg:
            raise TypeError(f"{self.name}: cfg must be a dict with a 'type' key, got {cfg!r}")
        args = dict(cfg)
        t = args.pop("type")
        cls = t if isinstance(t, type) else self.get(t)
        if cls is None:
            raise KeyError(f"'{t}' is not in the {self.name} registry")
        for k, v in default_args.items():
            args.setdefault(k, v)
        return cls(**args)


BACKBONES = Registry("backbone")
NECKS = Registry("neck")
HEADS = Registry("head")
DETECTORS = Registry("detector")


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_model(cfg, train_cfg=None, test_cfg=None):
    return DETECTORS.build(cfg)


class ConfigDict(dict):
    """dict with attribute access (what mmcv.Config hands to builders)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(v):
    if isinstance(v, dict):
        return ConfigDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, (list, tuple)):
        return type(v)(_wrap(x) for x in v)
    return v


def _merge(base, child):
    out = dict(base)
    for k, v in child.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.pop("_delete_", False):
            out[k] = _merge(out[k], v)
        else:
            out[k] = v
    return out


class Config:
    """``Config.fromfile(path)``: execute a python config, merge its ``_base_`` files (missing
    dataset bases are tolerated: the data layer is out of scope), expose keys as attributes."""

    def __init__(self, d, filename=None):
        self._cfg = _wrap(d)
        self.filename = filename

    @staticmethod
    def _load(path, strict_bases=False):
        ns = runpy.run_path(path)
        cfg = {k: v for k, v in ns.items() if not k.startswith("__") and not callable(v)
               and not isinstance(v, type(os))}
        bases = cfg.pop("_base_", [])
        if isinstance(bases, str):
            bases = [bases]
        merged = {}
        for b in bases:
            bp = os.path.normpath(os.path.join(os.path.dirname(path), b))
            if os.path.exists(bp):
                merged = _merge(merged, Config._load(bp, strict_bases))
            elif strict_bases:
                raise FileNotFoundError(bp)
        return _merge(merged, cfg)

    @classmethod
    def fromfile(cls, path, import_plugin=True):
        cfg = cls(cls._load(path), filename=path)
        if import_plugin and cfg.get("plugin", False):
            # the reference imports `plugin_dir` as a module path; here every registry-visible
            # class of the hot path lives in stereoscene_amd.plugin
            importlib.import_module("stereoscene_amd.plugin")
        return cfg

    def get(self, k, default=None):
        return self._cfg.get(k, default)

    def __getattr__(self, k):
        return getattr(self._cfg, k)

    def __getitem__(self, k):
        return self._cfg[k]

    def __contains__(self, k):
        return k in self._cfg
