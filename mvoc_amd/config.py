"""Minimal stand-in for the subset of OmegaConf the reference's drivers use (``i2vgen-xl/inverse.py:231-236,143``,
``composite.py:228-236,94``): ``OmegaConf.load`` of a YAML template, ``OmegaConf.create`` of a JSON entry,
``OmegaConf.merge`` (deep, later wins), ``${a.b}`` interpolation resolved lazily against the root on attribute
access, attribute / item access, assignment, ``OmegaConf.to_yaml``.  omegaconf itself is not installed in the
build or GPU environment; PyYAML is."""
import re

import yaml

_INTERP = re.compile(r"\$\{([^}]+)\}")


class Config:
    def __init__(self, data=None, root=None):
        object.__setattr__(self, "_data", {})
        object.__setattr__(self, "_root", root if root is not None else self)
        for k, v in (data or {}).items():
            self._data[k] = self._wrap(v)

    def _wrap(self, v):
        if isinstance(v, Config):
            return Config(v.to_container(resolve=False), self._root)
        if isinstance(v, dict):
            return Config(v, self._root)
        if isinstance(v, (list, tuple)):
            return [self._wrap(x) for x in v]
        return v

    def _rebind(self, root):
        object.__setattr__(self, "_root", root)
        for v in self._data.values():
            for x in (v if isinstance(v, list) else [v]):
                if isinstance(x, Config):
                    x._rebind(root)

    def _lookup(self, path):
        node = self._root
        for part in path.split("."):
            node = node[part]
        return node

    def _resolve(self, v):
        if isinstance(v, str):
            m = _INTERP.fullmatch(v)
            if m:  # whole-value reference keeps the referenced type (e.g. image_size: ${image_size})
                return self._resolve(self._lookup(m.group(1).strip()))
            return _INTERP.sub(lambda mm: str(self._resolve(self._lookup(mm.group(1).strip()))), v)
        if isinstance(v, list):
            return [self._resolve(x) for x in v]
        return v

    def __getattr__(self, k):
        try:
            return self._resolve(self._data[k])
        except KeyError:
            raise AttributeError(k) from None

    __getitem__ = __getattr__

    def __setattr__(self, k, v):
        self._data[k] = self._wrap(v)

    __setitem__ = __setattr__

    def __contains__(self, k):
        return k in self._data

    def keys(self):
        return self._data.keys()

    def items(self):
        return [(k, getattr(self, k)) for k in self._data]

    def get(self, k, default=None):
        return getattr(self, k) if k in self._data else default

    def to_container(self, resolve=True):
        def conv(v):
            if isinstance(v, Config):
                return v.to_container(resolve)
            if isinstance(v, list):
                return [conv(x) for x in v]
            return self._resolve(v) if resolve else v
        return {k: conv(v) for k, v in self._data.items()}

    def __repr__(self):
        return f"Config({self.to_container(resolve=False)!r})"


class OmegaConf:
    @staticmethod
    def load(path):
        with open(path) as f:
            return Config(yaml.safe_load(f) or {})

    @staticmethod
    def create(obj=None):
        return Config(obj or {})

    @staticmethod
    def merge(*cfgs):
        def deep(a, b):
            out = dict(a)
            for k, v in b.items():
                out[k] = deep(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else v
            return out
        acc = {}
        for c in cfgs:
            acc = deep(acc, c.to_container(resolve=False) if isinstance(c, Config) else dict(c))
        return Config(acc)

    @staticmethod
    def to_yaml(cfg, resolve=False):
        return yaml.safe_dump(cfg.to_container(resolve=resolve), sort_keys=False)

    @staticmethod
    def to_container(cfg, resolve=False):
        return cfg.to_container(resolve=resolve)
