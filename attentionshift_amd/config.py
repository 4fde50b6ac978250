"""Loader for the reference's mmcv-style python configs (configs/mae/*.py, configs/_base_/*.py):
`_base_` inheritance, `_delete_=True`, dotted `--cfg-options a.b=c` overrides (tools/train.py:54-64)."""
import ast
import copy
import os


class ConfigDict(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value


def _wrap(obj):
    if isinstance(obj, dict):
        return ConfigDict({k: _wrap(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return type(obj)(_wrap(v) for v in obj)
    return obj


def _merge(base, child):
    out = copy.deepcopy(base)
    for k, v in child.items():
        if isinstance(v, dict) and k in out and isinstance(out[k], dict) and not v.get("_delete_", False):
            out[k] = _merge(out[k], v)
        else:
            v = copy.deepcopy(v)
            if isinstance(v, dict):
                v.pop("_delete_", None)
            out[k] = v
    return out


def _load_file(path):
    src = open(path).read()
    ast.parse(src)                      # syntax check with a clear error
    scope = {}
    exec(compile(src, path, "exec"), scope)
    cfg = {k: v for k, v in scope.items() if not k.startswith("__") and not callable(v) and not hasattr(v, "__loader__")}
    bases = cfg.pop("_base_", [])
    if isinstance(bases, str):
        bases = [bases]
    merged = {}
    for b in bases:
        bp = os.path.join(os.path.dirname(path), b)
        if not os.path.exists(bp):
            raise FileNotFoundError(f"{path}: base config {b} not found")
        merged = _merge(merged, _load_file(bp))
    return _merge(merged, cfg)


class Config:
    def __init__(self, cfg_dict, filename=None):
        object.__setattr__(self, "_cfg", _wrap(cfg_dict))
        object.__setattr__(self, "filename", filename)

    @staticmethod
    def fromfile(path):
        return Config(_load_file(os.path.abspath(path)), path)

    def merge_from_dict(self, options):
        for key, value in options.items():
            d = self._cfg
            parts = key.split(".")
            for p in parts[:-1]:
                d = d.setdefault(p, ConfigDict())
            d[parts[-1]] = _wrap(value)

    def __getattr__(self, name):
        return getattr(self._cfg, name)

    def __getitem__(self, name):
        return self._cfg[name]

    def get(self, *a):
        return self._cfg.get(*a)

    def to_dict(self):
        return copy.deepcopy(dict(self._cfg))
