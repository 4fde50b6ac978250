"""mmcv-compatible registry surface (reference mmdet/models/builder.py:6-12, mmcv.utils.Registry).

The reference hangs every model component off `Registry.register_module()` + `build_from_cfg(dict(type=...))`.
mmcv is not installed on the GPU box, so this is a small API-compatible registry; when mmdet IS importable
`register_into_mmdet()` additionally registers the same classes into mmdet's own BACKBONES / HEADS so the
existing run_train.py pipeline builds them from configs/mae/*.py unchanged.
"""
import inspect


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return key in self._module_dict

    def __repr__(self):
        return f"Registry(name={self._name}, items={sorted(self._module_dict)})"

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def _register(self, cls, name=None, force=False):
        if not inspect.isclass(cls):
            raise TypeError(f"module must be a class, but got {type(cls)}")
        names = [cls.__name__] if name is None else ([name] if isinstance(name, str) else list(name))
        for n in names:
            if not force and n in self._module_dict:
                raise KeyError(f"{n} is already registered in {self._name}")
            self._module_dict[n] = cls

    def register_module(self, name=None, force=False, module=None):
        if not isinstance(force, bool):
            raise TypeError(f"force must be a boolean, but got {type(force)}")
        if module is not None:
            self._register(module, name, force)
            return module

        def _decorate(cls):
            self._register(cls, name, force)
            return cls

        return _decorate

    def build(self, cfg, default_args=None):
        return build_from_cfg(cfg, self, default_args)


def build_from_cfg(cfg, registry, default_args=None):
    """Same contract as mmcv.utils.build_from_cfg: cfg['type'] names (or is) the class."""
    if not isinstance(cfg, dict):
        raise TypeError(f"cfg must be a dict, but got {type(cfg)}")
    if "type" not in cfg and not (default_args and "type" in default_args):
        raise KeyError(f'`cfg` or `default_args` must contain the key "type", but got {cfg}\n{default_args}')
    args = dict(cfg)
    if default_args is not None:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop("type")
    if isinstance(obj_type, str):
        cls = registry.get(obj_type)
        if cls is None:
            raise KeyError(f"{obj_type} is not in the {registry.name} registry")
    elif inspect.isclass(obj_type):
        cls = obj_type
    else:
        raise TypeError(f"type must be a str or valid type, but got {type(obj_type)}")
    try:
        return cls(**args)
    except Exception as e:
        raise type(e)(f"{cls.__name__}: {e}")


BACKBONES = Registry("backbone")
HEADS = Registry("head")
ROI_HEADS = HEADS                      # mmdet 2.11: roi heads live in HEADS


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def register_into_mmdet():
    """Drop-in: register the MI355X classes under the reference's names in mmdet's registries."""
    from mmdet.models.builder import BACKBONES as MM_BACKBONES, HEADS as MM_HEADS  # noqa: raises if absent
    for reg, mine in ((MM_BACKBONES, BACKBONES), (MM_HEADS, HEADS)):
        for name, cls in mine.module_dict.items():
            reg.register_module(name=name, force=True, module=cls)
