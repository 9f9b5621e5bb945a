"""Model registry surface of the reference (projects/registry.py:1-3: ``MODELS = mmcv.utils.Registry('model')``).

mmcv is not a dependency of this repo; this is the minimal counterpart with the two calls the reference uses:
``@MODELS.register_module()`` (UMGen.py:51) and ``build_from_cfg(dict(type=<class or name>, config=...), MODELS)``
(evaluate.py:193).  When mmcv IS importable the drop-in class is additionally registered in a real mmcv Registry
(see INTEGRATION.md).
"""
from __future__ import annotations


class Registry:
    def __init__(self, name: str):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            key = name or cls.__name__
            if key in self.module_dict and not force:
                raise KeyError(f"{key} is already registered in {self.name}")
            self.module_dict[key] = cls
            return cls
        return deco(module) if module is not None else deco

    def get(self, key):
        return self.module_dict.get(key)


def build_from_cfg(cfg: dict, registry: Registry, default_args: dict = None):
    """mmcv.utils.build_from_cfg semantics: ``type`` may be a registered name or the class object itself."""
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    t = args.pop("type")
    cls = registry.get(t) if isinstance(t, str) else t
    if cls is None:
        raise KeyError(f"{t} is not in the {registry.name} registry")
    return cls(**args)


MODELS = Registry("model")
DATASETS = Registry("dataset")
