"""String -> class registries and `build_from_cfg`, the plugin mechanism of the reference
(mmcv/utils/registry.py:8-167): configs name components by `type`, the rest are kwargs."""
import inspect


class Registry:

    def __init__(self, name):
        self._name = name
        self._table = {}

    name = property(lambda self: self._name)
    module_dict = property(lambda self: self._table)

    def __len__(self):
        return len(self._table)

    def __contains__(self, key):
        return key in self._table

    def __repr__(self):
        return f'Registry(name={self._name}, items={sorted(self._table)})'

    def get(self, key):
        return self._table.get(key)

    def _add(self, cls, name, force):
        if not inspect.isclass(cls):
            raise TypeError(f'module must be a class, but got {type(cls)}')
        key = name or cls.__name__
        if key in self._table and not force:
            raise KeyError(f'{key} is already registered in {self._name}')
        self._table[key] = cls

    def register_module(self, name=None, force=False, module=None):
        """Decorator (`@R.register_module()`, `@R.register_module(name='x')`, or the legacy bare
        `@R.register_module`) or plain call (`R.register_module(module=Cls)`)."""
        if inspect.isclass(name):  # legacy: used without parentheses
            self._add(name, None, force)
            return name
        if not isinstance(force, bool):
            raise TypeError(f'force must be a boolean, but got {type(force)}')
        if name is not None and not isinstance(name, str):
            raise TypeError(f'name must be a str, but got {type(name)}')
        if module is not None:
            self._add(module, name, force)
            return module

        def deco(cls):
            self._add(cls, name, force)
            return cls

        return deco


def build_from_cfg(cfg, registry, default_args=None):
    """Instantiate `registry[cfg['type']](**rest_of_cfg, **default_args)`."""
    if not isinstance(cfg, dict):
        raise TypeError(f'cfg must be a dict, but got {type(cfg)}')
    if 'type' not in cfg:
        raise KeyError(f'the cfg dict must contain the key "type", but got {cfg}')
    if not isinstance(registry, Registry):
        raise TypeError(f'registry must be a Registry object, but got {type(registry)}')
    if not (default_args is None or isinstance(default_args, dict)):
        raise TypeError(f'default_args must be a dict or None, but got {type(default_args)}')
    args = dict(cfg)
    kind = args.pop('type')
    if isinstance(kind, str):
        cls = registry.get(kind)
        if cls is None:
            raise KeyError(f'{kind} is not in the {registry.name} registry')
    elif inspect.isclass(kind):
        cls = kind
    else:
        raise TypeError(f'type must be a str or valid type, but got {type(kind)}')
    for k, v in (default_args or {}).items():
        args.setdefault(k, v)
    return cls(**args)
