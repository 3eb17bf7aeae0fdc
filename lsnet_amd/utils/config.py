"""Python-file configs with `_base_` inheritance, `_delete_` and dotted-key overrides -- the
behaviour of mmcv.Config (mmcv/utils/config.py:58-388) that `configs/lsnet/*.py` rely on.

Own implementation: config files are executed with `runpy` (no temp-module import), the attribute
dict is a small dict subclass (no addict dependency)."""
import copy
import json
import os.path as osp
import runpy

BASE_KEY = '_base_'
DELETE_KEY = '_delete_'
RESERVED_KEYS = ('filename', 'text', 'pretty_text')


class ConfigDict(dict):
    """dict with attribute access; nested dicts (also inside lists/tuples) are converted on insert."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(i) for i in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(f"'{type(self).__name__}' object has no attribute '{k}'") from None

    def __delattr__(self, k):
        try:
            del self[k]
        except KeyError:
            raise AttributeError(k) from None

    def update(self, *args, **kwargs):
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def setdefault(self, k, default=None):
        if k not in self:
            self[k] = default
        return self[k]

    def copy(self):
        return type(self)(self)

    def __deepcopy__(self, memo):
        return type(self)({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def to_dict(self):
        def plain(v):
            if isinstance(v, dict):
                return {k: plain(x) for k, x in v.items()}
            if isinstance(v, (list, tuple)):
                return type(v)(plain(x) for x in v)
            return v
        return plain(self)


def _merge(child, base):
    """Values of `child` override `base` recursively; a child dict carrying `_delete_=True` replaces
    the base value instead of merging into it (mmcv/utils/config.py:158-175)."""
    out = dict(base)
    for k, v in child.items():
        if isinstance(v, dict) and k in out and not v.pop(DELETE_KEY, False):
            if not isinstance(out[k], dict):
                raise TypeError(f'{k}={v} in child config cannot inherit from base because {k} is a dict in '
                                f'the child config but is of type {type(out[k])} in base config. You may set '
                                f'`{DELETE_KEY}=True` to ignore the base config')
            out[k] = _merge(v, out[k])
        else:
            out[k] = v
    return out


def _load_file(filename):
    filename = osp.abspath(osp.expanduser(filename))
    if not osp.isfile(filename):
        raise FileNotFoundError(f'file "{filename}" does not exist')
    if filename.endswith('.py'):
        with open(filename) as f:
            compile(f.read(), filename, 'exec')  # surfaces SyntaxError with the config's name
        ns = runpy.run_path(filename)
        cfg = {k: v for k, v in ns.items() if not k.startswith('__') and not callable(v)
               and not isinstance(v, type(osp))}
    elif filename.endswith('.json'):
        with open(filename) as f:
            cfg = json.load(f)
    elif filename.endswith(('.yml', '.yaml')):
        import yaml
        with open(filename) as f:
            cfg = yaml.safe_load(f)
    else:
        raise IOError('Only py/yml/yaml/json type are supported now!')
    with open(filename) as f:
        text = filename + '\n' + f.read()
    if BASE_KEY in cfg:
        bases = cfg.pop(BASE_KEY)
        bases = bases if isinstance(bases, list) else [bases]
        merged, texts = {}, []
        for b in bases:
            bcfg, btext = _load_file(osp.join(osp.dirname(filename), b))
            if merged.keys() & bcfg.keys():
                raise KeyError('Duplicate key is not allowed among bases')
            merged.update(bcfg)
            texts.append(btext)
        cfg = _merge(cfg, merged)
        text = '\n'.join(texts + [text])
    return cfg, text


class Config:
    """`Config.fromfile(path)`; attribute and item access; `merge_from_dict({'a.b': v})`."""

    @staticmethod
    def fromfile(filename):
        cfg, text = _load_file(filename)
        return Config(cfg, cfg_text=text, filename=filename)

    def __init__(self, cfg_dict=None, cfg_text=None, filename=None):
        cfg_dict = {} if cfg_dict is None else cfg_dict
        if not isinstance(cfg_dict, dict):
            raise TypeError(f'cfg_dict must be a dict, but got {type(cfg_dict)}')
        for key in cfg_dict:
            if key in RESERVED_KEYS:
                raise KeyError(f'{key} is reserved for config file')
        object.__setattr__(self, '_cfg_dict', ConfigDict(cfg_dict))
        object.__setattr__(self, '_filename', filename)
        if cfg_text is None and filename:
            with open(filename) as f:
                cfg_text = f.read()
        object.__setattr__(self, '_text', cfg_text or '')

    filename = property(lambda self: self._filename)
    text = property(lambda self: self._text)

    @property
    def pretty_text(self):
        return json.dumps(self._cfg_dict.to_dict(), indent=4, default=str)

    def __repr__(self):
        return f'Config (path: {self.filename}): {self._cfg_dict!r}'

    def __len__(self):
        return len(self._cfg_dict)

    def __iter__(self):
        return iter(self._cfg_dict)

    def __contains__(self, k):
        return k in self._cfg_dict

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setattr__(self, name, value):
        self._cfg_dict[name] = value

    def __setitem__(self, name, value):
        self._cfg_dict[name] = value

    def get(self, k, default=None):
        return self._cfg_dict.get(k, default)

    def merge_from_dict(self, options):
        nested = {}
        for full_key, v in options.items():
            d = nested
            *parents, leaf = full_key.split('.')
            for p in parents:
                d = d.setdefault(p, {})
            d[leaf] = v
        object.__setattr__(self, '_cfg_dict', ConfigDict(_merge(nested, self._cfg_dict)))

    def dump(self, file=None):
        text = self.pretty_text
        if file is None:
            return text
        with open(file, 'w') as f:
            f.write(text)
