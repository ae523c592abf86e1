"""Minimal registry / config layer with the same surface the reference gets from mmcv
(`Registry`, `build_from_cfg`, `Config.fromfile`, `--cfg-options`), so that configs/fcos_semi/*.py
build unmodified: mmdet/models/builder.py:6-59, mmdet/apis/train.py:111-135, tools/train.py:55-91."""
import ast
import os


class Registry:
    def __init__(self, name, parent=None):
        self.name = name
        self._module_dict = {} if parent is None else parent._module_dict

    def register_module(self, name=None, force=False, module=None):
        def _reg(cls):
            key = name or cls.__name__
            if key in self._module_dict and not force:
                raise KeyError(f'{key} is already registered in {self.name}')
            self._module_dict[key] = cls
            return cls
        if module is not None:
            return _reg(module)
        return _reg

    def get(self, key):
        return self._module_dict.get(key)

    def __contains__(self, key):
        return key in self._module_dict

    def build(self, cfg, default_args=None):
        return build_from_cfg(cfg, self, default_args)


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict) or 'type' not in cfg:
        raise KeyError(f'cfg must be a dict with a "type" key, got {cfg!r}')
    args = dict(cfg)
    for k, v in (default_args or {}).items():
        args.setdefault(k, v)
    t = args.pop('type')
    cls = registry.get(t) if isinstance(t, str) else t
    if cls is None:
        raise KeyError(f'{t} is not in the {registry.name} registry')
    return cls(**args)


MODELS = Registry('models')
BACKBONES = NECKS = HEADS = LOSSES = DETECTORS = MODELS      # mmdet/models/builder.py:8-14
RUNNERS = Registry('runner')
HOOKS = Registry('hook')
OPTIMIZERS = Registry('optimizer')


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    """mmdet/models/builder.py:48-59."""
    assert cfg.get('train_cfg') is None or train_cfg is None
    assert cfg.get('test_cfg') is None or test_cfg is None
    return DETECTORS.build(cfg, default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg))


class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return v

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(v):
    if isinstance(v, dict):
        return ConfigDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    if isinstance(v, tuple):
        return tuple(_wrap(x) for x in v)
    return v


class Config:
    """Python-file configs, attribute access, dotted-key overrides."""

    def __init__(self, d, filename=None):
        object.__setattr__(self, '_cfg', _wrap(d))
        object.__setattr__(self, 'filename', filename)

    @staticmethod
    def fromfile(path):
        g = {'__file__': os.path.abspath(path)}
        with open(path) as f:
            exec(compile(f.read(), path, 'exec'), g)
        d = {k: v for k, v in g.items() if not k.startswith('__') and not callable(v) and not isinstance(v, type(os))}
        return Config(d, path)

    def merge_from_dict(self, options):
        for key, val in options.items():
            cur = self._cfg
            parts = key.split('.')
            for p in parts[:-1]:
                cur = cur.setdefault(p, ConfigDict()) if isinstance(cur, dict) else cur[int(p)]
            last = parts[-1]
            if isinstance(cur, list):
                cur[int(last)] = _wrap(val)
            else:
                cur[last] = _wrap(val)

    @staticmethod
    def parse_cfg_options(items):
        out = {}
        for it in items or []:
            k, v = it.split('=', 1)
            try:
                out[k] = ast.literal_eval(v)
            except (ValueError, SyntaxError):
                out[k] = v
        return out

    def __getattr__(self, k):
        return getattr(self._cfg, k)

    def __setattr__(self, k, v):
        self._cfg[k] = _wrap(v)

    def __getitem__(self, k):
        return self._cfg[k]

    def __setitem__(self, k, v):
        self._cfg[k] = _wrap(v)

    def get(self, k, default=None):
        return self._cfg.get(k, default)

    def __contains__(self, k):
        return k in self._cfg
