"""`Config.fromfile` / `merge_from_dict` for the reference's Python config files (tools/test.py:124-126)."""
import copy
import os
import runpy


class ConfigDict(dict):
    """dict with attribute access (mmcv ConfigDict behaviour the reference relies on: cfg.model, cfg.get)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(obj):
    if isinstance(obj, dict):
        return ConfigDict({k: _wrap(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [_wrap(v) for v in obj]
    if isinstance(obj, tuple):
        return tuple(_wrap(v) for v in obj)
    return obj


class Config(ConfigDict):
    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = _wrap(v)

    @staticmethod
    def fromfile(path):
        path = os.path.abspath(path)
        ns = runpy.run_path(path)
        cfg = Config()
        for k, v in ns.items():
            if k.startswith("__") or callable(v) or isinstance(v, type(os)):
                continue
            cfg[k] = _wrap(v)
        cfg["filename"] = path
        return cfg

    def merge_from_dict(self, options):
        """`--cfg-options a.b.c=v` semantics: dotted keys address nested dicts (lists by integer index)."""
        for key, value in options.items():
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                if isinstance(node, (list, tuple)):
                    node = node[int(p)]
                else:
                    if p not in node:
                        node[p] = ConfigDict()
                    node = node[p]
            last = parts[-1]
            if isinstance(node, list):
                node[int(last)] = _wrap(value)
            else:
                node[last] = _wrap(value)
        return self
