"""Attribute-access configuration tree on top of PyYAML.

Stands in for the OmegaConf objects of the reference (train.py:516-545,
utils/commons.py:181-182): attribute and item access, ``.get(key, default)``,
in-place assignment (models mutate their config, models/vqvae/vqvae.py:66-67),
recursive merge and a YAML save/load round trip.  PyYAML reads ``1e-9`` as the
string '1e-9' (no dot), so scalars that look like floats are coerced on load.
"""
import re

import yaml

_FLOAT_RE = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)([eE][+-]?\d+)?$")


class Config(dict):
    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as exc:
            raise AttributeError(key) from exc

    def __setattr__(self, key, value):
        self[key] = _wrap(value)

    def __delattr__(self, key):
        del self[key]

    def to_dict(self):
        return _unwrap(self)

    def copy(self):
        return _wrap(_unwrap(self))


def _coerce(value):
    if isinstance(value, str) and _FLOAT_RE.match(value) and not value.isdigit():
        return float(value)
    return value


def _wrap(obj):
    if isinstance(obj, Config):
        return obj
    if isinstance(obj, dict):
        return Config({k: _wrap(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return [_wrap(v) for v in obj]
    return _coerce(obj)


def _unwrap(obj):
    if isinstance(obj, dict):
        return {k: _unwrap(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_unwrap(v) for v in obj]
    return obj


def create(mapping):
    return _wrap(mapping)


def load(path):
    with open(path, encoding="utf-8") as f:
        return _wrap(yaml.safe_load(f) or {})


def save(config, path):
    with open(path, "w", encoding="utf-8") as f:
        yaml.safe_dump(_unwrap(config), f, sort_keys=False)


def merge(*configs):
    """Later configs win; nested mappings merge key by key (OmegaConf.merge semantics)."""
    out = Config()
    for cfg in configs:
        _merge_into(out, _wrap(cfg))
    return out


def _merge_into(dst, src):
    for key, value in src.items():
        if isinstance(value, Config) and isinstance(dst.get(key), Config):
            _merge_into(dst[key], value)
        else:
            dst[key] = value.copy() if isinstance(value, Config) else value
