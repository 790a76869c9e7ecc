"""YAML → attribute-dict configuration, same surface as the reference's pcdet/config.py:16-85
(cfg_from_yaml_file with _BASE_CONFIG_ includes and recursive merge, cfg_from_list typed overrides),
without the easydict dependency (absent from the image)."""
import os
from ast import literal_eval

import yaml


class AttrDict(dict):
    """dict with attribute access, recursive over nested dicts/lists (what easydict.EasyDict gives pcdet)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def update(self, *a, **kw):
        for k, v in dict(*a, **kw).items():
            self[k] = v


CFG_ROOT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cfgs")

cfg = AttrDict()
cfg.ROOT_DIR = os.path.dirname(CFG_ROOT)
cfg.LOCAL_RANK = 0


def _resolve(path, relative_to=None):
    cands = [path]
    if relative_to:
        cands.append(os.path.join(os.path.dirname(relative_to), path))
    cands.append(os.path.join(os.path.dirname(CFG_ROOT), path))      # 'cfgs/dataset_configs/x.yaml' from the repo root
    cands.append(os.path.join(CFG_ROOT, path))
    for c in cands:
        if os.path.isfile(c):
            return c
    raise FileNotFoundError(path)


def merge_new_config(config, new_config, _origin=None):
    if "_BASE_CONFIG_" in new_config:
        with open(_resolve(new_config["_BASE_CONFIG_"], _origin)) as f:
            base = yaml.safe_load(f)
        config.update(AttrDict(base))
    for key, val in new_config.items():
        if not isinstance(val, dict):
            config[key] = val
            continue
        if key not in config or not isinstance(config[key], dict):
            config[key] = AttrDict()
        merge_new_config(config[key], val, _origin)
    return config


def cfg_from_yaml_file(cfg_file, config):
    cfg_file = _resolve(cfg_file)
    with open(cfg_file) as f:
        new_config = yaml.safe_load(f)
    merge_new_config(config, new_config, cfg_file)
    return config


def cfg_from_list(cfg_list, config):
    """Set config keys via list (e.g. from the command line): ['A.B', 'value', ...]."""
    assert len(cfg_list) % 2 == 0
    for k, v in zip(cfg_list[0::2], cfg_list[1::2]):
        key_list = k.split(".")
        d = config
        for subkey in key_list[:-1]:
            assert subkey in d, "NotFoundKey: %s" % subkey
            d = d[subkey]
        subkey = key_list[-1]
        assert subkey in d, "NotFoundKey: %s" % subkey
        try:
            value = literal_eval(v)
        except Exception:
            value = v
        if isinstance(d[subkey], dict) and not isinstance(value, dict):
            key_val_list = value.split(",")
            for src in key_val_list:
                cur_key, cur_val = src.split(":")
                val_type = type(d[subkey][cur_key])
                d[subkey][cur_key] = val_type(cur_val)
        elif isinstance(d[subkey], list) and isinstance(value, str):
            val_list = value.split(",")
            for kk, x in enumerate(val_list):
                val_list[kk] = type(d[subkey][0])(x)
            d[subkey] = val_list
        else:
            assert type(value) == type(d[subkey]), "type {} does not match original type {}".format(type(value), type(d[subkey]))
            d[subkey] = value


def log_config_to_file(config, pre="cfg", logger=None):
    for key, val in config.items():
        if isinstance(val, dict):
            logger.info("----------- %s -----------" % key)
            log_config_to_file(val, pre=pre + "." + key, logger=logger)
            continue
        logger.info("%s.%s: %s" % (pre, key, val))
