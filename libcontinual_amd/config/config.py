"""YAML config loader with the reference's merge semantics (core/config/config.py:18-134):
default.yaml < user file (its `includes:` resolved relative to ./config/ or, failing that, to the packaged
copies of the reference headers; the file's own keys override its includes) < console (stub).  Scientific
notation such as `5e-4` parses as float (custom resolver, config.py:59-72)."""
import os
import re

import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_FILE = os.path.join(_HERE, "default.yaml")
PACKAGED_CONFIG_DIR = os.path.join(_HERE, "yaml")


def _loader():
    class L(yaml.SafeLoader):
        pass
    L.add_implicit_resolver(
        "tag:yaml.org,2002:float",
        re.compile(r"""^(?:[-+]?[0-9][0-9_]*\.[0-9_]*(?:[eE][-+]?[0-9]+)?
                       |[-+]?[0-9][0-9_]*[eE][-+]?[0-9]+
                       |\.[0-9_]+(?:[eE][-+][0-9]+)?
                       |[-+]?[0-9][0-9_]*(?::[0-5]?[0-9])+\.[0-9_]*
                       |[-+]?\.(?:inf|Inf|INF)
                       |\.(?:nan|NaN|NAN))$""", re.X),
        list("-+0123456789."))
    return L


def _find_include(name):
    for base in ("./config/", PACKAGED_CONFIG_DIR):
        p = os.path.join(base, name)
        if os.path.exists(p):
            return p
    raise FileNotFoundError(f"include {name!r} not found under ./config/ or {PACKAGED_CONFIG_DIR}")


class Config:
    def __init__(self, config_file=None):
        self.config_file = config_file
        self.default_dict = self._load_config_files(DEFAULT_FILE)
        self.file_dict = self._load_config_files(config_file)
        self.console_dict = None
        self.config_dict = {}
        for d in (self.default_dict, self.file_dict, self.console_dict):
            if d:
                self.config_dict.update(d)          # shallow, like the reference's _update (config.py:95-114)

    def get_config_dict(self):
        return self.config_dict

    @staticmethod
    def _load_config_files(config_file):
        cfg = {}
        L = _loader()
        if config_file is not None:
            with open(config_file, "r", encoding="utf-8") as f:
                cfg.update(yaml.load(f.read(), Loader=L) or {})
        own = dict(cfg)
        for inc in cfg.get("includes", None) or []:
            with open(_find_include(inc), "r", encoding="utf-8") as f:
                cfg.update(yaml.load(f.read(), Loader=L) or {})
        cfg.pop("includes", None)
        cfg.update(own)
        cfg.pop("includes", None)
        return cfg
