"""Import the reference's own Python files from /root/reference WITHOUT running its package
__init__s (which need torchvision / timm / continuum, absent here).  BUILD-CONTAINER ONLY: used by
`oracle/gen_golden.py` and `tests/test_oracle_vs_reference.py` (skipped when /root/reference is
absent, i.e. on the GPU box).  Recipe: SURVEY.md appendix A.  Nothing from the reference is copied.
"""
import importlib
import os
import sys
import types

REF_ROOT = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "core", "model"))


def install():
    """Register empty packages whose __path__ points into the reference tree."""
    if not available():
        raise RuntimeError("reference tree not present")
    sys.dont_write_bytecode = True
    pkgs = {
        "core": "core",
        "core.model": "core/model",
        "core.model.backbone": "core/model/backbone",
        "core.model.buffer": "core/model/buffer",
    }
    for name, rel in pkgs.items():
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(REF_ROOT, rel)]
            sys.modules[name] = m
    return sys.modules["core"]


def load(modname):
    install()
    return importlib.import_module(modname)
