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
        "core.model.backbone.petl": "core/model/backbone/petl",
        "core.model.backbone.tokenizer": "core/model/backbone/tokenizer",
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


def install_vit_standins():
    """Stand-ins for the third-party names the reference's ViT files import but this image lacks (SURVEY.md
    appendix A.4): timm 0.x (`PatchEmbed`, `trunc_normal_`, `DropPath`, `register_model`, ...), torchvision
    (imported, unused on this path) and ftfy (tokenizer of the CLIP branch, unused).  Each is restated from its
    published definition; none of them is reference code."""
    import torch
    import torch.nn as nn

    def mod(name, **attrs):
        m = sys.modules.get(name)
        if m is None:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
        for k, v in attrs.items():
            setattr(m, k, v)
        return m

    class PatchEmbed(nn.Module):
        """timm.models.vision_transformer.PatchEmbed: Conv2d(in, D, p, stride p) -> flatten(2).transpose(1, 2)"""

        def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
            super().__init__()
            self.img_size, self.patch_size = (img_size, img_size), (patch_size, patch_size)
            self.num_patches = (img_size // patch_size) ** 2
            self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

        def forward(self, x):
            return self.proj(x).flatten(2).transpose(1, 2)

    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.0):
            super().__init__()
            assert drop_prob == 0.0

        def forward(self, x):
            return x

    ident = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
    mod("timm", create_model=None)
    mod("timm.models")
    mod("timm.models.vision_transformer", PatchEmbed=PatchEmbed, _cfg=lambda **k: dict(k))
    mod("timm.models.layers", trunc_normal_=torch.nn.init.trunc_normal_, DropPath=DropPath)
    mod("timm.models.registry", register_model=ident)
    mod("timm.models.helpers", named_apply=None, adapt_input_conv=None)
    mod("torchvision")
    mod("torchvision.models")
    mod("ftfy", fix_text=lambda t: t)
    install()
