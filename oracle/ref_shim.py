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

    class Mlp(nn.Module):
        """timm.models.layers.Mlp: fc1 -> act -> drop -> fc2 -> drop"""

        def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
            super().__init__()
            assert drop == 0.0
            self.fc1 = nn.Linear(in_features, hidden_features or in_features)
            self.act = act_layer()
            self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)

        def forward(self, x):
            return self.fc2(self.act(self.fc1(x)))

    def named_apply(fn, module, name="", depth_first=True, include_root=False):
        """timm.models.helpers.named_apply"""
        if not depth_first and include_root:
            fn(module=module, name=name)
        for cn, cm in module.named_children():
            named_apply(fn, cm, ".".join((name, cn)) if name else cn, depth_first, True)
        if depth_first and include_root:
            fn(module=module, name=name)
        return module

    def lecun_normal_(t):
        fan_in = t.shape[1] if t.dim() > 1 else t.shape[0]
        return torch.nn.init.trunc_normal_(t, std=(1.0 / fan_in) ** 0.5 / .87962566103423978)

    ident = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
    mod("timm", create_model=None)
    mod("timm.models")
    mod("timm.models.vision_transformer", PatchEmbed=PatchEmbed, _cfg=lambda **k: dict(k))
    mod("timm.models.layers", trunc_normal_=torch.nn.init.trunc_normal_, DropPath=DropPath, PatchEmbed=PatchEmbed, Mlp=Mlp, lecun_normal_=lecun_normal_)
    mod("timm.models.registry", register_model=ident)
    mod("timm.models.helpers", named_apply=named_apply, adapt_input_conv=None, build_model_with_cfg=None, resolve_pretrained_cfg=None, checkpoint_seq=None)
    mod("timm.data", IMAGENET_DEFAULT_MEAN=(0.485, 0.456, 0.406), IMAGENET_DEFAULT_STD=(0.229, 0.224, 0.225),
        IMAGENET_INCEPTION_MEAN=(0.5, 0.5, 0.5), IMAGENET_INCEPTION_STD=(0.5, 0.5, 0.5))
    mod("torchvision")
    mod("torchvision.models")
    mod("ftfy", fix_text=lambda t: t)
    install()
