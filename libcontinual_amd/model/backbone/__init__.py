from .resnet import *  # noqa: F401,F403
from .vit import ViTZoo, VisionTransformer, vit_pt_imnet  # noqa: F401
from .sinet import SiNet_vit, ViT_lora_co  # noqa: F401
