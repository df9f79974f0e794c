from .resnet import *  # noqa: F401,F403
from .vit import ViTZoo, VisionTransformer, vit_pt_imnet  # noqa: F401
