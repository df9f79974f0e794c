"""Registry namespace: YAML `backbone.name` / `classifier.name` / `buffer.name` are attribute names looked
up here by `get_instance` (reference core/model/__init__.py:1-34, core/utils/utils.py:77-92)."""
from .backbone import *  # noqa: F401,F403
from .buffer import *  # noqa: F401,F403
from .finetune import Finetune  # noqa: F401
from .ewc import EWC  # noqa: F401
from .lwf import LWF  # noqa: F401
from .icarl import ICarl  # noqa: F401
from .lucir import LUCIR  # noqa: F401
from .wa import WA  # noqa: F401
from .der import DER  # noqa: F401
from .bic import bic, BiasLayer  # noqa: F401
from .l2p import L2P  # noqa: F401
from .inflora_opt import InfLoRA_OPT  # noqa: F401
from .inflora import InfLoRA  # noqa: F401
from .heads import HipLinear  # noqa: F401
