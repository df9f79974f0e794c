"""Linear classifier head on libclhip (replaces nn.Linear heads: ewc.py:50, lwf.py:29-40, icarl.py:31)."""
import torch.nn as nn

from .. import ops


class HipLinear(nn.Linear):
    """nn.Linear-compatible (same attributes, init and state_dict) but forward/backward are the
    clhip_linear_fwd / clhip_linear_bwd kernels."""

    def forward(self, input):
        return ops.linear(input, self.weight, self.bias)
