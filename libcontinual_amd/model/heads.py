"""Linear classifier head on libclhip (replaces nn.Linear heads: ewc.py:50, lwf.py:29-40, icarl.py:31) and the two
helpers every growing-head method shares: widen a head keeping the learned rows, and take a frozen teacher copy."""
import copy

import torch
import torch.nn as nn

from .. import ops


class HipLinear(nn.Linear):
    """nn.Linear-compatible (same attributes, init and state_dict) but forward/backward are the
    clhip_linear_fwd / clhip_linear_bwd kernels."""

    def forward(self, input):
        return ops.linear(input, self.weight, self.bias)


def widened(head, n_out, device=None):
    """a fresh HipLinear with `n_out` outputs whose first rows are `head`'s.  The new rows keep the
    nn.Linear default init, which is what the reference's `nn.Linear(...)` + row copy leaves there."""
    grown = HipLinear(head.in_features, n_out).to(device if device is not None else head.weight.device)
    keep = min(head.out_features, n_out)
    with torch.no_grad():
        grown.weight[:keep].copy_(head.weight[:keep])
        grown.bias[:keep].copy_(head.bias[:keep])
    return grown


def teacher_of(module, device=None):
    """deep copy with gradients off, switched to eval (callers that keep it as an nn.Module attribute get it flipped back to
    train mode by `model.train()`, exactly like the reference -- SURVEY.md section 8a quirk a10)"""
    t = copy.deepcopy(module)
    for q in t.parameters():
        q.requires_grad_(False)
    t.eval()
    return t.to(device) if device is not None else t
