"""TEST INFRASTRUCTURE (CPU, fp64): where does the product's bf16 parameter-gradient deviation come from?  

The bf16 mode's whole-gradient deviation from the fp64 oracle on a random-init CifarResNet-32 (the smoke scenario: batch 64, EWC task-0 step)
is 0.53 in relative L2 norm, 1.3 x the "operand-rounding yardstick" (the oracle itself with only conv weights and the input rounded to bf16:
0.40).  This script reproduces the product's STORAGE FORMAT inside the fp64 oracle, one rounding site at a time -- every tensor the HIP path
keeps in bf16 (conv outputs z, activations y, activation gradients dy, pre-BatchNorm gradients dz) is rounded to bf16 exactly where the product
stores it, all arithmetic stays fp64 -- and prints the deviation of each cumulative variant.  If the last variant lands where the product is,
the extra third is the price of bf16 STORAGE of activations and gradients (the format the benchmark line is quoted in), not an arithmetic
defect of a kernel.

    python -m oracle.bf16_floor_study [arch=cifar_resnet32] [batch=64]  > profiles/r04_bf16_floor_study.md
"""
import sys

import torch
import torch.nn.functional as F

from . import fixtures as fx, nets


def rb(t):
    return t.to(torch.bfloat16).to(t.dtype)


class Store(torch.autograd.Function):
    """y = round_bf16(x) if fwd else x;  dx = round_bf16(dy) if bwd else dy  (a tensor kept in bf16 on the way forward and / or back)"""

    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return rb(x) if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (rb(g) if ctx.bwd else g), None, None


def forward(name, P, Bf, x, sites):
    """nets.forward (train mode) with the product's bf16 storage sites switched on by name: "z" (conv output), "y" (activation),
    "dy" (gradient of an activation), "dz" (gradient of a conv output = the BatchNorm-backward result).  BatchNorm statistics are taken
    from the UNROUNDED conv output, as the product takes them from its fp32 accumulators."""
    units, _, _ = nets.arch(name)
    acts = {"input": x}
    for u in units:
        z = F.conv2d(acts[u.src], P[u.conv + ".weight"], None, u.stride, u.pad)
        mean = z.mean(dim=(0, 2, 3), keepdim=True)
        var = z.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
        zs = Store.apply(z, "z" in sites, "dz" in sites)                 # what the apply pass reads / what the backward writes as dz
        g, b = P[u.bn + ".weight"].view(1, -1, 1, 1), P[u.bn + ".bias"].view(1, -1, 1, 1)
        y = (zs - mean) / torch.sqrt(var + nets.BN_EPS) * g + b
        if u.res is not None:
            y = y + acts[u.res]
        if u.relu:
            y = F.relu(y)
        acts[u.dst] = Store.apply(y, "y" in sites, "dy" in sites)
    return acts[units[-1].dst].mean(dim=(2, 3))


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else "cifar_resnet32"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    with fx.use_dtype(torch.float64):
        P0, Bf = fx.det_backbone_state(arch, "smoke")
        fd = nets.arch(arch)[1]
        w0, b0 = fx.det_linear("smoke/head", 10, fd)
        x, y = fx.det_batch("smoke/batch", B, 0, 10)
    P0 = {k: v.double() for k, v in P0.items()}
    x, w0, b0 = x.double(), w0.double(), b0.double()

    def grads(sites, round_operands):
        P = {k: ((rb(v) if (round_operands and v.dim() == 4) else v).clone().requires_grad_(True)) for k, v in P0.items()}
        xin = rb(x) if round_operands else x
        f = forward(arch, P, Bf, xin, sites)
        loss = F.cross_entropy(f @ w0.T + b0, y)
        loss.backward()
        names = sorted(k for k, v in P.items() if v.grad is not None)
        return float(loss), torch.cat([P[k].grad.reshape(-1) for k in names]), {k: P[k].grad for k in names}
    l_ref, g_ref, per_ref = grads((), False)
    rows = [("operands only: conv weights + input rounded (the yardstick)", ()),
            ("+ conv outputs z stored in bf16", ("z",)),
            ("+ activations y stored in bf16", ("z", "y")),
            ("+ activation gradients dy stored in bf16", ("z", "y", "dy")),
            ("+ pre-BatchNorm gradients dz stored in bf16 (= every tensor the HIP path keeps in bf16)", ("z", "y", "dy", "dz"))]
    print(f"# Where the bf16 gradient deviation comes from: {arch}, batch {B}, CE step on random-init weights, everything computed in fp64\n")
    print("Deviation of the WHOLE parameter gradient from the fp64 oracle (relative L2 norm) when the tensors the HIP path stores in bf16 are rounded to")
    print("bf16 at their storage sites, cumulatively; arithmetic in fp64 throughout (`python -m oracle.bf16_floor_study`).\n")
    print("| storage format | loss rel. deviation | whole-gradient rel. L2 | / yardstick | worst single tensor |")
    print("|---|---|---|---|---|")
    yard = None
    for label, sites in rows:
        l, g, per = grads(sites, True)
        d = float((g - g_ref).norm() / g_ref.norm())
        yard = yard or d
        worst = max(per, key=lambda k: float((per[k] - per_ref[k]).norm() / per_ref[k].norm()))
        wv = float((per[worst] - per_ref[worst]).norm() / per_ref[worst].norm())
        print(f"| {label} | {abs(l - l_ref) / abs(l_ref):.2e} | {d:.3f} | {d / yard:.2f} | {wv:.3f} ({worst}) |")


if __name__ == "__main__":
    main()
