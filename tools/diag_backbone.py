"""Diagnostic (GPU): per-unit relative errors of z / y / dy and parameter gradients, HIP plan vs fp64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import libcontinual_amd.model as M
from oracle import nets

arch, dtype, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
g = torch.Generator().manual_seed(3)
P = nets.init_params(arch, g); Bf = nets.init_buffers(arch)
x = torch.randn(B, 3, 32, 32, generator=g); cw = torch.randn(B, nets.arch(arch)[1], generator=g)
if len(sys.argv) > 4 and sys.argv[4] == "det":
    from oracle import fixtures as fx
    P, Bf = fx.det_backbone_state(arch, "ewc"); x = fx.det_images("ewc/t0/0/x", B)
Pg = {k: v.double().requires_grad_(True) for k, v in P.items()}
Bo = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in Bf.items()}
f_ref, acts = nets.forward(arch, Pg, Bo, x.double(), True, return_acts=True)
for k, v in acts.items():
    if v.requires_grad: v.retain_grad()
(f_ref * cw.double()).sum().backward()
bb = {"cifar_resnet32": M.cifar_resnet32, "resnet32_V2": M.resnet32_V2}.get(arch, lambda **k: M.resnet18(args={"dataset": "cifar100"}, **k))(dtype=dtype)
bb.load_state_dict({**P, **Bf}); bb = bb.cuda(); bb.train()
f = bb(x.cuda())["features"]; (f * cw.cuda()).sum().backward()
rel = lambda a, b: float((a.double().cpu() - b).norm() / (b.norm() + 1e-300))
units = nets.arch(arch)[0]
grads = dict(bb.named_parameters())
for i, u in enumerate(units):
    z = bb.debug_read(i + 1, 1); y = bb.debug_read(i + 1, 0); dy = bb.debug_read(i + 1, 2)
    print(f"{i:2d} {u.conv:28s} z {rel(z, acts[u.dst+'#z'].detach()):.2e} y {rel(y, acts[u.dst].detach()):.2e} dy {rel(dy, acts[u.dst].grad):.2e} "
          f"| dW {rel(grads[u.conv+'.weight'].grad, Pg[u.conv+'.weight'].grad):.2e} dgamma {rel(grads[u.bn+'.weight'].grad, Pg[u.bn+'.weight'].grad):.2e} "
          f"dbeta {rel(grads[u.bn+'.bias'].grad, Pg[u.bn+'.bias'].grad):.2e}  |dgamma| {Pg[u.bn+'.weight'].grad.norm():.2e} |dy| {acts[u.dst].grad.norm():.2e}")
