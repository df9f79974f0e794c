"""BUILD CONTAINER ONLY (needs /root/reference): the CPU restatement `oracle/` that bench.py times as `cpu_baseline` (kind "port") next to the
imported reference classes on the same step, same threads -- SURVEY.md section 8(d): "the restatement is additionally timed side-by-side with
the imported reference classes to show they cost the same".  LwF task-0 step (forward + CE + backward + SGD), fp32, torch CPU.

    python -m oracle.cpu_vs_reference [batch=64] [steps=4] > profiles/r04_cpu_restatement_vs_reference.md
(TEST INFRASTRUCTURE: lives under oracle/ like acc_runs.py; nothing of the product imports it)
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    from oracle import gen_golden, methods as om, nets, ref_shim
    assert ref_shim.available(), "the reference tree is not here (this tool runs in the build container only)"
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(1)
    rows = []
    for arch, fd in (("resnet18", 512), ("cifar_resnet32", 64)):
        x = torch.rand(B, 3, 32, 32, generator=g)
        y = torch.randint(0, 50, (B,), generator=g)
        # ---- the restatement (what bench.py's cpu_baseline leg runs)
        torch.manual_seed(0)
        P = {k: v.requires_grad_(True) for k, v in nets.init_params(arch).items()}
        Bf = nets.init_buffers(arch)
        w, b = om.linear_default_init(50, fd)
        net = om.Net(arch, P, Bf, w.requires_grad_(True), b.requires_grad_(True))
        m = om.LWF(net, 50, 5)
        m.before_task(0, (w, b))
        opt = om.SGD(net.parameters(), 0.1)

        def step_o():
            _, _, loss = m.observe(x, y, True)
            opt.zero_grad(); loss.backward(); opt.step()
        # ---- the reference's own classes
        ns = gen_golden.reference_namespace()
        torch.manual_seed(0)
        bb = ns.resnet18(args={"dataset": "cifar100"}) if arch == "resnet18" else ns.cifar_resnet32()
        rm = ns.LWF(bb, fd, 100, init_cls_num=50, inc_cls_num=5, device="cpu")
        rm.before_task(0, None, None, None)
        rm.train()
        ropt = torch.optim.SGD(rm.parameters(), lr=0.1)

        def step_r():
            _, _, loss = rm.observe({"image": x, "label": y})
            ropt.zero_grad(); loss.backward(); ropt.step()
        res = {}
        for name, fn in (("restatement (oracle/)", step_o), ("imported reference classes", step_r), ("restatement, 2nd pass", step_o), ("reference, 2nd pass", step_r)):
            fn()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            res[name] = (time.perf_counter() - t0) / steps
        rows.append((arch, res))
    print(f"# CPU: the restatement `oracle/` against the imported reference classes (build container, {cores} cores, torch {torch.__version__}, fp32)\n")
    print(f"LwF task-0 step (forward, CE, backward, SGD) at batch {B}, {steps} timed steps after one warm-up, alternating.  `bench.py`'s `cpu_baseline`")
    print("(kind \"port\") times the restatement on the GPU box's host cores, where the reference's files do not exist; here both run side by side.\n")
    print("| backbone | restatement s/step | reference s/step | ratio | restatement img/s | reference img/s |")
    print("|---|---|---|---|---|---|")
    for arch, r in rows:
        o = 0.5 * (r["restatement (oracle/)"] + r["restatement, 2nd pass"])
        f = 0.5 * (r["imported reference classes"] + r["reference, 2nd pass"])
        print(f"| {arch} | {o:.3f} | {f:.3f} | {o / f:.2f} | {B / o:.1f} | {B / f:.1f} |")
    print("\nBoth are the same torch CPU operators in the same order (the restatement calls `F.conv2d` / `F.batch_norm` on explicit parameter dictionaries, the")
    print("reference its `nn.Module`s); the ratio is what module dispatch and the optimizer classes differ by.")


if __name__ == "__main__":
    main()
