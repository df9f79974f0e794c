"""Golden scenarios of the ViT path: backbone, L2P, InfLoRA_OPT (TEST INFRASTRUCTURE, see oracle/__init__ and
oracle/scenarios.py for the three executors: reference -> fixtures, oracle (CPU), product (MI355X)).

The reference's ViTZoo hard-codes ViT-B/16; the fixtures drive the same reference classes
(`VisionTransformer`, `MultiHeadAttention_LoRA`, `prompt.L2P`, `L2P`, `InfLoRA_OPT`, `SiNet`) at a small
configuration (oracle.vit.VIT_TINY) through `make_vit`, so that fp64 runs take seconds and fixtures stay small.
"""
import types

import numpy as np
import torch

from . import detrand
from . import fixtures as fx
from . import vit as ov
from .scenarios import ListLoader

CFG = ov.VIT_TINY
L2P_CFG = dict(init=3, inc=3, total=6, task_num=2, length=2, pool=6, top_k=3, coeff=1.0, bs=8, lr=0.01)
INF_CFG = dict(init=3, inc=3, task_num=2, lame=0.9, lamb=0.6, rank=4, bs=8, lr=0.05, momentum=0.9)


def det_images(tag, n, cfg=CFG):
    return fx._t(detrand.uniform(tag, (n, 3, cfg["img"], cfg["img"]), -1.0, 1.0))


def det_labels(tag, n, lo, hi):
    return torch.from_numpy(detrand.randint(tag, (n,), lo, hi).astype(np.int64))


def det_extra(tag, shapes, lo, hi):
    return {k: fx._t(detrand.uniform(f"{tag}/{k}", s, lo, hi)) for k, s in shapes.items()}


def backbone_params(tag, lora_rank=0):
    return {k: v.to(fx._DTYPE[0]) for k, v in ov.det_params(CFG, tag, lora_rank, torch.float64).items()}


class VitPluginAdapter:
    """ns: namespace with `make_vit(cfg, attn_layer, lora_rank)` -> ViTZoo-like backbone, `L2P`, `InfLoRA_OPT`;
    `optim(name, params, **kw)` -> optimizer"""

    kind = "plugin"

    def __init__(self, ns, device="cpu", optim=None):
        self.ns, self.device = ns, device
        self.optim = optim or (lambda name, params, **kw: getattr(torch.optim, name)(params, **kw))

    def batch(self, x, y):
        return {"image": x, "label": y}


class VitOracleAdapter:
    kind = "oracle"
    device = "cpu"


def _np(t):
    return t.detach().cpu().double().numpy().copy()


def _set_lora(bb, flag):
    for m in bb.modules():
        if hasattr(m, "apply_lora"):
            m.apply_lora = flag


# ------------------------------------------------------------------------------------------ backbone
def scenario_vit_backbone(adapter):
    """plain forward (cls feature) with and without an applied LoRA branch; per-block activations are not part of
    the plugin surface, so only the features are pinned"""
    tag = "vitbb"
    x = det_images(tag + "/x", 5)
    res = {}
    for name, rank in (("plain", 0), ("lora", 4)):
        P = backbone_params(tag, rank)
        if rank:
            for k in P:
                if "lora_B" in k:
                    P[k] = fx._t(detrand.uniform(f"{tag}/B/{k}", tuple(P[k].shape), -0.2, 0.2))
        if adapter.kind == "oracle":
            with torch.no_grad():
                f = ov.cls_features(P, x, CFG, lora=bool(rank))
        else:
            bb = adapter.ns.make_vit(CFG, "MultiHeadAttention_LoRA" if rank else "MultiHeadAttention", rank)
            bb.load_state_dict({k: v.clone() for k, v in P.items()}, strict=True)
            bb = bb.to(adapter.device)
            _set_lora(bb, bool(rank))
            bb.eval()
            with torch.no_grad():
                f = bb(x.to(adapter.device))
        res["feat_" + name] = _np(f)
    return res


# ---------------------------------------------------------------------------------------------- L2P
# the data tag is chosen so that no batch-majority vote has a tie at the top-k cut (its resolution is torch.topk
# implementation-defined, prompt.py:389); scenario_l2p on the oracle asserts it
L2P_TAG = ["l2p/v1"]


def _l2p_state(tag):
    c = L2P_CFG
    D = CFG["dim"]
    P = backbone_params(tag)
    # keys in (0,1) like nn.init.uniform_; prompts in (0,1) as well (l2p.py:64)
    P.update(det_extra(tag, {"prompt.prompt": (1, c["pool"], c["length"], D), "prompt.prompt_key": (c["pool"], D)}, 0.0, 1.0))
    b = 1.0 / np.sqrt(D)
    P.update(det_extra(tag, {"classifier.weight": (c["total"], D), "classifier.bias": (c["total"],)}, -b, b))
    return P


def scenario_l2p(adapter):
    """2 tasks x 2 steps of L2P (observe does backward + clip inside; Adam outside), inference after each task"""
    c = L2P_CFG
    tag = L2P_TAG[0]
    P = _l2p_state(tag)
    ov.TIES_AT_CUT.clear()
    xs = [det_images(f"{tag}/x{i}", c["bs"]) for i in range(4)]
    ys = [det_labels(f"{tag}/y{i}", c["bs"], 0 if i < 2 else c["init"], c["init"] if i < 2 else c["total"]) for i in range(4)]
    tx, ty = det_images(tag + "/tx", c["bs"]), det_labels(tag + "/ty", c["bs"], 0, c["total"])
    res, losses, preds = {}, [], []
    tr = ["prompt.prompt", "prompt.prompt_key", "classifier.weight", "classifier.bias"]
    if adapter.kind == "oracle":
        P = {k: v.clone() for k, v in P.items()}
        m = ov.L2P(P, CFG, c["init"], c["inc"], c["total"], c["top_k"], c["coeff"])
        opt = ov.Adam(m.parameters(), c["lr"])
        for t in range(2):
            m.before_task(t)
            for i in (2 * t, 2 * t + 1):
                pred, acc, loss, ids, norm = m.observe(xs[i], ys[i])
                if i == 0:
                    res["grad_prompt0"] = _np(P["prompt.prompt"].grad)
                    res["grad_key0"] = _np(P["prompt.prompt_key"].grad)
                    res["grad_cls_w0"] = _np(P["classifier.weight"].grad)
                opt.step()
                losses.append(float(loss.detach())); preds.append(pred.numpy())
            m.after_task(t)
            p, a = m.inference(tx, ty)
            res[f"test_pred{t}"] = p.numpy()
            for n in tr:
                res[f"{n}@{t}"] = _np(P[n])
        assert not ov.TIES_AT_CUT, f"prompt vote tie at the cut: {ov.TIES_AT_CUT}"
    else:
        ns = adapter.ns
        bb = ns.make_vit(CFG, "MultiHeadAttention", 0)
        m = ns.L2P(bb, adapter.device, init_cls_num=c["init"], inc_cls_num=c["inc"], num_class=c["total"], task_num=c["task_num"],
                   feat_dim=CFG["dim"], prompt_length=c["length"], pool_size=c["pool"], top_k=c["top_k"], pull_constraint_coeff=c["coeff"])
        sd = {("backbone." + k if not k.startswith("classifier") else k): v.clone() for k, v in P.items()}
        m.network.load_state_dict(sd, strict=True)
        m.network.to(adapter.device)
        opt = adapter.optim("Adam", m.get_parameters({}), lr=c["lr"], betas=(0.9, 0.999), weight_decay=0)
        named = dict(m.network.named_parameters())
        key = lambda n: n if n.startswith("classifier") else "backbone." + n
        for t in range(2):
            m.before_task(t, None, None, None)
            for i in (2 * t, 2 * t + 1):
                m.train()
                opt.zero_grad()
                pred, acc, loss = m.observe(adapter.batch(xs[i], ys[i]))
                if i == 0:
                    res["grad_prompt0"] = _np(named[key("prompt.prompt")].grad)
                    res["grad_key0"] = _np(named[key("prompt.prompt_key")].grad)
                    res["grad_cls_w0"] = _np(named[key("classifier.weight")].grad)
                opt.step()
                losses.append(float(loss.detach())); preds.append(pred.cpu().numpy())
            m.after_task(t, None, None, None)
            m.eval()
            with torch.no_grad():
                p, a = m.inference(adapter.batch(tx, ty))
            res[f"test_pred{t}"] = p.cpu().numpy()
            for n in tr:
                res[f"{n}@{t}"] = _np(named[key(n)])
    res["losses"] = np.asarray(losses, np.float64)
    res["preds"] = np.stack(preds)
    return res


# ------------------------------------------------------------------------------------------ InfLoRA
def _inflora_state(tag):
    c = INF_CFG
    D = CFG["dim"]
    P = backbone_params(tag, c["rank"])
    b = 1.0 / np.sqrt(D)
    for t in range(c["task_num"]):
        n = c["init"] if t == 0 else c["inc"]
        P.update(det_extra(tag, {f"classifier_pool.{t}.weight": (n, D), f"classifier_pool.{t}.bias": (n,)}, -b, b))
    return P


def scenario_inflora(adapter):
    """2 tasks of InfLoRA_OPT: before_task (input Gram -> SVD -> lora_A), 2 SGD steps on lora_B + the task head,
    after_task (merge into qkv, DualGPM feature update), inference.  Sign-invariant quantities only (SVD bases)."""
    c = INF_CFG
    tag = "inflora"
    P = _inflora_state(tag)
    D, depth = CFG["dim"], CFG["depth"]
    xs = [det_images(f"{tag}/x{i}", c["bs"]) for i in range(4)]
    ys = [det_labels(f"{tag}/y{i}", c["bs"], 0 if i < 2 else c["init"], c["init"] if i < 2 else c["init"] + c["inc"]) for i in range(4)]
    tx, ty = det_images(tag + "/tx", c["bs"]), det_labels(tag + "/ty", c["bs"], 0, c["init"] + c["inc"])
    res, losses, preds = {}, [], []
    blocks = [f"feat.transformer.blocks.{i}." for i in range(depth)]

    # fixtures stay small: D x D quantities are stored through a fixed 6-column probe (sign-invariant products only)
    probe = detrand.uniform(tag + "/probe", (D, 6), -1.0, 1.0)

    def record(t, get, feature_list, project_type):
        for i, b in enumerate(blocks):
            A = _np(get(b + "attn.lora_A_k.weight"))
            res[f"AtA{i}@{t}"] = A.T @ (A @ probe)
            res[f"qkv{i}@{t}"] = _np(get(b + "attn.qkv.weight")) @ probe
            f = np.asarray(feature_list[i], np.float64)
            res[f"proj{i}@{t}"] = f @ (f.T @ probe)
            res[f"rank{i}@{t}"] = np.asarray(f.shape[1])
        res[f"ptype@{t}"] = np.asarray([p == "retain" for p in project_type])

    if adapter.kind == "oracle":
        P = {k: v.clone() for k, v in P.items()}
        m = ov.InfLoRA(P, CFG, c["init"], c["inc"], c["task_num"], c["lame"], c["lamb"], c["rank"])
        from .methods import SGD
        for t in range(2):
            bx = [xs[2 * t], xs[2 * t + 1]]
            m.before_task(t, bx)
            opt = SGD(m.parameters(), c["lr"], c["momentum"], 0.0)
            for i in (2 * t, 2 * t + 1):
                pred, acc, loss = m.observe(xs[i], ys[i])
                opt.zero_grad(); loss.backward(); opt.step()
                losses.append(float(loss.detach())); preds.append(pred.numpy())
            m.after_task(t, bx)
            record(t, lambda n: m.P[n], m.feature_list, m.project_type)
            p, a = m.inference(tx, ty)
            res[f"test_pred{t}"] = p.numpy()
            res[f"head{t}@{t}"] = _np(m.P[f"classifier_pool.{t}.weight"])
    else:
        import os
        os.environ.setdefault("PYTHONHASHSEED", "0")
        ns = adapter.ns
        bb = ns.make_vit(CFG, "MultiHeadAttention_LoRA", c["rank"])
        m = ns.InfLoRA_OPT(bb, adapter.device, init_cls_num=c["init"], inc_cls_num=c["inc"], task_num=c["task_num"], lame=c["lame"],
                           lamb=c["lamb"], dataset="cifar100", use_ca=False, embd_dim=D)
        m._network.load_state_dict({("backbone." + k if k.startswith("feat.") else k): v.clone() for k, v in P.items()}, strict=True)
        m._network.to(adapter.device)
        named = lambda: dict(m._network.named_parameters())
        test_loader = ListLoader([(tx, ty)], c["bs"], types.SimpleNamespace(trfms=None))
        for t in range(2):
            loader = ListLoader([(xs[2 * t], ys[2 * t]), (xs[2 * t + 1], ys[2 * t + 1])], c["bs"], types.SimpleNamespace(trfms=None))
            m.before_task(t, None, loader, [test_loader])
            opt = adapter.optim("SGD", m.get_parameters({}), lr=c["lr"], momentum=c["momentum"])
            for i in (2 * t, 2 * t + 1):
                m.train()
                pred, acc, loss = m.observe(adapter.batch(xs[i], ys[i]))
                opt.zero_grad(); loss.backward(); opt.step()
                losses.append(float(loss.detach())); preds.append(pred.cpu().numpy())
            m.after_task(t, None, loader, [test_loader])
            nm = named()
            record(t, lambda n: nm["backbone." + n].detach().cpu().double(), m.feature_list, m.project_type)
            m.eval()
            with torch.no_grad():
                p, a = m.inference(adapter.batch(tx, ty))
            res[f"test_pred{t}"] = p.cpu().numpy()
            res[f"head{t}@{t}"] = _np(nm[f"classifier_pool.{t}.weight"])
    res["losses"] = np.asarray(losses, np.float64)
    res["preds"] = np.stack(preds)
    return res


# ------------------------------------------------------------------------------- InfLoRA (original, multi-branch)
# the reference resizes the inputs of its Gram passes to 224 (InfLoRA.py:147, 194) whatever the model: the fixture model
# therefore takes 224 x 224 images (4 x 4 patches of 56 pixels), everything else as small as the other ViT fixtures
ORIG_VIT = dict(img=224, patch=56, dim=128, depth=2, heads=2, mlp=512)
ORIG_CFG = dict(inc=3, total_sessions=3, lame=0.9, lamb=0.6, rank=4, bs=4, lr=0.05, momentum=0.9)


def _timm_name(k):
    """oracle / transformer.py parameter name -> timm tree of SiNet_vit.image_encoder"""
    k = k.replace("feat.transformer.blocks.", "blocks.").replace("feat.", "").replace(".ln_1.", ".norm1.").replace(".ln_2.", ".norm2.")
    return "image_encoder." + k


def _orig_state(tag):
    c, D = ORIG_CFG, ORIG_VIT["dim"]
    P = {k: v.to(fx._DTYPE[0]) for k, v in ov.det_params(ORIG_VIT, tag, 0, torch.float64).items()}
    b = 1.0 / np.sqrt(D)
    for i in range(ORIG_VIT["depth"]):
        blk = f"feat.transformer.blocks.{i}.attn."
        for t in range(c["total_sessions"]):
            P.update(det_extra(tag, {f"{blk}lora_A_k.{t}.weight": (c["rank"], D), f"{blk}lora_A_v.{t}.weight": (c["rank"], D)}, -b, b))
            P[f"{blk}lora_B_k.{t}.weight"] = torch.zeros(D, c["rank"], dtype=fx._DTYPE[0])
            P[f"{blk}lora_B_v.{t}.weight"] = torch.zeros(D, c["rank"], dtype=fx._DTYPE[0])
    for t in range(c["total_sessions"]):
        P.update(det_extra(tag, {f"classifier_pool.{t}.weight": (c["inc"], D), f"classifier_pool.{t}.bias": (c["inc"],)}, -b, b))
    return P


def scenario_inflora_orig(adapter):
    """2 of 3 tasks of the original InfLoRA: before_task (Gram -> [projection] -> SVD -> this task's lora_A), 2 SGD steps on this
    task's lora_B pair + head, after_task (DualGPM), inference over all heads with all pairs applied."""
    c, V = ORIG_CFG, ORIG_VIT
    tag = "inflora_orig"
    P = _orig_state(tag)
    D, depth = V["dim"], V["depth"]
    xs = [det_images(f"{tag}/x{i}", c["bs"], V) for i in range(4)]
    ys = [det_labels(f"{tag}/y{i}", c["bs"], 0 if i < 2 else c["inc"], c["inc"] if i < 2 else 2 * c["inc"]) for i in range(4)]
    tx, ty = det_images(tag + "/tx", c["bs"], V), det_labels(tag + "/ty", c["bs"], 0, 2 * c["inc"])
    res, losses, preds = {}, [], []
    blocks = [f"feat.transformer.blocks.{i}." for i in range(depth)]
    probe = detrand.uniform(tag + "/probe", (D, 6), -1.0, 1.0)

    def record(t, get, feature_list, project_type):
        for i, b in enumerate(blocks):
            A = _np(get(f"{b}attn.lora_A_k.{t}.weight"))
            res[f"AtA{i}@{t}"] = A.T @ (A @ probe)
            Bk, Bv = _np(get(f"{b}attn.lora_B_k.{t}.weight")), _np(get(f"{b}attn.lora_B_v.{t}.weight"))
            res[f"BAk{i}@{t}"] = Bk @ (A @ probe)                       # sign-invariant: B_t A_t applied to the probe
            res[f"BAv{i}@{t}"] = Bv @ (_np(get(f"{b}attn.lora_A_v.{t}.weight")) @ probe)
            f = np.asarray(feature_list[i], np.float64)
            res[f"proj{i}@{t}"] = f @ (f.T @ probe)
            res[f"rank{i}@{t}"] = np.asarray(f.shape[1])
        res[f"ptype@{t}"] = np.asarray([p == "retain" for p in project_type])

    if adapter.kind == "oracle":
        P = {k: v.clone() for k, v in P.items()}
        m = ov.InfLoRAOrig(P, V, c["inc"], c["total_sessions"], c["lame"], c["lamb"], c["rank"])
        from .methods import SGD
        for t in range(2):
            bx = [xs[2 * t], xs[2 * t + 1]]
            m.before_task(bx)
            opt = SGD(m.parameters(), c["lr"], c["momentum"], 0.0)
            for i in (2 * t, 2 * t + 1):
                pred, acc, loss = m.observe(xs[i], ys[i])
                opt.zero_grad(); loss.backward(); opt.step()
                losses.append(float(loss.detach())); preds.append(pred.numpy())
            m.after_task(bx)
            record(t, lambda n: m.P[n], m.feature_list, m.project_type)
            p, a = m.inference(tx, ty)
            res[f"test_pred{t}"] = p.numpy()
            res[f"head{t}@{t}"] = _np(m.P[f"classifier_pool.{t}.weight"])
    else:
        ns = adapter.ns
        net = ns.make_sinet(V, c["total_sessions"], c["rank"], c["inc"])
        m = ns.InfLoRA(net, 64, 2 * c["inc"], device=adapter.device, inc_cls_num=c["inc"], lame=c["lame"], lamb=c["lamb"],
                       total_sessions=c["total_sessions"])
        own = net.state_dict()
        sd = {(k if k.startswith("classifier_pool.") else _timm_name(k)): v.clone() for k, v in P.items()}
        assert all(k in own for k in sd), [k for k in sd if k not in own][:5]
        net.load_state_dict(sd, strict=False)
        net.to(adapter.device)
        named = lambda: dict(net.named_parameters())
        test_loader = ListLoader([(tx, ty)], c["bs"], types.SimpleNamespace(trfms=None))
        for t in range(2):
            loader = ListLoader([(xs[2 * t], ys[2 * t]), (xs[2 * t + 1], ys[2 * t + 1])], c["bs"], types.SimpleNamespace(trfms=None))
            m.before_task(t, None, loader, [test_loader])
            opt = adapter.optim("SGD", m.get_parameters({}), lr=c["lr"], momentum=c["momentum"])
            for i in (2 * t, 2 * t + 1):
                m.train()
                pred, acc, loss = m.observe(adapter.batch(xs[i], ys[i]))
                opt.zero_grad(); loss.backward(); opt.step()
                losses.append(float(loss.detach())); preds.append(pred.cpu().numpy())
            m.after_task(t, None, loader, [test_loader])
            nm = named()
            record(t, lambda n: nm[_timm_name(n)].detach().cpu().double(), m.feature_list, m.project_type)
            m.eval()
            with torch.no_grad():
                p, a = m.inference(adapter.batch(tx, ty))
            res[f"test_pred{t}"] = p.cpu().numpy()
            res[f"head{t}@{t}"] = _np(nm[f"classifier_pool.{t}.weight"])
    res["losses"] = np.asarray(losses, np.float64)
    res["preds"] = np.stack(preds)
    return res


# ---------------------------------------------------------------- full geometry: ViT-B/16, 224 x 224
# The small fixtures above cannot reach the 12-head packing of the attention kernels, the ragged 197- / 222-token key loops or the 768 /
# 3072-wide GEMM tilings.  This scenario runs the same plugin classes ONCE at the geometry of BASELINE.json's ViT configurations -- batch 2,
# deterministic weights -- and keeps only small summaries (the cls features, per-tensor norms and corner blocks of the gradients), so that the
# reference-generated fixture stays well below 1 MB.
FULL_L2P = dict(pool=10, length=5, top_k=5, total=100, init=10, inc=10, task_num=10, coeff=1.0)


def _summ(res, name, g):
    """norm + an 8 x 8 corner (of the last two dimensions' leading block) of one gradient tensor"""
    g = np.asarray(g, np.float64)
    res[name + "/norm"] = np.asarray([np.linalg.norm(g)], np.float64)
    g2 = g.reshape(-1, g.shape[-1])
    res[name + "/corner"] = g2[:8, :8].copy()
    res[name + "/rownorm"] = np.linalg.norm(g2, axis=1)[:64].copy()          # (row SUMS of the prompt gradient vanish: it leaves a LayerNorm)


def full_lora_inputs():
    cfg = ov.VIT_B16
    P = {k: v.to(torch.float64) for k, v in ov.det_params(cfg, "full/lora", 10, torch.float64).items()}
    x = fx._t(detrand.uniform("full/lora/x", (2, 3, cfg["img"], cfg["img"]), -1.0, 1.0)).to(torch.float64)
    cw = torch.from_numpy(np.asarray(detrand.uniform("full/lora/cw", (2, cfg["dim"]), -1.0, 1.0))).to(torch.float64)
    for k in P:
        if "lora_B" in k:
            P[k] = torch.from_numpy(np.asarray(detrand.uniform("full/lora/B/" + k, tuple(P[k].shape), -0.1, 0.1))).to(torch.float64)
    return cfg, P, x, cw


def full_l2p_inputs():
    """keys built so that every sample votes for the even pool entries by a wide margin (no tie at the vote's cut); the query feature that
    places them comes from the oracle's fp64 forward (a deterministic function of the tagged weights)"""
    cfg, c = ov.VIT_B16, FULL_L2P
    D = cfg["dim"]
    P = {k: v.to(torch.float64) for k, v in ov.det_params(cfg, "full/l2p", 0, torch.float64).items()}
    x = fx._t(detrand.uniform("full/l2p/x", (2, 3, cfg["img"], cfg["img"]), -1.0, 1.0)).to(torch.float64)
    y = torch.tensor([3, 7])
    with torch.no_grad():
        q = ov.cls_features(P, x, cfg)
    qm = torch.nn.functional.normalize(q.mean(0), dim=0)
    noise = torch.from_numpy(np.asarray(detrand.uniform("full/l2p/key", (c["pool"], D), -0.02, 0.02))).to(torch.float64)
    sign = torch.tensor([1.0 if j % 2 == 0 else -1.0 for j in range(c["pool"])], dtype=torch.float64)
    P["prompt.prompt_key"] = sign[:, None] * (1.0 + 0.05 * torch.arange(c["pool"], dtype=torch.float64)[:, None]) * qm[None, :] + noise
    P["prompt.prompt"] = torch.from_numpy(np.asarray(detrand.uniform("full/l2p/prompt", (1, c["pool"], c["length"], D), 0.0, 1.0))).to(torch.float64)
    b = 1.0 / np.sqrt(D)
    P["classifier.weight"] = torch.from_numpy(np.asarray(detrand.uniform("full/l2p/cw", (c["total"], D), -b, b))).to(torch.float64)
    P["classifier.bias"] = torch.from_numpy(np.asarray(detrand.uniform("full/l2p/cb", (c["total"],), -b, b))).to(torch.float64)
    return cfg, P, x, y


def scenario_vit_full(adapter, parts=("lora", "l2p")):
    """the reference's classes -> the fixture; the product -> the GPU test; the oracle (fp64) -> tests/test_oracle_vit_golden.py"""
    if adapter.kind == "oracle":
        return _vit_full_oracle(parts)
    ns, dev = adapter.ns, adapter.device
    dt = fx._DTYPE[0]
    res = {}
    if "lora" in parts:
        cfg, P, x, cw = full_lora_inputs()
        bb = ns.make_vit(cfg, "MultiHeadAttention_LoRA", 10)
        bb.load_state_dict({k: v.to(dt) for k, v in P.items()}, strict=True)
        bb = bb.to(dev)
        _set_lora(bb, True)
        names = [k for k in P if "lora_B" in k]
        for n_, p_ in bb.named_parameters():
            p_.requires_grad_(n_ in names)
        f = bb(x.to(dt).to(dev))
        (f * cw.to(dt).to(dev)).sum().backward()
        res["lora/feat"] = _np(f)
        got = dict(bb.named_parameters())
        for k in names:
            _summ(res, "lora/grad/" + k, _np(got[k].grad))
    if "l2p" in parts:
        cfg, P, x, y = full_l2p_inputs()
        c = FULL_L2P
        bb = ns.make_vit(cfg, "MultiHeadAttention", 0)
        m = ns.L2P(bb, dev, init_cls_num=c["init"], inc_cls_num=c["inc"], num_class=c["total"], task_num=c["task_num"], feat_dim=cfg["dim"],
                   prompt_length=c["length"], pool_size=c["pool"], top_k=c["top_k"], pull_constraint_coeff=c["coeff"])
        sd = {("backbone." + k if not k.startswith("classifier") else k): v.to(dt) for k, v in P.items()}
        m.network.load_state_dict(sd, strict=True)
        m.network.to(dev)
        m.before_task(0, None, None, None)
        m.train()
        pred, acc, loss = m.observe(adapter.batch(x.to(dt).to(dev), y.to(dev)))
        named = dict(m.network.named_parameters())
        res["l2p/loss"] = np.asarray([float(loss.detach())], np.float64)
        res["l2p/pred"] = pred.detach().cpu().numpy().astype(np.int64)
        gp = _np(named["backbone.prompt.prompt"].grad)
        res["l2p/touched"] = (np.abs(gp[0]).reshape(c["pool"], -1).max(1) > 0).astype(np.int64)
        for n_ in ("backbone.prompt.prompt", "backbone.prompt.prompt_key", "classifier.weight"):
            _summ(res, "l2p/grad/" + n_, _np(named[n_].grad))
    return res


def _vit_full_oracle(parts):
    res = {}
    if "lora" in parts:
        cfg, P, x, cw = full_lora_inputs()
        names = [k for k in P if "lora_B" in k]
        Pg = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in P.items()}
        f = ov.cls_features(Pg, x, cfg, lora=True)
        (f * cw).sum().backward()
        res["lora/feat"] = _np(f)
        for k in names:
            _summ(res, "lora/grad/" + k, _np(Pg[k].grad))
    if "l2p" in parts:
        cfg, P, x, y = full_l2p_inputs()
        c = FULL_L2P
        ov.TIES_AT_CUT.clear()
        Po = {k: v.clone() for k, v in P.items()}
        m = ov.L2P(Po, cfg, c["init"], c["inc"], c["total"], c["top_k"], c["coeff"])
        m.before_task(0)
        pred, acc, loss, ids, norm = m.observe(x, y)
        assert not ov.TIES_AT_CUT
        res["l2p/loss"] = np.asarray([float(loss.detach())], np.float64)
        res["l2p/pred"] = pred.numpy().astype(np.int64)
        gp = _np(Po["prompt.prompt"].grad)
        res["l2p/touched"] = (np.abs(gp[0]).reshape(c["pool"], -1).max(1) > 0).astype(np.int64)
        for n_, k_ in (("backbone.prompt.prompt", "prompt.prompt"), ("backbone.prompt.prompt_key", "prompt.prompt_key"), ("classifier.weight", "classifier.weight")):
            _summ(res, "l2p/grad/" + n_, _np(Po[k_].grad))
    return res
