"""CPU restatement of the reference's ViT path (TEST INFRASTRUCTURE, see oracle/__init__): ViT-B/16-style
backbone, the L2P prompt pool + method, and InfLoRA_OPT with its LoRA attention.  Plain torch (CPU, dtype of
the inputs), functional: parameters live in a dict keyed by the reference's names.

Reference followed (file:line under /root/reference/core/model):
  backbone/transformer.py:169-197   MultiHeadAttention.forward
  backbone/transformer.py:199-274   MultiHeadAttention_LoRA.{init_param, merge_weight, forward}
  backbone/transformer.py:1255-1273 Mlp
  backbone/transformer.py:1276-1336 ResidualAttentionBlock (pre-LN, eps 1e-5)
  backbone/transformer.py:2006-2018 Transformer.forward (L2P prompt tokens prepended at block 0)
  backbone/transformer.py:2222-2294 VisionTransformer.forward (l2p branch / plain branch, final LN eps 1e-6)
  backbone/vit.py:99-139            ViTZoo.forward
  backbone/prompt.py:369-406        prompt.L2P.forward
  l2p.py:36-122                     L2P
  InfLoRA_opt.py:46-139,141-369     SiNet, InfLoRA_OPT (observe, before_task, after_task, _update_feature)
The third-party pieces the reference takes from timm 0.x (absent here) are restated from their published
definition: PatchEmbed = Conv2d(in, D, p, stride p) -> flatten(2).transpose(1,2); DropPath(0) = identity.

Pinned by tests/golden/{vit_backbone,l2p,inflora}.npz = fp64 runs of the reference itself (oracle/gen_golden.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import detrand

VIT_B16 = dict(img=224, patch=16, dim=768, depth=12, heads=12, mlp=3072)
VIT_TINY = dict(img=32, patch=8, dim=128, depth=2, heads=2, mlp=512)      # the fixture configuration (head dim 64)


def n_patches(cfg):
    return (cfg["img"] // cfg["patch"]) ** 2


def param_shapes(cfg, lora_rank=0):
    """[(name, shape)] in the reference's registration order under ViTZoo (prefix `feat.`)."""
    D, p = cfg["dim"], cfg["patch"]
    out = [("feat.cls_token", (1, 1, D)), ("feat.pos_embed", (1, n_patches(cfg) + 1, D)),
           ("feat.patch_embed.proj.weight", (D, 3, p, p)), ("feat.patch_embed.proj.bias", (D,))]
    for i in range(cfg["depth"]):
        b = f"feat.transformer.blocks.{i}."
        out += [(b + "attn.qkv.weight", (3 * D, D)), (b + "attn.qkv.bias", (3 * D,)),
                (b + "attn.proj.weight", (D, D)), (b + "attn.proj.bias", (D,))]
        if lora_rank:
            out += [(b + "attn.lora_A_k.weight", (lora_rank, D)), (b + "attn.lora_B_k.weight", (D, lora_rank)),
                    (b + "attn.lora_A_v.weight", (lora_rank, D)), (b + "attn.lora_B_v.weight", (D, lora_rank))]
        out += [(b + "ln_1.weight", (D,)), (b + "ln_1.bias", (D,)),
                (b + "mlp.fc1.weight", (cfg["mlp"], D)), (b + "mlp.fc1.bias", (cfg["mlp"],)),
                (b + "mlp.fc2.weight", (D, cfg["mlp"])), (b + "mlp.fc2.bias", (D,)),
                (b + "ln_2.weight", (D,)), (b + "ln_2.bias", (D,))]
    out += [("feat.norm.weight", (D,)), ("feat.norm.bias", (D,))]
    return out


def det_params(cfg, tag, lora_rank=0, dtype=torch.float32):
    """deterministic weights at a trained-network-like scale (so that softmax / LN / GELU are exercised away
    from their trivial regimes); LoRA B starts at zero like init_param, A small random"""
    P = {}
    for name, shp in param_shapes(cfg, lora_rank):
        t = f"{tag}/{name}"
        if "ln_" in name or ".norm." in name:
            a = detrand.uniform(t, shp, 0.8, 1.2) if name.endswith("weight") else detrand.uniform(t, shp, -0.1, 0.1)
        elif "lora_B" in name:
            a = np.zeros(shp)
        elif name.endswith(".bias"):
            a = detrand.uniform(t, shp, -0.05, 0.05)
        elif "cls_token" in name or "pos_embed" in name:
            a = detrand.uniform(t, shp, -0.2, 0.2)
        else:
            fan_in = int(np.prod(shp[1:]))
            s = 1.0 / math.sqrt(fan_in)
            a = detrand.uniform(t, shp, -1.7 * s, 1.7 * s)
        P[name] = torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
    return P


# ------------------------------------------------------------------------------------------ backbone
def patch_embed(P, img, cfg):
    p = cfg["patch"]
    x = F.conv2d(img, P["feat.patch_embed.proj.weight"], P["feat.patch_embed.proj.bias"], stride=p)
    return x.flatten(2).transpose(1, 2)                              # [B, n_patches, D]


def qkv_weight(P, b, lora):
    """effective [3D, D] qkv weight: W_k + B_k A_k, W_v + B_v A_v while LoRA is applied (transformer.py:249-255)"""
    W = P[b + "attn.qkv.weight"]
    if not lora:
        return W
    q, k, v = W.chunk(3, dim=0)
    k = k + P[b + "attn.lora_B_k.weight"] @ P[b + "attn.lora_A_k.weight"]
    v = v + P[b + "attn.lora_B_v.weight"] @ P[b + "attn.lora_A_v.weight"]
    return torch.cat([q, k, v], dim=0)


def attention(P, b, x, heads, lora=False):
    """x [B,N,D] -> [B,N,D] (transformer.py:169-197 / 239-274); qkv bias carries no gradient on the LoRA path
    (`self.qkv.bias.data`) -- irrelevant here because it is frozen in both methods"""
    B, N, D = x.shape
    qkv = F.linear(x, qkv_weight(P, b, lora), P[b + "attn.qkv.bias"]).reshape(B, N, 3, heads, D // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    a = (q @ k.transpose(-2, -1)) * (D // heads) ** -0.5
    a = a.softmax(dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, N, D)
    return F.linear(o, P[b + "attn.proj.weight"], P[b + "attn.proj.bias"])


def block(P, i, x, heads, lora=False, gram=None, eps=1e-5):
    """pre-LN residual block (transformer.py:1331-1336); `gram` collects the attention input's X^T X and token
    count (transformer.py:241-244)"""
    b = f"feat.transformer.blocks.{i}."
    D = x.shape[-1]
    h = F.layer_norm(x, (D,), P[b + "ln_1.weight"], P[b + "ln_1.bias"], eps)
    if gram is not None:
        hd = h.detach()
        gram.append((torch.einsum("bni,bnj->ij", hd, hd), hd.shape[0] * hd.shape[1]))
    x = x + attention(P, b, h, heads, lora)
    h = F.layer_norm(x, (D,), P[b + "ln_2.weight"], P[b + "ln_2.bias"], eps)
    h = F.linear(F.gelu(F.linear(h, P[b + "mlp.fc1.weight"], P[b + "mlp.fc1.bias"])), P[b + "mlp.fc2.weight"], P[b + "mlp.fc2.bias"])
    return x + h


def tokens(P, img, cfg):
    x = patch_embed(P, img, cfg)
    cls = P["feat.cls_token"].expand(x.shape[0], -1, -1)
    x = torch.cat((cls, x), dim=1)
    return x + P["feat.pos_embed"][:, : x.shape[1]]


def encode(P, x, cfg, lora=False, gram=None, acts=None):
    for i in range(cfg["depth"]):
        x = block(P, i, x, cfg["heads"], lora, gram, cfg.get("block_eps", 1e-5))
        if acts is not None:
            acts.append(x)
    D = x.shape[-1]
    return F.layer_norm(x, (D,), P["feat.norm.weight"], P["feat.norm.bias"], 1e-6)


def cls_features(P, img, cfg, lora=False, gram=None, acts=None):
    """the plain forward: final-LN output at the cls token (l2p branch without prompt, transformer.py:2261;
    plain branch + ViTZoo's `out[:,0,:]`, vit.py:128-131)"""
    return encode(P, tokens(P, img, cfg), cfg, lora, gram, acts)[:, 0]


# ---------------------------------------------------------------------------------------------- L2P
def l2p_select(prompt_key, cls_feat, top_k, pool_size):
    """prompt.py:375-391: cosine similarity, per-sample top-k, batch-majority top-k (ties: torch.topk order on
    the count vector; lowest id first here -- the fixtures avoid ties at the cut) -> (ids [top_k], key_norm, q_norm)"""
    kn = F.normalize(prompt_key, p=2, dim=-1, eps=1e-12)
    qn = F.normalize(cls_feat, p=2, dim=-1, eps=1e-12)
    sim = qn @ kn.T
    _, idx = torch.topk(sim, top_k, dim=1)
    counts = torch.bincount(idx.reshape(-1), minlength=pool_size)
    # unique(sorted) + pad(count 0) + topk == top-k of the dense count vector restricted to ids that occur; ids that
    # never occur have count 0 and can only be reached when fewer than top_k ids occur, which per-sample top-k excludes
    order = sorted(range(pool_size), key=lambda j: (-int(counts[j]), j))
    ids = torch.tensor(order[:top_k], dtype=torch.long)
    if top_k < pool_size and int(counts[order[top_k - 1]]) == int(counts[order[top_k]]):
        TIES_AT_CUT.append(counts.tolist())          # tie-break is torch.topk-implementation-defined: fixtures must avoid it
    return ids, kn, qn


TIES_AT_CUT = []


def l2p_forward(P, img, cfg, top_k, ids=None):
    """ViTZoo.forward l2p branch (vit.py:102-118): no-grad query pass, prompt selection, prompted pass.
    -> (feature [B,D] = mean over the prompt-token outputs, reduce_sim, ids)"""
    with torch.no_grad():
        q = cls_features(P, img, cfg)
    prompt, key = P["prompt.prompt"], P["prompt.prompt_key"]
    sel, kn, qn = l2p_select(key.detach(), q, top_k, prompt.shape[1])
    if ids is None:
        ids = sel
    B = img.shape[0]
    kn = F.normalize(key, p=2, dim=-1, eps=1e-12)
    bp = prompt[0, ids].reshape(1, -1, prompt.shape[-1]).expand(B, -1, -1)       # [B, top_k*len, D], same for every sample
    reduce_sim = (kn[ids].unsqueeze(0) * qn.unsqueeze(1)).sum() / B
    x = torch.cat([bp, tokens(P, img, cfg)], dim=1)                              # prompts get no pos-embed
    x = encode(P, x, cfg)
    return x[:, : bp.shape[1]].mean(dim=1), reduce_sim, ids


class L2P:
    """l2p.py:46-122 (functional restatement: `P` holds backbone + prompt + classifier tensors)"""

    def __init__(self, P, cfg, init_cls, inc_cls, total_cls, top_k, coeff):
        self.P, self.cfg = P, cfg
        self.init_cls, self.inc_cls, self.total_cls, self.top_k, self.coeff = init_cls, inc_cls, total_cls, top_k, coeff
        self.cur_task_id, self.known = 0, 0
        self.trainable = ["prompt.prompt", "prompt.prompt_key", "classifier.weight", "classifier.bias"]
        for n, t in P.items():
            t.requires_grad_(n in self.trainable)

    def parameters(self):
        return [self.P[n] for n in self.trainable]

    def before_task(self, t):
        self.cur_task_id = t

    def after_task(self, t):
        self.known += self.init_cls if t == 0 else self.inc_cls

    def logits(self, x):
        feat, rs, ids = l2p_forward(self.P, x, self.cfg, self.top_k)
        return F.linear(feat, self.P["classifier.weight"], self.P["classifier.bias"]), rs, ids

    def observe(self, x, y):
        """masked CE - coeff * reduce_sim, backward and clip_grad_norm_(1.0) INSIDE observe (l2p.py:87-109)"""
        logits, rs, ids = self.logits(x)
        lo = 0 if self.cur_task_id == 0 else self.known
        hi = self.init_cls if self.cur_task_id == 0 else self.known + self.inc_cls
        mask = torch.full_like(logits, float("-inf"))
        mask[:, lo:hi] = 0
        logits = logits + mask
        loss = F.cross_entropy(logits, y) - self.coeff * rs
        for p in self.parameters():
            p.grad = None
        loss.backward()
        norm = torch.nn.utils.clip_grad_norm_(self.parameters(), 1.0)
        pred = logits.argmax(1)
        return pred, (pred == y).sum().item() / x.shape[0], loss.detach(), ids, norm

    def inference(self, x, y):
        with torch.no_grad():
            logits, _, ids = self.logits(x)
        pred = logits.argmax(1)
        return pred, (pred == y).sum().item() / x.shape[0]


# ------------------------------------------------------------------------------------------ InfLoRA
class InfLoRA:
    """InfLoRA_opt.py:141-369 with a ViT backbone (use_ca False).  P: backbone incl. lora_* + `classifier_pool.{t}.*`"""

    def __init__(self, P, cfg, init_cls, inc_cls, task_num, lame, lamb, rank):
        self.P, self.cfg, self.rank = P, cfg, rank
        self.init_cls, self.inc_cls, self.task_num, self.lame, self.lamb = init_cls, inc_cls, task_num, lame, lamb
        self.known, self.cur_task = 0, -1
        self.feature_list, self.project_type = [], []
        self.apply_lora = False
        self.trainable = []

    def _blocks(self):
        return [f"feat.transformer.blocks.{i}." for i in range(self.cfg["depth"])]

    def features(self, x, gram=None):
        return cls_features(self.P, x, self.cfg, self.apply_lora, gram)

    def logits(self, x, inference=False):
        f = self.features(x)
        heads = range(self.cur_task + 1) if inference else [self.cur_task]
        return torch.cat([F.linear(f, self.P[f"classifier_pool.{t}.weight"], self.P[f"classifier_pool.{t}.bias"]) for t in heads], dim=1)

    def observe(self, x, y):
        y = y - self.known
        logits = self.logits(x)
        loss = F.cross_entropy(logits, y)
        pred = logits.argmax(1)
        return pred, (pred == y).sum().item() / y.shape[0], loss

    def inference(self, x, y):
        with torch.no_grad():
            logits = self.logits(x, True)
        pred = logits.argmax(1)
        return pred, (pred == y).sum().item() / y.shape[0]

    def parameters(self):
        return [self.P[n] for n in self.trainable]

    def _gram(self, batches):
        """running mean of X^T X over all tokens of all batches, per attention layer (transformer.py:241-244)"""
        cur = [torch.zeros(self.cfg["dim"], self.cfg["dim"], dtype=next(iter(self.P.values())).dtype) for _ in self._blocks()]
        n = [0] * len(cur)
        with torch.no_grad():
            for x in batches:
                g = []
                self.features(x, g)
                for i, (m, cnt) in enumerate(g):
                    cur[i] = (cur[i] * n[i] + m) / (n[i] + cnt)
                    n[i] += cnt
        return cur

    def before_task(self, t, batches):
        """InfLoRA_opt.py:205-274: B <- 0, A <- top singular vectors of the (projected) input Gram / sqrt(3)"""
        if t == 1:
            self.known = self.init_cls
        elif t > 1:
            self.known += self.inc_cls
        self.cur_task = t
        for b in self._blocks():
            for n in ("attn.lora_B_k.weight", "attn.lora_B_v.weight"):
                self.P[b + n] = torch.zeros_like(self.P[b + n])
        self.apply_lora = True
        self.trainable = [n for n in self.P if "lora_B" in n or n.startswith(f"classifier_pool.{t}.")]
        for n, p in self.P.items():
            p.requires_grad_(n in self.trainable)
        cur = self._gram(batches)                       # B = 0 -> the LoRA branch does not change this pass
        for i, b in enumerate(self._blocks()):
            m = cur[i]
            if t > 0:
                fm = torch.as_tensor(self.feature_list[i] @ self.feature_list[i].T).to(m.dtype)
                m = m - fm @ m if self.project_type[i] == "remove" else fm @ m
            U, _, _ = torch.linalg.svd(m, full_matrices=False)
            A = (U[:, : self.rank].T / math.sqrt(3)).clone()
            self.P[b + "attn.lora_A_k.weight"] = A.clone()
            self.P[b + "attn.lora_A_v.weight"] = A.clone()
        return cur

    def after_task(self, t, batches):
        """merge (transformer.py:228-234) then DualGPM feature update (InfLoRA_opt.py:290-369)"""
        with torch.no_grad():
            for b in self._blocks():
                self.P[b + "attn.qkv.weight"] = qkv_weight(self.P, b, True).detach().clone()
        self.apply_lora = False
        cur = self._gram(batches)
        thr = (self.lame - self.lamb) * t / self.task_num + self.lamb
        for i in range(len(cur)):
            act = cur[i].numpy()
            if t == 0:
                U, S, _ = np.linalg.svd(act, full_matrices=False)
                ratio = (S ** 2) / (S ** 2).sum()
                r = max(np.sum(np.cumsum(ratio) < thr), 1)
                self.feature_list.append(U[:, :r])
                self.project_type.append("remove")
                continue
            _, S, _ = np.linalg.svd(act, full_matrices=False)
            total = (S ** 2).sum()
            fm = self.feature_list[i] @ self.feature_list[i].T
            if self.project_type[i] == "remove":
                hat = act - fm @ act
                U, S, _ = np.linalg.svd(hat, full_matrices=False)
                ratio = (S ** 2) / total
                acc = (total - (S ** 2).sum()) / total
                if acc < thr:
                    r = np.sum(np.cumsum(ratio) + acc < thr) + 1
                    Ui = np.hstack((self.feature_list[i], U[:, :r]))
                    self.feature_list[i] = Ui[:, : min(Ui.shape[0], Ui.shape[1])]
            else:
                hat = fm @ act
                U, S, _ = np.linalg.svd(hat, full_matrices=False)
                ratio = (S ** 2) / total
                acc = (S ** 2).sum() / total
                if acc >= 1 - thr:
                    r = np.sum(acc - np.cumsum(ratio) >= 1 - thr) + 1
                    af = self.feature_list[i] - U[:, :r] @ U[:, :r].T @ self.feature_list[i]
                    U, _, _ = np.linalg.svd(af)
                    self.feature_list[i] = U[:, : self.feature_list[i].shape[1] - r]
        for i in range(len(self.feature_list)):
            f = self.feature_list[i]
            if self.project_type[i] == "remove" and f.shape[1] > f.shape[0] / 2:
                U, _, _ = np.linalg.svd(f)
                self.feature_list[i] = U[:, f.shape[1]:]
                self.project_type[i] = "retain"
        return cur


class Adam:
    """torch.optim.Adam(lr, betas, eps 1e-8, weight_decay 0) restated (the L2P optimizer, l2p-...yaml:38-43)"""

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8):
        self.params, self.lr, self.b1, self.b2, self.eps = list(params), lr, betas[0], betas[1], eps
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0

    def step(self):
        self.t += 1
        with torch.no_grad():
            for p, m, v in zip(self.params, self.m, self.v):
                if p.grad is None:
                    continue
                m.mul_(self.b1).add_(p.grad, alpha=1 - self.b1)
                v.mul_(self.b2).addcmul_(p.grad, p.grad, value=1 - self.b2)
                mh = m / (1 - self.b1 ** self.t)
                vh = v / (1 - self.b2 ** self.t)
                p.sub_(self.lr * mh / (vh.sqrt() + self.eps))


# --------------------------------------------------------------------------------- InfLoRA (original, multi-branch)
class InfLoRAOrig:
    """InfLoRA.py:36-308 over SiNet.py:62-156 / vit_inflora.py:176-263 (timm tree: every LayerNorm has eps 1e-6 ->
    cfg['block_eps']).  P: backbone under the oracle's names, `...attn.lora_{A,B}_{k,v}.{t}.weight` per task and
    `classifier_pool.{t}.*`; the forward applies the pairs of ALL tasks up to the running one."""

    def __init__(self, P, cfg, inc_cls, total_sessions, lame, lamb, rank):
        self.P, self.cfg, self.rank = P, dict(cfg, block_eps=1e-6), rank
        self.inc_cls, self.total_sessions, self.lame, self.lamb = inc_cls, total_sessions, lame, lamb
        self.known = self.total = 0
        self.cur_task = -1
        self.feature_list, self.project_type, self.feature_mat = [], [], []
        self.trainable = []

    def _blocks(self):
        return [f"feat.transformer.blocks.{i}." for i in range(self.cfg["depth"])]

    def _view(self):
        """parameter dict of a single-pair LoRA ViT equivalent to the multi-branch forward of task `cur_task`:
        finished pairs folded (without gradient) into qkv.weight, the running task's pair as the LoRA branch"""
        t, D = self.cur_task, self.cfg["dim"]
        V = dict(self.P)
        for b in self._blocks():
            W = self.P[b + "attn.qkv.weight"].detach().clone()
            for s_ in range(t):
                W[D:2 * D] += (self.P[f"{b}attn.lora_B_k.{s_}.weight"] @ self.P[f"{b}attn.lora_A_k.{s_}.weight"]).detach()
                W[2 * D:] += (self.P[f"{b}attn.lora_B_v.{s_}.weight"] @ self.P[f"{b}attn.lora_A_v.{s_}.weight"]).detach()
            V[b + "attn.qkv.weight"] = W
            for n in ("lora_A_k", "lora_B_k", "lora_A_v", "lora_B_v"):
                V[f"{b}attn.{n}.weight"] = self.P[f"{b}attn.{n}.{max(t, 0)}.weight"]
        return V

    def features(self, x, gram=None):
        return cls_features(self._view(), x, self.cfg, self.cur_task >= 0, gram)

    def _head(self, f, t):
        return F.linear(f, self.P[f"classifier_pool.{t}.weight"], self.P[f"classifier_pool.{t}.bias"])

    def observe(self, x, y):
        y = y - self.known
        logits = self._head(self.features(x), self.cur_task)
        loss = F.cross_entropy(logits, y)
        pred = logits.argmax(1)
        return pred, (pred == y).sum().item() / y.shape[0], loss

    def inference(self, x, y):
        with torch.no_grad():
            f = self.features(x)
            logits = torch.cat([self._head(f, t) for t in range(self.cur_task + 1)], 1)
        pred = logits.argmax(1)
        return pred, (pred == y).sum().item() / y.shape[0]

    def parameters(self):
        return [self.P[n] for n in self.trainable]

    def _gram(self, batches):
        cur = [torch.zeros(self.cfg["dim"], self.cfg["dim"], dtype=next(iter(self.P.values())).dtype) for _ in self._blocks()]
        n = [0] * len(cur)
        with torch.no_grad():
            for x in batches:
                g = []
                self.features(x, g)
                for i, (m, cnt) in enumerate(g):
                    cur[i] = (cur[i] * n[i] + m) / (n[i] + cnt)
                    n[i] += cnt
        return cur

    def before_task(self, batches):
        """InfLoRA.py:108-183"""
        self.known = self.total
        self.cur_task += 1
        self.total = self.known + self.inc_cls
        t = self.cur_task
        self.trainable = [n for n in self.P if n.startswith(f"classifier_pool.{t}.") or f"lora_B_k.{t}." in n or f"lora_B_v.{t}." in n]
        for n, p in self.P.items():
            p.requires_grad_(n in self.trainable)
        cur = self._gram(batches)                       # B_t = 0: the running pair does not change this pass
        for i, b in enumerate(self._blocks()):
            m = cur[i]
            if t > 0:
                fm = self.feature_mat[i].to(m.dtype)
                m = m - fm @ m if self.project_type[i] == "remove" else fm @ m
            U = torch.linalg.svd(m, full_matrices=False)[0]
            A = (U[:, : self.rank].T / math.sqrt(3)).clone()
            self.P[f"{b}attn.lora_A_k.{t}.weight"] = A.clone()
            self.P[f"{b}attn.lora_A_v.{t}.weight"] = A.clone()

    def after_task(self, batches):
        """InfLoRA.py:185-308"""
        cur = self._gram(batches)
        thr = (self.lame - self.lamb) * self.cur_task / self.total_sessions + self.lamb
        first = len(self.feature_list) == 0
        for i in range(len(cur)):
            act = cur[i].numpy()
            if first:
                U, S, _ = np.linalg.svd(act, full_matrices=False)
                r = int(np.sum(np.cumsum(S ** 2 / (S ** 2).sum()) < thr))
                self.feature_list.append(U[:, : max(r, 1)])
                self.project_type.append("remove" if r < act.shape[0] / 2 else "retain")
                continue
            total = (np.linalg.svd(act, compute_uv=False) ** 2).sum()
            f = self.feature_list[i]
            if self.project_type[i] == "remove":
                U, S, _ = np.linalg.svd(act - f @ f.T @ act, full_matrices=False)
                ratio, acc, r = S ** 2 / total, (total - (S ** 2).sum()) / total, 0
                for v in ratio:
                    if acc < thr:
                        acc += v; r += 1
                    else:
                        break
                if r:
                    Ui = np.hstack((f, U[:, :r]))
                    self.feature_list[i] = Ui[:, : Ui.shape[0]] if Ui.shape[1] > Ui.shape[0] else Ui
            else:
                U, S, _ = np.linalg.svd(f @ f.T @ act, full_matrices=False)
                ratio, acc, r = S ** 2 / total, (S ** 2).sum() / total, 0
                for v in ratio:
                    if acc >= 1 - thr:
                        acc -= v; r += 1
                    else:
                        break
                if r:
                    rest = f - U[:, :r] @ U[:, :r].T @ f
                    self.feature_list[i] = np.linalg.svd(rest)[0][:, : f.shape[1] - r]
        for i, f in enumerate(self.feature_list):
            if self.project_type[i] == "remove" and f.shape[1] > f.shape[0] / 2:
                self.feature_list[i] = np.linalg.svd(f)[0][:, f.shape[1]:]
                self.project_type[i] = "retain"
        self.feature_mat = [torch.as_tensor(f @ f.T) for f in self.feature_list]
