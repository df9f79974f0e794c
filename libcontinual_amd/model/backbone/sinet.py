"""SiNet_vit: the backbone of the original InfLoRA plugin (reference core/model/backbone/SiNet.py:62-156 over
core/model/backbone/vit_inflora.py:176-263, 275-296, 343-520) on the HIP ViT executor.

Same module tree and parameter names as the reference (timm layout: `image_encoder.blocks.{i}.norm1 / attn.qkv / attn.proj /
attn.lora_{A,B}_{k,v}.{task} / norm2 / mlp.fc1 / mlp.fc2`, `classifier_pool.{task}`), so its state dicts and the plugin's
name-based freezing carry over.  Every task owns a rank-r pair (A_t, B_t) for k and for v and the forward applies ALL pairs up
to the running task: k += x (sum_{s<=t} B_s A_s)^T (vit_inflora.py:235-239).

On the device that is the InfLoRA_OPT hot path with a different "frozen" weight: the pairs of the finished tasks never change
again, so they are folded once per task into a base qkv matrix  W + sum_{s<t} B_s A_s  and the executor (libclhip
clhip_vit_forward / backward, the same VisionTransformer object the other ViT plugins use) sees a plain rank-r LoRA layer whose
trainable pair is task t's.  The executor instance is private (not a sub-module): its parameters ARE this module's parameters
(same nn.Parameter objects under the executor's own names), except the folded base matrices it owns.
"""
import math
import os

import torch
import torch.nn as nn

from ..heads import HipLinear
from .vit import LazyGram, Mlp, VisionTransformer, _P, _PatchEmbed, lazy_gram_attrs

__all__ = ["Attention_LoRA", "ViT_lora_co", "SiNet_vit"]


class Attention_LoRA(nn.Module):
    """parameter holder + Gram bookkeeping of one attention layer (vit_inflora.py:176-263); the executor runs it"""

    def __init__(self, dim, num_heads=8, r=64, n_tasks=10):
        super().__init__()
        self.dim, self.num_heads, self.rank = dim, num_heads, r
        self.scale = (dim // num_heads) ** -0.5
        self.qkv, self.proj = _P((3 * dim, dim)), _P((dim, dim))
        self.lora_A_k = nn.ModuleList([_P((r, dim), False) for _ in range(n_tasks)])
        self.lora_B_k = nn.ModuleList([_P((dim, r), False) for _ in range(n_tasks)])
        self.lora_A_v = nn.ModuleList([_P((r, dim), False) for _ in range(n_tasks)])
        self.lora_B_v = nn.ModuleList([_P((dim, r), False) for _ in range(n_tasks)])
        # CPU running means of x^T x like the reference's -- read lazily: the per-batch sums stay in the backbone's device buffers
        self.__dict__["_gram_all"], self.__dict__["_gram_cur"] = LazyGram(dim), LazyGram(dim)

    matrix, n_matrix = lazy_gram_attrs("matrix", "n_matrix", "_gram_all")
    cur_matrix, n_cur_matrix = lazy_gram_attrs("cur_matrix", "n_cur_matrix", "_gram_cur")

    def init_param(self):
        for t in range(len(self.lora_A_k)):
            nn.init.kaiming_uniform_(self.lora_A_k[t].weight, a=math.sqrt(5))
            nn.init.kaiming_uniform_(self.lora_A_v[t].weight, a=math.sqrt(5))
            nn.init.zeros_(self.lora_B_k[t].weight)
            nn.init.zeros_(self.lora_B_v[t].weight)

    def get_matrix(self, task):
        return self.lora_B_k[task].weight @ self.lora_A_k[task].weight, self.lora_B_v[task].weight @ self.lora_A_v[task].weight

    @torch.no_grad()
    def get_pre_matrix(self, task):
        """sum over the tasks BEFORE `task` (vit_inflora.py:259-263); zeros when there is none"""
        wk = torch.zeros(self.dim, self.dim, device=self.qkv.weight.device)
        wv = torch.zeros_like(wk)
        for t in range(task):
            wk += self.lora_B_k[t].weight @ self.lora_A_k[t].weight
            wv += self.lora_B_v[t].weight @ self.lora_A_v[t].weight
        return wk, wv


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, n_tasks=10, r=64):
        super().__init__()
        self.norm1 = _P((dim,))
        self.attn = Attention_LoRA(dim, num_heads, r=r, n_tasks=n_tasks)
        self.norm2 = _P((dim,))
        self.mlp = Mlp(dim, int(dim * mlp_ratio))


class ViT_lora_co(nn.Module):
    """SiNet.py:7-35 + vit_inflora.py:343-520.  `forward(x, task_id)` -> ([B, 1, D] final-norm cls token, prompt_loss):
    the callers only read token 0 (SiNet.py:96, 116, 127), which is what the executor produces."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0,
                 n_tasks=10, rank=64, dtype=None, **kwargs):
        super().__init__()
        self.embed_dim = self.num_features = self.out_dim = embed_dim
        self.depth, self.n_tasks, self.rank = depth, n_tasks, rank
        self.patch_embed = _PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        n_patch = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.cls_token_grow = nn.Parameter(torch.zeros(1, 5000, embed_dim))             # kept for state-dict parity (vit_inflora.py:389-391)
        self.pos_embed = nn.Parameter(torch.zeros(1, n_patch + 1, embed_dim))
        self.pos_embed_grow = nn.Parameter(torch.zeros(1, n_patch + 1000, embed_dim))
        self.blocks = nn.Sequential(*[Block(embed_dim, num_heads, mlp_ratio, n_tasks, rank) for _ in range(depth)])
        self.norm = _P((embed_dim,))
        self.head = _P((num_classes, embed_dim)) if num_classes > 0 else None            # unused by SiNet (kept for checkpoints)
        ex = VisionTransformer(img_size=img_size, patch_size=patch_size, embed_dim=embed_dim, depth=depth, num_heads=num_heads,
                               attn_layer="MultiHeadAttention_LoRA", mlp_ratio=mlp_ratio, lora_rank=rank,
                               dtype=dtype or os.environ.get("CLHIP_DTYPE", "bf16"))
        ex.block_ln_eps = 1e-6                       # timm's norm_layer = LayerNorm(eps=1e-6) in every block (vit_inflora.py:375)
        self.__dict__["_ex"] = ex                    # private executor: not a sub-module, so its names never show up
        self.__dict__["_fold_sig"] = None
        self._share_parameters()
        self.reset_parameters()

    # -------------------------------------------------------------------------------------- init (vit_inflora.py:427-446)
    def reset_parameters(self):
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.trunc_normal_(self.pos_embed_grow, std=.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        nn.init.normal_(self.cls_token_grow, std=1e-6)
        w = self.patch_embed.proj.weight
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        nn.init.uniform_(self.patch_embed.proj.bias, -1 / math.sqrt(w[0].numel()), 1 / math.sqrt(w[0].numel()))
        for blk in self.blocks:
            for lin in (blk.attn.qkv, blk.attn.proj, blk.mlp.fc1, blk.mlp.fc2):
                nn.init.trunc_normal_(lin.weight, std=.02)
                nn.init.zeros_(lin.bias)
            for ln in (blk.norm1, blk.norm2):
                nn.init.ones_(ln.weight); nn.init.zeros_(ln.bias)
            for t in range(self.n_tasks):
                for ml in (blk.attn.lora_A_k, blk.attn.lora_A_v, blk.attn.lora_B_k, blk.attn.lora_B_v):
                    nn.init.trunc_normal_(ml[t].weight, std=.02)
        nn.init.ones_(self.norm.weight); nn.init.zeros_(self.norm.bias)
        if self.head is not None:
            nn.init.zeros_(self.head.weight); nn.init.zeros_(self.head.bias)

    # --------------------------------------------------------------------------------------------- executor plumbing
    def _share_parameters(self):
        """hand the executor THIS module's Parameter objects under its own names (the folded qkv base stays its own)"""
        ex = self._ex
        ex._parameters["cls_token"], ex._parameters["pos_embed"] = self.cls_token, self.pos_embed
        for name in ("weight", "bias"):
            ex.patch_embed.proj._parameters[name] = self.patch_embed.proj._parameters[name]
            ex.norm._parameters[name] = self.norm._parameters[name]
        for mine, his in zip(self.blocks, ex.transformer.blocks):
            for a, b in ((mine.norm1, his.ln_1), (mine.norm2, his.ln_2), (mine.attn.proj, his.attn.proj), (mine.mlp.fc1, his.mlp.fc1),
                         (mine.mlp.fc2, his.mlp.fc2)):
                for name in ("weight", "bias"):
                    b._parameters[name] = a._parameters[name]
            his.attn.qkv._parameters["bias"] = mine.attn.qkv.bias
            his.attn.qkv._parameters["weight"] = nn.Parameter(torch.zeros_like(mine.attn.qkv.weight), requires_grad=False)

    @torch.no_grad()
    def _select_task(self, task_id):
        """point the executor's trainable pair at task `task_id` and (re)fold the finished tasks into its base qkv weights"""
        ex = self._ex
        lora_on = task_id > -0.5
        t = int(task_id) if lora_on else 0
        sig_parts = [lora_on, t]
        for mine, his in zip(self.blocks, ex.transformer.blocks):
            a = mine.attn
            for ml, name in ((a.lora_A_k, "lora_A_k"), (a.lora_B_k, "lora_B_k"), (a.lora_A_v, "lora_A_v"), (a.lora_B_v, "lora_B_v")):
                getattr(his.attn, name)._parameters["weight"] = ml[t].weight
            his.attn.apply_lora = lora_on
            sig_parts.append((a.qkv.weight.data_ptr(), a.qkv.weight._version))
            for s in range(t if lora_on else 0):
                sig_parts += [(ml[s].weight.data_ptr(), ml[s].weight._version) for ml in (a.lora_A_k, a.lora_B_k, a.lora_A_v, a.lora_B_v)]
        sig = tuple(sig_parts)
        if sig == self._fold_sig:
            return
        D = self.embed_dim
        for mine, his in zip(self.blocks, ex.transformer.blocks):
            a, w = mine.attn, mine.attn.qkv.weight
            base = his.attn.qkv.weight
            if base.device != w.device:
                base = nn.Parameter(torch.empty_like(w), requires_grad=False)
                his.attn.qkv._parameters["weight"] = base
            base.copy_(w)
            if lora_on and t > 0:
                wk, wv = a.get_pre_matrix(t)
                base[D:2 * D] += wk
                base[2 * D:3 * D] += wv
        self.__dict__["_fold_sig"] = sig

    def forward(self, x, task_id, register_blk=-1, get_feat=False, get_cur_feat=False):
        ex = self._ex
        self._select_task(task_id)
        want_gram = get_feat or get_cur_feat
        gram = None
        if want_gram:
            # the executor ADDS every layer's X^T X into a resident [depth, D, D] buffer (one MFMA launch per forward); the running means of
            # vit_inflora.py:205-212 are taken when `matrix` / `cur_matrix` are read.  Both at once (never in the reference's plugin): the batch's
            # sums go through a scratch buffer and are added to both.
            both = get_feat and get_cur_feat
            gram = self._gram_buffer("_tmp" if both else ("_gram_all" if get_feat else "_gram_cur"), x.device)
            if both:
                gram.zero_()
        feat = ex.features(x, None, get_input_matrix=want_gram, gram_out=gram)
        if want_gram:
            n = x.shape[0] * (self.patch_embed.num_patches + 1)
            for slot, on in (("_gram_all", get_feat), ("_gram_cur", get_cur_feat)):
                if not on:
                    continue
                if both:
                    self._gram_buffer(slot, x.device).add_(gram)
                for blk in self.blocks:
                    blk.attn.__dict__[slot].n_dev += n
        prompt_loss = torch.zeros((1,), device=feat.device, requires_grad=True)
        return feat.unsqueeze(1), prompt_loss

    def _gram_buffer(self, slot, dev):
        """resident [depth, D, D] fp32 sums behind the blocks' `matrix` (slot "_gram_all") / `cur_matrix` ("_gram_cur") attributes"""
        bufs = self.__dict__.setdefault("_gram_bufs", {})
        b = bufs.get(slot)
        if b is not None and slot != "_tmp":             # (a deep copy of the module carries the buffers but detached holders)
            g0 = self.blocks[0].attn.__dict__[slot]
            if g0.dev is None or g0.dev.data_ptr() != b[0].data_ptr():
                b = None
        if b is None or b.device != dev:
            if slot != "_tmp":
                for blk in self.blocks:
                    blk.attn.__dict__[slot].fold()
            b = bufs[slot] = torch.zeros(self.depth, self.embed_dim, self.embed_dim, device=dev)
            if slot != "_tmp":
                for i, blk in enumerate(self.blocks):
                    g = blk.attn.__dict__[slot]
                    g.dev, g.n_dev = b[i], 0
        return b

    def load_timm_state_dict(self, sd):
        """timm `vit_base_patch16_224_in21k` keys are this module's keys: load what matches in name and shape"""
        own = self.state_dict()
        self.load_state_dict({k: v for k, v in sd.items() if k in own and tuple(v.shape) == tuple(own[k].shape)}, strict=False)


class SiNet_vit(nn.Module):
    """SiNet.py:62-156.  kwargs beyond the reference's (total_sessions, rank, init_cls, embd_dim): `pretrained` / `checkpoint`
    (there is no network here: the reference downloads vit_base_patch16_224_in21k, this loads a local file) and the
    ViT geometry (img_size, patch_size, depth, num_heads, dtype) so tests can build a small one."""

    def __init__(self, **args):
        super().__init__()
        geo = {k: args[k] for k in ("img_size", "patch_size", "depth", "num_heads", "dtype") if k in args}
        self.image_encoder = ViT_lora_co(embed_dim=args["embd_dim"], n_tasks=args["total_sessions"], rank=args["rank"], **geo)
        if args.get("pretrained", False):
            path = args.get("checkpoint") or os.environ.get("CLHIP_VIT_CHECKPOINT")
            if not path or not os.path.exists(path):
                raise FileNotFoundError("SiNet_vit(pretrained=True) needs backbone.kwargs.checkpoint or $CLHIP_VIT_CHECKPOINT (no network)")
            sd = torch.load(path, map_location="cpu")
            self.image_encoder.load_timm_state_dict(sd.get("state_dict", sd.get("model", sd)) if isinstance(sd, dict) else sd)
        self.class_num = args["init_cls"]
        self.classifier_pool = nn.ModuleList([HipLinear(args["embd_dim"], self.class_num, bias=True) for _ in range(args["total_sessions"])])
        self.classifier_pool_backup = nn.ModuleList([HipLinear(args["embd_dim"], self.class_num, bias=True) for _ in range(args["total_sessions"])])
        self.numtask = 0

    @property
    def feature_dim(self):
        return self.image_encoder.out_dim

    def extract_vector(self, image, task=None):
        feats, _ = self.image_encoder(image, self.numtask - 1 if task is None else task)
        return feats[:, 0, :]

    def forward(self, image, get_feat=False, get_cur_feat=False, fc_only=False):
        if fc_only:
            return torch.cat([self.classifier_pool[ti](image) for ti in range(self.numtask)], dim=1)
        feats, prompt_loss = self.image_encoder(image, task_id=self.numtask - 1, get_feat=get_feat, get_cur_feat=get_cur_feat)
        feats = feats[:, 0, :].reshape(feats.size(0), -1)
        return {"logits": self.classifier_pool[self.numtask - 1](feats), "features": feats, "prompt_loss": prompt_loss}

    def interface(self, image):
        feats, _ = self.image_encoder(image, task_id=self.numtask - 1)
        feats = feats[:, 0, :].reshape(feats.size(0), -1)
        return torch.cat([head(feats) for head in self.classifier_pool[: self.numtask]], 1)

    def update_fc(self, nb_classes):
        self.numtask += 1

    def classifier_backup(self, task_id):
        self.classifier_pool_backup[task_id].load_state_dict(self.classifier_pool[task_id].state_dict())

    def freeze(self):
        for q in self.parameters():
            q.requires_grad = False
        return self.eval()
