"""ViT backbone of the L2P / InfLoRA_OPT path on the HIP executor (clhip_vit_*).

Mirror of the reference's object graph and parameter names so that checkpoints, YAML
kwargs and plugin code written against it keep working:

    ViTZoo                     core/model/backbone/vit.py:43-139     (.feat, .prompt, .prompt_flag, create_prompt, forward)
      feat: VisionTransformer  core/model/backbone/transformer.py:2147-2294
        patch_embed.proj, cls_token, pos_embed, transformer.blocks[i].{ln_1, attn.{qkv, proj[, lora_*]}, ln_2, mlp.{fc1, fc2}}, norm
    MultiHeadAttention_LoRA    transformer.py:199-274   (apply_lora, init_param, merge_weight, reset_input_matrix, cur_matrix)
    L2PPrompt                  core/model/backbone/prompt.py:345-406 (prompt, prompt_key)

The modules below only OWN parameters (fp32 masters on the device); they have no forward of their own.  All
compute happens in `VisionTransformer.features()`: ONE C call for the whole forward and ONE for the whole backward
(csrc/vit_plan.hip), wrapped in a single autograd.Function per use (plain / LoRA, L2P-prompted).  What differs from
the reference by design: tokens stay batch-first [B*N, D] in bf16 (or fp32 parity mode) with no permutes; the frozen
weights are kept as compute-dtype copies in both orientations so the backward is GEMMs only; LoRA's B gradient uses
the rank-r shortcut instead of a dense [3D, D] dW; the prompt vote, gather, pull loss and key gradient are one kernel.
"""
import ctypes as C
import math
import os

import torch
import torch.nn as nn

from ... import _lib
from ..._lib import call, require_gpu

_DT = {"bf16": (_lib.BF16, torch.bfloat16), "f32": (_lib.F32, torch.float32)}


def _st():
    return torch.cuda.current_stream().cuda_stream


class _P(nn.Module):
    """parameter holder with nn.Linear-style attribute names"""

    def __init__(self, w_shape, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*w_shape))
        self.bias = nn.Parameter(torch.zeros(w_shape[0])) if bias else None

    def forward(self, *a, **k):
        raise RuntimeError("parameter holder: the HIP executor (VisionTransformer.features) runs the layer")


class LazyGram:
    """running mean of X^T X over the tokens seen so far (transformer.py:241-244) whose per-batch sums stay ON THE DEVICE: the executor adds
    every batch's X^T X into `dev` (a [D, D] fp32 view of the backbone's resident [depth, D, D] buffer) and `n_dev` counts the tokens in it;
    the host tensor the reference keeps (`cur_matrix`, the SVD's input) is brought up to date when somebody READS it --
    (host * n_host + dev) / (n_host + n_dev), the reference's per-batch update with all pending batches as one -- i.e. one transfer per
    task boundary instead of one per batch and layer."""

    def __init__(self, dim):
        self.host, self.n_host = torch.zeros(dim, dim), 0
        self.dev, self.n_dev = None, 0

    def fold(self):
        if self.n_dev:
            self.host = (self.host * self.n_host + self.dev.cpu()) / (self.n_host + self.n_dev)
            self.n_host += self.n_dev
            self.drop_pending()

    def drop_pending(self):
        if self.n_dev:
            self.dev.zero_()
        self.n_dev = 0

    def __deepcopy__(self, memo):          # a copied module starts from the folded host state (device views are never shared)
        self.fold()
        c = LazyGram(self.host.shape[0])
        c.host, c.n_host = self.host.clone(), self.n_host
        return c


def lazy_gram_attrs(matrix_name, count_name, slot):
    """(matrix, count) properties over a LazyGram kept in `self.__dict__[slot]`: reads fold the pending device sums, `x = zeros` /
    `n = 0` drop them -- the plugin code written against the reference's plain attributes runs unchanged"""

    def get_m(self):
        g = self.__dict__[slot]
        g.fold()
        return g.host

    def set_m(self, v):
        g = self.__dict__[slot]
        g.drop_pending()
        g.host = v

    def get_n(self):
        g = self.__dict__[slot]
        return g.n_host + g.n_dev

    def set_n(self, v):
        g = self.__dict__[slot]
        if v == 0:
            g.drop_pending()
        else:
            g.fold()
        g.n_host = v

    return property(get_m, set_m), property(get_n, set_n)


class MultiHeadAttention(nn.Module):
    def __init__(self, dim, num_heads, **kw):
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = _P((3 * dim, dim))
        self.proj = _P((dim, dim))


class MultiHeadAttention_LoRA(MultiHeadAttention):
    """transformer.py:199-274: LoRA on k and v; `cur_matrix` is the running mean of the attention input's Gram"""

    def __init__(self, dim, num_heads, lora_rank=10, lora_bias=False, **kw):
        super().__init__(dim, num_heads)
        assert not lora_bias
        self.lora_rank = lora_rank
        self.lora_A_k, self.lora_B_k = _P((lora_rank, dim), False), _P((dim, lora_rank), False)
        self.lora_A_v, self.lora_B_v = _P((lora_rank, dim), False), _P((dim, lora_rank), False)
        self.apply_lora = False
        self.__dict__["_gram"] = LazyGram(dim)           # `cur_matrix` (CPU, the SVD input, like the reference) / `n_cur_matrix` live in it

    cur_matrix, n_cur_matrix = lazy_gram_attrs("cur_matrix", "n_cur_matrix", "_gram")

    def init_param(self):
        nn.init.kaiming_uniform_(self.lora_A_k.weight, a=math.sqrt(5))
        nn.init.kaiming_uniform_(self.lora_A_v.weight, a=math.sqrt(5))
        nn.init.zeros_(self.lora_B_k.weight)
        nn.init.zeros_(self.lora_B_v.weight)
        self.apply_lora = True

    @torch.no_grad()
    def merge_weight(self):
        w = self.qkv.weight
        require_gpu(w)
        call("clhip_lora_merge", w.data_ptr(), self.lora_A_k.weight.data_ptr(), self.lora_B_k.weight.data_ptr(),
             self.lora_A_v.weight.data_ptr(), self.lora_B_v.weight.data_ptr(), self.dim, self.lora_rank, _st())
        w.add_(0)                                        # bump the version counter: the executor refreshes its copies
        self.apply_lora = False

    def reset_input_matrix(self):
        self.n_cur_matrix = 0                            # (drops the pending device sums first: nothing is transferred for a reset)
        self.cur_matrix.zero_()


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1, self.fc2 = _P((hidden, dim)), _P((dim, hidden))


class ResidualAttentionBlock(nn.Module):
    def __init__(self, dim, heads, mlp_ratio, attn_layer, **kw):
        super().__init__()
        self.attn = attn_layer(dim, heads, **kw)
        self.ln_1 = _P((dim,))
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.ln_2 = _P((dim,))


class Transformer(nn.Module):
    def __init__(self, dim, depth, heads, mlp_ratio, attn_layer, **kw):
        super().__init__()
        self.blocks = nn.ModuleList([ResidualAttentionBlock(dim, heads, mlp_ratio, attn_layer, **kw) for _ in range(depth)])


class _PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = _P((embed_dim, in_chans, patch_size, patch_size))


_ATTN = {"MultiHeadAttention": MultiHeadAttention, "MultiHeadAttention_LoRA": MultiHeadAttention_LoRA}


class _Scratch:
    """non-module state (C handle, device buffers): never copied / moved with the module, rebuilt on demand"""

    def __init__(self):
        self.handle = None
        self.shadow = None
        self.ws = None
        self.ws_key = None
        self.sig = None
        self.cparams = None
        self.keep = None
        self.gram = None

    def __deepcopy__(self, memo):
        return _Scratch()


class _VitFn(torch.autograd.Function):
    """features = ViT(images [, prompt tokens]); backward -> d prompt tokens and / or d lora_B of every layer"""

    @staticmethod
    def forward(ctx, vit, images, prompt_tokens, gram, need, *lora_b):
        feat = vit._run_forward(images, prompt_tokens, need, gram)
        ctx.vit, ctx.n_lora, ctx.has_prompt = vit, len(lora_b), prompt_tokens is not None
        ctx.token = vit._fwd_token
        ctx.lora_shapes = [tuple(b.shape) for b in lora_b]
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        vit = ctx.vit
        if vit._fwd_token != ctx.token:
            raise RuntimeError("the ViT workspace was overwritten by a later forward before this backward ran")
        dprompt, dlora = vit._run_backward(dfeat, ctx.has_prompt, ctx.n_lora > 0)
        return (None, None, dprompt, None, None) + tuple(dlora if dlora else ())


class VisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12, attn_layer="MultiHeadAttention",
                 mlp_ratio=4.0, dtype="bf16", lora_rank=0, **kwargs):
        super().__init__()
        assert in_chans == 3
        if isinstance(attn_layer, str):
            if attn_layer not in _ATTN:
                raise NotImplementedError(f"attn_layer {attn_layer} is outside the hot-path scope (SURVEY.md section 8)")
            attn_layer = _ATTN[attn_layer]
        self.img_size, self.patch_size, self.embed_dim, self.depth, self.num_heads = img_size, patch_size, embed_dim, depth, num_heads
        self.num_features = embed_dim
        self.mlp_dim = int(embed_dim * mlp_ratio)
        self.lora_rank = lora_rank if attn_layer is MultiHeadAttention_LoRA else 0
        self.compute_dtype = dtype
        self.patch_embed = _PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        kw = {"lora_rank": lora_rank} if self.lora_rank else {}
        self.transformer = Transformer(embed_dim, depth, num_heads, mlp_ratio, attn_layer, **kw)
        self.norm = _P((embed_dim,))
        self._s = _Scratch()
        self._fwd_token = 0
        self.reset_parameters()

    # ------------------------------------------------------------------ init (transformer.py:2201-2213, timm PatchEmbed)
    def reset_parameters(self):
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.trunc_normal_(self.cls_token, std=.02)
        w = self.patch_embed.proj.weight
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        bound = 1 / math.sqrt(w[0].numel())
        nn.init.uniform_(self.patch_embed.proj.bias, -bound, bound)
        for blk in self.transformer.blocks:
            for lin in (blk.attn.qkv, blk.attn.proj, blk.mlp.fc1, blk.mlp.fc2):
                nn.init.trunc_normal_(lin.weight, std=.02)
                nn.init.zeros_(lin.bias)
            for ln in (blk.ln_1, blk.ln_2):
                nn.init.ones_(ln.weight); nn.init.zeros_(ln.bias)
            if self.lora_rank:
                for n in ("lora_A_k", "lora_B_k", "lora_A_v", "lora_B_v"):
                    nn.init.trunc_normal_(getattr(blk.attn, n).weight, std=.02)
        nn.init.ones_(self.norm.weight); nn.init.zeros_(self.norm.bias)

    # ------------------------------------------------------------------------------------ executor state
    def attention_modules(self):
        return [b.attn for b in self.transformer.blocks]

    def _frozen_tensors(self):
        out = [self.patch_embed.proj.weight]
        for b in self.transformer.blocks:
            out += [b.attn.qkv.weight, b.attn.proj.weight, b.mlp.fc1.weight, b.mlp.fc2.weight]
        return out

    def _all_tensors(self):
        return [p for p in self.parameters()]

    def _ensure(self, dev):
        s = self._s
        if s.handle is None:
            desc = _lib.VitDesc(self.img_size, self.patch_size, self.embed_dim, self.depth, self.num_heads, self.mlp_dim, self.lora_rank,
                                float(getattr(self, "block_ln_eps", 0.0)))
            h = _lib.lib().clhip_vit_create(C.byref(desc), _DT[self.compute_dtype][0])
            if not h:
                raise _lib.ClhipError(_lib.lib().clhip_last_error().decode())
            s.handle = h
        if s.shadow is None or s.shadow.device != dev:
            s.shadow = torch.empty(_lib.lib().clhip_vit_shadow_bytes(s.handle), dtype=torch.uint8, device=dev)
            s.sig = None
        # parameter pointer table + staleness signature of the compute-dtype weight copies
        ptrs = tuple(p.data_ptr() for p in self._all_tensors())
        if s.cparams is None or s.keep != ptrs:
            for p in self._all_tensors():
                require_gpu(p)
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise _lib.ClhipError("ViT master parameters must be contiguous fp32")
            layers = (_lib.VitLayerParams * self.depth)()
            for i, b in enumerate(self.transformer.blocks):
                L = layers[i]
                L.qkv_w, L.qkv_b = b.attn.qkv.weight.data_ptr(), b.attn.qkv.bias.data_ptr()
                L.proj_w, L.proj_b = b.attn.proj.weight.data_ptr(), b.attn.proj.bias.data_ptr()
                L.ln1_w, L.ln1_b = b.ln_1.weight.data_ptr(), b.ln_1.bias.data_ptr()
                L.fc1_w, L.fc1_b = b.mlp.fc1.weight.data_ptr(), b.mlp.fc1.bias.data_ptr()
                L.fc2_w, L.fc2_b = b.mlp.fc2.weight.data_ptr(), b.mlp.fc2.bias.data_ptr()
                L.ln2_w, L.ln2_b = b.ln_2.weight.data_ptr(), b.ln_2.bias.data_ptr()
                if self.lora_rank:
                    L.lora_a_k, L.lora_b_k = b.attn.lora_A_k.weight.data_ptr(), b.attn.lora_B_k.weight.data_ptr()
                    L.lora_a_v, L.lora_b_v = b.attn.lora_A_v.weight.data_ptr(), b.attn.lora_B_v.weight.data_ptr()
            cp = _lib.VitParams(self.cls_token.data_ptr(), self.pos_embed.data_ptr(), self.patch_embed.proj.weight.data_ptr(),
                                self.patch_embed.proj.bias.data_ptr(), self.norm.weight.data_ptr(), self.norm.bias.data_ptr(), layers)
            s.cparams, s.keep, s._layers = cp, ptrs, layers
            s.sig = None
        lora_on = bool(self.lora_rank) and any(a.apply_lora for a in self.attention_modules())
        if lora_on and not all(a.apply_lora for a in self.attention_modules()):
            raise _lib.ClhipError("apply_lora must be set on all attention layers or none")
        frozen = self._frozen_tensors()
        if self.lora_rank:            # lora_A is fixed within a task: a change (init_param / SVD in before_task) forces the full preparation
            for a in self.attention_modules():
                frozen = frozen + [a.lora_A_k.weight, a.lora_A_v.weight]
        sig = (tuple((t.data_ptr(), t._version) for t in frozen), lora_on)
        if s.sig != sig:
            call("clhip_vit_prep_weights", s.handle, C.byref(s.cparams), s.shadow.data_ptr(), int(lora_on), 0, _st())
            s.sig = sig
        elif lora_on:
            # lora_B moves every optimizer step: refresh only the effective qkv copies (transformer.py:249-255)
            call("clhip_vit_prep_weights", s.handle, C.byref(s.cparams), s.shadow.data_ptr(), 1, 1, _st())
        return s

    def _workspace(self, s, B, n_prompt, save, dev):
        """`save`: bit 0 = keep what the backward needs, bit 1 = keep every layer's attention input for the Gram launch"""
        key = (B, n_prompt, int(save))
        need = _lib.lib().clhip_vit_workspace_bytes(s.handle, B, n_prompt, int(save))
        if need == 0:
            raise _lib.ClhipError(f"invalid ViT launch: batch {B}, {n_prompt} prompt tokens (max 256 tokens)")
        if s.ws is None or s.ws.device != dev or s.ws.numel() < need:
            s.ws = None
            s.ws = torch.empty(need, dtype=torch.uint8, device=dev)
        s.ws_key = key
        return s.ws

    def _run_forward(self, images, prompt_tokens, save, gram):
        require_gpu(images)
        images = images.float().contiguous()
        B = images.shape[0]
        if tuple(images.shape[1:]) != (3, self.img_size, self.img_size):
            raise _lib.ClhipError(f"expected images [B,3,{self.img_size},{self.img_size}], got {tuple(images.shape)}")
        dev = images.device
        s = self._ensure(dev)
        n_prompt = 0
        if prompt_tokens is not None:
            prompt_tokens = prompt_tokens.detach().float().contiguous()
            n_prompt = prompt_tokens.shape[0]
        ws = self._workspace(s, B, n_prompt, int(bool(save)) | (2 if gram is not None else 0), dev)
        feat = torch.empty(B, self.embed_dim, device=dev, dtype=torch.float32)
        call("clhip_vit_forward", s.handle, C.byref(s.cparams), s.shadow.data_ptr(), ws.data_ptr(), images.data_ptr(), B,
             prompt_tokens.data_ptr() if n_prompt else None, n_prompt, int(save), gram.data_ptr() if gram is not None else None,
             feat.data_ptr(), _st())
        self._fwd_token += 1
        self._last = (B, n_prompt)
        return feat

    def _run_backward(self, dfeat, want_prompt, want_lora):
        s = self._s
        B, n_prompt = self._last
        dfeat = dfeat.float().contiguous()
        dev = dfeat.device
        dprompt = torch.empty(n_prompt, self.embed_dim, device=dev) if want_prompt else None
        dl, arr = None, None
        if want_lora:
            dl = list(torch.zeros(2 * self.depth, self.embed_dim, self.lora_rank, device=dev).unbind(0))     # one fill, 2*depth views
            arr = (C.c_void_p * (2 * self.depth))(*[t.data_ptr() for t in dl])
        call("clhip_vit_backward", s.handle, C.byref(s.cparams), s.shadow.data_ptr(), s.ws.data_ptr(), dfeat.data_ptr(),
             dprompt.data_ptr() if want_prompt else None, arr, _st())
        return dprompt, dl

    # --------------------------------------------------------------------------------------- public forward
    def _gram_buffer(self, dev):
        """the resident [depth, D, D] fp32 sums of X^T X; layer i's LazyGram (if its attention module has one) owns row i"""
        s = self._s
        if s.gram is not None:                           # (holders re-created or detached since: a deep copy, a new attention module)
            g0 = self.attention_modules()[0].__dict__.get("_gram")
            if g0 is not None and (g0.dev is None or g0.dev.data_ptr() != s.gram[0].data_ptr()):
                s.gram = None
        if s.gram is None or s.gram.device != dev:
            for a in self.attention_modules():
                g = a.__dict__.get("_gram")
                if g is not None:
                    g.fold()
            s.gram = torch.zeros(self.depth, self.embed_dim, self.embed_dim, device=dev)
            for i, a in enumerate(self.attention_modules()):
                g = a.__dict__.get("_gram")
                if g is not None:
                    g.dev, g.n_dev = s.gram[i], 0
        return s.gram

    def features(self, images, prompt_tokens=None, get_input_matrix=False, gram_out=None):
        """[B, D] fp32: final-LN output at the cls token, or (with L2P prompt tokens [P, D]) the mean over the P prompt
        token outputs (transformer.py:2254-2261).  Differentiable w.r.t. prompt_tokens and the lora_B weights.
        get_input_matrix: every layer's X^T X (X = the attention input) is ADDED to a device-resident [depth, D, D] fp32 buffer by one
        MFMA launch at the end of the forward -- the modules' own (`cur_matrix`, read lazily) or the caller's `gram_out`."""
        gram = None
        if get_input_matrix:
            gram = gram_out if gram_out is not None else self._gram_buffer(images.device)
        lora_b = []
        if self.lora_rank and any(a.apply_lora for a in self.attention_modules()):
            for a in self.attention_modules():
                lora_b += [a.lora_B_k.weight, a.lora_B_v.weight]
        need = torch.is_grad_enabled() and ((prompt_tokens is not None and prompt_tokens.requires_grad) or any(b.requires_grad for b in lora_b))
        feat = _VitFn.apply(self, images, prompt_tokens, gram, need, *lora_b)
        if get_input_matrix and gram_out is None:
            # the running mean over tokens (transformer.py:241-244) is taken when `cur_matrix` is read; here only the token count moves
            cnt = images.shape[0] * (self.patch_embed.num_patches + 1 + (0 if prompt_tokens is None else prompt_tokens.shape[0]))
            for a in self.attention_modules():
                g = a.__dict__.get("_gram")
                if g is not None:
                    g.n_dev += cnt
        return feat

    def forward(self, x, prompt=None, prompt_flag="", cls_features=None, get_input_matrix=False, **kwargs):
        """the two branches of transformer.py:2222-2294 that the in-scope methods use"""
        if prompt_flag == "l2p":
            if prompt is None:
                return self.features(x)
            tokens, reduce_sim = prompt(None, cls_features=cls_features)
            return self.features(x, tokens), reduce_sim
        if prompt is not None:
            raise NotImplementedError("CODA / DualPrompt prompting is outside the hot-path scope (SURVEY.md section 8)")
        return self.features(x, None, get_input_matrix), None

    def debug_read(self, layer, which):
        s = self._s
        B, n_prompt = self._last
        M = B * (n_prompt + 1 + self.patch_embed.num_patches)
        width = {0: self.embed_dim, 1: 3 * self.embed_dim, 2: self.embed_dim, 3: self.embed_dim, 4: self.mlp_dim}[which]
        out = torch.empty(M, width, device=s.ws.device)
        call("clhip_vit_read_act", s.handle, s.ws.data_ptr(), layer, which, out.data_ptr(), _st())
        return out

    def __del__(self):
        s = getattr(self, "_s", None)
        if s is not None and s.handle is not None:
            try:
                _lib.lib().clhip_vit_destroy(s.handle)
            except Exception:
                pass
            s.handle = None


class _L2PSelectFn(torch.autograd.Function):
    """prompt.L2P.forward (prompt.py:369-406) as one kernel: -> (prompt tokens [top_k*length, D], reduce_sim)"""

    @staticmethod
    def forward(ctx, prompt, key, cls_features, top_k):
        require_gpu(prompt)
        _, pool, length, D = prompt.shape
        q = cls_features.detach().float().contiguous()
        B = q.shape[0]
        dev = prompt.device
        ids = torch.empty(top_k, dtype=torch.int32, device=dev)
        tokens = torch.empty(top_k * length, D, device=dev)
        rs = torch.empty(1, device=dev)
        dkey = torch.empty(pool, D, device=dev)
        scratch = torch.empty(B + pool + D + B * pool, device=dev)
        call("clhip_l2p_select", q.data_ptr(), key.detach().contiguous().data_ptr(), prompt.detach().contiguous().data_ptr(), B, D, pool, top_k, length,
             ids.data_ptr(), tokens.data_ptr(), rs.data_ptr(), dkey.data_ptr(), scratch.data_ptr(), _st())
        ctx.save_for_backward(ids, dkey)
        ctx.dims = (pool, top_k, length, D)
        ctx.mark_non_differentiable(ids)
        return tokens, rs.view(()), ids

    @staticmethod
    def backward(ctx, dtokens, drs, _):
        ids, dkey = ctx.saved_tensors
        pool, top_k, length, D = ctx.dims
        dpool = None
        if dtokens is not None:
            dpool = torch.empty(1, pool, length, D, device=ids.device)
            call("clhip_l2p_scatter", dtokens.float().contiguous().data_ptr(), ids.data_ptr(), dpool.data_ptr(), pool, top_k, length, D, _st())
        gk = None
        if drs is not None:
            gk = torch.empty_like(dkey)
            call("clhip_scale_dev", dkey.data_ptr(), gk.data_ptr(), dkey.numel(), 1.0, drs.reshape(1).float().contiguous().data_ptr(), _st())
        return dpool, gk, None, None


class L2PPrompt(nn.Module):
    """prompt.L2P (prompt.py:345-406): prompt pool [num_layers=1, pool, length, D] and keys [pool, D]"""

    def __init__(self, length, prompt_init=nn.init.uniform_, prompt_key=False, pool_size=None, top_k=None, num_layers=1, embed_dim=768):
        super().__init__()
        assert num_layers == 1
        self.length, self.pool_size, self.top_k, self.num_layers, self.embed_dim = length, pool_size, top_k, num_layers, embed_dim
        self.prompt = nn.Parameter(torch.empty(num_layers, pool_size, length, embed_dim))
        self.prompt_key = nn.Parameter(torch.empty(pool_size, embed_dim))
        prompt_init(self.prompt)
        prompt_init(self.prompt_key)
        self.last_ids = None

    def forward(self, x_embed, cls_features=None):
        tokens, reduce_sim, ids = _L2PSelectFn.apply(self.prompt, self.prompt_key, cls_features, self.top_k)
        self.last_ids = ids
        return tokens, reduce_sim


class ViTZoo(nn.Module):
    """core/model/backbone/vit.py:43-139.  Extra kwargs (img_size, patch_size, embed_dim, depth, num_heads, dtype) size
    the model for tests; the defaults are the reference's hard-coded ViT-B/16."""

    def __init__(self, pretrained=False, model_name="vit_base_patch16_224", attn_layer="MultiHeadAttention", checkpoint=None, img_size=224,
                 patch_size=16, embed_dim=768, depth=12, num_heads=12, dtype="bf16", **kwargs):
        super().__init__()
        kwargs.pop("num_classes", None)
        kwargs.pop("device", None)
        self.task_id = None
        self.feat_dim = embed_dim
        self.feat = VisionTransformer(img_size=img_size, patch_size=patch_size, embed_dim=embed_dim, depth=depth, num_heads=num_heads,
                                      attn_layer=attn_layer, dtype=dtype, **kwargs)
        if pretrained:
            self.load_pretrained(model_name, checkpoint)
        self.prompt = None
        self.prompt_flag = ""

    def load_pretrained(self, model_name, checkpoint=None):
        """timm state dict -> reference key names (vit.py:69-84).  There is no network here: the checkpoint must be a
        local file (`checkpoint` kwarg, $CLHIP_VIT_CHECKPOINT, or torch hub's cache directory)."""
        cands = [checkpoint, os.environ.get("CLHIP_VIT_CHECKPOINT"),
                 os.path.expanduser(f"~/.cache/torch/hub/checkpoints/{model_name}.pt"),
                 os.path.expanduser(f"~/.cache/torch/hub/checkpoints/{model_name}.pth")]
        path = next((c for c in cands if c and os.path.exists(c)), None)
        if path is None:
            raise FileNotFoundError(f"pretrained ViT weights for {model_name} not found (looked in {[c for c in cands if c]}); "
                                    "pass backbone.kwargs.checkpoint or set pretrained: false")
        sd = torch.load(path, map_location="cpu")
        sd = sd.get("state_dict", sd.get("model", sd)) if isinstance(sd, dict) else sd
        out = {}
        for k, v in sd.items():
            for old, new in ((".norm1.", ".ln_1."), (".norm2.", ".ln_2."), ("blocks.", "transformer.blocks.")):
                if old in k:
                    k = k.replace(old, new)
            out[k] = v
        own = self.feat.state_dict()
        self.feat.load_state_dict({k: v for k, v in out.items() if k in own and tuple(v.shape) == tuple(own[k].shape)}, strict=False)

    def create_prompt(self, prompt_flag, **kwargs):
        self.prompt_flag = prompt_flag
        if prompt_flag != "l2p":
            raise NotImplementedError("only the L2P prompt pool is on the hot path (SURVEY.md section 8)")
        self.prompt = L2PPrompt(**kwargs)

    def forward(self, image, text=None, pen=False, train=False, **kwargs):
        if self.prompt_flag == "l2p":
            with torch.no_grad():
                cls_features = self.feat(image, prompt_flag="l2p")
            return self.feat(image, prompt=self.prompt, cls_features=cls_features, prompt_flag="l2p")
        if self.prompt is not None:
            raise NotImplementedError
        out, _ = self.feat(image, **kwargs)
        return out.view(out.size(0), -1)


def vit_pt_imnet(pretrained=False, **kwargs):
    return ViTZoo(pretrained, **kwargs)
