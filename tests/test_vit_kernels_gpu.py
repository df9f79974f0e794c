"""ViT-path kernels through the C ABI vs plain torch fp32/fp64 math on the same inputs: GEMM + epilogues, attention
fwd/bwd (MFMA and generic paths), LayerNorm fwd/bwd, LN+pool, patchify / assemble, weight prep + LoRA, LoRA gradient,
Gram, L2P selection.  fp32 mode: tight; bf16 mode: inputs rounded to bf16 first, tolerance = bf16 output rounding."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from libcontinual_amd import _lib           # noqa: E402
from libcontinual_amd._lib import call      # noqa: E402

DEV = "cuda"
TD = {"bf16": torch.bfloat16, "f32": torch.float32}
CODE = {"bf16": _lib.BF16, "f32": _lib.F32}
TOL = {"bf16": 2e-2, "f32": 2e-5}


def st():
    return torch.cuda.current_stream().cuda_stream


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + int(np.prod(shape)) % 1000)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def relerr(got, want):
    got, want = got.double(), want.double()
    return float((got - want).abs().max() / (want.abs().max() + 1e-30))


def p(t):
    return t.data_ptr() if t is not None else None


@pytest.mark.parametrize("dt", ["bf16", "f32"])
@pytest.mark.parametrize("M,N,K", [(197 * 2, 768, 768), (300, 192, 64), (128, 128, 128), (1000, 2304, 768), (77, 64, 3072)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3, 4])
def test_gemm_nt(dt, M, N, K, epi):
    """the register-staged tile kernels (every tile shape pick_tile chooses); gemm8.hip has its own test below"""
    td = TD[dt]
    A = rnd(M, K, seed=1).to(td)
    B = rnd(N, K, scale=1 / math.sqrt(K), seed=2).to(td)
    bias = rnd(N, seed=3)
    R = rnd(M, N, seed=4).to(td)
    Hin = rnd(M, N, seed=5).to(td)
    Cc = torch.empty(M, N, device=DEV, dtype=td)
    Hout = torch.zeros(M, N, device=DEV, dtype=td)
    ref = A.double() @ B.double().T
    if epi in (1, 2, 3):
        ref = ref + bias.double()
    if epi == 2:
        ref = ref + R.double()
    pre = ref.clone()
    if epi == 3:        # C = gelu(pre), H = gelu'(pre)
        ref = F.gelu(ref)
        pre = 0.5 * (1 + torch.erf(pre / math.sqrt(2))) + pre * torch.exp(-0.5 * pre * pre) / math.sqrt(2 * math.pi)
    if epi == 4:        # C = acc * H
        ref = ref * Hin.double()
    Hp = Hout if epi == 3 else (Hin if epi == 4 else None)
    call("clhip_gemm_nt", p(A), p(B), p(Cc), p(bias) if epi in (1, 2, 3) else None, p(R) if epi == 2 else None, p(Hp), M, N, K, K, K, N, N, N, epi, CODE[dt], st())
    torch.cuda.synchronize()
    assert relerr(Cc, ref) < TOL[dt]
    if epi == 3:
        assert relerr(Hout, pre) < TOL[dt]


@pytest.mark.parametrize("M,N,K", [(300, 256, 256), (1000, 512, 384), (513, 768, 768), (256 * 9 + 7, 2304, 768), (77, 256, 3072), (256 * 33, 768, 256), (256 * 70 + 131, 1024, 512)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3, 4])
def test_gemm8_every_epilogue(M, N, K, epi):
    """the 256 x 256 eight-phase kernel (gemm8.hip, 64-deep K tiles in quadrant half tiles, DMA seven phases ahead) forced onto shapes it would not
    pick: ragged last row tile, one to nine column tiles, 4 to 48 K tiles, fewer tiles than workgroups, several tiles per persistent workgroup
    (the DMA queue runs across tile boundaries) and more tiles than 256 workgroups"""
    _lib.lib().clhip_gemm8_config(2)
    try:
        A = rnd(M, K, seed=1).to(torch.bfloat16)
        B = rnd(N, K, scale=1 / math.sqrt(K), seed=2).to(torch.bfloat16)
        bias = rnd(N, seed=3)
        R = rnd(M, N, seed=4).to(torch.bfloat16)
        Hin = rnd(M, N, seed=5).to(torch.bfloat16)
        Cc = torch.full((M + 1, N), 7.0, device=DEV, dtype=torch.bfloat16)          # one guard row behind the output
        Hout = torch.full((M + 1, N), 7.0, device=DEV, dtype=torch.bfloat16)
        ref = A.double() @ B.double().T
        if epi in (1, 2, 3):
            ref = ref + bias.double()
        if epi == 2:
            ref = ref + R.double()
        pre = ref.clone()
        if epi == 3:
            ref = F.gelu(ref)
            pre = 0.5 * (1 + torch.erf(pre / math.sqrt(2))) + pre * torch.exp(-0.5 * pre * pre) / math.sqrt(2 * math.pi)
        if epi == 4:
            ref = ref * Hin.double()
        Hp = Hout if epi == 3 else (Hin if epi == 4 else None)
        call("clhip_gemm_nt", p(A), p(B), p(Cc), p(bias) if epi in (1, 2, 3) else None, p(R) if epi == 2 else None, p(Hp), M, N, K, K, K, N, N, N, epi, CODE["bf16"], st())
        torch.cuda.synchronize()
        assert relerr(Cc[:M], ref) < TOL["bf16"]
        assert float((Cc[M].float() - 7.0).abs().max()) == 0.0
        if epi == 3:
            assert relerr(Hout[:M], pre) < TOL["bf16"]
            assert float((Hout[M].float() - 7.0).abs().max()) == 0.0
    finally:
        _lib.lib().clhip_gemm8_config(-1)


@pytest.mark.parametrize("M,N,K", [(256 * 86 + 100, 768, 256), (256 * 40, 2304, 256), (256 * 64, 1024, 384), (300, 256, 256)])
@pytest.mark.parametrize("epi", [0, 2, 3, 4])
def test_gemm8_whole_rounds_and_a_register_staged_tail(M, N, K, epi):
    """mode 1 (what CLHIP_GEMM8=1 runs): gemm8.hip computes the row panels that fill whole rounds of its 256 workgroups, the register-staged
    kernel the remaining rows (261 tiles -> 85 panels + 356 rows; 360 tiles -> 28 panels + 12 panels; 256 tiles -> all; 2 tiles -> none): one
    result, every row written once, the guard row behind the output untouched"""
    _lib.lib().clhip_gemm8_config(1)
    try:
        A = rnd(M, K, seed=11).to(torch.bfloat16)
        B = rnd(N, K, scale=1 / math.sqrt(K), seed=12).to(torch.bfloat16)
        bias = rnd(N, seed=13)
        R = rnd(M, N, seed=14).to(torch.bfloat16)
        Hin = rnd(M, N, seed=15).to(torch.bfloat16)
        Cc = torch.full((M + 1, N), 7.0, device=DEV, dtype=torch.bfloat16)
        Hout = torch.full((M + 1, N), 7.0, device=DEV, dtype=torch.bfloat16)
        ref = A.double() @ B.double().T
        if epi in (1, 2, 3):
            ref = ref + bias.double()
        if epi == 2:
            ref = ref + R.double()
        pre = ref.clone()
        if epi == 3:
            ref = F.gelu(ref)
            pre = 0.5 * (1 + torch.erf(pre / math.sqrt(2))) + pre * torch.exp(-0.5 * pre * pre) / math.sqrt(2 * math.pi)
        if epi == 4:
            ref = ref * Hin.double()
        Hp = Hout if epi == 3 else (Hin if epi == 4 else None)
        call("clhip_gemm_nt", p(A), p(B), p(Cc), p(bias) if epi in (1, 2, 3) else None, p(R) if epi == 2 else None, p(Hp), M, N, K, K, K, N, N, N, epi, CODE["bf16"], st())
        torch.cuda.synchronize()
        assert relerr(Cc[:M], ref) < TOL["bf16"]
        # row blocks: both kernels' parts separately (a missing part would hide in the norm of the whole)
        for lo in range(0, M, 4096):
            assert relerr(Cc[lo:min(M, lo + 4096)], ref[lo:min(M, lo + 4096)]) < TOL["bf16"]
        assert float((Cc[M].float() - 7.0).abs().max()) == 0.0
        if epi == 3:
            assert relerr(Hout[:M], pre) < TOL["bf16"]
            assert float((Hout[M].float() - 7.0).abs().max()) == 0.0
    finally:
        _lib.lib().clhip_gemm8_config(-1)


@pytest.mark.parametrize("M,N,K", [(3552, 768, 3072), (3552, 768, 2304), (777, 768, 1536), (130, 256, 4096)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3, 4])
def test_gemm_nt_split_k(M, N, K, epi):
    """few output tiles and a long K (the L2P batch-16 shapes first): 128 x 128 tiles over 2-4 K slices + one reduce / epilogue pass -- ragged last row tile,
    every epilogue through the second pass, a guard row behind the outputs, and the same bits from a second call (fixed slice order)"""
    A = rnd(M, K, seed=1).to(torch.bfloat16)
    B = rnd(N, K, scale=1 / math.sqrt(K), seed=2).to(torch.bfloat16)
    bias = rnd(N, seed=3)
    R = rnd(M, N, seed=4).to(torch.bfloat16)
    Hin = rnd(M, N, seed=5).to(torch.bfloat16)
    ref = A.double() @ B.double().T
    if epi in (1, 2, 3):
        ref = ref + bias.double()
    if epi == 2:
        ref = ref + R.double()
    pre = ref.clone()
    if epi == 3:
        ref = F.gelu(ref)
        pre = 0.5 * (1 + torch.erf(pre / math.sqrt(2))) + pre * torch.exp(-0.5 * pre * pre) / math.sqrt(2 * math.pi)
    if epi == 4:
        ref = ref * Hin.double()
    outs = []
    for _ in range(2):
        Cc = torch.full((M + 1, N), 7.0, device=DEV, dtype=torch.bfloat16)
        Hout = torch.full((M + 1, N), 7.0, device=DEV, dtype=torch.bfloat16)
        Hp = Hout if epi == 3 else (Hin if epi == 4 else None)
        call("clhip_gemm_nt", p(A), p(B), p(Cc), p(bias) if epi in (1, 2, 3) else None, p(R) if epi == 2 else None, p(Hp), M, N, K, K, K, N, N, N, epi, CODE["bf16"], st())
        torch.cuda.synchronize()
        assert relerr(Cc[:M], ref) < TOL["bf16"]
        assert float((Cc[M].float() - 7.0).abs().max()) == 0.0
        if epi == 3:
            assert relerr(Hout[:M], pre) < TOL["bf16"]
            assert float((Hout[M].float() - 7.0).abs().max()) == 0.0
        outs.append((Cc.clone(), Hout.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("M,N,K", [(25216, 2304, 64), (25216, 768, 2304), (8192, 4096, 64), (25216, 3072, 768), (25216, 2304, 768)])
@pytest.mark.parametrize("epi", [0, 2, 3, 4])
def test_gemm_nt_full_size_tilings(M, N, K, epi):
    """the BASELINE-size shapes take paths small cases never reach: whole rounds of 256 x 256 tiles + a small-tile tail launch
    with offset A / C / R / H pointers (short K, wide N and narrow N, long K), and the grouped tile rasterisation (N >= 4096)"""
    A = rnd(M, K, seed=1).to(torch.bfloat16)
    B = rnd(N, K, scale=1 / math.sqrt(K), seed=2).to(torch.bfloat16)
    bias = rnd(N, seed=3)
    R = rnd(M, N, seed=4).to(torch.bfloat16)
    Hin = rnd(M, N, seed=5).to(torch.bfloat16)
    Cc = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    Hout = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    Hp = Hout if epi == 3 else (Hin if epi == 4 else None)
    call("clhip_gemm_nt", p(A), p(B), p(Cc), p(bias) if epi in (2, 3) else None, p(R) if epi == 2 else None, p(Hp), M, N, K, K, K, N, N, N, epi, CODE["bf16"], st())
    ref = A.float() @ B.float().T                      # fp32 product of the bf16 operands on the device
    if epi in (2, 3):
        ref = ref + bias
    if epi == 2:
        ref = ref + R.float()
    if epi == 3:
        pre = ref
        ref = F.gelu(pre)
        dref = 0.5 * (1 + torch.erf(pre / math.sqrt(2))) + pre * torch.exp(-0.5 * pre * pre) / math.sqrt(2 * math.pi)
        assert relerr(Hout, dref) < TOL["bf16"]
    if epi == 4:
        ref = ref * Hin.float()
    torch.cuda.synchronize()
    assert relerr(Cc, ref) < TOL["bf16"]
    # every row panel was written (head launch, tail launch, last ragged tile)
    assert torch.isfinite(Cc.float()).all() and float(Cc[-1].float().abs().max()) > 0 and float(Cc[21759:21761].float().abs().max() if M > 21761 else 1.0) > 0


def attn_ref(qkv, B, N, H, D):
    hd = D // H
    q, k, v = qkv.double().reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    a = ((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(-1)
    return (a @ v).transpose(1, 2).reshape(B * N, D)


@pytest.mark.parametrize("dt", ["bf16", "f32"])
@pytest.mark.parametrize("B,N,H,hd", [(3, 197, 12, 64), (2, 222, 2, 64), (4, 17, 2, 64), (2, 23, 3, 64), (1, 256, 1, 64), (2, 50, 2, 32)])
def test_attention_fwd_bwd(dt, B, N, H, hd):
    """head dim 64 in bf16 = the MFMA kernels; fp32 and head dim 32 = the generic kernels"""
    td = TD[dt]
    D = H * hd
    qkv = (rnd(B * N, 3 * D, seed=7) * 1.5).to(td)
    dout = rnd(B * N, D, seed=8).to(td)
    out = torch.empty(B * N, D, device=DEV, dtype=td)
    lse = torch.empty(B, H, N, device=DEV)
    dqkv = torch.zeros(B * N, 3 * D, device=DEV, dtype=td)
    dsum = torch.empty(B, H, N, device=DEV)
    qd = qkv.double().requires_grad_(True)
    ref = attn_ref(qd, B, N, H, D)
    (ref * dout.double()).sum().backward()
    call("clhip_attn_fwd", p(qkv), p(out), p(lse), B, N, H, D, CODE[dt], st())
    call("clhip_attn_bwd", p(qkv), p(out), p(lse), p(dout), p(dqkv), p(dsum), B, N, H, D, CODE[dt], st())
    torch.cuda.synchronize()
    assert relerr(out, ref.detach()) < TOL[dt]
    g = qd.grad
    for i, nm in enumerate("qkv"):
        assert relerr(dqkv[:, i * D:(i + 1) * D], g[:, i * D:(i + 1) * D]) < (4e-2 if dt == "bf16" else 5e-5), nm


@pytest.mark.parametrize("B,N,H", [(5, 197, 12), (2, 222, 3), (3, 17, 2), (2, 240, 1), (2, 33, 2)])
def test_attention_bwd_variants_are_bit_identical(B, N, H):
    """attn_bwd_mfma3_kernel (one-round-trip prologue, no key masks, read-ahead; 197 / 222 tokens) issues the MFMAs of every stored element on the same operands in the
    same order as the kernel of rounds 2-5 (ATTN_BWD=1; also the path of every other token count)"""
    D = H * 64
    qkv = (rnd(B * N, 3 * D, seed=17) * 1.5).to(torch.bfloat16)
    dout = rnd(B * N, D, seed=18).to(torch.bfloat16)
    out = torch.empty(B * N, D, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(B, H, N, device=DEV)
    dsum = torch.empty(B, H, N, device=DEV)
    call("clhip_attn_fwd", p(qkv), p(out), p(lse), B, N, H, D, CODE["bf16"], st())
    got = {}
    for v in (b"1", b"3"):
        dqkv = torch.full((B * N, 3 * D), 3.0, device=DEV).to(torch.bfloat16)
        assert _lib.lib().clhip_config(b"ATTN_BWD", v) == 0
        try:
            call("clhip_attn_bwd", p(qkv), p(out), p(lse), p(dout), p(dqkv), p(dsum), B, N, H, D, CODE["bf16"], st())
            torch.cuda.synchronize()
        finally:
            _lib.lib().clhip_config(b"ATTN_BWD", None)
        got[v] = dqkv
    assert torch.isfinite(got[b"3"].float()).all() and float(got[b"3"].float().abs().max()) > 0
    assert torch.equal(got[b"1"], got[b"3"])


@pytest.mark.parametrize("dt", ["bf16", "f32"])
@pytest.mark.parametrize("M,D", [(197 * 3, 768), (50, 128), (7, 64), (33, 2048)])
def test_layernorm(dt, M, D):
    td = TD[dt]
    x = (rnd(M, D, seed=1) * 2 + 0.3).to(td)
    gamma, beta = rnd(D, seed=2) * 0.2 + 1, rnd(D, seed=3) * 0.1
    dy = rnd(M, D, seed=4).to(td)
    g0 = rnd(M, D, seed=5).to(td)
    y = torch.empty_like(x)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    call("clhip_ln_fwd", p(x), p(gamma), p(beta), p(y), p(mean), p(rstd), M, D, 1e-5, CODE[dt], st())
    xd = x.double().requires_grad_(True)
    ref = F.layer_norm(xd, (D,), gamma.double(), beta.double(), 1e-5)
    (ref * dy.double()).sum().backward()
    g = g0.clone()
    call("clhip_ln_bwd", p(dy), p(x), p(gamma), p(mean), p(rstd), p(g), M, D, CODE[dt], st())
    torch.cuda.synchronize()
    assert relerr(y, ref.detach()) < TOL[dt]
    assert relerr(mean, x.double().mean(1)) < 1e-5
    assert relerr(g, g0.double() + xd.grad) < TOL[dt]


@pytest.mark.parametrize("dt", ["bf16", "f32"])
@pytest.mark.parametrize("B,N,D,P", [(4, 197, 768, 1), (3, 222, 768, 25), (5, 23, 128, 6)])
def test_ln_pool(dt, B, N, D, P):
    td = TD[dt]
    x = (rnd(B * N, D, seed=1) * 2).to(td)
    gamma, beta = rnd(D, seed=2) * 0.2 + 1, rnd(D, seed=3) * 0.1
    dfeat = rnd(B, D, seed=4)
    feat = torch.empty(B, D, device=DEV)
    g = torch.full((B * N, D), 7.0, device=DEV).to(td)
    call("clhip_ln_pool_fwd", p(x), p(gamma), p(beta), p(feat), B, N, D, P, 1e-6, CODE[dt], st())
    call("clhip_ln_pool_bwd", p(dfeat), p(x), p(gamma), p(g), B, N, D, P, 1e-6, CODE[dt], st())
    xd = x.double().requires_grad_(True)
    ref = F.layer_norm(xd.reshape(B, N, D), (D,), gamma.double(), beta.double(), 1e-6)[:, :P].mean(1)
    (ref * dfeat.double()).sum().backward()
    torch.cuda.synchronize()
    assert relerr(feat, ref.detach()) < 1e-5
    assert relerr(g, xd.grad) < TOL[dt]


@pytest.mark.parametrize("dt", ["bf16", "f32"])
def test_patchify_assemble_promptgrad(dt):
    td = TD[dt]
    B, S, pz, D, P = 3, 32, 8, 128, 6
    npch = (S // pz) ** 2
    img = rnd(B, 3, S, S, seed=1)
    patches = torch.empty(B * npch, 3 * pz * pz, device=DEV, dtype=td)
    call("clhip_patchify", p(img), p(patches), B, S, pz, CODE[dt], st())
    ref = F.unfold(img, pz, stride=pz).transpose(1, 2).reshape(B * npch, -1)     # (c, i, j) column order
    torch.cuda.synchronize()
    assert relerr(patches, ref) < (4e-3 if dt == "bf16" else 1e-7)
    pe = rnd(B * npch, D, seed=2).to(td)
    cls, pos, prompt = rnd(D, seed=3), rnd(npch + 1, D, seed=4), rnd(P, D, seed=5)
    for n_prompt in (0, P):
        N = n_prompt + 1 + npch
        x = torch.empty(B * N, D, device=DEV, dtype=td)
        call("clhip_vit_assemble", p(pe), p(cls), p(pos), p(prompt) if n_prompt else None, p(x), B, npch, n_prompt, D, CODE[dt], st())
        parts = [prompt[:n_prompt].expand(B, -1, -1), (cls + pos[0]).expand(B, 1, -1), pe.float().reshape(B, npch, D) + pos[1:]]
        torch.cuda.synchronize()
        assert relerr(x, torch.cat(parts, 1).reshape(B * N, D)) < (4e-3 if dt == "bf16" else 1e-7)
    g = rnd(B * N, D, seed=6).to(td)
    dprompt = torch.empty(P, D, device=DEV)
    call("clhip_vit_prompt_grad", p(g), p(dprompt), B, N, P, D, CODE[dt], st())
    torch.cuda.synchronize()
    assert relerr(dprompt, g.float().reshape(B, N, D)[:, :P].sum(0)) < 1e-5


@pytest.mark.parametrize("dt", ["bf16", "f32"])
def test_weight_prep_and_lora(dt):
    td = TD[dt]
    D, r = 128, 10
    w = rnd(3 * D, D, seed=1)
    Ak, Bk, Av, Bv = rnd(r, D, seed=2), rnd(D, r, seed=3), rnd(r, D, seed=4), rnd(D, r, seed=5)
    wt = torch.empty(3 * D, D, device=DEV, dtype=td)
    wtt = torch.empty(D, 3 * D, device=DEV, dtype=td)
    eff = w.clone()
    eff[D:2 * D] += Bk @ Ak
    eff[2 * D:] += Bv @ Av
    call("clhip_weight_prep2", p(w), p(wt), p(wtt), 3 * D, D, p(Ak), p(Bk), p(Av), p(Bv), r, CODE[dt], st())
    torch.cuda.synchronize()
    assert relerr(wt, eff) < (4e-3 if dt == "bf16" else 1e-6)
    assert relerr(wtt, eff.T) < (4e-3 if dt == "bf16" else 1e-6)
    call("clhip_weight_prep2", p(w), p(wt), p(wtt), 3 * D, D, None, None, None, None, 0, CODE[dt], st())
    torch.cuda.synchronize()
    assert relerr(wt, w) < (4e-3 if dt == "bf16" else 1e-7)
    wm = w.clone()
    call("clhip_lora_merge", p(wm), p(Ak), p(Bk), p(Av), p(Bv), D, r, st())
    torch.cuda.synchronize()
    assert relerr(wm, eff) < 1e-6
    # B gradient through the rank-r shortcut
    M = 2700
    x = rnd(M, D, seed=6).to(td)
    dqkv = rnd(M, 3 * D, seed=7).to(td)
    xd, dd = x.double(), dqkv.double()
    acat = torch.empty(32, D, device=DEV, dtype=td)
    call("clhip_lora_acat", p(Ak), p(Av), p(acat), D, r, CODE[dt], st())
    for fast in (False, True):          # generic slab kernels / MFMA path (bf16 only; P is rounded to bf16 there)
        dBk = torch.ones(D, r, device=DEV)
        dBv = torch.zeros(D, r, device=DEV)
        ws = torch.empty(_lib.lib().clhip_lora_grad_ws_bytes(M, D, r), dtype=torch.uint8, device=DEV)
        call("clhip_lora_grad", p(x), p(dqkv), p(Ak), p(Av), p(acat) if fast else None, p(dBk), p(dBv), p(ws), M, D, r, CODE[dt], st())
        torch.cuda.synchronize()
        tol = 1e-2 if (fast and dt == "bf16") else 1e-4
        assert relerr(dBk, 1 + dd[:, D:2 * D].T @ (xd @ Ak.double().T)) < tol
        assert relerr(dBv, dd[:, 2 * D:].T @ (xd @ Av.double().T)) < tol


@pytest.mark.parametrize("dt", ["bf16", "f32"])
def test_gram(dt):
    td = TD[dt]
    M, D = 2500, 128
    x = rnd(M, D, seed=1).to(td)
    G = torch.ones(D, D, device=DEV)
    call("clhip_gram_accum", p(x), p(G), M, D, CODE[dt], st())
    torch.cuda.synchronize()
    assert relerr(G, 1 + x.double().T @ x.double()) < 1e-4


@pytest.mark.parametrize("M,D,L", [(2500, 192, 3), (197 * 4, 768, 2), (33, 64, 1)])
def test_gram_batched_mfma(M, D, L):
    """clhip_gram_accum_batched: X_l^T X_l of all layers in one MFMA launch (transformer.py:241-244) -- ragged row
    count, a partial 128-tile (D = 192), a layer stride larger than M * D, accumulation onto existing sums, bitwise reproducible"""
    stride = M * D + 256
    buf = torch.zeros(L * stride, device=DEV, dtype=torch.bfloat16)
    xs = []
    for l in range(L):
        x = rnd(M, D, seed=10 + l).to(torch.bfloat16)
        buf[l * stride:l * stride + M * D] = x.reshape(-1)
        xs.append(x.double())
    outs = []
    for _ in range(2):
        G = torch.full((L, D, D), 0.5, device=DEV)
        call("clhip_gram_accum_batched", p(buf), stride, L, p(G), M, D, CODE["bf16"], st())
        call("clhip_gram_accum_batched", p(buf), stride, L, p(G), M, D, CODE["bf16"], st())      # resident sums: a second batch adds on top
        torch.cuda.synchronize()
        outs.append(G.clone())
    assert torch.equal(outs[0], outs[1])
    for l in range(L):
        want = 0.5 + 2 * (xs[l].T @ xs[l])
        assert relerr(outs[0][l], want) < 1e-5
        assert torch.equal(outs[0][l], outs[0][l].T) or relerr(outs[0][l], outs[0][l].T) < 1e-6


def test_l2p_select():
    B, D, pool, top_k, length = 16, 128, 10, 5, 5
    q = rnd(B, D, seed=1)
    key, prompt = torch.rand(pool, D, device=DEV), torch.rand(pool, length, D, device=DEV)
    ids = torch.empty(top_k, dtype=torch.int32, device=DEV)
    tokens = torch.empty(top_k * length, D, device=DEV)
    rs = torch.empty(1, device=DEV)
    dkey = torch.empty(pool, D, device=DEV)
    scratch = torch.empty(B + pool + D + B * pool, device=DEV)
    call("clhip_l2p_select", p(q), p(key), p(prompt), B, D, pool, top_k, length, p(ids), p(tokens), p(rs), p(dkey), p(scratch), st())
    torch.cuda.synchronize()
    kd = key.double().requires_grad_(True)
    kn, qn = F.normalize(kd, dim=-1), F.normalize(q.double(), dim=-1)
    _, idx = torch.topk(qn @ kn.T, top_k, dim=1)
    counts = torch.bincount(idx.reshape(-1), minlength=pool).tolist()
    order = sorted(range(pool), key=lambda j: (-counts[j], j))[:top_k]
    assert ids.tolist() == order
    sel = torch.tensor(order, device=DEV)
    ref = (kn[sel].unsqueeze(0) * qn.unsqueeze(1)).sum() / B
    ref.backward()
    assert relerr(rs, ref.detach().reshape(1)) < 1e-5
    assert relerr(dkey, kd.grad) < 1e-4
    assert torch.equal(tokens, prompt[sel].reshape(-1, D))
    dtok = rnd(top_k * length, D, seed=3)
    dpool = torch.empty(pool, length, D, device=DEV)
    call("clhip_l2p_scatter", p(dtok), p(ids), p(dpool), pool, top_k, length, D, st())
    torch.cuda.synchronize()
    want = torch.zeros(pool, length, D, device=DEV)
    want[sel] = dtok.reshape(top_k, length, D)
    assert torch.equal(dpool, want)
