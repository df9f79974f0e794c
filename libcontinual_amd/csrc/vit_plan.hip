// vit_plan.hip -- whole-backbone executor for the ViT path: one C call runs the complete forward (patch embedding,
// token assembly with optional L2P prompt tokens, depth x [LN -> qkv -> attention -> proj(+res) -> LN -> fc1(GELU)
// -> fc2(+res)], final LN + pooling) or the complete activation-gradient backward (every weight of the backbone is
// frozen in L2P and InfLoRA_OPT; the only parameter gradients are the prompt tokens and the LoRA B matrices).
// Replaces VisionTransformer.forward / Transformer.forward / ResidualAttentionBlock.forward
// (core/model/backbone/transformer.py:2222-2294, 2006-2018, 1331-1336) and the autograd graph behind
// loss.backward() (l2p.py:103, trainer.py:604).  Nothing here allocates or synchronises: fixed workspace offsets.
#include <new>
#include <vector>

#include "common.h"

namespace {
size_t al(size_t v) { return (v + 255) / 256 * 256; }

enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_RES = 2, EPI_BIAS_GELU = 3, EPI_MUL = 4 };

struct LayerShadow { size_t qkv_f, qkv_b, proj_f, proj_b, fc1_f, fc1_b, fc2_f, fc2_b, acat; };

struct Layout {          // byte offsets into the workspace for one (B, n_prompt, save) configuration
    int B, P, N, M, save;
    size_t patches, pe_out, ln_out, act, g, dtmp, dbig, dqkv, dsum, lora_ws, total;
    std::vector<size_t> x_in, x_mid, qkv, attn_o, hpre, h1, st1, st2, lse;   // per layer (x_in has depth+1 entries)
};

template <typename T>
__global__ __launch_bounds__(256) void to_f32_kernel(const T* __restrict__ x, float* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = Elem<T>::ld(x + i);
}
}  // namespace

struct clhip_vit {
    clhip_vit_desc d;
    int dtype, esize, np, Kp;
    size_t shadow_bytes, pe_f;
    std::vector<LayerShadow> sh;
    Layout last;          // configuration of the most recent forward
    bool have_last;
    hipStream_t side;     // stream of the LoRA weight gradients (leaves of the backward chain), created on first use
    hipEvent_t ev_q, ev_l;
};

// `flags`: bit 0 = keep what the backward needs; bit 1 (CLHIP_VIT_KEEP_ATTN_IN) = keep every layer's attention input (LN1 output) until the
// end of the forward -- the Gram pass multiplies all of them in one launch
static void make_layout(const clhip_vit* v, int B, int P, int flags, Layout& L) {
    const clhip_vit_desc& d = v->d;
    const size_t e = v->esize;
    const int save = flags & 1;
    L.B = B; L.P = P; L.N = P + 1 + v->np; L.M = B * L.N; L.save = save;
    const size_t M = L.M, D = d.dim;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    L.patches = take((size_t)B * v->np * v->Kp * e);
    L.pe_out = take((size_t)B * v->np * D * e);
    L.ln_out = take(M * D * e);
    L.act = take(M * d.mlp * e);
    const int sets = save ? d.depth : 1;
    L.x_in.assign(d.depth + 1, 0); L.x_mid.assign(d.depth, 0); L.qkv.assign(d.depth, 0); L.attn_o.assign(d.depth, 0);
    L.hpre.assign(d.depth, 0); L.h1.assign(d.depth, 0); L.st1.assign(d.depth, 0); L.st2.assign(d.depth, 0); L.lse.assign(d.depth, 0);
    const bool keep_h1 = save && d.lora_rank > 0;
    for (int s = 0; s < sets; ++s) {
        L.x_mid[s] = take(M * D * e);
        L.qkv[s] = take(M * 3 * D * e);
        L.attn_o[s] = take(M * D * e);
        L.hpre[s] = save ? take(M * d.mlp * e) : 0;
        L.h1[s] = keep_h1 ? take(M * D * e) : L.ln_out;
        L.st1[s] = take(2 * M * sizeof(float));
        L.st2[s] = take(2 * M * sizeof(float));
        L.lse[s] = take((size_t)B * d.heads * L.N * sizeof(float));
    }
    if (save) {
        for (int l = 0; l <= d.depth; ++l) L.x_in[l] = take(M * D * e);
    } else {
        const size_t a = take(M * D * e), b = take(M * D * e);
        for (int l = 0; l <= d.depth; ++l) L.x_in[l] = (l & 1) ? b : a;
        for (int l = 1; l < d.depth; ++l) {
            L.x_mid[l] = L.x_mid[0]; L.qkv[l] = L.qkv[0]; L.attn_o[l] = L.attn_o[0]; L.hpre[l] = 0; L.h1[l] = L.ln_out;
            L.st1[l] = L.st1[0]; L.st2[l] = L.st2[0]; L.lse[l] = L.lse[0];
        }
    }
    if ((flags & 2) && !keep_h1) {                       // one contiguous region, layer stride M * D elements
        const size_t base = take((size_t)d.depth * al(M * D * e));
        for (int l = 0; l < d.depth; ++l) L.h1[l] = base + (size_t)l * al(M * D * e);
    }
    if (save) {
        L.g = take(M * D * e);
        L.dtmp = take(M * D * e);
        L.dbig = take(M * d.mlp * e);
        L.dqkv = take(M * 3 * D * e);
        L.dsum = take((size_t)B * d.heads * L.N * sizeof(float));
        L.lora_ws = d.lora_rank > 0 ? take(clhip_lora_grad_ws_bytes(L.M, d.dim, d.lora_rank)) : 0;
    } else {
        L.g = L.dtmp = L.dbig = L.dqkv = L.dsum = L.lora_ws = 0;
    }
    L.total = off;
}

extern "C" clhip_vit* clhip_vit_create(const clhip_vit_desc* desc, int dtype) {
    if (!desc || (dtype != CLHIP_BF16 && dtype != CLHIP_F32) || desc->dim <= 0 || desc->depth <= 0 || desc->heads <= 0 || desc->dim % desc->heads ||
        desc->patch <= 0 || desc->img % desc->patch || desc->dim % 64 || desc->mlp % 64 || (3 * desc->patch * desc->patch) % 64 ||
        desc->dim / desc->heads > 64 || desc->lora_rank < 0 || desc->lora_rank > 16) {
        clhip_set_error("clhip_vit_create: invalid descriptor (dim, mlp and 3*patch^2 must be multiples of 64; head dim <= 64; rank <= 16)");
        return nullptr;
    }
    clhip_vit* v = new (std::nothrow) clhip_vit();
    if (!v) { clhip_set_error("out of host memory"); return nullptr; }
    v->d = *desc; v->dtype = dtype; v->esize = dtype == CLHIP_BF16 ? 2 : 4;
    const int g = desc->img / desc->patch;
    v->np = g * g; v->Kp = 3 * desc->patch * desc->patch;
    if (v->np + 1 > 256) { clhip_set_error("clhip_vit_create: more than 256 tokens"); delete v; return nullptr; }
    const size_t e = v->esize, D = desc->dim, H = desc->mlp;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    v->pe_f = take(D * v->Kp * e);
    v->sh.resize(desc->depth);
    for (auto& s : v->sh) {
        s.qkv_f = take(3 * D * D * e); s.qkv_b = take(3 * D * D * e);
        s.proj_f = take(D * D * e); s.proj_b = take(D * D * e);
        s.fc1_f = take(H * D * e); s.fc1_b = take(H * D * e);
        s.fc2_f = take(H * D * e); s.fc2_b = take(H * D * e);
        s.acat = desc->lora_rank > 0 ? take(32 * D * e) : 0;
    }
    v->shadow_bytes = off;
    v->have_last = false;
    return v;
}

extern "C" void clhip_vit_destroy(clhip_vit* v) {
    if (v && v->side) {
        (void)hipStreamSynchronize(v->side);
        (void)hipEventDestroy(v->ev_q);
        (void)hipEventDestroy(v->ev_l);               // (the stream is the process-wide one: clhip_shared_stream)
    }
    delete v;
}
extern "C" size_t clhip_vit_shadow_bytes(const clhip_vit* v) { return v ? v->shadow_bytes : 0; }
extern "C" size_t clhip_vit_workspace_bytes(const clhip_vit* v, int B, int n_prompt, int save) {
    if (!v || B <= 0 || n_prompt < 0 || n_prompt + 1 + v->np > 256) return 0;
    Layout L;
    make_layout(v, B, n_prompt, save, L);
    return L.total;
}

#define TRY(x) do { int rc_ = (x); if (rc_ != CLHIP_OK) return rc_; } while (0)

extern "C" int clhip_vit_prep_weights(clhip_vit* v, const clhip_vit_params* P, void* shadow, int apply_lora, int qkv_only, void* stream) {
    CLHIP_CHECK_ARG(v && P && P->layers && shadow);
    char* sh = static_cast<char*>(shadow);
    const int D = v->d.dim, H = v->d.mlp, r = apply_lora ? v->d.lora_rank : 0;
    if (qkv_only && r > 0) {
        // between optimizer steps only lora_B moves: refresh the k / v rows of all layers' qkv copies in one launch
        const int n = v->d.depth;
        std::vector<const float*> w(n), ak(n), bk(n), av(n), bv(n);
        std::vector<void*> wt(n), wtt(n);
        for (int l = 0; l < n; ++l) {
            const clhip_vit_layer_params& p = P->layers[l];
            w[l] = p.qkv_w; ak[l] = p.lora_a_k; bk[l] = p.lora_b_k; av[l] = p.lora_a_v; bv[l] = p.lora_b_v;
            wt[l] = sh + v->sh[l].qkv_f; wtt[l] = sh + v->sh[l].qkv_b;
        }
        return clhip_lora_qkv_refresh(n, w.data(), ak.data(), bk.data(), av.data(), bv.data(), wt.data(), wtt.data(), D, r, v->dtype, stream);
    }
    if (!qkv_only) TRY(clhip_weight_prep2(P->pe_w, sh + v->pe_f, nullptr, D, v->Kp, nullptr, nullptr, nullptr, nullptr, 0, v->dtype, stream));
    for (int l = 0; l < v->d.depth; ++l) {
        const clhip_vit_layer_params& p = P->layers[l];
        const LayerShadow& s = v->sh[l];
        if (r > 0) {
            CLHIP_CHECK_ARG(p.lora_a_k && p.lora_b_k && p.lora_a_v && p.lora_b_v);
            TRY(clhip_lora_acat(p.lora_a_k, p.lora_a_v, sh + s.acat, D, r, v->dtype, stream));
        }
        TRY(clhip_weight_prep2(p.qkv_w, sh + s.qkv_f, sh + s.qkv_b, 3 * D, D, p.lora_a_k, p.lora_b_k, p.lora_a_v, p.lora_b_v, r, v->dtype, stream));
        if (qkv_only) continue;
        TRY(clhip_weight_prep2(p.proj_w, sh + s.proj_f, sh + s.proj_b, D, D, nullptr, nullptr, nullptr, nullptr, 0, v->dtype, stream));
        TRY(clhip_weight_prep2(p.fc1_w, sh + s.fc1_f, sh + s.fc1_b, H, D, nullptr, nullptr, nullptr, nullptr, 0, v->dtype, stream));
        TRY(clhip_weight_prep2(p.fc2_w, sh + s.fc2_f, sh + s.fc2_b, D, H, nullptr, nullptr, nullptr, nullptr, 0, v->dtype, stream));
    }
    return CLHIP_OK;
}

extern "C" int clhip_vit_forward(clhip_vit* v, const clhip_vit_params* P, const void* shadow, void* workspace, const float* images, int B,
                                 const float* prompt_tokens, int n_prompt, int save, float* gram, float* feat, void* stream) {
    CLHIP_CHECK_ARG(v && P && P->layers && shadow && workspace && images && feat && B > 0 && n_prompt >= 0);
    CLHIP_CHECK_ARG(n_prompt == 0 || prompt_tokens != nullptr);
    CLHIP_CHECK_ARG(n_prompt + 1 + v->np <= 256);
    const clhip_vit_desc& d = v->d;
    Layout& L = v->last;
    make_layout(v, B, n_prompt, (save ? 1 : 0) | (gram ? 2 : 0), L);
    v->have_last = true;
    char* ws = static_cast<char*>(workspace);
    const char* sh = static_cast<const char*>(shadow);
    const int D = d.dim, Hm = d.mlp, M = L.M, N = L.N, dt = v->dtype;
    const float beps = d.block_ln_eps > 0.f ? d.block_ln_eps : 1e-5f;
    // patch embedding (timm PatchEmbed = Conv2d(3, D, p, stride p)) as patchify + GEMM, then token assembly
    TRY(clhip_patchify(images, ws + L.patches, B, d.img, d.patch, dt, stream));
    TRY(clhip_gemm_nt(ws + L.patches, sh + v->pe_f, ws + L.pe_out, P->pe_b, nullptr, nullptr, B * v->np, D, v->Kp, v->Kp, v->Kp, D, 0, 0, EPI_BIAS, dt, stream));
    TRY(clhip_vit_assemble(ws + L.pe_out, P->cls_token, P->pos_embed, prompt_tokens, ws + L.x_in[0], B, v->np, n_prompt, D, dt, stream));
    for (int l = 0; l < d.depth; ++l) {
        const clhip_vit_layer_params& p = P->layers[l];
        const LayerShadow& s = v->sh[l];
        float* st1 = reinterpret_cast<float*>(ws + L.st1[l]);
        float* st2 = reinterpret_cast<float*>(ws + L.st2[l]);
        char* h1 = ws + L.h1[l];
        TRY(clhip_ln_fwd(ws + L.x_in[l], p.ln1_w, p.ln1_b, h1, st1, st1 + M, M, D, beps, dt, stream));
        TRY(clhip_gemm_nt(h1, sh + s.qkv_f, ws + L.qkv[l], p.qkv_b, nullptr, nullptr, M, 3 * D, D, D, D, 3 * D, 0, 0, EPI_BIAS, dt, stream));
        TRY(clhip_attn_fwd(ws + L.qkv[l], ws + L.attn_o[l], reinterpret_cast<float*>(ws + L.lse[l]), B, N, d.heads, D, dt, stream));
        TRY(clhip_gemm_nt(ws + L.attn_o[l], sh + s.proj_f, ws + L.x_mid[l], p.proj_b, ws + L.x_in[l], nullptr, M, D, D, D, D, D, D, 0, EPI_BIAS_RES, dt, stream));
        TRY(clhip_ln_fwd(ws + L.x_mid[l], p.ln2_w, p.ln2_b, ws + L.ln_out, st2, st2 + M, M, D, beps, dt, stream));
        TRY(clhip_gemm_nt(ws + L.ln_out, sh + s.fc1_f, ws + L.act, p.fc1_b, nullptr, save ? ws + L.hpre[l] : nullptr, M, Hm, D, D, D, Hm, 0, Hm, EPI_BIAS_GELU, dt,
                          stream));
        TRY(clhip_gemm_nt(ws + L.act, sh + s.fc2_f, ws + L.x_in[l + 1], p.fc2_b, ws + L.x_mid[l], nullptr, M, D, Hm, Hm, Hm, D, D, 0, EPI_BIAS_RES, dt, stream));
    }
    if (gram) {
        // every layer's attention input is still there (make_layout bit 1, or the backward's own copies): X_l^T X_l of all layers, one launch
        bool strided = true;
        for (int l = 1; l < d.depth; ++l) strided = strided && (L.h1[l] - L.h1[l - 1]) == (L.h1[1] - L.h1[0]) && (L.h1[1] - L.h1[0]) % v->esize == 0;
        if (strided && d.depth > 1) TRY(clhip_gram_accum_batched(ws + L.h1[0], (L.h1[1] - L.h1[0]) / v->esize, d.depth, gram, M, D, dt, stream));
        else for (int l = 0; l < d.depth; ++l) TRY(clhip_gram_accum(ws + L.h1[l], gram + (size_t)l * D * D, M, D, dt, stream));
    }
    return clhip_ln_pool_fwd(ws + L.x_in[d.depth], P->norm_w, P->norm_b, feat, B, N, D, n_prompt > 0 ? n_prompt : 1, 1e-6f, dt, stream);
}

extern "C" int clhip_vit_backward(clhip_vit* v, const clhip_vit_params* P, const void* shadow, void* workspace, const float* dfeat, float* dprompt_tokens,
                                  float* const* d_lora_b, void* stream) {
    CLHIP_CHECK_ARG(v && P && P->layers && shadow && workspace && dfeat);
    CLHIP_CHECK_ARG(v->have_last && v->last.save);
    const clhip_vit_desc& d = v->d;
    const Layout& L = v->last;
    CLHIP_CHECK_ARG(dprompt_tokens == nullptr || L.P > 0);
    CLHIP_CHECK_ARG(d_lora_b == nullptr || d.lora_rank > 0);
    char* ws = static_cast<char*>(workspace);
    const char* sh = static_cast<const char*>(shadow);
    const int D = d.dim, Hm = d.mlp, M = L.M, N = L.N, B = L.B, dt = v->dtype;
    char* g = ws + L.g;
    // the lora_B gradients are leaves of the chain: they run on a side stream (ordered by events against the single dqkv buffer)
    static const bool two_streams = !(clhip_cfg("WGRAD_STREAM") && atoi(clhip_cfg("WGRAD_STREAM")) == 0);
    hipStream_t main_s = static_cast<hipStream_t>(stream);
    // (inside a stream capture everything stays on the captured stream, as plan.hip does: no fork / join nodes, no stream probe)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(main_s, &cap);
    const bool side_on = two_streams && d_lora_b != nullptr && cap == hipStreamCaptureStatusNone;
    hipStream_t shared_side = side_on ? clhip_shared_stream(0, main_s, false) : nullptr;
    if (side_on && shared_side == nullptr) { clhip_set_error("clhip_vit_backward: cannot create the side stream"); return CLHIP_EHIP; }
    const bool first_side = side_on && !v->side;
    if (side_on) v->side = shared_side;
    if (first_side) {
        (void)hipEventCreateWithFlags(&v->ev_q, hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&v->ev_l, hipEventDisableTiming);
    }
    bool lora_pending = false;
    TRY(clhip_ln_pool_bwd(dfeat, ws + L.x_in[d.depth], P->norm_w, g, B, N, D, L.P > 0 ? L.P : 1, 1e-6f, dt, stream));
    for (int l = d.depth - 1; l >= 0; --l) {
        const clhip_vit_layer_params& p = P->layers[l];
        const LayerShadow& s = v->sh[l];
        const float* st1 = reinterpret_cast<const float*>(ws + L.st1[l]);
        const float* st2 = reinterpret_cast<const float*>(ws + L.st2[l]);
        // MLP branch: x_out = x_mid + fc2(gelu(fc1(LN2(x_mid))))
        TRY(clhip_gemm_nt(g, sh + s.fc2_b, ws + L.dbig, nullptr, nullptr, ws + L.hpre[l], M, Hm, D, D, D, Hm, 0, Hm, EPI_MUL, dt, stream));
        TRY(clhip_gemm_nt(ws + L.dbig, sh + s.fc1_b, ws + L.dtmp, nullptr, nullptr, nullptr, M, D, Hm, Hm, Hm, D, 0, 0, EPI_NONE, dt, stream));
        TRY(clhip_ln_bwd(ws + L.dtmp, ws + L.x_mid[l], p.ln2_w, st2, st2 + M, g, M, D, dt, stream));
        // attention branch: x_mid = x_in + proj(attn(qkv(LN1(x_in))))
        TRY(clhip_gemm_nt(g, sh + s.proj_b, ws + L.dtmp, nullptr, nullptr, nullptr, M, D, D, D, D, D, 0, 0, EPI_NONE, dt, stream));
        if (lora_pending) { (void)hipStreamWaitEvent(main_s, v->ev_l, 0); lora_pending = false; }     // the previous layer's dB has read dqkv
        TRY(clhip_attn_bwd(ws + L.qkv[l], ws + L.attn_o[l], reinterpret_cast<const float*>(ws + L.lse[l]), ws + L.dtmp, ws + L.dqkv,
                           reinterpret_cast<float*>(ws + L.dsum), B, N, d.heads, D, dt, stream));
        if (d_lora_b) {
            CLHIP_CHECK_ARG(p.lora_a_k && p.lora_a_v && d_lora_b[2 * l] && d_lora_b[2 * l + 1]);
            void* ls = stream;
            if (side_on) {
                (void)hipEventRecord(v->ev_q, main_s);
                (void)hipStreamWaitEvent(v->side, v->ev_q, 0);
                ls = v->side;
            }
            TRY(clhip_lora_grad(ws + L.h1[l], ws + L.dqkv, p.lora_a_k, p.lora_a_v, sh + s.acat, d_lora_b[2 * l], d_lora_b[2 * l + 1], ws + L.lora_ws, M, D,
                                d.lora_rank, dt, ls));
            if (side_on) { (void)hipEventRecord(v->ev_l, v->side); lora_pending = true; }
        }
        if (l == 0 && dprompt_tokens == nullptr) break;           // nothing below the first block needs a gradient
        TRY(clhip_gemm_nt(ws + L.dqkv, sh + s.qkv_b, ws + L.dtmp, nullptr, nullptr, nullptr, M, D, 3 * D, 3 * D, 3 * D, D, 0, 0, EPI_NONE, dt, stream));
        TRY(clhip_ln_bwd(ws + L.dtmp, ws + L.x_in[l], p.ln1_w, st1, st1 + M, g, M, D, dt, stream));
    }
    if (dprompt_tokens) TRY(clhip_vit_prompt_grad(g, dprompt_tokens, B, N, L.P, D, dt, stream));
    if (lora_pending) (void)hipStreamWaitEvent(main_s, v->ev_l, 0);          // the caller's stream owns the gradients again
    return CLHIP_OK;
}

extern "C" int clhip_vit_read_act(clhip_vit* v, void* workspace, int layer, int which, float* out, void* stream) {
    CLHIP_CHECK_ARG(v && workspace && out && v->have_last && v->last.save);
    CLHIP_CHECK_ARG(layer >= 0 && layer <= v->d.depth && which >= 0 && which <= 4 && (layer < v->d.depth || which == 0));
    const Layout& L = v->last;
    const size_t D = v->d.dim, M = L.M;
    size_t off, n;
    switch (which) {
        case 0: off = L.x_in[layer]; n = M * D; break;
        case 1: off = L.qkv[layer]; n = M * 3 * D; break;
        case 2: off = L.attn_o[layer]; n = M * D; break;
        case 3: off = L.x_mid[layer]; n = M * D; break;
        default: off = L.hpre[layer]; n = M * v->d.mlp; break;
    }
    const char* src = static_cast<const char*>(workspace) + off;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int blocks = (int)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
    if (v->dtype == CLHIP_BF16) hipLaunchKernelGGL(to_f32_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)src, out, n);
    else hipLaunchKernelGGL(to_f32_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)src, out, n);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}
