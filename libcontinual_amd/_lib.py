"""ctypes binding of libclhip.so (C ABI declared in include/clhip.h).

There is NO fallback: if the shared library is missing, or a compute entry point is called without a
HIP device, this module raises.  torch is used only to own device memory and streams.
"""
import ctypes as C
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libclhip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "clhip.h")

BF16, F32 = 0, 1


class ClhipError(RuntimeError):
    pass


class UnitDesc(C.Structure):
    _fields_ = [("cin", C.c_int32), ("cout", C.c_int32), ("ksize", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("src", C.c_int32), ("res", C.c_int32), ("relu", C.c_int32),
                ("w_off", C.c_int64), ("gamma_off", C.c_int64), ("beta_off", C.c_int64),
                ("rm_off", C.c_int64), ("rv_off", C.c_int64)]


class VitDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("img", "patch", "dim", "depth", "heads", "mlp", "lora_rank")] + [("block_ln_eps", C.c_float)]


class VitLayerParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("qkv_w", "qkv_b", "proj_w", "proj_b", "ln1_w", "ln1_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b",
                                          "ln2_w", "ln2_b", "lora_a_k", "lora_b_k", "lora_a_v", "lora_b_v")]


class VitParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("cls_token", "pos_embed", "pe_w", "pe_b", "norm_w", "norm_b")] + [("layers", C.POINTER(VitLayerParams))]


def build(force=False):
    """Compile libclhip.so with hipcc for gfx950 (cross-compiles without a GPU)."""
    script = os.path.join(_HERE, "csrc", "build.sh")
    if force:
        import shutil
        shutil.rmtree(os.path.join(_HERE, "csrc", "obj"), ignore_errors=True)
    subprocess.run(["bash", script], check=True)


_lib = None
_p, _i, _l, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t

# name -> (restype, argtypes); every int-returning function is wrapped with an error check
_PROTOS = {
    "clhip_last_error": (C.c_char_p, []),
    "clhip_version": (_i, []),
    "clhip_nchw_to_nhwc": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "clhip_nhwc_to_nchw": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "clhip_conv_weight_prep": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "clhip_conv_fwd_tiles": (_i, [_i] * 8),
    "clhip_conv_fwd": (_i, [_p, _p, _p, _p] + [_i] * 9 + [_p]),
    "clhip_conv_fwd_acc": (_i, [_p, _p, _p, _p] + [_i] * 10 + [_p]),
    "clhip_bn_apply_train": (_i, [_p, _p, _i, _l, _i, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _i, _i, _p]),
    "clhip_bn_apply_train_mask": (_i, [_p, _p, _i, _l, _i, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _p, _i, _p]),
    "clhip_bn_bwd_acc": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _l, _i, _i, _p, _i, _i, _p]),
    "clhip_bn_bwd_acc_zmask": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _l, _i, _p, _i, _i, _p]),
    "clhip_conv_dgrad": (_i, [_p, _p, _p, _i] + [_i] * 9 + [_p]),
    "clhip_conv_dgrad_bn_reduce_supported": (_i, [_i] * 9),
    "clhip_conv_dgrad_bn_reduce": (_i, [_p, _p, _p, _i, _p, _p, _p, _p, _p, _i] + [_i] * 9 + [_p]),
    "clhip_conv_dgrad_bn_reduce_overlapped": (_i, [_i] * 9),
    "clhip_conv_dgrad_bn_reduce_ex": (_i, [_p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i] + [_i] * 9 + [_p]),
    "clhip_bn_bwd_apply_acc": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _l, _i, _i, _p, _i, _i, _p]),
    "clhip_conv_wgrad_ws_bytes": (_sz, [_i] * 10),
    "clhip_conv_dgrad_wgrad_supported": (_i, [_i] * 10),
    "clhip_conv_bn_input_supported": (_i, [_i] * 9),
    "clhip_conv_fwd_acc_bn_input": (_i, [_p, _p, _p, _p, _p, _i] + [_i] * 9 + [_p]),
    "clhip_conv_fwd_acc_bn_res_input": (_i, [_p, _p, _p, _p, _p, _p, _i] + [_i] * 9 + [_p]),
    "clhip_conv_dgrad_wgrad_bn_input": (_i, [_p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _i] + [_i] * 10 + [_p]),
    "clhip_conv_dgrad_wgrad_bn_grad": (_i, [_p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _i] + [_i] * 10 + [_p]),
    "clhip_conv_dgrad_pair_supported": (_i, [_i] * 6),
    "clhip_conv_dgrad_pair_packed_bytes": (_sz, [_i, _i]),
    "clhip_conv_dgrad_pair_pack": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "clhip_conv_dgrad_pair": (_i, [_p, _p, _p, _p, _i] + [_i] * 6 + [_p]),
    "clhip_conv_dgrad_pair_bn_reduce_supported": (_i, [_i] * 6),
    "clhip_conv_dgrad_pair_bn_reduce": (_i, [_p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _i] + [_i] * 6 + [_p]),
    "clhip_conv_wgrad_pair_supported": (_i, [_i] * 6),
    "clhip_conv_fwd_acc_pair_supported": (_i, [_i] * 6),
    "clhip_conv_fwd_acc_pair": (_i, [_p, _p, _p, _p, _p, _p, _i, _p, _i] + [_i] * 6 + [_p]),
    "clhip_conv_wgrad_pair": (_i, [_p] * 7 + [_i] * 6 + [_p]),
    "clhip_conv_dgrad_wgrad": (_i, [_p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _i] + [_i] * 10 + [_p]),
    "clhip_conv_wgrad": (_i, [_p, _p, _p, _p] + [_i] * 10 + [_p]),
    "clhip_bn_stats_finalize": (_i, [_p, _i, _l, _i, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _p]),
    "clhip_bn_eval_affine": (_i, [_p, _p, _p, _p, _f, _i, _p, _p, _p]),
    "clhip_bn_apply": (_i, [_p, _p, _p, _p, _p, _l, _i, _i, _i, _p]),
    "clhip_bn_apply_eval": (_i, [_p, _p, _p, _p, _p, _f, _p, _p, _l, _i, _i, _i, _p]),
    "clhip_bn_bwd_blocks": (_i, [_l, _i]),
    "clhip_bn_bwd_ws_floats": (_sz, [_l, _i]),
    "clhip_bn_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _l, _i, _i, _p, _i, _p]),
    "clhip_avgpool_fwd": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "clhip_avgpool_bwd": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "clhip_avgpool_bwd_bn_reduce_supported": (_i, [_i] * 4),
    "clhip_avgpool_bwd_bn_reduce": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "clhip_avgpool_win_fwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "clhip_avgpool_win_bwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "clhip_add_stats_blocks": (_i, [_l, _i]),
    "clhip_add_stats": (_i, [_p, _p, _p, _i, _l, _i, _i, _p]),
    "clhip_add_inplace": (_i, [_p, _p, _l, _i, _p]),
    "clhip_plan_create": (_p, [C.POINTER(UnitDesc), _i, _i, _i, _i, _i, _i]),
    "clhip_plan_create_ex": (_p, [C.POINTER(UnitDesc), _i, _i, _i, _i, _i, _i, _i]),
    "clhip_plan_destroy": (None, [_p]),
    "clhip_plan_workspace_bytes": (_sz, [_p]),
    "clhip_plan_shadow_bytes": (_sz, [_p]),
    "clhip_plan_feat_dim": (_i, [_p]),
    "clhip_plan_stage_status": (_i, [_p]),
    "clhip_plan_stage_info": (C.c_longlong, [_p, _i]),
    "clhip_plan_stage_trace": (_i, [_p, _p]),
    "clhip_plan_prep_weights": (_i, [_p, _p, _p, _p]),
    "clhip_plan_forward": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _p]),
    "clhip_plan_forward_ex": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _p, _p]),
    "clhip_plan_backward": (_i, [_p, _p, _p, _p, _p, _p, _p]),
    "clhip_plan_num_units": (_i, [_p]),
    "clhip_plan_backward_range": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "clhip_plan_read_act": (_i, [_p, _p, _i, _i, _p, _p]),
    "clhip_linear_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "clhip_linear_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "clhip_ce_slice": (_i, [_p, _p, _i, _i, _i, _i, _i, _f, _p, _i, _p, _i, _p, _p, _p]),
    "clhip_ce_window": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _f, _p, _i, _p, _i, _p, _p, _p]),
    "clhip_kd_loss": (_i, [_p, _i, _p, _i, _i, _i, _f, _f, _p, _i, _p, _i, _p]),
    "clhip_cosine_linear_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "clhip_cosine_linear_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "clhip_cos_embed_loss": (_i, [_p, _p, _i, _i, _f, _p, _i, _p, _i, _p]),
    "clhip_margin_rank_loss": (_i, [_p, _p, _i, _i, _i, _i, _f, _f, _p, _i, _p, _i, _p, _p]),
    "clhip_ewc_penalty": (_i, [_p, _p, _p, _l, _f, _p, _i, _p]),
    "clhip_ewc_grad": (_i, [_p, _p, _p, _p, _l, _f, _p, _p]),
    "clhip_ewc_penalty_multi": (_i, [_i, _p, _p, _p, _p, _f, _p, _i, _p]),
    "clhip_ewc_grad_multi": (_i, [_i, _p, _p, _p, _p, _p, _f, _p, _p]),
    "clhip_fisher_accum": (_i, [_p, _p, _l, _f, _p]),
    "clhip_fisher_merge": (_i, [_p, _p, _l, _f, _p]),
    "clhip_sgd_step": (_i, [_p, _p, _p, _l, _f, _f, _f, _f, _p, _p, _f, _p]),
    "clhip_sgd_step_multi": (_i, [_i, _p, _p, _p, _p, _f, _f, _f, _f, _p]),
    "clhip_sgd_step_multi_zero": (_i, [_i, _p, _p, _p, _p, _f, _f, _f, _f, C.c_uint, _p]),
    "clhip_adam_step": (_i, [_p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _f, _i, _p]),
    "clhip_sq_norm": (_i, [_p, _l, _p, _i, _p]),
    "clhip_scale": (_i, [_p, _l, _f, _p]),
    "clhip_scale_dev": (_i, [_p, _p, _l, _f, _p, _p]),
    "clhip_sigma_scale_fwd": (_i, [_p, _p, _p, _l, _p]),
    "clhip_sigma_scale_bwd": (_i, [_p, _p, _p, _p, _p, _i, _l, _p]),
    "clhip_l2_normalize_rows": (_i, [_p, _p, _i, _i, _p]),
    "clhip_ncm_classify": (_i, [_p, _p, _i, _i, _i, _p, _p]),
    "clhip_herding_select": (_i, [_p, _i, _i, _i, _p, _p, _p]),
    "clhip_herding_select_batched": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p]),
    "clhip_augment_crop_flip": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), _p]),
    "clhip_augment_rrc_flip": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), _p]),
    "clhip_augment_rrc_aa_ws_bytes": (_sz, [_i, _i, _i]),
    "clhip_augment_rrc_aa": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), _p]),
    "clhip_gemm_nt": (_i, [_p, _p, _p, _p, _p, _p] + [_i] * 10 + [_p]),
    "clhip_config": (_i, [C.c_char_p, C.c_char_p]),
    "clhip_config_get": (C.c_char_p, [C.c_char_p]),
    "clhip_conv_bn_input_wt_supported": (_i, [_i] * 9),
    "clhip_conv_fwd_acc_bn_input_wt": (_i, [_p, _p, _p, _p, _p, _p, _i] + [_i] * 9 + [_p]),
    "clhip_gram_accum_batched": (_i, [_p, C.c_size_t, _i, _p, _i, _i, _i, _p]),
    "clhip_gemm8_config": (None, [_i]),
    "clhip_wgrad4_config": (None, [_i]),
    "clhip_attn_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "clhip_attn_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "clhip_ln_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _f, _i, _p]),
    "clhip_ln_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "clhip_ln_pool_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p]),
    "clhip_ln_pool_bwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p]),
    "clhip_patchify": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "clhip_vit_assemble": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "clhip_vit_prompt_grad": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "clhip_weight_prep2": (_i, [_p, _p, _p, _i, _i, _p, _p, _p, _p, _i, _i, _p]),
    "clhip_lora_qkv_refresh": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "clhip_lora_merge": (_i, [_p, _p, _p, _p, _p, _i, _i, _p]),
    "clhip_lora_grad_ws_bytes": (_sz, [_i, _i, _i]),
    "clhip_lora_acat": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "clhip_lora_grad": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "clhip_gram_accum": (_i, [_p, _p, _i, _i, _i, _p]),
    "clhip_l2p_select": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p]),
    "clhip_l2p_scatter": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "clhip_vit_create": (_p, [C.POINTER(VitDesc), _i]),
    "clhip_vit_destroy": (None, [_p]),
    "clhip_vit_shadow_bytes": (_sz, [_p]),
    "clhip_vit_workspace_bytes": (_sz, [_p, _i, _i, _i]),
    "clhip_vit_prep_weights": (_i, [_p, C.POINTER(VitParams), _p, _i, _i, _p]),
    "clhip_vit_forward": (_i, [_p, C.POINTER(VitParams), _p, _p, _p, _i, _p, _i, _i, _p, _p, _p]),
    "clhip_vit_backward": (_i, [_p, C.POINTER(VitParams), _p, _p, _p, _p, C.POINTER(C.c_void_p), _p]),
    "clhip_vit_read_act": (_i, [_p, _p, _i, _i, _p, _p]),
}

_CHECKED = {}


def header_symbols():
    """Every function name declared in include/clhip.h."""
    with open(HEADER_PATH) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(clhip_[a-z0-9_]+)\s*\(", src)))


def lib():
    """The loaded library; raises ClhipError (never falls back) if it cannot be loaded."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ClhipError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        try:
            L = C.CDLL(LIB_PATH)
        except OSError as e:
            raise ClhipError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in _PROTOS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def call(name, *args):
    """Call an int-returning entry point and raise with clhip_last_error() on failure."""
    fn = _CHECKED.get(name)
    if fn is None:
        fn = getattr(lib(), name)
        _CHECKED[name] = fn
    rc = fn(*args)
    if rc != 0:
        raise ClhipError(f"{name} failed ({rc}): {lib().clhip_last_error().decode()}")
    return rc


def require_gpu(t):
    if not t.is_cuda:
        raise ClhipError("libcontinual_amd runs its hot path as HIP kernels on an MI355X; got a "
                         f"{t.device} tensor and there is no CPU fallback")
