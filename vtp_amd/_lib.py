"""ctypes binding of libvtp_hip.so (include/vtp_hip.h).  The product path has NO fallback: if the library is
missing or a call is rejected, a RuntimeError is raised (never a silent eager/CPU path)."""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_long, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VTP_HIP_LIB") or os.path.join(_HERE, "lib", "libvtp_hip.so")  # override: A/B runs of two builds

_P, _I, _L, _F = c_void_p, c_int, c_long, c_float

# name -> argtypes (must mirror include/vtp_hip.h exactly; tests/test_abi.py checks the export list)
SIGNATURES = {
    "vtp_gemm_nt": [_P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P],
    "vtp_norm_fwd": [_P, _P, _P, _P, _P, _I, _I, _F, _I, _P],
    "vtp_norm_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],  # dy x w stats dres dx dxb dw db dxsum M D kind stream
    "vtp_rope_qk": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vtp_attn_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _L, _L, _L, _L, _F, _I, _P],
    "vtp_attn_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _L, _F, _I, _P],
    "vtp_im2col16": [_P, _P, _I, _I, _I, _P],
    "vtp_im2col16_rows": [_P, _P, _I, _I, _I, _I, _P],
    "vtp_col2im16": [_P, _P, _I, _I, _I, _P],
    "vtp_assemble_tokens": [_P, _P, _P, _P, _I, _I, _I, _P],
    "vtp_transpose_bf16": [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _P],  # in ld out ld colsum swiglu_h in_grp in_pre R C stream
    "vtp_strided_rowsum": [_P, _L, _P, _I, _I, _P],
    "vtp_mask_rows_bwd": [_P, _P, _P, _P, _I, _I, _I, _P],
    "vtp_token_rows_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "vtp_cast_f32_bf16": [_P, _P, _L, _P],
    "vtp_cast_transpose_f32_bf16": [_P, _P, _I, _I, _P],
    "vtp_prep_weights": [_P, _I, _I, _P],
    "vtp_prep_weights_range": [_P, _I, _I, _I, _P],
    "vtp_swiglu_bwd": [_P, _P, _P, _P, _I, _I, _P],
    "vtp_gemm_dgrad_swiglu": [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P],
    "vtp_gelu_bwd": [_P, _P, _P, _L, _P],
    "vtp_quick_gelu_bwd": [_P, _P, _P, _L, _P],
    "vtp_pixel_shuffle16": [_P, _P, _I, _I, _I, _P],
    "vtp_pixel_unshuffle16": [_P, _P, _I, _I, _I, _P],
    "vtp_l1_loss_fwd_bwd": [_P, _P, _P, _P, _I, _I, _I, _F, _P],
    "vtp_adamw": [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _F, _P],
    "vtp_adamw_dev": [_P, _P, _P, _P, _P, _L, _P, _P],
    "vtp_adamw_dev_masked": [_P, _P, _P, _P, _P, _P, _L, _P, _P],
    "vtp_reduce_slabs": [_P, _L, _I, _P, _L, _I, _P],
    "vtp_gemm_splits": [_I, _I],
    "vtp_gemm_tn_splits": [_I, _I, _I],
    "vtp_gemm_qkv_rope": [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P],
    "vtp_gemm_tn": [_P, _I, _P, _I, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "vtp_gemm_tn_grouped": [_P, _I, _I, _I, _I, _P, _P, _P],
    "vtp_gemm_tn_grouped_k": [_P, _I, _I, _I, _I, _P, _P, _I, _P],
    "vtp_gemm_tn_grouped_items": [_P, _I, _I, _I, _P, _I, _I, _P, _P, _P],
    "vtp_colsum_bf16": [_P, _I, _P, _I, _I, _I, _I, _I, _P],
    "vtp_colsum_bf16_rows": [_P, _I, _P, _P, _I, _I, _P],
    "vtp_gather_image_rows": [_P, _P, _P, _P, _I, _L, _I, _F, _P],
    "vtp_scatter_image_rows": [_P, _P, _P, _I, _L, _I, _F, _I, _P],
    "vtp_layerscale_wgrad": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "vtp_scaled_transpose": [_P, _P, _P, _I, _I, _P],
    "vtp_qk_norm_fwd": [_P, _P, _P, _P, _P, _L, _I, _F, _P],
    "vtp_qk_norm_bwd": [_P, _P, _P, _P, _P, _P, _P, _L, _I, _P],
    "vtp_set_gemm_tuning": [_I, _I],
    "vtp_gemm_nt_config": [_I, _I, _I, _I],
    "vtp_gemm_debug": [_P, _I, _I],
    "vtp_attn_debug": [_P, _I, _I, _I],
    "vtp_cu_thief": [_I, _I, _P, _P],
    "vtp_set_gemm_dynamic": [_I],
    "vtp_norm_fwd_e4m3": [_P, _P, _P, _P, _P, _P, _I, _I, _F, _I, _P],
    "vtp_quantize_e4m3": [_P, _I, _P, _L, _P, _F, _P],
    "vtp_amax": [_P, _I, _L, _P, _P],
    "vtp_dequantize_e4m3": [_P, _P, _L, _F, _P],
    "vtp_gemm_nt_fp8": [_P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _I, _P],
    "vtp_u8_to_images": [_P, _P, _L, _I, _I, _P, _P, _I, _P],
    "vtp_images_to_u8": [_P, _P, _L, _I, _I, _P, _P, _P],
    "vtp_latent_channel_stats": [_P, _P, _L, _I, _I, _P],
    "vtp_ema": [_P, _P, _L, _F, _P],
    "vtp_gather_token_rows": [_P, _P, _P, _I, _I, _P],
    "vtp_scatter_token_rows": [_P, _P, _P, _I, _I, _P],
    "vtp_weight_norm_prep": [_P, _P, _P, _P, _P, _I, _I, _P],
    "vtp_weight_norm_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _P],
    "vtp_softmax_center": [_P, _P, _F, _P, _I, _I, _P],
    "vtp_softmax_center_dev": [_P, _P, _P, _P, _I, _I, _P],
    "vtp_dino_ce": [_P, _P, _P, _P, _P, _F, _P, _P, _I, _I, _P],
    "vtp_center_ema": [_P, _P, _F, _P, _F, _I, _P],
    "vtp_conv3x3": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "vtp_lpips_unfold3": [_P, _P, _P, _I, _I, _I, _P, _P, _P],
    "vtp_lpips_fold3_bwd": [_P, _P, _I, _I, _I, _P, _P],
    "vtp_maxpool2_fwd": [_P, _P, _I, _I, _I, _I, _P],
    "vtp_maxpool2_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vtp_lpips_tap": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "vtp_ema_dev": [_P, _P, _L, _P, _P],
    "vtp_adamw_ema_dev": [_P, _P, _P, _P, _P, _P, _L, _P, _P],
    "vtp_embed_tokens": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "vtp_embed_tokens_bwd": [_P, _P, _P, _P, _I, _I, _I, _P],
    "vtp_gather_rows": [_P, _P, _P, _I, _I, _I, _P],
    "vtp_scatter_rows": [_P, _P, _P, _P, _I, _I, _I, _P],
    "vtp_l2norm_fwd": [_P, _P, _P, _I, _I, _F, _P],
    "vtp_l2norm_bwd": [_P, _P, _P, _P, _I, _I, _P],
    "vtp_clip_logits": [_P, _P, _P, _P, _I, _I, _I, _P],
    "vtp_clip_grad_rows": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vtp_clip_grad_cols": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vtp_siglip_pairs": [_P, _P, _I, _I, _I, _F, _P, _P, _P, _P],
    "vtp_koleo": [_P, _P, _P, _P, _I, _I, _F, _F, _P],
    "vtp_sinkhorn_knopp": [_P, _F, _P, _P, _P, _P, _I, _I, _F, _P, _P, _I, _I, _P],
    "vtp_clip_loss": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P],
}

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"vtp_amd: {LIB_PATH} not found. The HIP library is mandatory (there is no fallback path): "
            "run `python -c 'import __graft_entry__ as g; g.build()'` or `python vtp_amd/build.py`.")
    # On a GPU box the HIP runtime must be up BEFORE the library's fat binaries register themselves (dlopen): loading it first
    # (e.g. __graft_entry__.build() followed by smoke() in one process) left every later launch of ours failing with "no
    # ROCm-capable device is detected" although torch's own kernels ran (ROCm 7.2, measured round 4).  No GPU: nothing to do.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:  # the ABI / symbol check of tests/test_abi.py does not need torch
        pass
    lib = ctypes.CDLL(LIB_PATH)
    lib.vtp_abi_version.restype = c_int
    lib.vtp_last_error.restype = c_char_p
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale -> loud
        fn.argtypes = args
        fn.restype = c_int
    if lib.vtp_abi_version() != 1:
        raise RuntimeError("vtp_amd: ABI version mismatch between vtp_amd/_lib.py and libvtp_hip.so")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = _lib.vtp_last_error().decode(errors="replace") if _lib is not None else ""
        raise RuntimeError(f"{what} failed (rc={rc}): {msg}")
