"""`torch.ops.vtp_hip.*` -- the gfx950 kernels of libvtp_hip.so registered with `torch.library` ("registered ...
so torch.ops.vtp_hip.* are callable from Python").  Every op is the C-ABI entry point of the same name (include/vtp_hip.h)
behind a torch schema: tensors provide device memory, the current HIP stream is used, outputs are pre-allocated by the caller
(`mutates_args`), so the ops compose with torch code, `torch.cuda.graph` capture and the dispatcher's tooling (fake tensors,
schema checks).  Differentiable tower-level entry points live in vtp_amd/autograd.py; these are the raw kernels.

    import vtp_amd.torch_ops            # registers the library (idempotent)
    torch.ops.vtp_hip.gemm_nt(a, b, c, bias, None, None, M, N, K, epilogue)
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops

_lib = torch.library.Library("vtp_hip", "DEF")
_defined = set()


def _define(schema: str, fn):
    name = schema.split("(")[0]
    if name in _defined:
        return
    _lib.define(schema)
    _lib.impl(name, fn, "CUDA")  # ROCm devices dispatch under the CUDA key
    _defined.add(name)


def _gemm_nt(a, b, c, bias, gamma, resid, M, N, K, epilogue):
    ops.gemm_nt(a, b, c, M=M, N=N, K=K, bias=bias, gamma=gamma, resid=resid, epi=epilogue)


def _gemm_swiglu(a, w12, b12, hidden, x12, M, N2, K):
    ops.gemm_nt(a, w12, hidden, M=M, N=N2, K=K, c2=x12, bias=b12, epi=ops.EPI_SWIGLU)


def _gemm_tn(a, b, c, M, N, K, accumulate, a_colsum):
    ops.gemm_tn(a, b, c, M=M, N=N, K=K, lda=a.stride(0), ldb=b.stride(0), ldc=c.stride(0), resid=c if accumulate else None,
                epi=ops.EPI_F32, a_colsum=a_colsum)


def _gemm_qkv_rope(a, w, bias, c, M, N, K, rope_pos, rope_sin, rope_cos, rope_cols):
    ops.gemm_qkv_rope(a, w, bias, c, M, N, K, rope_pos, rope_sin, rope_cos, rope_cols)


def _norm_fwd(x, w, b, y, stats, eps, kind):
    ops.norm_fwd(x, w, b, y, stats, x.shape[0], x.shape[1], eps, kind)


def _norm_bwd(dy, x, w, stats, dres, dx, dx_bf16, dw, db, kind):
    ops.norm_bwd(dy, x, w, stats, dres, dx, dx_bf16, dw, db, x.shape[0], x.shape[1], kind)


def _rope_qk(qkv, sin, cos, B, N, heads, prefix, inverse):
    ops.rope_qk(qkv, sin, cos, B, N, heads, prefix, inverse)


def _attn_fwd(qkv, o, lse, B, N, heads, scale, causal):
    D = heads * 64
    ops.attn_fwd(qkv, qkv[:, D:], qkv[:, 2 * D:], o, lse, B, N, heads, N * 3 * D, 3 * D, N * D, D, scale, causal)


def _attn_bwd(qkv, o, d_o, lse, delta, dqkv, B, N, heads, scale, causal):
    D = heads * 64
    ops.attn_bwd(qkv, qkv[:, D:], qkv[:, 2 * D:], o, d_o, lse, delta, dqkv, dqkv[:, D:], dqkv[:, 2 * D:], B, N, heads, N * 3 * D, 3 * D,
                 N * D, D, scale, causal)


def _swiglu_bwd(dh, x12, dx12, M, H):
    ops.swiglu_bwd(dh, x12, dx12, M, H)


def _im2col16(img, patches):
    ops.im2col16(img, patches, img.shape[0], img.shape[2], img.shape[3])


def _pixel_shuffle16(t, img, B, h, w):
    ops.pixel_shuffle16(t, img, B, h, w)


def _pixel_unshuffle16(d_img, dt, B, h, w):
    ops.pixel_unshuffle16(d_img, dt, B, h, w)


def _l1_loss_fwd_bwd(t, target, dt, loss_sum, B, h, w, gscale):
    ops.l1_loss_fwd_bwd(t, target, dt, loss_sum, B, h, w, gscale)


def _adamw(p, g, m, v, n, hyper):
    ops.adamw_dev(p, g, m, v, None, n, hyper)


def _ema(t, s, n, momentum):
    ops.ema(t, s, n, momentum)


def _clip_loss(img_l, txt_l, img_all, txt_all, logit_scale, label_offset, loss_sum, d_img_l, d_txt_l, d_img_all, d_txt_all,
               d_logit_scale, scratch):
    ops.clip_loss(img_l, txt_l, img_all, txt_all, logit_scale, img_l.shape[0], img_all.shape[0], img_l.shape[1], label_offset, loss_sum,
                  d_img_l, d_txt_l, d_img_all, d_txt_all, d_logit_scale, scratch)


def _dino_ce(s_logits, t_probs, t0, t1, w, inv_temp, loss_sum, d_logits):
    ops.dino_ce(s_logits, t_probs, t0, t1, w, inv_temp, loss_sum, d_logits, s_logits.shape[0], s_logits.shape[1])


_define("gemm_nt(Tensor a, Tensor b, Tensor(c!) c, Tensor? bias, Tensor? gamma, Tensor? resid, int M, int N, int K, int epilogue) -> ()",
        _gemm_nt)
_define("gemm_swiglu(Tensor a, Tensor w12, Tensor b12, Tensor(h!) hidden, Tensor(x!)? x12, int M, int N2, int K) -> ()", _gemm_swiglu)
_define("gemm_tn(Tensor a, Tensor b, Tensor(c!) c, int M, int N, int K, bool accumulate, Tensor(s!)? a_colsum) -> ()", _gemm_tn)
_define("gemm_qkv_rope(Tensor a, Tensor w, Tensor? bias, Tensor(c!) c, int M, int N, int K, Tensor rope_pos, Tensor rope_sin, "
        "Tensor rope_cos, int rope_cols) -> ()", _gemm_qkv_rope)
_define("norm_fwd(Tensor x, Tensor w, Tensor? b, Tensor(y!) y, Tensor(s!) stats, float eps, int kind) -> ()", _norm_fwd)
_define("norm_bwd(Tensor dy, Tensor x, Tensor w, Tensor stats, Tensor? dres, Tensor(a!) dx, Tensor(b!)? dx_bf16, Tensor(c!) dw, "
        "Tensor(d!)? db, int kind) -> ()", _norm_bwd)
_define("rope_qk(Tensor(q!) qkv, Tensor sin, Tensor cos, int B, int N, int heads, int prefix, bool inverse) -> ()", _rope_qk)
_define("attn_fwd(Tensor qkv, Tensor(o!) o, Tensor(l!) lse, int B, int N, int heads, float scale, bool causal) -> ()", _attn_fwd)
_define("attn_bwd(Tensor qkv, Tensor o, Tensor d_o, Tensor lse, Tensor(d!) delta, Tensor(g!) dqkv, int B, int N, int heads, float scale, "
        "bool causal) -> ()", _attn_bwd)
_define("swiglu_bwd(Tensor dh, Tensor x12, Tensor(d!) dx12, int M, int H) -> ()", _swiglu_bwd)
_define("im2col16(Tensor img, Tensor(p!) patches) -> ()", _im2col16)
_define("pixel_shuffle16(Tensor t, Tensor(i!) img, int B, int h, int w) -> ()", _pixel_shuffle16)
_define("pixel_unshuffle16(Tensor d_img, Tensor(t!) dt, int B, int h, int w) -> ()", _pixel_unshuffle16)
_define("l1_loss_fwd_bwd(Tensor t, Tensor target, Tensor(d!) dt, Tensor(l!) loss_sum, int B, int h, int w, float gscale) -> ()",
        _l1_loss_fwd_bwd)
_define("adamw(Tensor(p!) p, Tensor g, Tensor(m!) m, Tensor(v!) v, int n, Tensor hyper) -> ()", _adamw)
_define("ema(Tensor(t!) t, Tensor s, int n, float momentum) -> ()", _ema)
_define("clip_loss(Tensor img_l, Tensor txt_l, Tensor img_all, Tensor txt_all, Tensor logit_scale, int label_offset, Tensor(a!) loss_sum, "
        "Tensor(b!) d_img_l, Tensor(c!) d_txt_l, Tensor(d!) d_img_all, Tensor(e!) d_txt_all, Tensor(f!) d_logit_scale, "
        "Tensor(g!) scratch) -> ()", _clip_loss)
_define("dino_ce(Tensor s_logits, Tensor t_probs, Tensor t0, Tensor t1, Tensor w, float inv_temp, Tensor(l!) loss_sum, "
        "Tensor(d!) d_logits) -> ()", _dino_ce)

OPS = sorted(_defined)
