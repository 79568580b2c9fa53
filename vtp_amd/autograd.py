"""Autograd visibility of the hand-scheduled towers: `loss.backward()` through `VTPModel` / `VTP` works like it does through
the reference's nn.Modules (the reference's users call `model(...)` and then `loss.backward()`, e.g.
tools/test_linear_probing_hf.py:285-290; training: VTP.forward(forward_type=...), vtp/models/vtp.py:323-338).

Granularity: one `torch.autograd.Function` per tower pass (trunk -> tokens / latents, pixel decoder, text tower, SSL student).
Each forward runs the engine's kernel sequence with activations saved in its static workspace; each backward runs the engine's
hand-written backward, which ACCUMULATES the parameter gradients into the flat fp32 gradient buffer -- and every
`nn.Parameter.grad` of the model is a view of that buffer (engine.ParamStore), so after `loss.backward()` `p.grad` holds what
torch autograd would have produced and any torch optimizer (or the fused VTPTrainer AdamW) can consume it.  Parameters are not
inputs of the Functions (600 tensors per call would cost more host time than the kernels); a scalar `anchor` tensor that
requires grad keeps the Functions on the autograd tape.

Static workspaces mean ONE forward per (tower, tag, shape) can wait for its backward; a later forward of the same kind
overwrites the saved activations and a backward through the earlier one raises (see _Flight).
The small heads on top (cls / mean pooling, bottleneck, visual_proj, F.normalize, logits) are kernel-backed Functions as well
(HeadLinear / SumTokens / L2Normalize / ClipLogits at the end of this file).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from .engine import BF, F32


def _drain(gen):
    """run an engine backward generator (it yields gradient-bucket boundaries for the DDP trainer) to completion"""
    try:
        while True:
            next(gen)
    except StopIteration as stop:
        return stop.value


class _Flight:
    """The engines keep activations in static workspaces: a later forward of the same (tower, tag, shape) overwrites what an
    earlier one saved.  Like a freed autograd graph, that is fine as long as nobody calls backward on the earlier one: every
    forward takes a ticket, backward checks that its ticket is still the current one."""

    def __init__(self):
        self.current = {}
        self.n = 0

    def take(self, key) -> int:
        self.n += 1
        self.current[key] = self.n
        return self.n

    def check(self, key, ticket):
        if self.current.get(key) != ticket:
            raise RuntimeError(f"vtp_amd: backward through a forward pass of {key} whose saved activations were overwritten by a "
                               "later forward of the same kind and shape (static workspaces: one pass per (tower, shape) can "
                               "wait for its backward; run inference under torch.no_grad() / model.eval())")


def _flight(model) -> _Flight:
    f = getattr(model, "_ag_flight", None)
    if f is None:
        f = model._ag_flight = _Flight()
    return f


def anchor(model) -> torch.Tensor:
    a = getattr(model, "_ag_anchor", None)
    dev = model.trunk.cls_token.device
    if a is None or a.device != dev:
        a = model._ag_anchor = torch.zeros((), device=dev, requires_grad=True)
    return a


def grad_mode(model) -> bool:
    """differentiable path: autograd recording is on and the module is in training mode (inference callers use model.eval()
    and / or torch.no_grad(), like the reference's tools do)"""
    return torch.is_grad_enabled() and model.training


# ---------------------------------------------------------------------------------------------------------------- trunk
class TrunkTokens(torch.autograd.Function):
    """image [B,3,H,W] -> final-norm tokens f32 [B, 1+hw, D]  (DinoVisionTransformer.forward_features, vision_transformer.py:221-258)"""

    @staticmethod
    def forward(ctx, image, anchor_, model, tag):
        st = model._fresh()
        st.sync_grad_views()
        img = model._img(image)
        B, _, H, W = img.shape
        key = ("trunk", tag, tuple(img.shape))
        ctx.ticket = _flight(model).take(key)
        xnf = model._trunk.forward(img, train=True, tag=tag)
        ctx.model, ctx.key, ctx.tctx, ctx.shape = model, key, model._trunk.ctx(), (B, (H // 16) * (W // 16) + 1)
        ctx.img_dtype = image.dtype
        return xnf.float().view(B, ctx.shape[1], -1).clone()

    @staticmethod
    def backward(ctx, d_tokens):
        model, tr = ctx.model, ctx.model._trunk
        _flight(model).check(ctx.key, ctx.ticket)
        model._store.sync_grad_views()
        d_xnf = tr.d_xnf_buffer(ctx.tctx)
        d_xnf.copy_(d_tokens.reshape(d_xnf.shape))  # f32 -> bf16, every row written
        want = ctx.needs_input_grad[0]  # the image itself requires grad (the reference's autograd would deliver d/d image)
        _drain(tr.backward(None, ctx=ctx.tctx, want_dimg=want))
        return (ctx.tctx.take_d_img(0).to(ctx.img_dtype) if want else None), None, None, None


class EncodeLatents(torch.autograd.Function):
    """image -> bottleneck latents f32 [B, 64, h, w]  (get_reconstruction_latents, modeling_vtp.py:337-360)"""

    @staticmethod
    def forward(ctx, image, anchor_, model, tag):
        st = model._fresh()
        st.sync_grad_views()
        img = model._img(image)
        B, _, H, W = img.shape
        h, w = H // 16, W // 16
        key = ("trunk", tag, tuple(img.shape))
        ctx.ticket = _flight(model).take(key)
        model._trunk.forward(img, train=True, tag=tag)
        lat = model._trunk.latents(out_f32=True)
        ctx.model, ctx.key, ctx.tctx, ctx.dims = model, key, model._trunk.ctx(), (B, h, w)
        ctx.img_dtype = image.dtype
        return lat.view(B, h * w, -1).transpose(1, 2).reshape(B, -1, h, w).clone()

    @staticmethod
    def backward(ctx, d_lat):
        model, tr = ctx.model, ctx.model._trunk
        _flight(model).check(ctx.key, ctx.ticket)
        model._store.sync_grad_views()
        B, h, w = ctx.dims
        d_tok = d_lat.reshape(B, -1, h * w).transpose(1, 2).to(BF).contiguous().view(B * h * w, -1)
        d_xnf = tr.d_xnf_buffer(ctx.tctx)
        d_xnf.zero_()  # the bottleneck dgrad writes the patch rows only: the cls rows carry no gradient on this path
        want = ctx.needs_input_grad[0]
        _drain(tr.backward(d_tok, ctx=ctx.tctx, want_dimg=want))
        return (ctx.tctx.take_d_img(0).to(ctx.img_dtype) if want else None), None, None, None


# ---------------------------------------------------------------------------------------------------------------- pixel decoder
class DecodeLatents(torch.autograd.Function):
    """latents f32 [B, 64, h, w] -> image f32 [B, 3, 16h, 16w]  (DinoV3PixelDecoder.forward, pixel_decoder.py:134-162)"""

    @staticmethod
    def forward(ctx, latents, anchor_, model):
        st = model._fresh()
        st.sync_grad_views()
        B, C, h, w = latents.shape
        key = ("decoder", (B, h, w))
        ctx.ticket = _flight(model).take(key)
        lat = latents.detach().reshape(B, C, h * w).transpose(1, 2).to(BF).contiguous().view(B * h * w, C)
        t = model._decoder.forward(lat, B, h, w, train=True)
        img = torch.empty(B, 3, h * 16, w * 16, dtype=F32, device=lat.device)
        ops.pixel_shuffle16(t, img, B, h, w)
        ctx.model, ctx.key, ctx.dctx, ctx.dims = model, key, model._decoder._ctx, (B, C, h, w)
        return img

    @staticmethod
    def backward(ctx, d_img):
        model, dec = ctx.model, ctx.model._decoder
        _flight(model).check(ctx.key, ctx.ticket)
        model._store.sync_grad_views()
        B, C, h, w = ctx.dims
        dec._ctx = ctx.dctx
        dt = ctx.dctx[0].get("b.dt", (B * h * w, 768), BF)
        ops.pixel_unshuffle16(d_img.contiguous().float(), dt, B, h, w)
        d_lat = _drain(dec.backward(dt))  # bf16 [B*hw, C]
        return d_lat.float().view(B, h * w, C).transpose(1, 2).reshape(B, C, h, w), None, None


# ---------------------------------------------------------------------------------------------------------------- text tower
class TextFeature(torch.autograd.Function):
    """token ids [B, T] -> un-normalised text features f32 [B, D_t]  (get_clip_text_feature / encode_text, vtp.py:293-312)"""

    @staticmethod
    def forward(ctx, ids, anchor_, model):
        st = model._fresh()
        st.sync_grad_views()
        key = ("text", tuple(ids.shape))
        ctx.ticket = _flight(model).take(key)
        f = model._text.forward(ids, train=True)
        ctx.model, ctx.key, ctx.tctx = model, key, model._text._ctx
        return f.clone()

    @staticmethod
    def backward(ctx, d_feat):
        model = ctx.model
        _flight(model).check(ctx.key, ctx.ticket)
        model._store.sync_grad_views()
        model._text._ctx = ctx.tctx
        _drain(model._text.backward(d_feat.contiguous().float()))
        return None, None, None


# ---------------------------------------------------------------------------------------------------------------- SSL student
class SSLStudent(torch.autograd.Function):
    """The student side of VTP.forward_ssl_learning (vtp.py:452-484) as one differentiable op: masked global crops + local
    crops through the trunk (one list forward) and the DINO head.  Returns the four tensors of the reference's
    student_outputs dict (logits as f32 copies of the bf16 kernel outputs); the teacher side carries no gradient."""

    @staticmethod
    def forward(ctx, anchor_, model, global_crops, local_crops, masks_u8, plan):
        from .vtp import ssl_forward
        st = model._fresh()
        st.sync_grad_views()
        key = ("ssl", tuple(global_crops.shape), tuple(local_crops.shape), plan["Ts"])
        ctx.ticket = _flight(model).take(key)
        out = ssl_forward(model, global_crops, local_crops, masks_u8, plan, train=True)
        nm, nl, B2 = plan["n_masked"], out["nl"], out["B2"]
        sl = out["student_logits"].float()
        ctx.model, ctx.key, ctx.out, ctx.plan = model, key, out, plan
        ctx.teacher = out["teacher_logits"].float()
        ctx.mark_non_differentiable(ctx.teacher)
        return (sl[:nl].clone(), sl[nl:nl + B2].clone(), out["student_global_cls"].float().clone(), sl[nl + B2:nl + B2 + nm].clone(),
                ctx.teacher)

    @staticmethod
    def backward(ctx, d_local, d_global, d_cls, d_patch, _d_teacher):
        model, out, plan = ctx.model, ctx.out, ctx.plan
        _flight(model).check(ctx.key, ctx.ticket)
        st = model._store
        st.sync_grad_views()
        head, tr = model._head, model._trunk
        Ts, nl, B2, nm = out["Ts"], out["nl"], out["B2"], plan["n_masked"]
        K, D = head.K, tr.D
        d_logits = out["ws"].get("d_logits", (Ts, K), BF)
        d_logits.zero_()
        if d_local is not None:
            d_logits[:nl].copy_(d_local)
        if d_global is not None:
            d_logits[nl:nl + B2].copy_(d_global)
        if d_patch is not None and nm > 0:
            d_logits[nl + B2:nl + B2 + nm].copy_(d_patch)
        dX = head.backward(d_logits, out["head_ctx"])  # bf16 [Ts, D]
        if d_cls is not None:  # student_global_cls_tokens are the head's own input rows
            dX[nl:nl + B2] += d_cls.to(BF)
        tctx = out["ctx"]
        d_xnf = tr.d_xnf_buffer(tctx)
        d_xnf.zero_()
        seg_g, seg_l = tctx.segs[-2], tctx.segs[-1]
        idx = out["idx"]
        ops.scatter_token_rows(dX, idx["student_local_src"], d_xnf[seg_l.row0:], nl, D)
        ops.scatter_token_rows(dX[nl:], idx["student_global_src"], d_xnf[seg_g.row0:], Ts - nl, D)
        _drain(tr.backward(None, ctx=tctx))
        return None, None, None, None, None, None


# ---------------------------------------------------------------------------------------------------------------- feature heads
# The [B, D]-sized heads behind get_clip_image_feature / get_clip_logits (modeling_vtp.py:244-333): bottleneck / visual_proj
# projections, token pooling, F.normalize and the logit matrix -- on the same kernels as the fused trainer (bf16 MFMA GEMM with
# fp32 accumulation = what the reference's Linear layers do under bf16 autocast), each with a hand-written backward.
def _ws(model):
    return model._clip.ws


class HeadLinear(torch.autograd.Function):
    """y f32 [R, N] = alpha * bf16(x) W^T for a bias-free projection held by the engine (`lin`: engine.Lin with bf16 W / W^T copies
    and the flat-gradient view); backward: dx = alpha * dy W, dW += alpha * dy^T x accumulated into the flat gradient buffer --
    unless the weight is frozen (`wgrad` False: its nn.Parameter does not require a gradient; the GEMM is skipped)."""

    @staticmethod
    def forward(ctx, x, anchor_, model, lin, alpha, wgrad=True):
        st = model._fresh()
        st.sync_grad_views()
        R, K = x.shape
        assert K == lin.K
        x_b = torch.empty(R, K, dtype=BF, device=x.device)
        ops.cast_f32_bf16(x.contiguous().float(), x_b, R * K)
        y = torch.empty(R, lin.N, dtype=F32, device=x.device)
        ops.gemm_nt(x_b, lin.w, y, M=R, N=lin.N, K=K, epi=ops.EPI_F32, alpha=alpha)
        ctx.model, ctx.lin, ctx.alpha, ctx.x_b, ctx.wgrad = model, lin, alpha, x_b, bool(wgrad)
        return y

    @staticmethod
    def backward(ctx, dy):
        lin, x_b = ctx.lin, ctx.x_b
        ctx.model._store.sync_grad_views()
        R, K = x_b.shape
        dy_b = torch.empty(R, lin.N, dtype=BF, device=dy.device)
        ops.cast_f32_bf16(dy.contiguous().float(), dy_b, R * lin.N)
        dx = torch.empty(R, K, dtype=F32, device=dy.device)
        ops.gemm_nt(dy_b, lin.wT, dx, M=R, N=K, K=lin.N, epi=ops.EPI_F32, alpha=ctx.alpha)
        if ctx.wgrad:
            if ctx.alpha != 1.0:  # dW += alpha * dy^T x: fold alpha into dy (one more rounding of a [R, N] matrix)
                ops.cast_f32_bf16((dy * ctx.alpha).contiguous().float(), dy_b, R * lin.N)
            ops.gemm_tn(dy_b, x_b, lin.gw, M=lin.N, N=K, K=R, lda=lin.N, ldb=K, ldc=K, resid=lin.gw, epi=ops.EPI_F32)
        return dx, None, None, None, None, None


class SumTokens(torch.autograd.Function):
    """tokens f32 [B, n, D] -> their sum over n, f32 [B, D] (the mean pooling of vision_clip_feat = 'pooled'; the 1 / n rides in the
    following projection's alpha)"""

    @staticmethod
    def forward(ctx, tokens):
        B, n, D = tokens.shape
        t = tokens.contiguous().float()
        out = torch.zeros(B, D, dtype=F32, device=t.device)
        for b in range(B):
            ops.strided_rowsum(t[b], D, out[b], n, D)
        ctx.shape = (B, n, D)
        return out

    @staticmethod
    def backward(ctx, d):
        B, n, D = ctx.shape
        return d[:, None, :].expand(B, n, D).contiguous()


class L2Normalize(torch.autograd.Function):
    """F.normalize(x, dim=-1) (eps 1e-12) on the l2norm kernels"""

    @staticmethod
    def forward(ctx, x):
        B, D = x.shape
        x = x.contiguous().float()
        y, inv = torch.empty_like(x), torch.empty(B, dtype=F32, device=x.device)
        ops.l2norm_fwd(x, y, inv, B, D, 1e-12)
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        dx = torch.empty_like(y)
        ops.l2norm_bwd(dy.contiguous().float(), y, inv, dx, y.shape[0], y.shape[1])
        return dx


class ClipLogits(torch.autograd.Function):
    """logits f32 [M, N] = exp(logit_scale) * I T^T (modeling_vtp.py:326-329) on the clip_logits kernel; backward through the
    clip_grad_rows / clip_grad_cols kernels, d logit_scale = sum(dL . logits)"""

    @staticmethod
    def forward(ctx, i, t, logit_scale):
        i, t = i.contiguous().float(), t.contiguous().float()
        M, D = i.shape
        N = t.shape[0]
        out = torch.empty(M, N, dtype=F32, device=i.device)
        ops.clip_logits(i, t, logit_scale.detach().reshape(1), out, M, N, D)
        ctx.save_for_backward(i, t, logit_scale.detach().reshape(1), out)
        return out

    @staticmethod
    def backward(ctx, dL):
        i, t, ls, logits = ctx.saved_tensors
        M, D = i.shape
        N = t.shape[0]
        G = dL.contiguous().float()
        di, dt = torch.empty_like(i), torch.empty_like(t)
        L = ops._lib_()
        from . import _lib
        s = ops._s()
        _lib.check(L.vtp_clip_grad_rows(ops._p(G), ops._p(t), ops._p(ls), ops._p(di), M, N, D, 0, s), "vtp_clip_grad_rows")
        _lib.check(L.vtp_clip_grad_cols(ops._p(G), ops._p(i), ops._p(ls), ops._p(dt), M, N, D, 0, s), "vtp_clip_grad_cols")
        d_ls = torch.dot(G.reshape(-1), logits.reshape(-1)).reshape(())
        return di, dt, d_ls
