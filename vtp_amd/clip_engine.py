"""CLIP side of the VTP hot path on the gfx950 kernels: text tower (TextTransformer pieces as re-hung by
VTPModel._init_text_components, modeling_vtp.py:135-178) and the image/text projection + normalisation heads
(modeling_vtp.py:244-333).  The contrastive loss itself is our spec (OpenCLIP convention, parity unpinned)."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops
from .engine import BF, F32, OVERLAP, ParamStore, Stack, Workspace, linear_bwd
from .ops import EPI_BF16, EPI_F32

I32 = torch.int32


class TextEngine:
    def __init__(self, store: ParamStore, cfg):
        self.store = store
        self.D, self.heads, self.depth, self.T = cfg.text_embed_dim, cfg.text_num_heads, cfg.text_depth, cfg.text_num_pos
        self.H = int(self.D * cfg.text_mlp_ratio)
        self.Dout = cfg.text_embed_dim  # output_dim = text_embed_dim (modeling_vtp.py:150)
        self.stack = Stack(store, "text_transformer.resblocks.", self.depth, self.D, self.heads, self.H, "layernorm",
                           style="text")
        # text_transformer.py:285-288: no_causal_mask drops the additive causal mask (full attention over the T tokens);
        # text_global_pool (:213-228): the pooled row is the first / last token or the arg-max id (EOT) of every caption
        self.stack.causal = not cfg.text_no_causal_mask
        self.stack.quick_gelu = bool(cfg.text_quick_gelu)  # act_layer = QuickGELU (modeling_vtp.py:139)
        self.pool = cfg.text_pool_type
        # text_projection is stored [width, output_dim] and applied as x @ P (modeling_vtp.py:308): as a Lin with
        # N = width, K = output_dim its bf16 copy `w` is P and `wT` is P^T (the K-contiguous operand of the forward GEMM)
        self.proj = store.lin("text_projection", None, self.D, self.Dout)
        self.ws: Dict[tuple, Workspace] = {}

    def workspace(self, B) -> Workspace:
        if B not in self.ws:
            self.ws[B] = Workspace(self.store.device)
        return self.ws[B]

    def forward(self, ids: torch.Tensor, train: bool) -> torch.Tensor:
        """ids int64 [B, T] (device) -> un-normalised text features f32 [B, Dout]  (modeling_vtp.py:278-310); text_pool_type = "none"
        (text_global_pool's fall-through, text_transformer.py:225-226): every token, f32 [B * T, Dout]."""
        st = self.store
        B, T = ids.shape
        D = self.D
        ws = self.workspace(B)
        x0 = ws.get("x0", (B * T, D), F32)
        eot = ws.get("eot", (B,), I32)
        ops.embed_tokens(ids, st.p("token_embedding.weight"), st.p("positional_embedding"), x0, eot, B, T, D)
        if self.pool in ("first", "last"):  # the kernel wrote the arg-max position of every row; 'first' / 'last' pool a fixed position
            eot.fill_(0 if self.pool == "first" else T - 1)
        xl = self.stack.forward(ws, x0, B, T, None, 0, train)
        if self.pool == "none":  # ln_final and the projection over all B * T rows
            xn = ws.get("all_n", (B * T, D), BF)
            stf = ws.get("all_stf", (B * T, 2), F32)
            ops.norm_fwd(xl, st.p("ln_final.weight"), st.p("ln_final.bias"), xn, stf, B * T, D, 1e-5, ops.NORM_LN)
            feat = ws.get("all_feat", (B * T, self.Dout), F32)
            ops.gemm_nt(xn, self.proj.wT, feat, M=B * T, N=self.Dout, K=D, epi=EPI_F32)
            self._ctx = (ws, ids, B, T, None, xl, xn, stf)
            return feat
        pooled = ws.get("pooled", (B, D), F32)
        ops.gather_rows(xl, eot, pooled, B, T, D)  # ln_final is row-wise: pool first, normalise B rows instead of B*T
        pn = ws.get("pooled_n", (B, D), BF)
        stf = ws.get("stf", (B, 2), F32)
        ops.norm_fwd(pooled, st.p("ln_final.weight"), st.p("ln_final.bias"), pn, stf, B, D, 1e-5, ops.NORM_LN)
        feat = ws.get("feat", (B, self.Dout), F32)
        ops.gemm_nt(pn, self.proj.wT, feat, M=B, N=self.Dout, K=D, epi=EPI_F32)
        self._ctx = (ws, ids, B, T, eot, pooled, pn, stf)
        return feat

    def backward(self, d_feat: torch.Tensor):
        """d_feat f32 [B, Dout].  Generator (see Stack.backward): yields "tail" then ("block", i); parameter gradients
        accumulate into store.flat_g."""
        st = self.store
        ws, ids, B, T, eot, pooled, pn, stf = self._ctx
        D = self.D
        if eot is None:  # text_pool_type = "none": the same head over all B * T rows, no pooling scatter
            R = B * T
            d_feat_b = ws.get("b.all_d_feat_b", (R, self.Dout), BF)
            ops.cast_f32_bf16(d_feat, d_feat_b, R * self.Dout)
            linear_bwd(ws, "tproj", None, pn, d_feat_b, R, None, need_dx=False, N=D, K=self.Dout, gw=self.proj.gw, gb=None, wT=None)
            d_pn = ws.get("b.all_d_pn", (R, D), BF)
            ops.gemm_nt(d_feat_b, self.proj.w, d_pn, M=R, N=D, K=self.Dout, epi=EPI_BF16)
            dx = ws.get("b.dxt", (R, D), F32)
            dx_b = ws.get("b.dxt_b", (R, D), BF)
            ops.norm_bwd(d_pn, pooled, st.p("ln_final.weight"), stf, None, dx, dx_b, st.g("ln_final.weight"), st.g("ln_final.bias"), R, D,
                         ops.NORM_LN)
            OVERLAP.join()
            yield "tail"
            dx0, _ = yield from self.stack.backward(ws, dx, dx_b, B, T, None, 0)
            ops.embed_tokens_bwd(ids, dx0, st.g("token_embedding.weight"), st.g("positional_embedding"), B, T, D)
            OVERLAP.join()
            return
        d_feat_b = ws.get("b.d_feat_b", (B, self.Dout), BF)
        ops.cast_f32_bf16(d_feat, d_feat_b, B * self.Dout)
        # dP [D, Dout] += pn^T d_feat  (roles of "dy" and "x" swapped so the result lands in the parameter's layout)
        linear_bwd(ws, "tproj", None, pn, d_feat_b, B, None, need_dx=False, N=D, K=self.Dout, gw=self.proj.gw, gb=None,
                   wT=None)
        d_pn = ws.get("b.d_pn", (B, D), BF)
        ops.gemm_nt(d_feat_b, self.proj.w, d_pn, M=B, N=D, K=self.Dout, epi=EPI_BF16)  # d_pn = d_feat P^T
        d_pooled = ws.get("b.d_pooled", (B, D), F32)
        ops.norm_bwd(d_pn, pooled, st.p("ln_final.weight"), stf, None, d_pooled, None, st.g("ln_final.weight"),
                     st.g("ln_final.bias"), B, D, ops.NORM_LN)
        dx = ws.get("b.dxt", (B * T, D), F32)
        dx_b = ws.get("b.dxt_b", (B * T, D), BF)
        ops.scatter_rows(d_pooled, eot, dx, dx_b, B, T, D)
        OVERLAP.join()
        yield "tail"
        dx0, _ = yield from self.stack.backward(ws, dx, dx_b, B, T, None, 0)
        ops.embed_tokens_bwd(ids, dx0, st.g("token_embedding.weight"), st.g("positional_embedding"), B, T, D)
        OVERLAP.join()


class ClipHead:
    """visual_proj + F.normalize on both modalities + contrastive loss with (optional) cross-rank feature exchange."""

    def __init__(self, store: ParamStore, vproj, Dv: int, Dt: int):
        self.store, self.vproj, self.Dv, self.Dt = store, vproj, Dv, Dt
        self.fused_ok = vproj.K == Dv  # cls token of the un-bottlenecked trunk output (the GEMMs below read K = Dv columns)
        self.ws = Workspace(store.device)

    def image_features(self, xnf: torch.Tensor, B: int, N: int) -> torch.Tensor:
        """cls rows of the final-norm token matrix (bf16 [B*N, Dv]) -> un-normalised image features f32 [B, Dt]."""
        assert self.fused_ok, "ClipHead: visual_proj input width != trunk width (bottlenecked CLIP feature): use the autograd path"
        f = self.ws.get(f"f_img{B}", (B, self.Dt), F32)
        ops.gemm_nt(xnf, self.vproj.w, f, M=B, N=self.Dt, K=self.Dv, lda=N * self.Dv, epi=EPI_F32)
        return f

    def normalize(self, f: torch.Tensor, tag: str):
        B, D = f.shape
        y = self.ws.get(f"n_{tag}{B}", (B, D), F32)
        inv = self.ws.get(f"inv_{tag}{B}", (B,), F32)
        ops.l2norm_fwd(f, y, inv, B, D, 1e-12)
        return y, inv

    def normalize_bwd(self, dy, y, inv, tag: str):
        B, D = y.shape
        dx = self.ws.get(f"dn_{tag}{B}", (B, D), F32)
        ops.l2norm_bwd(dy, y, inv, dx, B, D)
        return dx

    def image_backward(self, d_f: torch.Tensor, xnf: torch.Tensor, d_xnf: torch.Tensor, B: int, N: int):
        """d_f f32 [B, Dt] -> dW(visual_proj) and the cls rows of d_xnf (bf16 [B*N, Dv])."""
        d_f_b = self.ws.get(f"d_f_b{B}", (B, self.Dt), BF)
        ops.cast_f32_bf16(d_f, d_f_b, B * self.Dt)
        cls_rows = xnf.view(B, N * self.Dv)[:, :self.Dv]        # strided views: row b = token b*N
        d_cls_rows = d_xnf.view(B, N * self.Dv)[:, :self.Dv]
        linear_bwd(self.ws, "vproj", self.vproj, d_f_b, cls_rows, B, d_cls_rows)
