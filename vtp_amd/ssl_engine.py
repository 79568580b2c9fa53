"""Self-supervised (DINO / iBOT) side of the VTP hot path on the gfx950 kernels.

Reference pieces this replaces: DINOHead (vtp/models/heads/dino_head.py:7-89, weight-normed last layer :47-49), the
teacher / student token-buffer assembly of VTP.get_teacher_forward_outputs / get_student_ssl_outputs
(vtp/models/vtp.py:410-484) and VTP.update_teacher (vtp.py:388-401).  The SSL *losses* are not in the reference: ours
follow DINOv2 (softmax-centred teacher targets, student temperature 0.1, cross-view DINO term + masked-patch iBOT term)
and are an explicit spec here (parity unpinned)."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import ops
from .engine import BF, F32, OVERLAP, ParamStore, Workspace, linear_bwd
from .ops import EPI_BF16, EPI_F32, EPI_GELU

I32 = torch.int32


class DinoHeadEngine:
    """MLP(in -> hidden -> hidden -> bottleneck, GELU) -> L2-normalise -> weight-normed Linear(bottleneck -> K)."""

    def __init__(self, store: ParamStore, prefix: str, Din: int, hidden: int, bott: int, K: int):
        self.store, self.prefix, self.Din, self.hidden, self.bott, self.K = store, prefix, Din, hidden, bott, K
        self.l1 = store.lin(prefix + "mlp.0.weight", prefix + "mlp.0.bias", hidden, Din)
        self.l2 = store.lin(prefix + "mlp.2.weight", prefix + "mlp.2.bias", hidden, hidden)
        self.l3 = store.lin(prefix + "mlp.4.weight", prefix + "mlp.4.bias", bott, hidden)
        self.v, self.g = store.p(prefix + "last_layer.weight_v"), store.p(prefix + "last_layer.weight_g")
        self.gv, self.gg = store.g(prefix + "last_layer.weight_v"), store.g(prefix + "last_layer.weight_g")
        dev = store.device
        self.weff = torch.empty(K, bott, dtype=BF, device=dev)
        self.weffT = torch.empty(bott, K, dtype=BF, device=dev)
        self.inv_norm = torch.empty(K, dtype=F32, device=dev)
        if not hasattr(store, "prep_hooks"):
            store.prep_hooks = []
        store.prep_hooks.append(self.prep)
        # what the hook reads: the trainer's optimizer lane runs it right behind the bucket update that covers these ranges (beside the
        # backward) instead of in the serial tail of the step
        ov, kv = store.offsets[prefix + "last_layer.weight_v"]
        og, kg = store.offsets[prefix + "last_layer.weight_g"]
        store.__dict__.setdefault("hook_deps", {})[self.prep] = [(ov, ov + kv), (og, og + kg)]
        self.ws: Dict[tuple, Workspace] = {}

    def prep(self):
        ops.weight_norm_prep(self.v, self.g, self.weff, self.weffT, self.inv_norm, self.K, self.bott)

    def workspace(self, T: int, tag: str) -> Workspace:
        key = (T, tag)
        if key not in self.ws:
            self.ws[key] = Workspace(self.store.device)
        return self.ws[key]

    def forward(self, X: torch.Tensor, T: int, tag: str = ""):
        """X bf16 [T, Din] -> prototype logits bf16 [T, K]; returns (logits, ctx)."""
        ws = self.workspace(T, tag)
        hid, bott, K = self.hidden, self.bott, self.K
        pre1, h1 = ws.get("pre1", (T, hid), BF), ws.get("h1", (T, hid), BF)
        pre2, h2 = ws.get("pre2", (T, hid), BF), ws.get("h2", (T, hid), BF)
        z, zn = ws.get("z", (T, bott), F32), ws.get("zn", (T, bott), F32)
        inv_z = ws.get("inv_z", (T,), F32)
        zn_b = ws.get("zn_b", (T, bott), BF)
        logits = ws.get("logits", (T, K), BF)
        ops.gemm_nt(X, self.l1.w, h1, M=T, N=hid, K=self.Din, c2=pre1, ldc2=hid, bias=self.l1.bias, epi=EPI_GELU)
        ops.gemm_nt(h1, self.l2.w, h2, M=T, N=hid, K=hid, c2=pre2, ldc2=hid, bias=self.l2.bias, epi=EPI_GELU)
        ops.gemm_nt(h2, self.l3.w, z, M=T, N=bott, K=hid, bias=self.l3.bias, epi=EPI_F32)
        ops.l2norm_fwd(z, zn, inv_z, T, bott, 1e-12)  # F.normalize(x, eps=1e-12)  (dino_head.py:82-84)
        ops.cast_f32_bf16(zn, zn_b, T * bott)
        ops.gemm_nt(zn_b, self.weff, logits, M=T, N=K, K=bott, epi=EPI_BF16)
        return logits, (ws, X, T, pre1, h1, pre2, h2, zn, inv_z, zn_b)

    def backward(self, d_logits: torch.Tensor, ctx) -> torch.Tensor:
        """d_logits bf16 [T, K] -> dX bf16 [T, Din]; parameter gradients accumulate into store.flat_g."""
        ws, X, T, pre1, h1, pre2, h2, zn, inv_z, zn_b = ctx
        hid, bott, K = self.hidden, self.bott, self.K
        # last layer (weight-normed): dW_eff = d_logits^T zn ; d_zn = d_logits W_eff
        dWeff = ws.get("b.dWeff", (K, bott), F32)
        ops.gemm_tn(d_logits, zn_b, dWeff, M=K, N=bott, K=T, lda=K, ldb=bott, ldc=bott, epi=EPI_F32)
        ops.weight_norm_bwd(dWeff, self.v, self.g, self.inv_norm, self.gv, self.gg, K, bott)
        d_zn = ws.get("b.d_zn", (T, bott), F32)
        # reduction over K = 65536 prototypes with only T/128 x bott/128 output tiles: split-K slabs + one reduce
        S = ops.gemm_splits(K, max(1, min(16, K // 2048)))
        if S == 1:
            ops.gemm_nt(d_logits, self.weffT, d_zn, M=T, N=bott, K=K, epi=EPI_F32)
        else:
            slab = ws.get("b.d_zn_slab", (S * T * bott,), F32)
            ops.gemm_nt(d_logits, self.weffT, slab, M=T, N=bott, K=K, ldc=bott, ldc2=T * bott // 4, epi=ops.EPI_F32_SLAB,
                        splits=S)
            ops.reduce_slabs(slab, T * bott, S, d_zn, T * bott, accumulate=False)
        dz = ws.get("b.dz", (T, bott), F32)
        ops.l2norm_bwd(d_zn, zn, inv_z, dz, T, bott)
        dz_b = ws.get("b.dz_b", (T, bott), BF)
        ops.cast_f32_bf16(dz, dz_b, T * bott)
        dh = ws.get("b.dh", (T, hid), BF)
        # one d(pre-activation) buffer per layer: the weight-gradient branch of l2 (side stream, linear_bwd) may still be
        # reading its dy while the main stream already produces l1's
        dpre2, dpre1 = ws.get("b.dpre2", (T, hid), BF), ws.get("b.dpre1", (T, hid), BF)
        linear_bwd(ws, "l3", self.l3, dz_b, h2, T, dh)
        ops.gelu_bwd(dh, pre2, dpre2, T * hid)
        linear_bwd(ws, "l2", self.l2, dpre2, h1, T, dh)
        ops.gelu_bwd(dh, pre1, dpre1, T * hid)
        dX = ws.get("b.dX", (T, self.Din), BF)
        linear_bwd(ws, "l1", self.l1, dpre1, X, T, dX)
        OVERLAP.join()
        return dX


def build_ssl_indices(masks: np.ndarray, B: int, hw: int, n_local: int, hw_local: int, dino_weight: float,
                      ibot_weight: float, pad_to: int = 64, upperbound: Optional[int] = None):
    """Host-side index plan of one SSL batch (the data pipeline produces `masks` on the host anyway).

    masks: bool [2B, hw] (global crops, view-major).  Token rows refer to the [.., 1 + hw, D] final-norm streams.
    Returns a dict of int32 / f32 numpy arrays:
      student rows  = [local cls (n_local*B) | global cls (2B) | masked patches (Tm, padded)]
      teacher rows  = [global cls with the two views swapped (2B, vtp.py:425-426) | masked patches (Tm)]"""
    N = hw + 1
    n_g = 2
    flat = np.flatnonzero(masks.reshape(-1))
    n_masked = int(flat.size)
    if upperbound is not None:  # the reference's fixed-size masked-token buffers (vtp.py:432-439), rounded up to 64 rows
        if n_masked > upperbound:
            raise ValueError(f"{n_masked} masked patches exceed upperbound={upperbound}")
        pad_to, n_cap = 64, int(upperbound)
    else:
        n_cap = n_masked
    Tm = max(pad_to, (n_cap + pad_to - 1) // pad_to * pad_to)
    img_of = flat // hw
    masked_rows = (img_of * N + 1 + flat % hw).astype(np.int32)
    per_img = np.maximum(masks.sum(1), 1)
    mw = (1.0 / per_img[img_of]).astype(np.float32)
    pad = np.full(Tm - n_masked, -1, np.int32)
    # teacher buffer
    t_cls = np.concatenate([np.arange(B, 2 * B), np.arange(0, B)]).astype(np.int32) * N  # swapped views
    teacher_src = np.concatenate([t_cls, masked_rows, pad])
    # student buffer (two source streams: the local-crop pass and the global-crop pass)
    local_src = (np.arange(n_local * B) * (hw_local + 1)).astype(np.int32)  # cls rows of the local pass
    terms = n_g * (n_g - 1) + n_local * n_g
    w_cls = dino_weight / (B * terms)
    Ts = n_local * B + 2 * B + Tm
    t0 = np.full(Ts, -1, np.int32)
    t1 = np.full(Ts, -1, np.int32)
    w = np.zeros(Ts, np.float32)
    b_idx = np.tile(np.arange(B), n_local)
    t0[:n_local * B] = b_idx            # teacher row b     = the other-view cls of image b (view 1)
    t1[:n_local * B] = B + b_idx        # teacher row B + b = view 0 of image b
    w[:n_local * B] = w_cls
    g0 = n_local * B
    t0[g0:g0 + 2 * B] = np.arange(2 * B)  # swapped buffer: row v*B+b holds the teacher's view (1-v) of image b
    w[g0:g0 + 2 * B] = w_cls
    m0 = g0 + 2 * B
    t0[m0:m0 + n_masked] = 2 * B + np.arange(n_masked)
    w[m0:m0 + n_masked] = ibot_weight * mw / B
    student_global_src = np.concatenate([np.arange(2 * B, dtype=np.int32) * N, masked_rows, pad])
    return dict(n_masked=n_masked, Tm=Tm, Ts=Ts, teacher_src=teacher_src, student_local_src=local_src,
                student_global_src=student_global_src, t0=t0, t1=t1, w=w)
