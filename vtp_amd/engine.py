"""Hand-scheduled forward/backward of the VTP towers on the gfx950 kernels (no autograd inside).

Data layout in HBM (one GPU, per tower):
  * residual stream   f32  [B*N, D]   (token-major; encoder N = 1 + hw with the cls row first, decoder N = hw)
  * GEMM operands     bf16 [rows, K]  K-contiguous; weights are bf16 copies (W and W^T) of the fp32 masters
  * qkv               bf16 [B*N, 3D]  packed exactly like attention.py:115 (q | k | v, head-major inside each);
                                      RoPE is applied in place, attention reads it with strides (no head transposes)
  * saved for backward per block: x_in f32, xn1/xn2 bf16, (mean,rstd) f32, qkv bf16, o bf16, lse f32, x_mid f32,
    x12 bf16 [M,2H] (SwiGLU pre-activations, 8|8 interleaved), hidden bf16 [M,H]  (~28 MB/image-batch-row for VTP-B;
    288 GB of HBM3E makes recomputation pointless at these sizes)
  * master params / grads: two flat f32 buffers (one fused AdamW launch, contiguous gradient buckets for RCCL).

Reference call stack this replaces: SelfAttentionBlock._forward_list (block.py:290-296) ->
SelfAttention.forward (attention.py:91-126) / SwiGLUFFN.forward (ffn.py:77-81), and their autograd backward.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import torch

from . import ops
from .ops import EPI_BF16, EPI_F32, EPI_F32_SLAB, EPI_GELU, EPI_SWIGLU


# =====================================================================================================================
# parameters: flat fp32 masters + flat fp32 grads + bf16 compute copies refreshed by ONE batched kernel
# =====================================================================================================================
class Lin:
    """Compute-side record of one linear map y = x W^T + b  (W fp32 master [N,K]).  w32: the fp32 master as [N,K]; wTs: bf16
    (gamma (.) W)^T, the dgrad operand when a LayerScale follows the linear (refreshed by a prep hook)."""
    __slots__ = ("N", "K", "w", "wT", "bias", "gw", "gb", "w32", "wTs")


class SwiGLULin:
    """w1/w2 of SwiGLUFFN fused into one interleaved [2H, D] matrix (16-row groups = 8 rows w1 | 8 rows w2)."""
    __slots__ = ("H", "K", "w12", "w12T", "b12", "gw1", "gb1")


def _flat_order(names: List[str]) -> List[str]:
    """Registration order, except that (w1.weight, w2.weight) and (w1.bias, w2.bias) are made adjacent (the SwiGLU
    wgrad / bias-grad kernels de-interleave into `w1 | w2` with a single base pointer)."""
    out, seen = [], set()
    for n in names:
        if n in seen:
            continue
        if n.endswith("mlp.w1.weight") or n.endswith("mlp.w1.bias"):
            sib = n.replace("mlp.w1.", "mlp.w2.")
            out += [n, sib]
            seen.update((n, sib))
        else:
            out.append(n)
            seen.add(n)
    return out


class ParamStore:
    def __init__(self, module: torch.nn.Module, device: torch.device):
        self.device = device
        params = dict(module.named_parameters())
        order = _flat_order(list(params.keys()))
        self.offsets: Dict[str, Tuple[int, int]] = {}
        off = 0
        for n in order:
            k = params[n].numel()
            self.offsets[n] = (off, k)
            off += (k + 3) // 4 * 4
        self.numel = off
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=device)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=device)
        self.params = params
        with torch.no_grad():
            for n in order:
                o, k = self.offsets[n]
                prm = params[n]
                view = self.flat_p[o:o + k].view(prm.shape)
                view.copy_(prm.detach().to(device=device, dtype=torch.float32))
                prm.data = view
                prm.grad = self.flat_g[o:o + k].view(prm.shape)
        self._descs: List[List[int]] = []
        self._bf16_elems = 0
        self._bf16_allocs: List[Tuple[object, str, int, tuple]] = []
        self._f32_extra: List[Tuple[object, str, int]] = []
        self._desc_dev = None
        self._tiles = 0
        self._keep = []

    # ---- views
    def p(self, name: str) -> torch.Tensor:
        o, k = self.offsets[name]
        return self.flat_p[o:o + k]

    def g(self, name: str) -> torch.Tensor:
        o, k = self.offsets[name]
        return self.flat_g[o:o + k]

    def has(self, name: str) -> bool:
        return name in self.offsets

    # ---- registration of compute copies (finalize() allocates them in one flat bf16 buffer)
    def _want_bf16(self, obj, attr, n_elems, shape):
        self._bf16_allocs.append((obj, attr, self._bf16_elems, shape))
        self._bf16_elems += (n_elems + 7) // 8 * 8

    def lin(self, wname: str, bname: Optional[str], N: int, K: int, need_T: bool = True) -> Lin:
        L = Lin()
        L.N, L.K = N, K
        assert self.offsets[wname][1] == N * K, (wname, N, K)
        L.bias = self.p(bname) if bname and self.has(bname) else None
        L.gw = self.g(wname)
        L.gb = self.g(bname) if bname and self.has(bname) else None
        L.wT = None
        L.w32, L.wTs = self.p(wname).view(N, K), None
        self._want_bf16(L, "w", N * K, (N, K))
        if need_T:
            self._want_bf16(L, "wT", N * K, (K, N))
        self._descs.append(["lin", L, wname, need_T])
        return L

    def swiglu(self, prefix: str, H: int, K: int) -> SwiGLULin:
        S = SwiGLULin()
        S.H, S.K = H, K
        o1, k1 = self.offsets[prefix + "w1.weight"]
        o2, _ = self.offsets[prefix + "w2.weight"]
        assert o2 == o1 + k1 and k1 == H * K
        ob1, kb1 = self.offsets[prefix + "w1.bias"]
        ob2, _ = self.offsets[prefix + "w2.bias"]
        assert ob2 == ob1 + kb1 and kb1 == H
        S.gw1 = self.flat_g[o1:o1 + 2 * k1]
        S.gb1 = self.flat_g[ob1:ob1 + 2 * H]
        self._want_bf16(S, "w12", 2 * H * K, (2 * H, K))
        self._want_bf16(S, "w12T", 2 * H * K, (K, 2 * H))
        self._descs.append(["swiglu", S, prefix])
        return S

    def finalize(self):
        self.flat_bf16 = torch.zeros(max(self._bf16_elems, 8), dtype=torch.bfloat16, device=self.device)
        for obj, attr, off, shape in self._bf16_allocs:
            n = shape[0] * shape[1]
            setattr(obj, attr, self.flat_bf16[off:off + n].view(shape))
        rows, tiles = [], 0
        self._desc_src = []  # flat-buffer offset of each record's (first) fp32 source: which gradient bucket refreshes it (desc_runs)

        def add(src, src2, dst, dstT, R, C, mode, name):
            nonlocal tiles
            rows.append([src, src2, dst, dstT, R, C, mode, tiles])
            self._desc_src.append(self.offsets[name][0])
            tiles += (R + 255) // 256 if mode == 2 else ((R + 63) // 64) * ((C + 63) // 64)

        for d in self._descs:
            if d[0] == "lin":
                _, L, wname, need_T = d
                add(self.p(wname).data_ptr(), 0, L.w.data_ptr(), L.wT.data_ptr() if need_T else 0, L.N, L.K, 0, wname)
            else:
                _, S, prefix = d
                b12 = torch.zeros(2 * S.H, dtype=torch.float32, device=self.device)
                S.b12 = b12
                self._keep.append(b12)
                add(self.p(prefix + "w1.weight").data_ptr(), self.p(prefix + "w2.weight").data_ptr(), S.w12.data_ptr(),
                    S.w12T.data_ptr(), 2 * S.H, S.K, 1, prefix + "w1.weight")
                add(self.p(prefix + "w1.bias").data_ptr(), self.p(prefix + "w2.bias").data_ptr(), b12.data_ptr(), 0,
                    2 * S.H, 1, 2, prefix + "w1.bias")
        self._desc_dev = torch.tensor(rows, dtype=torch.int64, device=self.device)
        self._ndesc = len(rows)
        self._tiles = tiles
        self._desc_tile0 = [r[7] for r in rows] + [tiles]
        self.prep()

    def prep(self):
        """Refresh every bf16 compute copy from the fp32 masters (one kernel launch)."""
        ops.prep_weights(self._desc_dev, self._ndesc, self._tiles)
        for hook in getattr(self, "prep_hooks", ()):
            hook()
        self._prepped_version = self.flat_p._version

    def desc_runs(self, ranges):
        """runs [d0, d1) of the weight-refresh table whose fp32 sources lie in the given flat ranges (the layers of a gradient bucket)"""
        hit = [i for i, o in enumerate(self._desc_src) if any(lo <= o < hi for lo, hi in ranges)]
        runs = []
        for i in hit:
            if runs and runs[-1][1] == i:
                runs[-1][1] = i + 1
            else:
                runs.append([i, i + 1])
        return [tuple(r) for r in runs]

    def prep_runs(self, runs):
        """refresh the bf16 compute copies of the table runs only (one launch per run): the per-bucket optimizer lane of the trainer"""
        for d0, d1 in runs:
            ops.prep_weights_range(self._desc_dev[d0:], d1 - d0, self._desc_tile0[d0], self._desc_tile0[d1] - self._desc_tile0[d0])

    def mark_prepped(self, done=()):
        """every compute copy is current (the trainer refreshed them bucket by bucket): the derived-weight hooks that did not ride
        along with a bucket (`done`) + the version stamp"""
        for hook in getattr(self, "prep_hooks", ()):
            if hook not in done:
                hook()
        self._prepped_version = self.flat_p._version

    def ensure_fresh(self):
        if self.flat_p._version != self._prepped_version:
            self.prep()

    def zero_grad(self):
        self.flat_g.zero_()

    def sync_grad_views(self):
        """Every parameter's .grad must be its slice of the flat gradient buffer (the engines accumulate there).  A torch
        optimizer's zero_grad(set_to_none=True) drops the views: re-attach them, zeroed ("None" means a fresh gradient)."""
        for n, prm in self.params.items():
            if prm.grad is None and prm.requires_grad:
                o, k = self.offsets[n]
                view = self.flat_g[o:o + k].view(prm.shape)
                view.zero_()
                prm.grad = view


# =====================================================================================================================
# buffers
# =====================================================================================================================
class Workspace:
    """Named device buffers, allocated once per (name, shape, dtype) and reused every step (static memory plan)."""

    def __init__(self, device):
        self.device = device
        self._bufs: Dict[tuple, torch.Tensor] = {}

    def get(self, name: str, shape, dtype, zero: bool = False) -> torch.Tensor:
        key = (name, tuple(shape), dtype)
        t = self._bufs.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
            self._bufs[key] = t
        return t

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self._bufs.values())


BF, F32 = torch.bfloat16, torch.float32


def _env_flag(name: str, default: str = "1") -> bool:
    return os.environ.get(name, default) not in ("0", "false", "off")


class Overlap:
    """Side HIP streams for work that is independent of the main stream's next kernels and fills the CUs their tails leave
    idle: the weight-gradient branch of every linear backward (column sums + split-K TN GEMM + slab reduce run concurrently
    with the dgrad GEMM), the EMA teacher's forward, and -- on lane 1 -- the text tower, which the trainer issues on its own
    stream with its own wgrad side stream.  fork/join are event waits (parallel hipGraph branches under stream capture)."""

    enabled = _env_flag("VTP_OVERLAP")  # VTP_OVERLAP=0: everything in-line on one stream (clean per-kernel profiles)

    def __init__(self):
        self._sides = {}
        self._lane = 0
        self._deferred = []

    @property
    def side(self):
        return self._sides.get(self._lane)

    def _side(self):
        if self._lane not in self._sides:
            self._sides[self._lane] = torch.cuda.Stream()
        return self._sides[self._lane]

    def fork(self):
        self._side().wait_stream(torch.cuda.current_stream())

    def join(self):
        self.run_deferred()
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)

    # Launch order inside a captured segment decides which hardware queue a kernel replays on: the HIP graph executor walks the
    # segment's root nodes in capture order, hands each root the next queue and lets a node inherit the queue of the first
    # predecessor that reaches it.  A side branch captured in FRONT of the main stream's first kernel therefore takes the main
    # chain's join nodes onto its own queue, and the main chain pays a cross-queue signal (10-20 us of idle chip) on the way in and
    # out of every such node.  Side work is therefore forked where its inputs are ready but ISSUED behind the main stream's next
    # kernel: defer(fn) queues the issue, run_deferred() -- called by Stack.backward behind the first dgrad GEMM of a block, and by
    # whoever joins -- performs it.  (VTP_FORK_LATE=0: issue at the fork, the order of rounds 2-5.)
    def defer(self, fn):
        if FORK_LATE:
            self._deferred.append(fn)
        else:
            fn()

    def run_deferred(self):
        while self._deferred:
            self._deferred.pop(0)()

    def lane(self, k: int):
        """context: fork / join / side refer to side stream k while a second tower is being issued on another stream"""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            prev, self._lane = self._lane, k
            try:
                yield
            finally:
                self._lane = prev
        return ctx()


FORK_LATE = _env_flag("VTP_FORK_LATE")
# VTP_PE_IN_GROUP=0: the patch-embed weight gradient as its own split-K launches (one set per list item) behind block 0's group
PE_IN_GROUP = _env_flag("VTP_PE_IN_GROUP")
OVERLAP = Overlap()
# Decided by same-box A/B runs of rounds 1-3 and no longer switchable: apply_rope in the qkv GEMM's epilogue, the SwiGLU backward
# in the w3 dgrad's epilogue, weight gradients as TN GEMMs straight from the activation layouts (no transposed copies, no fp32
# atomics from the MFMA epilogue: 285 vs 572 images/s), bias-gradient column sums beside (not inside) the per-layer wgrad GEMMs.
FUSE_SWIGLU_BWD = True
FUSE_ROPE = True
# the four weight gradients of a transformer block as ONE grouped launch with the split-K combine and the bias-gradient column
# sums inside it (ops.WgradGroup), issued one block late so that it runs beside the NEXT block's dgrad / attention kernels
# (the per-layer path -- one split-K GEMM + slab reduce + column-sum launch per linear layer, beside that layer's dgrad -- is what
# LayerScale, stochastic depth and tiny token counts use)
WGRAD_GROUPED = True
# experiment switch (round 6): VTP_WGRAD_INLINE=1 issues the grouped launch on the main stream (no side stream) -- what the overlap of the
# dominant kernel with the next block's dgrad chain is worth in the step (0.4 %: a kernel that holds every CU it runs on hides little)
WGRAD_INLINE = _env_flag("VTP_WGRAD_INLINE", default="0")


def linear_bwd(ws: Workspace, tag: str, L, dy_b, x_b, M: int, d_in, *, need_dx: bool = True, dy_remap=(0, 0),
               x_remap=(0, 0), dx_remap=(0, 0), gw=None, gb=None, N=None, K=None, wT=None, swiglu_h: int = 0,
               bias_grad_done: bool = False, ls=None, dgrad_swiglu=None, defer=None):
    """Backward of y[M,N] = x[M,K] W^T + b given dy (bf16 [M,N]):  dW += dy^T x,  db += colsum(dy),  dx = dy W.
    Reaches the NT GEMM through transposed operands: dy^T and x^T are produced by the LDS transpose kernel (the
    column sums for db ride along), W^T is the cached transposed weight.  The wgrad GEMM is split-K over the token
    dimension; slices write private fp32 slabs (plain stores) that one reduce kernel folds into the flat gradient."""
    if L is not None:
        N = L.N if N is None else N
        K = L.K if K is None else K
        gw = L.gw if gw is None else gw
        gb = L.gb if gb is None else gb
        wT = L.wT if wT is None else wT
    if ls is not None:
        # y = gamma (.) (x W^T + b) (LayerScale, misc.py:24-25) and dy_b is the gradient of y: the branch output f is not stored
        # -- G = dy^T x and cs = colsum(dy) give dW = gamma (.) G, db = gamma (.) cs, dgamma = rowsum(W (.) G) + b (.) cs
        gamma, g_gamma = ls
        assert L is not None and not swiglu_h and dy_remap == (0, 0) and x_remap == (0, 0)
        cs = ws.get("T.ls_cs", (N,), F32)
        cs.zero_()
        G = ws.get("T.ls_G", (N * K,), F32)

        def wgrad_ls():
            kw = dict(M=N, N=K, K=M, lda=dy_b.stride(0), ldb=x_b.stride(0), ldc=K, a_colsum=cs)
            St = ops.gemm_tn_splits(N, K, M)
            if St == 1:
                ops.gemm_tn(dy_b, x_b, G, epi=EPI_F32, **kw)
            else:
                slab = ws.get("T.slab", (St * N * K,), F32)
                ops.gemm_tn(dy_b, x_b, slab, ldc2=N * K // 4, epi=EPI_F32_SLAB, splits=St, **kw)
                ops.reduce_slabs(slab, N * K, St, G, N * K, accumulate=False)
            ops.layerscale_wgrad(G, L.w32, L.bias, cs, gamma, gw, gb, g_gamma, N, K)

        OVERLAP.join()
        wgrad_ls()
        if need_dx:
            ops.gemm_nt(dy_b, L.wTs, d_in, M=M, N=K, K=N, lda=dy_b.stride(0), ldb=N, ldc=d_in.stride(0), epi=EPI_BF16,
                        c_remap=dx_remap)
        return
    if bias_grad_done:  # the kernel that produced dy_b already accumulated its column sums into the bias gradient
        gb = None
    if defer is not None:  # grouped weight gradients: record the problem, the caller launches the block's group later
        assert dy_remap == (0, 0) and x_remap == (0, 0)
        defer.append(dict(dy=dy_b, x=x_b, gw=gw, gb=gb, N=N, K=K, swiglu_h=swiglu_h))
    c_remap = (-1, swiglu_h) if swiglu_h else (0, 0)

    def wgrad():
        # dW[N,K] = dy[M,N]^T x[M,K] straight from the activation layouts (LDS transpose reads inside the GEMM); db = colsum(dy) as a
        # column-sum launch beside it
        if gb is not None:
            ops.colsum_bf16(dy_b, dy_b.stride(0), gb, M, N, swiglu_h=swiglu_h, in_remap=dy_remap)
        kw = dict(M=N, N=K, K=M, lda=dy_b.stride(0), ldb=x_b.stride(0), ldc=K, a_remap=dy_remap, b_remap=x_remap, c_remap=c_remap)
        St = ops.gemm_tn_splits(N, K, M)  # tile-configuration aware (8-phase 256x256 kernel: tiles x splits = 256 CUs)
        if St == 1:
            ops.gemm_tn(dy_b, x_b, gw, resid=gw, epi=EPI_F32, **kw)
        else:
            n_el = N * K
            slab = ws.get("T.slab", (St * n_el,), F32)
            ops.gemm_tn(dy_b, x_b, slab, ldc2=n_el // 4, epi=EPI_F32_SLAB, splits=St, **kw)
            ops.reduce_slabs(slab, n_el, St, gw, n_el, accumulate=True)

    if defer is not None:
        pass
    elif OVERLAP.enabled and need_dx:
        OVERLAP.join()   # previous layer's dW branch is done with the shared T.* scratch (and with its inputs)
        OVERLAP.fork()   # the side stream sees dy_b / x_b complete
        with torch.cuda.stream(OVERLAP.side):
            wgrad()
    else:
        OVERLAP.join()
        wgrad()
    if need_dx and dgrad_swiglu is not None:
        # this linear is the w3 of a SwiGLU FFN: the dgrad GEMM's epilogue applies the activation backward with the saved x12 and
        # writes dx12 directly (d_in = the [M, 2K] pre-activation gradient); dh never exists in HBM
        assert dy_remap == (0, 0) and dx_remap == (0, 0)
        ops.gemm_dgrad_swiglu(dy_b, wT, dgrad_swiglu, d_in, M, K, N)
    elif need_dx:
        ops.gemm_nt(dy_b, wT, d_in, M=M, N=K, K=N, lda=dy_b.stride(0), ldb=N, ldc=d_in.stride(0), epi=EPI_BF16,
                    a_remap=dy_remap, c_remap=dx_remap)


# =====================================================================================================================
# a run of SelfAttentionBlocks (block.py:137-308) -- shared by the ViT trunk and the pixel decoder
# =====================================================================================================================
class BlockW:
    pass


# parameter names per block style (reference module trees: block.py:159-187 SelfAttentionBlock, block.py:382-399
# ResidualAttentionBlock + nn.MultiheadAttention)
_NAMES = {
    "vit": dict(n1="norm1", n2="norm2", qkv_w="attn.qkv.weight", qkv_b="attn.qkv.bias", proj_w="attn.proj.weight",
                proj_b="attn.proj.bias", ls1="ls1.gamma", ls2="ls2.gamma"),
    "text": dict(n1="ln_1", n2="ln_2", qkv_w="attn.in_proj_weight", qkv_b="attn.in_proj_bias",
                 proj_w="attn.out_proj.weight", proj_b="attn.out_proj.bias", ls1="ls_1.gamma", ls2="ls_2.gamma"),
}


class Stack:
    """A run of pre-norm transformer blocks.  style "vit": SelfAttentionBlock (RoPE attention + SwiGLU FFN);
    style "text": CLIP ResidualAttentionBlock (causal attention, no RoPE, LayerNorm eps 1e-5, erf-GELU MLP of width H)."""

    def __init__(self, store: ParamStore, prefix: str, depth: int, D: int, heads: int, H: int, norm: str,
                 style: str = "vit", ffn: str = "swiglu"):
        self.store, self.depth, self.D, self.heads, self.H, self.style = store, depth, D, heads, H, style
        # FFN of the blocks: SwiGLUFFN (w1 | w2 fused + w3) or the erf-GELU MLP -- `mlp.fc1 / fc2` in ViT blocks with ffn_layer = "mlp"
        # (ffn.py:21-48), `mlp.c_fc / c_proj` in the text tower
        self.swiglu = style == "vit" and ffn != "mlp"
        self.quick_gelu = False  # GELU MLPs: QuickGELU instead of erf-GELU (text_quick_gelu; set by the owner)
        self.kind = ops.NORM_RMS if norm == "rmsnorm" else ops.NORM_LN
        if style == "text":
            self.eps = 1e-5  # nn.LayerNorm default (normalization.py:25-31)
        else:
            self.eps = 1e-5 if norm == "rmsnorm" else 1e-6  # RMSNorm default / partial(nn.LayerNorm, eps=1e-6)
        self.causal = style == "text"
        nm = _NAMES[style]
        self.blocks: List[BlockW] = []
        for i in range(depth):
            b = BlockW()
            pre = f"{prefix}{i}."
            b.n1w, b.gn1w = store.p(pre + nm["n1"] + ".weight"), store.g(pre + nm["n1"] + ".weight")
            b.n2w, b.gn2w = store.p(pre + nm["n2"] + ".weight"), store.g(pre + nm["n2"] + ".weight")
            if self.kind == ops.NORM_LN:
                b.n1b, b.gn1b = store.p(pre + nm["n1"] + ".bias"), store.g(pre + nm["n1"] + ".bias")
                b.n2b, b.gn2b = store.p(pre + nm["n2"] + ".bias"), store.g(pre + nm["n2"] + ".bias")
            else:
                b.n1b = b.gn1b = b.n2b = b.gn2b = None
            b.qkv = store.lin(pre + nm["qkv_w"], pre + nm["qkv_b"], 3 * D, D)
            b.proj = store.lin(pre + nm["proj_w"], pre + nm["proj_b"], D, D)
            if self.swiglu:
                b.w12 = store.swiglu(pre + "mlp.", H, D)
                b.w3 = store.lin(pre + "mlp.w3.weight", pre + "mlp.w3.bias", D, H)
            else:
                n1, n2 = ("fc1", "fc2") if style == "vit" else ("c_fc", "c_proj")
                b.fc = store.lin(f"{pre}mlp.{n1}.weight", f"{pre}mlp.{n1}.bias", H, D)
                b.w3 = store.lin(f"{pre}mlp.{n2}.weight", f"{pre}mlp.{n2}.bias", D, H)
            b.ls1 = store.p(pre + nm["ls1"]) if store.has(pre + nm["ls1"]) else None  # LayerScale (misc.py:7-26; text: block.py:388,399)
            b.ls2 = store.p(pre + nm["ls2"]) if store.has(pre + nm["ls2"]) else None
            b.gls1 = store.g(pre + nm["ls1"]) if b.ls1 is not None else None
            b.gls2 = store.g(pre + nm["ls2"]) if b.ls2 is not None else None
            # QK normalisation (attention.py:67-68): RMSNorm(head_dim) weights of q and k, shared by the heads
            qn = pre + "attn.q_norm.weight"
            if store.has(qn):
                b.qn_w, b.kn_w = store.p(qn), store.p(pre + "attn.k_norm.weight")
                b.g_qn, b.g_kn = store.g(qn), store.g(pre + "attn.k_norm.weight")
            else:
                b.qn_w = b.kn_w = b.g_qn = b.g_kn = None
            for L, gam in ((b.proj, b.ls1), (b.w3, b.ls2)):
                if gam is not None:  # dgrad operand (gamma (.) W)^T, refreshed with the other bf16 copies
                    L.wTs = torch.empty(L.K, L.N, dtype=BF, device=store.device)
                    if not hasattr(store, "prep_hooks"):
                        store.prep_hooks = []
                    store.prep_hooks.append(lambda L=L, gam=gam: ops.scaled_transpose(L.w32, gam, L.wTs, L.N, L.K))
            self.blocks.append(b)
        self.qk_norm = any(b.qn_w is not None for b in self.blocks)
        self.drop_plan = None  # stochastic depth (set per step by set_drop_plan)
        # grouped weight-gradient launches write dW = ... instead of dW += ... while this is set: valid only where the caller has zeroed the
        # gradients and runs this stack's backward ONCE before it reads them (VTPTrainer's step sets it; the autograd boundary, which
        # must accumulate like nn.Parameter.grad does, never does).  Saves the 256-KiB read of every output tile in the epilogue of the
        # step's dominant kernel: 38 -> 24 us per last-arriving workgroup (tools/wgrad_timeline.py)
        self.wgrad_overwrite = False

    def _rope_plan(self, ws: Workspace, segs, prefix_tokens: int, M: int):
        """(rope_pos int32 [M], sin, cos) for the fused qkv + RoPE epilogue: rope_pos[m] = row of the concatenated per-segment
        tables that rotates token row m, -1 for prefix (cls) rows.  Built once per workspace (static segment structure).  Under
        train-time RoPE augmentation the segments carry RopeAugTabs: sin / cos are then the static per-block buffers [depth, P, 64]
        themselves (refreshed in place every step; block i uses [i]), not a concatenated copy."""
        if self.style != "vit" or all(rp is None for _, _, rp in segs) or (2 * self.D) % 128 or self.qk_norm:
            return None  # (with QK normalisation the rotation follows the norm: it cannot ride in the projection's epilogue)
        hit = getattr(ws, "_rope_plan", None)
        aug = [rp for _, _, rp in segs if isinstance(rp, RopeAugTabs)]
        key = tuple((b, n, None if rp is None else (rp.sin_all.data_ptr(), rp.off) if isinstance(rp, RopeAugTabs) else rp[0].data_ptr())
                    for b, n, rp in segs) + (prefix_tokens,)
        if hit is not None and hit[0] == key:
            return hit[1]
        pos, tabs_s, tabs_c, base = [], [], [], 0
        for Bs, Ns, rp in segs:
            if rp is None:
                pos.append(torch.full((Bs * Ns,), -1, dtype=torch.int32))
                continue
            hw = _rope_hw(rp)
            assert Ns - prefix_tokens == hw, (Ns, prefix_tokens, hw)
            if aug:  # rows of the shared per-block buffers: the item's offset there
                assert isinstance(rp, RopeAugTabs) and rp.sin_all is aug[0].sin_all, "augmented and plain RoPE tables cannot be mixed in one pass"
                base = rp.off
            one = torch.cat([torch.full((prefix_tokens,), -1, dtype=torch.int32), torch.arange(hw, dtype=torch.int32) + base])
            pos.append(one.repeat(Bs))
            if not aug:
                tabs_s.append(rp[0])
                tabs_c.append(rp[1])
                base += hw
        dev = self.store.device
        if aug:
            plan = (torch.cat(pos).to(dev), aug[0].sin_all, aug[0].cos_all)
        else:
            plan = (torch.cat(pos).to(dev), torch.cat(tabs_s).contiguous(), torch.cat(tabs_c).contiguous())
        assert plan[0].numel() == M
        ws._rope_plan = (key, plan)
        return plan

    @staticmethod
    def _plan_tabs(plan, i: int):
        """sin / cos operands of the fused epilogue for block i"""
        if plan[1].ndim == 3:  # static per-block buffers of RopeAugTabs (refreshed in place); [1, P, 64]: one draw for every block
            i = min(i, plan[1].shape[0] - 1)
            return plan[1][i], plan[2][i]
        return plan[1], plan[2]

    # ------------------------------------------------------------------------------------------------ stochastic depth
    # block.py:20-118 (get_branges_scales) and :207-289: in training with drop_ratio > 0 every residual branch of every block runs
    # on a fresh random subset of the images of each list item and is added back with alpha = batch / kept.
    @staticmethod
    def drop_allocation(b: int, ratio: float, world: int = 1, rank: int = 0):
        """(images kept on this rank, residual scale factor) -- the reference's allocation rule: without DDP keep =
        max(int(b (1 - r)), 1), scale = b / keep; with DDP the GLOBAL keep count max(int(b W (1 - r)), W) is spread evenly over the
        ranks and scale = global batch / global kept (block.py:44-62; deterministic, so no broadcast is needed here)."""
        if world <= 1:
            keep = max(int(b * (1 - ratio)), 1)
            return keep, b / keep
        gb = b * world
        gkeep = max(int(gb * (1 - ratio)), world)
        base, extra = gkeep // world, gkeep % world
        alloc = [min(base + (1 if i < extra else 0), b) for i in range(world)]
        return alloc[rank], gb / max(sum(alloc), 1)

    def make_drop_plan(self, segs, ratio, generator: Optional[torch.Generator] = None, world: int = 1, rank: int = 0):
        """Host side, once per step: a random image subset per (block, branch, list item).  `ratio`: one rate, or one per list item
        (the reference has a rate per objective -- clip_drop_rate / ssl_drop_rate / rec_drop_rate, vtp.py:205-207 -- and the
        objectives are items of one list forward here).  Returns a dict with the int32 index tensor (CPU; copy it into the static
        device buffer with set_drop_plan) and the static shape information."""
        ratios = list(ratio) if isinstance(ratio, (list, tuple)) else [ratio] * len(segs)
        assert len(ratios) == len(segs)
        keeps, scales = [], []
        for (B, _, _), r in zip(segs, ratios):
            k, sc = self.drop_allocation(B, r, world, rank)
            keeps.append(k)
            scales.append(sc)
        per = sum(keeps)
        idx = torch.empty(self.depth * 2 * per, dtype=torch.int32)
        o = 0
        for _ in range(self.depth * 2):
            for (B, _, _), k in zip(segs, keeps):
                idx[o:o + k] = torch.randperm(B, generator=generator)[:k].to(torch.int32)
                o += k
        return dict(idx=idx, keeps=keeps, scales=scales, per=per, ratio=ratio, batches=[B for B, _, _ in segs])

    def set_drop_plan(self, plan, ws_key="drop"):
        """activate / update (plan) or switch off (None) stochastic depth for the following forward + backward.  The device index
        buffer of a given size is allocated ONCE and kept for the lifetime of the stack: captured hipGraph segments bake its
        address, so it is only ever refreshed in place -- switching the plan off, or alternating between plans of different sizes
        (steps with / without SSL crops), never frees or moves a buffer a graph may still read."""
        if plan is not None and self.style != "vit":
            raise NotImplementedError("stochastic depth is a property of the ViT blocks (block.py:207-289); the text tower has none")
        if plan is None:
            if self.drop_plan is not None:
                self.last_drop_plan = self.drop_plan  # introspection (tests replay the subsets of the step that just ran)
            self.drop_plan = None
            return
        bufs = self.__dict__.setdefault("_drop_bufs", {})
        n = plan["idx"].numel()
        buf = bufs.get(n)
        if buf is None:
            buf = bufs[n] = torch.empty(n, dtype=torch.int32, device=self.store.device)
        buf.copy_(plan["idx"], non_blocking=True)
        cur = dict(plan)
        cur["idx_dev"] = buf
        self.drop_plan = cur

    @staticmethod
    def _check_drop_plan(p, segs):
        """the plan was drawn for specific per-item batch sizes: a pass with other shapes must not run on it (out-of-range image
        indices / silently truncated item lists)"""
        got = [B for B, _, _ in segs]
        if p.get("batches") != got:
            raise RuntimeError(f"stochastic-depth plan drawn for item batch sizes {p.get('batches')} used on a pass with {got}")

    def _drop_idx(self, i: int, branch: int, s: int):
        p = self.drop_plan
        o = (i * 2 + branch) * p["per"] + sum(p["keeps"][:s])
        return p["idx_dev"][o:o + p["keeps"][s]]

    def forward_drop(self, ws: Workspace, x, segs, prefix_tokens: int):
        """training forward with stochastic depth; `x` (f32 [M, D]) is updated IN PLACE block after block (the branch inputs that
        backward needs are the gathered compact copies).  segs = [(B_i, N_i, rope_i)]."""
        D, H, heads = self.D, self.H, self.heads
        p = self.drop_plan
        keeps, scales = p["keeps"], p["scales"]
        self._check_drop_plan(p, segs)
        csegs = [(k, N, rp) for (B, N, rp), k in zip(segs, keeps)]  # the compact (gathered) list
        Mc = sum(k * N for k, N, _ in csegs)
        scale = 1.0 / math.sqrt(64.0)
        rope_plan = self._rope_plan(ws, csegs, prefix_tokens, Mc) if FUSE_ROPE else None
        saved_all = []
        for i, b in enumerate(self.blocks):
            t = f"{i}.d."
            xs1, xs2 = ws.get(t + "xs1", (Mc, D), F32), ws.get(t + "xs2", (Mc, D), F32)
            xn1, st1 = ws.get(t + "xn1", (Mc, D), BF), ws.get(t + "st1", (Mc, 2), F32)
            qkv, o = ws.get(t + "qkv", (Mc, 3 * D), BF), ws.get(t + "o", (Mc, D), BF)
            lse = ws.get(t + "lse", (Mc * heads,), F32)
            xn2, st2 = ws.get(t + "xn2", (Mc, D), BF), ws.get(t + "st2", (Mc, 2), F32)
            pre, hid = ws.get(t + "x12", (Mc, 2 * H if self.swiglu else H), BF), ws.get(t + "hid", (Mc, H), BF)
            delta = ws.get("d.delta", (Mc, D), F32)
            # ---- attention branch on the kept images
            self._gather(x, segs, csegs, i, 0, xs1, None, 1.0)
            ops.norm_fwd(xs1, b.n1w, b.n1b, xn1, st1, Mc, D, self.eps, self.kind)
            if rope_plan is not None:
                ps, pc = self._plan_tabs(rope_plan, i)
                ops.gemm_qkv_rope(xn1, b.qkv.w, b.qkv.bias, qkv, Mc, 3 * D, D, rope_plan[0], ps, pc, 2 * D)
            elif b.qn_w is not None:  # QK normalisation (attention.py:67-68,119-120) on the kept images: projection -> norm -> RoPE below
                qkv_pre = ws.get(t + "qkv_pre", (Mc, 3 * D), BF)
                qinv = ws.get(t + "qinv", (Mc, 2 * heads), F32)
                ops.gemm_nt(xn1, b.qkv.w, qkv_pre, M=Mc, N=3 * D, K=D, bias=b.qkv.bias, epi=EPI_BF16)
                ops.qk_norm_fwd(qkv_pre, b.qn_w, b.kn_w, qkv, qinv, Mc, D)
            else:
                ops.gemm_nt(xn1, b.qkv.w, qkv, M=Mc, N=3 * D, K=D, bias=b.qkv.bias, epi=EPI_BF16)
            for r0, Bs, Ns, rp in self._rows(csegs):
                q_s = qkv[r0:r0 + Bs * Ns]
                if rp is not None and rope_plan is None:
                    rs, rc = rope_at(rp, i)
                    ops.rope_qk(q_s, rs, rc, Bs, Ns, heads, prefix_tokens)
                ops.attn_fwd(q_s, q_s[:, D:], q_s[:, 2 * D:], o[r0:r0 + Bs * Ns], lse[r0 * heads:], Bs, Ns, heads, Ns * 3 * D, 3 * D,
                             Ns * D, D, scale, self.causal)
            ops.gemm_nt(o, b.proj.w, delta, M=Mc, N=D, K=D, bias=b.proj.bias, gamma=b.ls1, epi=EPI_F32)
            self._scatter(delta, x, segs, csegs, i, 0, scales, accumulate=True)  # x_attn = index_add(x, residual, alpha)
            # ---- FFN branch on an independent subset
            self._gather(x, segs, csegs, i, 1, xs2, None, 1.0)
            ops.norm_fwd(xs2, b.n2w, b.n2b, xn2, st2, Mc, D, self.eps, self.kind)
            if self.swiglu:
                ops.gemm_nt(xn2, b.w12.w12, hid, M=Mc, N=2 * H, K=D, c2=pre, ldc2=2 * H, bias=b.w12.b12, epi=EPI_SWIGLU)
            else:  # Mlp FFN (ffn_layer = "mlp", ffn.py:21-48): fc1 -> GELU -> fc2 on the kept images
                ops.gemm_nt(xn2, b.fc.w, hid, M=Mc, N=H, K=D, c2=pre, ldc2=H, bias=b.fc.bias,
                            epi=ops.EPI_QUICK_GELU if self.quick_gelu else EPI_GELU)
            ops.gemm_nt(hid, b.w3.w, delta, M=Mc, N=D, K=H, bias=b.w3.bias, gamma=b.ls2, epi=EPI_F32)
            self._scatter(delta, x, segs, csegs, i, 1, scales, accumulate=True)
            saved_all.append((xs1, xn1, st1, qkv, o, lse, xs2, xn2, st2, pre, hid))
        self.last_saved = saved_all
        return x

    def _gather(self, x, segs, csegs, i, branch, dst, dst_b, scale_by_seg):
        D = self.D
        r_full, r_c = 0, 0
        for s, ((B, N, _), (k, _, _)) in enumerate(zip(segs, csegs)):
            sc = scale_by_seg[s] if isinstance(scale_by_seg, (list, tuple)) else scale_by_seg
            ops.gather_image_rows(x[r_full:], self._drop_idx(i, branch, s), None if dst is None else dst[r_c:],
                                  None if dst_b is None else dst_b[r_c:], k, N, D, sc)
            r_full += B * N
            r_c += k * N

    def _scatter(self, src, x, segs, csegs, i, branch, alpha_by_seg, accumulate):
        D = self.D
        r_full, r_c = 0, 0
        for s, ((B, N, _), (k, _, _)) in enumerate(zip(segs, csegs)):
            al = alpha_by_seg[s] if isinstance(alpha_by_seg, (list, tuple)) else alpha_by_seg
            ops.scatter_image_rows(src[r_c:], self._drop_idx(i, branch, s), x[r_full:], k, N, D, al, accumulate)
            r_full += B * N
            r_c += k * N

    def backward_drop(self, ws: Workspace, dy, segs, prefix_tokens: int, saved):
        """backward of forward_drop; `dy` (f32 [M, D], gradient of the stack output) is updated IN PLACE into the gradient of the
        stack input.  Generator (see backward)."""
        D, H, heads = self.D, self.H, self.heads
        p = self.drop_plan
        keeps, scales = p["keeps"], p["scales"]
        self._check_drop_plan(p, segs)
        csegs = [(k, N, rp) for (B, N, rp), k in zip(segs, keeps)]
        Mc = sum(k * N for k, N, _ in csegs)
        scale = 1.0 / math.sqrt(64.0)
        dh = ws.get("b.d.dh", (Mc, H), BF)
        dpre = ws.get("b.d.dx12", (Mc, 2 * H if self.swiglu else H), BF)
        dxn = ws.get("b.d.dxn", (Mc, D), BF)
        d_o = ws.get("b.d.do", (Mc, D), BF)
        dqkv = ws.get("b.d.dqkv", (Mc, 3 * D), BF)
        delta = ws.get("b.d.delta", (Mc * heads,), F32)
        gdy_b = ws.get("b.d.gdy_b", (Mc, D), BF)
        gdy = ws.get("b.d.gdy", (Mc, D), F32)
        dxc = ws.get("b.d.dxc", (Mc, D), F32)
        for i in range(self.depth - 1, -1, -1):
            b = self.blocks[i]
            xs1, xn1, st1, qkv, o, lse, xs2, xn2, st2, pre, hid = saved[i]
            # ---- FFN branch: d(delta2) = alpha * dy[kept rows]; the branch-input gradient lands on the same rows
            self._gather(dy, segs, csegs, i, 1, gdy, gdy_b, scales)
            linear_bwd(ws, "w3", b.w3, gdy_b, hid, Mc, dh, ls=(b.ls2, b.gls2) if b.ls2 is not None else None)
            if self.swiglu:
                ops.swiglu_bwd(dh, pre, dpre, Mc, H)
                linear_bwd(ws, "w12", None, dpre, xn2, Mc, dxn, N=2 * H, K=D, gw=b.w12.gw1, gb=b.w12.gb1, wT=b.w12.w12T, swiglu_h=H)
            else:
                ops.gelu_bwd(dh, pre, dpre, Mc * H, quick=self.quick_gelu)
                linear_bwd(ws, "fc", b.fc, dpre, xn2, Mc, dxn)
            ops.norm_bwd(dxn, xs2, b.n2w, st2, gdy, dxc, None, b.gn2w, b.gn2b, Mc, D, self.kind)   # dxc = dy[kept] + norm bwd
            self._scatter(dxc, dy, segs, csegs, i, 1, 1.0, accumulate=False)
            # ---- attention branch
            self._gather(dy, segs, csegs, i, 0, gdy, gdy_b, scales)
            linear_bwd(ws, "proj", b.proj, gdy_b, o, Mc, d_o, ls=(b.ls1, b.gls1) if b.ls1 is not None else None)
            for r0, Bs, Ns, rp in self._rows(csegs):
                r1 = r0 + Bs * Ns
                q_s, dq_s = qkv[r0:r1], dqkv[r0:r1]
                ops.attn_bwd(q_s, q_s[:, D:], q_s[:, 2 * D:], o[r0:r1], d_o[r0:r1], lse[r0 * heads:], delta[r0 * heads:], dq_s,
                             dq_s[:, D:], dq_s[:, 2 * D:], Bs, Ns, heads, Ns * 3 * D, 3 * D, Ns * D, D, scale, self.causal,
                             rope=None if rp is None else rope_at(rp, i), rope_prefix=prefix_tokens)
            if b.qn_w is not None:
                ops.qk_norm_bwd(dqkv, ws.get(f"{i}.d.qkv_pre", (Mc, 3 * D), BF), ws.get(f"{i}.d.qinv", (Mc, 2 * heads), F32), b.qn_w, b.kn_w,
                                b.g_qn, b.g_kn, Mc, D)
            linear_bwd(ws, "qkv", b.qkv, dqkv, xn1, Mc, dxn)
            ops.norm_bwd(dxn, xs1, b.n1w, st1, gdy, dxc, None, b.gn1w, b.gn1b, Mc, D, self.kind)
            self._scatter(dxc, dy, segs, csegs, i, 0, 1.0, accumulate=False)
            OVERLAP.join()
            yield ("block", i)
        return dy

    @staticmethod
    def _rows(segs):
        """(first row, B, N, rope) of every segment of a row-concatenated token buffer."""
        r0 = 0
        for B, N, rope in segs:
            yield r0, B, N, rope
            r0 += B * N

    @staticmethod
    def _attn_rows(segs):
        """_rows with neighbouring segments of equal sequence length and the same RoPE tables merged into one batch (the clean
        images and the global crops of a list forward: one attention launch instead of two, a better last round of workgroups)."""
        out = []
        for r0, B, N, rope in Stack._rows(segs):
            same = out and (out[-1][3] is rope or (isinstance(out[-1][3], tuple) and isinstance(rope, tuple)
                                                   and out[-1][3][0] is rope[0] and out[-1][3][1] is rope[1]))
            if out and out[-1][2] == N and same:  # (RopeAugTabs of different items never merge: each item has its own draws)
                out[-1] = (out[-1][0], out[-1][1] + B, N, rope)
            else:
                out.append((r0, B, N, rope))
        return out

    # x: f32 [M, D] input residual.  Returns the output residual (f32 [M, D]).
    def forward(self, ws: Workspace, x, B: int, N: int, rope, prefix_tokens: int, train: bool, segs=None):
        """`segs` = [(B_i, N_i, rope_i)]: several batches of different sequence length concatenated along the token-row
        axis (the reference's list path, block.py:235-298 / utils.py:14-25 cat_keep_shapes): every linear / norm runs once
        over all rows, RoPE and attention run per segment on its row range."""
        D, H, heads = self.D, self.H, self.heads
        segs = [(B, N, rope)] if segs is None else segs
        M = sum(b * n for b, n, _ in segs)
        scale = 1.0 / math.sqrt(64.0)
        vit = self.swiglu  # (FFN kind; RoPE / naming follow self.style)
        fp8 = getattr(self, "fp8", None)
        if fp8 is not None and fp8["ready"] and not train:
            return self.forward_fp8(ws, x, segs, prefix_tokens, M)
        calib = fp8["amax"] if (fp8 is not None and not fp8["ready"] and not train) else None
        saved_all = []
        rope_plan = self._rope_plan(ws, segs, prefix_tokens, M) if FUSE_ROPE else None
        for i, b in enumerate(self.blocks):
            t = f"{i}." if train else ""
            xn1 = ws.get(t + "xn1", (M, D), BF)
            st1 = ws.get(t + "st1", (M, 2), F32)
            qkv = ws.get(t + "qkv", (M, 3 * D), BF)
            o = ws.get(t + "o", (M, D), BF)
            lse = ws.get(t + "lse", (M * heads,), F32)  # per segment [B_i, heads, N_i]
            xmid = ws.get(t + "xmid", (M, D), F32)
            xn2 = ws.get(t + "xn2", (M, D), BF)
            st2 = ws.get(t + "st2", (M, 2), F32)
            pre = ws.get(t + "x12", (M, 2 * H if vit else H), BF) if train else None  # FFN pre-activations
            hid = ws.get(t + "hid", (M, H), BF)
            xout = ws.get((f"{i}.xout" if train else f"xout{i & 1}"), (M, D), F32)
            x_in = x  # block input (previous block's xout buffer) is kept for norm1 backward

            ops.norm_fwd(x, b.n1w, b.n1b, xn1, st1, M, D, self.eps, self.kind)
            if rope_plan is not None:  # apply_rope rides in the epilogue of the qkv projection (all segments, one launch)
                ps, pc = self._plan_tabs(rope_plan, i)
                ops.gemm_qkv_rope(xn1, b.qkv.w, b.qkv.bias, qkv, M, 3 * D, D, rope_plan[0], ps, pc, 2 * D)
            elif b.qn_w is not None:  # projection -> QK norm (pre-norm values and 1/rms kept for the backward) -> RoPE below
                qkv_pre = ws.get(t + "qkv_pre", (M, 3 * D), BF)
                qinv = ws.get(t + "qinv", (M, 2 * heads), F32)
                ops.gemm_nt(xn1, b.qkv.w, qkv_pre, M=M, N=3 * D, K=D, bias=b.qkv.bias, epi=EPI_BF16)
                ops.qk_norm_fwd(qkv_pre, b.qn_w, b.kn_w, qkv, qinv, M, D)
            else:
                ops.gemm_nt(xn1, b.qkv.w, qkv, M=M, N=3 * D, K=D, bias=b.qkv.bias, epi=EPI_BF16)
            for r0, Bs, Ns, rp in self._attn_rows(segs):
                q_s, o_s = qkv[r0:r0 + Bs * Ns], o[r0:r0 + Bs * Ns]
                if rp is not None and rope_plan is None:
                    rs, rc = rope_at(rp, i)
                    ops.rope_qk(q_s, rs, rc, Bs, Ns, heads, prefix_tokens)
                ops.attn_fwd(q_s, q_s[:, D:], q_s[:, 2 * D:], o_s, lse[r0 * heads:], Bs, Ns, heads, Ns * 3 * D, 3 * D, Ns * D, D,
                             scale, self.causal)
            ops.gemm_nt(o, b.proj.w, xmid, M=M, N=D, K=D, bias=b.proj.bias, gamma=b.ls1, resid=x, epi=EPI_F32)
            ops.norm_fwd(xmid, b.n2w, b.n2b, xn2, st2, M, D, self.eps, self.kind)
            if vit:
                ops.gemm_nt(xn2, b.w12.w12, hid, M=M, N=2 * H, K=D, c2=pre, ldc2=2 * H, bias=b.w12.b12, epi=EPI_SWIGLU)
            else:
                ops.gemm_nt(xn2, b.fc.w, hid, M=M, N=H, K=D, c2=pre, ldc2=H, bias=b.fc.bias,
                            epi=ops.EPI_QUICK_GELU if self.quick_gelu else EPI_GELU)
            ops.gemm_nt(hid, b.w3.w, xout, M=M, N=D, K=H, bias=b.w3.bias, gamma=b.ls2, resid=xmid, epi=EPI_F32)
            if calib is not None:  # fp8 calibration pass: amax of the four GEMM inputs of this block
                for j, tns in enumerate((xn1, o, xn2, hid)):
                    ops.amax(tns, calib[i, j:j + 1])
            if train:
                saved_all.append((x_in, xn1, st1, qkv, o, lse, xmid, xn2, st2, pre, hid))
            x = xout
        self.last_saved = saved_all  # per-call context: several forward passes may be in flight before their backward
        return x

    # ---- fp8 (e4m3) inference forward: BASELINE config 5 (per-tensor scales: weights from their own amax, activations from a
    # calibration pass of the bf16 path over representative images)
    def fp8_begin_calibration(self):
        if not self.swiglu or self.D % 16 or self.H % 16 or any(b.ls1 is not None or b.ls2 is not None for b in self.blocks):
            raise NotImplementedError("fp8 forward: SwiGLU ViT blocks with D, H multiples of 16, no LayerScale")
        dev = self.store.device
        self.fp8 = {"ready": False, "amax": torch.zeros(self.depth, 4, dtype=F32, device=dev)}

    def fp8_finalize(self):
        """weights -> e4m3 with scale 448 / amax(W); activation scales 448 / calibrated amax (x 1 / margin)"""
        f, dev = self.fp8, self.store.device
        amax = f["amax"].clamp_min(1e-12)
        f["act_scale"] = (ops.E4M3_MAX / amax).contiguous()        # device [depth, 4]: read by the quantise kernels
        act_inv = (amax / ops.E4M3_MAX).cpu().tolist()               # host: the GEMM alphas
        f["w"], f["alpha"] = [], []
        for i, b in enumerate(self.blocks):
            ws8, al = [], []
            for j, w in enumerate((b.qkv.w, b.proj.w, b.w12.w12, b.w3.w)):
                wa = float(w.float().abs().max().clamp_min(1e-12))
                q = torch.empty(w.shape, dtype=torch.uint8, device=dev)
                ops.quantize_e4m3(w.contiguous(), q, ops.E4M3_MAX / wa)
                ws8.append(q)
                al.append(act_inv[i][j] * wa / ops.E4M3_MAX)
            f["w"].append(ws8)
            f["alpha"].append(al)
        f["ready"] = True

    def forward_fp8(self, ws: Workspace, x, segs, prefix_tokens: int, M: int):
        """Stack.forward(train=False) with the four linear maps of every block on the fp8 MFMA path: norm / attention / SwiGLU
        outputs are quantised (one extra row pass each) and multiplied against e4m3 weights; accumulation, bias, residual and
        the attention itself are unchanged (fp32 / bf16)."""
        D, H, heads, f = self.D, self.H, self.heads, self.fp8
        scale = 1.0 / math.sqrt(64.0)
        a8 = ws.get("f8.a", (M, max(D, H)), torch.uint8)
        qkv = ws.get("qkv", (M, 3 * D), BF)
        o = ws.get("o", (M, D), BF)
        lse = ws.get("lse", (M * heads,), F32)
        xmid = ws.get("xmid", (M, D), F32)
        hid = ws.get("hid", (M, H), BF)
        rope_plan = self._rope_plan(ws, segs, prefix_tokens, M) if FUSE_ROPE else None
        rope_arg = None if rope_plan is None else (rope_plan[0], rope_plan[1], rope_plan[2], 2 * D)
        for i, b in enumerate(self.blocks):
            w8, al, sc = f["w"][i], f["alpha"][i], f["act_scale"][i]
            xout = ws.get(f"xout{i & 1}", (M, D), F32)
            ops.norm_fwd_e4m3(x, b.n1w, b.n1b, a8, sc[0:1], None, M, D, self.eps, self.kind)  # norm + quantise in one pass
            if b.qn_w is not None:  # QK normalisation (round 5): e4m3 projection -> bf16 pre-norm q, k -> RMSNorm(head_dim) -> RoPE below
                qkv_pre = ws.get("f8.qkv_pre", (M, 3 * D), BF)
                ops.gemm_nt_fp8(a8, w8[0], qkv_pre, M=M, N=3 * D, K=D, alpha=al[0], bias=b.qkv.bias, epi=EPI_BF16)
                ops.qk_norm_fwd(qkv_pre, b.qn_w, b.kn_w, qkv, ws.get("f8.qinv", (M, 2 * heads), F32), M, D)
            else:
                ops.gemm_nt_fp8(a8, w8[0], qkv, M=M, N=3 * D, K=D, alpha=al[0], bias=b.qkv.bias, epi=EPI_BF16, rope=rope_arg)
            for r0, Bs, Ns, rp in self._attn_rows(segs):
                q_s, o_s = qkv[r0:r0 + Bs * Ns], o[r0:r0 + Bs * Ns]
                if rp is not None and rope_arg is None:
                    ops.rope_qk(q_s, rp[0], rp[1], Bs, Ns, heads, prefix_tokens)
                ops.attn_fwd(q_s, q_s[:, D:], q_s[:, 2 * D:], o_s, lse[r0 * heads:], Bs, Ns, heads, Ns * 3 * D, 3 * D, Ns * D, D,
                             scale, self.causal)
            ops.quantize_e4m3(o, a8, sc[1:2])
            ops.gemm_nt_fp8(a8, w8[1], xmid, M=M, N=D, K=D, alpha=al[1], bias=b.proj.bias, resid=x, epi=EPI_F32)
            ops.norm_fwd_e4m3(xmid, b.n2w, b.n2b, a8, sc[2:3], None, M, D, self.eps, self.kind)
            ops.gemm_nt_fp8(a8, w8[2], hid, M=M, N=2 * H, K=D, alpha=al[2], bias=b.w12.b12, epi=EPI_SWIGLU)
            ops.quantize_e4m3(hid, a8, sc[3:4])
            ops.gemm_nt_fp8(a8, w8[3], xout, M=M, N=D, K=H, alpha=al[3], bias=b.w3.bias, resid=xmid, epi=EPI_F32)
            x = xout
        return x

    def w3_colsum_target(self, i: int):
        """where the norm backward in front of block i's w3 gradient may add the column sums of its bf16 output (= w3's bias
        gradient); None when a LayerScale sits between them (the bias gradient then needs the gamma factor) or under
        stochastic depth (the FFN branch sees a gathered subset of the rows)"""
        b = self.blocks[i]
        return b.w3.gb if (b.ls2 is None and self.drop_plan is None) else None

    # dy: f32 [M,D] grad of the stack output, dy_b: its bf16 copy.  Returns (dx f32, dx bf16) for the stack input.
    def dx_buffers(self, ws: Workspace, M: int):
        """(f32, bf16) buffers backward() returns the stack-input gradient in (block 0's outputs; static like every workspace buffer)"""
        return ws.get("b.dx0", (M, self.D), F32), ws.get("b.dx_b0", (M, self.D), BF)

    def backward(self, ws: Workspace, dy, dy_b, B: int, N: int, rope, prefix_tokens: int, saved=None, segs=None,
                 dy_colsum_done: bool = False, hold_last: bool = False, extra_last=None):
        """hold_last (grouped weight gradients only): block 0's group is neither launched nor announced here but left in self.held =
        (0, group) -- the caller launches it and yields ("block", 0) -- and extra_last, a list of linear_bwd-style problem records over the
        same token rows (TrunkEngine: the patch-embed weight gradient, whose dy is this stack's input gradient), rides in that launch.
        Generator: yields ("block", i) each time all parameter gradients of block i have been enqueued (a
        gradient-bucket / graph-segment boundary for the trainer); returns (dx f32, dx bf16) of the stack input.
        dy_colsum_done: the caller's norm backward already summed dy_b's columns into the last block's w3 bias gradient
        (self.blocks[-1].w3.gb as its dx_colsum)."""
        D, H, heads = self.D, self.H, self.heads
        segs = [(B, N, rope)] if segs is None else segs
        M = sum(b * n for b, n, _ in segs)
        scale = 1.0 / math.sqrt(64.0)
        vit = self.swiglu  # (FFN kind)
        # grouped weight gradients: the four dW of block i are ONE launch (split-K combine and bias-gradient column sums inside
        # it), issued at the start of block i - 1's backward so that it overlaps that block's dgrad / attention kernels; the dy
        # operands it reads (dpre, dmid_b, dqkv; dy_b already alternates) are therefore double-buffered by block parity
        grouped = WGRAD_GROUPED and M >= 256 and all(b.ls1 is None and b.ls2 is None for b in self.blocks) \
            and ops.wgrad_group_fits(M, max(3 * D, 2 * H if vit else H)) and D % 8 == 0 and H % 8 == 0  # else: per-layer launches (ring-kernel fallback inside)
        par = (lambda i: f".{i & 1}") if grouped else (lambda i: "")
        dh = ws.get("b.dh", (M, H), BF)
        d_o = ws.get("b.do", (M, D), BF)
        dxn = ws.get("b.dxn", (M, D), BF)
        delta = ws.get("b.delta", (M * heads,), F32)
        dmid = ws.get("b.dmid", (M, D), F32)
        saved = self.last_saved if saved is None else saved
        groups = ws.__dict__.setdefault("_wgrad_groups", {})
        scratch = self.__dict__.setdefault("_wgrad_scratch", {})
        pending = None  # (block index, ops.WgradGroup) whose launch is due

        def launch_pending():
            """fork now (the side stream picks up behind the norm backward that closed the previous block); the launch itself is
            issued behind this block's first main-stream kernel (Overlap.defer)"""
            if pending is None:
                return
            if OVERLAP.enabled and not WGRAD_INLINE:
                OVERLAP.fork()
                side, grp = OVERLAP.side, pending[1]

                def issue():
                    with torch.cuda.stream(side):
                        grp.launch()
                OVERLAP.defer(issue)
            else:
                pending[1].launch()

        for i in range(self.depth - 1, -1, -1):
            b = self.blocks[i]
            x_in, xn1, st1, qkv, o, lse, xmid, xn2, st2, pre, hid = saved[i]
            dxo = ws.get(f"b.dx{i & 1}", (M, D), F32)
            dxo_b = ws.get(f"b.dx_b{i & 1}", (M, D), BF)
            dpre = ws.get("b.dx12" + par(i), (M, 2 * H if vit else H), BF)
            dmid_b = ws.get("b.dmid_b" + par(i), (M, D), BF)
            dqkv = ws.get("b.dqkv" + par(i), (M, 3 * D), BF)
            probs = [] if grouped else None
            if grouped:
                launch_pending()  # dW of block i + 1, beside this block's kernels
            # ---- FFN: x_out = x_mid + w3(act(...))
            fused_act = vit and FUSE_SWIGLU_BWD and b.ls2 is None
            linear_bwd(ws, "w3", b.w3, dy_b, hid, M, dpre if fused_act else dh,
                       bias_grad_done=dy_colsum_done if i == self.depth - 1 else self.w3_colsum_target(i) is not None,
                       ls=(b.ls2, b.gls2) if b.ls2 is not None else None, dgrad_swiglu=pre if fused_act else None, defer=probs)
            OVERLAP.run_deferred()  # side work forked at the block boundary: behind the block's first main-stream kernel
            if vit:
                if not fused_act:
                    ops.swiglu_bwd(dh, pre, dpre, M, H)
                linear_bwd(ws, "w12", None, dpre, xn2, M, dxn, N=2 * H, K=D, gw=b.w12.gw1, gb=b.w12.gb1, wT=b.w12.w12T,
                           swiglu_h=H, defer=probs)
            else:
                ops.gelu_bwd(dh, pre, dpre, M * H, quick=self.quick_gelu)
                linear_bwd(ws, "fc", b.fc, dpre, xn2, M, dxn, defer=probs)
            # the norm backward kernels also sum the columns of their bf16 output = the bias gradient of the linear layer that
            # takes it as dy (proj here; the previous block's w3 below)
            ops.norm_bwd(dxn, xmid, b.n2w, st2, dy, dmid, dmid_b, b.gn2w, b.gn2b, M, D, self.kind,
                         dx_colsum=b.proj.gb if b.ls1 is None else None)
            # ---- attention: x_mid = x_in + proj(attn(rope(qkv(xn1))))
            linear_bwd(ws, "proj", b.proj, dmid_b, o, M, d_o, bias_grad_done=b.proj.gb is not None,
                       ls=(b.ls1, b.gls1) if b.ls1 is not None else None, defer=probs)
            for r0, Bs, Ns, rp in self._attn_rows(segs):
                r1 = r0 + Bs * Ns
                q_s, dq_s = qkv[r0:r1], dqkv[r0:r1]
                # dq / dk come back as gradients w.r.t. the un-rotated q, k (inverse RoPE fused into the attention backward)
                ops.attn_bwd(q_s, q_s[:, D:], q_s[:, 2 * D:], o[r0:r1], d_o[r0:r1], lse[r0 * heads:], delta[r0 * heads:], dq_s,
                             dq_s[:, D:], dq_s[:, 2 * D:], Bs, Ns, heads, Ns * 3 * D, 3 * D, Ns * D, D, scale, self.causal,
                             rope=None if rp is None else rope_at(rp, i), rope_prefix=prefix_tokens)
            if b.qn_w is not None:  # gradient w.r.t. the normalised q, k -> w.r.t. the projection output (in place), + dw
                ops.qk_norm_bwd(dqkv, ws.get(f"{i}.qkv_pre", (M, 3 * D), BF), ws.get(f"{i}.qinv", (M, 2 * heads), F32), b.qn_w, b.kn_w,
                                b.g_qn, b.g_kn, M, D)
            linear_bwd(ws, "qkv", b.qkv, dqkv, xn1, M, dxn, defer=probs)
            if grouped:  # dW of block i + 1 is done: the norm backward below overwrites the dy operand it read (b.dx_b, same parity)
                OVERLAP.join()
            ops.norm_bwd(dxn, x_in, b.n1w, st1, dmid, dxo, dxo_b, b.gn1w, b.gn1b, M, D, self.kind,
                         dx_colsum=self.w3_colsum_target(i - 1) if i > 0 else None)
            dy, dy_b = dxo, dxo_b
            if not grouped:
                OVERLAP.join()
                yield ("block", i)
                continue
            extra = extra_last if (i == 0 and hold_last and extra_last) else []
            gkey = (i, bool(self.wgrad_overwrite), len(extra))
            grp = groups.get(gkey)
            if grp is None:
                grp = ops.WgradGroup(M)
                for pr in probs + list(extra):
                    grp.add(pr["dy"], pr["x"], pr["gw"], pr["gb"], pr["N"], pr["K"], pr["swiglu_h"], accumulate=not self.wgrad_overwrite)
                groups[gkey] = grp.finalize(self.store.device, scratch)
            if pending is not None:
                yield ("block", pending[0])  # its weight gradients are complete (joined above)
            pending = (i, grp)
        self.held = None
        if pending is not None and hold_last:
            OVERLAP.run_deferred()
            self.held = pending
        elif pending is not None:  # block 0's group: nothing of this stack is left to run beside it
            pending[1].launch()
            yield ("block", pending[0])
        return dy, dy_b


# =====================================================================================================================
# RoPE tables (host, cached): exact op order of RopePositionEmbedding.forward (embeddings.py:131-180) in the dtype of
# the `periods` buffer (bf16) -- recomputing sin/cos in fp32 on the device would NOT match the reference
# =====================================================================================================================
_ROPE_CACHE: Dict[tuple, Tuple[torch.Tensor, torch.Tensor]] = {}


def _rope_host(per: torch.Tensor, H: int, W: int, aug=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """(sin, cos) bf16 [H*W, 64] on the host, op for op RopePositionEmbedding.forward (embeddings.py:131-180) in the dtype of `per`.
    aug = {"shift": [2] | None, "jitter": [2] | None, "rescale": [1] | None}: the train-time coordinate augmentations (:155-171) with
    GIVEN draws (shift added per axis; jitter -- already exp() of the log-uniform draw -- multiplied per axis; rescale multiplied)."""
    dd = {"dtype": per.dtype}
    ch = torch.arange(0.5, H, **dd) / H
    cw = torch.arange(0.5, W, **dd) / W
    coords = torch.stack(torch.meshgrid(ch, cw, indexing="ij"), dim=-1).flatten(0, 1)
    coords = 2.0 * coords - 1.0
    if aug is not None:
        if aug.get("shift") is not None:
            coords += aug["shift"][None, :]
        if aug.get("jitter") is not None:
            coords *= aug["jitter"][None, :]
        if aug.get("rescale") is not None:
            coords *= aug["rescale"]
    ang = 2 * math.pi * coords[:, :, None] / per[None, None, :]
    ang = ang.flatten(1, 2).tile(2)
    return torch.sin(ang).to(torch.bfloat16).contiguous(), torch.cos(ang).to(torch.bfloat16).contiguous()


def rope_tables(periods: torch.Tensor, H: int, W: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    per = periods.detach().to("cpu")
    key = (H, W, str(device), per.dtype, tuple(per.float().tolist()))
    hit = _ROPE_CACHE.get(key)
    if hit is not None:
        return hit
    sin, cos = _rope_host(per, H, W)
    sin, cos = sin.to(device), cos.to(device)
    _ROPE_CACHE[key] = (sin, cos)
    return sin, cos


class RopeAugTabs:
    """RoPE tables of ONE list item under train-time augmentation: rows [off, off + hw) of the static per-block buffers
    sin_all / cos_all bf16 [depth, P, 64] (P = patch tokens of all items).  The trunk evaluates rope_embed inside its block loop
    (vision_transformer.py:228-233), so every block rotates with its own draw: Stack asks at(i) for block i's tables."""
    __slots__ = ("sin_all", "cos_all", "off", "hw")

    def __init__(self, sin_all, cos_all, off, hw):
        self.sin_all, self.cos_all, self.off, self.hw = sin_all, cos_all, off, hw

    def at(self, i: int):
        i = min(i, self.sin_all.shape[0] - 1)  # one shared draw for all blocks (pixel decoder): buffers [1, P, 64]
        return self.sin_all[i, self.off:self.off + self.hw], self.cos_all[i, self.off:self.off + self.hw]


def rope_at(rp, i: int):
    """(sin, cos) [hw, 64] of block i: per-block tables under augmentation, the shared pair otherwise"""
    return rp.at(i) if isinstance(rp, RopeAugTabs) else rp


def _rope_hw(rp) -> int:
    return rp.hw if isinstance(rp, RopeAugTabs) else rp[0].shape[0]


class RopeAugmenter:
    """Train-time RoPE coordinate augmentations (RopePositionEmbedding shift / jitter / rescale, embeddings.py:155-171; constructor
    arguments pos_embed_rope_*_coords of the ViT classes, reachable through the legacy YAML's vision_encoder / pixel_decoder sections):
    host-side draws in the rope dtype, tables built with the reference's op order, uploaded into STATIC device buffers that are only ever
    refreshed in place (captured hipGraph segments bake their addresses).  per_block: one draw per block and list item (trunk) or one
    per forward (pixel decoder, pixel_decoder.py:144).  An eager forward refreshes its buffer itself; under stream capture nothing
    is drawn -- the trainer calls refresh_all() before every replay."""

    def __init__(self, periods: torch.Tensor, depth: int, per_block: bool, shift, jitter, rescale, seed: int = 0):
        self.per, self.depth, self.per_block = periods.detach().to("cpu"), depth, per_block
        self.cfg = (shift, jitter, rescale)
        self.engine_id = int(seed)
        self.gen = torch.Generator().manual_seed(int(seed))
        self.records: Dict[tuple, dict] = {}
        self.touched = set()    # record keys handed out since the trainer last cleared it (what ONE captured step reads)
        self.last_draws = None  # [[draw dict per item] per block] of the latest refresh (tests replay them into the oracle)

    def reseed(self, base_seed: int, rank: int = 0):
        """the reference draws from the per-rank device RNG: data-parallel ranks must not share their augmentations.  The trainer
        calls this with its seed and rank; the stream of a (seed, rank, engine) triple is reproducible."""
        self.gen.manual_seed((int(base_seed) * 1000003 + int(rank)) * 8 + self.engine_id)

    def get_state(self) -> torch.Tensor:
        return self.gen.get_state().clone()

    def set_state(self, state: torch.Tensor):
        self.gen.set_state(state.to("cpu", torch.uint8))

    @property
    def active(self) -> bool:
        return any(v is not None for v in self.cfg)

    def _draw(self):
        import numpy as np
        shift, jitter, rescale = self.cfg
        dd = {"dtype": self.per.dtype}
        d = {"shift": None, "jitter": None, "rescale": None}
        if shift is not None:
            d["shift"] = torch.empty(2, **dd).uniform_(-shift, shift, generator=self.gen)
        if jitter is not None:
            jm = float(np.log(jitter))
            d["jitter"] = torch.empty(2, **dd).uniform_(-jm, jm, generator=self.gen).exp()
        if rescale is not None:
            rm = float(np.log(rescale))
            d["rescale"] = torch.empty(1, **dd).uniform_(-rm, rm, generator=self.gen).exp()
        return d

    def record(self, key, hws, device):
        """static buffers for a forward whose list items have (h, w) = hws"""
        rec = self.records.get(key)
        if rec is None:
            P = sum(h * w for h, w in hws)
            nb = self.depth if self.per_block else 1
            rec = dict(hws=list(hws), sin=torch.zeros(nb, P, 64, dtype=torch.bfloat16, device=device),
                       cos=torch.zeros(nb, P, 64, dtype=torch.bfloat16, device=device),
                       host=torch.zeros(2, nb, P, 64, dtype=torch.bfloat16).pin_memory(), event=None, fresh=False)
            self.records[key] = rec
        return rec

    def refresh(self, rec):
        """new draws -> host tables -> one non-blocking upload per table"""
        if rec["event"] is not None:
            rec["event"].synchronize()  # the pinned staging rows are rewritten only after the previous upload has executed
        nb = rec["sin"].shape[0]
        draws = []
        for i in range(nb):
            row, off = [], 0
            for h, w in rec["hws"]:
                d = self._draw()
                sin, cos = _rope_host(self.per, h, w, d)
                rec["host"][0, i, off:off + h * w].copy_(sin)
                rec["host"][1, i, off:off + h * w].copy_(cos)
                off += h * w
                row.append(d)
            draws.append(row)
        rec["sin"].copy_(rec["host"][0], non_blocking=True)
        rec["cos"].copy_(rec["host"][1], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        rec["event"], rec["fresh"] = ev, True
        self.last_draws = draws

    def tables(self, key, hws, device):
        """the record of this forward, refreshed unless the stream is being captured (then the trainer refreshes before each replay)"""
        rec = self.record(key, hws, device)
        self.touched.add(key)
        if torch.cuda.is_current_stream_capturing():
            if not rec["fresh"]:
                raise RuntimeError("RoPE augmentation tables must exist before stream capture (run the step eagerly once)")
        else:
            self.refresh(rec)
        return rec

    def refresh_all(self, keys=None):
        """new draws for the records a replayed graph reads (`keys`: what tables() handed out while that graph was captured; None:
        every record ever created -- host cost and one event wait per record, so the trainer passes the keys)"""
        for k, rec in self.records.items():
            if keys is None or k in keys:
                self.refresh(rec)


# =====================================================================================================================
# ViT trunk with bottleneck (vision_transformer.py:189-264, vision_transformer_bottleneck.py:48-79)
# =====================================================================================================================
class TrunkEngine:
    def __init__(self, store: ParamStore, cfg, periods: torch.Tensor, prefix: str = "trunk."):
        from .config import ffn_hidden
        self.store = store
        self.prefix = prefix  # "trunk." (student) or "teacher_trunk." (EMA teacher, vtp.py:253-268)
        self.D, self.heads, self.depth = cfg.vision_embed_dim, cfg.vision_num_heads, cfg.vision_depth
        self.H = ffn_hidden(self.D, cfg.vision_mlp_ratio, cfg.vision_ffn_layer)
        self.kind = ops.NORM_RMS if cfg.vision_norm_layer == "rmsnorm" else ops.NORM_LN
        self.eps = 1e-5 if self.kind == ops.NORM_RMS else 1e-6
        self.periods = periods.detach().to("cpu")  # host copy: rope_tables must not touch the device during graph capture
        self.pe = store.lin(self.prefix + "patch_embed.proj.weight", self.prefix + "patch_embed.proj.bias", self.D, 768)  # W^T: the input-image gradient
        self.stack = Stack(store, self.prefix + "blocks.", self.depth, self.D, self.heads, self.H, cfg.vision_norm_layer,
                           ffn=cfg.vision_ffn_layer)
        self.bott_dim = cfg.vision_feature_bottleneck
        self.bott = store.lin(self.prefix + "feature_bottleneck.weight", None, self.bott_dim, self.D) \
            if store.has(self.prefix + "feature_bottleneck.weight") else None
        self.ws: Dict[tuple, Workspace] = {}
        # train-time RoPE coordinate augmentations (embeddings.py:155-171): one draw per block and list item
        self.rope_aug = RopeAugmenter(self.periods, self.depth, True, getattr(cfg, "vision_rope_shift_coords", None),
                                      getattr(cfg, "vision_rope_jitter_coords", None), getattr(cfg, "vision_rope_rescale_coords", None),
                                      seed=1 if prefix == "trunk." else 2)

    def workspace(self, key) -> Workspace:
        if key not in self.ws:
            self.ws[key] = Workspace(self.store.device)
        return self.ws[key]

    def forward(self, img: torch.Tensor, train: bool, masks: Optional[torch.Tensor] = None, tag: str = "", rope_aug: Optional[bool] = None):
        """img f32 [B,3,H,W] -> final-norm tokens xnf bf16 [B*N, D] (N = 1 + hw).  masks: uint8 [B, hw] (1 = replace the
        patch embedding by mask_token, vision_transformer.py:194-196).  `tag` separates the buffers of passes that are in
        flight at the same time with the same shape; self.ctx() returns the handle backward() needs for such passes."""
        return self.forward_list([(img, masks)], train, tag, rope_aug=rope_aug)

    def forward_list(self, items, train: bool, tag: str = "", rope_aug: Optional[bool] = None):
        """items = [(img f32 [B_i,3,H_i,W_i], masks_i or None)]: the reference's list forward (forward_features_list,
        vision_transformer.py:221-258): batches of different resolution go through the blocks as ONE row-concatenated
        token buffer (one GEMM / norm launch per layer for all of them, attention + RoPE per segment).  Returns xnf bf16
        [sum_i B_i*N_i, D]; ctx().segs[i].row0 is the first row of item i."""
        st = self.store
        D = self.D
        segs, r0, p0 = [], 0, 0
        for img, masks in items:
            B, _, Hh, Ww = img.shape
            h, w = Hh // 16, Ww // 16
            sg = TrunkSeg()
            sg.B, sg.h, sg.w, sg.hw, sg.N, sg.row0, sg.prow0, sg.masks, sg.img = B, h, w, h * w, h * w + 1, r0, p0, masks, img
            sg.rope = rope_tables(self.periods, h, w, st.device)
            segs.append(sg)
            r0 += B * sg.N
            p0 += B * sg.hw
        M, P = r0, p0
        ws = self.workspace((tuple((g.B, g.h, g.w) for g in segs), tag))
        # rope_aug: None = follow `train` (RopePositionEmbedding augments when its module is in training mode); the EMA teacher's pass
        # stores nothing for a backward (train=False) but belongs to a training step: its caller passes rope_aug=True
        if self.rope_aug.active and (train if rope_aug is None else rope_aug):
            rec = self.rope_aug.tables((tuple((g.B, g.h, g.w) for g in segs), tag), [(g.h, g.w) for g in segs], st.device)
            off = 0
            for g in segs:
                g.rope = RopeAugTabs(rec["sin"], rec["cos"], off, g.hw)
                off += g.hw
        # im2col rows in the TOKEN layout (row 0 of every image is never written and stays zero): the patch-embed weight gradient is then
        # one product over the same rows as the blocks' weight gradients -- d_tokens^T patches with the cls rows contributing zero -- and
        # rides in block 0's grouped launch (backward()) instead of three few-tile split-K launches at the end of the step
        patches = ws.get("patches_tok", (M, 768), BF, zero=True)
        x0 = ws.get("x0", (M, D), F32)
        for g in segs:
            pt, xs = patches[g.row0:g.row0 + g.B * g.N], x0[g.row0:g.row0 + g.B * g.N]
            ops.im2col16_rows(g.img, pt, g.B, g.h * 16, g.w * 16, 1)
            ops.gemm_nt(pt, self.pe.w, xs, M=g.B * g.hw, N=D, K=768, bias=self.pe.bias, epi=EPI_F32, a_remap=(g.hw, 1), c_remap=(g.hw, 1))
            ops.assemble_tokens(xs, st.p(self.prefix + "cls_token"), st.p(self.prefix + "mask_token"), g.masks, g.B, g.N, D)
        stack_segs = [(g.B, g.N, g.rope) for g in segs]
        dropped = train and self.stack.drop_plan is not None
        if dropped:  # stochastic depth (block.py:207-289): residual branches on random image subsets, x0 updated in place
            xl = self.stack.forward_drop(ws, x0, stack_segs, 1)
        else:
            xl = self.stack.forward(ws, x0, 0, 0, None, 1, train, segs=stack_segs)
        xnf = ws.get("xnf", (M, D), BF)
        stf = ws.get("stf", (M, 2), F32)
        ops.norm_fwd(xl, st.p(self.prefix + "norm.weight"), st.p(self.prefix + "norm.bias") if self.kind == ops.NORM_LN else None, xnf, stf,
                     M, D, self.eps, self.kind)
        c = TrunkCtx()
        c.ws, c.segs, c.M, c.xl, c.xnf, c.stf, c.patches, c.stack_saved, c.x0, c.stack_segs = \
            ws, segs, M, xl, xnf, stf, patches, self.stack.last_saved, x0, stack_segs
        c.dropped = dropped
        self._ctx = c
        return xnf

    def ctx(self):
        return self._ctx

    def latents(self, out_f32: bool = False, seg: int = 0) -> torch.Tensor:
        """bottleneck on the patch rows of item `seg` of the last forward -> [B*hw, 64] (bf16, or f32 for the API)."""
        c = self._ctx
        g = c.segs[seg]
        lat = c.ws.get("lat32" if out_f32 else "lat", (g.B * g.hw, self.bott_dim), F32 if out_f32 else BF)
        ops.gemm_nt(c.xnf[g.row0:], self.bott.w, lat, M=g.B * g.hw, N=self.bott_dim, K=self.D,
                    epi=EPI_F32 if out_f32 else EPI_BF16, a_remap=(g.hw, 1))
        return lat

    def d_xnf_buffer(self, ctx=None) -> torch.Tensor:
        """bf16 [M, D] gradient w.r.t. the final-norm tokens of the last forward (all items, row-concatenated).  The patch
        rows of the latent item are written by backward() (bottleneck dgrad); every other row must be written (or zeroed)
        by the heads before backward()."""
        c = self._ctx if ctx is None else ctx
        return c.ws.get("b.d_xnf", (c.M, self.D), BF, zero=True)

    def backward(self, d_lat: Optional[torch.Tensor], ctx=None, lat_seg: int = 0, want_dimg: bool = False):
        """d_lat: bf16 [B*hw, 64] grad of latents(seg=lat_seg), or None.  Accumulates every trunk parameter gradient into
        store.flat_g.  Generator (see Stack.backward): yields "tail", then ("block", i) per block.
        want_dimg: also form the gradient w.r.t. the input images (PatchEmbed backward, embeddings.py:61-70: d_tokens[patch rows]
        W_pe folded back to pixels) -- ctx.d_img[i] f32 [B_i,3,H_i,W_i] for list item i (freshly allocated: this path is the eager
        autograd boundary, never a captured segment; the caller pops the entry -- ctx.take_d_img(i) -- so the tensor does not outlive the
        autograd call on the cached ctx); masked patches get zero (they never saw the pixels)."""
        st = self.store
        if ctx is not None:
            self._ctx = ctx
        c = self._ctx
        ws, M, D = c.ws, c.M, self.D
        d_xnf = ws.get("b.d_xnf", (M, D), BF, zero=True)  # rows stay zero unless a head wrote them before backward()
        if d_lat is not None:
            g = c.segs[lat_seg]
            linear_bwd(ws, "bott", self.bott, d_lat, c.xnf[g.row0:], g.B * g.hw, d_xnf[g.row0:], x_remap=(g.hw, 1),
                       dx_remap=(g.hw, 1))
        dx = ws.get("b.dxt", (M, D), F32)
        dx_b = ws.get("b.dxt_b", (M, D), BF)
        ops.norm_bwd(d_xnf, c.xl, st.p(self.prefix + "norm.weight"), c.stf, None, dx, dx_b, st.g(self.prefix + "norm.weight"),
                     st.g(self.prefix + "norm.bias") if self.kind == ops.NORM_LN else None, M, D, self.kind,
                     dx_colsum=self.stack.w3_colsum_target(self.depth - 1))
        OVERLAP.join()
        yield "tail"
        g_cls, g_mask = st.g(self.prefix + "cls_token"), st.g(self.prefix + "mask_token")
        if getattr(c, "dropped", False):
            dx0 = yield from self.stack.backward_drop(ws, dx, c.stack_segs, 1, c.stack_saved)
            dx0_b = dx_b
            ops.cast_f32_bf16(dx0, dx0_b, M * D)
            self.stack.held = None
        else:
            # the patch-embed weight (and bias) gradient as a fifth problem of block 0's grouped launch: dy = the stack's input gradient
            # (cls and masked rows zeroed by token_rows_bwd below, before the launch), x = the im2col rows in the token layout
            extra = None
            if PE_IN_GROUP and not want_dimg and D % 8 == 0:
                extra = [dict(dy=self.stack.dx_buffers(ws, M)[1], x=c.patches, gw=self.pe.gw, gb=self.pe.gb, N=self.pe.N, K=self.pe.K, swiglu_h=0)]
            dx0, dx0_b = yield from self.stack.backward(ws, dx, dx_b, 0, 0, None, 1, c.stack_saved, segs=c.stack_segs,
                                                        dy_colsum_done=self.stack.w3_colsum_target(self.depth - 1) is not None,
                                                        hold_last=extra is not None, extra_last=extra)
        held, self.stack.held = getattr(self.stack, "held", None), None
        # backward of the token assembly (vision_transformer.py:189-219): masked rows -> mask_token, row 0 -> cls_token, both zeroed in the
        # bf16 copy (what is left there is the gradient of the patch-embed output, in token rows)
        for g in c.segs:
            d_s, d_sb = dx0[g.row0:g.row0 + g.B * g.N], dx0_b[g.row0:g.row0 + g.B * g.N]
            ops.token_rows_bwd(d_s, d_sb, g.masks, g_mask if g.masks is not None else None, g_cls, g.B, g.N, D)
        if held is not None:
            assert dx0_b.data_ptr() == self.stack.dx_buffers(ws, M)[1].data_ptr()
            held[1].launch()
            yield ("block", held[0])
        else:
            for item, g in enumerate(c.segs):  # patch embed: dW += d_tokens[patch rows]^T patches, one launch sequence per item
                d_sb = dx0_b[g.row0:g.row0 + g.B * g.N]
                linear_bwd(ws, "pe", self.pe, d_sb, c.patches[g.row0:g.row0 + g.B * g.N], g.B * g.hw, None, need_dx=False,
                           dy_remap=(g.hw, 1), x_remap=(g.hw, 1))
                if want_dimg:
                    dpt = ws.get(f"b.dpatch{g.prow0}", (g.B * g.hw, 768), F32)
                    ops.gemm_nt(d_sb, self.pe.wT, dpt, M=g.B * g.hw, N=768, K=D, epi=EPI_F32, a_remap=(g.hw, 1))
                    dimg = torch.empty(g.B, 3, g.h * 16, g.w * 16, dtype=F32, device=st.device)
                    ops.col2im16(dpt, dimg, g.B, g.h * 16, g.w * 16)
                    c.__dict__.setdefault("d_img", {})[item] = dimg
        OVERLAP.join()


class TrunkSeg:
    """One item of a list forward: B images of (16h x 16w) pixels -> rows [row0, row0 + B*N) of the token buffers."""


class TrunkCtx:
    """Saved state of one trunk forward (what backward() needs)."""

    def take_d_img(self, item: int = 0):
        """the input-image gradient of list item `item` formed by backward(want_dimg=True); removes it from the ctx"""
        return self.__dict__.get("d_img", {}).pop(item)


# =====================================================================================================================
# pixel decoder (pixel_decoder.py:134-162): 1x1 conv -> blocks -> LN -> 1x1 conv -> PixelShuffle(16)
# =====================================================================================================================
class DecoderEngine:
    def __init__(self, store: ParamStore, cfg, periods: torch.Tensor):
        from .config import ffn_hidden
        self.store = store
        self.D, self.heads, self.depth = cfg.decoder_embed_dim, cfg.decoder_num_heads, cfg.decoder_depth
        self.H = ffn_hidden(self.D, 4.0, cfg.decoder_ffn_layer)
        self.kind = ops.NORM_RMS if cfg.decoder_norm_layer == "rmsnorm" else ops.NORM_LN
        self.eps = 1e-5 if self.kind == ops.NORM_RMS else 1e-6
        self.periods = periods.detach().to("cpu")
        self.cin = cfg.vision_feature_bottleneck
        self.pin = store.lin("pixel_decoder.proj_in.weight", "pixel_decoder.proj_in.bias", self.D, self.cin)
        self.stack = Stack(store, "pixel_decoder.blocks.", self.depth, self.D, self.heads, self.H, cfg.decoder_norm_layer,
                           ffn=cfg.decoder_ffn_layer)
        self.pout = store.lin("pixel_decoder.proj_out.weight", "pixel_decoder.proj_out.bias", 768, self.D)
        self.ws: Dict[tuple, Workspace] = {}
        # train-time RoPE augmentations: the decoder evaluates rope_embed ONCE per forward (pixel_decoder.py:144)
        self.rope_aug = RopeAugmenter(self.periods, self.depth, False, getattr(cfg, "decoder_rope_shift_coords", None),
                                      getattr(cfg, "decoder_rope_jitter_coords", None), getattr(cfg, "decoder_rope_rescale_coords", None), seed=3)

    def workspace(self, B, h, w) -> Workspace:
        key = (B, h, w)
        if key not in self.ws:
            self.ws[key] = Workspace(self.store.device)
        return self.ws[key]

    def forward(self, lat: torch.Tensor, B: int, h: int, w: int, train: bool) -> torch.Tensor:
        """lat bf16 [B*hw, 64] token-major -> t bf16 [B*hw, 768] (pre-PixelShuffle, token-major)."""
        st = self.store
        M, D = B * h * w, self.D
        ws = self.workspace(B, h, w)
        x0 = ws.get("x0", (M, D), F32)
        ops.gemm_nt(lat, self.pin.w, x0, M=M, N=D, K=self.cin, bias=self.pin.bias, epi=EPI_F32)
        rope = rope_tables(self.periods, h, w, st.device)
        if train and self.rope_aug.active:
            rec = self.rope_aug.tables((B, h, w), [(h, w)], st.device)
            # (RopeAugTabs, not a plain pair: _rope_plan keeps a COPY of plain tables, keyed by their address -- a buffer that is
            # refreshed in place would leave the fused qkv + RoPE epilogue rotating with the first step's draw)
            rope = RopeAugTabs(rec["sin"], rec["cos"], 0, h * w)
        dropped = train and self.stack.drop_plan is not None
        if dropped:
            xl = self.stack.forward_drop(ws, x0, [(B, h * w, rope)], 0)
        else:
            xl = self.stack.forward(ws, x0, B, h * w, rope, 0, train)
        xnf = ws.get("xnf", (M, D), BF)
        stf = ws.get("stf", (M, 2), F32)
        ops.norm_fwd(xl, st.p("pixel_decoder.norm.weight"),
                     st.p("pixel_decoder.norm.bias") if self.kind == ops.NORM_LN else None, xnf, stf, M, D, self.eps, self.kind)
        t = ws.get("t", (M, 768), BF)
        ops.gemm_nt(xnf, self.pout.w, t, M=M, N=768, K=D, bias=self.pout.bias, epi=EPI_BF16)
        self._ctx = (ws, B, h, w, lat, xl, xnf, stf, rope, self.stack.last_saved)
        self._dropped = dropped
        return t

    def backward(self, dt: torch.Tensor):
        """dt bf16 [B*hw, 768] -> d_lat bf16 [B*hw, 64]; parameter grads accumulate into store.flat_g.
        Generator: yields "tail", then ("block", i) per block; returns d_lat."""
        st = self.store
        ws, B, h, w, lat, xl, xnf, stf, rope, stack_saved = self._ctx
        M, D = B * h * w, self.D
        d_xnf = ws.get("b.d_xnf", (M, D), BF)
        linear_bwd(ws, "pout", self.pout, dt, xnf, M, d_xnf)
        dx = ws.get("b.dxt", (M, D), F32)
        dx_b = ws.get("b.dxt_b", (M, D), BF)
        ops.norm_bwd(d_xnf, xl, st.p("pixel_decoder.norm.weight"), stf, None, dx, dx_b, st.g("pixel_decoder.norm.weight"),
                     st.g("pixel_decoder.norm.bias") if self.kind == ops.NORM_LN else None, M, D, self.kind,
                     dx_colsum=self.stack.w3_colsum_target(self.depth - 1))
        OVERLAP.join()
        yield "tail"
        if getattr(self, "_dropped", False):
            dx0 = yield from self.stack.backward_drop(ws, dx, [(B, h * w, rope)], 0, stack_saved)
            dx0_b = dx_b
            ops.cast_f32_bf16(dx0, dx0_b, M * D)
        else:
            dx0, dx0_b = yield from self.stack.backward(ws, dx, dx_b, B, h * w, rope, 0, stack_saved,
                                                        dy_colsum_done=self.stack.w3_colsum_target(self.depth - 1) is not None)
        d_lat = ws.get("b.d_lat", (M, self.cin), BF)
        linear_bwd(ws, "pin", self.pin, dx0_b, lat, M, d_lat)
        OVERLAP.join()
        return d_lat
