"""Thin torch-tensor front end over the C ABI (include/vtp_hip.h).  Tensors only provide device memory and the
current HIP stream; every op below is a hand-written gfx950 kernel in libvtp_hip.so."""
from __future__ import annotations

import torch

from . import _lib

EPI_BF16, EPI_F32, EPI_SWIGLU, EPI_GELU, EPI_F32_ATOMIC, EPI_F32_SLAB = 0, 1, 2, 3, 4, 5
EPI_QUICK_GELU = 8  # the GELU epilogue with x * sigmoid(1.702 x)
NORM_RMS, NORM_LN = 0, 1


def _s():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _lib_():
    return _lib.load()


def pad8(n: int) -> int:
    return (n + 7) // 8 * 8


def gemm_nt(a, b, c, *, M=None, N=None, K=None, lda=None, ldb=None, ldc=None, c2=None, ldc2=0, bias=None, gamma=None,
            resid=None, epi=EPI_BF16, a_remap=(0, 0), c_remap=(0, 0), splits=1, alpha=1.0):
    """c[M,N] = epi(alpha * a[M,K] @ b[N,K]^T).  a, b bf16 (K-contiguous rows)."""
    M = a.shape[0] if M is None else M
    K = a.shape[1] if K is None else K
    N = b.shape[0] if N is None else N
    lda = a.stride(0) if lda is None else lda
    ldb = b.stride(0) if ldb is None else ldb
    ldc = c.stride(0) if ldc is None else ldc
    if c2 is not None and not ldc2:
        ldc2 = c2.stride(0)
    rc = _lib_().vtp_gemm_nt(_p(a), lda, _p(b), ldb, _p(c), ldc, _p(c2), ldc2, _p(bias), _p(gamma), _p(resid), M, N, K,
                             epi, a_remap[0], a_remap[1], c_remap[0], c_remap[1], splits, alpha, _s())
    _lib.check(rc, "vtp_gemm_nt")


def gemm_qkv_rope(a, w, bias, c, M, N, K, rope_pos, rope_sin, rope_cos, rope_cols):
    """c bf16 [M, N] = a @ w^T + bias with apply_rope fused (rows with rope_pos >= 0, columns < rope_cols)."""
    _lib.check(_lib_().vtp_gemm_qkv_rope(_p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(c), c.stride(0), M, N, K,
                                         _p(rope_pos), _p(rope_sin), _p(rope_cos), rope_cols, _s()), "vtp_gemm_qkv_rope")


def norm_fwd(x, w, b, y, stats, M, D, eps, kind):
    _lib.check(_lib_().vtp_norm_fwd(_p(x), _p(w), _p(b), _p(y), _p(stats), M, D, eps, kind, _s()), "vtp_norm_fwd")


def norm_bwd(dy, x, w, stats, dres, dx, dxb, dw, db, M, D, kind, dx_colsum=None):
    _lib.check(_lib_().vtp_norm_bwd(_p(dy), _p(x), _p(w), _p(stats), _p(dres), _p(dx), _p(dxb), _p(dw), _p(db), _p(dx_colsum),
                                    M, D, kind, _s()), "vtp_norm_bwd")


def rope_qk(qkv, sin, cos, B, N, heads, prefix, inverse=False):
    _lib.check(_lib_().vtp_rope_qk(_p(qkv), _p(sin), _p(cos), B, N, heads, prefix, int(inverse), _s()), "vtp_rope_qk")


def attn_fwd(q, k, v, o, lse, B, N, heads, sb, sn, sbo, sno, scale, causal=False):
    _lib.check(_lib_().vtp_attn_fwd(_p(q), _p(k), _p(v), _p(o), _p(lse), B, N, heads, sb, sn, sbo, sno, scale, int(causal),
                                    _s()), "vtp_attn_fwd")


def attn_bwd(q, k, v, o, d_o, lse, delta, dq, dk, dv, B, N, heads, sb, sn, sbo, sno, scale, causal=False, rope=None,
             rope_prefix=0):
    """rope = (sin, cos): dq / dk come back as gradients w.r.t. the un-rotated q, k (inverse RoPE fused / appended)."""
    rs, rc = (rope[0], rope[1]) if rope is not None else (None, None)
    _lib.check(_lib_().vtp_attn_bwd(_p(q), _p(k), _p(v), _p(o), _p(d_o), _p(lse), _p(delta), _p(dq), _p(dk), _p(dv), _p(rs),
                                    _p(rc), rope_prefix, B, N, heads, sb, sn, sbo, sno, scale, int(causal), _s()), "vtp_attn_bwd")


def im2col16(img, patches, B, H, W):
    _lib.check(_lib_().vtp_im2col16(_p(img), _p(patches), B, H, W, _s()), "vtp_im2col16")


def im2col16_rows(img, rows, B, H, W, prefix=1):
    """patch p of image b -> row b * (hw + prefix) + prefix + p of rows bf16 [B * (hw + prefix), 768] (prefix rows untouched)"""
    _lib.check(_lib_().vtp_im2col16_rows(_p(img), _p(rows), B, H, W, prefix, _s()), "vtp_im2col16_rows")


def col2im16(dpatches, dimg, B, H, W):
    """d_img f32 [B,3,H,W] <- d_patches f32 [B*hw, 768] (inverse of im2col16: the PatchEmbed input gradient)"""
    _lib.check(_lib_().vtp_col2im16(_p(dpatches), _p(dimg), B, H, W, _s()), "vtp_col2im16")


def assemble_tokens(x, cls, mask_token, masks, B, N, D):
    _lib.check(_lib_().vtp_assemble_tokens(_p(x), _p(cls), _p(mask_token), _p(masks), B, N, D, _s()), "vtp_assemble_tokens")


def transpose_bf16(inp, ld_in, out, ld_out, R, C, colsum=None, swiglu_h=0, in_remap=(0, 0)):
    _lib.check(_lib_().vtp_transpose_bf16(_p(inp), ld_in, _p(out), ld_out, _p(colsum), swiglu_h, in_remap[0], in_remap[1],
                                          R, C, _s()), "vtp_transpose_bf16")


def strided_rowsum(inp, stride, out, B, D):
    _lib.check(_lib_().vtp_strided_rowsum(_p(inp), stride, _p(out), B, D, _s()), "vtp_strided_rowsum")


def token_rows_bwd(dx, dxb, masks, d_mask_token, d_cls_token, B, N, D):
    """backward of assemble_tokens: masked rows -> d_mask_token (masks may be None), row 0 of every image -> d_cls_token; both kinds of
    row zeroed in the bf16 copy"""
    _lib.check(_lib_().vtp_token_rows_bwd(_p(dx), _p(dxb), _p(masks), None if masks is None else _p(d_mask_token), _p(d_cls_token), B, N, D,
                                          _s()), "vtp_token_rows_bwd")


def mask_rows_bwd(dx, dxb, masks, d_mask_token, B, N, D):
    _lib.check(_lib_().vtp_mask_rows_bwd(_p(dx), _p(dxb), _p(masks), _p(d_mask_token), B, N, D, _s()), "vtp_mask_rows_bwd")


def cast_f32_bf16(inp, out, n):
    _lib.check(_lib_().vtp_cast_f32_bf16(_p(inp), _p(out), n, _s()), "vtp_cast_f32_bf16")


def cast_transpose_f32_bf16(inp, out, R, C):
    _lib.check(_lib_().vtp_cast_transpose_f32_bf16(_p(inp), _p(out), R, C, _s()), "vtp_cast_transpose_f32_bf16")


def prep_weights(descs, n, total_tiles):
    _lib.check(_lib_().vtp_prep_weights(_p(descs), n, total_tiles, _s()), "vtp_prep_weights")


def prep_weights_range(descs_run, n, tile_base, n_tiles):
    """descs_run: the descriptor table sliced at the run's first record (a view: rows [d0:d1] of the int64 [*, 8] table)"""
    _lib.check(_lib_().vtp_prep_weights_range(_p(descs_run), n, tile_base, n_tiles, _s()), "vtp_prep_weights_range")


def swiglu_bwd(dh, x12, dx12, M, H, db12=None):
    _lib.check(_lib_().vtp_swiglu_bwd(_p(dh), _p(x12), _p(dx12), _p(db12), M, H, _s()), "vtp_swiglu_bwd")


def gelu_bwd(dy, pre, dx, n, quick: bool = False):
    if quick:
        _lib.check(_lib_().vtp_quick_gelu_bwd(_p(dy), _p(pre), _p(dx), n, _s()), "vtp_quick_gelu_bwd")
    else:
        _lib.check(_lib_().vtp_gelu_bwd(_p(dy), _p(pre), _p(dx), n, _s()), "vtp_gelu_bwd")


def pixel_shuffle16(t, img, B, h, w):
    _lib.check(_lib_().vtp_pixel_shuffle16(_p(t), _p(img), B, h, w, _s()), "vtp_pixel_shuffle16")


def pixel_unshuffle16(d_img, dt, B, h, w):
    _lib.check(_lib_().vtp_pixel_unshuffle16(_p(d_img), _p(dt), B, h, w, _s()), "vtp_pixel_unshuffle16")


def l1_loss_fwd_bwd(t, target, dt, loss_sum, B, h, w, gscale):
    _lib.check(_lib_().vtp_l1_loss_fwd_bwd(_p(t), _p(target), _p(dt), _p(loss_sum), B, h, w, gscale, _s()),
               "vtp_l1_loss_fwd_bwd")


def adamw(p, g, m, v, p_bf16, n, lr, beta1, beta2, eps, wd, step, grad_scale=1.0):
    _lib.check(_lib_().vtp_adamw(_p(p), _p(g), _p(m), _p(v), _p(p_bf16), n, lr, beta1, beta2, eps, wd, step, grad_scale,
                                 _s()), "vtp_adamw")


def adamw_dev(p, g, m, v, p_bf16, n, hyper, nodecay4=None):
    """nodecay4: uint8 [n / 4], non-zero = the four elements are exempt from weight decay"""
    if nodecay4 is None:
        _lib.check(_lib_().vtp_adamw_dev(_p(p), _p(g), _p(m), _p(v), _p(p_bf16), n, _p(hyper), _s()), "vtp_adamw_dev")
    else:
        _lib.check(_lib_().vtp_adamw_dev_masked(_p(p), _p(g), _p(m), _p(v), _p(p_bf16), _p(nodecay4), n, _p(hyper), _s()),
                   "vtp_adamw_dev_masked")


def gemm_tn(a, b, c, *, M, N, K, lda, ldb, ldc, ldc2=0, resid=None, epi=EPI_F32, a_remap=(0, 0), b_remap=(0, 0),
            c_remap=(0, 0), splits=1, a_colsum=None):
    """c[M,N] f32 = a[K,M]^T @ b[K,N]  (a, b bf16 row-major, K = token rows).  a_colsum (f32 [M], accumulated): column sums
    of a over the tokens (the bias gradient of the same linear layer)."""
    rc = _lib_().vtp_gemm_tn(_p(a), lda, _p(b), ldb, _p(c), ldc, ldc2, _p(resid), M, N, K, epi, a_remap[0], a_remap[1],
                             b_remap[0], b_remap[1], c_remap[0], c_remap[1], splits, _p(a_colsum), _s())
    _lib.check(rc, "vtp_gemm_tn")


def wgrad_group_fits(Ktok: int, max_ld: int) -> bool:
    """the grouped launch stages its operands with 32-bit byte offsets (k row x leading dimension x 2 B): rows x widest leading
    dimension must stay below 4 GiB -- otherwise the caller takes the per-layer path, which has the ring-kernel fallback"""
    return int(Ktok) * int(max_ld) * 2 < (1 << 32)


def gemm_cus() -> int:
    """workgroup slots the one-workgroup-per-CU GEMM launches may use: 256, or VTP_GEMM_CUS (the library's persistent kernels read the
    same variable) when CUs are to be left to a long-running communication kernel (INTEGRATION.md "Running beside RCCL")"""
    import os
    try:
        v = int(os.environ.get("VTP_GEMM_CUS", "0") or 0)
    except ValueError:
        v = 0
    return v - v % 8 if 8 <= v < 256 else 256


def wgrad_group_splits(ntiles: int, Ktok: int):
    """split rule of a grouped launch: tiles x splits fills one round of the CUs (256, or VTP_GEMM_CUS), every K slice keeps >= 16
    k-tiles; returns (k rows per slice, effective slice count) -- the launcher (vtp_gemm_tn_grouped) rounds the same way"""
    s = max(1, min(gemm_cus() // ntiles, Ktok // 1024))
    ks = ((Ktok + s - 1) // s + 63) // 64 * 64
    return ks, (Ktok + ks - 1) // ks


def wgrad_group_kernel(ntiles: int, splits: int, Ktok: int) -> int:
    """which kernel runs a grouped weight-gradient launch: 0 = the 8-phase kernel; 1 = the one-wave-per-SIMD kernel (gemm4w_tn.hip) on the
    same tiles x slices geometry, when the token count allows its 8-row staging pieces and a K slice is long enough to amortise its
    pipeline fill.  VTP_GEMM4W_TN=0 / 1 forces the choice (same-box A/B of the step); a token count that is not a multiple of 8
    always takes the 8-phase kernel."""
    import os
    e = os.environ.get("VTP_GEMM4W_TN")
    if Ktok % 8 != 0 or e == "0":
        return 0
    if e == "1":
        return 1
    return 1 if Ktok // max(splits, 1) >= W4_TN_MIN_SLICE else 0


def wgrad_group_items(rows, Ktok: int, base_splits: int, cus: int = None):
    """work-item list of a grouped launch on the one-wave-per-SIMD kernel (vtp_gemm_tn_grouped_items).  rows: the GroupProblem records
    (WgradGroup.rows).  Every tile is cut into `base_splits` K ranges, except the tiles that also form a bias gradient -- the first tile
    column of a problem with a colsum target: their k loop carries 64 v_dot2 per k-tile beside its 64 MFMAs and measures 27 % slower
    (tools/wgrad_timeline.py: 470 vs 365 us at 34 144 rows) -- which get ONE MORE slice when the launch still fits one round of the CUs
    and the shorter slice keeps >= 16 k-tiles.  Returns (items [n, 8] int32 rows {tile, kbeg, kcount, nparts, part, 0, 0, 0}, slots).
    Order: per problem, the heavier-cut tiles first, slice-major inside a group -- the kernel deals contiguous chunks of the list to
    the XCDs, so the workgroups that stream the same K range of the same operand panels sit behind one L2."""
    cus = gemm_cus() if cus is None else cus

    def cuts(n):
        ks = ((Ktok + n - 1) // n + 63) // 64 * 64
        return [(b, min(Ktok, b + ks) - b) for b in range(0, Ktok, ks)]

    tiles = []  # (tile index, problem, has colsum work)
    for pi, r in enumerate(rows):
        gb, N, K, tile0 = r[3], r[7], r[8], r[11]
        tn = (K + 255) // 256  # the C of a problem is [N, K] (dW): "M" = N rows of dy's columns, "N" = K columns of x
        for lt in range(((N + 255) // 256) * tn):
            tiles.append((tile0 + lt, pi, gb != 0 and lt % tn == 0))
    ncs = sum(1 for t in tiles if t[2])
    more = ncs > 0 and len(tiles) * base_splits + ncs <= cus and Ktok // (base_splits + 1) >= 1024
    items, slots = [], 1
    for pi in range(len(rows)):
        for heavy in (True, False):
            grp = [t for t in tiles if t[1] == pi and t[2] == heavy]
            if not grp:
                continue
            cs = cuts(base_splits + 1 if (heavy and more) else base_splits)
            slots = max(slots, len(cs))
            for z, (kb, kc) in enumerate(cs):
                items += [[t[0], kb, kc, len(cs), z, 0, 0, 0] for t in grp]
    return items, slots


W4_TN_MIN_SLICE = 4096  # token rows per K slice from which the one-wave-per-SIMD kernel is taken (tools/wgrad_kernel_ab.py against the two-phase 8-phase kernel: x0.97 at 1232 .. 2048 rows, x1.01 at 4096, x1.06 at 8224, x1.08 .. 1.12 at 17072)


class WgradGroup:
    """The weight gradients of one transformer block as ONE launch (vtp_gemm_tn_grouped): problems dW_g[N_g, K_g] (+)= dy_g^T x_g
    over the same token rows.  add() the problems, finalize() once (the operand buffers are static: the device descriptor table
    is built a single time), launch() every step."""

    def __init__(self, Ktok: int):
        self.Ktok, self.rows, self.ntiles, self.keep = int(Ktok), [], 0, []
        self.table = self.part = self.ticket = None

    def add(self, dy, x, gw, gb, N: int, K: int, swiglu_h: int = 0, accumulate: bool = True):
        """dy bf16 [Ktok, N] (row stride dy.stride(0)), x bf16 [Ktok, K], gw f32 [N * K], gb f32 [N] or None (bias gradient =
        column sums of dy, fused).  swiglu_h: dy's columns are the 8|8-interleaved w1|w2 pre-activation gradients; the rows of gw
        (and gb) are w1's H rows followed by w2's."""
        # limits of BOTH grouped kernels (16-B staging pieces, clamped tail columns): checked here because the kernels read the records
        # from device memory and cannot refuse them -- e.g. ffn_layer="mlp" with an odd int(D * ratio) takes the per-layer path
        if N % 8 or K % 8 or dy.stride(0) % 8 or x.stride(0) % 8 or N < 8 or K < 8:
            raise ValueError(f"WgradGroup.add: N = {N}, K = {K} and the leading dimensions ({dy.stride(0)}, {x.stride(0)}) must be multiples "
                             "of 8 (use the per-layer weight-gradient path: linear_bwd without `defer`)")
        self.rows.append([dy.data_ptr(), x.data_ptr(), gw.data_ptr(), 0 if gb is None else gb.data_ptr(), dy.stride(0), x.stride(0), K,
                          N, K, -1 if swiglu_h else 0, swiglu_h, self.ntiles, int(accumulate), 0, 0, 0])
        self.ntiles += ((N + 255) // 256) * ((K + 255) // 256)
        self.keep += [dy, x, gw, gb]

    def finalize(self, device, scratch=None):
        """split factor: tiles x splits fills one round of the 256 CUs, every K slice keeps >= 16 k-tiles.  scratch: a dict shared by
        the groups that never run concurrently (one partial-sum / ticket buffer for all of them)."""
        import torch
        assert 1 <= len(self.rows) <= 8
        if not wgrad_group_fits(self.Ktok, max(max(r[4], r[5]) for r in self.rows)):
            raise ValueError("WgradGroup: token rows x leading dimension exceed the 32-bit staging offsets of vtp_gemm_tn_grouped; "
                             "use the per-layer weight-gradient path (linear_bwd without `defer`)")
        _, self.splits = wgrad_group_splits(self.ntiles, self.Ktok)
        self.table = torch.tensor(self.rows, dtype=torch.int64, device=device)
        self.kernel = wgrad_group_kernel(self.ntiles, self.splits, self.Ktok)
        self.items, self.slots = None, self.splits
        import os
        if self.kernel == 1 and os.environ.get("VTP_WGRAD_ITEMS", "1") not in ("0", "false", "off"):
            # uneven cut: the tiles with bias-gradient work get one slice more (wgrad_group_items); VTP_WGRAD_ITEMS=0: uniform geometry
            items, self.slots = wgrad_group_items(self.rows, self.Ktok, self.splits)
            self.nitems = len(items)
            self.items = torch.tensor(items, dtype=torch.int32, device=device)
        if self.slots > 1:
            scratch = {} if scratch is None else scratch
            need = self.ntiles * self.slots * 65536
            if scratch.get("part") is None or scratch["part"].numel() < need:
                scratch["part"] = torch.empty(need, dtype=torch.float32, device=device)
            if scratch.get("ticket") is None or scratch["ticket"].numel() < self.ntiles:
                scratch["ticket"] = torch.zeros(max(self.ntiles, 256), dtype=torch.int32, device=device)
            self.part, self.ticket = scratch["part"], scratch["ticket"]
        return self

    def launch(self, kernel=None):
        """kernel: 0 = the 8-phase kernel, 1 = the one-wave-per-SIMD kernel (gemm4w_tn.hip), None = the measured choice made by finalize()
        (ops.wgrad_group_kernel)"""
        if kernel is None:
            kernel = self.kernel
        if kernel == 1 and self.items is not None:
            _lib.check(_lib_().vtp_gemm_tn_grouped_items(_p(self.table), len(self.rows), self.ntiles, self.Ktok, _p(self.items), self.nitems,
                                                         self.slots, _p(self.part), _p(self.ticket), _s()), "vtp_gemm_tn_grouped_items")
            return
        _lib.check(_lib_().vtp_gemm_tn_grouped_k(_p(self.table), len(self.rows), self.ntiles, self.Ktok, self.splits, _p(self.part),
                                                  _p(self.ticket), kernel, _s()), "vtp_gemm_tn_grouped")


def colsum_bf16(inp, ld, out, R, C, swiglu_h=0, in_remap=(0, 0)):
    _lib.check(_lib_().vtp_colsum_bf16(_p(inp), ld, _p(out), swiglu_h, in_remap[0], in_remap[1], R, C, _s()), "vtp_colsum_bf16")


def colsum_bf16_rows(inp, ld, out, n_rows_dev, R_max, C):
    _lib.check(_lib_().vtp_colsum_bf16_rows(_p(inp), ld, _p(out), _p(n_rows_dev), R_max, C, _s()), "vtp_colsum_bf16_rows")


def gather_image_rows(src, idx, dst, dst_bf16, n_img, N, D, scale=1.0):
    _lib.check(_lib_().vtp_gather_image_rows(_p(src), _p(idx), _p(dst), _p(dst_bf16), n_img, N, D, scale, _s()), "vtp_gather_image_rows")


def scatter_image_rows(src, idx, dst, n_img, N, D, alpha=1.0, accumulate=True):
    _lib.check(_lib_().vtp_scatter_image_rows(_p(src), _p(idx), _p(dst), n_img, N, D, alpha, int(accumulate), _s()),
               "vtp_scatter_image_rows")


def layerscale_wgrad(G, W, bias, colsum, gamma, dW, db, dgamma, N, K):
    _lib.check(_lib_().vtp_layerscale_wgrad(_p(G), _p(W), _p(bias), _p(colsum), _p(gamma), _p(dW), _p(db), _p(dgamma), N, K, _s()),
               "vtp_layerscale_wgrad")


def scaled_transpose(W, gamma, dstT, N, K):
    _lib.check(_lib_().vtp_scaled_transpose(_p(W), _p(gamma), _p(dstT), N, K, _s()), "vtp_scaled_transpose")


def gemm_splits(K, splits):
    return _lib_().vtp_gemm_splits(K, splits)


def gemm_tn_splits(M, N, K):
    return _lib_().vtp_gemm_tn_splits(M, N, K)


def reduce_slabs(slabs, stride, S, dst, n, accumulate=True):
    _lib.check(_lib_().vtp_reduce_slabs(_p(slabs), stride, S, _p(dst), n, int(accumulate), _s()), "vtp_reduce_slabs")


def ema(t, s, n, momentum):
    _lib.check(_lib_().vtp_ema(_p(t), _p(s), n, momentum, _s()), "vtp_ema")


def embed_tokens(ids, table, pos, x, eot, B, T, D):
    _lib.check(_lib_().vtp_embed_tokens(_p(ids), _p(table), _p(pos), _p(x), _p(eot), B, T, D, _s()), "vtp_embed_tokens")


def embed_tokens_bwd(ids, dx, d_table, d_pos, B, T, D):
    _lib.check(_lib_().vtp_embed_tokens_bwd(_p(ids), _p(dx), _p(d_table), _p(d_pos), B, T, D, _s()), "vtp_embed_tokens_bwd")


def gather_rows(x, idx, out, B, T, D):
    _lib.check(_lib_().vtp_gather_rows(_p(x), _p(idx), _p(out), B, T, D, _s()), "vtp_gather_rows")


def scatter_rows(dy, idx, dx, dxb, B, T, D):
    _lib.check(_lib_().vtp_scatter_rows(_p(dy), _p(idx), _p(dx), _p(dxb), B, T, D, _s()), "vtp_scatter_rows")


def l2norm_fwd(x, y, inv, B, D, eps=1e-12):
    _lib.check(_lib_().vtp_l2norm_fwd(_p(x), _p(y), _p(inv), B, D, eps, _s()), "vtp_l2norm_fwd")


def l2norm_bwd(dy, y, inv, dx, B, D):
    _lib.check(_lib_().vtp_l2norm_bwd(_p(dy), _p(y), _p(inv), _p(dx), B, D, _s()), "vtp_l2norm_bwd")


def clip_loss(img_l, txt_l, img_all, txt_all, logit_scale, Bl, Bg, D, label_offset, loss_sum, d_img_l, d_txt_l, d_img_all,
              d_txt_all, d_logit_scale, scratch):
    _lib.check(_lib_().vtp_clip_loss(_p(img_l), _p(txt_l), _p(img_all), _p(txt_all), _p(logit_scale), Bl, Bg, D, label_offset,
                                     _p(loss_sum), _p(d_img_l), _p(d_txt_l), _p(d_img_all), _p(d_txt_all), _p(d_logit_scale),
                                     _p(scratch), _s()), "vtp_clip_loss")


def qk_norm_fwd(qkv, wq, wk, out, inv, M, D, eps=1e-5):
    _lib.check(_lib_().vtp_qk_norm_fwd(_p(qkv), _p(wq), _p(wk), _p(out), _p(inv), M, D, eps, _s()), "vtp_qk_norm_fwd")


def qk_norm_bwd(dqkv, qkv, inv, wq, wk, dwq, dwk, M, D):
    _lib.check(_lib_().vtp_qk_norm_bwd(_p(dqkv), _p(qkv), _p(inv), _p(wq), _p(wk), _p(dwq), _p(dwk), M, D, _s()), "vtp_qk_norm_bwd")


def gemm_dgrad_swiglu(dy, wT, x12, dx12, M, H, K):
    """dx12[M, 2H] = SwiGLU'(x12) (.) (dy[M, K] @ wT[H, K]^T): the w3 dgrad with swiglu_bwd in its epilogue"""
    _lib.check(_lib_().vtp_gemm_dgrad_swiglu(_p(dy), dy.stride(0), _p(wT), wT.stride(0), _p(x12), x12.stride(0), _p(dx12), dx12.stride(0),
                                             M, H, K, _s()), "vtp_gemm_dgrad_swiglu")


def clip_logits(a, b, logit_scale, out, M, N, D):
    """out f32 [M, N] = exp(logit_scale) * a[M, D] @ b[N, D]^T  (the logits of modeling_vtp.py:326-329)"""
    _lib.check(_lib_().vtp_clip_logits(_p(a), _p(b), _p(logit_scale), _p(out), M, N, D, _s()), "vtp_clip_logits")


def siglip_loss(img_l, txt_all, logit_scale, logit_bias, Bl, Bg, D, label_offset, loss_sum, d_img_l, d_txt_all, d_logit_scale,
                d_logit_bias, scratch):
    """SigLIP (pairwise sigmoid) loss of the local images against all (gathered) texts: loss, feature gradients (local image rows
    / gathered text columns -- reduce-scatter the latter across ranks), d log-scale, d bias.  weight = 1 / B_local."""
    L, s = _lib_(), _s()
    _lib.check(L.vtp_clip_logits(_p(img_l), _p(txt_all), _p(logit_scale), _p(scratch), Bl, Bg, D, s), "vtp_clip_logits")
    _lib.check(L.vtp_siglip_pairs(_p(scratch), _p(logit_bias), Bl, Bg, label_offset, 1.0 / Bl, _p(loss_sum), _p(d_logit_scale),
                                  _p(d_logit_bias), s), "vtp_siglip_pairs")
    _lib.check(L.vtp_clip_grad_rows(_p(scratch), _p(txt_all), _p(logit_scale), _p(d_img_l), Bl, Bg, D, 0, s), "vtp_clip_grad_rows")
    _lib.check(L.vtp_clip_grad_cols(_p(scratch), _p(img_l), _p(logit_scale), _p(d_txt_all), Bl, Bg, D, 0, s), "vtp_clip_grad_cols")


def koleo(xn, nn_scratch, d_xn, loss_sum, B, D, weight, eps=1e-8):
    _lib.check(_lib_().vtp_koleo(_p(xn), _p(nn_scratch), _p(d_xn), _p(loss_sum), B, D, weight, eps, _s()), "vtp_koleo")


def sinkhorn_knopp(logits, inv_temp, probs, u, v, scratch, T, K, count, n_iters=3, phase=-1, count_dev=None, n_rows_dev=None):
    _lib.check(_lib_().vtp_sinkhorn_knopp(_p(logits), inv_temp, _p(probs), _p(u), _p(v), _p(scratch), T, K, float(count), _p(count_dev),
                                          _p(n_rows_dev), n_iters, phase, _s()), "vtp_sinkhorn_knopp")


def gather_token_rows(src, idx, dst, T, D):
    _lib.check(_lib_().vtp_gather_token_rows(_p(src), _p(idx), _p(dst), T, D, _s()), "vtp_gather_token_rows")


def scatter_token_rows(d_dst, idx, d_src, T, D):
    _lib.check(_lib_().vtp_scatter_token_rows(_p(d_dst), _p(idx), _p(d_src), T, D, _s()), "vtp_scatter_token_rows")


def weight_norm_prep(v, g, weff, weffT, inv_norm, K, C):
    _lib.check(_lib_().vtp_weight_norm_prep(_p(v), _p(g), _p(weff), _p(weffT), _p(inv_norm), K, C, _s()), "vtp_weight_norm_prep")


def weight_norm_bwd(dW, v, g, inv_norm, dv, dg, K, C):
    _lib.check(_lib_().vtp_weight_norm_bwd(_p(dW), _p(v), _p(g), _p(inv_norm), _p(dv), _p(dg), K, C, _s()), "vtp_weight_norm_bwd")


def softmax_center(logits, center, inv_temp, probs, T, K):
    """inv_temp: float, or a device f32 tensor (read by the kernel: graph-replay safe schedules)"""
    if isinstance(inv_temp, torch.Tensor):
        _lib.check(_lib_().vtp_softmax_center_dev(_p(logits), _p(center), _p(inv_temp), _p(probs), T, K, _s()), "vtp_softmax_center_dev")
    else:
        _lib.check(_lib_().vtp_softmax_center(_p(logits), _p(center), inv_temp, _p(probs), T, K, _s()), "vtp_softmax_center")


def dino_ce(s_logits, t_probs, t0, t1, w, inv_temp, loss_sum, d_logits, T, K):
    _lib.check(_lib_().vtp_dino_ce(_p(s_logits), _p(t_probs), _p(t0), _p(t1), _p(w), inv_temp, _p(loss_sum), _p(d_logits), T, K,
                                   _s()), "vtp_dino_ce")


def center_ema(center, col_sum, inv_count, momentum, K, count=None):
    _lib.check(_lib_().vtp_center_ema(_p(center), _p(col_sum), inv_count, _p(count), momentum, K, _s()), "vtp_center_ema")


def ema_dev(t, s, n, momentum_dev):
    _lib.check(_lib_().vtp_ema_dev(_p(t), _p(s), n, _p(momentum_dev), _s()), "vtp_ema_dev")


def adamw_ema_dev(p, g, m, v, teacher, n, hyper, nodecay4=None):
    """AdamW over one gradient bucket with the EMA update of the teacher's copy of the same elements fused in (teacher: f32 [n] or
    None; momentum = hyper[9]) -- the per-bucket optimizer lane of VTPTrainer"""
    _lib.check(_lib_().vtp_adamw_ema_dev(_p(p), _p(g), _p(m), _p(v), _p(teacher), _p(nodecay4), n, _p(hyper), _s()), "vtp_adamw_ema_dev")


# ---- LPIPS (vtp_amd/csrc/lpips.hip, conv mode of gemm.hip) ---------------------------------------------------------------
def _f3(v):
    import ctypes
    return (ctypes.c_float * 3)(*[float(x) for x in v])


def conv3x3(x, w, bias, y, NB, H, W, Cin, Cout, taps=9, mode=0, relu_mask=None):
    _lib.check(_lib_().vtp_conv3x3(_p(x), _p(w), _p(bias), _p(y), _p(relu_mask), NB, H, W, Cin, Cout, taps, mode, _s()), "vtp_conv3x3")


def lpips_unfold3(tok, img, out, n, H, W, shift, scale):
    _lib.check(_lib_().vtp_lpips_unfold3(_p(tok), _p(img), _p(out), n, H, W, _f3(shift), _f3(scale), _s()), "vtp_lpips_unfold3")


def lpips_fold3_bwd(dA, dt, n, H, W, scale):
    _lib.check(_lib_().vtp_lpips_fold3_bwd(_p(dA), _p(dt), n, H, W, _f3(scale), _s()), "vtp_lpips_fold3_bwd")


def maxpool2_fwd(x, y, NB, H, W, C):
    _lib.check(_lib_().vtp_maxpool2_fwd(_p(x), _p(y), NB, H, W, C, _s()), "vtp_maxpool2_fwd")


def maxpool2_bwd(Y, dP, tap, dY, NB, H, W, C):
    _lib.check(_lib_().vtp_maxpool2_bwd(_p(Y), _p(dP), _p(tap), _p(dY), NB, H, W, C, _s()), "vtp_maxpool2_bwd")


def lpips_tap(f0, f1, w, val, df0, n, H, W, C, gscale):
    _lib.check(_lib_().vtp_lpips_tap(_p(f0), _p(f1), _p(w), _p(val), _p(df0), n, H, W, C, float(gscale), _s()), "vtp_lpips_tap")


def _f3(vals):
    import ctypes
    return (ctypes.c_float * 3)(*[float(v) for v in vals])


def u8_to_images(u8_nhwc, img_nchw, mean, std, flip=False):
    """ToTensor + Normalize (+ horizontal flip): uint8 [B,H,W,3] -> f32 [B,3,H,W]  (vtp_tokenizer.py:74-81)"""
    B, H, W, _ = u8_nhwc.shape
    m, s = _f3(mean), _f3(std)
    _lib.check(_lib_().vtp_u8_to_images(_p(u8_nhwc), _p(img_nchw), B, H, W, m, s, int(bool(flip)), _s()), "vtp_u8_to_images")


def images_to_u8(img_nchw, u8_nhwc, sub, div):
    """Normalize(sub, div) -> *255 -> clamp -> uint8 -> NHWC  (vtp_tokenizer.py:105-111)"""
    B, _, H, W = img_nchw.shape
    a, d = _f3(sub), _f3(div)
    _lib.check(_lib_().vtp_images_to_u8(_p(img_nchw), _p(u8_nhwc), B, H, W, a, d, _s()), "vtp_images_to_u8")


def latent_channel_stats(latents, sums):
    B, C = latents.shape[:2]
    hw = latents[0, 0].numel()
    _lib.check(_lib_().vtp_latent_channel_stats(_p(latents), _p(sums), B, C, hw, _s()), "vtp_latent_channel_stats")


# ---- fp8 (e4m3) forward path: BASELINE config 5
E4M3_MAX = 448.0


def quantize_e4m3(src, dst, scale):
    """dst uint8 [n] = e4m3(clamp(src * scale)); src bf16 or f32; scale: float or a device f32 tensor"""
    n = src.numel()
    sd, sf = (scale, 0.0) if isinstance(scale, torch.Tensor) else (None, float(scale))
    _lib.check(_lib_().vtp_quantize_e4m3(_p(src), int(src.dtype == torch.float32), _p(dst), n, _p(sd), sf, _s()), "vtp_quantize_e4m3")


def norm_fwd_e4m3(x, w, b, y8, q_scale, stats, M, D, eps, kind):
    _lib.check(_lib_().vtp_norm_fwd_e4m3(_p(x), _p(w), _p(b), _p(y8), _p(q_scale), _p(stats), M, D, eps, kind, _s()), "vtp_norm_fwd_e4m3")


def amax(src, out):
    _lib.check(_lib_().vtp_amax(_p(src), int(src.dtype == torch.float32), src.numel(), _p(out), _s()), "vtp_amax")


def dequantize_e4m3(src, dst, inv_scale):
    _lib.check(_lib_().vtp_dequantize_e4m3(_p(src), _p(dst), src.numel(), float(inv_scale), _s()), "vtp_dequantize_e4m3")


def gemm_nt_fp8(a8, b8, c, *, M, N, K, alpha, lda=None, ldb=None, ldc=None, c2=None, ldc2=0, bias=None, resid=None, epi=EPI_BF16,
                rope=None):
    """c[M,N] = alpha * a8[M,K] @ b8[N,K]^T (+ bias, + resid | SwiGLU); a8 / b8 uint8 tensors holding e4m3;
    rope = (rope_pos, rope_sin, rope_cos, rope_cols): apply_rope fused into the bf16 epilogue"""
    lda = K if lda is None else lda
    ldb = K if ldb is None else ldb
    ldc = c.stride(0) if ldc is None else ldc
    rp = (None, None, None, 0) if rope is None else rope
    _lib.check(_lib_().vtp_gemm_nt_fp8(_p(a8), lda, _p(b8), ldb, _p(c), ldc, _p(c2), ldc2, _p(bias), _p(resid), M, N, K, epi, float(alpha),
                                       _p(rp[0]), _p(rp[1]), _p(rp[2]), rp[3], _s()), "vtp_gemm_nt_fp8")
