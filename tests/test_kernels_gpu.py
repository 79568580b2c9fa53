"""Per-kernel parity tests (MI355X).  Every test calls the HIP kernels through the C ABI (vtp_amd.ops -> ctypes ->
libvtp_hip.so) and compares with an fp32 PyTorch statement of the same op evaluated on the SAME bf16-rounded inputs.
Tolerance protocol: |err| <= 1e-3 * max|ref| + one bf16 ulp of the output (2^-8 relative) for bf16
outputs; integer / copy kernels are bit-exact."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from vtp_amd import _lib
    _lib.load()  # loud failure if the HIP library is missing


def ops():
    from vtp_amd import ops as o
    return o


def bf(x):
    return x.to(torch.bfloat16)


def check(out, ref, name, bf16_out=True, scale=1e-3):
    out = out.float()
    ref = ref.float()
    err = (out - ref).abs()
    tol = scale * ref.abs().max() + (2.0 ** -7 * ref.abs() if bf16_out else 1e-6 * ref.abs())  # bf16 ulp <= 2^-7 rel
    bad = (err > tol)
    worst = float((err - tol).max())
    print(f"[{name}] max|err|={float(err.max()):.3e} max|ref|={float(ref.abs().max()):.3e} relF={float(err.norm() / (ref.norm() + 1e-30)):.3e}")
    assert not torch.isnan(out).any(), f"{name}: NaN in output"
    assert not bad.any(), f"{name}: {int(bad.sum())} / {bad.numel()} elements out of tolerance (worst excess {worst:.3e})"


# ----------------------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (8224, 2304, 768), (257, 384, 128), (129, 344, 128), (300, 128, 344),
                                   (1000, 64, 768), (64, 768, 64)])
def test_gemm_bias_bf16(M, N, K):
    o = ops()
    g = torch.Generator(device=DEV).manual_seed(M * 7 + N * 3 + K)
    a = bf(torch.randn(M, K, device=DEV, generator=g))
    b = bf(torch.randn(N, K, device=DEV, generator=g) * 0.1)
    bias = torch.randn(N, device=DEV, generator=g)
    # asymmetric, transpose-detecting: a has a strong per-row ramp
    a = bf(a.float() + torch.linspace(-1, 1, M, device=DEV)[:, None])
    c = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    o.gemm_nt(a, b, c, bias=bias, epi=o.EPI_BF16)
    ref = a.float() @ b.float().T + bias
    check(c, ref, f"gemm_bf16 {M}x{N}x{K}")


def test_gemm_f32_residual_gamma_and_remaps():
    o = ops()
    B, hw, D, K = 3, 36, 128, 768
    N_tok = hw + 1
    g = torch.Generator(device=DEV).manual_seed(1)
    a = bf(torch.randn(B * hw, K, device=DEV, generator=g))
    w = bf(torch.randn(D, K, device=DEV, generator=g) * 0.05)
    bias = torch.randn(D, device=DEV, generator=g)
    gamma = torch.rand(D, device=DEV, generator=g) + 0.5
    x = torch.randn(B * N_tok, D, device=DEV, generator=g)
    x0 = x.clone()
    # write only the patch rows of a [B, 1+hw, D] stream: out = resid + gamma*(acc+bias), in place
    o.gemm_nt(a, w, x, M=B * hw, bias=bias, gamma=gamma, resid=x, epi=o.EPI_F32, c_remap=(hw, 1))
    ref = x0.clone().view(B, N_tok, D)
    ref[:, 1:] += ((a.float() @ w.float().T + bias) * gamma).view(B, hw, D)
    check(x, ref.view(-1, D), "gemm_f32 resid+gamma+c_remap", bf16_out=False, scale=1e-5)
    # A-side remap: read only the patch rows
    full = bf(torch.randn(B * N_tok, D, device=DEV, generator=g))
    w2 = bf(torch.randn(64, D, device=DEV, generator=g) * 0.1)
    out = torch.zeros(B * hw, 64, device=DEV)
    o.gemm_nt(full, w2, out, M=B * hw, epi=o.EPI_F32, a_remap=(hw, 1))
    ref2 = full.view(B, N_tok, D)[:, 1:].reshape(-1, D).float() @ w2.float().T
    check(out, ref2, "gemm_f32 a_remap", bf16_out=False, scale=1e-5)
    # strided A (cls rows only): lda = N_tok * D
    out3 = torch.zeros(B, 64, device=DEV)
    o.gemm_nt(full, w2, out3, M=B, lda=N_tok * D, epi=o.EPI_F32)
    check(out3, full.view(B, N_tok, D)[:, 0].float() @ w2.float().T, "gemm_f32 strided cls rows", bf16_out=False, scale=1e-5)


def interleave(w1, w2):
    H = w1.shape[0]
    out = torch.empty(2 * H, *w1.shape[1:], dtype=w1.dtype, device=w1.device)
    g = torch.arange(2 * H, device=w1.device)
    j = (g // 16) * 8 + (g % 8)
    sel = (g % 16) >= 8
    out[~sel] = w1[j[~sel]]
    out[sel] = w2[j[sel]]
    return out


@pytest.mark.parametrize("M,D,H", [(257, 128, 344), (1028, 768, 2048)])
def test_gemm_swiglu(M, D, H):
    o = ops()
    g = torch.Generator(device=DEV).manual_seed(5)
    x = bf(torch.randn(M, D, device=DEV, generator=g))
    w1 = bf(torch.randn(H, D, device=DEV, generator=g) * 0.08)
    w2 = bf(torch.randn(H, D, device=DEV, generator=g) * 0.08)
    b1 = torch.randn(H, device=DEV, generator=g) * 0.1
    b2 = torch.randn(H, device=DEV, generator=g) * 0.1
    w12, b12 = interleave(w1, w2).contiguous(), interleave(b1, b2).contiguous()
    hid = torch.full((M, H), float("nan"), dtype=torch.bfloat16, device=DEV)
    x12 = torch.full((M, 2 * H), float("nan"), dtype=torch.bfloat16, device=DEV)
    o.gemm_nt(x, w12, hid, N=2 * H, c2=x12, bias=b12, epi=o.EPI_SWIGLU)
    x1 = bf(x.float() @ w1.float().T + b1)
    x2 = bf(x.float() @ w2.float().T + b2)
    ref = bf(F.silu(x1.float())).float() * x2.float()
    check(hid, ref, f"swiglu hidden {M}x{D}x{H}", scale=4e-3)  # one bf16 ulp of x1/x2 propagates
    check(x12, interleave(x1.T.contiguous(), x2.T.contiguous()).T, "swiglu x12 (interleaved pre-activations)")


def test_gemm_gelu_and_atomic_splitk():
    o = ops()
    g = torch.Generator(device=DEV).manual_seed(9)
    M, N, K = 500, 512, 128
    a = bf(torch.randn(M, K, device=DEV, generator=g))
    w = bf(torch.randn(N, K, device=DEV, generator=g) * 0.1)
    bias = torch.randn(N, device=DEV, generator=g) * 0.1
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    o.gemm_nt(a, w, out, c2=pre, bias=bias, epi=o.EPI_GELU)
    p = bf(a.float() @ w.float().T + bias)
    check(pre, p, "gelu pre")
    check(out, F.gelu(p.float()), "gelu out", scale=4e-3)
    # split-K atomic accumulate (wgrad shape): C[N1,N2] += A[N1,Kbig] B[N2,Kbig]^T with a K tail
    N1, N2, Kb = 344, 128, 8224
    A = bf(torch.randn(N1, Kb, device=DEV, generator=g))
    Bm = bf(torch.randn(N2, Kb, device=DEV, generator=g))
    C = torch.ones(N1, N2, device=DEV)
    o.gemm_nt(A, Bm, C, epi=o.EPI_F32_ATOMIC, splits=7)
    check(C, 1.0 + A.float() @ Bm.float().T, "atomic split-K", bf16_out=False, scale=2e-5)


# ----------------------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("kind,D", [(0, 128), (0, 384), (0, 768), (1, 768), (1, 1024), (1, 128)])
def test_norm_fwd_bwd(kind, D):
    o = ops()
    M = 515
    g = torch.Generator(device=DEV).manual_seed(D + kind)
    x = torch.randn(M, D, device=DEV, generator=g) * 2 + 0.3
    w = torch.rand(D, device=DEV, generator=g) + 0.5
    b = torch.randn(D, device=DEV, generator=g) * 0.1 if kind else None
    eps = 1e-5 if kind == 0 else 1e-6
    y = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
    st = torch.empty(M, 2, device=DEV)
    o.norm_fwd(x, w, b, y, st, M, D, eps, kind)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if kind else None
    if kind == 0:
        ref = xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + eps) * wr
    else:
        ref = F.layer_norm(xr, (D,), wr, br, eps)
    check(y, ref.detach(), f"norm_fwd kind={kind} D={D}")
    dy = bf(torch.randn(M, D, device=DEV, generator=g))
    dres = torch.randn(M, D, device=DEV, generator=g)
    ref.backward(dy.float())
    dx = torch.empty(M, D, device=DEV)
    dxb = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
    dw = torch.zeros(D, device=DEV)
    db = torch.zeros(D, device=DEV) if kind else None
    dxs = torch.full((D,), 3.0, device=DEV)  # accumulated into: column sums of the bf16 output (the next linear's bias grad)
    o.norm_bwd(dy, x, w, st, dres, dx, dxb, dw, db, M, D, kind, dx_colsum=dxs)
    cs = dxb.float().sum(0) + 3.0
    assert float((dxs - cs).abs().max()) <= 1e-4 * float(dxb.float().abs().sum(0).max()) + 1e-4, "norm_bwd dx_colsum"
    check(dx, xr.grad + dres, "norm_bwd dx", bf16_out=False, scale=2e-5)
    check(dxb, bf(xr.grad + dres), "norm_bwd dx bf16")
    check(dw, wr.grad, "norm_bwd dw", bf16_out=False, scale=1e-4)
    if kind:
        check(db, br.grad, "norm_bwd db", bf16_out=False, scale=1e-4)


# ----------------------------------------------------------------------------------------------------------- RoPE
def test_rope_bit_exact_vs_oracle():
    from oracle import vtp_oracle as O
    o = ops()
    B, heads, h, w = 2, 3, 5, 7
    N, D = h * w + 1, heads * 64
    g = torch.Generator().manual_seed(3)
    qkv = bf(torch.randn(B * N, 3 * D, generator=g))
    sin, cos = O.rope_table(h, w, O.rope_periods(64))
    q, k, v = qkv.view(B, N, 3, heads, 64).unbind(2)
    qr, kr = O.apply_rope(q.transpose(1, 2), k.transpose(1, 2), sin, cos)
    ref = torch.stack([qr.transpose(1, 2), kr.transpose(1, 2), v], dim=2).reshape(B * N, 3 * D)
    dev = qkv.to(DEV).clone()
    o.rope_qk(dev, sin.to(DEV), cos.to(DEV), B, N, heads, 1)
    assert torch.equal(dev.cpu().view(torch.int16), ref.contiguous().view(torch.int16)), "RoPE must be bit-exact vs eager bf16"
    # inverse = transpose of the rotation: <R x, y> == <x, R^T y> up to bf16 rounding
    x = bf(torch.randn(B * N, 3 * D, generator=g)).to(DEV)
    y = bf(torch.randn(B * N, 3 * D, generator=g)).to(DEV)
    rx, rty = x.clone(), y.clone()
    o.rope_qk(rx, sin.to(DEV), cos.to(DEV), B, N, heads, 1)
    o.rope_qk(rty, sin.to(DEV), cos.to(DEV), B, N, heads, 1, inverse=True)
    lhs = (rx.float()[:, :2 * D] * y.float()[:, :2 * D]).sum()
    rhs = (x.float()[:, :2 * D] * rty.float()[:, :2 * D]).sum()
    assert abs(float(lhs - rhs)) < 2e-2 * float(lhs.abs() + 10), (float(lhs), float(rhs))


# ----------------------------------------------------------------------------------------------------------- attention
def _attn_ref(q, k, v, causal):
    # q,k,v: [B,N,h,64] bf16 -> fp32 math
    qf, kf, vf = (t.float().transpose(1, 2) for t in (q, k, v))
    return F.scaled_dot_product_attention(qf, kf, vf, is_causal=causal).transpose(1, 2)


def _sdpa_bf16_errors(q, k, v, d_o, causal, ref_grads):
    """(relF, max|err|) of the bf16 execution of the reference op (F.scaled_dot_product_attention + autograd, attention.py:124)
    against the fp32 gradients, per SDPA backend; returns the largest per tensor.  Test infrastructure only."""
    from torch.nn.attention import SDPBackend, sdpa_kernel
    worst = {}
    ran = []
    for be in (SDPBackend.FLASH_ATTENTION, SDPBackend.EFFICIENT_ATTENTION, SDPBackend.MATH):
        try:
            qb, kb, vb = (t.detach().clone().transpose(1, 2).requires_grad_(True) for t in (q, k, v))  # [B,h,N,64] bf16
            with sdpa_kernel([be]):
                ob = F.scaled_dot_product_attention(qb, kb, vb, is_causal=causal)
            ob.backward(d_o.transpose(1, 2).to(ob.dtype))
        except RuntimeError:
            continue
        ran.append(be.name)
        for nm, gb, r in zip(("dq", "dk", "dv"), (qb.grad, kb.grad, vb.grad), ref_grads):
            d = gb.transpose(1, 2).float() - r
            e = (float(d.norm() / r.norm()), float(d.abs().max()))
            w = worst.get(nm, (0.0, 0.0))
            worst[nm] = (max(w[0], e[0]), max(w[1], e[1]))
    assert ran, "no SDPA backend ran under bf16"
    return worst


@pytest.mark.parametrize("B,N,heads,causal", [(2, 257, 3, False), (1, 256, 2, False), (3, 17, 2, False), (2, 77, 2, True),
                                              (1, 1025, 2, False), (2, 130, 1, True), (1, 64, 1, False),
                                              # the launches of the benchmarked list forward (VTP-B, 12 heads): local 96^2 crops
                                              # (N = 37) and the merged clean-image + global-crop segment (32 + 64 images, N = 257)
                                              (16, 37, 12, False), (96, 257, 12, False),
                                              # persistent backward kernels: two rows in the odd 9th block, several heads per
                                              # workgroup at 8 row blocks, the last block partly filled
                                              (3, 258, 2, False), (40, 256, 8, False), (2, 230, 2, False),
                                              # ... and its two-wave instantiation (33 .. 66 tokens)
                                              (4, 64, 2, False), (2, 66, 3, False), (5, 40, 2, False)])
def test_attention_fwd_bwd(B, N, heads, causal):
    o = ops()
    D = heads * 64
    g = torch.Generator(device=DEV).manual_seed(N + heads)
    qkv = bf(torch.randn(B * N, 3 * D, device=DEV, generator=g))
    # spike one key against one query so the online-softmax rescale path is exercised hard
    qkv[N // 2, :64] *= 6
    qkv[min(N - 1, 70), D:D + 64] = qkv[N // 2, :64]
    out = torch.full((B * N, D), float("nan"), dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B, heads, N, device=DEV)
    scale = 0.125
    o.attn_fwd(qkv, qkv[:, D:], qkv[:, 2 * D:], out, lse, B, N, heads, N * 3 * D, 3 * D, N * D, D, scale, causal)
    q, k, v = qkv.view(B, N, 3, heads, 64).unbind(2)
    qr, kr, vr = (t.float().detach().requires_grad_(True) for t in (q, k, v))
    ref = _attn_ref(qr, kr, vr, causal)
    check(out.view(B, N, heads, 64), ref.detach(), f"attn_fwd N={N} causal={causal}", scale=4e-3)
    s = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float()) * scale
    if causal:
        s = s.masked_fill(torch.ones(N, N, device=DEV, dtype=torch.bool).triu(1), float("-inf"))
    check(lse, torch.logsumexp(s, -1), "attn lse", bf16_out=False, scale=1e-4)
    d_o = bf(torch.randn(B * N, D, device=DEV, generator=g))
    ref.backward(d_o.view(B, N, heads, 64).float())
    dqkv = torch.full((B * N, 3 * D), float("nan"), dtype=torch.bfloat16, device=DEV)
    delta = torch.empty(B, heads, N, device=DEV)
    o.attn_bwd(qkv, qkv[:, D:], qkv[:, 2 * D:], out, d_o, lse, delta, dqkv, dqkv[:, D:], dqkv[:, 2 * D:], B, N, heads,
               N * 3 * D, 3 * D, N * D, D, scale, causal)
    dq, dk, dv = dqkv.view(B, N, 3, heads, 64).unbind(2)
    # Bar: E_ours <= slack x E_ref, where E_ref is the error of the SAME op under bf16 -- stock PyTorch-ROCm
    # F.scaled_dot_product_attention forward + autograd backward on the identical bf16 q, k, v, dO, every SDPA backend that
    # accepts the shape (flash / mem-efficient / math), the largest of them -- against the fp32 reference.  No absolute bounds.
    # dS = P (dP - delta) is rounded to bf16 before the dQ / dK MFMAs and delta is formed from the bf16 O: the queries that attend
    # almost only to the spiked key are a cancellation in ANY bf16 execution, and a handful of such rows carries most of the error.
    # Measured over the 15 shapes x 3 tensors (profiles/r04_parity.log): ours / reference = 1.00 to three digits in 36 of 45 cases
    # (same roundings at the same places), 0.95 .. 1.07 in 8, 1.27 in one (dq, N = 66: the rows in question see a differently
    # rounded O from OUR forward than the reference backward sees from ITS forward -- a few draws, not an aggregate).  Hence
    # slack 1.5 on the Frobenius error per (shape, tensor) and 2 on max|err| (an extreme-value statistic of one run); the
    # aggregated model-level comparisons (tests/test_parity_*_gpu.py) keep 1.25.
    e_ref = _sdpa_bf16_errors(q, k, v, d_o.view(B, N, heads, 64), causal, (qr.grad, kr.grad, vr.grad))
    for nm, a, r in (("dq", dq, qr.grad), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
        assert not torch.isnan(a.float()).any(), f"attn_bwd {nm}: NaN"
        d = a.float() - r
        eF, eM = float(d.norm() / r.norm()), float(d.abs().max())
        rF, rM = e_ref[nm]
        print(f"[attn_bwd {nm} N={N} B={B} h={heads} causal={causal}] relF ours={eF:.3e} ref(bf16 SDPA)={rF:.3e} ratio={eF / rF:.2f} | "
              f"max|err| ours={eM:.3e} ref={rM:.3e} ratio={eM / rM:.2f}")
        assert eF <= 1.5 * rF, f"attn_bwd {nm}: relF {eF:.3e} > 1.5 x E_ref {rF:.3e}"
        assert eM <= 2.0 * rM, f"attn_bwd {nm}: max|err| {eM:.3e} > 2 x E_ref {rM:.3e}"
    if not causal:
        # inverse RoPE fused into the backward (short sequences: in the kernels' stores; long: appended pass) is bit-identical
        # to running vtp_rope_qk(inverse) on the plain result; the cls row (prefix 1) stays un-rotated
        sin = bf(torch.randn(N - 1, 64, device=DEV, generator=g))
        cos = bf(torch.randn(N - 1, 64, device=DEV, generator=g))
        want = dqkv.clone()
        o.rope_qk(want, sin, cos, B, N, heads, 1, inverse=True)
        got = torch.full_like(dqkv, float("nan"))
        o.attn_bwd(qkv, qkv[:, D:], qkv[:, 2 * D:], out, d_o, lse, delta, got, got[:, D:], got[:, 2 * D:], B, N, heads,
                   N * 3 * D, 3 * D, N * D, D, scale, causal, rope=(sin, cos), rope_prefix=1)
        assert torch.equal(got, want), float((got.float() - want.float()).abs().max())


@pytest.mark.parametrize("B,N,heads,prefix", [(3, 256, 2, 0), (2, 257, 3, 1), (6, 37, 2, 1), (2, 64, 2, 0), (2, 1025, 2, 1)])
def test_attention_bwd_fused_inverse_rope_prefixes(B, N, heads, prefix):
    """the inverse RoPE in the stores of the (fused) backward kernels with and without a cls prefix (the pixel decoder has none):
    bit-identical to vtp_rope_qk(inverse) applied to the plain backward result"""
    o = ops()
    D = heads * 64
    g = torch.Generator(device=DEV).manual_seed(100 + N + prefix)
    qkv = bf(torch.randn(B * N, 3 * D, device=DEV, generator=g))
    out = torch.empty(B * N, D, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B, heads, N, device=DEV)
    o.attn_fwd(qkv, qkv[:, D:], qkv[:, 2 * D:], out, lse, B, N, heads, N * 3 * D, 3 * D, N * D, D, 0.125, False)
    d_o = bf(torch.randn(B * N, D, device=DEV, generator=g))
    delta = torch.empty(B, heads, N, device=DEV)
    plain = torch.full((B * N, 3 * D), float("nan"), dtype=torch.bfloat16, device=DEV)
    o.attn_bwd(qkv, qkv[:, D:], qkv[:, 2 * D:], out, d_o, lse, delta, plain, plain[:, D:], plain[:, 2 * D:], B, N, heads,
               N * 3 * D, 3 * D, N * D, D, 0.125, False)
    assert torch.isfinite(plain.float()).all()
    sin = bf(torch.randn(N - prefix, 64, device=DEV, generator=g))
    cos = bf(torch.randn(N - prefix, 64, device=DEV, generator=g))
    want = plain.clone()
    o.rope_qk(want, sin, cos, B, N, heads, prefix, inverse=True)
    got = torch.full_like(plain, float("nan"))
    o.attn_bwd(qkv, qkv[:, D:], qkv[:, 2 * D:], out, d_o, lse, delta, got, got[:, D:], got[:, 2 * D:], B, N, heads,
               N * 3 * D, 3 * D, N * D, D, 0.125, False, rope=(sin, cos), rope_prefix=prefix)
    assert torch.equal(got, want), float((got.float() - want.float()).abs().max())
    # a second launch gives the same bits (the odd block's partial sums are added in a fixed order)
    again = torch.full_like(plain, float("nan"))
    o.attn_bwd(qkv, qkv[:, D:], qkv[:, 2 * D:], out, d_o, lse, delta, again, again[:, D:], again[:, 2 * D:], B, N, heads,
               N * 3 * D, 3 * D, N * D, D, 0.125, False, rope=(sin, cos), rope_prefix=prefix)
    assert torch.equal(again, got)


# ----------------------------------------------------------------------------------------------------------- data movement
def test_im2col_pixelshuffle_l1():
    o = ops()
    B, H, W = 2, 64, 96
    h, w = H // 16, W // 16
    g = torch.Generator(device=DEV).manual_seed(2)
    img = torch.randn(B, 3, H, W, device=DEV, generator=g)
    patches = torch.empty(B * h * w, 768, dtype=torch.bfloat16, device=DEV)
    o.im2col16(img, patches, B, H, W)
    ref = F.unfold(img, 16, stride=16).transpose(1, 2).reshape(B * h * w, 768)  # K order (c, ky, kx)
    assert torch.equal(patches, bf(ref)), "im2col16 must be exact"
    t = bf(torch.randn(B * h * w, 768, device=DEV, generator=g))
    out = torch.empty(B, 3, H, W, device=DEV)
    o.pixel_shuffle16(t, out, B, h, w)
    ref_img = F.pixel_shuffle(t.float().view(B, h, w, 768).permute(0, 3, 1, 2), 16)
    assert torch.equal(out, ref_img), "pixel_shuffle16 must be exact"
    dt = torch.empty_like(t)
    loss = torch.zeros(1, device=DEV)
    gs = 1.0 / img.numel()
    o.l1_loss_fwd_bwd(t, img, dt, loss, B, h, w, gs)
    tr = t.float().clone().requires_grad_(True)
    l = (F.pixel_shuffle(tr.view(B, h, w, 768).permute(0, 3, 1, 2), 16) - img).abs().mean()
    l.backward()
    assert abs(float(loss) * gs - float(l)) < 1e-5 * float(l) + 1e-7
    check(dt, bf(tr.grad), "l1 grad (token-major)")


def test_transpose_colsum_and_remaps():
    o = ops()
    g = torch.Generator(device=DEV).manual_seed(4)
    R, C = 517, 352
    x = bf(torch.randn(R, C, device=DEV, generator=g))
    ld = (R + 7) // 8 * 8
    out = torch.full((C, ld), float("nan"), dtype=torch.bfloat16, device=DEV)
    cs = torch.ones(C, device=DEV)
    o.transpose_bf16(x, C, out, ld, R, C, colsum=cs)
    assert torch.equal(out[:, :R], x.T), "transpose must be exact"
    assert float(out[:, R:].float().abs().max()) == 0.0, "tail columns must be zero-filled"
    check(cs, 1 + x.float().sum(0), "colsum", bf16_out=False, scale=1e-5)
    # swiglu de-interleave of the column sums
    H = C // 2
    cs2 = torch.zeros(C, device=DEV)
    o.transpose_bf16(x, C, out, ld, R, C, colsum=cs2, swiglu_h=H)
    full = x.float().sum(0)
    gidx = torch.arange(C, device=DEV)
    dst = (gidx // 16) * 8 + (gidx % 8) + ((gidx % 16) >= 8) * H
    ref = torch.zeros(C, device=DEV)
    ref[dst] = full
    check(cs2, ref, "colsum swiglu remap", bf16_out=False, scale=1e-5)
    # input row remap (patch rows of a [B, 1+hw, D] stream)
    B, hw, D = 3, 20, 128
    full_t = bf(torch.randn(B * (hw + 1), D, device=DEV, generator=g))
    ld2 = (B * hw + 7) // 8 * 8
    out2 = torch.empty(D, ld2, dtype=torch.bfloat16, device=DEV)
    o.transpose_bf16(full_t, D, out2, ld2, B * hw, D, in_remap=(hw, 1))
    assert torch.equal(out2[:, :B * hw], full_t.view(B, hw + 1, D)[:, 1:].reshape(-1, D).T)


def test_prep_weights_casts_and_interleave():
    o = ops()
    g = torch.Generator(device=DEV).manual_seed(6)
    H, D = 344, 128
    w = torch.randn(200, 136, device=DEV, generator=g)
    w1 = torch.randn(H, D, device=DEV, generator=g)
    w2 = torch.randn(H, D, device=DEV, generator=g)
    b1 = torch.randn(H, device=DEV, generator=g)
    b2 = torch.randn(H, device=DEV, generator=g)
    wb = torch.empty(200, 136, dtype=torch.bfloat16, device=DEV)
    wT = torch.empty(136, 200, dtype=torch.bfloat16, device=DEV)
    w12 = torch.empty(2 * H, D, dtype=torch.bfloat16, device=DEV)
    w12T = torch.empty(D, 2 * H, dtype=torch.bfloat16, device=DEV)
    b12 = torch.empty(2 * H, device=DEV)
    rows, tiles = [], 0
    wo = torch.randn(203, 134, device=DEV, generator=g)  # odd shape: the scalar path of the kernel
    wob = torch.empty(203, 134, dtype=torch.bfloat16, device=DEV)
    woT = torch.empty(134, 203, dtype=torch.bfloat16, device=DEV)
    for src, src2, dst, dstT, R, C, mode in [(w, None, wb, wT, 200, 136, 0), (w1, w2, w12, w12T, 2 * H, D, 1),
                                             (b1, b2, b12, None, 2 * H, 1, 2), (wo, None, wob, woT, 203, 134, 0)]:
        rows.append([src.data_ptr(), src2.data_ptr() if src2 is not None else 0, dst.data_ptr(),
                     dstT.data_ptr() if dstT is not None else 0, R, C, mode, tiles])
        tiles += (R + 255) // 256 if mode == 2 else ((R + 63) // 64) * ((C + 63) // 64)
    desc = torch.tensor(rows, dtype=torch.int64, device=DEV)
    o.prep_weights(desc, len(rows), tiles)
    assert torch.equal(wb, bf(w)) and torch.equal(wT, bf(w).T)
    assert torch.equal(wob, bf(wo)) and torch.equal(woT, bf(wo).T)
    assert torch.equal(w12, bf(interleave(w1, w2))) and torch.equal(w12T, bf(interleave(w1, w2)).T)
    assert torch.equal(b12, interleave(b1, b2))
    out = torch.empty(1000, dtype=torch.bfloat16, device=DEV)
    src = torch.randn(1000, device=DEV, generator=g)
    o.cast_f32_bf16(src, out, 1000)
    assert torch.equal(out, bf(src))


def test_swiglu_gelu_bwd_adamw_ema_assemble():
    from oracle import vtp_oracle as O
    o = ops()
    g = torch.Generator(device=DEV).manual_seed(8)
    M, H = 130, 344
    x1 = bf(torch.randn(M, H, device=DEV, generator=g))
    x2 = bf(torch.randn(M, H, device=DEV, generator=g))
    dh = bf(torch.randn(M, H, device=DEV, generator=g))
    x12 = interleave(x1.T.contiguous(), x2.T.contiguous()).T.contiguous()
    dx12 = torch.empty_like(x12)
    db12 = torch.full((2 * H,), 2.0, device=DEV)
    o.swiglu_bwd(dh, x12, dx12, M, H, db12=db12)
    g_ = torch.arange(2 * H, device=DEV)
    dest = (g_ // 16) * 8 + g_ % 8 + torch.where((g_ % 16) >= 8, H, 0)   # interleaved column -> [b1 | b2] slot
    cs = torch.zeros(2 * H, device=DEV).index_add_(0, dest, dx12.float().sum(0)) + 2.0
    assert float((db12 - cs).abs().max()) <= 1e-4 * float(dx12.float().abs().sum(0).max()) + 1e-4, "swiglu_bwd bias grads"
    dx12_plain = torch.empty_like(x12)
    o.swiglu_bwd(dh, x12, dx12_plain, M, H)  # default kernel (no fused bias gradients): identical output
    assert torch.equal(dx12_plain, dx12)
    a, b = x1.float().requires_grad_(True), x2.float().requires_grad_(True)
    (F.silu(a) * b).backward(dh.float())
    ref = interleave(bf(a.grad).T.contiguous(), bf(b.grad).T.contiguous()).T
    check(dx12, ref, "swiglu_bwd", scale=6e-3)
    pre = bf(torch.randn(M * 8, device=DEV, generator=g))
    dy = bf(torch.randn(M * 8, device=DEV, generator=g))
    dx = torch.empty_like(pre)
    o.gelu_bwd(dy, pre, dx, M * 8)
    pr = pre.float().requires_grad_(True)
    F.gelu(pr).backward(dy.float())
    check(dx, pr.grad, "gelu_bwd", scale=2e-3)
    # AdamW vs the oracle's torch.optim.AdamW restatement
    n = 4096 + 8
    p = torch.randn(n, device=DEV, generator=g)
    m = torch.zeros(n, device=DEV)
    v = torch.zeros(n, device=DEV)
    pc, mc, vc = p.cpu().clone(), m.cpu().clone(), v.cpu().clone()
    for step in range(1, 4):
        gr = torch.randn(n, device=DEV, generator=g)
        o.adamw(p, gr, m, v, None, n, 1e-3, 0.9, 0.95, 1e-8, 0.05, step, 0.5)
        O.adamw_step(pc, gr.cpu() * 0.5, mc, vc, step, 1e-3, 0.9, 0.95, 1e-8, 0.05)
    check(p, pc.to(DEV), "adamw", bf16_out=False, scale=1e-5)
    t = torch.randn(n, device=DEV, generator=g)
    s = torch.randn(n, device=DEV, generator=g)
    ref_t = 0.99 * t + 0.01 * s
    o.ema(t, s, n, 0.99)
    check(t, ref_t, "ema", bf16_out=False, scale=1e-6)
    B, N, D = 3, 10, 128
    x = torch.randn(B * N, D, device=DEV, generator=g)
    cls = torch.randn(D, device=DEV, generator=g)
    mt = torch.randn(D, device=DEV, generator=g)
    masks = (torch.rand(B, N - 1, device=DEV, generator=g) < 0.4).to(torch.uint8)
    ref_x = x.clone().view(B, N, D)
    ref_x[:, 0] = cls
    ref_x[:, 1:][masks.bool()] = mt
    o.assemble_tokens(x, cls, mt, masks, B, N, D)
    assert torch.equal(x.view(B, N, D), ref_x)
    acc = torch.ones(D, device=DEV)
    o.strided_rowsum(x, N * D, acc, B, D)
    check(acc, 1 + x.view(B, N, D)[:, 0].sum(0), "strided_rowsum", bf16_out=False, scale=1e-6)


def test_c_abi_rejects_bad_arguments():
    o = ops()
    a = torch.zeros(16, 12, dtype=torch.bfloat16, device=DEV)  # K = 12 is not a multiple of 8
    c = torch.zeros(16, 16, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="multiples of 8"):
        o.gemm_nt(a, a, c)
    with pytest.raises(RuntimeError, match="prefix"):
        o.rope_qk(c, c, c, 1, 4, 1, 9)


# ----------------------------------------------------------------------------------------------------------- CLIP head
def test_text_glue_kernels():
    o = ops()
    g = torch.Generator(device=DEV).manual_seed(12)
    B, T, D, V = 5, 16, 128, 300
    ids = torch.randint(1, V - 1, (B, T), device=DEV, generator=g)
    eot_pos = torch.tensor([3, 15, 7, 1, 9], device=DEV)
    ids[torch.arange(B), eot_pos] = V - 1
    table = torch.randn(V, D, device=DEV, generator=g)
    pos = torch.randn(T, D, device=DEV, generator=g)
    x = torch.empty(B * T, D, device=DEV)
    eot = torch.empty(B, dtype=torch.int32, device=DEV)
    o.embed_tokens(ids, table, pos, x, eot, B, T, D)
    assert torch.equal(x.view(B, T, D), table[ids] + pos)
    assert torch.equal(eot.long(), ids.argmax(-1)) and torch.equal(eot.long(), eot_pos)
    pooled = torch.empty(B, D, device=DEV)
    o.gather_rows(x, eot, pooled, B, T, D)
    assert torch.equal(pooled, x.view(B, T, D)[torch.arange(B), eot_pos])
    dy = torch.randn(B, D, device=DEV, generator=g)
    dx = torch.full((B * T, D), float("nan"), device=DEV)
    dxb = torch.full((B * T, D), float("nan"), dtype=torch.bfloat16, device=DEV)
    o.scatter_rows(dy, eot, dx, dxb, B, T, D)
    ref = torch.zeros(B, T, D, device=DEV)
    ref[torch.arange(B), eot_pos] = dy
    assert torch.equal(dx.view(B, T, D), ref) and torch.equal(dxb.view(B, T, D), bf(ref))
    dxx = torch.randn(B * T, D, device=DEV, generator=g)
    d_table = torch.zeros(V, D, device=DEV)
    d_pos = torch.ones(T, D, device=DEV)
    o.embed_tokens_bwd(ids, dxx, d_table, d_pos, B, T, D)
    ref_t = torch.zeros(V, D, device=DEV).index_add_(0, ids.reshape(-1), dxx)
    check(d_table, ref_t, "embed bwd table", bf16_out=False, scale=1e-6)
    check(d_pos, 1 + dxx.view(B, T, D).sum(0), "embed bwd pos", bf16_out=False, scale=1e-6)
    xx = torch.randn(B, D, device=DEV, generator=g) * 3
    y = torch.empty_like(xx)
    inv = torch.empty(B, device=DEV)
    o.l2norm_fwd(xx, y, inv, B, D)
    xr = xx.clone().requires_grad_(True)
    yr = F.normalize(xr, dim=-1)
    check(y, yr.detach(), "l2norm fwd", bf16_out=False, scale=1e-6)
    gy = torch.randn(B, D, device=DEV, generator=g)
    yr.backward(gy)
    dxo = torch.empty_like(xx)
    o.l2norm_bwd(gy, y, inv, dxo, B, D)
    check(dxo, xr.grad, "l2norm bwd", bf16_out=False, scale=1e-5)


@pytest.mark.parametrize("world,Bl,D", [(1, 6, 128), (2, 4, 128), (4, 32, 768)])
def test_clip_loss_with_emulated_ranks(world, Bl, D):
    """Every 'rank' runs vtp_clip_loss on its local rows against the gathered features; the total feature gradient of
    rank r = d_local(r) + sum_r' d_all(r')[rows of r]  must equal autograd of sum_r' L_r' (DDP then averages the
    parameter gradients), and sum_r loss_r / world the OpenCLIP global loss."""
    o = ops()
    g = torch.Generator(device=DEV).manual_seed(world * 100 + Bl)
    Bg = world * Bl
    I = F.normalize(torch.randn(Bg, D, device=DEV, generator=g), dim=-1)
    T = F.normalize(torch.randn(Bg, D, device=DEV, generator=g) + 0.5 * I, dim=-1)
    ls = torch.tensor([2.0], device=DEV)
    Ir, Tr, lsr = I.clone().requires_grad_(True), T.clone().requires_grad_(True), ls.clone().requires_grad_(True)
    total = 0.0
    for r in range(world):
        sl = slice(r * Bl, (r + 1) * Bl)
        labels = torch.arange(r * Bl, (r + 1) * Bl, device=DEV)
        li = lsr.exp() * Ir[sl] @ Tr.T
        lt = lsr.exp() * Tr[sl] @ Ir.T
        total = total + 0.5 * (F.cross_entropy(li, labels) + F.cross_entropy(lt, labels))
    total.backward()
    dI = torch.zeros_like(I)
    dT = torch.zeros_like(T)
    dls = torch.zeros(1, device=DEV)
    loss = torch.zeros(1, device=DEV)
    for r in range(world):
        sl = slice(r * Bl, (r + 1) * Bl)
        d_il, d_tl = torch.empty(Bl, D, device=DEV), torch.empty(Bl, D, device=DEV)
        d_ia, d_ta = torch.empty(Bg, D, device=DEV), torch.empty(Bg, D, device=DEV)
        scratch = torch.empty(2 * Bl * Bg, device=DEV)
        o.clip_loss(I[sl].contiguous(), T[sl].contiguous(), I, T, ls, Bl, Bg, D, r * Bl, loss, d_il, d_tl, d_ia, d_ta, dls, scratch)
        dI[sl] += d_il
        dT[sl] += d_tl
        dI += d_ia
        dT += d_ta
    check(loss, total.detach().reshape(1), "clip loss", bf16_out=False, scale=1e-5)
    check(dI, Ir.grad, "clip dI", bf16_out=False, scale=2e-5)
    check(dT, Tr.grad, "clip dT", bf16_out=False, scale=2e-5)
    check(dls, lsr.grad, "clip d logit_scale", bf16_out=False, scale=2e-5)


# ----------------------------------------------------------------------------------------------------------- TN GEMM (wgrad)
@pytest.mark.parametrize("Mo,No,K,splits", [(128, 128, 64, 1), (768, 768, 8224, 11), (2304, 768, 8224, 4), (344, 128, 515, 1),
                                            (64, 768, 8192, 3), (128, 688, 1000, 2)])
def test_gemm_tn_matches_reference(Mo, No, K, splits):
    """C[Mo,No] = A[K,Mo]^T B[K,No] (weight gradient from untransposed activations) incl. token/column tails."""
    o = ops()
    g = torch.Generator(device=DEV).manual_seed(Mo + No + K)
    A = bf(torch.randn(K, Mo, device=DEV, generator=g) + torch.linspace(-1, 1, Mo, device=DEV)[None, :])
    Bm = bf(torch.randn(K, No, device=DEV, generator=g))
    ref = A.float().T @ Bm.float()
    S = o.gemm_splits(K, splits)
    if S == 1:
        C = torch.ones(Mo, No, device=DEV)
        o.gemm_tn(A, Bm, C, M=Mo, N=No, K=K, lda=Mo, ldb=No, ldc=No, resid=C, epi=o.EPI_F32)
        check(C, 1 + ref, f"gemm_tn {Mo}x{No}x{K}", bf16_out=False, scale=2e-5)
    else:
        slab = torch.full((S * Mo * No,), float("nan"), device=DEV)
        o.gemm_tn(A, Bm, slab, M=Mo, N=No, K=K, lda=Mo, ldb=No, ldc=No, ldc2=Mo * No // 4, epi=o.EPI_F32_SLAB, splits=S)
        C = torch.ones(Mo, No, device=DEV)
        o.reduce_slabs(slab, Mo * No, S, C, Mo * No, accumulate=True)
        check(C, 1 + ref, f"gemm_tn split {Mo}x{No}x{K} S={S}", bf16_out=False, scale=2e-5)


def test_gemm_tn_remaps_and_colsum():
    o = ops()
    g = torch.Generator(device=DEV).manual_seed(21)
    B, hw, D, H = 3, 36, 128, 176
    full = bf(torch.randn(B * (hw + 1), D, device=DEV, generator=g))      # token stream with cls rows
    dy = bf(torch.randn(B * hw, 2 * H, device=DEV, generator=g))          # compact, SwiGLU-interleaved columns
    C = torch.zeros(2 * H, D, device=DEV)
    o.gemm_tn(dy, full, C, M=2 * H, N=D, K=B * hw, lda=2 * H, ldb=D, ldc=D, resid=C, epi=o.EPI_F32, b_remap=(hw, 1),
              c_remap=(-1, H))
    patches = full.view(B, hw + 1, D)[:, 1:].reshape(-1, D).float()
    ref_i = dy.float().T @ patches                                        # rows in interleaved order
    gi = torch.arange(2 * H, device=DEV)
    dst = (gi // 16) * 8 + (gi % 8) + ((gi % 16) >= 8) * H
    ref = torch.zeros_like(ref_i)
    ref[dst] = ref_i
    check(C, ref, "gemm_tn b_remap + swiglu c_remap", bf16_out=False, scale=2e-5)
    cs = torch.ones(2 * H, device=DEV)
    o.colsum_bf16(dy, 2 * H, cs, B * hw, 2 * H, swiglu_h=H)
    refc = torch.zeros(2 * H, device=DEV)
    refc[dst] = dy.float().sum(0)
    check(cs, 1 + refc, "colsum swiglu", bf16_out=False, scale=1e-5)
    cs2 = torch.zeros(D, device=DEV)
    o.colsum_bf16(full, D, cs2, B * hw, D, in_remap=(hw, 1))
    check(cs2, patches.sum(0), "colsum in_remap", bf16_out=False, scale=1e-5)


# ----------------------------------------------------------------------------------------------------------- fused qkv + RoPE
@pytest.mark.parametrize("cfg", [-1, 0, 4, 5, 8])
def test_gemm_qkv_rope_bit_identical_to_gemm_then_rope(cfg):
    """vtp_gemm_qkv_rope (apply_rope in the GEMM epilogue; list forward: two resolutions + cls prefix in ONE launch) must be
    bit-identical to vtp_gemm_nt followed by vtp_rope_qk per segment -- for every tile configuration that can run it."""
    from vtp_amd import _lib
    from vtp_amd.engine import rope_tables
    o = ops()
    lib = _lib.load()
    heads, D = 4, 256
    segs = [(3, 16, 16), (5, 6, 6), (2, 16, 16)]  # (images, h, w): N = 1 + h*w tokens each
    per = (100.0 ** (2 * torch.arange(16, dtype=torch.bfloat16) / 32))
    g = torch.Generator(device=DEV).manual_seed(3)
    M = sum(b * (1 + h * w) for b, h, w in segs)
    xn = bf(torch.randn(M, D, device=DEV, generator=g))
    w = bf(torch.randn(3 * D, D, device=DEV, generator=g) * 0.1)
    bias = torch.randn(3 * D, device=DEV, generator=g)
    lib.vtp_set_gemm_tuning(cfg, 3)
    try:
        ref = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=DEV)
        o.gemm_nt(xn, w, ref, M=M, N=3 * D, K=D, bias=bias, epi=o.EPI_BF16)
        pos, ts, tc, r0, base = [], [], [], 0, 0
        for b, h, wd in segs:
            sin, cos = rope_tables(per, h, wd, torch.device(DEV))
            N = 1 + h * wd
            o.rope_qk(ref[r0:r0 + b * N], sin, cos, b, N, heads, 1)
            pos.append(torch.cat([torch.tensor([-1], dtype=torch.int32), torch.arange(h * wd, dtype=torch.int32) + base]).repeat(b))
            ts.append(sin)
            tc.append(cos)
            r0 += b * N
            base += h * wd
        out = torch.full((M, 3 * D), float("nan"), dtype=torch.bfloat16, device=DEV)
        o.gemm_qkv_rope(xn, w, bias, out, M, 3 * D, D, torch.cat(pos).to(DEV), torch.cat(ts).contiguous(), torch.cat(tc).contiguous(),
                        2 * D)
    finally:
        lib.vtp_set_gemm_tuning(-1, 3)
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16)), \
        f"cfg {cfg}: {int((out.view(torch.int16) != ref.view(torch.int16)).sum())} elements differ"


@pytest.mark.parametrize("M,H,D", [(8192, 2048, 768), (2500, 344 - 344 % 8, 128), (16448, 2048, 768), (130, 1024, 384)])
def test_gemm_dgrad_swiglu_fused_equals_two_kernels(M, H, D):
    """w3 dgrad with swiglu_bwd in its epilogue == vtp_gemm_nt (bf16 dh) followed by vtp_swiglu_bwd, bit for bit (the fused epilogue
    rounds dh to bf16 at the same point), on the 8-phase and on the ring tile configurations, with row / column tails"""
    from vtp_amd import ops
    torch.manual_seed(M + H)
    dy = (torch.randn(M, D, device=DEV) * 0.5).to(torch.bfloat16)
    wT = (torch.randn(H, D, device=DEV) * 0.05).to(torch.bfloat16)
    x12 = torch.randn(M, 2 * H, device=DEV).to(torch.bfloat16)
    dh = torch.empty(M, H, dtype=torch.bfloat16, device=DEV)
    ref = torch.empty(M, 2 * H, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(dy, wT, dh, M=M, N=H, K=D, epi=ops.EPI_BF16)
    ops.swiglu_bwd(dh, x12, ref, M, H)
    out = torch.full((M, 2 * H), 7.0, dtype=torch.bfloat16, device=DEV)
    ops.gemm_dgrad_swiglu(dy, wT, x12, out, M, H, D)
    assert torch.equal(out, ref)
