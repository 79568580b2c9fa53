"""The one-wave-per-SIMD 256x256 GEMM with the hand-scheduled k loop (vtp_amd/csrc/gemm4w.hip + gemm4w_ktile.inc, tile configuration 10)
-- against a plain fp32 torch reference on bf16-rounded inputs, and BIT FOR BIT against the 8-phase 256x256 kernel (configuration 8):
both accumulate every output element over the k-steps in the same order with the same MFMA and share the epilogue code, so any
difference is a staging / synchronisation / hazard bug of the asm loop.  Ragged M tails (clamped staging), one- and two-k-tile
streams, several tiles per workgroup (the staging cursor jumps between output tiles inside the asm), every epilogue incl. fused
RoPE and SwiGLU backward.  Shapes the kernel does not take (N % 256, K % 128, A row remap) fall back to configuration 8 (still checked)."""
import pytest
import torch
import torch.nn.functional as F

from test_kernels_gpu import DEV, bf, check, interleave, ops  # noqa: F401  (same helpers / tolerance)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    yield
    from vtp_amd import _lib
    _lib.load().vtp_set_gemm_tuning(-1, 3)


def _cfg(c):
    from vtp_amd import _lib
    _lib.load().vtp_set_gemm_tuning(c, 3)


@pytest.mark.parametrize("M,N,K", [(128, 256, 128), (256, 256, 256), (512, 768, 384), (300, 512, 256), (100, 256, 128), (257, 256, 128),
                                   (8224, 2304, 768), (70000, 256, 128), (2500, 4096, 256), (34144, 768, 768), (2464, 768, 3072),
                                   (34144, 768, 2304), (256, 256, 192), (64, 768, 72), (1000, 64, 768)])  # (the last three: fallback)
def test_gemm4w_bias_bf16(M, N, K):
    o = ops()
    g = torch.Generator(device=DEV).manual_seed(M * 7 + N * 3 + K)
    a = bf(torch.randn(M, K, device=DEV, generator=g) + torch.linspace(-1, 1, M, device=DEV)[:, None])
    b = bf(torch.randn(N, K, device=DEV, generator=g) * 0.1)
    bias = torch.randn(N, device=DEV, generator=g)
    outs = {}
    for cfg in (10, 8):
        _cfg(cfg)
        c = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        for rep in range(2):  # a second launch must give the same result (no state carried between launches)
            o.gemm_nt(a, b, c, bias=bias, epi=o.EPI_BF16)
        outs[cfg] = c
    check(outs[10], a.float() @ b.float().T + bias, f"gemm4w bf16 {M}x{N}x{K}")
    for cx in (10,):
        assert torch.equal(outs[cx].view(torch.int16), outs[8].view(torch.int16)), \
            f"cfg {cx} vs 8: {int((outs[cx].view(torch.int16) != outs[8].view(torch.int16)).sum())} elements differ"


@pytest.mark.parametrize("cfg", [10])
def test_gemm4w_repeatable_under_load(cfg):
    """race screen: 30 launches of a many-tiles-per-workgroup shape interleaved with a memory-bound kernel; all results identical"""
    o = ops()
    _cfg(cfg)
    g = torch.Generator(device=DEV).manual_seed(0)
    M, N, K = 34144, 2304, 768
    a = bf(torch.randn(M, K, device=DEV, generator=g))
    b = bf(torch.randn(N, K, device=DEV, generator=g) * 0.1)
    junk = torch.empty(64 << 20, device=DEV)
    first = None
    for i in range(30):
        c = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        if i % 3 == 0:
            junk.normal_()
        o.gemm_nt(a, b, c, epi=o.EPI_BF16)
        if first is None:
            first = c
            check(c, a.float() @ b.float().T, "gemm4w big bf16")
        else:
            assert torch.equal(c.view(torch.int16), first.view(torch.int16)), f"launch {i} differs"


def test_gemm4w_f32_residual_gamma_and_remaps():
    o = ops()
    B, hw, D, K = 9, 256, 384, 768
    N_tok = hw + 1
    g = torch.Generator(device=DEV).manual_seed(1)
    a = bf(torch.randn(B * hw, K, device=DEV, generator=g))
    w = bf(torch.randn(D, K, device=DEV, generator=g) * 0.05)
    bias = torch.randn(D, device=DEV, generator=g)
    gamma = torch.rand(D, device=DEV, generator=g) + 0.5
    x0 = torch.randn(B * N_tok, D, device=DEV, generator=g)
    res = {}
    for cfg in (10, 8):
        _cfg(cfg)
        x = x0.clone()
        o.gemm_nt(a, w, x, M=B * hw, bias=bias, gamma=gamma, resid=x, epi=o.EPI_F32, c_remap=(hw, 1))
        res[cfg] = x
    ref = x0.clone().view(B, N_tok, D)
    ref[:, 1:] += ((a.float() @ w.float().T + bias) * gamma).view(B, hw, D)
    check(res[10], ref.view(-1, D), "gemm4w f32 resid+gamma+c_remap", bf16_out=False, scale=1e-5)
    assert torch.equal(res[10], res[8])
    _cfg(10)
    full = bf(torch.randn(B * N_tok, D, device=DEV, generator=g))
    w2 = bf(torch.randn(256, D, device=DEV, generator=g) * 0.1)
    out = torch.zeros(B * hw, 256, device=DEV)
    o.gemm_nt(full, w2, out, M=B * hw, epi=o.EPI_F32, a_remap=(hw, 1))
    ref2 = full.view(B, N_tok, D)[:, 1:].reshape(-1, D).float() @ w2.float().T
    check(out, ref2, "gemm4w f32 a_remap", bf16_out=False, scale=1e-5)


@pytest.mark.parametrize("M,D,H", [(257, 128, 344), (2056, 768, 2048), (8192, 768, 2048)])
def test_gemm4w_swiglu(M, D, H):
    o = ops()
    g = torch.Generator(device=DEV).manual_seed(5)
    x = bf(torch.randn(M, D, device=DEV, generator=g))
    w1 = bf(torch.randn(H, D, device=DEV, generator=g) * 0.08)
    w2 = bf(torch.randn(H, D, device=DEV, generator=g) * 0.08)
    b1 = torch.randn(H, device=DEV, generator=g) * 0.1
    b2 = torch.randn(H, device=DEV, generator=g) * 0.1
    w12, b12 = interleave(w1, w2).contiguous(), interleave(b1, b2).contiguous()
    res = {}
    for cfg in (10, 8):
        _cfg(cfg)
        hid = torch.full((M, H), float("nan"), dtype=torch.bfloat16, device=DEV)
        x12 = torch.full((M, 2 * H), float("nan"), dtype=torch.bfloat16, device=DEV)
        o.gemm_nt(x, w12, hid, N=2 * H, c2=x12, bias=b12, epi=o.EPI_SWIGLU)
        res[cfg] = (hid, x12)
    x1 = bf(x.float() @ w1.float().T + b1)
    x2 = bf(x.float() @ w2.float().T + b2)
    ref = bf(F.silu(x1.float())).float() * x2.float()
    check(res[10][0], ref, f"gemm4w swiglu hidden {M}x{D}x{H}", scale=4e-3)
    check(res[10][1], interleave(x1.T.contiguous(), x2.T.contiguous()).T, "gemm4w swiglu x12")
    for cx in (10,):
        assert torch.equal(res[cx][0].view(torch.int16), res[8][0].view(torch.int16)) and torch.equal(res[cx][1].view(torch.int16), res[8][1].view(torch.int16))


@pytest.mark.parametrize("cfg", [10])
def test_gemm4w_gelu(cfg):
    o = ops()
    _cfg(cfg)
    g = torch.Generator(device=DEV).manual_seed(9)
    M, N, K = 2464, 3072, 768
    a = bf(torch.randn(M, K, device=DEV, generator=g))
    w = bf(torch.randn(N, K, device=DEV, generator=g) * 0.1)
    bias = torch.randn(N, device=DEV, generator=g) * 0.1
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    o.gemm_nt(a, w, out, c2=pre, bias=bias, epi=o.EPI_GELU)
    p = bf(a.float() @ w.float().T + bias)
    check(pre, p, "gemm4w gelu pre")
    check(out, F.gelu(p.float()), "gemm4w gelu out", scale=4e-3)


@pytest.mark.parametrize("M,H,D", [(8192, 2048, 768), (2500, 344 - 344 % 8, 128), (16448, 2048, 768)])
def test_gemm4w_dgrad_swiglu_and_qkv_rope_equal_cfg8(M, H, D):
    """the fused extras of the bf16 store path (XMODE 1 / 2) on configuration 9 == configuration 8, bit for bit"""
    from vtp_amd.engine import rope_tables
    o = ops()
    torch.manual_seed(M + H)
    dy = (torch.randn(M, D, device=DEV) * 0.5).to(torch.bfloat16)
    wT = (torch.randn(H, D, device=DEV) * 0.05).to(torch.bfloat16)
    x12 = torch.randn(M, 2 * H, device=DEV).to(torch.bfloat16)
    outs = {}
    for cfg in (10, 8):
        _cfg(cfg)
        out = torch.full((M, 2 * H), 7.0, dtype=torch.bfloat16, device=DEV)
        o.gemm_dgrad_swiglu(dy, wT, x12, out, M, H, D)
        outs[cfg] = out
    assert torch.equal(outs[10].view(torch.int16), outs[8].view(torch.int16))
    # qkv + RoPE: 64-wide heads, cls prefix rows (rope_pos < 0)
    heads, Dm, hw = 4, 256, 256
    Nt = hw + 1
    Bq = max(1, min(16, M // Nt))
    Mq = Bq * Nt
    per = (100.0 ** (2 * torch.arange(16, dtype=torch.bfloat16) / 32))
    sin, cos = rope_tables(per, 16, 16, torch.device(DEV))
    pos = torch.cat([torch.tensor([-1], dtype=torch.int32), torch.arange(hw, dtype=torch.int32)]).repeat(Bq).to(DEV)
    xn = (torch.randn(Mq, Dm, device=DEV)).to(torch.bfloat16)
    w = (torch.randn(3 * Dm, Dm, device=DEV) * 0.1).to(torch.bfloat16)
    bias = torch.randn(3 * Dm, device=DEV)
    outs = {}
    for cfg in (10, 8):
        _cfg(cfg)
        out = torch.full((Mq, 3 * Dm), float("nan"), dtype=torch.bfloat16, device=DEV)
        o.gemm_qkv_rope(xn, w, bias, out, Mq, 3 * Dm, Dm, pos, sin.contiguous(), cos.contiguous(), 2 * Dm)
        outs[cfg] = out
    assert torch.equal(outs[10].view(torch.int16), outs[8].view(torch.int16))
