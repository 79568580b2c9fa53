"""The 256x256 8-phase GEMM main loop (vtp_amd/csrc/gemm8p.hip, tile configuration 8) against a plain fp32 torch reference on
bf16-rounded inputs: every epilogue, ragged M / N / K tails, row remaps, split-K, several tiles per workgroup (persistent
stream across tiles) and the transposed (weight-gradient) form.  Same tolerance as tests/test_kernels_gpu.py."""
import pytest
import torch
import torch.nn.functional as F

from test_kernels_gpu import DEV, bf, check, interleave, ops  # noqa: F401  (same helpers / tolerance)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _force_cfg8():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from vtp_amd import _lib
    lib = _lib.load()
    lib.vtp_set_gemm_tuning(8, 3)
    yield
    lib.vtp_set_gemm_tuning(-1, 3)


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (256, 256, 128), (512, 768, 192), (300, 344, 200), (8224, 2304, 768),
                                   (1000, 64, 768), (64, 768, 72), (70000, 256, 128), (2500, 4096, 256)])
def test_gemm8p_bias_bf16(M, N, K):
    o = ops()
    g = torch.Generator(device=DEV).manual_seed(M * 7 + N * 3 + K)
    a = bf(torch.randn(M, K, device=DEV, generator=g) + torch.linspace(-1, 1, M, device=DEV)[:, None])
    b = bf(torch.randn(N, K, device=DEV, generator=g) * 0.1)
    bias = torch.randn(N, device=DEV, generator=g)
    c = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    for rep in range(2):  # a second launch must give the same result (no state carried between launches)
        o.gemm_nt(a, b, c, bias=bias, epi=o.EPI_BF16)
    ref = a.float() @ b.float().T + bias
    check(c, ref, f"gemm8p bf16 {M}x{N}x{K}")


def test_gemm8p_f32_residual_gamma_and_remaps():
    o = ops()
    B, hw, D, K = 9, 256, 384, 768
    N_tok = hw + 1
    g = torch.Generator(device=DEV).manual_seed(1)
    a = bf(torch.randn(B * hw, K, device=DEV, generator=g))
    w = bf(torch.randn(D, K, device=DEV, generator=g) * 0.05)
    bias = torch.randn(D, device=DEV, generator=g)
    gamma = torch.rand(D, device=DEV, generator=g) + 0.5
    x = torch.randn(B * N_tok, D, device=DEV, generator=g)
    x0 = x.clone()
    o.gemm_nt(a, w, x, M=B * hw, bias=bias, gamma=gamma, resid=x, epi=o.EPI_F32, c_remap=(hw, 1))
    ref = x0.clone().view(B, N_tok, D)
    ref[:, 1:] += ((a.float() @ w.float().T + bias) * gamma).view(B, hw, D)
    check(x, ref.view(-1, D), "gemm8p f32 resid+gamma+c_remap", bf16_out=False, scale=1e-5)
    full = bf(torch.randn(B * N_tok, D, device=DEV, generator=g))
    w2 = bf(torch.randn(64, D, device=DEV, generator=g) * 0.1)
    out = torch.zeros(B * hw, 64, device=DEV)
    o.gemm_nt(full, w2, out, M=B * hw, epi=o.EPI_F32, a_remap=(hw, 1))
    ref2 = full.view(B, N_tok, D)[:, 1:].reshape(-1, D).float() @ w2.float().T
    check(out, ref2, "gemm8p f32 a_remap", bf16_out=False, scale=1e-5)
    out3 = torch.zeros(B, 64, device=DEV)
    o.gemm_nt(full, w2, out3, M=B, lda=N_tok * D, epi=o.EPI_F32)
    check(out3, full.view(B, N_tok, D)[:, 0].float() @ w2.float().T, "gemm8p f32 strided cls rows", bf16_out=False, scale=1e-5)


@pytest.mark.parametrize("M,D,H", [(257, 128, 344), (2056, 768, 2048)])
def test_gemm8p_swiglu(M, D, H):
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
    check(hid, ref, f"gemm8p swiglu hidden {M}x{D}x{H}", scale=4e-3)
    check(x12, interleave(x1.T.contiguous(), x2.T.contiguous()).T, "gemm8p swiglu x12")


def test_gemm8p_gelu_slab_and_atomic_splitk():
    o = ops()
    g = torch.Generator(device=DEV).manual_seed(9)
    M, N, K = 700, 512, 128
    a = bf(torch.randn(M, K, device=DEV, generator=g))
    w = bf(torch.randn(N, K, device=DEV, generator=g) * 0.1)
    bias = torch.randn(N, device=DEV, generator=g) * 0.1
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    o.gemm_nt(a, w, out, c2=pre, bias=bias, epi=o.EPI_GELU)
    p = bf(a.float() @ w.float().T + bias)
    check(pre, p, "gemm8p gelu pre")
    check(out, F.gelu(p.float()), "gemm8p gelu out", scale=4e-3)
    N1, N2, Kb = 344, 128, 8224
    A = bf(torch.randn(N1, Kb, device=DEV, generator=g))
    Bm = bf(torch.randn(N2, Kb, device=DEV, generator=g))
    C = torch.ones(N1, N2, device=DEV)
    o.gemm_nt(A, Bm, C, epi=o.EPI_F32_ATOMIC, splits=7)
    check(C, 1.0 + A.float() @ Bm.float().T, "gemm8p atomic split-K", bf16_out=False, scale=2e-5)
    S = o.gemm_splits(Kb, 5)
    slab = torch.full((S * N1 * N2,), float("nan"), device=DEV)
    o.gemm_nt(A, Bm, slab, M=N1, N=N2, K=Kb, ldc=N2, ldc2=N1 * N2 // 4, epi=o.EPI_F32_SLAB, splits=S)
    check(slab.view(S, -1).sum(0).view(N1, N2), A.float() @ Bm.float().T, "gemm8p slab split-K", bf16_out=False, scale=2e-5)


@pytest.mark.parametrize("Mo,No,K,splits", [(256, 256, 64, 1), (768, 768, 8224, 11), (2304, 768, 8224, 4), (344, 128, 515, 1),
                                            (64, 768, 8192, 3), (128, 688, 1000, 2), (4096, 768, 34144, 5)])
def test_gemm8p_tn_matches_reference(Mo, No, K, splits):
    o = ops()
    g = torch.Generator(device=DEV).manual_seed(Mo + No + K)
    A = bf(torch.randn(K, Mo, device=DEV, generator=g) + torch.linspace(-1, 1, Mo, device=DEV)[None, :])
    Bm = bf(torch.randn(K, No, device=DEV, generator=g))
    ref = A.float().T @ Bm.float()
    S = o.gemm_splits(K, splits)
    if S == 1:
        C = torch.ones(Mo, No, device=DEV)
        o.gemm_tn(A, Bm, C, M=Mo, N=No, K=K, lda=Mo, ldb=No, ldc=No, resid=C, epi=o.EPI_F32)
        check(C, 1 + ref, f"gemm8p_tn {Mo}x{No}x{K}", bf16_out=False, scale=2e-5)
    else:
        slab = torch.full((S * Mo * No,), float("nan"), device=DEV)
        o.gemm_tn(A, Bm, slab, M=Mo, N=No, K=K, lda=Mo, ldb=No, ldc=No, ldc2=Mo * No // 4, epi=o.EPI_F32_SLAB, splits=S)
        C = torch.ones(Mo, No, device=DEV)
        o.reduce_slabs(slab, Mo * No, S, C, Mo * No, accumulate=True)
        check(C, 1 + ref, f"gemm8p_tn split {Mo}x{No}x{K} S={S}", bf16_out=False, scale=2e-5)


def test_gemm8p_tn_swiglu_c_remap():
    o = ops()
    g = torch.Generator(device=DEV).manual_seed(21)
    T, D, H = 1200, 128, 176
    x = bf(torch.randn(T, D, device=DEV, generator=g))
    dy = bf(torch.randn(T, 2 * H, device=DEV, generator=g))
    C = torch.zeros(2 * H, D, device=DEV)
    o.gemm_tn(dy, x, C, M=2 * H, N=D, K=T, lda=2 * H, ldb=D, ldc=D, resid=C, epi=o.EPI_F32, c_remap=(-1, H))
    ref_i = dy.float().T @ x.float()
    gi = torch.arange(2 * H, device=DEV)
    dst = (gi // 16) * 8 + (gi % 8) + ((gi % 16) >= 8) * H
    ref = torch.zeros_like(ref_i)
    ref[dst] = ref_i
    check(C, ref, "gemm8p_tn swiglu c_remap", bf16_out=False, scale=2e-5)


@pytest.mark.parametrize("Mo,No,K,splits,swiglu", [(768, 768, 8224, 9, False), (352, 128, 4100, 1, True), (4096, 768, 34144, 5, True),
                                                   (2304, 768, 34144, 9, False)])
def test_gemm8p_tn_fused_bias_gradient(Mo, No, K, splits, swiglu):
    """a_colsum: db[m] += sum_t dy[t, m] inside the weight-gradient GEMM (accumulating, SwiGLU row map, split-K, K tail)."""
    o = ops()
    g = torch.Generator(device=DEV).manual_seed(Mo + K)
    A = bf(torch.randn(K, Mo, device=DEV, generator=g) + 0.25)
    Bm = bf(torch.randn(K, No, device=DEV, generator=g))
    H = Mo // 2
    S = o.gemm_splits(K, splits)
    db = torch.ones(Mo, device=DEV)
    kw = dict(c_remap=(-1, H)) if swiglu else {}
    if S == 1:
        C = torch.zeros(Mo, No, device=DEV)
        o.gemm_tn(A, Bm, C, M=Mo, N=No, K=K, lda=Mo, ldb=No, ldc=No, resid=C, epi=o.EPI_F32, a_colsum=db, **kw)
    else:
        slab = torch.empty(S * Mo * No, device=DEV)
        o.gemm_tn(A, Bm, slab, M=Mo, N=No, K=K, lda=Mo, ldb=No, ldc=No, ldc2=Mo * No // 4, epi=o.EPI_F32_SLAB, splits=S, a_colsum=db,
                  **kw)
        C = slab.view(S, Mo, No).sum(0)
    ref_c = A.float().T @ Bm.float()
    ref_b = A.float().sum(0)
    if swiglu:
        gi = torch.arange(Mo, device=DEV)
        dst = (gi // 16) * 8 + (gi % 8) + ((gi % 16) >= 8) * H
        rc, rb = torch.zeros_like(ref_c), torch.zeros_like(ref_b)
        rc[dst], rb[dst] = ref_c, ref_b
        ref_c, ref_b = rc, rb
    check(C, ref_c, f"gemm8p_tn + colsum C {Mo}x{No}x{K}", bf16_out=False, scale=2e-5)
    check(db, 1 + ref_b, f"gemm8p_tn fused bias gradient {Mo}x{No}x{K}", bf16_out=False, scale=2e-5)


@pytest.mark.parametrize("Ktok,D,H", [(2134, 768, 2048), (514, 384, 1024), (34144, 768, 2048), (1100, 128, 344)])
def test_grouped_wgrad_matches_torch(Ktok, D, H):
    """ops.WgradGroup (vtp_gemm_tn_grouped): the four weight gradients of a block in ONE launch -- in-launch split-K combine by the
    last arriver (1, 2 or more slices), SwiGLU row de-interleave, fused bias-gradient column sums, C += and C = modes, and a second
    launch on the same scratch (tickets must be back at zero)."""
    o = ops()
    g = torch.Generator(device=DEV).manual_seed(Ktok + D)
    dqkv = bf(torch.randn(Ktok, 3 * D, device=DEV, generator=g))
    dmid = bf(torch.randn(Ktok, D, device=DEV, generator=g))
    dpre = bf(torch.randn(Ktok, 2 * H, device=DEV, generator=g))     # interleaved 8 | 8 (w1 | w2) columns
    dy = bf(torch.randn(Ktok, D, device=DEV, generator=g))
    xn1 = bf(torch.randn(Ktok, D, device=DEV, generator=g))
    att = bf(torch.randn(Ktok, D, device=DEV, generator=g))
    xn2 = bf(torch.randn(Ktok, D, device=DEV, generator=g))
    hid = bf(torch.randn(Ktok, H, device=DEV, generator=g))
    probs = [(dy, hid, D, H, 0, False), (dpre, xn2, 2 * H, D, H, True), (dmid, att, D, D, 0, False), (dqkv, xn1, 3 * D, D, 0, True)]
    gws = [torch.randn(N * K, device=DEV, generator=g) for _, _, N, K, _, _ in probs]
    gbs = [torch.randn(N, device=DEV, generator=g) if cs else None for _, _, N, _, _, cs in probs]
    gw0 = [t.clone() for t in gws]
    gb0 = [None if t is None else t.clone() for t in gbs]
    grp = o.WgradGroup(Ktok)
    for (a, x, N, K, sh, _), gw, gb in zip(probs, gws, gbs):
        grp.add(a, x, gw, gb, N, K, sh)
    scratch = {}
    grp.finalize(DEV, scratch)
    print(f"grouped wgrad Ktok={Ktok}: {grp.ntiles} tiles x {grp.splits} slices")
    grp.launch()
    grp.launch()  # accumulates twice
    torch.cuda.synchronize()
    if grp.splits > 1:
        assert int(scratch["ticket"].abs().sum()) == 0, "tickets must return to zero"
    for (a, x, N, K, sh, cs), gw, gb, w0, b0 in zip(probs, gws, gbs, gw0, gb0):
        ref = a.float().T @ x.float()                                    # [N, K]
        col = a.float().sum(0)
        if sh:  # de-interleave the GEMM's rows: 16-row groups = 8 rows of w1 | 8 rows of w2
            idx = torch.arange(N, device=DEV)
            dst = ((idx >> 4) << 3) + (idx & 7) + torch.where((idx & 8) != 0, sh, 0)
            r2, c2 = torch.empty_like(ref), torch.empty_like(col)
            r2[dst], c2[dst] = ref, col
            ref, col = r2, c2
        check(gw.view(N, K), w0.view(N, K) + 2 * ref, f"grouped dW N={N} K={K}", bf16_out=False, scale=1e-4)
        if cs:
            check(gb, b0 + 2 * col, f"grouped db N={N}", bf16_out=False, scale=1e-4)
    # overwrite mode + a group with a forced 3-way split through the C ABI's splits argument
    out = torch.full((D * H,), float("nan"), device=DEV)
    g1 = o.WgradGroup(Ktok)
    g1.add(dy, hid, out, None, D, H, 0, accumulate=False)
    g1.finalize(DEV, {})
    g1.splits = 3  # the launcher rounds the slices to whole k-tiles (the effective count may come out lower)
    g1.part = torch.empty(g1.ntiles * 3 * 65536, device=DEV)
    g1.ticket = torch.zeros(max(g1.ntiles, 256), dtype=torch.int32, device=DEV)
    g1.launch()
    check(out.view(D, H), dy.float().T @ hid.float(), "grouped dW overwrite", bf16_out=False, scale=1e-4)


@pytest.mark.parametrize("M,N,K,epi_name", [(8192, 768, 4096, "bf16"), (8100, 1024, 4096, "f32"), (4000, 1024, 4160, "gelu"),
                                            (12288, 512, 8192, "bf16")])
def test_in_launch_split_k_combine(M, N, K, epi_name):
    """few-tile / long-K shapes (text tower, pixel-decoder dgrads, DINO head): the 256 x 256 kernel cuts K into slices and the
    last-arriving slice of a tile combines them and runs the epilogue (heuristic path, no forced configuration); repeated launches
    must agree (tickets return to zero)"""
    from vtp_amd import _lib
    o = ops()
    _lib.load().vtp_set_gemm_tuning(-1, 3)  # the module fixture forces cfg 8 with no K split: back to the dispatcher's choice
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a = bf(torch.randn(M, K, device=DEV, generator=g))
    w = bf(torch.randn(N, K, device=DEV, generator=g) * 0.05)
    bias = torch.randn(N, device=DEV, generator=g)
    ref = a.float() @ w.float().T + bias
    for rep in range(3):
        if epi_name == "f32":
            x0 = torch.randn(M, N, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
            out = torch.full((M, N), float("nan"), device=DEV)
            o.gemm_nt(a, w, out, bias=bias, resid=x0, epi=o.EPI_F32)
            check(out, ref + x0, f"combine f32 {M}x{N}x{K}", bf16_out=False, scale=2e-5)
        elif epi_name == "bf16":
            out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            o.gemm_nt(a, w, out, bias=bias, epi=o.EPI_BF16)
            check(out, ref, f"combine bf16 {M}x{N}x{K}")
        else:
            out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            pre = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            o.gemm_nt(a, w, out, c2=pre, bias=bias, epi=o.EPI_GELU)
            check(pre, ref, f"combine gelu pre {M}x{N}x{K}")
            check(out, F.gelu(bf(ref).float()), f"combine gelu {M}x{N}x{K}", scale=4e-3)


def test_split_k_combine_run_to_run_bound():
    """The in-launch split-K combine adds the other slices' partial tiles to the registers of whichever slice arrives LAST, in slice
    order around it (csrc/gemm8p.hip): with TWO slices that is a commutative fp32 add -- every launch gives the same bits; with THREE
    the association depends on the arrival order, so weight gradients may differ from run to run by fp32 rounding of the partial
    sums.  Stated bound: |run_i - run_0| <= 4 * 2^-24 * (|dy|^T |x|) element-wise, i.e. a few fp32 ulps of the
    sum of absolute products -- never more than the error of ANY fixed summation order."""
    o = ops()
    Ktok, D, H = 34144, 768, 768
    g = torch.Generator(device=DEV).manual_seed(77)
    dy = bf(torch.randn(Ktok, D, device=DEV, generator=g))
    x = bf(torch.randn(Ktok, H, device=DEV, generator=g))
    bound = 4.0 * 2.0 ** -24 * (dy.float().abs().T @ x.float().abs())
    for forced, must_be_identical in ((2, True), (3, False)):
        outs = []
        for rep in range(8):
            out = torch.full((D * H,), float("nan"), device=DEV)
            grp = o.WgradGroup(Ktok)
            grp.add(dy, x, out, None, D, H, 0, accumulate=False)
            grp.finalize(DEV, {})
            grp.splits = forced
            grp.part = torch.empty(grp.ntiles * forced * 65536, device=DEV)
            grp.ticket = torch.zeros(max(grp.ntiles, 256), dtype=torch.int32, device=DEV)
            junk = torch.empty(32 << 20, device=DEV).normal_()  # perturb the arrival order between repetitions
            grp.launch()
            del junk
            outs.append(out.view(D, H).clone())
        check(outs[0], dy.float().T @ x.float(), f"grouped split-{forced}", bf16_out=False, scale=1e-4)
        diffs = [(t - outs[0]).abs() for t in outs[1:]]
        worst = max(float((d / bound).max()) for d in diffs)
        nbits = sum(int((d > 0).sum()) for d in diffs)
        print(f"split-K combine, {forced} slices: {nbits} elements differ over 7 repetitions, worst |diff| / bound = {worst:.3f}")
        assert all(bool((d <= bound).all()) for d in diffs), f"{forced} slices: run-to-run difference above the stated bound"
        if must_be_identical:
            assert nbits == 0, "two slices: the combine is a commutative add and must be bit-reproducible"
