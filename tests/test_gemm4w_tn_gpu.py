"""The grouped weight-gradient launch on the one-wave-per-SIMD kernel (vtp_amd/csrc/gemm4w_tn.hip + gemm4w_tn_ktile.inc, `kernel = 1` of
vtp_gemm_tn_grouped_k) -- against fp32 torch on the bf16 operands, and against the 8-phase kernel (`kernel = 0`): BIT FOR BIT where the
launch has one K slice (same k order per output element, same MFMA, same epilogue), within the run-to-run bound of the in-launch combine
where it has several (the last-arriving slice differs from run to run, in both kernels).  K tails inside a k-tile, odd k-tile counts
(padded with a zero k-tile), M / N tails (clamped staging columns), SwiGLU row de-interleave, fused bias-gradient column sums, C += and
C = modes, tickets back at zero."""
import pytest
import torch

from test_kernels_gpu import DEV, bf, check, ops  # noqa: F401  (same helpers / tolerance)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _problems(Ktok, D, H, g):
    dqkv = bf(torch.randn(Ktok, 3 * D, device=DEV, generator=g))
    dmid = bf(torch.randn(Ktok, D, device=DEV, generator=g))
    dpre = bf(torch.randn(Ktok, 2 * H, device=DEV, generator=g))  # interleaved 8 | 8 (w1 | w2) columns
    dy = bf(torch.randn(Ktok, D, device=DEV, generator=g))
    xn1, att, xn2 = (bf(torch.randn(Ktok, D, device=DEV, generator=g)) for _ in range(3))
    hid = bf(torch.randn(Ktok, H, device=DEV, generator=g))
    return [(dy, hid, D, H, 0, False), (dpre, xn2, 2 * H, D, H, True), (dmid, att, D, D, 0, False), (dqkv, xn1, 3 * D, D, 0, True)]


@pytest.mark.parametrize("Ktok,D,H", [(2136, 768, 2048), (512, 384, 1024), (34144, 768, 2048), (1096, 128, 344), (8192, 768, 2048),
                                      (2464, 768, 3072), (4104, 256, 688)])
def test_grouped_wgrad_one_wave_kernel(Ktok, D, H):
    o = ops()
    g = torch.Generator(device=DEV).manual_seed(Ktok + D)
    probs = _problems(Ktok, D, H, g)
    gw0 = [torch.randn(N * K, device=DEV, generator=g) for _, _, N, K, _, _ in probs]
    gb0 = [torch.randn(N, device=DEV, generator=g) if cs else None for _, _, N, _, _, cs in probs]
    res = {}
    for kernel in (1, 0):
        gws = [t.clone() for t in gw0]
        gbs = [None if t is None else t.clone() for t in gb0]
        grp = o.WgradGroup(Ktok)
        for (a, x, N, K, sh, _), gw, gb in zip(probs, gws, gbs):
            grp.add(a, x, gw, gb, N, K, sh)
        scratch = {}
        grp.finalize(DEV, scratch)
        grp.launch(kernel=kernel)
        grp.launch(kernel=kernel)  # accumulates twice
        torch.cuda.synchronize()
        if grp.splits > 1:
            assert int(scratch["ticket"].abs().sum()) == 0, "tickets must return to zero"
        res[kernel] = (gws, gbs, max(grp.splits, grp.slots), grp.ntiles)
    print(f"grouped wgrad Ktok={Ktok}: {res[1][3]} tiles x {res[1][2]} slices")
    for n, ((a, x, N, K, sh, cs), w0, b0) in enumerate(zip(probs, gw0, gb0)):
        ref = a.float().T @ x.float()  # [N, K]
        col = a.float().sum(0)
        if sh:  # de-interleave the GEMM's rows: 16-row groups = 8 rows of w1 | 8 rows of w2
            idx = torch.arange(N, device=DEV)
            dst = ((idx >> 4) << 3) + (idx & 7) + torch.where((idx & 8) != 0, sh, 0)
            r2, c2 = torch.empty_like(ref), torch.empty_like(col)
            r2[dst], c2[dst] = ref, col
            ref, col = r2, c2
        gw1, gw8 = res[1][0][n], res[0][0][n]
        check(gw1.view(N, K), w0.view(N, K) + 2 * ref, f"one-wave grouped dW N={N} K={K}", bf16_out=False, scale=1e-4)
        if res[1][2] == 1:
            assert torch.equal(gw1, gw8), f"problem {n}: {int((gw1 != gw8).sum())} elements differ from the 8-phase kernel"
        else:  # several slices: the sum order depends on which slice arrives last (in both kernels)
            assert float((gw1 - gw8).abs().max()) <= 2e-5 * float(gw8.abs().max())
        if cs:
            check(res[1][1][n], b0 + 2 * col, f"one-wave grouped db N={N}", bf16_out=False, scale=1e-4)


def test_grouped_wgrad_one_wave_overwrite_forced_splits_and_refusal():
    o = ops()
    Ktok, D, H = 4104, 768, 2048
    g = torch.Generator(device=DEV).manual_seed(7)
    dy = bf(torch.randn(Ktok, D, device=DEV, generator=g))
    hid = bf(torch.randn(Ktok, H, device=DEV, generator=g))
    outs = {}
    for kernel in (1, 0):
        out = torch.full((D * H,), float("nan"), device=DEV)
        g1 = o.WgradGroup(Ktok)
        g1.add(dy, hid, out, None, D, H, 0, accumulate=False)
        g1.finalize(DEV, {})
        g1.splits = 3  # the launcher rounds the slices to whole k-tiles
        g1.part = torch.empty(g1.ntiles * 3 * 65536, device=DEV)
        g1.ticket = torch.zeros(max(g1.ntiles, 256), dtype=torch.int32, device=DEV)
        g1.launch(kernel=kernel)
        outs[kernel] = out
    check(outs[1].view(D, H), dy.float().T @ hid.float(), "one-wave grouped dW overwrite, 3 slices", bf16_out=False, scale=1e-4)
    assert float((outs[1] - outs[0]).abs().max()) <= 2e-5 * float(outs[0].abs().max())
    # a token count that is not a multiple of 8 has no 8-row staging pieces: the measured choice is the 8-phase kernel, a forced 1 is refused
    assert o.wgrad_group_kernel(24, 2, 2134) == 0
    Kt = 2134
    a, x = bf(torch.randn(Kt, 256, device=DEV, generator=g)), bf(torch.randn(Kt, 256, device=DEV, generator=g))
    g2 = o.WgradGroup(Kt)
    g2.add(a, x, torch.zeros(256 * 256, device=DEV), None, 256, 256)
    g2.finalize(DEV, {})
    with pytest.raises(RuntimeError):
        g2.launch(kernel=1)
    g2.launch()
    check(g2.keep[2].view(256, 256), a.float().T @ x.float(), "fallback to the 8-phase kernel", bf16_out=False, scale=1e-4)
