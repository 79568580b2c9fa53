"""Loss-head variants (SigLIP, KoLeo, Sinkhorn-Knopp) against fp64 restatements of the upstream
definitions (oracle/loss_oracle.py) -- values and gradients.  The reference ships no loss: parity is unpinned by construction."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-300))


@pytest.mark.parametrize("Bl,Bg,off,D", [(8, 8, 0, 128), (6, 24, 12, 768)])
def test_siglip_loss_and_gradients(Bl, Bg, off, D):
    from oracle import loss_oracle as LO
    from vtp_amd import ops as o
    g = torch.Generator().manual_seed(Bl + D)
    txt = torch.nn.functional.normalize(torch.randn(Bg, D, generator=g), dim=-1)
    img = torch.nn.functional.normalize(torch.randn(Bl, D, generator=g) + 0.5 * txt[off:off + Bl], dim=-1)
    ls, bias = torch.tensor(2.3), torch.tensor(-10.0)
    i64, t64, l64, b64 = (x.double().requires_grad_(True) for x in (img, txt, ls, bias))
    loss_ref = LO.siglip_loss(i64, t64, l64, b64, off)
    loss_ref.backward()
    loss = torch.zeros(1, device=DEV)
    d_img, d_txt = torch.empty(Bl, D, device=DEV), torch.empty(Bg, D, device=DEV)
    d_ls, d_b = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    scratch = torch.empty(Bl * Bg, device=DEV)
    o.siglip_loss(img.to(DEV), txt.to(DEV), ls.to(DEV).reshape(1), bias.to(DEV).reshape(1), Bl, Bg, D, off, loss, d_img, d_txt, d_ls, d_b,
                  scratch)
    print(f"siglip: loss {float(loss):.6f} ref {float(loss_ref):.6f}; d_img {rel(d_img, i64.grad):.2e} d_txt {rel(d_txt, t64.grad):.2e} "
          f"d_ls {rel(d_ls, l64.grad.reshape(1)):.2e} d_bias {rel(d_b, b64.grad.reshape(1)):.2e}")
    assert abs(float(loss) - float(loss_ref)) < 1e-5 * abs(float(loss_ref))
    assert rel(d_img, i64.grad) < 1e-5 and rel(d_txt, t64.grad) < 1e-5
    assert rel(d_ls, l64.grad.reshape(1)) < 1e-4 and rel(d_b, b64.grad.reshape(1)) < 1e-4


@pytest.mark.parametrize("B,D", [(16, 128), (64, 768)])
def test_koleo_loss_and_gradient(B, D):
    from oracle import loss_oracle as LO
    from vtp_amd import ops as o
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, D, generator=g)
    x64 = x.double().requires_grad_(True)
    loss_ref, nn_ref = LO.koleo_loss(x64)
    loss_ref.backward()
    xd = x.to(DEV)
    xn, inv = torch.empty_like(xd), torch.empty(B, device=DEV)
    o.l2norm_fwd(xd, xn, inv, B, D, 1e-8)
    nn = torch.empty(B, dtype=torch.int32, device=DEV)
    d_xn, loss = torch.zeros(B, D, device=DEV), torch.zeros(1, device=DEV)
    o.koleo(xn, nn, d_xn, loss, B, D, 1.0 / B)
    dx = torch.empty_like(xd)
    o.l2norm_bwd(d_xn, xn, inv, dx, B, D)
    assert torch.equal(nn.cpu().long(), nn_ref)
    print(f"koleo: loss {float(loss):.6f} ref {float(loss_ref):.6f} dx {rel(dx, x64.grad):.2e}")
    assert abs(float(loss) - float(loss_ref)) < 2e-5 * abs(float(loss_ref)) and rel(dx, x64.grad) < 2e-4


@pytest.mark.parametrize("T,K,pad", [(64, 4096, 0), (200, 65536, 40)])
def test_sinkhorn_knopp_targets(T, K, pad):
    """probs rows sum to 1, match the fp64 algorithm; padded rows (device-side row count) do not disturb the valid ones."""
    from oracle import loss_oracle as LO
    from vtp_amd import ops as o
    g = torch.Generator().manual_seed(K)
    logits = (torch.randn(T + pad, K, generator=g) * 0.8).to(torch.bfloat16)
    temp = 0.07
    ref = LO.sinkhorn_knopp(logits[:T].float(), temp, 3)
    lg = logits.to(DEV)
    probs = torch.empty(T + pad, K, dtype=torch.bfloat16, device=DEV)
    u, v = torch.empty(T + pad, device=DEV), torch.empty(K, device=DEV)
    scratch = torch.empty(8 + K + T + pad, device=DEV)
    if pad:
        n_rows = torch.tensor([T], dtype=torch.int32, device=DEV)
        cnt = torch.tensor([float(T)], device=DEV)
        o.sinkhorn_knopp(lg, 1.0 / temp, probs, u, v, scratch, T + pad, K, 0.0, 3, count_dev=cnt, n_rows_dev=n_rows)
    else:
        o.sinkhorn_knopp(lg, 1.0 / temp, probs, u, v, scratch, T, K, float(T), 3)
    p = probs[:T].double().cpu()
    e = rel(p, ref)
    print(f"sinkhorn-knopp T={T} K={K}: rel {e:.2e}; row sums in [{float(p.sum(1).min()):.4f}, {float(p.sum(1).max()):.4f}]")
    assert e < 8e-3  # bf16 output rounding (2^-9 relative per entry)
    assert float((p.sum(1) - 1).abs().max()) < 5e-3
