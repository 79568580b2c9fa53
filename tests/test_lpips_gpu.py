"""LPIPS on the gfx950 kernels: implicit-GEMM 3x3 convolutions, pool / head kernels and the whole
perceptual term against the oracle and the golden outputs of the real reference class (seeded weights: the pretrained
vgg.pth is a download, so parity is structural)."""
import os

import pytest
import torch
import torch.nn.functional as F
from safetensors.torch import load_file

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def _stack(x_nchw):
    """f32 NCHW -> (flat bf16 buffer with guards, view [NB*(H+2)*(W+2), C]) in the kernels' zero-bordered NHWC layout."""
    NB, C, H, W = x_nchw.shape
    rows, g = NB * (H + 2) * (W + 2), W + 3
    buf = torch.zeros((rows + 2 * g) * C, dtype=BF, device=DEV)
    v = buf[g * C:(g + rows) * C].view(NB, H + 2, W + 2, C)
    v[:, 1:-1, 1:-1, :] = x_nchw.permute(0, 2, 3, 1).to(BF)
    return buf, v.view(rows, C)


def _unstack(t, NB, H, W, C):
    return t.view(NB, H + 2, W + 2, C)[:, 1:-1, 1:-1, :].permute(0, 3, 1, 2).float()


@pytest.mark.parametrize("NB,H,W,Cin,Cout", [(2, 16, 16, 64, 64), (3, 8, 24, 64, 128), (1, 32, 16, 128, 256), (2, 4, 4, 512, 512)])
def test_conv3x3_fwd_and_input_grad(NB, H, W, Cin, Cout):
    from vtp_amd import ops
    g = torch.Generator(device=DEV).manual_seed(NB * 100 + H)
    x = torch.randn(NB, Cin, H, W, device=DEV, generator=g).to(BF).float()
    w = (torch.randn(Cout, Cin, 3, 3, device=DEV, generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(BF).float()
    b = torch.randn(Cout, device=DEV, generator=g) * 0.1
    _, xs = _stack(x)
    ybuf, ys = _stack(torch.zeros(NB, Cout, H, W, device=DEV))
    ys.fill_(7.0)  # border rows must be overwritten with zeros
    wf = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).to(BF).contiguous()
    ops.conv3x3(xs, wf, b, ys, NB, H, W, Cin, Cout, taps=9, mode=0)
    ref = F.relu(F.conv2d(x, w, b, padding=1))
    out = _unstack(ys, NB, H, W, Cout)
    tol = 1e-3 * float(ref.abs().max()) + 2.0 ** -7 * ref.abs()
    assert bool(((out - ref).abs() <= tol).all()), float((out - ref).abs().max())
    full = ys.view(NB, H + 2, W + 2, Cout).float()
    assert float(full[:, 0].abs().max()) == 0 and float(full[:, -1].abs().max()) == 0
    assert float(full[:, :, 0].abs().max()) == 0 and float(full[:, :, -1].abs().max()) == 0
    # input gradient with the ReLU mask of the layer input
    dy = torch.randn(NB, Cout, H, W, device=DEV, generator=g).to(BF).float()
    xin = torch.randn(NB, Cin, H, W, device=DEV, generator=g).to(BF).float()
    _, dys = _stack(dy)
    _, ms = _stack(xin)
    _, dxs = _stack(torch.zeros(NB, Cin, H, W, device=DEV))
    wd = w.flip(2, 3).permute(1, 2, 3, 0).reshape(Cin, 9 * Cout).to(BF).contiguous()
    ops.conv3x3(dys, wd, None, dxs, NB, H, W, Cout, Cin, taps=9, mode=1, relu_mask=ms)
    ref_dx = F.conv_transpose2d(dy, w, padding=1) * (xin > 0)
    got = _unstack(dxs, NB, H, W, Cin)
    tol = 1e-3 * float(ref_dx.abs().max()) + 2.0 ** -7 * ref_dx.abs()
    assert bool(((got - ref_dx).abs() <= tol).all()), float((got - ref_dx).abs().max())


def test_maxpool_fwd_bwd():
    from vtp_amd import ops
    NB, H, W, C = 2, 8, 12, 64
    g = torch.Generator(device=DEV).manual_seed(0)
    y = F.relu(torch.randn(NB, C, H, W, device=DEV, generator=g)).to(BF).float()
    y[:, :, 0:2, 0:2] = y[:, :, 0:1, 0:1]  # a positive tie: the gradient must go to the first maximum (ATen semantics)
    _, ys = _stack(y)
    _, ps = _stack(torch.zeros(NB, C, H // 2, W // 2, device=DEV))
    ops.maxpool2_fwd(ys, ps, NB, H, W, C)
    assert torch.equal(_unstack(ps, NB, H // 2, W // 2, C), F.max_pool2d(y, 2, 2))
    dp = torch.randn(NB, C, H // 2, W // 2, device=DEV, generator=g).to(BF).float()
    tap = torch.randn(NB, C, H, W, device=DEV, generator=g).to(BF).float()
    _, dps = _stack(dp)
    _, ts = _stack(tap)
    _, dys = _stack(torch.zeros(NB, C, H, W, device=DEV))
    ops.maxpool2_bwd(ys, dps, ts, dys, NB, H, W, C)
    yr = y.clone().requires_grad_(True)
    F.max_pool2d(yr, 2, 2).backward(dp)
    ref = ((yr.grad + tap).to(BF).float()) * (y > 0)
    got = _unstack(dys, NB, H, W, C)
    assert float((got - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())


@pytest.fixture(scope="module")
def lp():
    from oracle import lpips_oracle as L
    from vtp_amd import LPIPS
    g = load_file(os.path.join(os.path.dirname(__file__), "golden", "lpips_tiny.safetensors"))
    sd = L.make_state(int(g["seed"]))
    m = LPIPS(use_dropout=True)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV), sd, g


def _tokens(x):
    """inverse PixelShuffle(16): [B,3,H,W] -> token-major [B*hw, 768] (c*256 + dy*16 + dx), the decoder's output layout."""
    B, _, H, W = x.shape
    return F.pixel_unshuffle(x, 16).permute(0, 2, 3, 1).reshape(B * (H // 16) * (W // 16), 768)


def test_lpips_forward_vs_golden_and_oracle(lp):
    from oracle import lpips_oracle as L
    m, sd, g = lp
    val = m(g["x0"].to(DEV), g["x1"].to(DEV)).cpu()
    ref = g["lpips"]  # outputs of the real reference class
    with torch.no_grad(), torch.autocast("cpu", dtype=BF):
        ref_bf = L.lpips(sd, g["x0"], g["x1"]).float()
    e_ref = float((ref_bf - ref).abs().max())
    e = float((val - ref).abs().max())
    assert e <= 1.5 * e_ref + 2e-3 * float(ref.abs().max()), (e, e_ref, ref.flatten().tolist(), val.flatten().tolist())


def test_lpips_state_dict_keys_match_reference_layout(lp):
    m, sd, _ = lp
    assert set(m.state_dict().keys()) == set(sd.keys())
    with pytest.raises(ValueError):
        m(torch.zeros(1, 3, 24, 24, device=DEV), torch.zeros(1, 3, 24, 24, device=DEV))


def test_lpips_loss_and_grad(lp):
    """trainer entry: token-major decoder output -> per-image LPIPS and the gradient accumulated into dt."""
    from oracle import lpips_oracle as L
    m, sd, g = lp
    x0, x1 = g["x0"], g["x1"]
    B, _, H, W = x0.shape
    tok = _tokens(x0).to(DEV).to(BF).contiguous()
    x0r = F.pixel_shuffle(tok.float().cpu().view(B, H // 16, W // 16, 768).permute(0, 3, 1, 2), 16)  # bf16-rounded input
    d0 = (1e-5 * torch.randn(tok.shape[0], 768, generator=torch.Generator().manual_seed(2))).to(BF)
    dt = d0.clone().to(DEV)  # stands for the L1 gradient already in dt: must be accumulated into, not overwritten
    weight = 3.0
    val = m.loss_and_grad(tok, x1.to(DEV).contiguous(), dt, weight, B, H, W).cpu()
    xr = x0r.clone().requires_grad_(True)
    ref_val = L.lpips(sd, xr, x1)
    (weight * ref_val.mean()).backward()
    ref_grad = _tokens(xr.grad)
    xb = x0r.clone().requires_grad_(True)
    with torch.autocast("cpu", dtype=BF):
        vb = L.lpips(sd, xb, x1)
        (weight * vb.float().mean()).backward()
    e_ref_v = float((vb.float().detach().flatten() - ref_val.detach().flatten()).abs().max())
    assert float((val - ref_val.detach().flatten()).abs().max()) <= 1.5 * e_ref_v + 2e-3 * float(ref_val.abs().max())
    got = dt.float().cpu() - d0.float()
    err = (got - ref_grad).norm() / ref_grad.norm()
    e_ref = (_tokens(xb.grad) - ref_grad).norm() / ref_grad.norm()
    assert float(err) <= 1.5 * float(e_ref) + 2.0 ** -6, (float(err), float(e_ref))


def test_trainer_with_perceptual_term_graphs_equal_eager(lp):
    """rec step with L1 + LPIPS: the hipGraph driver replays exactly what the eager driver computes, and the perceptual
    gradient reaches the parameters (differs from the L1-only step)."""
    from oracle.ref_stubs import TINY
    from vtp_amd import VTPConfig, VTPModel, VTPTrainer
    m, _, _ = lp
    img = torch.randn(2, 3, 64, 64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
    outs = {}
    for mode in ("eager", "graphs", "l1"):
        torch.manual_seed(0)
        model = VTPModel(VTPConfig(**TINY)).to(DEV)
        tr = VTPTrainer(model, lr=1e-3, use_graphs=(mode == "graphs"), lpips=None if mode == "l1" else m,
                        perceptual_weight=0.0 if mode == "l1" else 5.0)
        for _ in range(3):
            tr.step(img)
        torch.cuda.synchronize()
        outs[mode] = (model._store.flat_p.clone(), None if mode == "l1" else tr.lpips_val.clone())
    rel = float((outs["eager"][0] - outs["graphs"][0]).norm() / outs["eager"][0].norm())
    assert rel < 1e-4, rel  # only the fp32 atomics of the bias / norm gradients (and of the LPIPS mean) may reorder
    torch.testing.assert_close(outs["eager"][1], outs["graphs"][1], rtol=1e-3, atol=1e-6)
    assert float(outs["eager"][1].min()) > 0
    assert float((outs["eager"][0] - outs["l1"][0]).norm() / outs["eager"][0].norm()) > 10 * rel
