"""fp8 (e4m3) forward path of BASELINE config 5: quantisation kernels bit-exact against torch's float8_e4m3fn cast, the fp8 MFMA GEMM
against an fp32 matmul of the SAME quantised operands (so only accumulation order differs), and the end-to-end encode -> decode
error against the bf16 path."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def relF(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_quantize_matches_torch_e4m3(dtype):
    from vtp_amd import ops
    torch.manual_seed(0)
    x = (torch.randn(4096 * 8, device=DEV) * 3).to(dtype)
    x[:8] = torch.tensor([0.0, 1e-9, -1e-9, 448.0, 500.0, -1000.0, 0.0156, 240.0], device=DEV).to(dtype)
    am = torch.zeros(1, device=DEV)
    ops.amax(x, am)
    assert float(am) == float(x.float().abs().max())
    scale = 448.0 / float(am) * 16  # some values saturate
    q = torch.empty(x.numel(), dtype=torch.uint8, device=DEV)
    ops.quantize_e4m3(x, q, scale)
    ref = (x.float() * scale).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(q, ref)
    back = torch.empty(x.numel(), device=DEV)
    ops.dequantize_e4m3(q, back, 1.0 / scale)
    assert torch.equal(back, ref.view(torch.float8_e4m3fn).float() / scale) or relF(back, ref.view(torch.float8_e4m3fn).float() / scale) < 1e-7
    sdev = torch.tensor([scale], device=DEV)
    q2 = torch.empty_like(q)
    ops.quantize_e4m3(x, q2, sdev)
    assert torch.equal(q2, q)


@pytest.mark.parametrize("M,N,K", [(512, 768, 1024), (16448, 3072, 1024), (300, 1024, 2736), (8192, 1024, 1024)])
def test_gemm_fp8_vs_fp32_of_quantised_operands(M, N, K):
    from vtp_amd import ops
    torch.manual_seed(1)
    a = torch.randn(M, K, device=DEV)
    w = torch.randn(N, K, device=DEV) * 0.05
    sa, sw = 448.0 / float(a.abs().max()), 448.0 / float(w.abs().max())
    a8 = torch.empty(M, K, dtype=torch.uint8, device=DEV)
    w8 = torch.empty(N, K, dtype=torch.uint8, device=DEV)
    ops.quantize_e4m3(a, a8, sa)
    ops.quantize_e4m3(w, w8, sw)
    ad, wd = a8.view(torch.float8_e4m3fn).float() / sa, w8.view(torch.float8_e4m3fn).float() / sw
    bias = torch.randn(N, device=DEV)
    resid = torch.randn(M, N, device=DEV)
    ref = ad @ wd.T
    # bf16 output + bias
    c = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt_fp8(a8, w8, c, M=M, N=N, K=K, alpha=1.0 / (sa * sw), bias=bias, epi=ops.EPI_BF16)
    e = relF(c, ref + bias)
    print(f"fp8 gemm {M}x{N}x{K}: bf16-out rel {e:.2e}; quantisation error vs unquantised {relF(ref, a @ w.T):.2e}")
    assert e < 4e-3  # bf16 rounding of the output only
    # f32 output + bias + residual
    c32 = torch.empty(M, N, device=DEV)
    ops.gemm_nt_fp8(a8, w8, c32, M=M, N=N, K=K, alpha=1.0 / (sa * sw), bias=bias, resid=resid, epi=ops.EPI_F32)
    assert relF(c32, ref + bias + resid) < 5e-5  # fp32 accumulation order (the torch fp32 matmul is no more exact)


def test_fp8_encode_decode_vs_bf16_path():
    """end to end: fp8 forward (trunk + pixel decoder) against the bf16 path of the same model on the same images; the stated
    bar for config 5 is the error of per-tensor e4m3 through 2 x depth blocks, printed here and bounded loosely"""
    from vtp_amd import VTPConfig, VTPModel
    torch.manual_seed(0)
    cfg = VTPConfig(image_size=64, vision_embed_dim=192, vision_depth=4, vision_num_heads=3, text_embed_dim=128, text_depth=1,
                    text_num_heads=2, text_vocab_size=64, text_context_length=8, decoder_embed_dim=192, decoder_depth=4,
                    decoder_num_heads=3)
    m = VTPModel(cfg).to(DEV).eval()
    img = torch.randn(8, 3, 64, 64, device=DEV)
    with torch.no_grad():
        lat_b = m.get_reconstruction_latents(img)
        rec_b = m.get_latents_decoded_images(lat_b)
        m.enable_fp8_forward(img[:4])
        lat_8 = m.get_reconstruction_latents(img)
        rec_8 = m.get_latents_decoded_images(lat_b)
        e_lat, e_rec = relF(lat_8, lat_b), relF(rec_8, rec_b)
        print(f"fp8 vs bf16: latents rel {e_lat:.3e}, reconstruction (same latents) rel {e_rec:.3e}")
        assert 1e-4 < e_lat < 0.15 and 1e-4 < e_rec < 0.15  # it must differ (fp8 really ran) and stay a small perturbation
        m.disable_fp8_forward()
        assert torch.equal(m.get_reconstruction_latents(img), lat_b)
    with pytest.raises(NotImplementedError):
        bad = VTPModel(VTPConfig(image_size=64, vision_embed_dim=128, vision_depth=1, vision_num_heads=2, text_embed_dim=128,
                                 text_depth=1, text_num_heads=2, text_vocab_size=64, text_context_length=8, decoder_embed_dim=128,
                                 decoder_depth=1, decoder_num_heads=2)).to(DEV).eval()
        bad.enable_fp8_forward(img[:2])
