"""fp8 (e4m3) forward path of BASELINE config 5: quantisation kernels bit-exact against torch's float8_e4m3fn cast, the fp8 MFMA GEMM
against an fp32 matmul of the SAME quantised operands (so only accumulation order differs), the end-to-end encode -> decode
error against the bf16 path, and -- the parity statement of the configuration -- encode / decode outputs
against the ORACLE in fp32 with E_ref taken from the oracle's e4m3-simulated forward (oracle/fp8_oracle.py: the reference
algorithm with the same four GEMM operand pairs per block round-tripped through torch.float8_e4m3fn at per-tensor scales)."""
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


_TINY4 = dict(image_size=64, vision_embed_dim=192, vision_depth=4, vision_num_heads=3, text_embed_dim=128, text_depth=1,
              text_num_heads=2, text_vocab_size=64, text_context_length=8, decoder_embed_dim=192, decoder_depth=4,
              decoder_num_heads=3)
_LARGE24 = dict(vision_embed_dim=1024, vision_depth=24, vision_num_heads=16, decoder_embed_dim=1024, decoder_depth=24,
                decoder_num_heads=16, text_embed_dim=128, text_depth=1, text_num_heads=2, text_vocab_size=64, text_context_length=8)


@pytest.mark.parametrize("name,cfg_kw,heads,B,res", [("4-block D=192", _TINY4, 3, 8, 64), ("VTP-L 24-block (config 5)", _LARGE24, 16, 2, 256),
                                                     ("4-block D=192 with QK normalisation", dict(_TINY4, vision_use_qk_norm=True, decoder_use_qk_norm=True), 3, 8, 64)])
def test_fp8_encode_decode_vs_e4m3_simulated_oracle(name, cfg_kw, heads, B, res):
    """|ours_fp8 - oracle_fp32| <= 1.25 x |oracle_fp8sim - oracle_fp32| on the latents, on the decoder output for the SAME
    (reference) latents and end to end; E_ref = the larger of the simulated forward under CPU and CUDA bf16 autocast (the
    like-for-like precision of everything outside the GEMM operands).  The fp32-everything-else simulation is printed too."""
    from oracle import fp8_oracle as F8
    from vtp_amd import VTPConfig, VTPModel
    torch.manual_seed(3)
    m = VTPModel(VTPConfig(**cfg_kw))
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.ndim <= 1 and n != "logit_scale":
                p.add_(0.02 * torch.randn_like(p))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(DEV).eval()
    img = torch.randn(B, 3, res, res, generator=torch.Generator().manual_seed(5))
    calib = img[: max(2, B // 2)]
    with torch.no_grad():
        lat_ref, rec_ref = F8.encode_decode(sd, img, heads, heads)               # fp32 reference algorithm
        sim = F8.calibrate(sd, calib, heads, heads)
        sims = {"fp32": (F8.encode_decode(sd, img, heads, heads, lin=sim), F8.O.decoder_forward(sd, lat_ref, heads, lin=sim))}
        with torch.autocast("cpu", dtype=torch.bfloat16):
            sims["cpu16"] = (F8.encode_decode(sd, img, heads, heads, lin=sim), F8.O.decoder_forward(sd, lat_ref, heads, lin=sim))
        sd_g = {k: v.to(DEV) for k, v in sd.items()}
        sim_g = F8.Fp8Sim(sd_g)
        sim_g.amax, sim_g.mode = dict(sim.amax), "apply"
        with torch.autocast("cuda", dtype=torch.bfloat16):
            sims["gpu16"] = (F8.encode_decode(sd_g, img.to(DEV), heads, heads, lin=sim_g),
                             F8.O.decoder_forward(sd_g, lat_ref.to(DEV), heads, lin=sim_g))
        m.enable_fp8_forward(calib.to(DEV))
        lat = m.get_reconstruction_latents(img.to(DEV))
        rec_e2e = m.get_latents_decoded_images(lat)
        rec_same = m.get_latents_decoded_images(lat_ref.to(DEV))
    ours = {"latents": (lat, lat_ref), "decoder(same latents)": (rec_same, rec_ref), "end-to-end": (rec_e2e, rec_ref)}
    pick = {"latents": lambda t: t[0][0], "decoder(same latents)": lambda t: t[1], "end-to-end": lambda t: t[0][1]}
    for what, (o, ref) in ours.items():
        e = relF(o, ref)
        er = {tag: relF(pick[what](sims[tag]), ref) for tag in sims}
        e_ref = max(er["cpu16"], er["gpu16"])
        print(f"PARITY fp8 [{name}] {what}: E_ours={e:.3e} E_ref(fp8-sim: fp32 rest)={er['fp32']:.3e} (cpu autocast)={er['cpu16']:.3e} "
              f"(cuda autocast)={er['gpu16']:.3e} E_ours/E_ref={e / e_ref:.2f}")
        assert e <= 1.25 * e_ref, (name, what, e, e_ref)
        assert e > 0.2 * er["fp32"], "suspiciously exact: did the fp8 path run?"
