"""The drop-in boundary on the GPU: `loss.backward()` through `VTPModel` / `VTP` (vtp_amd/autograd.py) must
reproduce the hand-scheduled trainer's flat gradients and the reference's autograd (golden `grad2.*`); the legacy training
class surface (vtp/models/vtp.py:275-407: forward(forward_type), encode_image / encode_text / get_logits, `proj` /
`teacher_proj` checkpoint keys + their EMA) is pinned to golden outputs of the REAL legacy class
(tests/golden/vtp_tiny_legacy.safetensors, oracle/make_golden_legacy.py)."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def relF(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _tiny(golden_sd):
    from oracle.ref_stubs import TINY
    from vtp_amd import VTPConfig, VTPModel
    m = VTPModel(VTPConfig(**TINY))
    m.load_state_dict(golden_sd, strict=True)
    return m.to(DEV)


def _clip_loss(i, t, scale):
    logits = scale * i @ t.T
    labels = torch.arange(logits.shape[0], device=logits.device)
    return 0.5 * (F.cross_entropy(logits, labels) + F.cross_entropy(logits.T, labels))


def test_loss_backward_through_model_matches_trainer_and_reference_autograd(golden, golden_sd):
    """rec (L1) + clip (InfoNCE) built by the CALLER from model(...) outputs with torch ops, then loss.backward()."""
    from oracle import vtp_oracle as O
    from oracle.make_golden import GRAD2_KEYS
    from vtp_amd import VTPTrainer
    img, txt = golden["in.image"].to(DEV), golden["in.text"].to(DEV)
    m = _tiny(golden_sd)
    m.train()
    m.zero_grad()
    rec = m(image=img, forward_type="rec")
    assert set(rec) == {"latents", "reconstructed_image", "target_image"} and rec["reconstructed_image"].requires_grad
    clip = m(image=img, text=txt, forward_type="clip")
    assert clip["image_features"].requires_grad and clip["text_features"].requires_grad
    l1 = (rec["reconstructed_image"] - img).abs().mean()
    lc = _clip_loss(clip["image_features"], clip["text_features"], clip["logit_scale"])
    (l1 + lc).backward()
    torch.cuda.synchronize()
    print(f"autograd path: L1 {float(l1):.5f} (golden {float(golden['out.rec_l1_loss']):.5f}) clip {float(lc):.5f} "
          f"(golden {float(golden['out.clip_loss']):.5f})")
    assert abs(float(l1) - float(golden["out.rec_l1_loss"])) < 2e-3 * float(golden["out.rec_l1_loss"])
    assert abs(float(lc) - float(golden["out.clip_loss"])) < 5e-3 * float(golden["out.clip_loss"])
    g_auto = m._store.flat_g.clone()
    params = dict(m.named_parameters())
    # (a) the fused trainer on an identical model
    m2 = _tiny(golden_sd)
    tr = VTPTrainer(m2, lr=0.0, weight_decay=0.0)
    tr.step(img, txt)
    torch.cuda.synchronize()
    g_tr = m2._store.flat_g
    rel = float((g_auto - g_tr).norm() / g_tr.norm())
    print(f"flat gradient: autograd path vs VTPTrainer rel diff = {rel:.3e}")
    assert rel < 1.5e-2  # two trunk passes + torch heads vs one shared pass + fused heads: bf16 rounding of d_xnf only
    # (b) the reference's autograd (golden grad2.*), noise floor = the oracle's bf16-autocast backward
    sd = {k: v.clone().requires_grad_(v.dtype == torch.float32) for k, v in golden_sd.items()}
    with torch.autocast("cpu", dtype=torch.bfloat16):
        a, b = O.rec_clip_train_loss(sd, golden["in.image"], golden["in.text"], 2, 2, 2)
        (a + b).backward()
    for k in GRAD2_KEYS:
        ref = golden["grad2." + k]
        e, e_ref = relF(params[k].grad, ref), relF(sd[k].grad, ref)
        print(f"autograd grad2 {k}: E_ours={e:.3e} E_ref={e_ref:.3e}")
        assert e <= max(1.5 * e_ref, 2.5e-2), k
    # a torch optimizer consumes the gradients (views of the flat buffer), set_to_none included
    opt = torch.optim.SGD(m.parameters(), lr=1e-2)
    w0 = params["trunk.blocks.0.attn.qkv.weight"].detach().clone()
    opt.step()
    m.refresh_weights()
    assert float((params["trunk.blocks.0.attn.qkv.weight"] - w0).abs().max()) > 0
    opt.zero_grad(set_to_none=True)
    assert params["trunk.norm.weight"].grad is None
    rec = m(image=img, forward_type="rec")
    (rec["reconstructed_image"] - img).abs().mean().backward()
    g = params["trunk.norm.weight"].grad
    assert g is not None and g.data_ptr() == m._store.g("trunk.norm.weight").data_ptr() and float(g.abs().max()) > 0
    assert float(params["ln_final.weight"].grad.abs().max()) == 0.0  # fresh (zeroed) view: the text tower was not used


def test_inference_modes_and_stale_backward(golden, golden_sd):
    m = _tiny(golden_sd)
    img = golden["in.image"].to(DEV)
    m.eval()
    lat = m.get_reconstruction_latents(img)
    assert not lat.requires_grad
    m.train()
    with torch.no_grad():
        assert not m.get_reconstruction_latents(img).requires_grad
    a = m.get_reconstruction_latents(img)
    b = m.get_reconstruction_latents(img)  # same workspace: `a`'s saved activations are gone
    assert torch.equal(a, b) and relF(a, golden["out.latents"]) < 1.5e-2
    with pytest.raises(RuntimeError, match="overwritten"):
        a.sum().backward()
    b.sum().backward()


@pytest.mark.parametrize("clip_feat,ae_only", [("pooled", True), ("cls", False), ("pooled", False)])
def test_clip_feature_variants(golden, golden_sd, clip_feat, ae_only):
    """vision_clip_feat='pooled' / vision_bottleneck_ae_only=False (modeling_vtp.py:262-276) against the oracle, fwd + grads"""
    from oracle import vtp_oracle as O
    from oracle.ref_stubs import TINY
    from vtp_amd import VTPConfig, VTPModel
    cfg = VTPConfig(**{**TINY, "vision_clip_feat": clip_feat, "vision_bottleneck_ae_only": ae_only})
    torch.manual_seed(1)
    m = VTPModel(cfg)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(DEV)
    img = golden["in.image"]
    m.eval()
    f = m.get_clip_image_feature(img.to(DEV))
    ref = O.clip_image_feature(sd, img, 2, clip_feat=clip_feat, ae_only=ae_only)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        rb = O.clip_image_feature(sd, img, 2, clip_feat=clip_feat, ae_only=ae_only)
    print(f"clip feature {clip_feat}/ae_only={ae_only}: E_ours={relF(f, ref):.3e} E_ref={relF(rb, ref):.3e}")
    assert f.shape == ref.shape and relF(f, ref) <= max(1.5 * relF(rb, ref), 8e-3)
    m.train()
    m.zero_grad()
    w = torch.linspace(-1, 1, f.shape[1], device=DEV)
    (m.get_clip_image_feature(img.to(DEV)) * w).sum().backward()
    s2 = {k: v.clone().requires_grad_(v.dtype == torch.float32) for k, v in sd.items()}
    (O.clip_image_feature(s2, img, 2, clip_feat=clip_feat, ae_only=ae_only) * w.cpu()).sum().backward()
    params = dict(m.named_parameters())
    for k in ("visual_proj.weight", "trunk.blocks.1.mlp.w3.weight", "trunk.patch_embed.proj.weight"):
        e = relF(params[k].grad, s2[k].grad)
        print(f"   grad {k}: {e:.3e}")
        assert e < 4e-2, k


# ------------------------------------------------------------------------------------------------------------ legacy class
@pytest.fixture(scope="module")
def legacy():
    from safetensors.torch import load_file
    g = load_file(os.path.join(ROOT, "tests", "golden", "vtp_tiny_legacy.safetensors"))
    return g, {k[3:]: v for k, v in g.items() if k.startswith("sd.")}


def _legacy_model(sd):
    from oracle.make_golden_legacy import CFG as C
    from vtp_amd import VTP, VTPConfig
    cfg = VTPConfig(image_size=C["R"], vision_embed_dim=C["embed_dim"], vision_depth=C["depth"], vision_num_heads=C["heads"],
                    text_embed_dim=C["embed_dim"], text_depth=C["text_layers"], text_num_heads=C["text_heads"],
                    text_vocab_size=C["vocab"], text_context_length=C["ctx"], decoder_embed_dim=C["embed_dim"],
                    decoder_depth=C["dec_depth"], decoder_num_heads=C["dec_heads"])
    m = VTP(cfg, dino_out_dim=C["K"], dino_hidden_dim=C["hidden"], dino_bottleneck_dim=C["bott"])
    out = m.load_state_dict(sd, strict=True)  # legacy key layout: proj / teacher_proj / transformer.resblocks.*
    assert not out.missing_keys and not out.unexpected_keys
    return m.to(DEV)


def test_legacy_checkpoint_loads_strict_and_forward_types_match_reference(legacy):
    g, sd = legacy
    assert "proj.weight" in sd and "teacher_proj.weight" in sd and any(k.startswith("transformer.resblocks.") for k in sd)
    m = _legacy_model(sd)
    assert set(m.legacy_state_dict()) == set(sd)
    for k, v in m.legacy_state_dict().items():
        assert torch.equal(v.cpu(), sd[k]), k
    m.eval()
    img, txt = g["in.image"].to(DEV), g["in.text"].to(DEV)
    out = m(image=img, text=txt, forward_type="clip")
    assert set(out) == {"image_features", "text_features", "logit_scale"}
    e_i, e_t = relF(out["image_features"], g["clip.image_features"]), relF(out["text_features"], g["clip.text_features"])
    print(f"legacy forward_type=clip: image {e_i:.3e} text {e_t:.3e}")
    assert e_i < 1.5e-2 and e_t < 1.5e-2 and abs(float(out["logit_scale"]) - float(g["clip.logit_scale"])) < 1e-5
    r = m(reconstruction_image=img, forward_type="rec")
    assert set(r) == {"reconstructed_image", "target_image"} and r["target_image"] is img
    e_r = relF(r["reconstructed_image"], g["rec.reconstructed_image"])
    print(f"legacy forward_type=rec: {e_r:.3e}")
    assert e_r < 2e-2
    assert relF(m.encode_image(img), g["enc.image"]) < 1.5e-2 and relF(m.encode_text(txt), g["enc.text"]) < 1.5e-2
    li, lt = m.get_logits(img, txt)
    assert float((li.cpu() - g["logits.image"]).abs().max()) < 3e-2 * max(1.0, float(g["logits.image"].abs().max()))
    assert torch.equal(lt, li.T)
    with pytest.raises(AssertionError):
        m(image=img, forward_type="feature")
    # EMA: trunk, proj -> teacher_proj (vtp.py:396-398) and dino_head
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m.update_teacher(0.9)
    after = m.state_dict()
    for t_k, s_k in (("teacher_proj.weight", "visual_proj.weight"), ("teacher_trunk.blocks.1.mlp.w3.weight", "trunk.blocks.1.mlp.w3.weight"),
                     ("teacher_dino_head.mlp.2.bias", "dino_head.mlp.2.bias")):
        want = 0.9 * before[t_k] + 0.1 * before[s_k]
        assert torch.allclose(after[t_k], want, rtol=1e-6, atol=1e-7), t_k
        assert torch.equal(after[s_k], before[s_k])


def test_legacy_ssl_forward_is_differentiable(legacy):
    """forward_type='ssl' in training mode: our loss spec in torch on the returned dicts, loss.backward(), gradients against
    the oracle's autograd on the same (legacy-layout) weights."""
    from oracle import vtp_oracle as O
    from vtp_amd.data import collate_ssl_batch
    import numpy as np
    g, sd = legacy
    m = _legacy_model(sd)
    m.train()
    rng = np.random.default_rng(5)
    gen = torch.Generator().manual_seed(9)
    B = 3
    batch = collate_ssl_batch([torch.randn(B, 3, 64, 64, generator=gen) for _ in range(2)],
                              [torch.randn(B, 3, 32, 32, generator=gen) for _ in range(4)], rng=rng)
    masks = batch["masks"]
    ssl_dict = {k: batch[k] for k in ("global_crops", "n_global_crops", "mask_indices_list", "n_masked_patches", "upperbound",
                                      "local_crops", "masks")}
    ssl_dict["global_crops"], ssl_dict["local_crops"] = batch["global_crops"].to(DEV), batch["local_crops"].to(DEV)
    K = sd["dino_head.last_layer.weight_v"].shape[0]
    c_d, c_i = torch.zeros(K), torch.zeros(K)
    m.zero_grad()
    t_out, s_out = m(ssl_dict=ssl_dict, forward_type="ssl")
    assert set(s_out) == {"student_local_cls_tokens_after_head", "student_global_cls_tokens_after_head", "student_global_cls_tokens",
                          "student_global_masked_patch_tokens_after_head"}
    assert s_out["student_global_cls_tokens_after_head"].requires_grad and not t_out["teacher_cls_tokens_after_head"].requires_grad
    loss = O.ssl_loss({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in t_out.items()},
                      {k: v.cpu() for k, v in s_out.items()}, masks, c_d, c_i, n_local=4)
    loss.backward()
    torch.cuda.synchronize()
    hf = {("visual_proj." + k[5:] if k.startswith("proj.") else "text_transformer." + k[12:] if k.startswith("transformer.") else k): v
          for k, v in sd.items()}
    ref = {k: v.clone().requires_grad_(v.dtype == torch.float32 and not k.startswith("teacher_")) for k, v in hf.items()}
    t_r, s_r = O.ssl_outputs(ref, batch["global_crops"], batch["local_crops"], masks, 2)
    loss_r = O.ssl_loss(t_r, s_r, masks, c_d, c_i, n_local=4)
    loss_r.backward()
    print(f"legacy ssl autograd: loss ours {float(loss):.5f} oracle {float(loss_r):.5f}")
    assert abs(float(loss) - float(loss_r)) < 5e-3 * abs(float(loss_r))
    params = dict(m.named_parameters())
    for k in ("dino_head.mlp.0.weight", "dino_head.last_layer.weight_v", "dino_head.last_layer.weight_g", "trunk.blocks.0.attn.qkv.weight",
              "trunk.blocks.1.mlp.w2.weight", "trunk.mask_token", "trunk.cls_token", "trunk.patch_embed.proj.weight"):
        e = relF(params[k].grad, ref[k].grad)
        print(f"   ssl grad {k}: {e:.3e}")
        assert e < 5e-2, k


def test_torch_ops_vtp_hip_callable():
    """the raw kernels through torch.ops.vtp_hip (torch.library registration, vtp_amd/torch_ops.py)"""
    import vtp_amd.torch_ops  # noqa: F401
    g = torch.Generator(device=DEV).manual_seed(0)
    M, N, K = 300, 256, 128
    a = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.1).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV, generator=g)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    torch.ops.vtp_hip.gemm_nt(a, w, c, bias, None, None, M, N, K, 0)
    ref = a.float() @ w.float().T + bias
    assert relF(c, ref) < 5e-3
    x = torch.randn(M, N, device=DEV, generator=g)
    wn = torch.rand(N, device=DEV, generator=g) + 0.5
    y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    st = torch.empty(M, 2, device=DEV)
    torch.ops.vtp_hip.norm_fwd(x, wn, None, y, st, 1e-5, 0)
    refn = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5) * wn
    assert relF(y, refn) < 5e-3
    dw = torch.zeros(N, K, device=DEV)
    db = torch.zeros(N, device=DEV)
    dy = torch.randn(M, N, device=DEV, generator=g).to(torch.bfloat16)
    torch.ops.vtp_hip.gemm_tn(dy, a, dw, N, K, M, True, db)
    assert relF(dw, dy.float().T @ a.float()) < 1e-3 and relF(db, dy.float().sum(0)) < 1e-3


def test_patch_model_rebinds_a_reference_style_instance(golden, golden_sd):
    """vtp_amd.patch_model: an object with the reference class's surface (here: a torch module with the reference's parameter
    names, a PretrainedConfig-like config with extra bookkeeping keys, and methods that must NOT be called afterwards) keeps its
    identity and serves every API call from the HIP path"""
    from oracle.ref_stubs import TINY
    from vtp_amd import VTPConfig, VTPModel, patch_model

    class Cfg:
        def __init__(self, d):
            self.d = d

        def to_dict(self):
            return dict(self.d, transformers_version="4.x", model_type="vtp", architectures=["VTPModel"])

    class RefLike(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.config = Cfg(VTPConfig(**TINY).to_dict())
            self.holder = torch.nn.ParameterDict()  # flat storage under the reference's key names
            self._sd = {k: v.clone() for k, v in golden_sd.items()}

        def state_dict(self, *a, **k):
            return self._sd

        def get_reconstruction_latents(self, image):
            raise AssertionError("the reference path must not run after patch_model")

        def forward(self, *a, **k):
            raise AssertionError("the reference path must not run after patch_model")

    ref = RefLike().eval()
    out = patch_model(ref)
    assert out is ref and isinstance(ref._vtp_amd, VTPModel) and not ref._vtp_amd.training
    img, txt = golden["in.image"].to(DEV), golden["in.text"].to(DEV)
    direct = VTPModel(VTPConfig(**TINY))
    direct.load_state_dict(golden_sd)
    direct = direct.to(DEV).eval()
    with torch.no_grad():
        lat = ref.get_reconstruction_latents(img)
        assert torch.equal(lat, direct.get_reconstruction_latents(img))
        assert torch.equal(ref.get_latents_decoded_images(lat), direct.get_latents_decoded_images(lat))
        assert torch.equal(ref.get_clip_text_feature(txt), direct.get_clip_text_feature(txt))
        r1, r2 = ref(image=img, forward_type="rec"), direct(image=img, forward_type="rec")
        assert set(r1) == set(r2) and torch.equal(r1["reconstructed_image"], r2["reconstructed_image"])
    assert set(dict(ref.named_parameters())) == set(dict(direct.named_parameters()))
    ref.train()
    assert ref._vtp_amd.training
    loss = (ref.get_latents_decoded_images(ref.get_reconstruction_latents(img)) - img).abs().mean()
    loss.backward()  # differentiable through the patched object
    assert float(dict(ref.named_parameters())["trunk.blocks.0.attn.qkv.weight"].grad.abs().sum()) > 0


def test_gradient_with_respect_to_the_input_image(golden, golden_sd):
    """VERDICT r3 'missing' item 5: the reference's autograd delivers d loss / d image when the image requires grad (PatchEmbed is an
    nn.Conv2d, embeddings.py:61-70).  Here: d(L1 reconstruction + a CLIP-feature functional) / d image through EncodeLatents /
    TrunkTokens (dgrad of the patch-embed GEMM + vtp_col2im16) against the oracle's autograd in fp32; E_ref = the oracle under CPU
    bf16 autocast; bar E_ours <= 1.5 E_ref (tiny model)."""
    from oracle import vtp_oracle as O
    img0, tgt = golden["in.image"], golden["in.image"].flip(0)

    def oracle_grad(autocast):
        x = img0.clone().requires_grad_(True)
        import contextlib
        ctx = torch.autocast("cpu", dtype=torch.bfloat16) if autocast else contextlib.nullcontext()
        with ctx:
            lat = O.reconstruction_latents(golden_sd, x, 2)
            rec = O.decoder_forward(golden_sd, lat, 2)
            feat = O.clip_image_feature(golden_sd, x, 2, normalize=True)
            loss = (rec.float() - tgt).abs().mean() + feat.float()[:, :7].sum() * 0.1
        loss.backward()
        return x.grad.clone(), float(loss)

    g_ref, l_ref = oracle_grad(False)
    g_16, _ = oracle_grad(True)
    m = _tiny(golden_sd)
    m.train()
    m.zero_grad()
    x = img0.clone().to(DEV).requires_grad_(True)
    rec = m(image=x, forward_type="rec")["reconstructed_image"]
    feat = m.get_clip_image_feature(x, normalize=True)
    loss = (rec - tgt.to(DEV)).abs().mean() + feat[:, :7].sum() * 0.1
    loss.backward()
    torch.cuda.synchronize()
    assert x.grad is not None and x.grad.shape == img0.shape and x.grad.dtype == torch.float32
    e, e_ref = relF(x.grad, g_ref), relF(g_16, g_ref)
    print(f"d loss / d image: loss ours {float(loss):.5f} oracle {l_ref:.5f}; E_ours={e:.3e} E_ref={e_ref:.3e} ratio={e / e_ref:.2f}")
    assert e <= 1.5 * e_ref
    # without requires_grad on the image nothing is computed or returned
    m.zero_grad()
    y = img0.clone().to(DEV)
    m(image=y, forward_type="rec")["reconstructed_image"].mean().backward()
    assert y.grad is None
