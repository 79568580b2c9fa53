"""Pins the oracle restatement against the REAL reference imported from /root/reference (authoring
container only -- skipped on the GPU box where the tree is absent)."""
import pytest
import torch

from oracle import vtp_oracle as O
from oracle.ref_stubs import TINY, load_reference, reference_available

pytestmark = pytest.mark.skipif(not reference_available(), reason="/root/reference not present")


@pytest.fixture(scope="module")
def ref_model():
    ns = load_reference()
    torch.manual_seed(3)
    cfg = dict(TINY)
    cfg.update(image_size=96, vision_depth=3)
    m = ns.VTPModel(ns.VTPConfig(**cfg)).eval()
    with torch.no_grad():
        for p in m.parameters():
            if p.ndim <= 1 and p.numel() > 1:
                p.add_(0.05 * torch.randn_like(p))
    return m


@pytest.mark.parametrize("res", [(96, 96), (64, 128)])
def test_encode_decode(ref_model, res):
    sd = ref_model.state_dict()
    img = torch.randn(2, 3, *res)
    with torch.no_grad():
        lat_ref = ref_model.get_reconstruction_latents(img)
        rec_ref = ref_model.get_latents_decoded_images(lat_ref)
        lat = O.reconstruction_latents(sd, img, 2)
        rec = O.decoder_forward(sd, lat, 2)
    torch.testing.assert_close(lat, lat_ref, rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(rec, rec_ref, rtol=2e-4, atol=2e-5)


def test_clip(ref_model):
    sd = ref_model.state_dict()
    img = torch.randn(2, 3, 96, 96)
    text = torch.randint(1, 500, (2, 16))
    text[:, 7] = 511
    with torch.no_grad():
        li, _ = ref_model.get_clip_logits(img, text)
        lo = O.clip_logits(sd, img, text, 2, 2)
    torch.testing.assert_close(lo, li, rtol=1e-4, atol=1e-4)


def test_masked_trunk(ref_model):
    sd = ref_model.state_dict()
    img = torch.randn(2, 3, 96, 96)
    masks = torch.rand(2, 36) < 0.3
    with torch.no_grad():
        r = ref_model.trunk(img, is_training=True, masks=masks, use_bottleneck=False)
        o = O.trunk_forward(sd, img, 2, use_bottleneck=False, masks=masks)
    torch.testing.assert_close(o["x_norm_patchtokens"], r["x_norm_patchtokens"], rtol=2e-4, atol=2e-5)


def test_lpips():
    """oracle LPIPS == the reference's LPIPS class (vtp/utils/lpips.py) on the same seeded weights, eval mode."""
    from oracle import lpips_oracle as L
    from oracle.ref_stubs import load_reference_lpips
    LPIPS = load_reference_lpips()
    m = LPIPS(use_dropout=True).eval()
    sd = L.make_state(3)
    m.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(5)
    x0, x1 = torch.rand(2, 3, 32, 48, generator=g) * 2 - 1, torch.rand(2, 3, 32, 48, generator=g) * 2 - 1
    with torch.no_grad():
        torch.testing.assert_close(L.lpips(sd, x0, x1), m(x0, x1), rtol=1e-5, atol=1e-7)


def test_intermediate_layers(ref_model):
    sd = ref_model.state_dict()
    img = torch.randn(2, 3, 96, 96)
    with torch.no_grad():
        for kw in (dict(n=2, reshape=False, return_class_token=True, norm=True), dict(n=[0, 2], reshape=True, return_class_token=False, norm=False)):
            ref = ref_model.get_intermediate_layers_feature(img, **kw)
            got = O.intermediate_layers(sd, img, 2, **kw)
            assert len(ref) == len(got)
            for r, g_ in zip(ref, got):
                if kw["return_class_token"]:
                    torch.testing.assert_close(g_[0], r[0], rtol=2e-4, atol=2e-5)
                    torch.testing.assert_close(g_[1], r[1], rtol=2e-4, atol=2e-5)
                else:
                    torch.testing.assert_close(g_, r, rtol=2e-4, atol=2e-5)


def test_layerscale_and_stochastic_depth_branch():
    """LayerScale (vision_init_values) and the training branch with sample drop (block.py:207-233) of the REAL trunk vs the
    oracle fed with the image subsets the reference drew (torch.randperm, replayed from the same seed in call order)."""
    ns = load_reference()
    torch.manual_seed(5)
    cfg = dict(TINY)
    cfg.update(image_size=64, vision_depth=3, vision_init_values=0.3, decoder_init_values=0.2)
    m = ns.VTPModel(ns.VTPConfig(**cfg))
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.ndim <= 1 and p.numel() > 1:
                p.add_(0.05 * torch.randn_like(p))
    sd = m.state_dict()
    assert "trunk.blocks.0.ls1.gamma" in sd and "pixel_decoder.blocks.1.ls2.gamma" in sd
    img = torch.randn(5, 3, 64, 64)
    m.eval()
    with torch.no_grad():
        lat_ref = m.get_reconstruction_latents(img)
        rec_ref = m.get_latents_decoded_images(lat_ref)
        torch.testing.assert_close(O.reconstruction_latents(sd, img, 2), lat_ref, rtol=2e-4, atol=2e-5)
        torch.testing.assert_close(O.decoder_forward(sd, lat_ref, 2), rec_ref, rtol=2e-4, atol=2e-5)
    # training branch: drop_ratio 0.4 on 5 images -> keep 3, alpha 5/3
    m.train()
    ratio, B = 0.4, 5
    torch.manual_seed(11)
    with torch.no_grad():
        r = m.trunk(img, is_training=True, use_bottleneck=False, drop_ratio=ratio)
    torch.manual_seed(11)
    keep = max(int(B * (1 - ratio)), 1)
    drop = []
    for _ in range(3):
        i1 = torch.randperm(B)[:keep]
        i2 = torch.randperm(B)[:keep]
        drop.append((i1, B / keep, i2, B / keep))
    with torch.no_grad():
        o = O.trunk_forward(sd, img, 2, use_bottleneck=False, drop=drop)
    torch.testing.assert_close(o["x_norm_patchtokens"], r["x_norm_patchtokens"], rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(o["x_norm_clstoken"], r["x_norm_clstoken"], rtol=2e-4, atol=2e-5)


def test_center_crop_arr_matches_reference():
    """vtp_amd.tokenizer.center_crop_arr vs vtp/utils/image_utils.py:5-33 (pure PIL / numpy: imported straight from the tree)"""
    import importlib.util

    import numpy as np
    from PIL import Image
    spec = importlib.util.spec_from_file_location("_ref_image_utils", "/root/reference/vtp/utils/image_utils.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    from vtp_amd.tokenizer import center_crop_arr
    rng = np.random.default_rng(0)
    for (h, w), size in [((300, 500), 64), ((64, 64), 64), ((130, 97), 48), ((1000, 700), 96), ((50, 80), 64)]:
        im = Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
        a, b = np.asarray(center_crop_arr(im, size)), np.asarray(ref.center_crop_arr(im, size))
        assert a.shape == (size, size, 3) and np.array_equal(a, b)


def test_from_vtp_yaml_matches_reference(tmp_path):
    """VTPConfig.from_vtp_yaml against the reference classmethod (configuration_vtp.py:169-234) on a synthetic legacy YAML; the
    reference reads it with OmegaConf, which is stubbed here by a tiny attribute-dict loader over PyYAML"""
    import sys

    import yaml
    from vtp_amd import VTPConfig
    doc = {"data": {"image_size": 224},
           "training": {"train_clip": True, "train_reconstruction": True, "init_logit_scale": 2.0, "nonscalar_logit_scale": False},
           "vtp_model": {
               "vision_encoder": {"patch_size": 16, "embed_dim": 384, "depth": 12, "num_heads": 6, "mlp_ratio": 4.0, "ffn_layer": "swiglu",
                                  "norm_type": "rmsnorm", "vit_feature_bottleneck": 64, "bottleneck_ae_only": True, "clip_feat": "cls"},
               "text_encoder": {"context_length": 77, "vocab_size": 49408, "embed_dim": 384, "heads": 6, "layers": 12, "mlp_ratio": 4.0,
                                "embed_cls": False, "pad_id": 0, "no_causal_mask": False, "pool_type": "argmax", "proj_type": "linear",
                                "proj_bias": False, "output_tokens": False, "quick_gelu": False},
               "pixel_decoder": {"embed_dim": 384, "num_heads": 6, "depth": 12, "ffn_layer": "swiglu", "norm_layer": "layernorm"}}}
    path = str(tmp_path / "legacy.yaml")
    with open(path, "w") as fh:
        yaml.safe_dump(doc, fh)
    ours = VTPConfig.from_vtp_yaml(path)

    class AttrDict(dict):
        def __getattr__(self, k):
            v = self[k]
            return AttrDict(v) if isinstance(v, dict) else v

    ns = load_reference()  # installs the omegaconf import stub of oracle/ref_stubs.py
    om = sys.modules["omegaconf"]
    had = getattr(om.OmegaConf, "load", None)
    om.OmegaConf.load = staticmethod(lambda p: AttrDict(yaml.safe_load(open(p))))
    try:
        ref = ns.VTPConfig.from_vtp_yaml(path)
    finally:
        if had is not None:
            om.OmegaConf.load = had
    rd = ref.to_dict()
    for k, v in ours.to_dict().items():
        assert rd[k] == v, (k, rd[k], v)
    bad = dict(doc)
    bad["data"] = {}
    with open(path, "w") as fh:
        yaml.safe_dump(bad, fh)
    with pytest.raises(KeyError):
        VTPConfig.from_vtp_yaml(path)


def test_qk_norm_branch_matches_reference():
    """use_qk_norm (attention.py:67-68,119-120) in trunk and pixel decoder: oracle == reference, outputs and gradients"""
    ns = load_reference()
    torch.manual_seed(5)
    cfg = dict(TINY)
    cfg.update(image_size=64, vision_use_qk_norm=True, decoder_use_qk_norm=True)
    m = ns.VTPModel(ns.VTPConfig(**cfg)).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "q_norm" in n or "k_norm" in n:
                p.copy_(1.0 + 0.3 * torch.randn_like(p))
    sd = {k: v.clone().requires_grad_(v.dtype == torch.float32) for k, v in m.state_dict().items()}
    assert any("q_norm.weight" in k for k in sd)
    img = torch.randn(2, 3, 64, 64)
    lat_ref = m.get_reconstruction_latents(img)
    rec_ref = m.get_latents_decoded_images(lat_ref)
    lat = O.reconstruction_latents(sd, img, 2)
    rec = O.decoder_forward(sd, lat, 2)
    torch.testing.assert_close(lat, lat_ref, rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(rec, rec_ref, rtol=2e-4, atol=2e-5)
    (rec_ref - img).abs().mean().backward()
    (rec - img).abs().mean().backward()
    params = dict(m.named_parameters())
    for k in ("trunk.blocks.0.attn.q_norm.weight", "trunk.blocks.1.attn.k_norm.weight", "pixel_decoder.blocks.0.attn.q_norm.weight",
              "trunk.blocks.0.attn.qkv.weight"):
        torch.testing.assert_close(sd[k].grad, params[k].grad, rtol=2e-3, atol=1e-6)


def test_patch_model_config_and_keys_against_the_real_class(ref_model):
    """the host half of vtp_amd.patch_model on the REAL reference instance: its PretrainedConfig translates to our VTPConfig and
    its state_dict loads strict=True into our parameter tree (the device half runs in tests/test_boundary_gpu.py)"""
    from vtp_amd import VTPModel
    from vtp_amd.patch import API_METHODS, _our_config
    cfg = _our_config(ref_model.config)
    assert cfg.vision_embed_dim == ref_model.config.vision_embed_dim and cfg.vision_depth == ref_model.config.vision_depth
    ours = VTPModel(cfg)
    ours.load_state_dict({k: v.detach() for k, v in ref_model.state_dict().items()}, strict=True)
    for name in API_METHODS:
        assert callable(getattr(ref_model, name)) and callable(getattr(ours, name)), name


@pytest.mark.parametrize("clip_feat,ae_only", [("pooled", True), ("cls", False), ("pooled", False)])
def test_clip_image_feature_variants_match_reference(clip_feat, ae_only):
    """vision_clip_feat='pooled' (mean of the patch tokens, modeling_vtp.py:269) and vision_bottleneck_ae_only=False (the CLIP
    head sees the bottlenecked features, :252-262) -- the oracle branches behind vtp_amd's torch-head path -- plus the SigLIP
    logit bias (:330-331)"""
    ns = load_reference()
    torch.manual_seed(7)
    cfg = dict(TINY)
    cfg.update(image_size=64, vision_clip_feat=clip_feat, vision_bottleneck_ae_only=ae_only, init_logit_bias=-2.5)
    m = ns.VTPModel(ns.VTPConfig(**cfg)).eval()
    sd = m.state_dict()
    img = torch.randn(2, 3, 64, 64)
    text = torch.randint(1, 500, (2, 16))
    text[:, 9] = 511
    with torch.no_grad():
        f_ref = m.get_clip_image_feature(img)
        f = O.clip_image_feature(sd, img, 2, clip_feat=clip_feat, ae_only=ae_only)
        torch.testing.assert_close(f, f_ref, rtol=1e-4, atol=1e-5)
        li, _ = m.get_clip_logits(img, text)
        t = O.clip_text_feature(sd, text, 2)
        lo = sd["logit_scale"].exp() * f @ t.T + sd["logit_bias"]
        torch.testing.assert_close(lo, li, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("pool,no_causal,ls,quick", [("first", False, None, False), ("last", False, None, False), ("argmax", True, None, False),
                                                      ("last", True, None, False), ("argmax", False, 0.1, False), ("argmax", False, None, True)])
def test_text_pooling_and_mask_variants_match_reference(pool, no_causal, ls, quick):
    """text_global_pool first / last, no_causal_mask, LayerScale and QuickGELU (text_transformer.py:213-228,285-288; block.py:388,399,
    425-426; layers/activation.py:5-12) --
    the oracle branches behind the round-4 text-tower variants, against the real class (state_dict keys / shapes of ours too)"""
    from oracle.ref_stubs import TINY, load_reference
    from vtp_amd import VTPConfig, VTPModel
    ref = load_reference()
    torch.manual_seed(5)
    kw = dict(TINY, text_pool_type=pool, text_no_causal_mask=no_causal, text_ls_init_value=ls, text_quick_gelu=quick)
    m = ref.VTPModel(ref.VTPConfig(**kw)).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith(".gamma"):  # LayerScale gammas are torch.empty until reset_parameters: give them values
                p.copy_(0.5 + torch.rand_like(p))
    sd = m.state_dict()
    assert {k: tuple(v.shape) for k, v in VTPModel(VTPConfig(**kw)).state_dict().items()} == {k: tuple(v.shape) for k, v in sd.items()}
    text = torch.randint(1, 500, (3, 16))
    text[:, 9] = 511
    with torch.no_grad():
        want = m.get_clip_text_feature(text)
        got = O.clip_text_feature(sd, text, 2, pool_type=pool, causal=not no_causal, quick_gelu=quick)
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("embed_cls,pool", [(True, "argmax"), (True, "last"), (False, "none"), (True, "none")])
def test_text_embed_cls_and_pool_none_match_reference(embed_cls, pool):
    """text_embed_cls and text_pool_type = "none" AT THE CLASS BOUNDARY (round 5).  What the reference class does with them
    (modeling_vtp.py:142-170,296-310): it re-hangs TextTransformer's parts, so embed_cls leaves a positional table / causal mask of
    context_length + 1 positions but NO cls_emb and no padding mask (text_transformer.py:340-357 is never reached) -- captions must then
    carry context_length + 1 ids (the stock length fails in `x + positional_embedding`); pool "none" returns every token's projected,
    normalised feature [B, T, D] and get_clip_logits fails in its matmul.  Pinned here: state_dict keys / shapes of ours, the oracle
    against the real class, and the error cases."""
    from oracle.ref_stubs import TINY, load_reference
    from vtp_amd import VTPConfig, VTPModel
    ref = load_reference()
    torch.manual_seed(6)
    kw = dict(TINY, text_embed_cls=embed_cls, text_pool_type=pool)
    m = ref.VTPModel(ref.VTPConfig(**kw)).eval()
    sd = m.state_dict()
    ours = VTPModel(VTPConfig(**kw))
    assert {k: tuple(v.shape) for k, v in ours.state_dict().items()} == {k: tuple(v.shape) for k, v in sd.items()}
    L = TINY["text_context_length"]
    T = L + 1 if embed_cls else L
    assert ours.config.text_num_pos == T == sd["positional_embedding"].shape[0] and not any("cls_emb" in k for k in sd)
    text = torch.randint(1, 500, (3, T))
    text[:, 9] = 511
    with torch.no_grad():
        want = m.get_clip_text_feature(text)
        got = O.clip_text_feature(sd, text, 2, pool_type=pool)
    assert want.shape == ((3, T, TINY["text_embed_dim"]) if pool == "none" else (3, TINY["text_embed_dim"]))
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5)
    with pytest.raises(RuntimeError):  # the other caption length does not fit the positional table
        m.get_clip_text_feature(torch.randint(1, 500, (3, L if embed_cls else L + 1)))
    if pool == "none":
        with pytest.raises(RuntimeError):
            m.get_clip_logits(torch.randn(3, 3, 64, 64), text)


def test_rope_train_time_augmentations_match_reference():
    """RopePositionEmbedding's shift / jitter / rescale of the patch coordinates in training mode (embeddings.py:155-171): the oracle's
    augmented tables and where the towers draw them -- the trunk inside its block loop (vision_transformer.py:228-233: new coordinates
    per block), the pixel decoder once per forward (pixel_decoder.py:144) -- against the real class consuming the same global RNG
    stream: bit-identical latents and reconstructions"""
    ref = load_reference()
    torch.manual_seed(0)
    m = ref.VTPModel(ref.VTPConfig(**TINY))
    sd = m.state_dict()
    for mod in (m.trunk.rope_embed, m.pixel_decoder.rope_embed):
        mod.shift_coords, mod.jitter_coords, mod.rescale_coords = 0.1, 1.2, 1.5
    m.train()
    img = torch.randn(2, 3, 64, 64)
    torch.manual_seed(11)
    with torch.no_grad():
        lat_ref = m.get_reconstruction_latents(img)
    torch.manual_seed(11)
    draws = [O.rope_aug_draw(0.1, 1.2, 1.5) for _ in range(TINY["vision_depth"])]
    with torch.no_grad():
        pt = O.trunk_forward(sd, img, 2, rope_aug=draws)["x_norm_patchtokens"]
        lat = pt.transpose(1, 2).reshape(2, -1, 4, 4)
        plain = O.reconstruction_latents(sd, img, 2)
    assert torch.equal(lat, lat_ref) and not torch.allclose(plain, lat_ref, atol=1e-5)
    torch.manual_seed(12)
    with torch.no_grad():
        rec_ref = m.get_latents_decoded_images(lat_ref)
    torch.manual_seed(12)
    d = O.rope_aug_draw(0.1, 1.2, 1.5)
    with torch.no_grad():
        rec = O.decoder_forward(sd, lat_ref, 2, rope_aug=d)
    assert torch.equal(rec, rec_ref)
    m.eval()  # evaluation mode: no augmentation
    with torch.no_grad():
        torch.testing.assert_close(O.reconstruction_latents(sd, img, 2), m.get_reconstruction_latents(img), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("vis,dec", [("mlp", "mlp"), ("swiglu64", "swiglu")])
def test_ffn_layer_variants_match_reference(vis, dec):
    """ffn_layer = "mlp" (GELU Mlp, ffn.py:21-48) and the aligned SwiGLU widths (vision_transformer.py:22-28): parameter names /
    shapes of vtp_amd.VTPModel against the real class, and the oracle's FFN branch against its encode -> decode"""
    from vtp_amd import VTPConfig, VTPModel
    ref = load_reference()
    kw = dict(TINY, vision_ffn_layer=vis, decoder_ffn_layer=dec)
    torch.manual_seed(9)
    m = ref.VTPModel(ref.VTPConfig(**kw)).eval()
    sd = m.state_dict()
    ours = VTPModel(VTPConfig(**kw)).state_dict()
    assert {k: tuple(v.shape) for k, v in ours.items()} == {k: tuple(v.shape) for k, v in sd.items()}
    img = torch.randn(2, 3, 64, 64)
    with torch.no_grad():
        lat_ref = m.get_reconstruction_latents(img)
        rec_ref = m.get_latents_decoded_images(lat_ref)
        lat = O.reconstruction_latents(sd, img, 2)
        rec = O.decoder_forward(sd, lat, 2)
    torch.testing.assert_close(lat, lat_ref, rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(rec, rec_ref, rtol=2e-4, atol=2e-5)
