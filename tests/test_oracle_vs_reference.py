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
