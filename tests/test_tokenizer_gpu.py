"""f3: VTP_Tokenizer + latent shards on the MI355X (generation/tokenizer/vtp_tokenizer.py, generation/tools/
extract_features_vtp.py): byte kernels bit-exact against the restated torch chain, encode / decode equal to the model API,
shard files in the reference's layout."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


@pytest.fixture(scope="module")
def tok(golden_sd):
    from oracle.ref_stubs import TINY
    from vtp_amd import VTPConfig, VTPModel, VTP_Tokenizer
    m = VTPModel(VTPConfig(**TINY))
    m.load_state_dict(golden_sd, strict=True)
    return VTP_Tokenizer(m, img_size=TINY["image_size"], normalize_type="imagenet")


@pytest.mark.parametrize("norm", ["imagenet", "half"])
@pytest.mark.parametrize("shape", [(3, 32, 32), (2, 64, 48), (1, 256, 256), (5, 16, 20)])
def test_byte_kernels_bit_exact(norm, shape):
    from oracle import tokenizer_oracle as TO
    from vtp_amd import ops
    from vtp_amd.tokenizer import NORMALIZE_HALF, NORMALIZE_IMAGENET
    cfg = NORMALIZE_IMAGENET if norm == "imagenet" else NORMALIZE_HALF
    B, H, W = shape
    rng = np.random.default_rng(B * 1000 + H)
    u8 = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    u8[0, 0, :4] = [[0, 0, 0], [255, 255, 255], [1, 254, 127], [128, 2, 253]]
    for flip in (False, True):
        ref = TO.to_tensor_normalize(u8, cfg["mean"], cfg["std"], flip)
        out = torch.empty(B, 3, H, W, dtype=torch.float32, device=DEV)
        ops.u8_to_images(torch.from_numpy(u8).to(DEV), out, cfg["mean"], cfg["std"], flip)
        assert torch.equal(out.cpu(), ref), (norm, shape, flip)
    # decode tail: values across and beyond the representable range, exact integers and just-below-integer cases
    dec = torch.randn(B, 3, H, W, generator=torch.Generator().manual_seed(7)) * 1.5
    dec[0, :, 0, :4] = torch.tensor([-5.0, 5.0, 0.0, 1e-7])
    inv_mean = [-m / s for m, s in zip(cfg["mean"], cfg["std"])]
    inv_std = [1.0 / s for s in cfg["std"]]
    ref_u8 = TO.decode_tail(dec, cfg["mean"], cfg["std"])
    out_u8 = torch.empty(B, H, W, 3, dtype=torch.uint8, device=DEV)
    ops.images_to_u8(dec.to(DEV), out_u8, inv_mean, inv_std)
    assert np.array_equal(out_u8.cpu().numpy(), ref_u8), (norm, shape)
    # round trip: bytes -> normalised -> bytes is the identity up to the float floor (|diff| <= 1, and exact for >= 99.9 %)
    back = torch.empty(B, H, W, 3, dtype=torch.uint8, device=DEV)
    x = torch.empty(B, 3, H, W, dtype=torch.float32, device=DEV)
    ops.u8_to_images(torch.from_numpy(u8).to(DEV), x, cfg["mean"], cfg["std"], False)
    ops.images_to_u8(x, back, inv_mean, inv_std)
    diff = np.abs(back.cpu().numpy().astype(np.int32) - u8.astype(np.int32))
    assert diff.max() <= 1


def test_tokenizer_encode_decode_matches_model_api(tok, golden):
    from oracle import tokenizer_oracle as TO
    img = golden["in.image"]
    lat = tok.encode_images(img)
    assert lat.device.type == "cpu" and lat.dtype == torch.float32
    with torch.no_grad():
        ref_lat = tok.model.get_reconstruction_latents(img.to(DEV)).cpu()
    assert torch.equal(lat, ref_lat)
    assert tok._current_img_h == img.shape[2] and tok.latent_size == tok.img_size // 16 and tok.embed_dim == lat.shape[1]
    imgs = tok.decode_to_images(lat)
    assert imgs.dtype == np.uint8 and imgs.shape == (img.shape[0], img.shape[2], img.shape[3], 3)
    with torch.no_grad():
        dec = tok.model.get_latents_decoded_images(lat.to(DEV)).float().cpu()
    assert np.array_equal(imgs, TO.decode_tail(dec, tok.norm_mean, tok.norm_std))
    # golden reconstruction of the reference (oracle/make_golden.py) through the reference's own tail: byte images agree except
    # where bf16 noise crosses an integer boundary
    ref_imgs = TO.decode_tail(golden["out.reconstruction"], tok.norm_mean, tok.norm_std)
    if ref_imgs is not None:
        d = np.abs(imgs.astype(np.int32) - ref_imgs.astype(np.int32))
        print("byte diff vs golden: max", d.max(), "mean", d.mean())
        assert d.mean() < 1.5
    with pytest.raises(ValueError):
        tok._setup_normalization("bogus")


def test_extract_features_shards_and_stats(tok, tmp_path):
    from safetensors import safe_open
    from safetensors.torch import load_file
    from oracle import tokenizer_oracle as TO
    from vtp_amd.tokenizer import extract_features, shard_name
    S = tok.img_size
    rng = np.random.default_rng(5)
    samples = [(rng.integers(0, 256, (S, S, 3), dtype=np.uint8), i % 7) for i in range(11)]
    out = str(tmp_path / "lat")
    paths = extract_features(tok, samples, out, batch_size=2, batches_per_shard=3)
    assert [os.path.basename(p) for p in paths] == [shard_name(0, 0), shard_name(0, 1)]
    d0, d1 = load_file(paths[0]), load_file(paths[1])
    assert d0["latents"].shape[0] == 6 and d1["latents"].shape[0] == 5
    assert set(d0) == {"latents", "latents_flip", "labels"} and d0["labels"].dtype == torch.int64
    assert d0["labels"].tolist() + d1["labels"].tolist() == [y for _, y in samples]
    with safe_open(paths[0], "pt") as f:
        assert f.metadata() == {"total_size": "6", "dtype": "torch.float32", "device": "cpu"}
    # content: the same latents as a direct encode of the normalised (flipped) images
    u8 = np.stack([s for s, _ in samples[:2]])
    x = TO.to_tensor_normalize(u8, tok.norm_mean, tok.norm_std)
    assert torch.equal(d0["latents"][:2], tok.encode_images(x))
    assert torch.equal(d0["latents_flip"][:2], tok.encode_images(x.flip(-1)))
    st = torch.load(os.path.join(out, "latents_stats.pt"))
    ref = TO.latent_stats(torch.cat([d0["latents"], d1["latents"]]))
    assert st["mean"].shape == (1, tok.embed_dim, 1, 1)
    assert torch.allclose(st["mean"], ref["mean"], rtol=1e-5, atol=1e-6) and torch.allclose(st["std"], ref["std"], rtol=1e-5, atol=1e-6)


def test_img_transform_matches_device_path(tok):
    """VTP_Tokenizer.img_transform (the reference's dataset transform, vtp_tokenizer.py:74-81) as a host callable == crop_to_u8 +
    images_from_u8 on the device, bit for bit; transform_inv undoes the normalisation"""
    from PIL import Image
    rng = np.random.default_rng(3)
    pil = Image.fromarray(rng.integers(0, 256, (90, 130, 3), dtype=np.uint8))
    size = tok.img_size
    host = tok.img_transform(p_hflip=0)(pil)
    dev = tok.images_from_u8(tok.crop_to_u8(pil, size)[None])[0].cpu()
    assert host.shape == (3, size, size) and torch.equal(host, dev)
    flipped = tok.img_transform(p_hflip=1.0)(pil)
    assert torch.equal(flipped, tok.images_from_u8(tok.crop_to_u8(pil, size)[None], flip=True)[0].cpu())
    back = tok.transform_inv(host)
    u8 = torch.from_numpy(tok.crop_to_u8(pil, size).copy()).permute(2, 0, 1).float() / 255
    assert float((back - u8).abs().max()) < 1e-6
