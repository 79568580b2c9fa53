"""Generates tests/golden/vtp_tiny.safetensors from the REAL reference (run in the authoring
container only: needs /root/reference).  TEST INFRASTRUCTURE -- not product code.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

Contents: the full state_dict of a seeded random-init ``VTPModel(VTPConfig(**TINY))`` (prefixed
``sd.``), seeded inputs (image, text ids) and the reference's own outputs in fp32 for every
drop-in entry point of the hot path (latents, decoded image, clip features, logits), plus the
gradient of an L1 reconstruction loss w.r.t. a handful of parameters (reference autograd).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safetensors.torch import save_file

from oracle.ref_stubs import TINY, load_reference

GRAD2_KEYS = [  # gradients of (L1 + CLIP) -- the backward through the text tower / heads is the reference's autograd
    "logit_scale", "visual_proj.weight", "text_projection", "ln_final.weight", "ln_final.bias", "positional_embedding",
    "token_embedding.weight", "text_transformer.resblocks.0.attn.in_proj_weight", "text_transformer.resblocks.0.attn.in_proj_bias",
    "text_transformer.resblocks.1.attn.out_proj.weight", "text_transformer.resblocks.0.mlp.c_fc.weight",
    "text_transformer.resblocks.1.mlp.c_proj.bias", "text_transformer.resblocks.1.ln_2.weight",
    "trunk.cls_token", "trunk.blocks.1.attn.qkv.weight", "trunk.norm.weight", "trunk.patch_embed.proj.weight",
]

GRAD_KEYS = [
    "trunk.patch_embed.proj.weight", "trunk.cls_token", "trunk.blocks.0.attn.qkv.weight",
    "trunk.blocks.0.attn.qkv.bias", "trunk.blocks.1.mlp.w1.weight", "trunk.blocks.1.mlp.w3.bias",
    "trunk.blocks.0.norm1.weight", "trunk.norm.weight", "trunk.feature_bottleneck.weight",
    "pixel_decoder.proj_in.weight", "pixel_decoder.blocks.0.norm1.bias", "pixel_decoder.blocks.1.attn.proj.weight",
    "pixel_decoder.blocks.0.mlp.w2.weight", "pixel_decoder.norm.weight", "pixel_decoder.proj_out.weight",
    "pixel_decoder.proj_out.bias",
]


def main(out_path: str):
    ns = load_reference()
    torch.manual_seed(0)
    model = ns.VTPModel(ns.VTPConfig(**TINY)).eval()
    # make zero-init tensors non-trivial so parity tests see every bias / mask token
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.ndim <= 1 and n != "logit_scale" and float(p.abs().sum()) == 0.0:
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
            if n.endswith("norm1.weight") or n.endswith("norm2.weight") or n.endswith("norm.weight") \
                    or n.endswith("ln_1.weight") or n.endswith("ln_2.weight") or n == "ln_final.weight":
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            if n == "trunk.mask_token":
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
    T = TINY["text_context_length"]
    img = torch.randn(3, 3, TINY["image_size"], TINY["image_size"], generator=g)
    text = torch.zeros(3, T, dtype=torch.long)
    for b in range(3):
        ln = int(torch.randint(4, T - 1, (1,), generator=g))
        text[b, 0] = TINY["text_vocab_size"] - 2
        text[b, 1:ln] = torch.randint(1, TINY["text_vocab_size"] - 3, (ln - 1,), generator=g)
        text[b, ln] = TINY["text_vocab_size"] - 1  # EOT = max id -> argmax pooling hits it
    out = {"in.image": img, "in.text": text}
    with torch.no_grad():
        lat = model.get_reconstruction_latents(img)
        out["out.latents"] = lat.contiguous()
        out["out.reconstruction"] = model.get_latents_decoded_images(lat).contiguous()
        out["out.clip_image_feature"] = model.get_clip_image_feature(img)
        out["out.clip_text_feature"] = model.get_clip_text_feature(text)
        li, _ = model.get_clip_logits(img, text)
        out["out.clip_logits"] = li.contiguous()
        feat = model.get_last_layer_feature(img, use_bottleneck=False)
        out["out.last_cls"] = feat["cls_token"].contiguous()
        out["out.last_patch"] = feat["patch_tokens"].contiguous()
    # reference autograd: L1 reconstruction loss (the loss itself is our spec; the backward through the
    # model is the reference's)
    model.zero_grad()
    r = model(image=img, forward_type="rec")
    loss = (r["reconstructed_image"] - r["target_image"]).abs().mean()
    loss.backward()
    out["out.rec_l1_loss"] = loss.detach().reshape(1)
    params = dict(model.named_parameters())
    for k in GRAD_KEYS:
        out["grad." + k] = params[k].grad.detach().clone().contiguous()
    # L1 + CLIP (OpenCLIP ClipLoss on the reference's own 'clip' forward outputs); loss spec is ours, backward is the reference's
    model.zero_grad()
    r = model(image=img, forward_type="rec")
    l1 = (r["reconstructed_image"] - r["target_image"]).abs().mean()
    c = model(image=img, text=text, forward_type="clip")
    logits = c["logit_scale"] * c["image_features"] @ c["text_features"].T
    labels = torch.arange(logits.shape[0])
    lclip = 0.5 * (torch.nn.functional.cross_entropy(logits, labels) + torch.nn.functional.cross_entropy(logits.T, labels))
    (l1 + lclip).backward()
    out["out.clip_loss"] = lclip.detach().reshape(1)
    for k in GRAD2_KEYS:
        out["grad2." + k] = params[k].grad.detach().clone().contiguous()
    for k, v in model.state_dict().items():
        out["sd." + k] = v.detach().clone().contiguous()
    save_file(out, out_path)
    print("wrote", out_path, sum(v.numel() * v.element_size() for v in out.values()) / 1e6, "MB")


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    main(os.path.join(os.path.dirname(here), "tests", "golden", "vtp_tiny.safetensors"))
