"""Generates tests/golden/vtp_tiny_legacy.safetensors from the REAL reference's legacy training class (vtp/models/vtp.py
`VTP`) configured with ALL THREE objectives (train_clip + train_dinov2 + train_reconstruction).  Authoring container only.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_legacy.py

Contents: the legacy-layout state_dict (`proj`, `teacher_proj`, `transformer.resblocks.*`, `teacher_trunk.*`, ... -- the
checkpoint layout vtp_amd.VTP must load with strict=True), seeded inputs, and the reference outputs of
VTP.forward(forward_type='clip' / 'rec'), encode_image / encode_text (un-normalised) and get_logits (vtp.py:275-360,487-512)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safetensors.torch import save_file

from oracle.ref_stubs import load_reference

CFG = dict(embed_dim=128, depth=2, heads=2, K=512, hidden=128, bott=64, R=64, text_layers=2, text_heads=2, vocab=512, ctx=16,
           dec_depth=2, dec_heads=2)


def legacy_config(ns, c):
    class AD(ns.DictConfig):
        def __init__(s, d):
            super().__init__({k: AD(v) if isinstance(v, dict) else v for k, v in d.items()})
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__
    return AD(dict(
        data=dict(image_size=c["R"]),
        training=dict(train_clip=True, train_dinov2=True, train_reconstruction=True, cast_dtype=None, init_logit_scale=None,
                      init_logit_bias=None, nonscalar_logit_scale=False, clip_output_dict=True, clip_drop_rate=0.0,
                      ssl_drop_rate=0.0, rec_drop_rate=0.0),
        vtp_model=dict(
            vision_encoder=dict(model_type="dinov3", patch_size=16, embed_dim=c["embed_dim"], depth=c["depth"], num_heads=c["heads"],
                                mlp_ratio=4.0, ffn_layer="swiglu", norm_type="rmsnorm", init_values=None,
                                vit_feature_bottleneck=64, bottleneck_ae_only=True, clip_feat="cls"),
            text_encoder=dict(embed_dim=c["embed_dim"], context_length=c["ctx"], vocab_size=c["vocab"], heads=c["text_heads"],
                              layers=c["text_layers"], mlp_ratio=4.0, ls_init_value=None, embed_cls=False, no_causal_mask=False,
                              pad_id=0, pool_type="argmax", proj_type="linear", proj_bias=False, output_tokens=False,
                              quick_gelu=False, norm_kwargs={}, act_kwargs=None),
            dino_head=dict(out_dim=c["K"], nlayers=3, hidden_dim=c["hidden"], bottleneck_dim=c["bott"]),
            pixel_decoder=dict(model_type="dinov3", embed_dim=c["embed_dim"], depth=c["dec_depth"], num_heads=c["dec_heads"]))))


def main(out_path):
    ns = load_reference()
    c = CFG
    torch.manual_seed(0)
    cfg = legacy_config(ns, c)
    try:
        model = ns.VTP(vtp_config=cfg)
    except TypeError:  # DinoV3PixelDecoder(**decoder_kwargs) may reject the `model_type` key
        raise
    model.train()
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.startswith("teacher_"):
                p.add_(0.01 * torch.randn(p.shape, generator=g))
            elif p.ndim <= 1 and n != "logit_scale":
                p.add_(0.02 * torch.randn(p.shape, generator=g))
    img = torch.randn(3, 3, c["R"], c["R"], generator=g)
    txt = torch.randint(1, c["vocab"] - 2, (3, c["ctx"]), generator=g)
    txt[:, 0] = c["vocab"] - 2
    for b, ln in enumerate((5, 9, 15)):
        txt[b, ln] = c["vocab"] - 1
        txt[b, ln + 1:] = 0
    out = {"in.image": img, "in.text": txt}
    with torch.no_grad():
        o = model(image=img, text=txt, forward_type="clip")
        out["clip.image_features"], out["clip.text_features"], out["clip.logit_scale"] = \
            o["image_features"].contiguous(), o["text_features"].contiguous(), o["logit_scale"].reshape(1)
        r = model(reconstruction_image=img, forward_type="rec")
        out["rec.reconstructed_image"] = r["reconstructed_image"].contiguous()
        out["enc.image"] = model.encode_image(img).contiguous()
        out["enc.text"] = model.encode_text(txt).contiguous()
        li, lt = model.get_logits(img, txt)
        out["logits.image"] = li.contiguous()
    for k, v in model.state_dict().items():
        out["sd." + k] = v.detach().clone().contiguous()
    save_file(out, out_path)
    print("wrote", out_path, sum(v.numel() * v.element_size() for v in out.values()) / 1e6, "MB")
    print(sorted(k for k in out if k.startswith("sd.") and not k.startswith(("sd.trunk.blocks", "sd.teacher_trunk", "sd.transformer.", "sd.pixel_decoder.blocks")))[:60])


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    main(os.path.join(os.path.dirname(here), "tests", "golden", "vtp_tiny_legacy.safetensors"))
