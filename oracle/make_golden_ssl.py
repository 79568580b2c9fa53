"""Generates tests/golden/vtp_tiny_ssl.safetensors from the REAL reference's legacy training class
(vtp/models/vtp.py `VTP`, forward_type='ssl').  Authoring container only (needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_ssl.py

Contents: state_dict of a seeded tiny legacy VTP (trunk, dino_head, teacher_trunk, teacher_dino_head; the teacher is
perturbed so that it differs from the student), a seeded ssl_dict (2 global 64x64 crops + 4 local 32x32 crops per
image, iBOT masks), the reference's teacher / student outputs (vtp.py:446-448,479-484), and -- for OUR loss spec
(oracle.vtp_oracle.ssl_loss) -- the reference-autograd gradients of that loss through the reference model."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safetensors.torch import save_file

from oracle import vtp_oracle as O
from oracle.ref_stubs import load_reference

SSL_CFG = dict(embed_dim=128, depth=2, heads=2, K=512, hidden=128, bott=64, B=3, n_local=4, R=64, r=32)
SSL_GRAD_KEYS = ["dino_head.mlp.0.weight", "dino_head.mlp.2.bias", "dino_head.mlp.4.weight", "dino_head.last_layer.weight_g",
                 "dino_head.last_layer.weight_v", "trunk.mask_token", "trunk.cls_token", "trunk.patch_embed.proj.weight",
                 "trunk.blocks.0.attn.qkv.weight", "trunk.blocks.1.mlp.w2.weight", "trunk.norm.weight"]


def legacy_config(ns, c):
    class AD(ns.DictConfig):
        def __init__(s, d):
            super().__init__({k: AD(v) if isinstance(v, dict) else v for k, v in d.items()})
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__
    return AD(dict(
        data=dict(image_size=c["R"]),
        training=dict(train_clip=False, train_dinov2=True, train_reconstruction=False, cast_dtype=None, init_logit_scale=None,
                      init_logit_bias=None, nonscalar_logit_scale=False, clip_output_dict=True, clip_drop_rate=0.0,
                      ssl_drop_rate=0.0, rec_drop_rate=0.0),
        vtp_model=dict(
            vision_encoder=dict(model_type="dinov3", patch_size=16, embed_dim=c["embed_dim"], depth=c["depth"], num_heads=c["heads"],
                                mlp_ratio=4.0, ffn_layer="swiglu", norm_type="rmsnorm", init_values=None,
                                vit_feature_bottleneck=64, bottleneck_ae_only=True, clip_feat="cls"),
            text_encoder=dict(embed_dim=c["embed_dim"]),
            dino_head=dict(out_dim=c["K"], nlayers=3, hidden_dim=c["hidden"], bottleneck_dim=c["bott"]),
            pixel_decoder=dict(model_type="dinov3"))))


def make_ssl_batch(c, seed=7):
    g = torch.Generator().manual_seed(seed)
    B, hw = c["B"], (c["R"] // 16) ** 2
    global_crops = torch.randn(2 * B, 3, c["R"], c["R"], generator=g)
    local_crops = torch.randn(c["n_local"] * B, 3, c["r"], c["r"], generator=g)
    masks = torch.rand(2 * B, hw, generator=g) < 0.35
    masks[1] = False  # an un-masked image
    idx = masks.flatten().nonzero().flatten()
    return dict(global_crops=global_crops, n_global_crops=2, mask_indices_list=idx, n_masked_patches=int(idx.numel()),
                upperbound=int(idx.numel()) + 5, local_crops=local_crops, masks=masks)


def main(out_path):
    ns = load_reference()
    c = SSL_CFG
    torch.manual_seed(0)
    model = ns.VTP(vtp_config=legacy_config(ns, c))
    model.train()  # forward_ssl_learning is a training-time path (stochastic depth rates are 0)
    g = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.startswith("teacher_"):
                p.add_(0.01 * torch.randn(p.shape, generator=g))
            elif p.ndim <= 1 or n.endswith("mask_token"):
                p.add_(0.02 * torch.randn(p.shape, generator=g))
    batch = make_ssl_batch(c)
    t_out, s_out = model(forward_type="ssl", ssl_dict=batch)
    out = {"in.global_crops": batch["global_crops"], "in.local_crops": batch["local_crops"],
           "in.masks": batch["masks"].to(torch.uint8)}
    for k, v in t_out.items():
        if torch.is_tensor(v):
            out["teacher." + k] = v.detach().contiguous()
    for k, v in s_out.items():
        out["student." + k] = v.detach().contiguous()
    # our loss spec on the reference's outputs; gradients via the reference's autograd
    center_d = 0.05 * torch.randn(c["K"], generator=g)
    center_i = 0.05 * torch.randn(c["K"], generator=g)
    loss = O.ssl_loss(t_out, s_out, batch["masks"], center_d, center_i, n_local=c["n_local"])
    loss.backward()
    out["in.center_dino"], out["in.center_ibot"] = center_d, center_i
    out["out.ssl_loss"] = loss.detach().reshape(1)
    params = dict(model.named_parameters())
    for k in SSL_GRAD_KEYS:
        out["grad." + k] = params[k].grad.detach().clone().contiguous()
    for k, v in model.state_dict().items():
        out["sd." + k] = v.detach().clone().contiguous()
    save_file(out, out_path)
    print("wrote", out_path, sum(v.numel() * v.element_size() for v in out.values()) / 1e6, "MB; loss", float(loss))


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    main(os.path.join(os.path.dirname(here), "tests", "golden", "vtp_tiny_ssl.safetensors"))
