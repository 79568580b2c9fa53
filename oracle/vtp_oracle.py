"""CPU oracle for the VTP hot path -- TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32 unless stated) *restatement* of the reference algorithm
for the path named by BASELINE.json's ``north_star``: ViT trunk (patch-embed, RoPE attention,
SwiGLU blocks), bottleneck, pixel decoder, CLIP text tower + heads, and the (unshipped) loss heads.
Every function cites the reference file:line it follows (paths relative to /root/reference).

Rules (task §③):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
    this module -- the product path (``vtp_amd/``) must never import it and fails loudly when the
    HIP library is missing;
  * parity pinning: the restatement is checked (a) against the *real* reference imported from
    /root/reference in the authoring container (``tests/test_oracle_vs_reference.py``; skipped
    where the tree is absent) and (b) against golden fixtures generated from the real reference
    (``tests/golden/*.safetensors`` by ``oracle/make_golden.py``) -- these travel to the GPU box.
  * the loss heads (L1 reconstruction, CLIP InfoNCE) are NOT in the reference:
    their parity is **unpinned**; the definitions here follow OpenCLIP's ClipLoss and a plain L1.

The functions operate on a reference-format ``state_dict`` (exact checkpoint keys of
``VTPModel``, vtp/models/vtp_hf/modeling_vtp.py:51) so the same weights feed the oracle, the real
reference and the HIP path.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------------
# RoPE tables -- vtp/models/layers/embeddings.py:131-195
# --------------------------------------------------------------------------------------------
def rope_periods(head_dim: int = 64, base: float = 100.0, dtype=torch.bfloat16) -> Tensor:
    """periods = base ** (2*arange(D_head/4) / (D_head/2)), computed IN ``dtype`` (bf16 by default,
    vision_transformer.py:74).  embeddings.py:182-195."""
    return base ** (2 * torch.arange(head_dim // 4, dtype=dtype) / (head_dim // 2))


def rope_aug_draw(shift: Optional[float], jitter: Optional[float], rescale: Optional[float], dtype=torch.bfloat16, device="cpu",
                  generator=None) -> Dict[str, Optional[Tensor]]:
    """One call's random draws of RopePositionEmbedding.forward in training mode (embeddings.py:155-171), with the reference's own
    torch calls in its order -- shift: U(-s, s) per axis; jitter: exp(U(-ln j, ln j)) per axis; rescale: exp(U(-ln r, ln r)), one
    value -- drawn in `dtype` (the rope dtype, bf16).  With generator=None the GLOBAL generator is consumed exactly as the reference
    consumes it (the pinning test seeds it the same way on both sides)."""
    import numpy as np
    dd = {"dtype": dtype, "device": device}
    out: Dict[str, Optional[Tensor]] = {"shift": None, "jitter": None, "rescale": None}
    if shift is not None:
        out["shift"] = torch.empty(2, **dd).uniform_(-shift, shift, generator=generator)
    if jitter is not None:
        jm = np.log(jitter)
        out["jitter"] = torch.empty(2, **dd).uniform_(-jm, jm, generator=generator).exp()
    if rescale is not None:
        rm = np.log(rescale)
        out["rescale"] = torch.empty(1, **dd).uniform_(-rm, rm, generator=generator).exp()
    return out


def rope_table(H: int, W: int, periods: Tensor, aug: Optional[Dict[str, Optional[Tensor]]] = None) -> Tuple[Tensor, Tensor]:
    """(sin, cos), each [H*W, D_head], in periods.dtype -- 'separate' coordinate normalisation.
    embeddings.py:131-180 (exact op order, so bf16 rounding matches).  aug = rope_aug_draw(...): the train-time coordinate
    augmentations (:155-171) with GIVEN draws -- shift added, then the per-axis jitter and the global rescale multiplied in place."""
    dd = {"dtype": periods.dtype, "device": periods.device}  # device-aware: the GPU-autocast comparator of the tests
    coords_h = torch.arange(0.5, H, **dd) / H
    coords_w = torch.arange(0.5, W, **dd) / W
    coords = torch.stack(torch.meshgrid(coords_h, coords_w, indexing="ij"), dim=-1)
    coords = coords.flatten(0, 1)
    coords = 2.0 * coords - 1.0
    if aug is not None:
        if aug.get("shift") is not None:
            coords += aug["shift"].to(**dd)[None, :]
        if aug.get("jitter") is not None:
            coords *= aug["jitter"].to(**dd)[None, :]
        if aug.get("rescale") is not None:
            coords *= aug["rescale"].to(**dd)
    angles = 2 * math.pi * coords[:, :, None] / periods[None, None, :]
    angles = angles.flatten(1, 2)
    angles = angles.tile(2)
    return torch.sin(angles), torch.cos(angles)


def rope_rotate_half(x: Tensor) -> Tensor:
    """attention.py:12-16."""
    x1, x2 = x.chunk(2, dim=-1)
    return torch.cat([-x2, x1], dim=-1)


def apply_rope(q: Tensor, k: Tensor, sin: Tensor, cos: Tensor) -> Tuple[Tensor, Tensor]:
    """attention.py:70-89.  q,k: [B,h,N,d].  All arithmetic in sin.dtype (bf16): q and k -- *including
    the un-rotated prefix (cls) rows* -- are rounded to bf16 and cast back."""
    qd, kd, rd = q.dtype, k.dtype, sin.dtype
    q = q.to(rd)
    k = k.to(rd)
    prefix = q.shape[-2] - sin.shape[-2]
    assert prefix >= 0

    def rot(x):
        return (x * cos) + (rope_rotate_half(x) * sin)  # attention.py:19-23

    q = torch.cat((q[:, :, :prefix], rot(q[:, :, prefix:])), dim=-2)
    k = torch.cat((k[:, :, :prefix], rot(k[:, :, prefix:])), dim=-2)
    return q.to(qd), k.to(kd)


# --------------------------------------------------------------------------------------------
# norms -- vtp/models/layers/normalization.py
# --------------------------------------------------------------------------------------------
def rmsnorm(x: Tensor, w: Tensor, eps: float = 1e-5) -> Tensor:
    """normalization.py:17-22."""
    xf = x.float()
    return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).type_as(x) * w


def layernorm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    """nn.LayerNorm(eps=1e-6) for the decoder (vision_transformer.py:30-34); eps=1e-5 for the text
    tower (normalization.py:25-31, nn.LayerNorm default)."""
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


# --------------------------------------------------------------------------------------------
# blocks
# --------------------------------------------------------------------------------------------
def _plain_linear(x: Tensor, w: Tensor, b: Optional[Tensor], site: str) -> Tensor:
    return F.linear(x, w, b)


def self_attention(x: Tensor, sd: Dict[str, Tensor], pre: str, num_heads: int,
                   rope: Optional[Tuple[Tensor, Tensor]], lin=_plain_linear) -> Tensor:
    """SelfAttention.forward / compute_attention -- attention.py:91-96,110-126.
    lin(x, w, b, site): the linear map; F.linear unless a test substitutes a simulated low-precision GEMM (oracle/fp8_oracle.py)."""
    B, N, C = x.shape
    qkv = lin(x, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"], pre + "qkv")
    qkv = qkv.reshape(B, N, 3, num_heads, C // num_heads)
    q, k, v = torch.unbind(qkv, 2)
    q, k, v = [t.transpose(1, 2) for t in (q, k, v)]
    if pre + "q_norm.weight" in sd:  # use_qk_norm: RMSNorm(head_dim) on q and k (attention.py:67-68,119-120)
        q = rmsnorm(q, sd[pre + "q_norm.weight"])
        k = rmsnorm(k, sd[pre + "k_norm.weight"])
    if rope is not None:
        q, k = apply_rope(q, k, rope[0], rope[1])
    if q.dtype != v.dtype:  # under autocast the fp32 norm weight promotes q, k; SDPA is an autocast op and casts them back
        q, k = q.to(v.dtype), k.to(v.dtype)
    o = F.scaled_dot_product_attention(q, k, v)  # scale 1/sqrt(d), no mask  (attention.py:124)
    o = o.transpose(1, 2).reshape(B, N, C)
    return lin(o, sd[pre + "proj.weight"], sd[pre + "proj.bias"], pre + "proj")


def swiglu_ffn(x: Tensor, sd: Dict[str, Tensor], pre: str, lin=_plain_linear) -> Tensor:
    """SwiGLUFFN.forward -- ffn.py:77-81; with ffn_layer = "mlp" the block holds Mlp (fc1 -> nn.GELU -> fc2, ffn.py:21-48; dropout 0)."""
    if pre + "fc1.weight" in sd:
        h = F.gelu(lin(x, sd[pre + "fc1.weight"], sd[pre + "fc1.bias"], pre + "fc1"))
        return lin(h, sd[pre + "fc2.weight"], sd[pre + "fc2.bias"], pre + "fc2")
    x1 = lin(x, sd[pre + "w1.weight"], sd[pre + "w1.bias"], pre + "w1")
    x2 = lin(x, sd[pre + "w2.weight"], sd[pre + "w2.bias"], pre + "w2")
    return lin(F.silu(x1) * x2, sd[pre + "w3.weight"], sd[pre + "w3.bias"], pre + "w3")


def vit_block(x: Tensor, sd: Dict[str, Tensor], pre: str, num_heads: int, rope, norm: str, drop=None, lin=_plain_linear) -> Tensor:
    """SelfAttentionBlock._forward -- block.py:207-233.  Default: the eval / drop_ratio == 0 branch (:290-296); LayerScale
    (misc.py:24-25) is applied when the checkpoint has `ls1.gamma` / `ls2.gamma` (Identity otherwise, block.py:173,185).
    drop = (idx1, alpha1, idx2, alpha2): the training branch with stochastic depth for GIVEN image subsets (the reference
    draws them with torch.randperm, get_branges_scales block.py:20-118): the residual branch runs on x[idx] and is added back
    with torch.index_add(..., alpha=batch / kept) (:213-232)."""
    def nrm(t, name):
        if norm == "rmsnorm":
            return rmsnorm(t, sd[pre + name + ".weight"], 1e-5)
        return layernorm(t, sd[pre + name + ".weight"], sd[pre + name + ".bias"], 1e-6)

    ls1 = sd.get(pre + "ls1.gamma")
    ls2 = sd.get(pre + "ls2.gamma")
    f1 = (lambda t: t * ls1) if ls1 is not None else (lambda t: t)
    f2 = (lambda t: t * ls2) if ls2 is not None else (lambda t: t)
    if drop is None:
        x = x + f1(self_attention(nrm(x, "norm1"), sd, pre + "attn.", num_heads, rope, lin))
        x = x + f2(swiglu_ffn(nrm(x, "norm2"), sd, pre + "mlp.", lin))
        return x
    idx1, a1, idx2, a2 = drop
    r1 = self_attention(nrm(x[idx1], "norm1"), sd, pre + "attn.", num_heads, rope)
    # .to(x.dtype): a no-op in fp32; under bf16 autocast the branch output is bf16 while the stream is fp32 and index_add would
    # raise (the reference as written cannot run this branch under autocast with an fp32 stream) -- needed for the tests' E_ref
    x = torch.index_add(x, 0, idx1, f1(r1).to(x.dtype), alpha=a1)
    r2 = swiglu_ffn(nrm(x[idx2], "norm2"), sd, pre + "mlp.")
    return torch.index_add(x, 0, idx2, f2(r2).to(x.dtype), alpha=a2)


def _depth(sd: Dict[str, Tensor], pre: str) -> int:
    n = 0
    while f"{pre}{n}.norm1.weight" in sd or f"{pre}{n}.ln_1.weight" in sd:
        n += 1
    return n


# --------------------------------------------------------------------------------------------
# trunk -- vtp/models/encoders/vision_transformer.py + vision_transformer_bottleneck.py
# --------------------------------------------------------------------------------------------
def patch_embed(img: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """PatchEmbed.forward -- embeddings.py:61-70: conv k=s=16 -> [B, h*w, D] token order (y, x)."""
    x = F.conv2d(img, w, b, stride=w.shape[-1])
    return x.flatten(2).transpose(1, 2)


def trunk_forward(sd: Dict[str, Tensor], img: Tensor, num_heads: int, use_bottleneck: bool = True,
                  masks: Optional[Tensor] = None, pre: str = "trunk.", drop=None, lin=_plain_linear, rope_aug=None) -> Dict[str, Tensor]:
    """DinoVisionTransformerWithBottleneck.forward(is_training=True) for ONE resolution --
    vision_transformer.py:189-264, vision_transformer_bottleneck.py:48-79.
    rope_aug = [rope_aug_draw(...) per block]: the trunk calls rope_embed INSIDE its block loop (vision_transformer.py:228-233), so with
    train-time augmentations every block rotates with its own freshly drawn tables."""
    x = patch_embed(img, sd[pre + "patch_embed.proj.weight"], sd[pre + "patch_embed.proj.bias"])
    B, hw, D = x.shape
    H, W = img.shape[-2] // 16, img.shape[-1] // 16
    if masks is not None:  # vision_transformer.py:194-196
        x = torch.where(masks.unsqueeze(-1), sd[pre + "mask_token"].to(x.dtype).unsqueeze(0), x)
        cls = sd[pre + "cls_token"]
    else:
        cls = sd[pre + "cls_token"] + 0 * sd[pre + "mask_token"]  # :198
    x = torch.cat([cls.expand(B, -1, -1), x], dim=1)  # :210-217 (no storage tokens)
    rope = rope_table(H, W, sd[pre + "rope_embed.periods"])
    for i in range(_depth(sd, pre + "blocks.")):
        if rope_aug is not None:
            rope = rope_table(H, W, sd[pre + "rope_embed.periods"], rope_aug[i])
        x = vit_block(x, sd, f"{pre}blocks.{i}.", num_heads, rope, "rmsnorm", drop=None if drop is None else drop[i], lin=lin)
    xn = rmsnorm(x, sd[pre + "norm.weight"], 1e-5)  # :246
    cls_t, patch_t = xn[:, 0], xn[:, 1:]
    if use_bottleneck and (pre + "feature_bottleneck.weight") in sd:  # bottleneck.py:66-79
        wb = sd[pre + "feature_bottleneck.weight"]
        cls_t = F.linear(cls_t, wb)
        patch_t = F.linear(patch_t.reshape(-1, D), wb).reshape(B, hw, -1)
    return {"x_norm_clstoken": cls_t, "x_norm_patchtokens": patch_t, "x_prenorm": x}


def intermediate_layers(sd, img, num_heads, n=1, reshape=False, return_class_token=False, norm=True, pre="trunk."):
    """VTPModel.get_intermediate_layers_feature (modeling_vtp.py:214-240) = DinoVisionTransformer.get_intermediate_layers
    (vision_transformer.py:266-318, no storage tokens, tied cls/patch norm)."""
    x = patch_embed(img, sd[pre + "patch_embed.proj.weight"], sd[pre + "patch_embed.proj.bias"])
    B, hw, D = x.shape
    H, W = img.shape[-2] // 16, img.shape[-1] // 16
    cls = sd[pre + "cls_token"] + 0 * sd[pre + "mask_token"]
    x = torch.cat([cls.expand(B, -1, -1), x], dim=1)
    rope = rope_table(H, W, sd[pre + "rope_embed.periods"])
    depth = _depth(sd, pre + "blocks.")
    take = range(depth - n, depth) if isinstance(n, int) else n
    outs = []
    for i in range(depth):
        x = vit_block(x, sd, f"{pre}blocks.{i}.", num_heads, rope, "rmsnorm")
        if i in take:
            outs.append(x)
    if norm:
        outs = [rmsnorm(o, sd[pre + "norm.weight"], 1e-5) for o in outs]
    cls_t = [o[:, 0] for o in outs]
    outs = [o[:, 1:] for o in outs]
    if reshape:
        outs = [o.reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous() for o in outs]
    return tuple(zip(outs, cls_t)) if return_class_token else tuple(outs)


def reconstruction_latents(sd, img, num_heads, lin=_plain_linear) -> Tensor:
    """VTPModel.get_reconstruction_latents -- modeling_vtp.py:337-360,379-395."""
    out = trunk_forward(sd, img, num_heads, use_bottleneck=True, lin=lin)
    pt = out["x_norm_patchtokens"]
    B, N, C = pt.shape
    return pt.transpose(1, 2).reshape(B, C, img.shape[-2] // 16, img.shape[-1] // 16)


# --------------------------------------------------------------------------------------------
# pixel decoder -- vtp/models/decoders/pixel_decoder.py:134-162
# --------------------------------------------------------------------------------------------
def decoder_forward(sd: Dict[str, Tensor], latents: Tensor, num_heads: int, pre: str = "pixel_decoder.", drop=None,
                    lin=_plain_linear, rope_aug=None) -> Tensor:
    """DinoV3PixelDecoder.forward -- pixel_decoder.py:134-162.  rope_aug = ONE rope_aug_draw(...): the decoder evaluates rope_embed once,
    in front of its block loop (:144), so all blocks share one augmented table."""
    B, _, H, W = latents.shape
    x = F.conv2d(latents, sd[pre + "proj_in.weight"], sd[pre + "proj_in.bias"])  # :138
    D = x.shape[1]
    x = x.flatten(2).transpose(1, 2)  # :141
    rope = rope_table(H, W, sd[pre + "rope_embed.periods"], rope_aug)  # :144
    for i in range(_depth(sd, pre + "blocks.")):
        x = vit_block(x, sd, f"{pre}blocks.{i}.", num_heads, rope, "layernorm", drop=None if drop is None else drop[i], lin=lin)
    x = layernorm(x, sd[pre + "norm.weight"], sd[pre + "norm.bias"], 1e-6)  # :151
    x = x.transpose(1, 2).reshape(B, D, H, W)  # :154
    x = F.conv2d(x, sd[pre + "proj_out.weight"], sd[pre + "proj_out.bias"])  # :157
    return F.pixel_shuffle(x, 16)  # :160


# --------------------------------------------------------------------------------------------
# CLIP text tower + heads -- vtp/models/encoders/text_transformer.py, modeling_vtp.py:244-333
# --------------------------------------------------------------------------------------------
def text_block(x: Tensor, sd, pre: str, num_heads: int, causal: bool = True, quick_gelu: bool = False) -> Tensor:
    """ResidualAttentionBlock.forward -- block.py:416-427 with nn.MultiheadAttention (packed
    in_proj, additive causal mask text_transformer.py:334-338) and exact-erf GELU MLP."""
    B, T, C = x.shape
    h = layernorm(x, sd[pre + "ln_1.weight"], sd[pre + "ln_1.bias"], 1e-5)
    qkv = F.linear(h, sd[pre + "attn.in_proj_weight"], sd[pre + "attn.in_proj_bias"])
    q, k, v = qkv.reshape(B, T, 3, num_heads, C // num_heads).unbind(2)
    q, k, v = [t.transpose(1, 2) for t in (q, k, v)]
    o = F.scaled_dot_product_attention(q, k, v, is_causal=causal)
    o = o.transpose(1, 2).reshape(B, T, C)
    a = F.linear(o, sd[pre + "attn.out_proj.weight"], sd[pre + "attn.out_proj.bias"])
    x = x + (a * sd[pre + "ls_1.gamma"] if pre + "ls_1.gamma" in sd else a)  # LayerScale when ls_init_value is set (block.py:388,425)
    h = layernorm(x, sd[pre + "ln_2.weight"], sd[pre + "ln_2.bias"], 1e-5)
    h = F.linear(h, sd[pre + "mlp.c_fc.weight"], sd[pre + "mlp.c_fc.bias"])
    h = h * torch.sigmoid(1.702 * h) if quick_gelu else F.gelu(h)  # QuickGELU (layers/activation.py:5-12) | nn.GELU
    m = F.linear(h, sd[pre + "mlp.c_proj.weight"], sd[pre + "mlp.c_proj.bias"])
    return x + (m * sd[pre + "ls_2.gamma"] if pre + "ls_2.gamma" in sd else m)


def clip_text_feature(sd, text: Tensor, num_heads: int, normalize: bool = True, pool_type: str = "argmax",
                      causal: bool = True, quick_gelu: bool = False) -> Tensor:
    """VTPModel.get_clip_text_feature -- modeling_vtp.py:278-310; text_global_pool text_transformer.py:213-228 (argmax = EOT | first
    | last | none = every token, result [B, T, D]); causal = not text_no_causal_mask (text_transformer.py:285-288).
    text_embed_cls (text_transformer.py:268-272): the class keeps TextTransformer's positional_embedding [context_length + 1, D] and
    its (context_length + 1)^2 causal mask but NOT cls_emb / build_cls_mask (modeling_vtp.py:163-170 re-hangs the parts and never
    calls TextTransformer._embeds), so `text` must simply carry context_length + 1 ids -- nothing else changes here."""
    x = F.embedding(text, sd["token_embedding.weight"]) + sd["positional_embedding"]
    for i in range(_depth(sd, "text_transformer.resblocks.")):
        x = text_block(x, sd, f"text_transformer.resblocks.{i}.", num_heads, causal=causal, quick_gelu=quick_gelu)
    x = layernorm(x, sd["ln_final.weight"], sd["ln_final.bias"], 1e-5)
    if pool_type == "first":
        x = x[:, 0]
    elif pool_type == "last":
        x = x[:, -1]
    elif pool_type == "argmax":
        x = x[torch.arange(x.shape[0], device=x.device), text.argmax(dim=-1)]
    x = x @ sd["text_projection"]
    return F.normalize(x, dim=-1) if normalize else x


def clip_image_feature(sd, img: Tensor, num_heads: int, normalize: bool = True, clip_feat: str = "cls",
                       ae_only: bool = True) -> Tensor:
    """VTPModel.get_clip_image_feature -- modeling_vtp.py:244-276 (defaults vision_bottleneck_ae_only=True => no
    bottleneck, vision_clip_feat='cls'; 'pooled' = mean of the patch tokens, :269)."""
    out = trunk_forward(sd, img, num_heads, use_bottleneck=not ae_only)
    feat = out["x_norm_clstoken"] if clip_feat == "cls" else out["x_norm_patchtokens"].mean(dim=1)
    f = F.linear(feat, sd["visual_proj.weight"])
    return F.normalize(f, dim=-1) if normalize else f


def clip_logits(sd, img, text, vis_heads, txt_heads) -> Tensor:
    """VTPModel.get_clip_logits -- modeling_vtp.py:312-333."""
    i = clip_image_feature(sd, img, vis_heads)
    t = clip_text_feature(sd, text, txt_heads)
    return sd["logit_scale"].exp() * i @ t.T


# --------------------------------------------------------------------------------------------
# loss heads -- NOT in the reference (parity unpinned)
# --------------------------------------------------------------------------------------------
def l1_loss(rec: Tensor, target: Tensor) -> Tensor:
    return (rec.float() - target.float()).abs().mean()


def clip_loss(img_f: Tensor, txt_f: Tensor, logit_scale_exp: Tensor) -> Tensor:
    """OpenCLIP ClipLoss (single process): 0.5*(CE(s*I*T^T) + CE(s*T*I^T)), labels=arange."""
    logits = logit_scale_exp * img_f @ txt_f.T
    labels = torch.arange(logits.shape[0], device=logits.device)
    return 0.5 * (F.cross_entropy(logits, labels) + F.cross_entropy(logits.T, labels))


def rec_train_loss(sd, img, vis_heads, dec_heads) -> Tensor:
    """forward_type='rec' (vtp.py:487-512 / modeling_vtp.py:440-455) + L1."""
    lat = reconstruction_latents(sd, img, vis_heads)
    rec = decoder_forward(sd, lat, dec_heads)
    return l1_loss(rec, img)


def rec_clip_train_loss(sd, img, text, vis_heads, dec_heads, txt_heads) -> Tuple[Tensor, Tensor]:
    """rec (L1) + clip (InfoNCE) on the same images: returns (l1, clip).  The trunk is evaluated once per objective
    exactly like two VTP.forward calls (forward_type='rec' and 'clip', vtp.py:323-338); with drop rates 0 both see the
    same trunk activations."""
    l1 = rec_train_loss(sd, img, vis_heads, dec_heads)
    i = clip_image_feature(sd, img, vis_heads)
    t = clip_text_feature(sd, text, txt_heads)
    return l1, clip_loss(i, t, sd["logit_scale"].exp())


# --------------------------------------------------------------------------------------------
# SSL branch -- DINO head (dino_head.py), teacher/student token buffers (vtp.py:410-484); losses are OUR spec
# --------------------------------------------------------------------------------------------
def dino_head_forward(sd, pre: str, x: Tensor) -> Tensor:
    """DINOHead.forward (dino_head.py:65-89): MLP -> F.normalize(eps=1e-12) -> weight-normed linear
    (W = g * v / ||v||_row, torch.nn.utils.weight_norm dim=0; dino_head.py:47-49)."""
    h = F.gelu(F.linear(x, sd[pre + "mlp.0.weight"], sd[pre + "mlp.0.bias"]))
    h = F.gelu(F.linear(h, sd[pre + "mlp.2.weight"], sd[pre + "mlp.2.bias"]))
    z = F.linear(h, sd[pre + "mlp.4.weight"], sd[pre + "mlp.4.bias"])
    z = F.normalize(z, dim=-1, p=2, eps=1e-12)
    v, g = sd[pre + "last_layer.weight_v"], sd[pre + "last_layer.weight_g"]
    return F.linear(z, g * v / v.norm(dim=1, keepdim=True))


def ssl_outputs(sd, global_crops, local_crops, masks, vis_heads: int):
    """VTP.get_teacher_forward_outputs + get_student_ssl_outputs (vtp.py:410-484) for bottleneck_ae_only=True and
    drop rates 0.  The student's two resolutions go through the same trunk weights; batching them in one list call
    (block.py:235-298) is an efficiency device only, so they are evaluated as two passes here."""
    idx = masks.flatten().nonzero().flatten()
    with torch.no_grad():
        t = trunk_forward(sd, global_crops, vis_heads, use_bottleneck=False, pre="teacher_trunk.")
        cls = t["x_norm_clstoken"].chunk(2)
        cls = torch.cat((cls[1], cls[0]))                                   # vtp.py:425-426
        patches = t["x_norm_patchtokens"].flatten(0, 1)[idx]
        th = dino_head_forward(sd, "teacher_dino_head.", torch.cat([cls, patches]))
    n_cls = cls.shape[0]
    teacher = {"teacher_cls_tokens_after_head": th[:n_cls], "masked_teacher_patch_tokens_after_head": th[n_cls:]}
    sg = trunk_forward(sd, global_crops, vis_heads, use_bottleneck=False, masks=masks)
    sl = trunk_forward(sd, local_crops, vis_heads, use_bottleneck=False)
    student = {"student_local_cls_tokens_after_head": dino_head_forward(sd, "dino_head.", sl["x_norm_clstoken"]),
               "student_global_cls_tokens_after_head": dino_head_forward(sd, "dino_head.", sg["x_norm_clstoken"]),
               "student_global_cls_tokens": sg["x_norm_clstoken"],
               "student_global_masked_patch_tokens_after_head":
                   dino_head_forward(sd, "dino_head.", sg["x_norm_patchtokens"].flatten(0, 1)[idx])}
    return teacher, student


def ssl_loss(t_out, s_out, masks, center_dino, center_ibot, n_local: int, student_temp: float = 0.1,
             teacher_temp: float = 0.07, dino_weight: float = 1.0, ibot_weight: float = 1.0,
             centering: str = "softmax", koleo_weight: float = 0.0, sk_iterations: int = 3) -> Tensor:
    """OUR SSL loss spec (DINOv2 conventions; the reference ships none -> parity unpinned):
      teacher targets  p = softmax((z_t - center) / teacher_temp)            (separate centres for cls / patch tokens)
      DINO  = [ sum_{local crop j, view v} mean_b CE(s_loc[j,b], p[v,b]) + sum_v mean_b CE(s_glob[v,b], p[other(v),b]) ]
              / (n_g (n_g - 1) + n_local n_g)            with the teacher cls rows already view-swapped (vtp.py:425-426)
      iBOT  = (1 / B) sum_masked tokens CE(s_patch, p_patch) / n_masked_in_image
    Returns dino_weight * DINO + ibot_weight * iBOT.
    Variants (DINOv2 ssl_meta_arch conventions): centering="sinkhorn_knopp" replaces the centred softmax by
    loss_oracle.sinkhorn_knopp over the cls rows / the masked patch rows; koleo_weight adds
    koleo_weight * sum_v KoLeo(student global cls tokens of view v)."""
    from .loss_oracle import koleo_loss, sinkhorn_knopp
    tc = t_out["teacher_cls_tokens_after_head"].detach()
    tp = t_out["masked_teacher_patch_tokens_after_head"].detach()
    B2 = tc.shape[0]
    B = B2 // 2
    if centering == "sinkhorn_knopp":
        p_cls = sinkhorn_knopp(tc.float(), teacher_temp, sk_iterations).float()
    else:
        p_cls = F.softmax((tc - center_dino.to(tc.device)) / teacher_temp, dim=-1)
    lsm_g = F.log_softmax(s_out["student_global_cls_tokens_after_head"] / student_temp, dim=-1)
    lsm_l = F.log_softmax(s_out["student_local_cls_tokens_after_head"] / student_temp, dim=-1)
    terms = 2 * 1 + n_local * 2
    dino = -(p_cls * lsm_g).sum(-1).sum() / B               # sum over the two views of mean_b
    lsm_l = lsm_l.view(n_local, B, -1)
    for v in range(2):
        dino = dino - (p_cls[v * B:(v + 1) * B][None] * lsm_l).sum(-1).sum() / B
    dino = dino / terms
    ibot = tc.new_zeros(())  # (device-agnostic: the benchmarked-geometry parity test evaluates this oracle on the GPU in fp32)
    if tp.shape[0] > 0:
        if centering == "sinkhorn_knopp":
            p_pat = sinkhorn_knopp(tp.float(), teacher_temp, sk_iterations).float()
        else:
            p_pat = F.softmax((tp - center_ibot.to(tp.device)) / teacher_temp, dim=-1)
        lsm_p = F.log_softmax(s_out["student_global_masked_patch_tokens_after_head"] / student_temp, dim=-1)
        masks = masks.to(tp.device)
        per_img = masks.sum(1).clamp(min=1)
        img_of = masks.nonzero()[:, 0]
        w = 1.0 / per_img[img_of].float()
        ibot = -((p_pat * lsm_p).sum(-1) * w).sum() / B
    total = dino_weight * dino + ibot_weight * ibot
    if koleo_weight:
        for x in s_out["student_global_cls_tokens"].float().chunk(2):
            total = total + koleo_weight * koleo_loss(x)[0]
    return total


def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, b1: float, b2: float,
               eps: float, wd: float) -> None:
    """torch.optim.AdamW semantics (decoupled weight decay), in place, fp32."""
    p.mul_(1 - lr * wd)
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
