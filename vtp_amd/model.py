"""VTPModel -- MI355X-native drop-in for the reference ``VTPModel``
(vtp/models/vtp_hf/modeling_vtp.py:51-472): same constructor (a VTPConfig), same public methods, same
``state_dict`` keys / shapes (so reference checkpoints load unchanged), but every tower runs on the
hand-written gfx950 kernels of libvtp_hip.so through vtp_amd.engine (there is no PyTorch-eager fallback).
"""
from __future__ import annotations

import json
import math
import os
from typing import Dict, Optional, Tuple

import numpy as np
import torch
from torch import nn

from . import autograd as ag
from . import ops
from .config import VTPConfig, ffn_hidden, swiglu_hidden
from .engine import BF, F32, DecoderEngine, ParamStore, TrunkEngine


def _holder() -> nn.Module:
    return nn.Module()


def _param(*shape) -> nn.Parameter:
    return nn.Parameter(torch.empty(*shape))


def _linear(out_f: int, in_f: int, bias: bool = True) -> nn.Module:
    m = _holder()
    m.weight = _param(out_f, in_f)
    if bias:
        m.bias = _param(out_f)
    return m


def _norm(dim: int, bias: bool) -> nn.Module:
    m = _holder()
    m.weight = _param(dim)
    if bias:
        m.bias = _param(dim)
    return m


def _vit_block(D: int, H: int, norm: str, init_values=None, qk_norm: bool = False, ffn: str = "swiglu") -> nn.Module:
    """Parameter tree of SelfAttentionBlock (block.py:159-187); ls1 / ls2 = LayerScale gammas (misc.py:7-26) when `init_values`;
    attn.q_norm / attn.k_norm = RMSNorm(head_dim = 64) weights when `qk_norm` (attention.py:67-68); mlp = SwiGLUFFN (w1, w2, w3) or
    the GELU Mlp (fc1, fc2; ffn.py:21-48) when ffn == "mlp"."""
    b = _holder()
    b.norm1 = _norm(D, norm != "rmsnorm")
    b.attn = _holder()
    b.attn.qkv = _linear(3 * D, D)
    b.attn.proj = _linear(D, D)
    if qk_norm:
        b.attn.q_norm = _norm(64, False)
        b.attn.k_norm = _norm(64, False)
    if init_values:
        b.ls1 = _holder()
        b.ls1.gamma = _param(D)
        b.ls1.init_values = float(init_values)
    b.norm2 = _norm(D, norm != "rmsnorm")
    b.mlp = _holder()
    if ffn == "mlp":
        b.mlp.fc1 = _linear(H, D)
        b.mlp.fc2 = _linear(D, H)
    else:
        b.mlp.w1 = _linear(H, D)
        b.mlp.w2 = _linear(H, D)
        b.mlp.w3 = _linear(D, H)
    if init_values:
        b.ls2 = _holder()
        b.ls2.gamma = _param(D)
        b.ls2.init_values = float(init_values)
    return b


def _rope_periods(head_dim: int = 64, base: float = 100.0) -> torch.Tensor:
    """RopePositionEmbedding._init_weights (embeddings.py:182-195), bf16 like the reference default."""
    dt = torch.bfloat16
    return base ** (2 * torch.arange(head_dim // 4, dtype=dt) / (head_dim // 2))


class VTPModel(nn.Module):
    config_class = VTPConfig

    def __init__(self, config: VTPConfig):
        super().__init__()
        self.config = config
        c = config
        D, Dd, Dt = c.vision_embed_dim, c.decoder_embed_dim, c.text_embed_dim
        # ---- trunk (DinoVisionTransformerWithBottleneck)
        t = _holder()
        t.cls_token = _param(1, 1, D)
        t.mask_token = _param(1, D)
        t.patch_embed = _holder()
        t.patch_embed.proj = _holder()
        t.patch_embed.proj.weight = _param(D, 3, 16, 16)
        t.patch_embed.proj.bias = _param(D)
        t.rope_embed = _holder()
        t.rope_embed.register_buffer("periods", _rope_periods(), persistent=True)
        Hv = ffn_hidden(D, c.vision_mlp_ratio, c.vision_ffn_layer)
        t.blocks = nn.ModuleList([_vit_block(D, Hv, c.vision_norm_layer, c.vision_init_values, c.vision_use_qk_norm, c.vision_ffn_layer)
                                  for _ in range(c.vision_depth)])
        t.norm = _norm(D, c.vision_norm_layer != "rmsnorm")
        if c.vision_feature_bottleneck is not None and c.vision_feature_bottleneck != D:
            t.feature_bottleneck = _holder()
            t.feature_bottleneck.weight = _param(c.vision_feature_bottleneck, D)
        self.trunk = t
        eff = c.vision_feature_bottleneck if hasattr(t, "feature_bottleneck") else D
        if c.train_clip:
            self.visual_proj = _holder()
            self.visual_proj.weight = _param(Dt, D if c.vision_bottleneck_ae_only else eff)
        else:
            self.visual_proj = None
        # ---- pixel decoder (DinoV3PixelDecoder)
        if c.train_reconstruction:
            d = _holder()
            d.proj_in = _holder()
            d.proj_in.weight = _param(Dd, eff, 1, 1)
            d.proj_in.bias = _param(Dd)
            d.rope_embed = _holder()
            d.rope_embed.register_buffer("periods", _rope_periods(), persistent=True)
            Hd = ffn_hidden(Dd, 4.0, c.decoder_ffn_layer)
            d.blocks = nn.ModuleList([_vit_block(Dd, Hd, c.decoder_norm_layer, c.decoder_init_values, c.decoder_use_qk_norm, c.decoder_ffn_layer)
                                      for _ in range(c.decoder_depth)])
            d.norm = _norm(Dd, c.decoder_norm_layer != "rmsnorm")
            d.proj_out = _holder()
            d.proj_out.weight = _param(768, Dd, 1, 1)
            d.proj_out.bias = _param(768)
            self.pixel_decoder = d
        else:
            self.pixel_decoder = None
        # ---- CLIP text tower (TextTransformer pieces as re-hung by VTPModel._init_text_components)
        if c.train_clip:
            tt = _holder()
            blocks = []
            for _ in range(c.text_depth):
                r = _holder()
                r.ln_1 = _norm(Dt, True)
                r.attn = _holder()
                r.attn.in_proj_weight = _param(3 * Dt, Dt)
                r.attn.in_proj_bias = _param(3 * Dt)
                r.attn.out_proj = _linear(Dt, Dt)
                r.ln_2 = _norm(Dt, True)
                r.mlp = _holder()
                r.mlp.c_fc = _linear(int(Dt * c.text_mlp_ratio), Dt)
                r.mlp.c_proj = _linear(Dt, int(Dt * c.text_mlp_ratio))
                if c.text_ls_init_value is not None:  # LayerScale on both residual branches (block.py:388,399)
                    for nm_ in ("ls_1", "ls_2"):
                        ls = _holder()
                        ls.gamma = _param(Dt)
                        ls.init_values = float(c.text_ls_init_value)
                        setattr(r, nm_, ls)
                blocks.append(r)
            tt.resblocks = nn.ModuleList(blocks)
            self.text_transformer = tt
            self.token_embedding = _holder()
            self.token_embedding.weight = _param(c.text_vocab_size, Dt)
            self.positional_embedding = _param(c.text_num_pos, Dt)
            self.ln_final = _norm(Dt, True)
            self.text_projection = _param(Dt, Dt)
            self.context_length, self.vocab_size = c.text_context_length, c.text_vocab_size
            lshape = [1] if c.nonscalar_logit_scale else []
            self.logit_scale = nn.Parameter(torch.ones(lshape) * (c.init_logit_scale or np.log(1 / 0.07)))
        # SigLIP (modeling_vtp.py:177-180): a learnable logit bias next to the logit scale
        if c.train_clip and c.init_logit_bias is not None:
            self.logit_bias = nn.Parameter(torch.ones([1] if c.nonscalar_logit_scale else []) * c.init_logit_bias)
        else:
            self.logit_bias = None
        self.reset_parameters()
        self._store: Optional[ParamStore] = None

    # ------------------------------------------------------------------------------------------------ init
    @torch.no_grad()
    def reset_parameters(self):
        """Same distributions as the reference init (init_weights_vit vision_transformer.py:43-55, PatchEmbed
        embeddings.py:79-83, decoder pixel_decoder.py:122-131, text text_transformer.py:304-327)."""
        c = self.config

        def vit(prefix_mod):
            for name, p in prefix_mod.named_parameters():
                if name.endswith(("norm1.weight", "norm2.weight", "q_norm.weight", "k_norm.weight")) or name == "norm.weight":
                    p.fill_(1.0)
                elif name.endswith(".gamma"):  # LayerScale.reset_parameters (misc.py:21-22)
                    p.fill_(c.vision_init_values if prefix_mod is self.trunk else c.decoder_init_values)
                elif name.endswith(".bias"):
                    p.zero_()
                elif name.endswith(".weight") and p.ndim == 2:
                    nn.init.trunc_normal_(p, std=0.02)

        vit(self.trunk)
        nn.init.normal_(self.trunk.cls_token, std=0.02)
        self.trunk.mask_token.zero_()
        k = 1.0 / (3 * 16 * 16)
        self.trunk.patch_embed.proj.weight.uniform_(-math.sqrt(k), math.sqrt(k))
        self.trunk.patch_embed.proj.bias.uniform_(-math.sqrt(k), math.sqrt(k))
        if self.visual_proj is not None:
            nn.init.trunc_normal_(self.visual_proj.weight, std=0.02)
        if self.pixel_decoder is not None:
            vit(self.pixel_decoder)
            for m in (self.pixel_decoder.proj_in, self.pixel_decoder.proj_out):
                nn.init.trunc_normal_(m.weight, std=0.02)
                m.bias.zero_()
        if c.train_clip:
            W, L = c.text_embed_dim, c.text_depth
            nn.init.normal_(self.token_embedding.weight, std=0.02)
            nn.init.normal_(self.positional_embedding, std=0.01)
            proj_std, attn_std, fc_std = (W ** -0.5) * ((2 * L) ** -0.5), W ** -0.5, (2 * W) ** -0.5
            for r in self.text_transformer.resblocks:
                nn.init.normal_(r.attn.in_proj_weight, std=attn_std)
                r.attn.in_proj_bias.zero_()
                nn.init.normal_(r.attn.out_proj.weight, std=proj_std)
                r.attn.out_proj.bias.zero_()
                nn.init.normal_(r.mlp.c_fc.weight, std=fc_std)
                r.mlp.c_fc.bias.zero_()
                nn.init.normal_(r.mlp.c_proj.weight, std=proj_std)
                r.mlp.c_proj.bias.zero_()
                for ln in (r.ln_1, r.ln_2):
                    ln.weight.fill_(1.0)
                    ln.bias.zero_()
                if c.text_ls_init_value is not None:
                    r.ls_1.gamma.fill_(c.text_ls_init_value)
                    r.ls_2.gamma.fill_(c.text_ls_init_value)
            self.ln_final.weight.fill_(1.0)
            self.ln_final.bias.zero_()
            nn.init.normal_(self.text_projection, std=W ** -0.5)

    # ------------------------------------------------------------------------------------------------ engine plumbing
    def _engine(self) -> ParamStore:
        if self._store is None:
            dev = self.trunk.cls_token.device
            if dev.type != "cuda":
                raise RuntimeError("vtp_amd.VTPModel runs only on an MI355X (HIP) device: call model.cuda() first. "
                                   "There is no CPU / eager fallback.")
            st = ParamStore(self, dev)
            self._trunk = TrunkEngine(st, self.config, self.trunk.rope_embed.periods)
            self._decoder = DecoderEngine(st, self.config, self.pixel_decoder.rope_embed.periods) \
                if self.pixel_decoder is not None else None
            self._text = self._clip = None
            if self.visual_proj is not None:
                from .clip_engine import ClipHead, TextEngine
                self._vproj = st.lin("visual_proj.weight", None, self.visual_proj.weight.shape[0],
                                     self.visual_proj.weight.shape[1])
                self._text = TextEngine(st, self.config)
                self._clip = ClipHead(st, self._vproj, self.config.vision_embed_dim, self.config.text_embed_dim)
            self._build_extra_engines(st)
            st.finalize()
            self._store = st
            self._pver = self._param_version()
        return self._store

    def _build_extra_engines(self, st):
        """hook for subclasses (vtp_amd.VTP adds the SSL engines) -- runs before the compute copies are allocated"""

    def _param_version(self) -> int:
        return sum(p._version for p in self.parameters())

    def refresh_weights(self):
        """Re-derive the bf16 compute copies after the fp32 parameters were modified outside the fused optimizer."""
        self._engine().prep()
        self._pver = self._param_version()

    def _fresh(self) -> ParamStore:
        st = self._engine()
        if self._param_version() != self._pver:
            self.refresh_weights()
            if getattr(self, "_fp8_on", False):  # the e4m3 weight copies are derived data too
                for stack in self._fp8_stacks():
                    stack.fp8_finalize()
        return st

    # ---- fp8 (e4m3) inference forward of the trunk and the pixel decoder (BASELINE config 5)
    def _fp8_stacks(self):
        self._engine()
        return [self._trunk.stack] + ([self._decoder.stack] if self._decoder is not None else [])

    @torch.no_grad()
    def enable_fp8_forward(self, calibration_images: torch.Tensor):
        """Switch the inference path (eval / no_grad: get_reconstruction_latents, get_latents_decoded_images, features) to fp8
        MFMA GEMMs: one bf16 encode -> decode pass over `calibration_images` records the per-tensor activation maxima, the
        weights are quantised from their own maxima.  Training and the autograd path are untouched (bf16)."""
        self._fp8_on = False
        for stack in self._fp8_stacks():
            stack.fp8_begin_calibration()
        lat = self._latents_nograd(calibration_images)
        if self.pixel_decoder is not None:
            self._decode_nograd(lat)
        for stack in self._fp8_stacks():
            stack.fp8_finalize()
        self._fp8_on = True
        return self

    def disable_fp8_forward(self):
        self._fp8_on = False
        for stack in self._fp8_stacks():
            stack.fp8 = None
        return self

    def zero_grad(self, set_to_none: bool = False):
        """Gradients live in ONE flat buffer that every `p.grad` views (the fused optimizer / RCCL buckets use it directly):
        zeroed in place, the views stay attached whatever `set_to_none` says."""
        if self._store is not None:
            self._store.zero_grad()
            self._store.sync_grad_views()
        else:
            super().zero_grad(set_to_none=set_to_none)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        if self._store is not None:
            self.refresh_weights()
        return out

    # ------------------------------------------------------------------------------------------------ checkpoints
    def save_pretrained(self, path: str):
        from safetensors.torch import save_file
        self.config.save_pretrained(path)
        save_file({k: v.detach().contiguous().cpu() for k, v in self.state_dict().items()},
                  os.path.join(path, "model.safetensors"))

    @classmethod
    def from_pretrained(cls, path: str, device: Optional[str] = None) -> "VTPModel":
        from safetensors.torch import load_file
        model = cls(VTPConfig.from_pretrained(path))
        model.load_state_dict(load_file(os.path.join(path, "model.safetensors")), strict=True)
        return model.to(device) if device else model

    # ------------------------------------------------------------------------------------------------ API (inference)
    @staticmethod
    def _img(image: torch.Tensor) -> torch.Tensor:
        """API-boundary check: the kernels take raw device pointers, so shape, device, dtype and layout are settled here (any float
        dtype / memory format is converted; a host tensor is an error, never a silent copy or a fault)"""
        if not isinstance(image, torch.Tensor) or image.ndim != 4 or image.shape[1] != 3 or image.shape[2] % 16 or image.shape[3] % 16:
            raise ValueError(f"image must be a [B,3,H,W] tensor with H,W multiples of 16, got {tuple(getattr(image, 'shape', ()))}")
        if not image.is_cuda:
            raise ValueError("image must live on the MI355X (got a CPU tensor): move it with .cuda() -- there is no CPU path")
        if not image.is_floating_point():
            raise ValueError(f"image must be a floating-point tensor, got {image.dtype}")
        return image.detach().to(dtype=torch.float32).contiguous()

    def _ids(self, text: torch.Tensor, check_range: bool = True) -> torch.Tensor:
        """token ids -> int64 [B, context_length], contiguous, on the device; out-of-range ids raise (the embedding kernel indexes
        the table with them).  check_range costs one host sync: the inference API pays it, the trainer's hot path does not."""
        T = self.config.text_num_pos  # (context_length + 1 with text_embed_cls: the reference's positional table has that many rows)
        if not isinstance(text, torch.Tensor) or text.ndim != 2 or text.shape[1] != T:
            raise ValueError(f"text must be [B, {T}] token ids, got {tuple(getattr(text, 'shape', ()))}")
        if not text.is_cuda:
            raise ValueError("text must live on the MI355X (got a CPU tensor)")
        if text.is_floating_point() or text.dtype == torch.bool:
            raise ValueError(f"text must hold integer token ids, got {text.dtype}")
        ids = text.detach().to(dtype=torch.int64).contiguous()
        if check_range and ids.numel():
            lo, hi = int(ids.min()), int(ids.max())
            if lo < 0 or hi >= self.config.text_vocab_size:
                raise ValueError(f"token ids must be in [0, {self.config.text_vocab_size}), got [{lo}, {hi}]")
        return ids

    def get_reconstruction_latents(self, image: torch.Tensor) -> torch.Tensor:
        """modeling_vtp.py:337-360 -> [B, 64, H/16, W/16] (f32).  Differentiable (autograd.EncodeLatents) when autograd is
        recording and the module is in training mode; the inference path otherwise."""
        if ag.grad_mode(self):
            self._img(image)
            return ag.EncodeLatents.apply(image, ag.anchor(self), self, "ag.rec")
        return self._latents_nograd(image)

    @torch.no_grad()
    def _latents_nograd(self, image: torch.Tensor) -> torch.Tensor:
        self._fresh()
        img = self._img(image)
        B, _, H, W = img.shape
        self._trunk.forward(img, train=False)
        lat = self._trunk.latents(out_f32=True)  # [B*hw, 64]
        return lat.view(B, (H // 16) * (W // 16), -1).transpose(1, 2).reshape(B, -1, H // 16, W // 16).clone()

    def get_latents_decoded_images(self, latents: torch.Tensor) -> torch.Tensor:
        """modeling_vtp.py:362-377 -> [B, 3, H, W] (f32).  Differentiable in training mode (autograd.DecodeLatents)."""
        if self.pixel_decoder is None:
            raise RuntimeError("Reconstruction not enabled. Set train_reconstruction=True in config.")
        if ag.grad_mode(self):
            return ag.DecodeLatents.apply(latents, ag.anchor(self), self)
        return self._decode_nograd(latents)

    @torch.no_grad()
    def _decode_nograd(self, latents: torch.Tensor) -> torch.Tensor:
        self._fresh()
        B, C, h, w = latents.shape
        lat = latents.detach().reshape(B, C, h * w).transpose(1, 2).to(torch.bfloat16).contiguous().view(B * h * w, C)
        t = self._decoder.forward(lat, B, h, w, train=False)
        img = torch.empty(B, 3, h * 16, w * 16, dtype=torch.float32, device=lat.device)
        ops.pixel_shuffle16(t, img, B, h, w)
        return img

    @torch.no_grad()
    def get_last_layer_feature(self, image: torch.Tensor, use_bottleneck: bool = False) -> Dict[str, torch.Tensor]:
        """modeling_vtp.py:184-215."""
        self._fresh()
        img = self._img(image)
        B, _, H, W = img.shape
        hw = (H // 16) * (W // 16)
        xnf = self._trunk.forward(img, train=False).view(B, hw + 1, -1).float()
        cls_t, patch_t = xnf[:, 0], xnf[:, 1:]
        if use_bottleneck and self._trunk.bott is not None:
            wb = self.trunk.feature_bottleneck.weight
            patch_t = self._trunk.latents(out_f32=True).view(B, hw, -1).clone()
            cls_b = torch.empty(B, wb.shape[0], dtype=torch.float32, device=img.device)
            ops.gemm_nt(self._trunk.ctx().xnf, self._trunk.bott.w, cls_b, M=B, N=wb.shape[0], K=wb.shape[1],
                        lda=(hw + 1) * wb.shape[1], epi=ops.EPI_F32)
            cls_t = cls_b
        return {"cls_token": cls_t.contiguous(), "patch_tokens": patch_t.contiguous()}

    @torch.no_grad()
    def get_intermediate_layers_feature(self, image: torch.Tensor, n=1, reshape: bool = False, return_class_token: bool = False,
                                        norm: bool = True):
        """modeling_vtp.py:214-240 -> DinoVisionTransformer.get_intermediate_layers (vision_transformer.py:266-318): outputs
        of the last `n` blocks (or of the listed block indices), optionally through the trunk's final norm; patch tokens
        [B, hw, D] (or [B, D, h, w] with reshape), each paired with its class token [B, D] when return_class_token."""
        self._fresh()
        img = self._img(image)
        B, _, H, W = img.shape
        h, w = H // 16, W // 16
        tr = self._trunk
        depth = tr.depth
        # the reference walks the blocks in order and collects those listed (vision_transformer.py:266-281): ascending block order
        # whatever the order of `n`, and its length assert fails on duplicates or indices outside the trunk
        want = list(range(depth - n, depth)) if isinstance(n, int) else [int(i) for i in n]
        take = sorted(set(i for i in want if 0 <= i < depth))
        if not take or len(take) != len(want):
            raise AssertionError(f"only {len(take)} / {len(want)} blocks found")
        tr.forward(img, train=True, tag="intermediate")  # train=True keeps every block's output buffer
        c = tr.ctx()
        M, D = c.M, tr.D
        st = self._store
        outs = []
        for i in take:
            x = c.ws.get(f"{i}.xout", (M, D), torch.float32)
            if norm:
                y = c.ws.get("inter.y", (M, D), torch.bfloat16)
                stats = c.ws.get("inter.st", (M, 2), torch.float32)
                ops.norm_fwd(x, st.p(tr.prefix + "norm.weight"), st.p(tr.prefix + "norm.bias") if tr.kind == ops.NORM_LN else None,
                             y, stats, M, D, tr.eps, tr.kind)
                x = y
            x = x.float().view(B, h * w + 1, D)
            cls_t, patch = x[:, 0].contiguous(), x[:, 1:]
            patch = patch.reshape(B, h, w, D).permute(0, 3, 1, 2).contiguous() if reshape else patch.contiguous()
            outs.append((patch, cls_t) if return_class_token else patch)
        return tuple(outs)

    def get_clip_image_feature(self, image: torch.Tensor, normalize: bool = True) -> torch.Tensor:
        """modeling_vtp.py:244-276: cls (or mean-pooled patch) token of the final-norm trunk output -- bottlenecked first unless
        vision_bottleneck_ae_only -- through visual_proj, optionally L2-normalised.  Differentiable in training mode."""
        if self.visual_proj is None:
            raise RuntimeError("CLIP not enabled. Set train_clip=True in config.")
        c = self.config
        general = c.vision_clip_feat != "cls" or not c.vision_bottleneck_ae_only
        if ag.grad_mode(self) or general:
            # tokens from the trunk kernels; the [B, D] pooling / projection / normalisation heads on kernels too (autograd.HeadLinear /
            # SumTokens / L2Normalize: the arithmetic of modeling_vtp.py:262-276, bf16 MFMA with fp32 accumulation like the
            # reference's Linear layers under bf16 autocast)
            self._img(image)
            if ag.grad_mode(self):
                tokens = ag.TrunkTokens.apply(image, ag.anchor(self), self, "ag.clip")
            else:
                with torch.no_grad():
                    self._fresh()
                    img = self._img(image)
                    tokens = self._trunk.forward(img, train=False).float().view(img.shape[0], -1, c.vision_embed_dim)
            with torch.set_grad_enabled(ag.grad_mode(self)):
                a = ag.anchor(self)
                if c.vision_clip_feat == "cls":
                    feat, scale = tokens[:, 0], 1.0
                else:  # mean over the patch tokens: the sum here, 1 / hw in the next projection (linear maps commute with the mean)
                    feat, scale = ag.SumTokens.apply(tokens[:, 1:]), 1.0 / (tokens.shape[1] - 1)
                if not c.vision_bottleneck_ae_only and self._trunk.bott is not None:
                    feat, scale = ag.HeadLinear.apply(feat, a, self, self._trunk.bott, scale, self.trunk.feature_bottleneck.weight.requires_grad), 1.0
                feat = ag.HeadLinear.apply(feat, a, self, self._vproj, scale, self.visual_proj.weight.requires_grad)
                return ag.L2Normalize.apply(feat) if normalize else feat
        return self._clip_image_nograd(image, normalize)

    @torch.no_grad()
    def _clip_image_nograd(self, image: torch.Tensor, normalize: bool = True) -> torch.Tensor:
        self._fresh()
        c = self.config
        img = self._img(image)
        B, _, H, W = img.shape
        N = (H // 16) * (W // 16) + 1
        xnf = self._trunk.forward(img, train=False)  # [B*N, D] bf16; cls rows are b*N
        D = c.vision_embed_dim
        f = self._clip.image_features(xnf, B, N)
        if normalize:
            f, _ = self._clip.normalize(f, "img")
        return f.clone()

    def get_clip_text_feature(self, text: torch.Tensor, normalize: bool = True) -> torch.Tensor:
        """modeling_vtp.py:278-310.  Differentiable in training mode (autograd.TextFeature)."""
        if not self.config.train_clip:
            raise RuntimeError("CLIP not enabled. Set train_clip=True in config.")
        if ag.grad_mode(self):
            ids = self._ids(text)
            f = ag.TextFeature.apply(ids, ag.anchor(self), self)  # [B, D_t]; text_pool_type = "none": [B * T, D_t] (every token)
            f = ag.L2Normalize.apply(f) if normalize else f
            return f.view(ids.shape[0], ids.shape[1], -1) if self.config.text_pool_type == "none" else f
        return self._clip_text_nograd(text, normalize)

    @torch.no_grad()
    def _clip_text_nograd(self, text: torch.Tensor, normalize: bool = True) -> torch.Tensor:
        self._fresh()
        ids = self._ids(text)
        f = self._text.forward(ids, train=False)
        if normalize:
            f, _ = self._clip.normalize(f, "txt")
        f = f.clone()
        return f.view(ids.shape[0], ids.shape[1], -1) if self.config.text_pool_type == "none" else f  # per-token features [B, T, D_t]

    def get_clip_logits(self, image: torch.Tensor, text: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """modeling_vtp.py:312-333.  Inference: the [B_img, B_txt] logits come from the clip_logits kernel (exp(logit_scale) * I T^T,
        fp32); in training mode the same kernel inside autograd.ClipLogits (backward on the clip_grad kernels)."""
        if self.config.text_pool_type == "none":
            # the reference multiplies [B, D] image features with `text_features.T` of a [B, T, D] tensor and fails in the matmul
            # (modeling_vtp.py:329): same error class, stated
            raise RuntimeError("get_clip_logits needs pooled text features: text_pool_type = 'none' returns per-token features [B, T, D]")
        i = self.get_clip_image_feature(image, normalize=True)
        t = self.get_clip_text_feature(text, normalize=True)
        if ag.grad_mode(self):
            logits = ag.ClipLogits.apply(i, t, self.logit_scale)
        else:
            logits = torch.empty(i.shape[0], t.shape[0], dtype=torch.float32, device=i.device)
            ops.clip_logits(i.contiguous(), t.contiguous(), self._store.p("logit_scale"), logits, i.shape[0], t.shape[0], i.shape[1])
        if self.logit_bias is not None:
            logits = logits + self.logit_bias
        return logits, logits.T

    def forward(self, image=None, text=None, forward_type: str = "clip"):
        """modeling_vtp.py:399-472 (inference semantics)."""
        if forward_type == "rec":
            if image is None:
                raise ValueError("image is required for reconstruction")
            lat = self.get_reconstruction_latents(image)
            return {"latents": lat, "reconstructed_image": self.get_latents_decoded_images(lat), "target_image": image}
        if forward_type == "feature":
            if image is None:
                raise ValueError("image is required for feature extraction")
            f = self.get_last_layer_feature(image, use_bottleneck=True)
            return f
        if forward_type == "clip":
            out = {}
            if image is not None:
                out["image_features"] = self.get_clip_image_feature(image, normalize=True)
            if text is not None:
                out["text_features"] = self.get_clip_text_feature(text, normalize=True)
            out["logit_scale"] = self.logit_scale.exp()
            return out
        raise ValueError(f"Invalid forward_type: {forward_type}")
