"""VTPConfig -- field-for-field mirror of the reference configuration
(vtp/models/vtp_hf/configuration_vtp.py:67-166; defaults = VTP-Base f16d64) so a reference ``config.json`` loads
unchanged.  Plain Python (no transformers dependency on the hot path)."""
from __future__ import annotations

import json
import os
from typing import Optional


ROPE_AUG_KEYS = ("vision_rope_shift_coords", "vision_rope_jitter_coords", "vision_rope_rescale_coords",
                 "decoder_rope_shift_coords", "decoder_rope_jitter_coords", "decoder_rope_rescale_coords")


class VTPConfig:
    model_type = "vtp"

    def __init__(
        self,
        image_size: int = 256,
        train_clip: bool = True,
        train_reconstruction: bool = True,
        vision_patch_size: int = 16,
        vision_embed_dim: int = 768,
        vision_depth: int = 12,
        vision_num_heads: int = 12,
        vision_mlp_ratio: float = 4.0,
        vision_ffn_layer: str = "swiglu",
        vision_norm_layer: str = "rmsnorm",
        vision_init_values: Optional[float] = None,
        vision_use_qk_norm: bool = False,
        vision_feature_bottleneck: int = 64,
        vision_bottleneck_ae_only: bool = True,
        vision_clip_feat: str = "cls",
        text_context_length: int = 77,
        text_vocab_size: int = 49408,
        text_embed_dim: int = 768,
        text_num_heads: int = 12,
        text_depth: int = 12,
        text_mlp_ratio: float = 4.0,
        text_ls_init_value: Optional[float] = None,
        text_embed_cls: bool = False,
        text_pad_id: int = 0,
        text_no_causal_mask: bool = False,
        text_pool_type: str = "argmax",
        text_proj_type: str = "linear",
        text_proj_bias: bool = False,
        text_output_tokens: bool = False,
        text_quick_gelu: bool = False,
        decoder_embed_dim: int = 768,
        decoder_num_heads: int = 12,
        decoder_depth: int = 12,
        decoder_ffn_layer: str = "swiglu",
        decoder_norm_layer: str = "layernorm",
        decoder_init_values: Optional[float] = None,
        decoder_use_qk_norm: bool = False,
        init_logit_scale: Optional[float] = None,
        init_logit_bias: Optional[float] = None,
        nonscalar_logit_scale: bool = False,
        **kwargs,
    ):
        loc = dict(locals())
        for k in ("self", "kwargs", "__class__"):
            loc.pop(k, None)
        self.__dict__.update(loc)
        self.extra = dict(kwargs)  # unknown keys of a HF config.json (transformers_version, architectures, ...)
        # train-time RoPE coordinate augmentations (RopePositionEmbedding shift / jitter / rescale, embeddings.py:155-171): constructor
        # arguments `pos_embed_rope_*_coords` of the reference's ViT classes that its HF config does not carry -- they arrive through the
        # legacy training YAML (vision_encoder / pixel_decoder sections are passed on as **kwargs, vtp.py:196-237; from_vtp_yaml below)
        # or as extra keyword arguments here; None = off (the reference's default)
        for k in ROPE_AUG_KEYS:
            setattr(self, k, self.extra.pop(k, None))
        self._validate()

    # what the gfx950 kernels implement today; anything else is rejected loudly rather than silently approximated
    def _validate(self):
        def need(cond, msg):
            if not cond:
                raise ValueError(f"VTPConfig: {msg}")

        need(self.vision_patch_size == 16, "vision_patch_size must be 16 (f16 tokenizer)")
        for pre in ("vision", "decoder", "text"):
            d, h = getattr(self, f"{pre}_embed_dim"), getattr(self, f"{pre}_num_heads")
            need(d % h == 0 and d // h == 64, f"{pre}: head_dim must be 64 (got {d}/{h})")
        for pre in ("vision", "decoder"):  # ffn_layer_dict, vision_transformer.py:22-28
            need(getattr(self, f"{pre}_ffn_layer") in FFN_LAYERS, f"{pre}_ffn_layer must be one of {sorted(FFN_LAYERS)}")
        need(self.vision_norm_layer in ("rmsnorm", "layernorm"), "vision_norm_layer must be rmsnorm|layernorm")
        need(self.decoder_norm_layer in ("rmsnorm", "layernorm"), "decoder_norm_layer must be rmsnorm|layernorm")
        need(self.vision_clip_feat in ("cls", "pooled"), f"Invalid vision_clip_feat: {self.vision_clip_feat}")
        need(self.text_pool_type in ("argmax", "first", "last", "none"), "text_pool_type must be argmax | first | last | none")
        need(self.text_proj_type == "linear" and not self.text_proj_bias, "text projection must be the bias-free matrix")
        for k in ROPE_AUG_KEYS:
            v = getattr(self, k)
            need(v is None or (float(v) > 0 if k.endswith("shift_coords") else float(v) >= 1.0),
                 f"{k} must be None, or > 0 (shift) / >= 1 (jitter, rescale: log-uniform in [1/v, v])")

    @property
    def text_num_pos(self) -> int:
        """rows of positional_embedding = tokens get_clip_text_feature expects: context_length, + 1 with text_embed_cls -- the
        reference class keeps TextTransformer's enlarged positional table and causal mask (text_transformer.py:268-272) but neither
        cls_emb nor the padding mask of build_cls_mask (modeling_vtp.py:163-170), so the extra position is an ordinary token"""
        return int(self.text_context_length) + (1 if self.text_embed_cls else 0)

    def to_dict(self):
        d = {k: v for k, v in self.__dict__.items() if k != "extra" and not (k in ROPE_AUG_KEYS and v is None)}
        d["model_type"] = self.model_type
        return d

    @classmethod
    def from_dict(cls, d):
        d = dict(d)
        d.pop("model_type", None)
        return cls(**d)

    def save_pretrained(self, path: str):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as fh:
            json.dump(self.to_dict(), fh, indent=2)

    @classmethod
    def from_pretrained(cls, path: str):
        with open(os.path.join(path, "config.json")) as fh:
            return cls.from_dict(json.load(fh))

    # legacy training YAML -> HF config (configuration_vtp.py:169-234).  (yaml section, yaml key) -> constructor keyword; a missing
    # optional key keeps the constructor default.  PyYAML instead of the reference's OmegaConf: plain nested mappings suffice.
    _YAML_MAP = {
        ("data", "image_size"): "image_size",
        ("training", "train_clip"): "train_clip", ("training", "train_reconstruction"): "train_reconstruction",
        ("training", "init_logit_scale"): "init_logit_scale", ("training", "init_logit_bias"): "init_logit_bias",
        ("training", "nonscalar_logit_scale"): "nonscalar_logit_scale",
        ("vision", "patch_size"): "vision_patch_size", ("vision", "embed_dim"): "vision_embed_dim", ("vision", "depth"): "vision_depth",
        ("vision", "num_heads"): "vision_num_heads", ("vision", "mlp_ratio"): "vision_mlp_ratio", ("vision", "ffn_layer"): "vision_ffn_layer",
        ("vision", "norm_type"): "vision_norm_layer", ("vision", "init_values"): "vision_init_values",
        ("vision", "use_qk_norm"): "vision_use_qk_norm", ("vision", "vit_feature_bottleneck"): "vision_feature_bottleneck",
        ("vision", "bottleneck_ae_only"): "vision_bottleneck_ae_only", ("vision", "clip_feat"): "vision_clip_feat",
        ("text", "context_length"): "text_context_length", ("text", "vocab_size"): "text_vocab_size", ("text", "embed_dim"): "text_embed_dim",
        ("text", "heads"): "text_num_heads", ("text", "layers"): "text_depth", ("text", "mlp_ratio"): "text_mlp_ratio",
        ("text", "ls_init_value"): "text_ls_init_value", ("text", "embed_cls"): "text_embed_cls", ("text", "pad_id"): "text_pad_id",
        ("text", "no_causal_mask"): "text_no_causal_mask", ("text", "pool_type"): "text_pool_type", ("text", "proj_type"): "text_proj_type",
        ("text", "proj_bias"): "text_proj_bias", ("text", "output_tokens"): "text_output_tokens", ("text", "quick_gelu"): "text_quick_gelu",
        ("decoder", "embed_dim"): "decoder_embed_dim", ("decoder", "num_heads"): "decoder_num_heads", ("decoder", "depth"): "decoder_depth",
        ("decoder", "ffn_layer"): "decoder_ffn_layer", ("decoder", "norm_layer"): "decoder_norm_layer",
        ("decoder", "layerscale_init"): "decoder_init_values", ("decoder", "use_qk_norm"): "decoder_use_qk_norm",
        # (not in the reference's classmethod, whose HF config has no such fields: the legacy class forwards these sections' keys to the
        # ViT constructors as they are, vtp.py:196-237)
        ("vision", "pos_embed_rope_shift_coords"): "vision_rope_shift_coords", ("vision", "pos_embed_rope_jitter_coords"): "vision_rope_jitter_coords",
        ("vision", "pos_embed_rope_rescale_coords"): "vision_rope_rescale_coords",
        ("decoder", "pos_embed_rope_shift_coords"): "decoder_rope_shift_coords", ("decoder", "pos_embed_rope_jitter_coords"): "decoder_rope_jitter_coords",
        ("decoder", "pos_embed_rope_rescale_coords"): "decoder_rope_rescale_coords",
    }
    _YAML_OPTIONAL = {"vision_init_values", "vision_use_qk_norm", "text_ls_init_value", "decoder_init_values", "decoder_use_qk_norm",
                      "init_logit_scale", "init_logit_bias", "nonscalar_logit_scale"} | set(ROPE_AUG_KEYS)

    @classmethod
    def from_vtp_yaml(cls, yaml_path: str) -> "VTPConfig":
        """VTPConfig from a legacy VTP training YAML (sections data / training / vtp_model.{vision_encoder, text_encoder,
        pixel_decoder}) -- same key mapping as the reference's classmethod; a required key that is absent raises KeyError."""
        import yaml
        with open(yaml_path) as fh:
            cfg = yaml.safe_load(fh)
        sections = {"data": cfg["data"], "training": cfg["training"], "vision": cfg["vtp_model"]["vision_encoder"],
                    "text": cfg["vtp_model"]["text_encoder"], "decoder": cfg["vtp_model"]["pixel_decoder"]}
        kw = {}
        for (sec, key), arg in cls._YAML_MAP.items():
            if key in sections[sec]:
                kw[arg] = sections[sec][key]
            elif arg not in cls._YAML_OPTIONAL:
                raise KeyError(f"{yaml_path}: missing {sec}.{key}")
        return cls(**kw)


# ffn_layer_dict (vision_transformer.py:22-28): name -> alignment of the SwiGLU hidden width (None: the plain GELU Mlp, ffn.py:21-48)
FFN_LAYERS = {"mlp": None, "swiglu": 8, "swiglu32": 32, "swiglu64": 64, "swiglu128": 128}


def ffn_hidden(dim: int, ratio: float, ffn_layer: str) -> int:
    """hidden width of a block's FFN: Mlp int(dim * ratio) (block.py:176); SwiGLUFFN 2/3 of that, aligned (ffn.py:71-72)"""
    align = FFN_LAYERS[ffn_layer]
    return int(dim * ratio) if align is None else swiglu_hidden(dim, ratio, align)


def swiglu_hidden(dim: int, ratio: float = 4.0, align_to: int = 8) -> int:
    """SwiGLUFFN hidden width -- ffn.py:71-72 (with mlp_hidden_dim = int(dim*ffn_ratio), block.py:176)."""
    d = int(int(dim * ratio) * 2 / 3)
    return d + (-d % align_to)
