"""`patch_model`: put the MI355X path behind an existing instance of the REFERENCE's model class.

A code base that already holds `model = vtp.models.vtp_hf.VTPModel.from_pretrained(...)` (modeling_vtp.py) keeps that object:
its API methods, `forward`, parameter iteration and checkpoint methods are re-bound to a `vtp_amd.VTPModel` built from the same
config and state_dict, so every later call runs the gfx950 kernels (module-level swap, SURVEY.md §8b level A without touching
the import site).  The reference's own nn.Module tree is left in place but no longer consulted."""
from __future__ import annotations

import inspect

import torch

from .config import VTPConfig

API_METHODS = ("get_reconstruction_latents", "get_latents_decoded_images", "get_clip_image_feature", "get_clip_text_feature",
               "get_clip_logits", "get_last_layer_feature", "get_intermediate_layers_feature", "forward")
STATE_METHODS = ("parameters", "named_parameters", "state_dict", "load_state_dict", "zero_grad", "save_pretrained")


def _our_config(ref_config) -> VTPConfig:
    d = ref_config.to_dict() if hasattr(ref_config, "to_dict") else dict(vars(ref_config))
    ours = set(inspect.signature(VTPConfig.__init__).parameters) - {"self"}
    return VTPConfig(**{k: v for k, v in d.items() if k in ours})  # PretrainedConfig adds its own bookkeeping keys


def patch_model(ref_model, device="cuda"):
    """ref_model: an instance with the reference VTPModel's surface (`.config`, `.state_dict()`, the API methods).  Returns the same
    object, now backed by the HIP path; `ref_model._vtp_amd` is the backing `vtp_amd.VTPModel` (hand it to `VTPTrainer`)."""
    from .model import VTPModel
    ours = VTPModel(_our_config(ref_model.config))
    sd = {k: v.detach() for k, v in ref_model.state_dict().items()}
    ours.load_state_dict(sd, strict=True)
    ours = ours.to(device)
    ours.train(bool(getattr(ref_model, "training", False)))
    for name in API_METHODS + STATE_METHODS:
        object.__setattr__(ref_model, name, getattr(ours, name))  # instance attributes shadow the class's methods
    ref_train = ref_model.train if hasattr(ref_model, "train") else None

    def train(mode: bool = True):
        ours.train(mode)
        if ref_train is not None and isinstance(ref_model, torch.nn.Module):
            torch.nn.Module.train(ref_model, mode)
        return ref_model

    object.__setattr__(ref_model, "train", train)
    object.__setattr__(ref_model, "eval", lambda: train(False))
    object.__setattr__(ref_model, "_vtp_amd", ours)
    return ref_model
