"""VTP -- the *training* meta-architecture (reference: vtp/models/vtp.py:88-552) on the MI355X kernels.

`VTP` = `VTPModel` (trunk, pixel decoder, text tower, CLIP projection; same checkpoint keys as the HF class) plus the
self-supervised branch of the legacy class, under the legacy attribute names:
    dino_head.{mlp.0,mlp.2,mlp.4}.{weight,bias}, dino_head.last_layer.{weight_g,weight_v}     (dino_head.py:7-89)
    teacher_trunk.*  (frozen EMA copy of trunk.*),  teacher_dino_head.*                      (vtp.py:253-268)
and its methods `update_teacher(momentum)` (vtp.py:388-401), `get_ssl_params()` (vtp.py:403-407) and
`forward_ssl_learning(**ssl_dict)` (vtp.py:365-385, output dict keys of vtp.py:446-448,479-484).
The reference configures this class from an OmegaConf tree; here the same knobs are constructor arguments."""
from __future__ import annotations

import copy
from typing import Dict, Optional

import numpy as np
import torch
from torch import nn

from . import autograd as ag
from . import ops
from .config import VTPConfig
from .engine import BF, F32, OVERLAP, TrunkEngine
from .model import VTPModel, _holder, _param
from .ssl_engine import DinoHeadEngine, build_ssl_indices

I32 = torch.int32


def _dino_head_tree(in_dim: int, hidden: int, bott: int, K: int) -> nn.Module:
    h = _holder()
    h.mlp = nn.ModuleList([_holder() for _ in range(5)])  # indices 0, 2, 4 carry parameters (1, 3 are GELUs)
    for i, (o, k) in zip((0, 2, 4), ((hidden, in_dim), (hidden, hidden), (bott, hidden))):
        h.mlp[i].weight = _param(o, k)
        h.mlp[i].bias = _param(o)
    h.last_layer = _holder()
    h.last_layer.weight_g = _param(K, 1)
    h.last_layer.weight_v = _param(K, bott)
    return h


class VTP(VTPModel):
    def __init__(self, config: VTPConfig, dino_out_dim: int = 65536, dino_hidden_dim: int = 2048,
                 dino_bottleneck_dim: int = 256, dino_nlayers: int = 3):
        super().__init__(config)
        if dino_nlayers != 3:
            raise ValueError("only the 3-layer DINO head MLP is implemented")
        if not config.vision_bottleneck_ae_only:
            raise NotImplementedError("the DINO head reads the un-bottlenecked trunk features (bottleneck_ae_only=True)")
        D = config.vision_embed_dim
        self.dino_cfg = dict(in_dim=D, hidden=dino_hidden_dim, bott=dino_bottleneck_dim, K=dino_out_dim)
        self.dino_head = _dino_head_tree(D, dino_hidden_dim, dino_bottleneck_dim, dino_out_dim)
        with torch.no_grad():
            for i in (0, 2, 4):
                nn.init.trunc_normal_(self.dino_head.mlp[i].weight, std=0.02)
                self.dino_head.mlp[i].bias.zero_()
            nn.init.trunc_normal_(self.dino_head.last_layer.weight_v, std=0.02)
            self.dino_head.last_layer.weight_g.fill_(1.0)
        self.teacher_trunk = copy.deepcopy(self.trunk)
        self.teacher_dino_head = copy.deepcopy(self.dino_head)
        # legacy layout: the image projection is `proj` with an EMA copy `teacher_proj` (vtp.py:217,249); the HF class calls
        # it `visual_proj`.  One parameter under the HF name, `proj` as an alias, `teacher_proj` as its own frozen parameter.
        self.teacher_proj = copy.deepcopy(self.visual_proj) if self.visual_proj is not None else None
        frozen = list(self.teacher_trunk.parameters()) + list(self.teacher_dino_head.parameters())
        if self.teacher_proj is not None:
            frozen += list(self.teacher_proj.parameters())
        for p in frozen:
            p.requires_grad = False
        self.enable_teacher = True
        self.output_dict = True  # training.clip_output_dict (vtp.py:189)

    # ------------------------------------------------------------------------------------------------ legacy layout
    @property
    def proj(self):
        return self.visual_proj

    @property
    def transformer(self):
        return getattr(self, "text_transformer", None)

    _LEGACY_PREFIXES = (("proj.", "visual_proj."), ("transformer.", "text_transformer."))

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        """Accepts the HF-layout keys (VTPModel) and the legacy training-class layout (vtp/models/vtp.py: `proj`,
        `teacher_proj`, `transformer.resblocks.*`) -- a checkpoint of the reference's `VTP` loads with strict=True."""
        sd = {}
        for k, v in state_dict.items():
            for old, new in self._LEGACY_PREFIXES:
                if k.startswith(old):
                    k = new + k[len(old):]
                    break
            sd[k] = v
        return super().load_state_dict(sd, strict=strict, **kw)

    def legacy_state_dict(self) -> Dict[str, torch.Tensor]:
        """state_dict under the reference legacy class's key names (inverse of the mapping load_state_dict applies)."""
        out = {}
        for k, v in self.state_dict().items():
            for old, new in self._LEGACY_PREFIXES:
                if k.startswith(new):
                    k = old + k[len(new):]
                    break
            out[k] = v
        return out

    # ------------------------------------------------------------------------------------------------ engine
    def _build_extra_engines(self, st):
        c = self.dino_cfg
        self._t_trunk = TrunkEngine(st, self.config, self.teacher_trunk.rope_embed.periods, prefix="teacher_trunk.")
        self._head = DinoHeadEngine(st, "dino_head.", c["in_dim"], c["hidden"], c["bott"], c["K"])
        self._t_head = DinoHeadEngine(st, "teacher_dino_head.", c["in_dim"], c["hidden"], c["bott"], c["K"])

    # ------------------------------------------------------------------------------------------------ legacy API
    def get_ssl_params(self):
        return list(self.dino_head.parameters()) if self.enable_teacher else []

    @torch.no_grad()
    def update_teacher(self, momentum: float):
        """teacher = m * teacher + (1 - m) * student over trunk and dino_head parameters (vtp.py:388-401): two fused
        launches over contiguous ranges of the flat parameter buffer."""
        if not self.enable_teacher:
            return
        st = self._engine()
        for t_pref, s_pref in self.ema_pairs():
            (tlo, thi), (slo, shi) = _range(st, t_pref), _range(st, s_pref)
            assert thi - tlo == shi - slo
            ops.ema(st.flat_p[tlo:thi], st.flat_p[slo:shi], thi - tlo, momentum)
        st.prep()
        self._pver = self._param_version()

    def ema_pairs(self):
        """(teacher prefix, student prefix) of every EMA-tracked parameter group (vtp.py:388-401: trunk, proj, dino_head)"""
        pairs = [("teacher_trunk.", "trunk.")]
        if self.teacher_proj is not None:
            pairs.append(("teacher_proj.", "visual_proj."))
        pairs.append(("teacher_dino_head.", "dino_head."))
        return pairs

    # ------------------------------------------------------------------------------------------------ legacy forward surface
    def encode_image(self, image: torch.Tensor, normalize: bool = False) -> torch.Tensor:
        """vtp.py:275-291 (default normalize=False, unlike the HF class)."""
        return self.get_clip_image_feature(image, normalize=normalize)

    def encode_text(self, text: torch.Tensor, normalize: bool = False) -> torch.Tensor:
        """vtp.py:293-312."""
        return self.get_clip_text_feature(text, normalize=normalize)

    def get_logits(self, image: torch.Tensor, text: torch.Tensor):
        """vtp.py:314-321."""
        return self.get_clip_logits(image, text)

    def forward_clip(self, image: Optional[torch.Tensor], text: Optional[torch.Tensor]):
        """vtp.py:340-360."""
        image_features = self.encode_image(image, normalize=True) if image is not None else None
        text_features = self.encode_text(text, normalize=True) if text is not None else None
        with torch.set_grad_enabled(ag.grad_mode(self)):
            scale = self.logit_scale.exp()
        if self.output_dict:
            out = {"image_features": image_features, "text_features": text_features, "logit_scale": scale}
            if self.logit_bias is not None:
                out["logit_bias"] = self.logit_bias
            return out
        if self.logit_bias is not None:
            return image_features, text_features, scale, self.logit_bias
        return image_features, text_features, scale

    def forward_reconstruction(self, reconstruction_image: torch.Tensor):
        """vtp.py:362-363 -> get_reconstruction_outputs (vtp.py:487-512)."""
        if self.pixel_decoder is None:
            return {}
        lat = self.get_reconstruction_latents(reconstruction_image)
        return {"reconstructed_image": self.get_latents_decoded_images(lat), "target_image": reconstruction_image}

    def forward(self, image=None, text=None, ssl_dict: Optional[dict] = None, reconstruction_image=None,
                forward_type: str = "clip"):
        """The legacy training forward (vtp.py:323-338): forward_type in {clip, ssl, rec}; differentiable in training mode."""
        assert forward_type in ["clip", "ssl", "rec"], "Invalid forward type"
        if forward_type == "clip":
            return self.forward_clip(image, text)
        if forward_type == "ssl":
            return self.forward_ssl_learning(**ssl_dict)
        return self.forward_reconstruction(reconstruction_image)

    def forward_ssl_learning(self, global_crops, n_global_crops, mask_indices_list, n_masked_patches, upperbound,
                             local_crops, masks, masks_weight=None):
        """SSL forward with the legacy signature / outputs (vtp.py:365-385): (teacher_outputs, student_outputs) with the
        reference's dict keys; logits are f32 copies of the bf16 kernels' outputs.  In training mode the student outputs are
        differentiable (autograd.SSLStudent); the fused training step is VTPTrainer.step(..., ssl=...)."""
        if n_global_crops != 2:
            raise NotImplementedError("n_global_crops must be 2")
        if ag.grad_mode(self):
            return self._forward_ssl_autograd(global_crops, local_crops, masks, n_masked_patches, upperbound)
        return self._forward_ssl_nograd(global_crops, n_global_crops, mask_indices_list, n_masked_patches, upperbound, local_crops,
                                        masks)

    def _forward_ssl_autograd(self, global_crops, local_crops, masks, n_masked_patches, upperbound):
        self._fresh()
        B2 = global_crops.shape[0]
        B = B2 // 2
        hw = (global_crops.shape[-2] // 16) * (global_crops.shape[-1] // 16)
        hw_l = (local_crops.shape[-2] // 16) * (local_crops.shape[-1] // 16)
        n_local = local_crops.shape[0] // B
        plan = build_ssl_indices(masks.detach().cpu().numpy().astype(bool), B, hw, n_local, hw_l, 1.0, 1.0, pad_to=8)
        assert plan["n_masked"] == int(n_masked_patches) <= int(upperbound)
        m8 = masks.to(self._store.device).to(torch.uint8).contiguous()
        s_loc, s_glob, s_cls, s_patch, t_logits = ag.SSLStudent.apply(ag.anchor(self), self, self._img(global_crops),
                                                                      self._img(local_crops), m8, plan)
        nm = plan["n_masked"]
        teacher_outputs = {"teacher_cls_tokens_after_head": t_logits[:B2], "n_masked_patches": n_masked_patches,
                           "masked_teacher_patch_tokens_after_head": t_logits[B2:B2 + nm]}
        student_outputs = {"student_local_cls_tokens_after_head": s_loc, "student_global_cls_tokens_after_head": s_glob,
                           "student_global_cls_tokens": s_cls, "student_global_masked_patch_tokens_after_head": s_patch}
        return teacher_outputs, student_outputs

    @torch.no_grad()
    def _forward_ssl_nograd(self, global_crops, n_global_crops, mask_indices_list, n_masked_patches, upperbound, local_crops,
                            masks):
        self._fresh()
        B2 = global_crops.shape[0]
        B = B2 // 2
        hw = (global_crops.shape[-2] // 16) * (global_crops.shape[-1] // 16)
        hw_l = (local_crops.shape[-2] // 16) * (local_crops.shape[-1] // 16)
        n_local = local_crops.shape[0] // B
        plan = build_ssl_indices(masks.detach().cpu().numpy().astype(bool), B, hw, n_local, hw_l, 1.0, 1.0, pad_to=8)
        assert plan["n_masked"] == int(n_masked_patches) <= int(upperbound)
        out = ssl_forward(self, self._img(global_crops), self._img(local_crops), masks.to(self._store.device), plan, train=False)
        nm, nl = plan["n_masked"], n_local * B
        tl, sl = out["teacher_logits"].float(), out["student_logits"].float()
        teacher_outputs = {"teacher_cls_tokens_after_head": tl[:B2].clone(), "n_masked_patches": n_masked_patches,
                           "masked_teacher_patch_tokens_after_head": tl[B2:B2 + nm].clone()}
        student_outputs = {"student_local_cls_tokens_after_head": sl[:nl].clone(),
                           "student_global_cls_tokens_after_head": sl[nl:nl + B2].clone(),
                           "student_global_cls_tokens": out["student_global_cls"].float().clone(),
                           "student_global_masked_patch_tokens_after_head": sl[nl + B2:nl + B2 + nm].clone()}
        return teacher_outputs, student_outputs


def _range(st, prefix):
    offs = [(o, o + (k + 3) // 4 * 4) for n, (o, k) in st.offsets.items() if n.startswith(prefix)]
    lo, hi = min(o for o, _ in offs), max(h for _, h in offs)
    assert sum(h - o for o, h in offs) == hi - lo, f"{prefix} is not contiguous in the flat buffer"
    return lo, hi


def plan_to_device(plan, device):
    """int32 / f32 index tensors of one SSL batch on the device (done outside any graph capture)."""
    d = {k: torch.as_tensor(plan[k], device=device) for k in ("teacher_src", "student_local_src", "student_global_src", "t0", "t1")}
    d["w"] = torch.as_tensor(plan["w"], device=device)
    # the masked-token count travels in device memory: kernels read it there, so one captured graph serves every mask draw
    # that fits the same padded buffers
    d["n_masked_i"] = torch.tensor([plan["n_masked"]], dtype=torch.int32, device=device)
    d["n_masked_f"] = torch.tensor([float(plan["n_masked"])], dtype=torch.float32, device=device)
    return d


def ssl_forward(model: "VTP", global_crops, local_crops, masks_u8, plan, dev_plan=None, train: bool = False,
                lead_images=None):
    """Teacher + student forward of one SSL batch (vtp.py:410-484).  Returns a dict with the head logits and the
    context the backward needs.  masks_u8: uint8 [2B, hw] on the device; plan: build_ssl_indices() output.
    The student's passes (masked global crops, local crops and -- when `lead_images` is given -- the clean images of the
    rec / clip objectives, which use the same trunk weights) go through the trunk as ONE list forward (item order:
    lead, global, local)."""
    st = model._store
    B2 = global_crops.shape[0]
    hw = (global_crops.shape[-2] // 16) * (global_crops.shape[-1] // 16)
    N = hw + 1
    D = model.config.vision_embed_dim
    idx = plan_to_device(plan, st.device) if dev_plan is None else dev_plan
    Tm, Ts = plan["Tm"], plan["Ts"]
    Tt = B2 + Tm
    if masks_u8.dtype != torch.uint8:
        masks_u8 = masks_u8.to(torch.uint8).contiguous()
    # ---- teacher (EMA weights, clean input, no gradient): vtp.py:410-450.  Independent of the student pass until the loss:
    # it runs on the side stream and fills the CUs the student's GEMM tails leave idle (joined before the logits are used)
    ws = model._head.workspace(Ts, "ssl_io")
    Xt = ws.get("Xt", (Tt, D), BF)

    def teacher():
        xnf_t = model._t_trunk.forward(global_crops, train=False, tag="teacher", rope_aug=train)  # (teacher_trunk is in training mode too)
        ops.gather_token_rows(xnf_t, idx["teacher_src"], Xt, Tt, D)
        return model._t_head.forward(Xt, Tt, tag="teacher")[0]

    if OVERLAP.enabled:
        OVERLAP.join()
        OVERLAP.fork()
        with torch.cuda.stream(OVERLAP.side):
            t_logits = teacher()
    else:
        t_logits = teacher()
    # ---- student: masked global crops + local crops through the SAME trunk weights (vtp.py:452-484)
    # lead_images: one tensor, or a list of tensors (separate clip / reconstruction inputs, vtp.py:323-338): one list item each
    leads = [] if lead_images is None else (list(lead_images) if isinstance(lead_images, (list, tuple)) else [lead_images])
    items = [(im, None) for im in leads] + [(global_crops, masks_u8), (local_crops, None)]
    xnf = model._trunk.forward_list(items, train=train, tag="ssl")
    ctx = model._trunk.ctx()
    seg_g, seg_l = ctx.segs[-2], ctx.segs[-1]
    nl = int(plan["student_local_src"].shape[0])
    Xs = ws.get("Xs", (Ts, D), BF)
    ops.gather_token_rows(xnf[seg_l.row0:], idx["student_local_src"], Xs, nl, D)
    ops.gather_token_rows(xnf[seg_g.row0:], idx["student_global_src"], Xs[nl:], Ts - nl, D)
    s_logits, head_ctx = model._head.forward(Xs, Ts, tag="student")
    OVERLAP.join()  # teacher logits complete
    return dict(teacher_logits=t_logits, student_logits=s_logits, head_ctx=head_ctx, ctx=ctx, xnf=xnf, idx=idx,
                Xs=Xs, student_global_cls=Xs[nl:nl + B2], Tt=Tt, Ts=Ts, Tm=Tm, nl=nl, B2=B2, N=N, ws=ws)
