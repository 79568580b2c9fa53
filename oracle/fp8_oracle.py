"""e4m3-SIMULATED forward of the reference trunk + pixel decoder -- TEST INFRASTRUCTURE ONLY (same rules as vtp_oracle.py).

BASELINE config 5 ("fp8 MFMA forward") has no counterpart in the reference (it has no fp8 path), so its comparator is the
reference algorithm (oracle/vtp_oracle.py, pinned to /root/reference) with ONE change: the operands of the four linear maps
of every ViT block -- qkv, proj, w1|w2, w3 (vtp/models/layers/attention.py:92-96, ffn.py:77-81) -- take a round trip through
torch.float8_e4m3fn (OCP e4m3, saturating at 448) with per-tensor scales:

  * weights:      s_W = 448 / max|W|   (w1 and w2 share one scale: the product path multiplies the fused [2H, D] matrix);
  * activations:  s_A = 448 / amax, amax calibrated by a pass of the UNQUANTISED algorithm over calibration images
                  (`calibrate`), one value per (block, site) -- the recipe of vtp_amd.VTPModel.enable_fp8_forward;
  * product:      exact e4m3 x e4m3 products accumulated in fp32 (torch fp32 matmul of the e4m3 values, autocast off),
                  times 1 / (s_A s_W), plus the fp32 bias; the result takes the dtype the reference's linear would return
                  (bf16 under autocast, fp32 otherwise).

Everything else (norms, RoPE, SDPA, SwiGLU, residuals, patch embed, bottleneck, proj_in / proj_out) is the reference
algorithm untouched.  Parity statement built on it (tests/test_fp8_gpu.py): |ours_fp8 - ref_fp32| <= 1.25 x |sim_fp8 - ref_fp32|.
Parity pinning: inherits vtp_oracle's (the e4m3 cast is torch's own); the quantisation recipe itself is OUR spec (unpinned)."""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from . import vtp_oracle as O

E4M3_MAX = 448.0


def _q(x: torch.Tensor, scale: float) -> torch.Tensor:
    """e4m3 values of x * scale as fp32 (saturating, round-to-nearest-even: torch's cast)"""
    return (x.float() * scale).clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).float()


class Fp8Sim:
    """lin-hook for vtp_oracle.{trunk_forward, decoder_forward}: mode "calibrate" records max|activation| per site with plain
    F.linear; mode "apply" runs the simulated e4m3 GEMM."""

    def __init__(self, sd: Dict[str, torch.Tensor]):
        self.sd, self.amax, self.mode = sd, {}, "calibrate"
        self._wq = {}

    def _site(self, site: str) -> str:  # w1 and w2 read the same activation and share the fused weight's scale
        return site[:-2] + "w12" if site.endswith((".w1", ".w2")) else site

    def _weight(self, site: str, w: torch.Tensor):
        key = (site, w.device)
        if key not in self._wq:
            if site.endswith((".w1", ".w2")):
                pre = site[:-2]
                wa = float(torch.maximum(self.sd[pre + "w1.weight"].abs().max(), self.sd[pre + "w2.weight"].abs().max()))
            else:
                wa = float(w.abs().max())
            s = E4M3_MAX / max(wa, 1e-12)
            self._wq[key] = (_q(w.detach(), s), s)
        return self._wq[key]

    def __call__(self, x, w, b, site):
        k = self._site(site)
        if self.mode == "calibrate":
            self.amax[k] = max(self.amax.get(k, 0.0), float(x.detach().float().abs().max()))
            return F.linear(x, w, b)
        sa = E4M3_MAX / max(self.amax[k], 1e-12)
        wq, sw = self._weight(site, w)
        out_dtype = torch.bfloat16 if torch.is_autocast_enabled(x.device.type) else torch.float32
        with torch.autocast(x.device.type, enabled=False):
            y = F.linear(_q(x, sa), wq) * (1.0 / (sa * sw))
            if b is not None:
                y = y + b.float()
        return y.to(out_dtype)


def encode_decode(sd, img, vis_heads: int, dec_heads: int, lin=O._plain_linear):
    """get_reconstruction_latents -> get_latents_decoded_images (modeling_vtp.py:337-360,397-414) with the given linear hook"""
    lat = O.reconstruction_latents(sd, img, vis_heads, lin=lin)
    return lat, O.decoder_forward(sd, lat.float(), dec_heads, lin=lin)


def calibrate(sd, calib_img, vis_heads: int, dec_heads: int) -> Fp8Sim:
    sim = Fp8Sim(sd)
    with torch.no_grad():
        encode_decode(sd, calib_img, vis_heads, dec_heads, lin=sim)
    sim.mode = "apply"
    return sim
