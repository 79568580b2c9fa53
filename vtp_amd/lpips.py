"""LPIPS perceptual distance on the gfx950 kernels -- drop-in for `vtp.utils.lpips.LPIPS` (reference lpips.py:61-100).

Same module tree / state_dict keys as the reference class (`scaling_layer.{shift,scale}`, `net.slice{1..5}.{idx}.{weight,
bias}` with torchvision's vgg16.features indices, `lin{0..4}.model.1.weight`), so a `vgg.pth` loads with load_state_dict.
The reference downloads vgg.pth over HTTP in its constructor (lpips.py:76-82); here the constructor only allocates and the
caller loads weights.  Eval-mode semantics (Dropout = identity; every parameter frozen, lpips.py:74-75).

Compute: zero-bordered NHWC bf16 activation stacks, the 13 VGG convolutions as implicit GEMMs on the MFMA GEMM kernel
(`vtp_conv3x3`: 3x3 taps = row offsets, bias + ReLU + border mask in the epilogue), HBM-bound kernels for unfold / pool / head
(csrc/lpips.hip).  `forward` is the inference API; `loss_and_grad` is what the trainer calls: input = the decoder's
token-major output (PixelShuffle folded into the unfold), gradient accumulated into the token-major `dt`."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
from torch import nn

from . import ops

BF, F32 = torch.bfloat16, torch.float32

# (slice, index in vgg16.features, Cin, Cout); a new slice starts with the 2x2 max-pool (lpips.py:131-146)
VGG_CONVS = [(1, 0, 3, 64), (1, 2, 64, 64), (2, 5, 64, 128), (2, 7, 128, 128), (3, 10, 128, 256), (3, 12, 256, 256),
             (3, 14, 256, 256), (4, 17, 256, 512), (4, 19, 512, 512), (4, 21, 512, 512), (5, 24, 512, 512),
             (5, 26, 512, 512), (5, 28, 512, 512)]
CHNS = [64, 128, 256, 512, 512]
TAP_AFTER = [1, 3, 6, 9, 12]   # conv index whose ReLU output is tap k (relu1_2, relu2_2, relu3_3, relu4_3, relu5_3)


def _holder() -> nn.Module:
    return nn.Module()


class _Stack:
    """bf16 [NB, H+2, W+2, C] zero-bordered pixel rows with W+3 zero guard rows on both sides."""

    def __init__(self, NB: int, H: int, W: int, C: int, device):
        self.NB, self.H, self.W, self.C = NB, H, W, C
        self.rows = NB * (H + 2) * (W + 2)
        g = W + 3
        self.buf = torch.zeros((self.rows + 2 * g) * C, dtype=BF, device=device)
        self.t = self.buf[g * C:(g + self.rows) * C].view(self.rows, C)


class LPIPS(nn.Module):
    def __init__(self, use_dropout: bool = True):
        super().__init__()
        self.chns = list(CHNS)
        self.scaling_layer = _holder()
        self.scaling_layer.register_buffer("shift", torch.tensor([-0.030, -0.088, -0.188])[None, :, None, None])
        self.scaling_layer.register_buffer("scale", torch.tensor([0.458, 0.448, 0.450])[None, :, None, None])
        self.net = _holder()
        for sl in range(1, 6):
            setattr(self.net, f"slice{sl}", _holder())
        for sl, idx, cin, cout in VGG_CONVS:
            conv = _holder()
            conv.register_buffer("weight", torch.zeros(cout, cin, 3, 3))
            conv.register_buffer("bias", torch.zeros(cout))
            getattr(self.net, f"slice{sl}").add_module(str(idx), conv)
        for k, c in enumerate(CHNS):
            lin, model, conv = _holder(), _holder(), _holder()
            conv.register_buffer("weight", torch.zeros(1, c, 1, 1))
            model.add_module("1" if use_dropout else "0", conv)   # index 0 is nn.Dropout when use_dropout (lpips.py:121-123)
            lin.model = model
            setattr(self, f"lin{k}", lin)
        self._lin_idx = "1" if use_dropout else "0"
        self._prepped = None
        self._ws: Dict[tuple, dict] = {}

    # ------------------------------------------------------------------------------------------------ weights
    @torch.no_grad()
    def reset_parameters(self, seed: int = 0):
        """Seeded He-normal stand-in weights (benchmarks / tests; the trained vgg.pth is loaded with load_state_dict)."""
        g = torch.Generator().manual_seed(seed)
        for i, (_, _, cin, cout) in enumerate(VGG_CONVS):
            c = self._conv(i)
            c.weight.copy_(torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5)
            c.bias.copy_(0.05 * torch.randn(cout, generator=g))
        for k, ch in enumerate(CHNS):
            getattr(getattr(self, f"lin{k}").model, self._lin_idx).weight.copy_(torch.rand(1, ch, 1, 1, generator=g) * (2.0 / ch))
        return self

    @staticmethod
    def forward_gflop(H: int, W: int) -> float:
        """algorithmic FLOPs of one VGG16-features pass over one image (2 * MACs of the 13 convolutions)."""
        f, h, w, cur = 0.0, H, W, 1
        for sl, _, cin, cout in VGG_CONVS:
            if sl != cur:
                h, w, cur = h // 2, w // 2, sl
            f += 2.0 * h * w * 9 * cin * cout
        return f / 1e9

    def _conv(self, i: int) -> nn.Module:
        sl, idx, _, _ = VGG_CONVS[i]
        return getattr(getattr(self.net, f"slice{sl}"), str(idx))

    def _version(self) -> int:
        return sum(int(b._version) for b in self.buffers())

    def _prep(self):
        """bf16 GEMM operands: forward [Cout, 9*Cin] in (ky, kx, ci) order; input-gradient [Cin, 9*Cout] with flipped taps."""
        ver = self._version()
        if self._prepped == ver:
            return
        dev = self.scaling_layer.shift.device
        if dev.type != "cuda":
            raise RuntimeError("vtp_amd.LPIPS runs on the MI355X kernels only: move the module to a cuda device (no CPU fallback)")
        self._wf, self._wd, self._bias = [], [], []
        for i, (_, _, cin, cout) in enumerate(VGG_CONVS):
            w = self._conv(i).weight.detach().to(dev, F32)
            wf = w.permute(0, 2, 3, 1).reshape(cout, 9 * cin)
            if i == 0:  # unfolded first layer: K = 27 padded to 32
                wf = torch.cat([wf, torch.zeros(cout, 5, device=dev)], 1)
                wd = wf.t().contiguous()                                   # [32, 64]
            else:
                wd = w.flip(2, 3).permute(1, 2, 3, 0).reshape(cin, 9 * cout)
            self._wf.append(wf.to(BF).contiguous())
            self._wd.append(wd.to(BF).contiguous())
            self._bias.append(self._conv(i).bias.detach().to(dev, F32).contiguous())
        self._lin = [getattr(getattr(self, f"lin{k}").model, self._lin_idx).weight.detach().to(dev, F32).reshape(-1).contiguous()
                     for k in range(5)]
        self._shift = [float(v) for v in self.scaling_layer.shift.flatten().tolist()]
        self._scale = [float(v) for v in self.scaling_layer.scale.flatten().tolist()]
        self._prepped = ver

    # ------------------------------------------------------------------------------------------------ buffers
    def _workspace(self, NB: int, H: int, W: int, n_grad: int) -> dict:
        key = (NB, H, W, n_grad)
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        dev = self.scaling_layer.shift.device
        if H % 16 or W % 16:
            raise ValueError(f"LPIPS input size must be a multiple of 16 (four 2x2 max-pools), got {H}x{W}")
        ws = {"a0": _Stack(NB, H, W, 32, dev), "y": [], "pool": {}, "val": torch.zeros(NB // 2, dtype=F32, device=dev)}
        h, w, cur = H, W, 1
        for i, (sl, _, cin, cout) in enumerate(VGG_CONVS):
            if sl != cur:
                h, w, cur = h // 2, w // 2, sl
                ws["pool"][i] = _Stack(NB, h, w, cin, dev)
            ws["y"].append(_Stack(NB, h, w, cout, dev))
        if n_grad:
            ws["g"] = [_Stack(n_grad, s.H, s.W, s.C, dev) for s in ws["y"]]
            ws["tapg"] = {TAP_AFTER[k]: _Stack(n_grad, ws["y"][TAP_AFTER[k]].H, ws["y"][TAP_AFTER[k]].W, CHNS[k], dev) for k in range(4)}
            ws["dpool"] = {i: _Stack(n_grad, s.H, s.W, s.C, dev) for i, s in ws["pool"].items()}
            ws["da0"] = _Stack(n_grad, H, W, 32, dev)
        self._ws[key] = ws
        return ws

    # ------------------------------------------------------------------------------------------------ compute
    def _features(self, ws: dict, NB: int):
        """VGG16 trunk over the unfolded stack ws['a0'] (all NB images)."""
        x = ws["a0"]
        for i, (_, _, cin, cout) in enumerate(VGG_CONVS):
            y = ws["y"][i]
            if i in ws["pool"]:
                p = ws["pool"][i]
                ops.maxpool2_fwd(x.t, p.t, NB, x.H, x.W, cin)
                x = p
            if i == 0:
                ops.conv3x3(x.t, self._wf[0], self._bias[0], y.t, NB, y.H, y.W, 32, cout, taps=1, mode=0)
            else:
                ops.conv3x3(x.t, self._wf[i], self._bias[i], y.t, NB, y.H, y.W, cin, cout, taps=9, mode=0)
            x = y

    def _taps(self, ws: dict, n: int, grad_scale: Optional[float]):
        """LPIPS head on the five taps: ws['val'][b] = LPIPS(image b, image n + b); tap gradients when grad_scale is set."""
        ws["val"].zero_()
        for k, ci in enumerate(TAP_AFTER):
            y = ws["y"][ci]
            half = n * (y.H + 2) * (y.W + 2)
            df = None
            if grad_scale is not None:
                df = (ws["g"][12] if k == 4 else ws["tapg"][ci]).t
            ops.lpips_tap(y.t, y.t[half:], self._lin[k], ws["val"], df, n, y.H, y.W, CHNS[k],
                          0.0 if grad_scale is None else grad_scale / (y.H * y.W))

    def _backward(self, ws: dict, n: int):
        """input gradient of the first n images (the reconstruction half) down to the unfolded first layer."""
        for i in range(12, 0, -1):
            _, _, cin, cout = VGG_CONVS[i]
            g, y = ws["g"][i], ws["y"][i]
            if i in ws["pool"]:  # conv i reads max-pool(y[i-1]); y[i-1] is a tap
                dp = ws["dpool"][i]
                ops.conv3x3(g.t, self._wd[i], None, dp.t, n, y.H, y.W, cout, cin, taps=9, mode=1, relu_mask=None)
                prev = ws["y"][i - 1]
                ops.maxpool2_bwd(prev.t, dp.t, ws["tapg"][i - 1].t, ws["g"][i - 1].t, n, prev.H, prev.W, cin)
            else:
                ops.conv3x3(g.t, self._wd[i], None, ws["g"][i - 1].t, n, y.H, y.W, cout, cin, taps=9, mode=1,
                            relu_mask=ws["y"][i - 1].t)
        y0 = ws["y"][0]
        ops.conv3x3(ws["g"][0].t, self._wd[0], None, ws["da0"].t, n, y0.H, y0.W, 64, 32, taps=1, mode=1, relu_mask=None)

    # ------------------------------------------------------------------------------------------------ reference API
    @torch.no_grad()
    def forward(self, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        """LPIPS.forward(input, target) (lpips.py:84-100): two [B,3,H,W] images (in [-1,1]) -> [B,1,1,1]."""
        if input.shape != target.shape or input.dim() != 4 or input.shape[1] != 3:
            raise ValueError(f"LPIPS expects two [B,3,H,W] tensors of the same shape, got {tuple(input.shape)} / {tuple(target.shape)}")
        self._prep()
        B, _, H, W = input.shape
        ws = self._workspace(2 * B, H, W, 0)
        a0 = ws["a0"]
        rows = B * (H + 2) * (W + 2)
        ops.lpips_unfold3(None, input.to(F32).contiguous(), a0.t, B, H, W, self._shift, self._scale)
        ops.lpips_unfold3(None, target.to(F32).contiguous(), a0.t[rows:], B, H, W, self._shift, self._scale)
        self._features(ws, 2 * B)
        self._taps(ws, B, None)
        return ws["val"].clone().view(B, 1, 1, 1)

    # ------------------------------------------------------------------------------------------------ training entry
    def loss_and_grad(self, tok: torch.Tensor, target: torch.Tensor, dt: torch.Tensor, weight: float, B: int, H: int, W: int):
        """Perceptual term of the reconstruction loss: weight * mean_b LPIPS(decoded_b, target_b).
        tok bf16 [B*hw, 768]: the pixel decoder's token-major output (pre-PixelShuffle); target f32 [B,3,H,W];
        dt bf16 [B*hw, 768] is ACCUMULATED with the gradient w.r.t. tok.  Returns val f32 [B] (per-image LPIPS, device)."""
        self._prep()
        ws = self._workspace(2 * B, H, W, B)
        a0 = ws["a0"]
        rows = B * (H + 2) * (W + 2)
        ops.lpips_unfold3(tok, None, a0.t, B, H, W, self._shift, self._scale)
        ops.lpips_unfold3(None, target, a0.t[rows:], B, H, W, self._shift, self._scale)
        self._features(ws, 2 * B)
        self._taps(ws, B, weight / B)
        self._backward(ws, B)
        ops.lpips_fold3_bwd(ws["da0"].t, dt, B, H, W, self._scale)
        return ws["val"]
