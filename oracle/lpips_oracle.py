"""CPU restatement of the reference's LPIPS perceptual distance (vtp/utils/lpips.py).  TEST INFRASTRUCTURE -- only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product path is vtp_amd/lpips.py.

Pinned against the real reference class (tests/test_oracle_vs_reference.py::test_lpips, authoring container only) and
against tests/golden/lpips_tiny.safetensors (oracle/make_golden_lpips.py: outputs of the REAL reference class on seeded
weights and inputs).  The pretrained `vgg.pth` is an HTTP download (lpips.py:15-17) and not available offline, so the
weights are seeded random tensors with the reference's state_dict keys and shapes: structural parity only for a18.
"""
import math
from typing import Dict, List

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# torchvision vgg16().features: conv index -> (slice, name inside the slice) as lpips.py:131-146 assigns them
VGG_CONVS = [(1, 0, 3, 64), (1, 2, 64, 64),                       # slice1 = features[0:4]   -> relu1_2
             (2, 5, 64, 128), (2, 7, 128, 128),                   # slice2 = features[4:9]   -> relu2_2 (starts with maxpool 4)
             (3, 10, 128, 256), (3, 12, 256, 256), (3, 14, 256, 256),   # slice3 = features[9:16]  -> relu3_3
             (4, 17, 256, 512), (4, 19, 512, 512), (4, 21, 512, 512),   # slice4 = features[16:23] -> relu4_3
             (5, 24, 512, 512), (5, 26, 512, 512), (5, 28, 512, 512)]   # slice5 = features[23:30] -> relu5_3
CHNS = [64, 128, 256, 512, 512]  # lpips.py:67


def make_state(seed: int = 0) -> Dict[str, Tensor]:
    """Seeded stand-in for vgg.pth with the reference LPIPS module's state_dict keys (lpips.py:63-75,103-124)."""
    g = torch.Generator().manual_seed(seed)
    sd = {"scaling_layer.shift": torch.tensor([-0.030, -0.088, -0.188])[None, :, None, None],
          "scaling_layer.scale": torch.tensor([0.458, 0.448, 0.450])[None, :, None, None]}
    for sl, idx, cin, cout in VGG_CONVS:
        sd[f"net.slice{sl}.{idx}.weight"] = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * cin))
        sd[f"net.slice{sl}.{idx}.bias"] = 0.05 * torch.randn(cout, generator=g)
    for k, c in enumerate(CHNS):
        sd[f"lin{k}.model.1.weight"] = torch.rand(1, c, 1, 1, generator=g) * (2.0 / c)  # non-negative like the trained ones
    return sd


def vgg_features(sd: Dict[str, Tensor], x: Tensor) -> List[Tensor]:
    """lpips.py:148-166: the five ReLU taps of vgg16.features."""
    taps, h, cur = [], x, 1
    for sl, idx, _, _ in VGG_CONVS:
        if sl != cur:  # a new slice starts with the 2x2 max-pool (features[4], [9], [16], [23])
            taps.append(h)
            h = F.max_pool2d(h, 2, 2)
            cur = sl
        h = F.relu(F.conv2d(h, sd[f"net.slice{sl}.{idx}.weight"], sd[f"net.slice{sl}.{idx}.bias"], padding=1))
    taps.append(h)
    return taps


def normalize_tensor(x: Tensor, eps: float = 1e-10) -> Tensor:
    """lpips.py:169-171."""
    return x / (torch.sqrt(torch.sum(x ** 2, dim=1, keepdim=True)) + eps)


def lpips(sd: Dict[str, Tensor], inp: Tensor, target: Tensor) -> Tensor:
    """LPIPS.forward (lpips.py:84-100), eval mode (Dropout = identity): [B,3,H,W] x2 in [-1,1] -> [B,1,1,1]."""
    x0 = (inp - sd["scaling_layer.shift"]) / sd["scaling_layer.scale"]      # ScalingLayer, lpips.py:113-114
    x1 = (target - sd["scaling_layer.shift"]) / sd["scaling_layer.scale"]
    f0, f1 = vgg_features(sd, x0), vgg_features(sd, x1)
    val = None
    for k in range(len(CHNS)):
        d = (normalize_tensor(f0[k]) - normalize_tensor(f1[k])) ** 2
        r = F.conv2d(d, sd[f"lin{k}.model.1.weight"]).mean([2, 3], keepdim=True)  # NetLinLayer + spatial_average
        val = r if val is None else val + r
    return val


def lpips_loss(sd, inp: Tensor, target: Tensor) -> Tensor:
    """Perceptual term of the reconstruction objective as our trainer defines it (unpinned): batch mean of LPIPS."""
    return lpips(sd, inp, target).mean()
