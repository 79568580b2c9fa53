"""Generates tests/golden/lpips_tiny.safetensors from the REAL reference LPIPS class (authoring container only).
TEST INFRASTRUCTURE.   PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_lpips.py

The reference builds its backbone from torchvision.models.vgg16 and downloads vgg.pth; neither exists offline, so the
stub supplies a same-architecture `features` Sequential and the weights are oracle.lpips_oracle.make_state(seed) --
regenerated from the seed by the tests (59 MB of VGG weights do not belong in a fixture).  Stored: inputs, the reference's
LPIPS values and its autograd gradient of mean(LPIPS) w.r.t. the first input."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safetensors.torch import save_file

from oracle.lpips_oracle import make_state
from oracle.ref_stubs import load_reference_lpips

SEED, B, RES = 7, 2, 64


def main(out_path):
    LPIPS = load_reference_lpips()
    m = LPIPS(use_dropout=True).eval()
    missing, unexpected = m.load_state_dict(make_state(SEED), strict=True), None
    g = torch.Generator().manual_seed(11)
    x0 = (torch.rand(B, 3, RES, RES, generator=g) * 2 - 1).requires_grad_(True)
    x1 = torch.rand(B, 3, RES, RES, generator=g) * 2 - 1
    val = m(x0, x1)
    val.mean().backward()
    save_file({"x0": x0.detach().contiguous(), "x1": x1.contiguous(), "lpips": val.detach().contiguous(),
               "grad_x0": x0.grad.contiguous(), "seed": torch.tensor([SEED])}, out_path)
    print("wrote", out_path, "lpips =", val.flatten().tolist())


if __name__ == "__main__":
    main(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "lpips_tiny.safetensors"))
