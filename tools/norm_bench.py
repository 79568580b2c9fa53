"""norm_bwd / norm_fwd at the step's row counts: us and TB/s (algorithmic bytes).  VTP_NORM_BWD_BLOCKS=n caps the backward grid.
Usage (GPU box): python tools/norm_bench.py [tag]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vtp_amd import ops
from tools.gemm8p_bench import timeit


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("VTP_NORM_BWD_BLOCKS", "default")
    dev, D = "cuda", 768
    g = torch.Generator(device=dev).manual_seed(0)
    for M, kind in ((34144, ops.NORM_RMS), (16448, ops.NORM_RMS), (8192, ops.NORM_LN), (2464, ops.NORM_LN)):
        x = torch.randn(M, D, device=dev, generator=g)
        w, b = torch.randn(D, device=dev, generator=g), torch.randn(D, device=dev, generator=g)
        y = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
        st = torch.empty(M, 2, device=dev)
        dy = torch.randn(M, D, device=dev, generator=g).to(torch.bfloat16)
        dres = torch.randn(M, D, device=dev, generator=g)
        dx, dxb = torch.empty(M, D, device=dev), torch.empty(M, D, dtype=torch.bfloat16, device=dev)
        dw, db, cs = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
        bias = b if kind == ops.NORM_LN else None
        ops.norm_fwd(x, w, bias, y, st, M, D, 1e-5, kind)
        t = timeit({"fwd": lambda: ops.norm_fwd(x, w, bias, y, st, M, D, 1e-5, kind),
                    "bwd": lambda: ops.norm_bwd(dy, x, w, st, dres, dx, dxb, dw, db if kind == ops.NORM_LN else None, M, D, kind, dx_colsum=cs)})
        print(f"[{tag}] M={M:6d} kind={kind}: fwd {t['fwd']:6.1f} us {M * D * 6 / t['fwd'] / 1e6:5.2f} TB/s | bwd {t['bwd']:6.1f} us "
              f"{M * D * 16 / t['bwd'] / 1e6:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
