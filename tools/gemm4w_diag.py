"""Timing of tile configuration 10 (gemm4w) on a few shapes with whatever library VTP_HIP_LIB points at (the W4_DIAG experiment builds
give wrong results by construction: timing only).  Usage (GPU box): VTP_HIP_LIB=... python tools/gemm4w_diag.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vtp_amd import _lib, ops
from tools.gemm8p_bench import timeit


def main():
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(0)
    for M, N, K in [(34144, 768, 4096), (34144, 2304, 768), (8192, 8192, 4096), (16384, 4096, 8192)]:
        a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        b = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
        c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        fns = {}
        for name, cfg in (("cfg8", 8), ("cfg10", 10)):
            def run(cfg=cfg):
                lib.vtp_set_gemm_tuning(cfg, 3)
                ops.gemm_nt(a, b, c, M=M, N=N, K=K, epi=ops.EPI_BF16)
            fns[name] = run
        t = timeit(fns)
        fl = 2.0 * M * N * K
        print(f"M={M:5d} N={N:5d} K={K:5d}: cfg8 {t['cfg8']:7.1f} us {fl / t['cfg8'] / 1e6:7.1f} TF/s | cfg10 {t['cfg10']:7.1f} us "
              f"{fl / t['cfg10'] / 1e6:7.1f} TF/s", flush=True)
    lib.vtp_set_gemm_tuning(-1, 3)


if __name__ == "__main__":
    main()
