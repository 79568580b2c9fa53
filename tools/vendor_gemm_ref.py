"""Context for the GEMM numbers: the vendor library (torch.matmul -> hipBLASLt, bf16, plain C = A B^T, no fused epilogue) against
this repository's dispatch with the plain bf16 epilogue on the NT shapes of the VTP-B train step.  Measurement aid only -- the
product path never calls a library GEMM.  Interleaved rounds, median.  Usage (GPU box): python tools/vendor_gemm_ref.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vtp_amd import _lib, ops
from tools.gemm8p_bench import timeit

SHAPES = [(34144, 2304, 768), (34144, 4096, 768), (34144, 768, 768), (34144, 768, 2048), (34144, 768, 2304), (34144, 768, 4096),
          (34144, 2048, 768), (16448, 2304, 768), (16448, 4096, 768), (8192, 2304, 768), (8192, 768, 4096), (8192, 4096, 768),
          (2464, 2304, 768), (2464, 768, 3072), (8192, 8192, 4096)]


def main():
    lib = _lib.load()
    lib.vtp_set_gemm_tuning(-1, 3)
    g = torch.Generator(device="cuda").manual_seed(0)
    for M, N, K in SHAPES:
        a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        b = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
        c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        c2 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        bt = b.t()
        t = timeit({"ours": lambda: ops.gemm_nt(a, b, c, M=M, N=N, K=K, epi=ops.EPI_BF16), "vendor": lambda: torch.matmul(a, bt, out=c2)})
        fl = 2.0 * M * N * K
        d = float((c.float() - c2.float()).abs().max())
        print(f"M={M:5d} N={N:5d} K={K:5d}: ours {t['ours']:7.1f} us {fl / t['ours'] / 1e6:7.1f} TF/s | hipBLASLt {t['vendor']:7.1f} us "
              f"{fl / t['vendor'] / 1e6:7.1f} TF/s  ours/vendor speed x{t['vendor'] / t['ours']:.2f}  maxdiff {d:.2e}", flush=True)


if __name__ == "__main__":
    main()
