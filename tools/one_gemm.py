"""One GEMM shape, a few launches -- the target of rocprofv3 counter passes (scripts/gpu_pmc_gemm.sh).
usage: python tools/one_gemm.py <nt|tn> M N K cfg [splits]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vtp_amd import _lib, ops

kind, M, N, K, cfg = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
splits = int(sys.argv[6]) if len(sys.argv) > 6 else 1
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
lib.vtp_set_gemm_tuning(cfg, 3)
if kind == "nt":
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    b = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for _ in range(5):
        ops.gemm_nt(a, b, c, M=M, N=N, K=K, epi=ops.EPI_BF16)
else:
    a = torch.randn(K, M, device="cuda", generator=g).to(torch.bfloat16)
    b = torch.randn(K, N, device="cuda", generator=g).to(torch.bfloat16)
    S = ops.gemm_splits(K, splits)
    slab = torch.empty(S * M * N, device="cuda")
    for _ in range(5):
        ops.gemm_tn(a, b, slab, M=M, N=N, K=K, lda=M, ldb=N, ldc=N, ldc2=M * N // 4, epi=ops.EPI_F32_SLAB, splits=S)
torch.cuda.synchronize()
