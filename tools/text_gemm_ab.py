"""The text tower's GEMM shapes (M = 2464 = 32 x 77 rows) and the DINO head's (M = 2816): heuristic vs forced 8-phase kernel.
Usage (GPU box): python tools/text_gemm_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vtp_amd import _lib, ops

lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
SHAPES = [(2464, 3072, 768, ops.EPI_GELU), (2464, 3072, 768, ops.EPI_BF16), (2464, 2304, 768, ops.EPI_BF16), (2464, 768, 3072, ops.EPI_F32),
          (2464, 768, 3072, ops.EPI_BF16), (2464, 768, 2304, ops.EPI_BF16), (2464, 768, 768, ops.EPI_F32), (2464, 768, 768, ops.EPI_BF16),
          (2816, 2048, 2048, ops.EPI_GELU), (2816, 2048, 2048, ops.EPI_BF16), (2816, 2048, 768, ops.EPI_GELU), (2816, 256, 2048, ops.EPI_F32)]
if os.environ.get("SHAPES"):  # "M,N,K,epi;..."
    SHAPES = [tuple(int(v) for v in t.split(",")) for t in os.environ["SHAPES"].split(";")]
for M, N, K, epi in SHAPES:
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    b = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    if epi == ops.EPI_F32:
        c = torch.zeros(M, N, device="cuda")
        kw = dict(bias=bias, resid=c)
    elif epi == ops.EPI_GELU:
        c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        kw = dict(bias=bias, c2=torch.empty(M, N, dtype=torch.bfloat16, device="cuda"))
    else:
        c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        kw = dict(bias=bias)
    res = {}
    CFGS = tuple(int(v) for v in os.environ.get("CFGS", "-1,8").split(","))
    for cfg in CFGS:
        lib.vtp_set_gemm_tuning(cfg, 3)
        for _ in range(3):
            ops.gemm_nt(a, b, c, M=M, N=N, K=K, epi=epi, **kw)
        res[cfg] = []
    for _ in range(7):
        for cfg in CFGS:
            lib.vtp_set_gemm_tuning(cfg, 3)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.gemm_nt(a, b, c, M=M, N=N, K=K, epi=epi, **kw)
            e1.record()
            torch.cuda.synchronize()
            res[cfg].append(e0.elapsed_time(e1) * 100.0)
    lib.vtp_set_gemm_tuning(-1, 3)
    m = {k: sorted(v)[3] for k, v in res.items()}
    print(f"M={M} N={N} K={K} epi={epi}: " + " | ".join(f"cfg {c}: {m[c]:6.1f} us" for c in CFGS), flush=True)
