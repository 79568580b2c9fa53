"""The prototype-logit GEMM of the DINO head (M = head rows, N = 65536 prototypes, K = 256): tile configurations A/B.
Usage (GPU box): python tools/proto_gemm_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vtp_amd import _lib, ops

lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
for M in (2816, 2560, 128):
    N, K = 65536, 256
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    b = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ref = None
    out = []
    for cfg in (-1, 8, 4, 5, 2):
        lib.vtp_set_gemm_tuning(cfg, 3)
        for _ in range(3):
            ops.gemm_nt(a, b, c, M=M, N=N, K=K, epi=ops.EPI_BF16)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.gemm_nt(a, b, c, M=M, N=N, K=K, epi=ops.EPI_BF16)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 200.0)
        if ref is None:
            ref = c.clone()
        out.append(f"cfg {cfg}: {sorted(ts)[2]:7.1f} us (maxdiff {float((c.float() - ref.float()).abs().max()):.1e})")
    lib.vtp_set_gemm_tuning(-1, 3)
    print(f"M={M} N={N} K={K}: " + " | ".join(out), flush=True)
