import os, sys
sys.path.insert(0, "/root/repo")
import torch
from vtp_amd import _lib, ops
lib = _lib.load()
M, N, K = 34144, 2304, 768
a = torch.randn(M, K, device="cuda").bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.05).bfloat16(); c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
def t(cfg, n=20):
    lib.vtp_set_gemm_tuning(cfg, 3)
    for _ in range(3): ops.gemm_nt(a, b, c, M=M, N=N, K=K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.gemm_nt(a, b, c, M=M, N=N, K=K)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
print("cfg", sys.argv[1], "us", t(int(sys.argv[1])))
