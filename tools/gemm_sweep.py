"""Every tile configuration forced on every NT shape of the step (plain epilogues: bf16, fp32 residual, SwiGLU, GELU), against the
default dispatch: where do the thresholds of gemm.hip leave time on the table?  One line per shape: default us, best forced
configuration and its us.  Usage (GPU box): python tools/gemm_sweep.py [workload tag]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vtp_amd import _lib, ops
from tools.gemm8p_bench import timeit

CFGS = [0, 3, 4, 5, 7, 8, 9, 10, 21]
VTPB = [(M, N, K, k) for M in (34144, 16448, 8192) for (N, K, k) in
        ((2304, 768, "bf16"), (768, 768, "f32"), (4096, 768, "swiglu"), (768, 2048, "f32"), (768, 4096, "bf16"), (768, 2304, "bf16"),
         (768, 768, "bf16"), (2048, 768, "bf16"))] + \
       [(2464, N, K, k) for (N, K, k) in ((2304, 768, "bf16"), (768, 768, "f32"), (3072, 768, "gelu"), (768, 3072, "f32"), (768, 3072, "bf16"),
                                           (3072, 768, "bf16"), (768, 2304, "bf16"), (768, 768, "bf16"))]
VTPS = [(M, N, K, k) for M in (16384,) for (N, K, k) in
        ((1152, 384, "bf16"), (384, 384, "f32"), (2048, 384, "swiglu"), (384, 1024, "f32"), (384, 2048, "bf16"), (384, 1152, "bf16"),
         (384, 384, "bf16"), (1024, 384, "bf16"))]


def main():
    shapes = VTPS if len(sys.argv) > 1 and sys.argv[1] == "small" else VTPB
    lib = _lib.load()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    tot_d = tot_b = 0.0
    for M, N, K, kind in shapes:
        a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device=dev, generator=g)
        if kind == "f32":
            c = torch.zeros(M, N, device=dev)
            f = lambda: ops.gemm_nt(a, w, c, M=M, N=N, K=K, bias=bias, resid=c, epi=ops.EPI_F32)
        elif kind == "bf16":
            c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            f = lambda: ops.gemm_nt(a, w, c, M=M, N=N, K=K, bias=bias, epi=ops.EPI_BF16)
        elif kind == "gelu":
            c, c2 = (torch.empty(M, N, dtype=torch.bfloat16, device=dev) for _ in range(2))
            f = lambda: ops.gemm_nt(a, w, c, M=M, N=N, K=K, bias=bias, c2=c2, epi=ops.EPI_GELU)
        else:
            c = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev)
            c2 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            f = lambda: ops.gemm_nt(a, w, c, M=M, N=N, K=K, bias=bias, c2=c2, epi=ops.EPI_SWIGLU)
        res = {}
        lib.vtp_set_gemm_tuning(-1, 3)
        res["default"] = timeit({"x": f}, rounds=5, iters=8)["x"]
        for cfg in CFGS:
            lib.vtp_set_gemm_tuning(cfg, 3)
            try:
                res[cfg] = timeit({"x": f}, rounds=5, iters=8)["x"]
            except Exception:  # a configuration that refuses the shape
                pass
        lib.vtp_set_gemm_tuning(-1, 3)
        best = min((k for k in res if k != "default"), key=lambda k: res[k])
        tot_d += res["default"]
        tot_b += min(res[best], res["default"])
        print(f"M={M:6d} N={N:5d} K={K:5d} {kind:6s}: default {res['default']:7.1f} us | best cfg {best:>3} {res[best]:7.1f} us  x{res['default'] / res[best]:.3f} | "
              + " ".join(f"{k}:{v:.0f}" for k, v in res.items() if k != "default"), flush=True)
    print(f"sum default {tot_d:.1f} us, sum of per-shape best {tot_b:.1f} us ({100 * (1 - tot_b / tot_d):.1f} % on the table)")


if __name__ == "__main__":
    main()
