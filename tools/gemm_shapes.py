"""The step's NT GEMM launches through the default dispatch, one line per (shape, epilogue): us and TFLOP/s.  For A/B runs of two builds
(VTP_HIP_LIB=<other libvtp_hip.so>) in separate processes.  Usage (GPU box): python tools/gemm_shapes.py [tag]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vtp_amd import ops
from tools.gemm8p_bench import timeit

SHAPES = [  # (tag, M, N, K, kind)
    ("proj f32res", 34144, 768, 768, "f32"), ("w3 f32res", 34144, 768, 2048, "f32"), ("qkv+rope", 34144, 2304, 768, "rope"),
    ("w12 swiglu", 34144, 4096, 768, "swiglu"), ("w3 dgrad+swiglu bwd", 34144, 2048, 768, "dsw"), ("w12 dgrad", 34144, 768, 4096, "bf16"),
    ("qkv dgrad", 34144, 768, 2304, "bf16"), ("proj dgrad", 34144, 768, 768, "bf16"), ("qkv plain", 34144, 2304, 768, "bf16"),
    ("proj f32res T", 16448, 768, 768, "f32"), ("w3 f32res T", 16448, 768, 2048, "f32"), ("qkv+rope T", 16448, 2304, 768, "rope"),
    ("w12 swiglu T", 16448, 4096, 768, "swiglu"),
    ("proj f32res D", 8192, 768, 768, "f32"), ("w3 f32res D", 8192, 768, 2048, "f32"), ("qkv+rope D", 8192, 2304, 768, "rope"),
    ("w12 swiglu D", 8192, 4096, 768, "swiglu"), ("w3 dgrad+swiglu D", 8192, 2048, 768, "dsw"), ("w12 dgrad D", 8192, 768, 4096, "bf16"),
    ("qkv dgrad D", 8192, 768, 2304, "bf16"), ("proj dgrad D", 8192, 768, 768, "bf16"),
    ("text c_fc gelu", 2464, 3072, 768, "gelu"), ("text c_proj f32res", 2464, 768, 3072, "f32"), ("text qkv", 2464, 2304, 768, "bf16"),
    ("text out f32res", 2464, 768, 768, "f32"), ("text c_fc dgrad", 2464, 768, 3072, "bf16"), ("text c_proj dgrad", 2464, 3072, 768, "bf16"),
]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else ""
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    tot = 0.0
    for name, M, N, K, kind in SHAPES:
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
            c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            c2 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            f = lambda: ops.gemm_nt(a, w, c, M=M, N=N, K=K, bias=bias, c2=c2, epi=ops.EPI_GELU)
        elif kind == "swiglu":
            c = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev)
            c2 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            f = lambda: ops.gemm_nt(a, w, c, M=M, N=N, K=K, bias=bias, c2=c2, epi=ops.EPI_SWIGLU)
        elif kind == "rope":
            c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            hw = 256
            pos = (torch.arange(M, dtype=torch.int32, device=dev) % 257 - 1).clamp(min=-1)
            sin = torch.randn(hw, 64, device=dev, generator=g).to(torch.bfloat16)
            cos = torch.randn(hw, 64, device=dev, generator=g).to(torch.bfloat16)
            f = lambda: ops.gemm_qkv_rope(a, w, bias, c, M, N, K, pos, sin, cos, 2 * (N // 3))
        else:  # dsw: d_x12 = swiglu'(x12) (.) (dy W3): A = dy [M, K], W = w3^T [N = H, K], pre = x12 [M, 2H]
            pre = torch.randn(M, 2 * N, device=dev, generator=g).to(torch.bfloat16)
            c = torch.empty(M, 2 * N, dtype=torch.bfloat16, device=dev)
            f = lambda: ops.gemm_dgrad_swiglu(a, w, pre, c, M, N, K)
        t = timeit({"x": f})["x"]
        tot += t
        print(f"[{tag}] {name:22s} M={M:6d} N={N:5d} K={K:5d}  {t:7.1f} us  {2.0 * M * N * K / t / 1e6:7.1f} TF/s", flush=True)
    print(f"[{tag}] sum {tot:.1f} us")


if __name__ == "__main__":
    main()
