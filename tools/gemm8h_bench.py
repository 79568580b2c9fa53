"""A/B of the half-size two-workgroups-per-CU GEMM (cfg 9, gemm8h.hip) against the current dispatch ("auto": cfg 8 or the ring
kernels) on the NT shapes of the VTP-B train step: every epilogue incl. the fused RoPE / SwiGLU-backward ones.  Interleaved rounds, median.
Usage (GPU box): python tools/gemm8h_bench.py [quick] > gpurun_out/gemm8h_bench.log"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vtp_amd import _lib, ops
from tools.gemm8p_bench import timeit

LAYER = [  # (tag, N, K, kind)
    ("qkv_rope", 2304, 768, "rope"),
    ("proj_f32", 768, 768, "f32"),
    ("w12_swiglu", 4096, 768, "swiglu"),
    ("w3_f32", 768, 2048, "f32"),
    ("dgrad_w3_swiglu", 2048, 768, "dsw"),
    ("dgrad_w12", 768, 4096, "bf16"),
    ("dgrad_qkv", 768, 2304, "bf16"),
    ("dgrad_proj", 768, 768, "bf16"),
]
TEXT = [("t_qkv", 2304, 768, "bf16"), ("t_proj", 768, 768, "f32"), ("t_fc", 3072, 768, "gelu"), ("t_cproj", 768, 3072, "f32"),
        ("t_dgrad_fc", 768, 3072, "bf16"), ("t_dgrad_cproj", 3072, 768, "bf16")]


ALT = int(os.environ.get("ALT_CFG", "9"))  # the tile configuration measured against the dispatch (9: gemm8h, 10: gemm4w)


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    lib = _lib.load()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    cases = [(M, L) for M in ([34144] if quick else [34144, 16448, 8192]) for L in LAYER] + [(2464, L) for L in TEXT]
    for M, (tag, N, K, kind) in cases:
        a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        b = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device=dev, generator=g)
        fns = {}
        for name, cfg in (("auto", -1), ("8p", 8), ("8h", ALT)):
            if kind in ("bf16", "gelu"):
                c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
                kw = dict(bias=bias, epi=ops.EPI_BF16 if kind == "bf16" else ops.EPI_GELU)
                if kind == "gelu":
                    kw["c2"] = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
                call = lambda c=c, kw=kw: ops.gemm_nt(a, b, c, M=M, N=N, K=K, **kw)
            elif kind == "f32":
                c = torch.zeros(M, N, device=dev)
                call = lambda c=c: ops.gemm_nt(a, b, c, M=M, N=N, K=K, epi=ops.EPI_F32, bias=bias, resid=c)
            elif kind == "swiglu":
                c = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev)
                c2 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
                call = lambda c=c, c2=c2: ops.gemm_nt(a, b, c, M=M, N=N, K=K, epi=ops.EPI_SWIGLU, bias=bias, c2=c2)
            elif kind == "rope":
                c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
                pos = (torch.arange(M, dtype=torch.int32, device=dev) % 257) - 1
                sin = torch.randn(256, 64, device=dev, generator=g).to(torch.bfloat16)
                cos = torch.randn(256, 64, device=dev, generator=g).to(torch.bfloat16)
                call = lambda c=c, pos=pos, sin=sin, cos=cos: ops.gemm_qkv_rope(a, b, bias, c, M, N, K, pos, sin, cos, 2 * N // 3)
            else:  # dsw: A = dy [M, K], B = W3^T [H = N, K], x12 [M, 2N]
                x12 = torch.randn(M, 2 * N, device=dev, generator=g).to(torch.bfloat16)
                c = torch.empty(M, 2 * N, dtype=torch.bfloat16, device=dev)
                call = lambda c=c, x12=x12: ops.gemm_dgrad_swiglu(a, b, x12, c, M, N, K)

            def run(cfg=cfg, call=call):
                lib.vtp_set_gemm_tuning(cfg, 3)
                call()
            fns[name] = run
        t = timeit(fns)
        fl = 2.0 * M * N * K
        print(f"{tag:16s} M={M:5d} N={N:5d} K={K:5d}: auto {t['auto']:7.1f} us {fl / t['auto'] / 1e6:7.1f} TF/s | cfg{ALT} {t['8h']:7.1f} us "
              f"{fl / t['8h'] / 1e6:7.1f} TF/s  x{t['auto'] / t['8h']:.2f} | 8p {t['8p']:7.1f} us {fl / t['8p'] / 1e6:7.1f} TF/s", flush=True)
    lib.vtp_set_gemm_tuning(-1, 3)


if __name__ == "__main__":
    main()
