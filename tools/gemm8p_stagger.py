"""Launch time of the 256x256 persistent GEMM with every second workgroup started late (vtp_gemm_debug delay): the shapes whose
epilogues move the most HBM bytes per tile (fused SwiGLU backward, SwiGLU forward, fp32 residual) run k loop and epilogue in lock
step across the chip -- MFMA idle while every CU stores, HBM idle while every CU multiplies.  An offset between two halves of the
CUs interleaves the two phases.
Usage (GPU box): python tools/gemm8p_stagger.py > gpurun_out/gemm8p_stagger.log"""
import os

os.environ.setdefault("VTP_DIAG", "1")  # the library accepts its diagnostics hooks only in a process that asked for them
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vtp_amd import _lib, ops

SHAPES = [  # tag, M, N, K, kind
    ("dgrad_swiglu", 34144, 2048, 768, "dsw"),
    ("dgrad_swiglu", 8192, 2048, 768, "dsw"),
    ("w12_fwd swiglu", 34144, 4096, 768, ops.EPI_SWIGLU),
    ("w12_fwd swiglu", 16448, 4096, 768, ops.EPI_SWIGLU),
    ("w3_fwd f32res", 34144, 768, 2048, ops.EPI_F32),
    ("proj_fwd f32res", 34144, 768, 768, ops.EPI_F32),
    ("proj_fwd f32res", 16448, 768, 768, ops.EPI_F32),
    ("qkv_fwd bf16", 34144, 2304, 768, ops.EPI_BF16),
]


def make(M, N, K, kind, g):
    dev = "cuda"
    a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g)
    if kind == "dsw":
        x12 = torch.randn(M, 2 * N, device=dev, generator=g).to(torch.bfloat16)
        dx12 = torch.empty(M, 2 * N, dtype=torch.bfloat16, device=dev)
        return lambda: ops.gemm_dgrad_swiglu(a, b, x12, dx12, M, N, K)
    if kind == ops.EPI_BF16:
        c, kw = torch.empty(M, N, dtype=torch.bfloat16, device=dev), dict(bias=bias)
    elif kind == ops.EPI_F32:
        c = torch.zeros(M, N, device=dev)
        kw = dict(bias=bias, resid=c)
    else:
        c = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev)
        kw = dict(bias=bias, c2=torch.empty(M, N, dtype=torch.bfloat16, device=dev))
    return lambda: ops.gemm_nt(a, b, c, M=M, N=N, K=K, epi=kind, **kw)


def main():
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(0)
    delays = [0, 300, 600, 900, 1200, 1600, 2000, 2600]
    for tag, M, N, K, kind in SHAPES:
        f = make(M, N, K, kind, g)
        lib.vtp_set_gemm_tuning(8, 3)
        for _ in range(3):
            f()
        res = {d: [] for d in delays}
        for _ in range(5):
            for d in delays:
                lib.vtp_gemm_debug(None, 0, d)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    f()
                e1.record()
                torch.cuda.synchronize()
                res[d].append(e0.elapsed_time(e1) * 200.0)
        lib.vtp_gemm_debug(None, 0, 0)
        lib.vtp_set_gemm_tuning(-1, 3)
        print(f"== {tag} M={M} N={N} K={K}: " + "  ".join(f"{d / 100:.0f}us:{sorted(v)[2]:.1f}" for d, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
