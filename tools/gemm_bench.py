"""Micro-benchmark of vtp_gemm_nt tile configurations on the GEMM shapes of the VTP-B train step (M = 32 img x 257).
Usage (GPU box): python tools/gemm_bench.py > gpurun_out/gemm_bench.log"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vtp_amd import _lib, ops

SHAPES = [  # (tag, M, N, K, epilogue)
    ("qkv_fwd", 8224, 2304, 768, ops.EPI_BF16),
    ("proj_fwd", 8224, 768, 768, ops.EPI_F32),
    ("w12_fwd", 8224, 4096, 768, ops.EPI_SWIGLU),
    ("w3_fwd", 8224, 768, 2048, ops.EPI_F32),
    ("dgrad_w3", 8224, 2048, 768, ops.EPI_BF16),
    ("dgrad_w12", 8224, 768, 4096, ops.EPI_BF16),
    ("dgrad_qkv", 8224, 768, 2304, ops.EPI_BF16),
    ("wgrad_qkv", 2304, 768, 8224, ops.EPI_F32_SLAB),
    ("wgrad_w12", 4096, 768, 8224, ops.EPI_F32_SLAB),
    ("wgrad_proj", 768, 768, 8224, ops.EPI_F32_SLAB),
    ("big_4096", 4096, 4096, 4096, ops.EPI_BF16),
]
CFGS = {0: "128x128 4w s2", 5: "128x128 8w s2", 21: "128x128 8w s2 PIPE", 2: "256x128 8w s2", 4: "256x256 8w s2"}


def main():
    global SHAPES
    if len(sys.argv) > 1:  # python tools/gemm_bench.py <M>: the forward / dgrad shapes at another row count
        M2 = int(sys.argv[1])
        SHAPES = [(t, M2, n, k, e) for t, m, n, k, e in SHAPES if m == 8224]
    lib = _lib.load()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    for tag, M, N, K, epi in SHAPES:
        a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        b = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device=dev, generator=g)
        ref = None
        for cfg, name in CFGS.items():
            for swz in (3, 1):  # 3 = XCD tile order + LDS-staged stores; 1 = direct stores
                lib.vtp_set_gemm_tuning(cfg, swz)
                kw = {}
                splits = 1
                if epi == ops.EPI_BF16:
                    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
                    kw = dict(bias=bias)
                elif epi == ops.EPI_F32:
                    c = torch.zeros(M, N, device=dev)
                    kw = dict(bias=bias, resid=c)
                elif epi == ops.EPI_SWIGLU:
                    c = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev)
                    kw = dict(bias=bias, c2=torch.empty(M, N, dtype=torch.bfloat16, device=dev))
                else:
                    tiles = ((M + 127) // 128) * ((N + 127) // 128)
                    splits = ops.gemm_splits(K, max(1, min(round(384 / tiles), K // 512, 16)))
                    c = torch.empty(splits * M * N, device=dev)
                    kw = dict(ldc=N, ldc2=M * N // 4, splits=splits)
                def run():
                    ops.gemm_nt(a, b, c, M=M, N=N, K=K, epi=epi, **kw)
                for _ in range(3):
                    run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                iters = 20
                e0.record()
                for _ in range(iters):
                    run()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / iters
                tf = 2.0 * M * N * K / us / 1e6
                chk = ""
                if epi == ops.EPI_BF16:
                    if ref is None:
                        ref = c.float().clone()
                    else:
                        chk = f" maxdiff_vs_cfg0={float((c.float() - ref).abs().max()):.2e}"
                print(f"{tag:10s} M={M:5d} N={N:5d} K={K:5d} cfg={cfg} ({name}) swz={swz} splits={splits}: {us:8.1f} us  {tf:7.1f} TF/s{chk}", flush=True)
    lib.vtp_set_gemm_tuning(-1, 3)


if __name__ == "__main__":
    main()
