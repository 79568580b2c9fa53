"""Micro-benchmark of vtp_gemm_tn (weight-gradient GEMM from untransposed activations) tile configurations."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vtp_amd import _lib, ops

SHAPES = [("wgrad_qkv", 2304, 768, 8224), ("wgrad_w12", 4096, 768, 8224), ("wgrad_proj", 768, 768, 8224),
          ("wgrad_w3", 768, 2048, 8224)]
CFGS = {0: "128x128 4w s2", 5: "128x128 8w s2", 2: "256x128 8w s2", 3: "256x128 8w s3"}


def main():
    global SHAPES
    if len(sys.argv) > 1:  # token rows (the reduction length), e.g. 34144 for the row-concatenated list forward
        SHAPES = [(t, m, n, int(sys.argv[1])) for t, m, n, k in SHAPES]
    lib = _lib.load()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    for tag, M, N, K in SHAPES:
        a = torch.randn(K, M, device=dev, generator=g).to(torch.bfloat16)
        b = torch.randn(K, N, device=dev, generator=g).to(torch.bfloat16)
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        for want in sorted({max(1, min(round(t / tiles), K // 512, 16)) for t in (256, 384, 512, 768)}):
            S = ops.gemm_splits(K, want)
            for cfg, name in CFGS.items():
                lib.vtp_set_gemm_tuning(cfg, 3)
                c = torch.empty(S * M * N, device=dev)

                def run():
                    ops.gemm_tn(a, b, c, M=M, N=N, K=K, lda=M, ldb=N, ldc=N, ldc2=M * N // 4, epi=ops.EPI_F32_SLAB, splits=S)
                for _ in range(3):
                    run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    run()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / 20
                print(f"{tag:10s} M={M:5d} N={N:5d} K={K:5d} cfg={cfg} ({name}) splits={S}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s", flush=True)
    lib.vtp_set_gemm_tuning(-1, 3)


if __name__ == "__main__":
    main()
