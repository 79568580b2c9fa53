"""Timing of the qkv projection with apply_rope in its epilogue (vtp_gemm_qkv_rope) at the step's three row counts, tile configurations 8
(256 x 256) and 9 (half-size, two workgroups per CU), beside the plain bf16 epilogue of the same shape.  Interleaved rounds, median.
Usage (GPU box): VTP_HIP_LIB=... python tools/rope_epi_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vtp_amd import _lib, ops
from tools.gemm8p_bench import timeit


def main():
    lib = _lib.load()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    N, K = 2304, 768
    for M in (34144, 16448, 8192):
        a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        b = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device=dev, generator=g)
        pos = (torch.arange(M, dtype=torch.int32, device=dev) % 257) - 1
        sin = torch.randn(256, 64, device=dev, generator=g).to(torch.bfloat16)
        cos = torch.randn(256, 64, device=dev, generator=g).to(torch.bfloat16)
        outs, fns = {}, {}
        for name, cfg, rope in (("rope8", 8, True), ("rope9", 9, True), ("plain8", 8, False), ("plain9", 9, False)):
            c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            outs[name] = c
            if rope:
                call = lambda c=c: ops.gemm_qkv_rope(a, b, bias, c, M, N, K, pos, sin, cos, 2 * N // 3)
            else:
                call = lambda c=c: ops.gemm_nt(a, b, c, M=M, N=N, K=K, bias=bias, epi=ops.EPI_BF16)

            def run(cfg=cfg, call=call):
                lib.vtp_set_gemm_tuning(cfg, 3)
                call()
            fns[name] = run
        t = timeit(fns)
        torch.cuda.synchronize()
        same = torch.equal(outs["rope8"], outs["rope9"])
        print(os.path.basename(os.environ.get("VTP_HIP_LIB", "libvtp_hip.so")), M, {k: round(v, 1) for k, v in t.items()}, "8 == 9:", same,
              "sum", float(outs["rope8"].float().abs().sum()), flush=True)
    lib.vtp_set_gemm_tuning(-1, 3)


if __name__ == "__main__":
    main()
