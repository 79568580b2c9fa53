"""EXPERIMENT (round 5): the split-accumulator K = 768 GEMM (tile configuration 11, vtp_amd/csrc/gemm4s.hip) against the 8-phase kernel
(cfg 8), the half-size kernel (cfg 9), the dispatch and hipBLASLt (torch.matmul) on the plain-bf16 K = 768 shapes of the step:
correctness (bit-identical to cfg 8) first, then interleaved timing.  Run under `timeout` (a new kernel with barriers).
Usage (GPU box): timeout 300 python tools/gemm4s_probe.py > gpurun_out/gemm4s_probe.log"""
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
    K = 768
    shapes = [(512, 256), (300, 128), (34144, 2304), (34144, 768), (34144, 4096), (16448, 2304), (8192, 2304), (8192, 768)]
    for M, N in shapes:
        a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        b = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
        outs = {}
        for name, cfg in (("8p", 8), ("4s", 11)):
            c = torch.full((M + 8, N), 7.0, dtype=torch.bfloat16, device=dev)  # guard rows: nothing may be written past row M
            lib.vtp_set_gemm_tuning(cfg, 3)
            ops.gemm_nt(a, b, c, M=M, N=N, K=K, epi=ops.EPI_BF16)
            torch.cuda.synchronize()
            outs[name] = c
        lib.vtp_set_gemm_tuning(-1, 3)
        ref = (a.float() @ b.float().T)
        same = torch.equal(outs["4s"][:M], outs["8p"][:M])
        guard = bool((outs["4s"][M:] == 7.0).all())
        err = float((outs["4s"][:M].float() - ref).abs().max() / ref.abs().max())
        print(f"M={M:6d} N={N:5d}: bit-identical to cfg 8: {same}  guard rows untouched: {guard}  max rel err vs fp32 {err:.2e}", flush=True)
        if not same:
            d = (outs["4s"][:M] != outs["8p"][:M])
            idx = d.nonzero()
            print("   differing elements:", int(d.sum()), "first:", idx[:4].tolist())
            continue
        if M < 4096:
            continue
        c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        fns = {}
        for name, cfg in (("auto", -1), ("8p", 8), ("8h", 9), ("4s", 11)):
            def call(cfg=cfg):
                lib.vtp_set_gemm_tuning(cfg, 3)
                ops.gemm_nt(a, b, c, M=M, N=N, K=K, epi=ops.EPI_BF16)
            fns[name] = call
        bt = b.T.contiguous()
        fns["hipBLASLt"] = lambda: torch.matmul(a, bt)
        t = timeit(fns, rounds=5, iters=10)
        lib.vtp_set_gemm_tuning(-1, 3)
        fl = 2.0 * M * N * K
        print("   " + "  ".join(f"{k} {v:7.1f} us ({fl / v / 1e6:6.1f} TF/s)" for k, v in t.items()), flush=True)


if __name__ == "__main__":
    main()
