"""A/B of the 8-phase 256x256 GEMM main loop (cfg 8, gemm8p.hip) against the ring kernels' best configuration on the GEMM
shapes of the VTP-B train step at the row counts of the row-concatenated passes (M = 34144 student list forward, 16448 teacher,
8224 / 8192 single passes) and on the weight-gradient (TN) shapes.  Interleaved rounds, median.
Usage (GPU box): python tools/gemm8p_bench.py [quick] > gpurun_out/gemm8p_bench.log"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vtp_amd import _lib, ops

LAYER = [  # (tag, N, K, epilogue)
    ("qkv_fwd", 2304, 768, ops.EPI_BF16),
    ("proj_fwd", 768, 768, ops.EPI_F32),
    ("w12_fwd", 4096, 768, ops.EPI_SWIGLU),
    ("w3_fwd", 768, 2048, ops.EPI_F32),
    ("dgrad_w3", 2048, 768, ops.EPI_BF16),
    ("dgrad_w12", 768, 4096, ops.EPI_BF16),
    ("dgrad_qkv", 768, 2304, ops.EPI_BF16),
    ("dgrad_proj", 768, 768, ops.EPI_BF16),
]
WGRAD = [("wgrad_qkv", 2304, 768), ("wgrad_proj", 768, 768), ("wgrad_w12", 4096, 768), ("wgrad_w3", 768, 2048)]


def timeit(fns, rounds=7, iters=10):
    """fns: dict name -> callable.  Interleaved rounds; returns name -> median us."""
    res = {k: [] for k in fns}
    for k, f in fns.items():
        for _ in range(2):
            f()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) * 1e3 / iters)
    return {k: sorted(v)[len(v) // 2] for k, v in res.items()}


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] in ("quick", "var")
    var = len(sys.argv) > 1 and sys.argv[1] == "var"   # bf16-output NT + slab TN only
    lib = _lib.load()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    Ms = [34144] if quick else [34144, 16448, 8224]
    for M in Ms:
        for tag, N, K, epi in LAYER:
            if var and epi != ops.EPI_BF16:
                continue
            a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
            b = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
            bias = torch.randn(N, device=dev, generator=g)
            outs = {}

            def mk(cfg):
                if epi == ops.EPI_BF16:
                    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
                    kw = dict(bias=bias)
                elif epi == ops.EPI_F32:
                    c = torch.zeros(M, N, device=dev)
                    kw = dict(bias=bias, resid=c)
                else:
                    c = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev)
                    kw = dict(bias=bias, c2=torch.empty(M, N, dtype=torch.bfloat16, device=dev))
                outs[cfg] = (c, kw)

                def run():
                    lib.vtp_set_gemm_tuning(cfg, 3)
                    ops.gemm_nt(a, b, c, M=M, N=N, K=K, epi=epi, **kw)
                return run

            fns = {"auto": mk(-1), "8p": mk(8)}
            t = timeit(fns)
            # correctness of cfg 8 against the ring kernel (fresh outputs: the fp32-residual case accumulates in place)
            chk = ""
            if epi != ops.EPI_F32:
                d = float((outs[8][0].float() - outs[-1][0].float()).abs().max())
                chk = f" maxdiff(8p vs auto)={d:.2e}"
                if epi == ops.EPI_SWIGLU:
                    d2 = float((outs[8][1]["c2"].float() - outs[-1][1]["c2"].float()).abs().max())
                    chk += f" x12 diff={d2:.2e}"
            else:
                c0, c8 = torch.zeros(M, N, device=dev), torch.zeros(M, N, device=dev)
                lib.vtp_set_gemm_tuning(-1, 3)
                ops.gemm_nt(a, b, c0, M=M, N=N, K=K, epi=epi, bias=bias, resid=c0)
                lib.vtp_set_gemm_tuning(8, 3)
                ops.gemm_nt(a, b, c8, M=M, N=N, K=K, epi=epi, bias=bias, resid=c8)
                chk = f" maxdiff(8p vs auto)={float((c0 - c8).abs().max()):.2e}"
            fl = 2.0 * M * N * K
            print(f"{tag:10s} M={M:5d} N={N:5d} K={K:5d}: auto {t['auto']:7.1f} us {fl / t['auto'] / 1e6:7.1f} TF/s | 8p {t['8p']:7.1f} us "
                  f"{fl / t['8p'] / 1e6:7.1f} TF/s  x{t['auto'] / t['8p']:.2f}{chk}", flush=True)
    # weight gradients: C[Mo, No] = dy[K, Mo]^T x[K, No], K = tokens
    for Kt in ([34144] if quick else [34144, 8192]):
        for tag, Mo, No in WGRAD:
            A = torch.randn(Kt, Mo, device=dev, generator=g).to(torch.bfloat16)
            Bm = torch.randn(Kt, No, device=dev, generator=g).to(torch.bfloat16)
            tiles128 = ((Mo + 127) // 128) * ((No + 127) // 128)
            s_old = ops.gemm_splits(Kt, int(max(1, min(512 // tiles128, Kt // 512, 16))))
            tiles256 = ((Mo + 255) // 256) * ((No + 255) // 256)
            variants = {"auto": (-1, s_old)}
            for want in (256, 512):
                s8 = ops.gemm_splits(Kt, int(max(1, min(want // tiles256, Kt // 512))))
                variants[f"8p/{s8}"] = (8, s8)
            slabs, fns = {}, {}
            for name, (cfg, S) in variants.items():
                slab = torch.empty(S * Mo * No, device=dev)
                slabs[name] = (slab, S)

                def run(cfg=cfg, S=S, slab=slab):
                    lib.vtp_set_gemm_tuning(cfg, 3)
                    ops.gemm_tn(A, Bm, slab, M=Mo, N=No, K=Kt, lda=Mo, ldb=No, ldc=No, ldc2=Mo * No // 4, epi=ops.EPI_F32_SLAB, splits=S)
                fns[name] = run
            t = timeit(fns)
            ref = slabs["auto"][0].view(slabs["auto"][1], -1).sum(0)
            fl = 2.0 * Mo * No * Kt
            line = f"{tag:10s} Mo={Mo:5d} No={No:5d} K={Kt:5d}:"
            for name in variants:
                got = slabs[name][0].view(slabs[name][1], -1).sum(0)
                rel = float((got - ref).norm() / ref.norm())
                line += f" {name} {t[name]:7.1f} us {fl / t[name] / 1e6:7.1f} TF/s (rel {rel:.1e}) |"
            print(line, flush=True)
    lib.vtp_set_gemm_tuning(-1, 3)


if __name__ == "__main__":
    main()
