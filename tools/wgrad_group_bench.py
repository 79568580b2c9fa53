"""The four weight gradients of a VTP-B block: ONE grouped launch (ops.WgradGroup: in-launch split-K combine + fused bias-gradient
sums) against the per-layer path it replaces (split-K TN GEMM into slabs + reduce_slabs + colsum_bf16 per linear layer), alone on the
chip, at the token counts of the step (34 144 trunk, 8 192 decoder, 2 464 text); plus the in-kernel stamps of the grouped launch
(k loop / combine + epilogue per workgroup).  Usage (GPU box): python tools/wgrad_group_bench.py > gpurun_out/wgrad_group.log"""
import os

os.environ.setdefault("VTP_DIAG", "1")  # the library accepts its diagnostics hooks only in a process that asked for them
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vtp_amd import _lib, ops


def med(f, rounds=7, iters=5):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / iters)
    return sorted(ts)[len(ts) // 2]


def main():
    lib = _lib.load()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    D, H = 768, 2048
    for Ktok in (34144, 8192, 2464):
        bf = lambda *s: torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)
        dqkv, dmid, dpre, dy = bf(Ktok, 3 * D), bf(Ktok, D), bf(Ktok, 2 * H), bf(Ktok, D)
        xn1, att, xn2, hid = bf(Ktok, D), bf(Ktok, D), bf(Ktok, D), bf(Ktok, H)
        probs = [(dy, hid, D, H, 0, False), (dpre, xn2, 2 * H, D, H, True), (dmid, att, D, D, 0, False), (dqkv, xn1, 3 * D, D, 0, True)]
        gws = [torch.zeros(N * K, device=dev) for _, _, N, K, _, _ in probs]
        gbs = [torch.zeros(N, device=dev) if cs else None for _, _, N, _, _, cs in probs]
        grp = ops.WgradGroup(Ktok)
        for (a, x, N, K, sh, _), gw, gb in zip(probs, gws, gbs):
            grp.add(a, x, gw, gb, N, K, sh)
        grp.finalize(dev, {})
        grp_nocs = ops.WgradGroup(Ktok)
        for (a, x, N, K, sh, _), gw in zip(probs, gws):
            grp_nocs.add(a, x, gw, None, N, K, sh)
        grp_nocs.finalize(dev, {})
        slab = torch.empty(64 * 4096 * 768, device=dev)

        def old():
            for (a, x, N, K, sh, cs), gw, gb in zip(probs, gws, gbs):
                if cs:
                    ops.colsum_bf16(a, a.stride(0), gb, Ktok, N, swiglu_h=sh)
                St = ops.gemm_tn_splits(N, K, Ktok)
                kw = dict(M=N, N=K, K=Ktok, lda=a.stride(0), ldb=x.stride(0), ldc=K, c_remap=(-1, sh) if sh else (0, 0))
                if St == 1:
                    ops.gemm_tn(a, x, gw, resid=gw, epi=ops.EPI_F32, **kw)
                else:
                    ops.gemm_tn(a, x, slab, ldc2=N * K // 4, epi=ops.EPI_F32_SLAB, splits=St, **kw)
                    ops.reduce_slabs(slab, N * K, St, gw, N * K, accumulate=True)

        fl = sum(2.0 * N * K * Ktok for _, _, N, K, _, _ in probs)
        t_old, t_new, t_nocs = med(old), med(grp.launch), med(grp_nocs.launch)
        print(f"Ktok={Ktok}: per-layer path {t_old:8.1f} us {fl / t_old / 1e6:7.1f} TF/s | grouped ({grp.ntiles} tiles x {grp.splits}) "
              f"{t_new:8.1f} us {fl / t_new / 1e6:7.1f} TF/s | grouped without the bias sums {t_nocs:8.1f} us", flush=True)
        nwg = grp.ntiles * grp.splits
        tbuf = torch.zeros(max(nwg, 256) * 16 * 4, dtype=torch.int64, device=dev)
        lib.vtp_gemm_debug(tbuf.data_ptr(), 0, 0)
        grp.launch()
        torch.cuda.synchronize()
        lib.vtp_gemm_debug(None, 0, 0)
        t = tbuf.view(-1, 16, 4)[:nwg, 0].cpu().double() / 100.0
        t0 = t[:, 0].min()
        kl, ep = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1]
        print(f"   stamps: start spread {float(t[:, 0].max() - t0):.1f} us | k loop mean {float(kl.mean()):.1f} min {float(kl.min()):.1f} max "
              f"{float(kl.max()):.1f} | publish / combine / epilogue mean {float(ep.mean()):.1f} max {float(ep.max()):.1f} | last end "
              f"{float(t[:, 2].max() - t0):.1f} us", flush=True)


if __name__ == "__main__":
    main()
