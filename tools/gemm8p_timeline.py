"""Where a persistent 256x256 GEMM workgroup spends its time: per (workgroup, tile) s_memrealtime stamps written by the kernel
(vtp_gemm_debug) -> k-loop time, epilogue-issue time, spread of the epilogue start across workgroups (lock-step measure), and the
same launch with the persistent grid capped to 32 workgroups (an eighth of the chip: is the epilogue bound per CU or chip-wide?).
Usage (GPU box): python tools/gemm8p_timeline.py > gpurun_out/gemm8p_timeline.log"""
import os

os.environ.setdefault("VTP_DIAG", "1")  # the library accepts its diagnostics hooks only in a process that asked for them
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vtp_amd import _lib, ops

SHAPES = [  # tag, M, N, K, epilogue
    ("w12_fwd swiglu", 34144, 4096, 768, ops.EPI_SWIGLU),
    ("qkv_fwd bf16", 34144, 2304, 768, ops.EPI_BF16),
    ("dgrad_w12 bf16", 34144, 768, 4096, ops.EPI_BF16),
    ("w3_fwd f32res", 34144, 768, 2048, ops.EPI_F32),
    ("dgrad_w3 bf16", 34144, 2048, 768, ops.EPI_BF16),
]


def run(lib, tag, M, N, K, epi, grid, g, delay=0):
    dev = "cuda"
    a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g)
    if epi == ops.EPI_BF16:
        c, kw = torch.empty(M, N, dtype=torch.bfloat16, device=dev), dict(bias=bias)
    elif epi == ops.EPI_F32:
        c = torch.zeros(M, N, device=dev)
        kw = dict(bias=bias, resid=c)
    else:
        c = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev)
        kw = dict(bias=bias, c2=torch.empty(M, N, dtype=torch.bfloat16, device=dev))
    nwg = 256
    tbuf = torch.zeros(2 * nwg * 16 * 4, dtype=torch.int64, device=dev)  # [wg][16 tiles][4] stamps, then [wg][64]: k-tile starts of tile 1
    lib.vtp_set_gemm_tuning(8, 3)
    for _ in range(3):
        ops.gemm_nt(a, b, c, M=M, N=N, K=K, epi=epi, **kw)
    lib.vtp_gemm_debug(tbuf.data_ptr(), grid, delay)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.gemm_nt(a, b, c, M=M, N=N, K=K, epi=epi, **kw)
    e1.record()
    torch.cuda.synchronize()
    lib.vtp_gemm_debug(None, 0, 0)
    us = e0.elapsed_time(e1) * 1e3
    t = tbuf[:nwg * 64].view(nwg, 16, 4).cpu().double() / 100.0  # us (slot 3 holds cycles, read separately)
    G0 = grid if grid else 256
    kts = tbuf[G0 * 64:G0 * 64 + G0 * 64].view(G0, 64)[:, :16].cpu().double() / 100.0
    G = grid if grid else 256
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    t0 = t[:G, :, 0][t[:G, :, 0] > 0].min()
    rows = []
    for ti in range(16):
        ok = t[:G, ti, 0] > 0
        if ok.sum() == 0:
            break
        st, ke, ee = t[:G, ti, 0][ok] - t0, t[:G, ti, 1][ok] - t0, t[:G, ti, 2][ok] - t0
        nxt = None
        if ti + 1 < 16:
            ok2 = (t[:G, ti + 1, 0] > 0) & ok
            if ok2.sum() > 0:
                nxt = float((t[:G, ti + 1, 0][ok2] - t[:G, ti, 2][ok2]).mean())
        cyc = tbuf[:nwg * 64].view(nwg, 16, 4)[:G, ti, 3][ok].cpu().double()
        rows.append((ti, int(ok.sum()), float(st.mean()), float(st.std()), float((ke - st).mean()), float((ke - st).std()),
                     float((ee - ke).mean()), float((ee - ke).std()), float(ke.std()), float((cyc / (ke - st)).mean()) / 1e3))
    print(f"== {tag}{' [NO STORES]' if delay < 0 else ''}: M={M} N={N} K={K} grid={G} tiles={tiles} launch {us:.1f} us  ({2.0 * M * N * K / us / 1e6 * (1 if grid == 0 else 0):.0f} TF/s)")
    print("   tile  wgs   start(mean,sd)    kloop(mean,sd)   epi_issue(mean,sd)   sd(kloop end across wgs)   shader clock in the k loop (GHz)")
    for r in rows:
        print(f"   {r[0]:3d} {r[1]:5d}  {r[2]:8.1f} {r[3]:6.1f}   {r[4]:8.1f} {r[5]:6.1f}   {r[6]:8.1f} {r[7]:6.1f}    {r[8]:6.1f}      {r[9]:6.3f}")
    ok = kts[:, 0] > 0
    if int(ok.sum()) > 0:
        nkt = min(16, (K + 63) // 64)
        d = kts[ok][:, 1:nkt] - kts[ok][:, :nkt - 1]
        print("   tile 1, us per k-tile (mean over workgroups): " + " ".join(f"{float(v):.2f}" for v in d.mean(0)))
    sys.stdout.flush()


def delay_sweep(lib, tag, M, N, K, epi, g):
    """launch time with every second workgroup started late: if the epilogue store bursts of a lock-stepped chip are what costs,
    a half-tile offset between the two halves wins more than the idle start loses"""
    dev = "cuda"
    a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g)
    if epi == ops.EPI_BF16:
        c, kw = torch.empty(M, N, dtype=torch.bfloat16, device=dev), dict(bias=bias)
    elif epi == ops.EPI_F32:
        c = torch.zeros(M, N, device=dev)
        kw = dict(bias=bias, resid=c)
    else:
        c = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev)
        kw = dict(bias=bias, c2=torch.empty(M, N, dtype=torch.bfloat16, device=dev))
    lib.vtp_set_gemm_tuning(8, 3)
    delays = [0, 400, 800, 1200, 1700, 2400]
    res = {d: [] for d in delays}
    for _ in range(5):
        for d in delays:
            lib.vtp_gemm_debug(None, 0, d)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.gemm_nt(a, b, c, M=M, N=N, K=K, epi=epi, **kw)
            e1.record()
            torch.cuda.synchronize()
            res[d].append(e0.elapsed_time(e1) * 200.0)
    lib.vtp_gemm_debug(None, 0, 0)
    print(f"== delay sweep {tag} M={M} N={N} K={K}: " + "  ".join(f"{d / 100:.0f}us:{sorted(v)[2]:.1f}" for d, v in res.items()), flush=True)


def main():
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(0)
    for tag, M, N, K, epi in SHAPES:
        for grid in (0, 32):
            Mg = M if grid == 0 else (M // 8 // 256) * 256  # an eighth of the rows on an eighth of the chip
            run(lib, tag, Mg, N, K, epi, grid, g)
        # (round 3 also ran a NO-STORES build here -- delay = -1: -1 % -- that kernel mode was removed in round 4)
    lib.vtp_set_gemm_tuning(-1, 3)


if __name__ == "__main__":
    main()
