"""Where the workgroups of ONE grouped weight-gradient launch (one-wave-per-SIMD kernel, gemm4w_tn.hip) spend their time: 100-MHz stamps
per workgroup (vtp_gemm_debug) -> prologue / k loop / publish / combine + epilogue, and the SPREAD of the k-loop end across the
workgroups of an XCD (do the tiles that share an operand panel through L2 stay in step?).
Usage (GPU box): python tools/wgrad_timeline.py [Ktok ...]"""
import os

os.environ.setdefault("VTP_DIAG", "1")
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vtp_amd import _lib, ops


def main():
    lib = _lib.load()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    D, H = 768, 2048
    for Ktok in [int(a) for a in sys.argv[1:]] or [34144, 8192]:
        bf = lambda *s: torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)
        dqkv, dmid, dpre, dy = bf(Ktok, 3 * D), bf(Ktok, D), bf(Ktok, 2 * H), bf(Ktok, D)
        xn1, att, xn2, hid = bf(Ktok, D), bf(Ktok, D), bf(Ktok, D), bf(Ktok, H)
        probs = [(dy, hid, D, H, 0, False), (dpre, xn2, 2 * H, D, H, True), (dmid, att, D, D, 0, False), (dqkv, xn1, 3 * D, D, 0, True)]
        grp = ops.WgradGroup(Ktok)
        for a, x, N, K, sh, cs in probs:
            grp.add(a, x, torch.zeros(N * K, device=dev), torch.zeros(N, device=dev) if cs else None, N, K, sh,
                    accumulate=os.environ.get("ACC", "1") != "0")
        grp.finalize(dev, {})
        for _ in range(3):
            grp.launch(kernel=1)
        W = grp.nitems if grp.items is not None else grp.ntiles * grp.splits
        tbuf = torch.zeros((W + 256) * 64, dtype=torch.int64, device=dev)
        _lib.check(lib.vtp_gemm_debug(tbuf.data_ptr(), 0, 0), "vtp_gemm_debug")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        grp.launch(kernel=1)
        e1.record()
        torch.cuda.synchronize()
        lib.vtp_gemm_debug(None, 0, 0)
        t = tbuf[:W * 8].view(W, 8).cpu().double()
        t0 = t[:, 0].min()
        us = lambda c: (t[:, c] - t0) / 100.0
        start, landed, kend, pub, end = us(0), us(1), us(2), us(3), us(4)
        last = t[:, 4] > 0
        print(f"Ktok {Ktok}: {W} workgroups, launch {e0.elapsed_time(e1) * 1e3:.1f} us (events); start spread {start.max():.1f} us; first operands "
              f"{(landed - start).mean():.1f} us; k loop mean {(kend - landed).mean():.1f} (min {(kend - landed).min():.1f} max {(kend - landed).max():.1f}); "
              f"publish + ticket {(pub - kend)[t[:, 3] > 0].mean():.1f}; combine + epilogue of the last arrivers {(end - pub)[last & (t[:, 3] > 0)].mean():.1f} us; "
              f"last workgroup done at {max(end[last].max(), pub.max()):.1f} us")
        xcc = t[:, 7].long()
        for x in range(8):
            m = xcc == x
            if m.sum() == 0:
                continue
            ke = kend[m]
            tiles = sorted(set(t[m, 5].long().tolist()))
            print(f"   XCC {x}: {int(m.sum())} workgroups, tiles {tiles[0]}..{tiles[-1]}, slices {sorted(set(t[m, 6].long().tolist()))}, k-loop end "
                  f"{ke.min():.1f} .. {ke.max():.1f} us (spread {ke.max() - ke.min():.1f}), k-loop time {(kend - landed)[m].min():.1f} .. {(kend - landed)[m].max():.1f}")


if __name__ == "__main__":
    main()
