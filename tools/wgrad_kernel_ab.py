"""A/B of the two kernels of the grouped weight-gradient launch (ops.WgradGroup.launch(kernel=0 | 1): the 8-phase kernel against the
one-wave-per-SIMD kernel of gemm4w_tn.hip) on a VTP-B / VTP-L block at the token counts of the step.  Interleaved rounds, median.
Usage (GPU box): python tools/wgrad_kernel_ab.py > gpurun_out/r04_wgrad_kernel_ab.log"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vtp_amd import ops
from tools.gemm8p_bench import timeit


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    for D, H, toks in ((768, 2048, (34144, 16448, 8192, 4096, 2464)), (1024, 2736, (34144, 8192))):
        for Ktok in toks:
            bf = lambda *s: torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)
            Hp = (H + 7) // 8 * 8
            dqkv, dmid, dpre, dy = bf(Ktok, 3 * D), bf(Ktok, D), bf(Ktok, 2 * Hp), bf(Ktok, D)
            xn1, att, xn2, hid = bf(Ktok, D), bf(Ktok, D), bf(Ktok, D), bf(Ktok, Hp)
            probs = [(dy, hid, D, Hp, 0, False), (dpre, xn2, 2 * Hp, D, Hp, True), (dmid, att, D, D, 0, False), (dqkv, xn1, 3 * D, D, 0, True)]
            gws = [torch.zeros(N * K, device=dev) for _, _, N, K, _, _ in probs]
            gbs = [torch.zeros(N, device=dev) if cs else None for _, _, N, _, _, cs in probs]
            grp = ops.WgradGroup(Ktok)
            for (a, x, N, K, sh, _), gw, gb in zip(probs, gws, gbs):
                grp.add(a, x, gw, gb, N, K, sh)
            grp.finalize(dev, {})
            t = timeit({"k8": lambda: grp.launch(kernel=0), "k4w": lambda: grp.launch(kernel=1)})
            fl = sum(2.0 * N * K * Ktok for _, _, N, K, _, _ in probs)
            more = ""
            for sp in [int(x) for x in os.environ.get("MORE_SPLITS", "").split()]:  # the one-wave kernel with other slice counts
                g2 = ops.WgradGroup(Ktok)
                for (a, x, N, K, sh, _), gw, gb in zip(probs, gws, gbs):
                    g2.add(a, x, gw, gb, N, K, sh)
                g2.finalize(dev, {})
                g2.splits = sp
                g2.part = torch.empty(g2.ntiles * sp * 65536, device=dev)
                g2.ticket = torch.zeros(max(g2.ntiles, 256), dtype=torch.int32, device=dev)
                tt = timeit({"x": lambda: g2.launch(kernel=1)})["x"]
                more += f" | one-wave x{sp} slices {tt:7.1f} us"
            print(f"D={D} Ktok={Ktok:5d} ({grp.ntiles:3d} tiles x {grp.splits} slices): 8-phase {t['k8']:7.1f} us {fl / t['k8'] / 1e6:7.1f} TF/s | "
                  f"one-wave {t['k4w']:7.1f} us {fl / t['k4w'] / 1e6:7.1f} TF/s  x{t['k8'] / t['k4w']:.2f}  [auto: kernel {grp.kernel}]" + more, flush=True)


if __name__ == "__main__":
    main()
