"""Attention kernels (vtp_attn_fwd / vtp_attn_bwd) at the shapes of the VTP-B train step: median us and algorithmic TFLOP/s
(fwd 4 N^2 d per head, bwd 2.5x).  Usage (GPU box): python tools/attn_bench.py > gpurun_out/attn_bench.log"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vtp_amd import ops

SHAPES = [(32, 257, 12, "trunk lead / teacher half"), (64, 257, 12, "trunk global crops"), (256, 37, 12, "local crops"),
          (32, 256, 12, "decoder"), (8, 1025, 16, "L @ 512^2 (tiled kernels)")]


if os.environ.get("ATTN_SHAPES"):  # "B,N,h;B,N,h;..."
    SHAPES = [tuple(int(v) for v in t.split(",")) + ("custom",) for t in os.environ["ATTN_SHAPES"].split(";")]


def main():
    dev = "cuda"
    torch.manual_seed(0)
    for B, N, h, tag in SHAPES:
        D = 64 * h
        qkv = (torch.randn(B * N, 3 * D, device=dev) * 1.0).to(torch.bfloat16)
        o = torch.empty(B * N, D, dtype=torch.bfloat16, device=dev)
        d_o = torch.randn(B * N, D, device=dev).to(torch.bfloat16)
        lse = torch.empty(B * h * N, device=dev)
        delta = torch.empty(B * h * N, device=dev)
        dqkv = torch.empty_like(qkv)
        scale = 0.125

        def fwd():
            ops.attn_fwd(qkv, qkv[:, D:], qkv[:, 2 * D:], o, lse, B, N, h, N * 3 * D, 3 * D, N * D, D, scale, False)

        def bwd():
            ops.attn_bwd(qkv, qkv[:, D:], qkv[:, 2 * D:], o, d_o, lse, delta, dqkv, dqkv[:, D:], dqkv[:, 2 * D:], B, N, h, N * 3 * D,
                         3 * D, N * D, D, scale, False)

        res = {}
        for name, f in (("fwd", fwd), ("bwd", bwd)):
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            # 20 launches captured in one hipGraph: the replay is GPU-bound (python + ctypes launch overhead would otherwise hide
            # anything shorter than ~20 us)
            gr = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                with torch.cuda.graph(gr, stream=side):
                    for _ in range(20):
                        f()
            torch.cuda.synchronize()
            gr.replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                gr.replay()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 50.0)
            res[name] = sorted(ts)[3]
        fl = 4.0 * N * N * 64 * B * h
        print(f"B={B:4d} N={N:5d} h={h:3d} ({tag:28s}): fwd {res['fwd']:7.1f} us {fl / res['fwd'] / 1e6:7.1f} TF/s | bwd {res['bwd']:7.1f} us "
              f"{2.5 * fl / res['bwd'] / 1e6:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
