"""Where the resident attention backward kernels spend their time: per (workgroup, wave) s_memrealtime stamps (vtp_attn_debug)
-> staging, loop and store time per wave, workgroup lifetime, and how many workgroups are alive over the launch.
Usage (GPU box): python tools/attn_bwd_timeline.py [B N heads] > gpurun_out/attn_bwd_timeline.log"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vtp_amd import _lib, ops

PAD = int(os.environ.get("ATTN_LDS_PAD", "0"))  # extra LDS bytes per workgroup (occupancy experiments)
WPB = int(os.environ.get("ATTN_WPB", "0"))      # waves per workgroup override (0 = heuristic)
STAG = int(os.environ.get("ATTN_STAGGER", "0"))  # 10-ns ticks: every second first-round workgroup starts late
SHAPES = [(64, 257, 12), (32, 257, 12), (256, 37, 12), (32, 256, 12)]


def run(lib, B, N, h):
    dev = "cuda"
    D = 64 * h
    torch.manual_seed(0)
    qkv = torch.randn(B * N, 3 * D, device=dev).to(torch.bfloat16)
    o = torch.empty(B * N, D, dtype=torch.bfloat16, device=dev)
    d_o = torch.randn(B * N, D, device=dev).to(torch.bfloat16)
    lse = torch.empty(B * h * N, device=dev)
    delta = torch.empty(B * h * N, device=dev)
    dqkv = torch.empty_like(qkv)
    ops.attn_fwd(qkv, qkv[:, D:], qkv[:, 2 * D:], o, lse, B, N, h, N * 3 * D, 3 * D, N * D, D, 0.125, False)

    def bwd():
        ops.attn_bwd(qkv, qkv[:, D:], qkv[:, 2 * D:], o, d_o, lse, delta, dqkv, dqkv[:, D:], dqkv[:, 2 * D:], B, N, h, N * 3 * D, 3 * D,
                     N * D, D, 0.125, False)

    lib.vtp_attn_debug(None, PAD, WPB, STAG)
    for _ in range(3):
        bwd()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            bwd()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 100.0)
    print(f"   unstamped: {sorted(ts)[2]:.1f} us per backward (median of 5 x 10 launches)")
    nwg_max = B * h * 9
    nblk = (N + 31) // 32
    wpb = (nblk + 1) // 2 if (B * h <= 512 and nblk >= 4) else nblk  # res_waves_per_block (attention_resident.hip)
    if WPB > 0:
        wpb = min(WPB, nblk)
    nwg = B * h * ((nblk + wpb - 1) // wpb)
    if WPB == 0 and (nblk in (8, 2) or (nblk == 9 and N - 256 <= 2) or (nblk == 3 and N - 64 <= 2)):  # fused kernel: one workgroup per head
        nwg = B * h
    tbuf = torch.zeros(2 * nwg_max * 64, dtype=torch.int64, device=dev)
    lib.vtp_attn_debug(tbuf.data_ptr(), PAD, WPB, STAG)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    bwd()
    e1.record()
    torch.cuda.synchronize()
    lib.vtp_attn_debug(None, 0, 0, 0)
    print(f"== B={B} N={N} heads={h}: backward {e0.elapsed_time(e1) * 1e3:.1f} us (both kernels, stamped run)")
    t = tbuf.cpu().double() / 100.0  # us
    fused = WPB == 0 and (nblk in (8, 2) or (nblk == 9 and N - 256 <= 2) or (nblk == 3 and N - 64 <= 2))
    for ki, name in enumerate(("fused",) if fused else ("dQ", "dK/dV")):
        seg = t[ki * nwg * 64:(ki + 1) * nwg * 64].view(nwg, 16, 4)
        act = seg[:, :, 1] > 0
        t0 = seg[:, :, 0][seg[:, :, 0] > 0].min()
        done = seg[:, :, 3]
        w_loop = act & (seg[:, :, 2] > 0)
        stage = (seg[:, :, 1] - seg[:, :, 0])[act]
        loop = (seg[:, :, 2] - seg[:, :, 1])[w_loop]
        store = (seg[:, :, 3] - seg[:, :, 2])[w_loop]
        life = done.max(1).values - seg[:, 0, 0]
        end = done.max() - t0
        nw = int(act[0].sum())
        a, bb = ("phase 1 (dQ)", "phase 2 (dK/dV) + stores") if fused else ("loop", "stores")
        print(f"   {name:6s}: {nwg} workgroups x {nw} waves, kernel span {end:.1f} us | per wave: staging {stage.mean():.2f} us, {a} {loop.mean():.2f} "
              f"(min {loop.min():.2f} max {loop.max():.2f}), {bb} {store.mean():.2f} | workgroup lifetime {life.mean():.2f} us")
        starts, ends = seg[:, 0, 0] - t0, done.max(1).values - t0
        grid = torch.linspace(0, float(end), 9)[1:-1]
        alive = [int(((starts <= g) & (ends > g)).sum()) for g in grid]
        print(f"           workgroups alive at {', '.join(f'{float(g):.0f}' for g in grid)} us: {alive}")
    sys.stdout.flush()


def main():
    lib = _lib.load()
    shapes = SHAPES
    if len(sys.argv) == 4:
        shapes = [tuple(int(v) for v in sys.argv[1:4])]
    for B, N, h in shapes:
        run(lib, B, N, h)


if __name__ == "__main__":
    main()
