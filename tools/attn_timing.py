"""In-kernel phase stamps of the resident attention forward (s_memtime at: first data, barrier, end of the key loop, stores, odd-tile
share, merge) for one workgroup: VTP_ATTN_TIMING=1 AT_B=<images> python tools/attn_timing.py   (prints cycle counts per wave)."""
import os, sys
sys.path.insert(0, "/root/repo")
import torch
from vtp_amd import ops
B, N, h = int(os.environ.get("AT_B", "64")), 257, 12
D = 64 * h
torch.manual_seed(0)
qkv = torch.randn(B * N, 3 * D, device="cuda").to(torch.bfloat16)
o = torch.empty(B * N, D, dtype=torch.bfloat16, device="cuda")
lse = torch.empty(B * h * N, device="cuda")
for _ in range(3):
    ops.attn_fwd(qkv, qkv[:, D:], qkv[:, 2 * D:], o, lse, B, N, h, N * 3 * D, 3 * D, N * D, D, 0.125, False)
