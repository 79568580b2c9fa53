"""One grouped weight-gradient launch (a VTP-B block at 34 144 token rows), five times -- the target of a rocprofv3 counter pass.
usage: python tools/one_wgrad_group.py <kernel: 0 = 8-phase | 1 = one-wave-per-SIMD> [Ktok]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vtp_amd import ops

kernel = int(sys.argv[1])
Ktok = int(sys.argv[2]) if len(sys.argv) > 2 else 34144
D, H = 768, 2048
g = torch.Generator(device="cuda").manual_seed(0)
bf = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
dqkv, dmid, dpre, dy = bf(Ktok, 3 * D), bf(Ktok, D), bf(Ktok, 2 * H), bf(Ktok, D)
xn1, att, xn2, hid = bf(Ktok, D), bf(Ktok, D), bf(Ktok, D), bf(Ktok, H)
probs = [(dy, hid, D, H, 0, False), (dpre, xn2, 2 * H, D, H, True), (dmid, att, D, D, 0, False), (dqkv, xn1, 3 * D, D, 0, True)]
grp = ops.WgradGroup(Ktok)
for a, x, N, K, sh, cs in probs:
    grp.add(a, x, torch.zeros(N * K, device="cuda"), torch.zeros(N, device="cuda") if cs else None, N, K, sh)
grp.finalize("cuda", {})
for _ in range(5):
    grp.launch(kernel=kernel)
torch.cuda.synchronize()
