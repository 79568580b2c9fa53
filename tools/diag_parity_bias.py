"""Diagnostic for tests/test_parity_large_gpu.py: where does the DINO-head bias-gradient error sit?  For every seed: the error vectors
of ours / CPU-autocast / CUDA-autocast against the fp32 oracle for the head's bias gradients -- norms, ratios, cosines between the
error vectors (correlated = structural, uncorrelated = rounding noise) and the share of the five largest elements."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402

import test_parity_ssl_gpu as T  # noqa: E402
from test_parity_large_gpu import L2  # noqa: E402

KEYS = ["dino_head.mlp.0.bias", "dino_head.mlp.2.bias", "dino_head.mlp.4.bias", "dino_head.mlp.4.weight", "dino_head.last_layer.weight_v",
        "trunk.blocks.1.mlp.w2.bias", "trunk.blocks.1.attn.qkv.bias"]
for seed in [int(a) for a in sys.argv[1:]] or [41, 43]:
    c = T.Case(cfg_kw=L2, heads=(16, 16, 16), K=8192, res=512, seed=seed)
    tr, ssl = T._trainer(c, rec_weight=0.0)
    tr.step(c.img.to(T.DEV), None, ssl)
    torch.cuda.synchronize()
    params = dict(c.model.named_parameters())
    G = c.grads_ssl
    cos = lambda a, b: float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
    print(f"--- seed {seed}: ssl loss ours {float(tr.ssl_loss_sum):.6f} f32 {c.loss['f32']:.6f} cpu16 {c.loss['cpu16']:.6f} gpu16 {c.loss['gpu16']:.6f}")
    for k in KEYS:
        ref = G["f32"][k].flatten()
        eo, ec, eg = (params[k].grad.float().cpu().flatten() - ref), G["cpu16"][k].flatten() - ref, G["gpu16"][k].flatten() - ref
        top = lambda e: float(e.abs().topk(min(5, e.numel())).values.pow(2).sum() / e.pow(2).sum())
        print(f"{k:34s} |ref|={float(ref.norm()):.3e} E ours/cpu/gpu = {float(eo.norm() / ref.norm()):.3e} {float(ec.norm() / ref.norm()):.3e} "
              f"{float(eg.norm() / ref.norm()):.3e}  cos(o,c)={cos(eo, ec):+.2f} cos(o,g)={cos(eo, eg):+.2f} cos(c,g)={cos(ec, eg):+.2f}  "
              f"top5 share o/c/g = {top(eo):.2f} {top(ec):.2f} {top(eg):.2f}")
    del c, tr
    torch.cuda.empty_cache()
