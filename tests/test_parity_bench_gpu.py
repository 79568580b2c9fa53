"""Parity AT THE BENCHMARKED GEOMETRY and a statistically sound bar for the small gradient tensors.

1. `test_bench_step_parity_at_b32`: the step bench.py times -- VTP-B f16d64, 32 images, 2 global 256^2 + 8 local 96^2 crops per image,
   K = 65536 prototypes, block-wise iBOT masks, rec + clip + DINO/iBOT through ONE list forward of M = 34 144 token rows, hipGraph
   segments AND side streams on (i.e. the gemm4w NT path at >= 192 tiles, the 2-slice one-wave grouped weight gradients, the merged
   96 x 257 attention launches, the text tower on its own stream) -- built with bench.py's own seeds and builders, replayed twice, and
   compared with the ORACLE (oracle/vtp_oracle.py, the restated reference algorithm) evaluated in fp32 ON THE GPU with stock
   PyTorch-ROCm kernels (test infrastructure; the CPU needs ~8 s per image for this step).  E_ref = the error of the same oracle under
   torch.autocast("cuda", bf16) -- the reference's own bf16 path on this MI355X.  Compared: the three loss
   values, a depth-sampled set of gradient tensors per tower (bar 1.25 each for tensors with >= 4096 elements), the pooled small
   tensors (1.25), and ALL gradient tensors flat (1.25).  The GPU-fp32 oracle is itself pinned to the CPU-fp32 oracle at batch 2 in
   test 2 (gradients agree to <= 2e-5 relative: it is an fp32-grade reference, not a reduced-precision one).

2. `test_small_tensor_ratios_multi_seed`: the relative error of a SMALL gradient tensor (bias, gain, token: a few hundred sums of
   rounding draws; `dino_head.mlp.4.bias` came out 1.46 with one seed and 0.65 with the next) is a noisy single-draw statistic, so
   rounds 3-4 allowed "2.0 each, 1.25 pooled".  That cap is replaced here by what the claim actually is: over >= 5 independent input
   seeds (images, crops, captions, masks) the MEAN ratio E_ours / E_ref of every sampled small tensor is <= 1.25, and so is the ratio of
   the root-mean-square errors.  VTP-B at batch 2, the same full step, oracle fp32 + autocast on the GPU."""
import contextlib
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu
DEV = "cuda"
HV = HD = HT = 12
N_LOCAL = 8
SMALL = 4096


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def relF(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-30))


def oracle_full_step(sd, img, txt, gc, lc, masks, c_d, c_i, autocast, dev=DEV):
    """l1 + clip + (dino + ibot) of the oracle with autograd; returns (losses, {name: grad}) -- grads stay on `dev` (fp32)."""
    from oracle import vtp_oracle as O
    ctx = (lambda: torch.autocast("cuda" if dev != "cpu" else "cpu", dtype=torch.bfloat16)) if autocast else contextlib.nullcontext
    s2 = {k: v.detach().clone().to(dev).requires_grad_(v.dtype == torch.float32 and not k.startswith("teacher_")) for k, v in sd.items()}
    mv = lambda t: t.to(dev)
    with ctx():
        t_out, s_out = O.ssl_outputs(s2, mv(gc), mv(lc), mv(masks), HV)
        l_ssl = O.ssl_loss({k: v.float() for k, v in t_out.items()}, {k: v.float() for k, v in s_out.items()}, mv(masks), mv(c_d), mv(c_i),
                           N_LOCAL)
    l_ssl.backward()
    del t_out, s_out
    with ctx():
        l1, lc_ = O.rec_clip_train_loss(s2, mv(img), mv(txt), HV, HD, HT)
    (l1 + lc_).backward()
    losses = (float(l1.detach()), float(lc_.detach()), float(l_ssl.detach()))
    grads = {k: v.grad.detach().float() for k, v in s2.items() if v.grad is not None}
    del s2
    return losses, grads


def sampled_keys():
    blk = ("attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias", "mlp.w1.weight", "mlp.w2.bias", "mlp.w3.weight",
           "mlp.w3.bias", "norm1.weight", "norm2.weight")
    trunk = [f"trunk.blocks.{i}.{n}" for i in (0, 3, 6, 9, 11) for n in blk] + \
        ["trunk.patch_embed.proj.weight", "trunk.patch_embed.proj.bias", "trunk.cls_token", "trunk.mask_token", "trunk.norm.weight",
         "trunk.feature_bottleneck.weight"]
    dec = [f"pixel_decoder.blocks.{i}.{n}" for i in (0, 5, 11) for n in ("attn.qkv.weight", "attn.proj.bias", "mlp.w1.weight", "mlp.w3.weight",
                                                                        "norm2.weight", "norm2.bias")] + \
        ["pixel_decoder.proj_in.weight", "pixel_decoder.proj_out.weight", "pixel_decoder.proj_out.bias", "pixel_decoder.norm.weight"]
    txt = [f"text_transformer.resblocks.{i}.{n}" for i in (0, 6, 11) for n in ("attn.in_proj_weight", "attn.out_proj.bias", "mlp.c_fc.weight",
                                                                               "mlp.c_proj.weight", "ln_1.weight")] + \
        ["token_embedding.weight", "positional_embedding", "ln_final.weight", "text_projection"]
    head = ["dino_head.mlp.0.weight", "dino_head.mlp.0.bias", "dino_head.mlp.2.weight", "dino_head.mlp.4.weight", "dino_head.mlp.4.bias",
            "dino_head.last_layer.weight_v", "dino_head.last_layer.weight_g", "visual_proj.weight", "logit_scale"]
    return trunk + dec + txt + head


def compare(tag, ours, ref, ref16, keys):
    """ours / ref / ref16: name -> fp32 gradient.  Large sampled tensors 1.25 each; small ones pooled 1.25 (their per-tensor statement is
    the multi-seed test); one-element gradients floored at visual_proj.weight's E_ref; all tensors flat 1.25.  Returns the ratios."""
    vp = "visual_proj.weight"
    floor = relF(ref16[vp], ref[vp]) if vp in ref else 0.0
    pool = [0.0, 0.0, 0.0, 0]
    ratios, worst, worst_k = {}, 0.0, None
    for k in keys:
        if k not in ref:
            continue
        n = ref[k].numel()
        e, er = relF(ours[k], ref[k]), max(relF(ref16[k], ref[k]), floor if n == 1 else 0.0)
        r = e / max(er, 1e-30)
        ratios[k] = r
        print(f"PARITY {tag} grad {k} [{n}]: E_ours={e:.3e} E_ref(cuda autocast)={er:.3e} E_ours/E_ref={r:.2f}")
        if n >= SMALL:
            assert e <= 1.25 * er, f"{tag} {k}: E_ours {e:.3e} > 1.25 x E_ref {er:.3e}"
            if r > worst:
                worst, worst_k = r, k
        elif n > 1:
            pool[0] += float((ours[k] - ref[k]).pow(2).sum())
            pool[1] += float((ref16[k] - ref[k]).pow(2).sum())
            pool[2] += float(ref[k].pow(2).sum())
            pool[3] += 1
        else:
            assert e <= 1.25 * er, f"{tag} {k}: E_ours {e:.3e} > 1.25 x E_ref {er:.3e}"
    if pool[3]:
        e, er = (pool[0] / pool[2]) ** 0.5, (pool[1] / pool[2]) ** 0.5
        print(f"PARITY {tag} POOLED {pool[3]} sampled tensors with < {SMALL} elements: E_ours={e:.3e} E_ref={er:.3e} E_ours/E_ref={e / er:.2f}")
        assert e <= 1.25 * er
    num = den = rr = 0.0
    cnt = 0
    for k, g in ref.items():
        if k not in ours:
            continue
        num += float((ours[k] - g).pow(2).sum())
        rr += float((ref16[k] - g).pow(2).sum())
        den += float(g.pow(2).sum())
        cnt += 1
    e, er = (num / den) ** 0.5, (rr / den) ** 0.5
    print(f"PARITY {tag} ALL {cnt} gradient tensors (flat): E_ours={e:.3e} E_ref={er:.3e} E_ours/E_ref={e / er:.2f}; worst sampled large "
          f"tensor {worst_k}: {worst:.2f}")
    assert e <= 1.25 * er
    return ratios


def test_bench_step_parity_at_b32():
    import bench
    from vtp_amd import VTP, VTPConfig, VTPTrainer
    from vtp_amd.engine import OVERLAP
    assert OVERLAP.enabled, "the benchmarked step runs with the side streams on"
    B, res, K = 32, 256, 65536
    dev = torch.device("cuda", 0)
    # ---- exactly bench.py's builders and seeds (rank 0); lr = 0 / momentum 1 so that the replayed steps leave the weights where the
    # oracle reads them (the optimizer, EMA and weight-refresh kernels still run inside the captured step)
    torch.manual_seed(0)
    model = VTP(VTPConfig(), dino_out_dim=K).to(dev)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    trainer = VTPTrainer(model, lr=0.0, betas=(0.9, 0.95), weight_decay=0.0, teacher_momentum=1.0, use_graphs=True)
    img = torch.randn(B, 3, res, res, device=dev, generator=torch.Generator(device=dev).manual_seed(1234))
    txt = bench.synthetic_captions(B, model.config.text_context_length, model.config.text_vocab_size, dev, 4321)
    gc, lc = bench.synthetic_crops(B, res, dev, 777)
    masks, upper = bench.MaskStream(B, res, 555).draw()
    for rep in range(2):  # first call: eager warm-up + capture + replay; second: replay only
        # the teacher-softmax centres this step reads: zeros at step 0, the EMA of the first batch's statistics at the replayed step
        c_d, c_i = trainer.center_dino.detach().clone(), trainer.center_ibot.detach().clone()
        ssl = trainer.prepare_ssl(gc, lc, masks, upperbound=upper)
        l1, lcl = trainer.step(img, txt, ssl)
    torch.cuda.synchronize()
    assert float(c_d.abs().max()) > 0, "the replayed step must see the centres the first step left"
    assert trainer._graphs, "the step must have run from captured hipGraph segments"
    M = sum(g.B * g.N for g in trainer.trunk.ctx().segs)
    assert M == 32 * 257 + 64 * 257 + 256 * 37 == 34144
    ours_loss = (float(l1), float(lcl), float(trainer.ssl_loss_sum))
    ours = {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None and not n.startswith("teacher_")}
    masks_t = masks.bool() if torch.is_tensor(masks) else torch.as_tensor(np.asarray(masks), dtype=torch.bool)
    del trainer, model
    torch.cuda.empty_cache()
    # ---- the oracle on the same inputs, fp32 and under cuda autocast, on the GPU
    ref_loss, ref = oracle_full_step(sd, img, txt, gc, lc, masks_t, c_d, c_i, autocast=False)
    torch.cuda.empty_cache()
    r16_loss, ref16 = oracle_full_step(sd, img, txt, gc, lc, masks_t, c_d, c_i, autocast=True)
    torch.cuda.empty_cache()
    print(f"PARITY bench-geometry losses (L1, clip, ssl): ours={ours_loss} oracle fp32={ref_loss} oracle cuda-autocast={r16_loss}")
    for name, o, r, r16, floor in zip(("L1", "clip", "ssl"), ours_loss, ref_loss, r16_loss, (2e-3, 5e-3, 3e-4)):
        # (floors: the stated ones of tests/test_parity_ssl_gpu.py -- a loss value is ONE number, its E_ref a single draw)
        e, er = abs(o - r), abs(r16 - r)
        print(f"PARITY bench-geometry loss {name}: |err| ours={e:.3e} ref={er:.3e}")
        assert e <= max(1.25 * er, floor * abs(r)), f"{name} loss: {o} vs oracle {r} (autocast {r16})"
    missing = [k for k in ref if k not in ours]
    assert not missing, f"gradients the oracle has and the step does not: {missing[:5]}"
    compare("BENCH-GEOMETRY FULL step (B=32, graphs + side streams)", ours, ref, ref16, sampled_keys())


SEEDS = (101, 202, 303, 404, 505)


def test_small_tensor_ratios_multi_seed():
    import bench
    from vtp_amd import VTP, VTPConfig, VTPTrainer
    from vtp_amd.data import collate_ssl_masks
    B, res, K = 2, 256, 65536
    dev = torch.device("cuda", 0)
    torch.manual_seed(31)
    model = VTP(VTPConfig(), dino_out_dim=K)
    with torch.no_grad():  # a model mid-training: perturbed 1-D parameters, a teacher that differs from the student (as test_parity_ssl_gpu)
        for n, p in model.named_parameters():
            if p.ndim <= 1 and n != "logit_scale" and not n.startswith("teacher_"):
                p.add_(0.05 * torch.randn_like(p))
        sd0 = model.state_dict()
        for n, p in model.named_parameters():
            if n.startswith("teacher_trunk.") or n.startswith("teacher_dino_head."):
                s = sd0[n.replace("teacher_trunk.", "trunk.").replace("teacher_dino_head.", "dino_head.")]
                p.copy_(s + (0.002 if p.ndim >= 2 else 0.01) * torch.randn_like(s))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(dev)
    trainer = VTPTrainer(model, lr=0.0, weight_decay=0.0, teacher_momentum=1.0)
    keys = [k for k in sampled_keys() if 1 < sd[k].numel() < SMALL]
    assert "dino_head.mlp.4.bias" in keys and len(keys) >= 30
    g0 = torch.Generator().manual_seed(5)
    c_d, c_i = 0.3 * torch.randn(K, generator=g0), 0.3 * torch.randn(K, generator=g0)
    e_ours = {k: [] for k in keys}
    e_ref = {k: [] for k in keys}
    for si, seed in enumerate(SEEDS):
        g = torch.Generator().manual_seed(seed)
        img = torch.randn(B, 3, res, res, generator=g)
        gc = torch.randn(2 * B, 3, res, res, generator=g)
        lc = torch.randn(N_LOCAL * B, 3, 96, 96, generator=g)
        txt = bench.synthetic_captions(B, model.config.text_context_length, model.config.text_vocab_size, "cpu", seed + 1)
        col = collate_ssl_masks(2 * B, (res // 16, res // 16), 0.5, (0.1, 0.5), np.random.default_rng(seed + 2))
        masks = col["masks"]
        trainer.center_dino.copy_(c_d)
        trainer.center_ibot.copy_(c_i)
        ssl = trainer.prepare_ssl(gc.to(dev), lc.to(dev), masks, upperbound=col["upperbound"])
        trainer.step(img.to(dev), txt.to(dev), ssl)
        torch.cuda.synchronize()
        ours = {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if n in e_ours}
        _, ref = oracle_full_step(sd, img, txt, gc, lc, masks, c_d, c_i, autocast=False)
        _, ref16 = oracle_full_step(sd, img, txt, gc, lc, masks, c_d, c_i, autocast=True)
        if si == 0:  # pin the GPU-fp32 oracle to the CPU-fp32 oracle (same code, stock kernels of either device)
            _, ref_cpu = oracle_full_step(sd, img, txt, gc, lc, masks, c_d, c_i, autocast=False, dev="cpu")
            num = sum(float((ref[k].cpu() - v).pow(2).sum()) for k, v in ref_cpu.items())
            den = sum(float(v.pow(2).sum()) for v in ref_cpu.values())
            print(f"PARITY oracle fp32 on the GPU vs on the CPU, all gradients flat: rel {(num / den) ** 0.5:.2e}")
            assert (num / den) ** 0.5 <= 2e-5
            del ref_cpu
        for k in keys:
            e_ours[k].append(relF(ours[k], ref[k]))
            e_ref[k].append(relF(ref16[k], ref[k]))
        del ref, ref16
        torch.cuda.empty_cache()
    worst = 0.0
    for k in keys:
        eo, er = np.array(e_ours[k]), np.array(e_ref[k])
        mean_ratio, rms_ratio = float(np.mean(eo / er)), float(np.sqrt(np.mean(eo ** 2)) / np.sqrt(np.mean(er ** 2)))
        print(f"PARITY multi-seed small tensor {k} [{sd[k].numel()}]: per-seed ratios {np.round(eo / er, 2).tolist()} mean={mean_ratio:.2f} "
              f"rms-ratio={rms_ratio:.2f}")
        worst = max(worst, mean_ratio)
        assert mean_ratio <= 1.25 and rms_ratio <= 1.25, f"{k}: mean ratio {mean_ratio:.2f} / rms ratio {rms_ratio:.2f} over {len(SEEDS)} seeds"
    print(f"PARITY multi-seed: {len(keys)} small tensors x {len(SEEDS)} seeds, worst mean ratio {worst:.2f}")
