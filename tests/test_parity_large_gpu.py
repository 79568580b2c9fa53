"""Training-step parity at BASELINE config 4's SHAPES (VERDICT r3 "next round" item 1a): VTP-Large widths (D = 1024, 16 heads,
SwiGLU hidden 2736 -> K and N tails in every MLP GEMM, 4-chunk norm rows) at 512 x 512 (N = 1025 tokens: the TILED attention
kernels forward AND backward, csrc/attention.hip, not the LDS-resident ones), depth 2 in all three towers, B = 2 images with
2 global 512^2 crops + 8 local 96^2 crops each (N = 1025 and N = 37 segments in one list forward), iBOT masks on the 32 x 32
grid, DINO head on 1024-wide tokens (K = 8192 prototypes so that the CPU oracle stays at seconds).

Compared against the oracle (oracle/vtp_oracle.py: `ssl_outputs`, `ssl_loss`, `rec_clip_train_loss`; the reference lines are
vit_large vision_transformer.py:352-361, attention.py:110-126, vtp.py:388-401,410-484) in fp32:
  * the teacher / student head outputs,
  * the gradients of the SSL-only step and of the FULL step (rec + clip + ssl) -- sampled keys of every tower plus ALL
    gradient tensors flat.
Protocol = tests/test_parity_ssl_gpu.py: E_ours <= 1.25 x E_ref, E_ref = the reference algorithm under bf16 autocast on the
CPU and on this MI355X; no absolute floors (one stated rule for one-element gradients).  The depth is 2 and not 24 because the
oracle runs on the host; the kernels' launch shapes (M, N, K, tails, N = 1025 tiling) are those of the full model."""
import pytest
import torch

from test_parity_ssl_gpu import DEV, HEAD_KEYS, LOSS_REL_FLOOR, Case, _compare_grads, _trainer, check

pytestmark = pytest.mark.gpu

L2 = dict(image_size=512, vision_embed_dim=1024, vision_depth=2, vision_num_heads=16, decoder_embed_dim=1024, decoder_depth=2,
          decoder_num_heads=16, text_embed_dim=1024, text_depth=2, text_num_heads=16)
TRUNK_KEYS = [f"trunk.blocks.{i}.{n}" for i in (0, 1)
              for n in ("attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "mlp.w1.weight", "mlp.w2.bias", "mlp.w3.weight",
                        "norm1.weight", "norm2.weight")] + \
             ["trunk.patch_embed.proj.weight", "trunk.patch_embed.proj.bias", "trunk.cls_token", "trunk.mask_token", "trunk.norm.weight"]
_CASE = []


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def case() -> Case:
    if not _CASE:
        _CASE.append(Case(cfg_kw=L2, heads=(16, 16, 16), K=8192, res=512, seed=41))
    return _CASE[0]


def test_large_512_ssl_head_outputs():
    """teacher / student dicts at D = 1024, N = 1025 (tiled attention forward) + N = 37"""
    c = case()
    col = c.col
    assert c.model.config.vision_embed_dim == 1024 and c.gc.shape[-1] == 512
    with torch.no_grad():
        c.model.eval()
        t_out, s_out = c.model.forward_ssl_learning(c.gc.to(DEV), 2, col["mask_indices_list"].to(DEV), int(col["n_masked_patches"]),
                                                    col["upperbound"], c.lc.to(DEV), c.masks.to(DEV))
        c.model.train()
    ref_t, ref_s = c.out["f32"]
    for k in ("teacher_cls_tokens_after_head", "masked_teacher_patch_tokens_after_head"):
        check(f"L512 teacher {k}", t_out[k], ref_t[k], c.out["cpu16"][0][k], c.out["gpu16"][0][k])
    for k in ("student_local_cls_tokens_after_head", "student_global_cls_tokens_after_head", "student_global_cls_tokens",
              "student_global_masked_patch_tokens_after_head"):
        check(f"L512 student {k}", s_out[k], ref_s[k], c.out["cpu16"][1][k], c.out["gpu16"][1][k])


def test_large_512_ssl_only_gradients():
    """DINO + iBOT alone: the tiled attention BACKWARD (dQ and dK/dV kernels at N = 1025) and the H = 2736 weight gradients"""
    c = case()
    tr, ssl = _trainer(c, rec_weight=0.0)
    tr.step(c.img.to(DEV), None, ssl)
    torch.cuda.synchronize()
    loss = float(tr.ssl_loss_sum)
    e, e_ref = abs(loss - c.loss["f32"]), max(abs(c.loss["cpu16"] - c.loss["f32"]), abs(c.loss["gpu16"] - c.loss["f32"]))
    print(f"PARITY L512 SSL loss: ours={loss:.6f} oracle fp32={c.loss['f32']:.6f} |err| ours={e:.2e} ref={e_ref:.2e}")
    assert e <= max(1.25 * e_ref, LOSS_REL_FLOOR * abs(c.loss["f32"]))
    _compare_grads("L512 SSL-only", dict(c.model.named_parameters()), HEAD_KEYS + TRUNK_KEYS, c.grads_ssl)


def test_large_512_full_step_gradients():
    """rec + clip + ssl at config-4 shapes: one list forward, one trunk backward, decoder at N = 1024, text tower at D = 1024"""
    c = case()
    tr, ssl = _trainer(c)
    l1, lc = tr.step(c.img.to(DEV), c.txt.to(DEV), ssl)
    torch.cuda.synchronize()
    print(f"PARITY L512 full step losses: ours L1={float(l1):.6f} clip={float(lc):.6f} ssl={float(tr.ssl_loss_sum):.6f} | oracle fp32 "
          f"L1={c.loss_full[0]:.6f} clip={c.loss_full[1]:.6f} ssl={c.loss_full[2]:.6f}")
    assert abs(float(l1) - c.loss_full[0]) < 2e-3 * c.loss_full[0]
    assert abs(float(lc) - c.loss_full[1]) < 5e-3 * max(c.loss_full[1], 1e-3)
    assert abs(float(tr.ssl_loss_sum) - c.loss_full[2]) < 1e-3 * c.loss_full[2]
    dec = [f"pixel_decoder.blocks.{i}.{n}" for i in (0, 1) for n in ("attn.qkv.weight", "attn.proj.weight", "mlp.w3.weight", "norm2.weight")]
    txt = [f"text_transformer.resblocks.{i}.{n}" for i in (0, 1) for n in ("attn.in_proj_weight", "mlp.c_fc.weight")]
    _compare_grads("L512 FULL step", dict(c.model.named_parameters()),
                   HEAD_KEYS + TRUNK_KEYS + dec + txt + ["pixel_decoder.proj_out.weight", "trunk.feature_bottleneck.weight",
                                                        "visual_proj.weight", "logit_scale"], c.grads_full)


def test_large_512_small_tensors_multi_seed():
    """The per-tensor statement for the SMALL gradient tensors (< 4096 elements: biases, gains, tokens) at the L-width / 512 x 512
    geometry, which the single-draw tests above only pool (`dino_head.mlp.4.bias` came out at 1.46 with one seed in
    profiles/r05_parity.log): the full step on THREE input draws (images, crops, captions, masks), ratio E_ours / E_ref per tensor and
    draw against BOTH comparators, mean over the draws <= 1.25 against the larger one (the protocol's rule) -- reported against each."""
    import numpy as np
    from test_parity_ssl_gpu import SMALL, relF
    keys = None
    ratios = {}
    for si, iseed in enumerate((7, 107, 207)):
        c = case() if si == 0 else Case(cfg_kw=L2, heads=(16, 16, 16), K=8192, res=512, seed=41, input_seed=iseed)
        tr, ssl = _trainer(c)
        tr.step(c.img.to(DEV), c.txt.to(DEV), ssl)
        torch.cuda.synchronize()
        params = dict(c.model.named_parameters())
        G = c.grads_full
        if keys is None:
            keys = [k for k in HEAD_KEYS + TRUNK_KEYS if 1 < G["f32"][k].numel() < SMALL]
            assert "dino_head.mlp.4.bias" in keys
        for k in keys:
            e = relF(params[k].grad, G["f32"][k])
            ec, eg = relF(G["cpu16"][k], G["f32"][k]), relF(G["gpu16"][k], G["f32"][k])
            ratios.setdefault(k, []).append((e / ec, e / eg, e / max(ec, eg)))
        if si:
            del c, tr, ssl, params, G
            torch.cuda.empty_cache()
    worst = 0.0
    for k in keys:
        r = np.array(ratios[k])
        print(f"PARITY L512 multi-seed small tensor {k}: per-draw ratios vs the larger comparator {np.round(r[:, 2], 2).tolist()} "
              f"mean={r[:, 2].mean():.2f} | vs cpu autocast mean={r[:, 0].mean():.2f} | vs cuda autocast mean={r[:, 1].mean():.2f}")
        worst = max(worst, float(r[:, 2].mean()))
    print(f"PARITY L512 multi-seed: worst mean ratio over {len(keys)} small tensors = {worst:.2f}")
    assert worst <= 1.25
