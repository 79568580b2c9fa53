"""SelfAttentionBlock residual-wiring extras: LayerScale forward + backward (misc.py:7-26) and stochastic
depth (block.py:20-118,207-289) -- kernels vs torch, then the whole rec train step (trunk + pixel decoder) against the oracle's
autograd, the oracle being pinned to the REAL reference for exactly these branches (tests/test_oracle_vs_reference.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def relF(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_block_extra_kernels_vs_torch():
    from vtp_amd import ops as o
    g = torch.Generator(device=DEV).manual_seed(2)
    B, N, D = 7, 17, 128
    x = torch.randn(B * N, D, device=DEV, generator=g)
    idx = torch.tensor([5, 0, 3, 6], dtype=torch.int32, device=DEV)
    dst = torch.full((4 * N, D), float("nan"), device=DEV)
    dst_b = torch.full((4 * N, D), float("nan"), dtype=torch.bfloat16, device=DEV)
    o.gather_image_rows(x, idx, dst, dst_b, 4, N, D, 1.75)
    ref = x.view(B, N, D)[idx.long()].reshape(4 * N, D)
    assert torch.equal(dst, ref) and torch.equal(dst_b, (ref * 1.75).to(torch.bfloat16))
    y = x.clone()
    src = torch.randn(4 * N, D, device=DEV, generator=g)
    o.scatter_image_rows(src, idx, y, 4, N, D, 1.75, True)
    ref_y = torch.index_add(x.view(B, N, D), 0, idx.long(), src.view(4, N, D), alpha=1.75).view(-1, D)
    assert torch.allclose(y, ref_y, rtol=1e-6, atol=1e-6)
    o.scatter_image_rows(src, idx, y, 4, N, D, 1.0, False)
    assert torch.equal(y.view(B, N, D)[idx.long()].reshape(-1, D), src) and torch.equal(y.view(B, N, D)[1], ref_y.view(B, N, D)[1])
    # LayerScale wgrad finish + scaled transpose
    Nn, K = 96, 200
    G = torch.randn(Nn, K, device=DEV, generator=g)
    W = torch.randn(Nn, K, device=DEV, generator=g)
    bias, cs, gamma = (torch.randn(Nn, device=DEV, generator=g) for _ in range(3))
    dW, db, dg = torch.ones(Nn, K, device=DEV), torch.ones(Nn, device=DEV), torch.ones(Nn, device=DEV)
    o.layerscale_wgrad(G, W, bias, cs, gamma, dW, db, dg, Nn, K)
    assert torch.allclose(dW, 1 + gamma[:, None] * G, rtol=1e-5, atol=1e-5) and torch.allclose(db, 1 + gamma * cs, rtol=1e-5, atol=1e-5)
    assert torch.allclose(dg, 1 + (W * G).sum(1) + bias * cs, rtol=1e-4, atol=1e-3)
    wt = torch.empty(K, Nn, dtype=torch.bfloat16, device=DEV)
    o.scaled_transpose(W, gamma, wt, Nn, K)
    assert torch.equal(wt, (W * gamma[:, None]).T.to(torch.bfloat16))


def _model(init_values, **extra):
    from oracle.ref_stubs import TINY
    from vtp_amd import VTPConfig, VTPModel
    cfg = dict(TINY)
    cfg.update(vision_depth=3, decoder_depth=2, vision_init_values=init_values, decoder_init_values=init_values and 0.6 * init_values)
    cfg.update(extra)
    torch.manual_seed(8)
    m = VTPModel(VTPConfig(**cfg))
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.ndim <= 1 and p.numel() > 1:
                p.add_(0.05 * torch.randn_like(p))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    return m.to(DEV), sd


def _compare_grads(m, ref_sd, keys, tol, ref16=None):
    """E_ours vs the fp32 oracle autograd; with `ref16` (the same oracle graph under bf16 autocast) the bound is the usual
    max(1.5 E_ref, tol) of the parity protocol"""
    params = dict(m.named_parameters())
    for k in keys:
        e = relF(params[k].grad, ref_sd[k].grad)
        e_ref = relF(ref16[k].grad, ref_sd[k].grad) if ref16 is not None else 0.0
        print(f"   grad {k}: E_ours {e:.3e} E_ref {e_ref:.3e}")
        assert e <= max(1.5 * e_ref, tol), k


def test_layerscale_forward_backward_vs_oracle():
    from oracle import vtp_oracle as O
    from vtp_amd import VTPTrainer
    m, sd = _model(0.5)
    assert "trunk.blocks.0.ls1.gamma" in sd and "pixel_decoder.blocks.1.ls2.gamma" in sd
    img = torch.randn(4, 3, 64, 64, generator=torch.Generator().manual_seed(3))
    m.eval()
    lat = m.get_reconstruction_latents(img.to(DEV))
    with torch.no_grad():
        lat_ref = O.reconstruction_latents(sd, img, 2)
    print(f"LayerScale latents rel {relF(lat, lat_ref):.3e}")
    assert relF(lat, lat_ref) < 1.5e-2
    m.train()
    ref = {k: v.clone().requires_grad_(v.dtype == torch.float32) for k, v in sd.items()}
    loss_ref = O.rec_train_loss(ref, img, 2, 2)
    loss_ref.backward()
    tr = VTPTrainer(m, lr=0.0, weight_decay=0.0)
    loss = tr.step_rec(img.to(DEV))
    torch.cuda.synchronize()
    assert abs(float(loss) - float(loss_ref)) < 3e-3 * float(loss_ref)
    _compare_grads(m, ref, ["trunk.blocks.0.ls1.gamma", "trunk.blocks.2.ls2.gamma", "pixel_decoder.blocks.0.ls1.gamma",
                            "pixel_decoder.blocks.1.ls2.gamma", "trunk.blocks.1.attn.proj.weight", "trunk.blocks.1.attn.proj.bias",
                            "trunk.blocks.0.mlp.w3.weight", "trunk.blocks.2.mlp.w3.bias", "trunk.blocks.0.attn.qkv.weight",
                            "pixel_decoder.blocks.1.mlp.w3.bias", "pixel_decoder.blocks.0.mlp.w1.weight", "trunk.patch_embed.proj.weight"], 4e-2)


@pytest.mark.parametrize("init_values", [None, 0.5])
def test_stochastic_depth_step_vs_oracle(init_values):
    """drop_rate 0.4 on 5 images (keep 3, alpha 5/3) in trunk AND decoder: loss and gradients of the rec step vs the oracle's
    autograd evaluated with the SAME image subsets (read back from the trainer's drop plan)."""
    from oracle import vtp_oracle as O
    from vtp_amd import VTPTrainer
    m, sd = _model(init_values)
    B = 5
    img = torch.randn(B, 3, 64, 64, generator=torch.Generator().manual_seed(4))
    tr = VTPTrainer(m, lr=0.0, weight_decay=0.0, drop_rate=0.4, decoder_drop_rate=0.4, drop_seed=3)
    loss = tr.step_rec(img.to(DEV))
    torch.cuda.synchronize()

    def plan_of(stack):
        p = stack.last_drop_plan  # the trainer clears the active plan when the step ends
        keep, alpha = p["keeps"][0], p["scales"][0]
        assert keep == 3 and abs(alpha - B / 3) < 1e-9
        idx = p["idx_dev"].cpu().long().view(stack.depth, 2, keep)
        return [(idx[i, 0], alpha, idx[i, 1], alpha) for i in range(stack.depth)]

    d_tr, d_dec = plan_of(tr.trunk.stack), plan_of(tr.decoder.stack)
    ref = {k: v.clone().requires_grad_(v.dtype == torch.float32) for k, v in sd.items()}
    out = O.trunk_forward(ref, img, 2, use_bottleneck=True, drop=d_tr)
    lat = out["x_norm_patchtokens"].transpose(1, 2).reshape(B, -1, 4, 4)
    loss_ref = O.l1_loss(O.decoder_forward(ref, lat, 2, drop=d_dec), img)
    loss_ref.backward()
    ref16 = {k: v.clone().requires_grad_(v.dtype == torch.float32) for k, v in sd.items()}
    with torch.autocast("cpu", dtype=torch.bfloat16):
        o16 = O.trunk_forward(ref16, img, 2, use_bottleneck=True, drop=d_tr)
        l16 = O.l1_loss(O.decoder_forward(ref16, o16["x_norm_patchtokens"].transpose(1, 2).reshape(B, -1, 4, 4), 2, drop=d_dec), img)
    l16.backward()
    print(f"stochastic depth (LayerScale {init_values}): loss ours {float(loss):.5f} oracle {float(loss_ref):.5f}")
    assert abs(float(loss) - float(loss_ref)) < 3e-3 * float(loss_ref)
    keys = ["trunk.blocks.0.attn.qkv.weight", "trunk.blocks.1.attn.proj.bias", "trunk.blocks.2.mlp.w3.weight", "trunk.blocks.1.mlp.w1.bias",
            "trunk.blocks.0.norm1.weight", "trunk.blocks.2.norm2.weight", "trunk.patch_embed.proj.weight", "trunk.cls_token",
            "pixel_decoder.blocks.0.attn.qkv.bias", "pixel_decoder.blocks.1.mlp.w3.bias", "pixel_decoder.blocks.1.norm2.bias",
            "pixel_decoder.proj_in.weight"]
    if init_values:
        keys += ["trunk.blocks.1.ls1.gamma", "pixel_decoder.blocks.0.ls2.gamma"]
    _compare_grads(m, ref, keys, 3e-2, ref16)
    # a second step draws new subsets (same shapes: static index buffer refreshed in place) and the graph path agrees with eager
    i0 = tr.trunk.stack.last_drop_plan["idx_dev"].clone()
    tr.step_rec(img.to(DEV))
    assert not torch.equal(i0, tr.trunk.stack.last_drop_plan["idx_dev"])
    # the plan does not outlive the step: an evaluation pass with another batch size afterwards runs the plain path
    assert tr.trunk.stack.drop_plan is None and tr.decoder.stack.drop_plan is None
    feats = m.get_intermediate_layers_feature(img[:2].to(DEV), n=1)
    assert torch.isfinite(feats[0]).all()
    res = []
    for use_graphs in (False, True):
        m2, _ = _model(init_values)
        t2 = VTPTrainer(m2, lr=1e-3, weight_decay=0.0, drop_rate=0.4, decoder_drop_rate=0.4, drop_seed=9, use_graphs=use_graphs)
        res.append([float(t2.step_rec(img.to(DEV) + 0.01 * i)) for i in range(3)])
    print("drop eager:", res[0], "graphs:", res[1])
    for a, b in zip(*res):
        assert abs(a - b) < 2e-3 * abs(a)


def test_stochastic_depth_with_mlp_ffn_vs_oracle():
    """ffn_layer = "mlp" (fc1 -> GELU -> fc2, ffn.py:21-48) under stochastic depth (block.py:207-289), trunk AND decoder -- the
    combination round 4 rejected at set_drop_plan time: same protocol as the SwiGLU test above"""
    from oracle import vtp_oracle as O
    from vtp_amd import VTPTrainer
    m, sd = _model(None, vision_ffn_layer="mlp", decoder_ffn_layer="mlp")
    assert "trunk.blocks.0.mlp.fc1.weight" in sd and "pixel_decoder.blocks.1.mlp.fc2.bias" in sd
    B = 5
    img = torch.randn(B, 3, 64, 64, generator=torch.Generator().manual_seed(4))
    tr = VTPTrainer(m, lr=0.0, weight_decay=0.0, drop_rate=0.4, decoder_drop_rate=0.4, drop_seed=5)
    loss = tr.step_rec(img.to(DEV))
    torch.cuda.synchronize()

    def plan_of(stack):
        p = stack.last_drop_plan
        keep, alpha = p["keeps"][0], p["scales"][0]
        idx = p["idx_dev"].cpu().long().view(stack.depth, 2, keep)
        return [(idx[i, 0], alpha, idx[i, 1], alpha) for i in range(stack.depth)]

    d_tr, d_dec = plan_of(tr.trunk.stack), plan_of(tr.decoder.stack)
    ref = {k: v.clone().requires_grad_(v.dtype == torch.float32) for k, v in sd.items()}
    out = O.trunk_forward(ref, img, 2, use_bottleneck=True, drop=d_tr)
    loss_ref = O.l1_loss(O.decoder_forward(ref, out["x_norm_patchtokens"].transpose(1, 2).reshape(B, -1, 4, 4), 2, drop=d_dec), img)
    loss_ref.backward()
    ref16 = {k: v.clone().requires_grad_(v.dtype == torch.float32) for k, v in sd.items()}
    with torch.autocast("cpu", dtype=torch.bfloat16):
        o16 = O.trunk_forward(ref16, img, 2, use_bottleneck=True, drop=d_tr)
        l16 = O.l1_loss(O.decoder_forward(ref16, o16["x_norm_patchtokens"].transpose(1, 2).reshape(B, -1, 4, 4), 2, drop=d_dec), img)
    l16.backward()
    print(f"stochastic depth with the Mlp FFN: loss ours {float(loss):.5f} oracle {float(loss_ref):.5f}")
    assert abs(float(loss) - float(loss_ref)) < 3e-3 * float(loss_ref)
    _compare_grads(m, ref, ["trunk.blocks.0.mlp.fc1.weight", "trunk.blocks.1.mlp.fc1.bias", "trunk.blocks.2.mlp.fc2.weight",
                            "trunk.blocks.1.mlp.fc2.bias", "trunk.blocks.0.attn.qkv.weight", "trunk.blocks.2.norm2.weight",
                            "pixel_decoder.blocks.0.mlp.fc1.weight", "pixel_decoder.blocks.1.mlp.fc2.bias", "pixel_decoder.proj_in.weight",
                            "trunk.patch_embed.proj.weight"], 3e-2, ref16)
    res = []
    for use_graphs in (False, True):
        m2, _ = _model(None, vision_ffn_layer="mlp", decoder_ffn_layer="mlp")
        t2 = VTPTrainer(m2, lr=1e-3, weight_decay=0.0, drop_rate=0.4, decoder_drop_rate=0.4, drop_seed=9, use_graphs=use_graphs)
        res.append([float(t2.step_rec(img.to(DEV) + 0.01 * i)) for i in range(3)])
    for a, b in zip(*res):
        assert abs(a - b) < 2e-3 * abs(a)


def test_stochastic_depth_with_qk_norm_vs_oracle():
    """QK normalisation (attention.py:67-68,119-120) inside the sample-drop branch (block.py:207-289) -- the combination rounds 2-4 left
    out: projection -> per-head RMSNorm of q, k -> RoPE on the kept images, and its backward incl. the q / k norm weights"""
    from oracle import vtp_oracle as O
    from vtp_amd import VTPTrainer
    m, sd = _model(None, vision_use_qk_norm=True, decoder_use_qk_norm=True)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "q_norm" in n or "k_norm" in n:
                p.copy_(1.0 + 0.3 * torch.randn_like(p))
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    B = 5
    img = torch.randn(B, 3, 64, 64, generator=torch.Generator().manual_seed(4))
    tr = VTPTrainer(m, lr=0.0, weight_decay=0.0, drop_rate=0.4, decoder_drop_rate=0.4, drop_seed=6)
    loss = tr.step_rec(img.to(DEV))
    torch.cuda.synchronize()

    def plan_of(stack):
        p = stack.last_drop_plan
        keep, alpha = p["keeps"][0], p["scales"][0]
        idx = p["idx_dev"].cpu().long().view(stack.depth, 2, keep)
        return [(idx[i, 0], alpha, idx[i, 1], alpha) for i in range(stack.depth)]

    d_tr, d_dec = plan_of(tr.trunk.stack), plan_of(tr.decoder.stack)

    def ref_loss(s_):
        out = O.trunk_forward(s_, img, 2, use_bottleneck=True, drop=d_tr)
        return O.l1_loss(O.decoder_forward(s_, out["x_norm_patchtokens"].transpose(1, 2).reshape(B, -1, 4, 4), 2, drop=d_dec), img)

    ref = {k: v.clone().requires_grad_(v.dtype == torch.float32) for k, v in sd.items()}
    loss_ref = ref_loss(ref)
    loss_ref.backward()
    ref16 = {k: v.clone().requires_grad_(v.dtype == torch.float32) for k, v in sd.items()}
    with torch.autocast("cpu", dtype=torch.bfloat16):
        l16 = ref_loss(ref16)
    l16.backward()
    print(f"stochastic depth with QK normalisation: loss ours {float(loss):.5f} oracle {float(loss_ref):.5f}")
    assert abs(float(loss) - float(loss_ref)) < 3e-3 * float(loss_ref)
    _compare_grads(m, ref, ["trunk.blocks.0.attn.q_norm.weight", "trunk.blocks.1.attn.k_norm.weight", "pixel_decoder.blocks.0.attn.q_norm.weight",
                            "trunk.blocks.0.attn.qkv.weight", "trunk.blocks.2.attn.qkv.bias", "trunk.blocks.1.mlp.w3.weight",
                            "pixel_decoder.blocks.1.attn.qkv.weight", "trunk.patch_embed.proj.weight"], 3e-2, ref16)
    res = []
    for use_graphs in (False, True):
        m2, _ = _model(None, vision_use_qk_norm=True, decoder_use_qk_norm=True)
        t2 = VTPTrainer(m2, lr=1e-3, weight_decay=0.0, drop_rate=0.4, decoder_drop_rate=0.4, drop_seed=9, use_graphs=use_graphs)
        res.append([float(t2.step_rec(img.to(DEV) + 0.01 * i)) for i in range(3)])
    for a, b in zip(*res):
        assert abs(a - b) < 2e-3 * abs(a)


def test_rope_train_time_augmentations_vs_oracle():
    """RopePositionEmbedding shift / jitter / rescale (embeddings.py:155-171; `pos_embed_rope_*_coords` of the ViT classes, reachable
    through the legacy YAML): in training the trunk draws new coordinates per BLOCK (rope_embed sits inside its block loop,
    vision_transformer.py:228-233), the pixel decoder once per forward (pixel_decoder.py:144).  The step's loss and gradients against
    the oracle evaluated with the SAME draws (read back from the engines; the oracle's augmented tables are pinned to the real class
    bit for bit by tests/test_oracle_vs_reference.py), evaluation passes stay un-augmented, graphs agree with eager."""
    from oracle import vtp_oracle as O
    from vtp_amd import VTPTrainer
    kw = dict(vision_rope_shift_coords=0.2, vision_rope_jitter_coords=1.3, vision_rope_rescale_coords=1.5,
              decoder_rope_shift_coords=0.1, decoder_rope_rescale_coords=1.2)
    m, sd = _model(None, **kw)
    B = 3
    img = torch.randn(B, 3, 64, 64, generator=torch.Generator().manual_seed(4))
    m.eval()
    with torch.no_grad():  # evaluation: plain tables
        lat = m.get_reconstruction_latents(img.to(DEV))
        lat_ref = O.reconstruction_latents(sd, img, 2)
    assert relF(lat, lat_ref) < 1.5e-2
    m.train()
    tr = VTPTrainer(m, lr=0.0, weight_decay=0.0)
    KEYS = ["trunk.blocks.0.attn.qkv.weight", "trunk.blocks.2.attn.qkv.bias", "trunk.blocks.1.attn.proj.weight",
            "trunk.blocks.0.mlp.w3.weight", "trunk.patch_embed.proj.weight", "pixel_decoder.blocks.0.attn.qkv.weight",
            "pixel_decoder.blocks.1.attn.qkv.bias", "pixel_decoder.blocks.1.mlp.w1.weight", "pixel_decoder.proj_in.weight"]

    def check(trn, mod, loss, what):
        """loss / gradients of the step that just ran against the oracle evaluated with THAT step's draws"""
        torch.cuda.synchronize()
        d_tr, d_dec = trn.trunk.rope_aug.last_draws, trn.decoder.rope_aug.last_draws
        assert len(d_tr) == trn.trunk.depth and len(d_tr[0]) == 1 and len(d_dec) == 1
        assert not torch.equal(d_tr[0][0]["shift"], d_tr[1][0]["shift"]), "every block draws its own coordinates"
        aug_tr, aug_dec = [row[0] for row in d_tr], d_dec[0][0]

        def ref_loss(s_):
            out = O.trunk_forward(s_, img, 2, use_bottleneck=True, rope_aug=aug_tr)
            return O.l1_loss(O.decoder_forward(s_, out["x_norm_patchtokens"].transpose(1, 2).reshape(B, -1, 4, 4), 2, rope_aug=aug_dec), img)

        ref = {k: v.clone().requires_grad_(v.dtype == torch.float32) for k, v in sd.items()}
        loss_ref = ref_loss(ref)
        loss_ref.backward()
        ref16 = {k: v.clone().requires_grad_(v.dtype == torch.float32) for k, v in sd.items()}
        with torch.autocast("cpu", dtype=torch.bfloat16):
            l16 = ref_loss(ref16)
        l16.backward()
        print(f"RoPE augmentation [{what}]: loss ours {float(loss):.5f} oracle (same draws) {float(loss_ref):.5f}")
        assert abs(float(loss) - float(loss_ref)) < 3e-3 * float(loss_ref)
        _compare_grads(mod, ref, KEYS, 3e-2, ref16)
        return d_tr, d_dec

    loss = tr.step_rec(img.to(DEV))
    d_tr, d_dec = check(tr, m, loss, "eager, step 1")
    with torch.no_grad():
        plain = O.rec_train_loss(sd, img, 2, 2)
    print(f"   oracle without augmentation {float(plain):.5f}")
    # fresh draws every step -- and the forward rotates with THEM (the decoder's fused qkv + RoPE epilogue used to keep a
    # cached copy of the first step's table while the backward un-rotated with the fresh one): step 2 against the oracle with step 2's draws
    loss2 = tr.step_rec(img.to(DEV))
    d_tr2, d_dec2 = check(tr, m, loss2, "eager, step 2")
    assert not torch.equal(d_tr2[0][0]["shift"], d_tr[0][0]["shift"]) and not torch.equal(d_dec2[0][0]["shift"], d_dec[0][0]["shift"])
    # the hipGraph path re-draws into the static buffers before each replay: lr = 0, so every replayed step is comparable to the oracle
    m3, _ = _model(None, **kw)
    t3 = VTPTrainer(m3, lr=0.0, weight_decay=0.0, use_graphs=True)
    for _ in range(3):
        loss3 = t3.step_rec(img.to(DEV))
    check(t3, m3, loss3, "graphs, step 3")
    m2, _ = _model(None, **kw)
    t2 = VTPTrainer(m2, lr=1e-3, weight_decay=0.0, use_graphs=True)
    ls = [float(t2.step_rec(img.to(DEV))) for _ in range(4)]
    draws = t2.trunk.rope_aug.last_draws
    assert all(l == l for l in ls) and ls[-1] < ls[0] + 0.05 and len(draws) == t2.trunk.depth
    print("RoPE augmentation, graph path losses:", ls)


def test_qk_norm_forward_and_gradients_vs_oracle():
    """use_qk_norm in trunk and decoder: encode / decode outputs and the rec-step gradients (incl. the q / k norm weights) against
    the oracle's autograd; eager == hipGraph segments"""
    from oracle import vtp_oracle as O
    from oracle.ref_stubs import TINY
    from vtp_amd import VTPConfig, VTPModel, VTPTrainer
    torch.manual_seed(11)
    cfg = VTPConfig(**dict(TINY, vision_use_qk_norm=True, decoder_use_qk_norm=True))
    m = VTPModel(cfg)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "q_norm" in n or "k_norm" in n:
                p.copy_(1.0 + 0.3 * torch.randn_like(p))
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(DEV)
    img = torch.randn(2, 3, cfg.image_size, cfg.image_size)

    def oracle(autocast):
        sd = {k: v.clone().requires_grad_(v.dtype == torch.float32) for k, v in sd0.items()}
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            lat = O.reconstruction_latents(sd, img, 2)
            rec = O.decoder_forward(sd, lat.float(), 2)
            loss = O.l1_loss(rec, img)
        loss.backward()
        return sd, lat.detach().float(), rec.detach().float(), float(loss)

    ref, lat_r, rec_r, loss_r = oracle(False)
    noisy, lat_n, rec_n, _ = oracle(True)
    m.eval()
    with torch.no_grad():
        lat = m.get_reconstruction_latents(img.to(DEV))
        rec = m.get_latents_decoded_images(lat)
    e1, e1r = relF(lat, lat_r), relF(lat_n, lat_r)
    e2, e2r = relF(rec, rec_r), relF(rec_n, rec_r)
    print(f"qk-norm latents E_ours={e1:.3e} E_ref={e1r:.3e}; recon E_ours={e2:.3e} E_ref={e2r:.3e}")
    assert e1 <= max(1.5 * e1r, 1e-2) and e2 <= max(1.5 * e2r, 2e-2)
    m.train()
    tr = VTPTrainer(m, lr=0.0, weight_decay=0.0)
    loss = float(tr.step(img.to(DEV))[0])
    assert abs(loss - loss_r) < 3e-3 * loss_r
    params = dict(m.named_parameters())
    for k in ("trunk.blocks.0.attn.q_norm.weight", "trunk.blocks.1.attn.k_norm.weight", "pixel_decoder.blocks.0.attn.k_norm.weight",
              "trunk.blocks.0.attn.qkv.weight", "trunk.patch_embed.proj.weight", "pixel_decoder.blocks.0.attn.qkv.weight"):
        e, er = relF(params[k].grad, ref[k].grad), relF(noisy[k].grad, ref[k].grad)
        print(f"  grad {k}: E_ours={e:.3e} E_ref={er:.3e}")
        assert e <= max(1.5 * er, 3e-2), k
    res = []
    for use_graphs in (False, True):
        m2 = VTPModel(cfg)
        m2.load_state_dict(sd0)
        t2 = VTPTrainer(m2.to(DEV), lr=1e-3, weight_decay=0.0, use_graphs=use_graphs)
        res.append([float(t2.step(img.to(DEV))[0]) for _ in range(4)])
    assert res[0][-1] < res[0][0]
    for a, b in zip(*res):
        assert abs(a - b) < 1e-3 * abs(a)
