"""CPU tests of the host-side logic: config mirror, checkpoint-key contract, flat-parameter ordering, RoPE tables,
gradient bucketing ranges, and the world_size-2 gloo path of the bucketed all-reduce."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_defaults_are_vtp_base_and_validation():
    from vtp_amd.config import VTPConfig, swiglu_hidden
    c = VTPConfig()
    assert (c.vision_embed_dim, c.vision_depth, c.vision_num_heads) == (768, 12, 12)
    assert c.vision_feature_bottleneck == 64 and c.decoder_norm_layer == "layernorm" and c.vision_norm_layer == "rmsnorm"
    assert swiglu_hidden(768) == 2048 and swiglu_hidden(384) == 1024 and swiglu_hidden(1024) == 2736 and swiglu_hidden(128) == 344
    with pytest.raises(ValueError, match="head_dim"):
        VTPConfig(vision_embed_dim=768, vision_num_heads=8)
    with pytest.raises(ValueError, match="vision_clip_feat"):
        VTPConfig(vision_clip_feat="bogus")
    d = c.to_dict()
    assert d["model_type"] == "vtp" and VTPConfig.from_dict(d).to_dict() == d


def test_state_dict_contract_matches_reference_checkpoint(golden_sd):
    """Exact key names, shapes and dtypes of a reference VTPModel checkpoint."""
    from oracle.ref_stubs import TINY
    from vtp_amd import VTPConfig, VTPModel
    m = VTPModel(VTPConfig(**TINY))
    sd = m.state_dict()
    assert list(sd.keys()) != [] and set(sd.keys()) == set(golden_sd.keys())
    for k, v in sd.items():
        assert v.shape == golden_sd[k].shape and v.dtype == golden_sd[k].dtype, k
    m.load_state_dict(golden_sd, strict=True)
    assert torch.equal(m.trunk.rope_embed.periods, golden_sd["trunk.rope_embed.periods"])


def test_reference_checkpoint_directory_loads(tmp_path, golden_sd):
    """A directory written in the reference's HF layout (config.json + model.safetensors) loads unchanged."""
    import json
    from safetensors.torch import save_file
    from oracle.ref_stubs import TINY
    from vtp_amd import VTPModel
    cfg = dict(TINY, model_type="vtp", architectures=["VTPModel"], transformers_version="5.15.0")
    json.dump(cfg, open(tmp_path / "config.json", "w"))
    save_file({k: v.contiguous() for k, v in golden_sd.items()}, str(tmp_path / "model.safetensors"))
    m = VTPModel.from_pretrained(str(tmp_path))
    assert torch.equal(m.state_dict()["pixel_decoder.proj_out.weight"], golden_sd["pixel_decoder.proj_out.weight"])


def test_flat_order_keeps_swiglu_pairs_adjacent():
    from vtp_amd.engine import _flat_order
    names = ["a.norm1.weight", "a.mlp.w1.weight", "a.mlp.w1.bias", "a.mlp.w2.weight", "a.mlp.w2.bias", "a.mlp.w3.weight"]
    assert _flat_order(names) == ["a.norm1.weight", "a.mlp.w1.weight", "a.mlp.w2.weight", "a.mlp.w1.bias", "a.mlp.w2.bias",
                                  "a.mlp.w3.weight"]


def test_rope_tables_match_oracle():
    from oracle import vtp_oracle as O
    from vtp_amd.engine import rope_tables
    per = O.rope_periods(64)
    for H, W in [(16, 16), (6, 10), (32, 32)]:
        s, c = rope_tables(per, H, W, "cpu")
        so, co = O.rope_table(H, W, per)
        assert torch.equal(s, so) and torch.equal(c, co)


def test_range_helpers():
    from vtp_amd.train import merge_ranges, param_ranges, uncovered
    assert merge_ranges([(8, 12), (0, 4), (4, 8), (20, 24)]) == [(0, 12), (20, 24)]
    # EMA pairs the optimizer lane did not touch on a step (train.py, optimizer leg): the parts of a student range outside the updated ranges
    assert uncovered(10, 30, [(0, 12), (20, 24)]) == [(12, 20), (24, 30)]
    assert uncovered(10, 30, [(0, 40)]) == [] and uncovered(10, 30, []) == [(10, 30)] and uncovered(10, 30, [(30, 50), (0, 10)]) == [(10, 30)]
    offs = {"trunk.a": (0, 6), "visual_proj.weight": (8, 4), "pixel_decoder.b": (12, 8), "text.c": (20, 3)}
    assert param_ranges(offs, ("trunk.", "pixel_decoder.")) == [(0, 8), (12, 20)]


def test_rope_augmentation_host_tables_and_config(tmp_path):
    """train-time RoPE augmentations (embeddings.py:155-171): the engine's host tables equal the oracle's (which the reference pins) for
    the same draws, bit for bit; the options travel as extra VTPConfig keys and through the legacy YAML's pos_embed_rope_*_coords"""
    from oracle import vtp_oracle as O
    from vtp_amd import VTPConfig
    from vtp_amd.engine import RopeAugmenter, _rope_host
    per = O.rope_periods()
    aug = RopeAugmenter(per, 3, True, 0.2, 1.3, 1.5, seed=5)
    assert aug.active and not RopeAugmenter(per, 3, True, None, None, None).active
    for hw in ((4, 6), (16, 16)):
        d = aug._draw()
        assert d["shift"].dtype == torch.bfloat16 and float(d["shift"].abs().max()) <= 0.2 and 1 / 1.3 <= float(d["jitter"].min()) and float(d["rescale"]) <= 1.5
        s1, c1 = _rope_host(per, *hw, d)
        s2, c2 = O.rope_table(*hw, per, d)
        assert torch.equal(s1, s2.to(torch.bfloat16)) and torch.equal(c1, c2.to(torch.bfloat16))
    s1, _ = _rope_host(per, 16, 16, None)
    assert torch.equal(s1, O.rope_table(16, 16, per)[0].to(torch.bfloat16))
    c = VTPConfig(vision_rope_shift_coords=0.1, decoder_rope_jitter_coords=1.2)
    assert c.vision_rope_shift_coords == 0.1 and c.vision_rope_jitter_coords is None and c.decoder_rope_jitter_coords == 1.2
    assert VTPConfig.from_dict(c.to_dict()).decoder_rope_jitter_coords == 1.2
    assert not [k for k in VTPConfig().to_dict() if "rope" in k]  # defaults: the reference's HF config keys only
    with pytest.raises(ValueError):
        VTPConfig(vision_rope_jitter_coords=0.5)
    import yaml
    y = {"data": {"image_size": 64},
         "training": {"train_clip": True, "train_reconstruction": True},
         "vtp_model": {"vision_encoder": {"patch_size": 16, "embed_dim": 128, "depth": 2, "num_heads": 2, "mlp_ratio": 4.0, "ffn_layer": "swiglu",
                                          "norm_type": "rmsnorm", "vit_feature_bottleneck": 64, "bottleneck_ae_only": True, "clip_feat": "cls",
                                          "pos_embed_rope_shift_coords": 0.25, "pos_embed_rope_rescale_coords": 2.0},
                       "text_encoder": {"context_length": 16, "vocab_size": 512, "embed_dim": 128, "heads": 2, "layers": 2, "mlp_ratio": 4.0,
                                        "embed_cls": False, "pad_id": 0, "no_causal_mask": False, "pool_type": "argmax", "proj_type": "linear",
                                        "proj_bias": False, "output_tokens": False, "quick_gelu": False},
                       "pixel_decoder": {"embed_dim": 128, "num_heads": 2, "depth": 2, "ffn_layer": "swiglu", "norm_layer": "layernorm",
                                         "pos_embed_rope_jitter_coords": 1.1}}}
    f = tmp_path / "vtp.yaml"
    f.write_text(yaml.safe_dump(y))
    c = VTPConfig.from_vtp_yaml(str(f))
    assert (c.vision_rope_shift_coords, c.vision_rope_jitter_coords, c.vision_rope_rescale_coords) == (0.25, None, 2.0)
    assert c.decoder_rope_jitter_coords == 1.1


def test_optimizer_lane_pieces_and_table_runs():
    """per-bucket optimizer lane (vtp_amd/train.py): a bucket's flat ranges are cut at the borders of the EMA-tracked groups, each piece
    carries the offset of the teacher's copy; ParamStore.desc_runs picks the weight-refresh records whose sources lie in given ranges"""
    from vtp_amd.train import lane_pieces
    # student groups: trunk [0, 100) -> teacher at 1000; head [200, 260) -> teacher at 1100
    pairs = [(1000, 0, 100), (1100, 200, 260)]
    assert lane_pieces([(40, 60)], pairs) == [(40, 60, 1040)]
    assert lane_pieces([(80, 120)], pairs) == [(80, 100, 1080), (100, 120, None)]       # leaves the trunk group
    assert lane_pieces([(150, 300)], pairs) == [(150, 200, None), (200, 260, 1100), (260, 300, None)]
    assert lane_pieces([(0, 100), (200, 260)], []) == [(0, 100, None), (200, 260, None)]  # no teacher: plain AdamW
    covered = lane_pieces([(0, 300)], pairs)
    assert [a for a, _, _ in covered] == [0, 100, 200, 260] and covered[-1][1] == 300
    from vtp_amd.engine import ParamStore
    st = ParamStore.__new__(ParamStore)
    st._desc_src = [0, 16, 48, 64, 200, 232, 1000, 1016]
    assert st.desc_runs([(0, 50)]) == [(0, 3)]
    assert st.desc_runs([(16, 17), (200, 260)]) == [(1, 2), (4, 6)]
    assert st.desc_runs([(40, 48)]) == []
    assert st.desc_runs([(0, 50)]) + st.desc_runs([(1000, 1050)]) == [(0, 3), (6, 8)]


def test_bucket_waits_can_be_partial():
    """GradBucketer.wait(works) takes in a given list (the previous bucket event's reductions) and leaves the rest outstanding"""
    from vtp_amd.train import GradBucketer

    class W:
        def __init__(self):
            self.done = 0

        def wait(self):
            self.done += 1
    gb = GradBucketer.__new__(GradBucketer)
    a, b = W(), W()
    hit = []
    gb.works = [(b, None)]
    gb.wait([(a, lambda: hit.append(1))])
    assert a.done == 1 and b.done == 0 and hit == [1] and len(gb.works) == 1
    gb.wait()
    assert b.done == 1 and gb.works == []


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from vtp_amd.train import GradBucketer
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    flat = torch.arange(40, dtype=torch.float32) * (rank + 1)
    gb = GradBucketer(flat)
    gb.reduce_range(4, 16)   # bucket launched "during backward"
    gb.reduce_range(24, 40)
    gb.wait()
    out[rank] = flat.clone()
    dist.destroy_process_group()


def test_bucketed_allreduce_world2_gloo():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    base = torch.arange(40, dtype=torch.float32)
    for r in range(world):
        exp = base * (r + 1)
        exp[4:16] = base[4:16] * 3   # sum over ranks of (rank+1)
        exp[24:40] = base[24:40] * 3
        assert torch.equal(out[r], exp)


def _shard_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from vtp_amd.train import GradBucketer
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    n = 100
    flat_g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    flat_p = torch.zeros(n)
    gb = GradBucketer(flat_g, shard=True)
    ranges = [(4, 44), (60, 100)]          # 40 elements each: chunks of 16 (8-aligned) -> the last chunk is partly padding
    for lo, hi in ranges:
        gb.reduce_scatter_range(lo, hi)
    gb.wait()
    recs = [gb._rec(lo, hi) for lo, hi in ranges]
    own = []
    for rec in recs:                        # "optimizer" on the own chunk: p = -g_sum
        k = rec.b - rec.a
        own.append((rec.a, rec.b))
        flat_p[rec.a:rec.b] = -rec.g32[:k]
        rec.p_send.zero_()
        rec.p_send[:k] = flat_p[rec.a:rec.b]
    gb.all_gather_params(flat_p, recs)
    for rec in recs:
        flat_p[rec.lo:rec.hi] = rec.p_recv[:rec.hi - rec.lo]
    out[rank] = (flat_p.clone(), own, [r.chunk for r in recs])
    dist.destroy_process_group()


def test_sharded_bucket_exchange_world2_gloo():
    """reduce-scatter + own-chunk update + all-gather of the sharded optimizer (train.GradBucketer) on CPU tensors: every rank ends
    with the full updated buffer, ownership is disjoint and covers each bucket, chunks are 8-element aligned"""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_shard_worker, args=(world, port, out), nprocs=world, join=True)
    base = torch.arange(100, dtype=torch.float32)
    exp = torch.zeros(100)
    for lo, hi in ((4, 44), (60, 100)):
        exp[lo:hi] = -3 * base[lo:hi]
    for r in range(world):
        assert torch.equal(out[r][0], exp), r
        assert all(c % 8 == 0 for c in out[r][2])
    for i, (lo, hi) in enumerate(((4, 44), (60, 100))):
        (a0, b0), (a1, b1) = out[0][1][i], out[1][1][i]
        assert a0 == lo and b0 == a1 and b1 == hi  # disjoint cover in rank order


def test_ssl_index_plan_matches_reference_buffer_layout():
    """build_ssl_indices (host): teacher cls rows are view-swapped (vtp.py:425-426), masked patch rows address the
    [B, 1+hw, D] stream, per-image iBOT weights are 1 / n_masked(image), padding rows are inert."""
    import numpy as np
    from vtp_amd.ssl_engine import build_ssl_indices
    B, hw, n_local, hw_l = 3, 16, 2, 4
    rng = np.random.default_rng(0)
    masks = np.zeros((2 * B, hw), bool)
    masks[0, [1, 5]] = True
    masks[4, [0, 2, 3]] = True
    p = build_ssl_indices(masks, B, hw, n_local, hw_l, dino_weight=1.0, ibot_weight=2.0, pad_to=8)
    N = hw + 1
    assert p["n_masked"] == 5 and p["Tm"] == 8 and p["Ts"] == n_local * B + 2 * B + 8
    assert p["teacher_src"][:2 * B].tolist() == [b * N for b in (3, 4, 5, 0, 1, 2)]
    assert p["teacher_src"][2 * B:2 * B + 5].tolist() == [0 * N + 2, 0 * N + 6, 4 * N + 1, 4 * N + 3, 4 * N + 4]
    assert (p["teacher_src"][2 * B + 5:] == -1).all()
    assert p["student_local_src"].tolist() == [i * (hw_l + 1) for i in range(n_local * B)]
    assert p["student_global_src"][:2 * B].tolist() == [b * N for b in range(2 * B)]
    terms = 2 * 1 + n_local * 2
    w = p["w"]
    assert np.allclose(w[:n_local * B + 2 * B], 1.0 / (B * terms))
    m0 = n_local * B + 2 * B
    assert np.allclose(w[m0:m0 + 5], [2.0 / 2 / B] * 2 + [2.0 / 3 / B] * 3) and (w[m0 + 5:] == 0).all()
    # local crops are compared with both teacher views, global crops with the other view only
    assert p["t0"][:n_local * B].tolist() == [0, 1, 2, 0, 1, 2] and p["t1"][:n_local * B].tolist() == [3, 4, 5, 3, 4, 5]
    assert (p["t1"][n_local * B:] == -1).all()


def test_wgrad_group_split_rule_and_overlap_lanes():
    """ops.WgradGroup picks the slice count host-side (no GPU needed up to finalize's allocations): tiles x slices fills one round of
    the 256 CUs with every slice >= 16 k-tiles; the library's per-layer rule (vtp_gemm_tn_splits) stays within its kernels'
    operating points"""
    from vtp_amd import _lib
    from vtp_amd.engine import OVERLAP
    lib = _lib.load()
    for rows, cols, k in ((2304, 768, 34144), (768, 768, 34144), (4096, 768, 8224), (768, 2048, 34144), (65536, 256, 3000), (64, 64, 100)):
        s = lib.vtp_gemm_tn_splits(rows, cols, k)
        t256, t128 = -(-rows // 256) * -(-cols // 256), -(-rows // 128) * -(-cols // 128)
        assert s >= 1 and (s == 1 or t256 * s <= 256 or (t128 * s <= 512 and s <= 16)), (rows, cols, k, s)
    from vtp_amd import ops
    for ktok, want in ((34144, 2), (8192, 2), (2464, 2), (514, 1), (1100, 1)):  # VTP-B block: 108 tiles of 256 x 256
        ks, s = ops.wgrad_group_splits(108, ktok)  # the helper WgradGroup.finalize calls (test the code, not a copy of it)
        assert s == want and ks % 64 == 0 and ks * s >= ktok, (ktok, ks, s)
    # 32-bit staging offsets of the grouped launch: VTP-L at 512^2, 16 img/GPU (M = 16 x 1025 x ..., ld = 2H = 5472) fits; a token count
    # x leading dimension beyond 4 GiB does not -- Stack.backward then takes the per-layer path, WgradGroup.finalize refuses
    assert ops.wgrad_group_fits(34144, 4096) and ops.wgrad_group_fits(16 * 1025 * 3 + 128 * 37, 5472)
    assert not ops.wgrad_group_fits(400000, 5472)
    g = ops.WgradGroup(400000)
    g.rows.append([0, 0, 0, 0, 5472, 1024, 1024, 5472, 1024, 0, 0, 0, 1, 0, 0, 0])
    g.ntiles = 88
    with pytest.raises(ValueError, match="32-bit"):
        g.finalize("cpu")
    assert OVERLAP._lane == 0
    with OVERLAP.lane(1):
        assert OVERLAP._lane == 1
        with OVERLAP.lane(0):
            assert OVERLAP._lane == 0
        assert OVERLAP._lane == 1
    assert OVERLAP._lane == 0


def test_lpips_module_has_the_reference_state_dict_layout_and_no_cpu_fallback():
    from oracle import lpips_oracle as L
    from vtp_amd import LPIPS
    m = LPIPS(use_dropout=True)
    sd = L.make_state(1)
    assert set(m.state_dict().keys()) == set(sd.keys())
    assert all(tuple(m.state_dict()[k].shape) == tuple(v.shape) for k, v in sd.items())
    m.load_state_dict(sd, strict=True)
    assert abs(LPIPS.forward_gflop(224, 224) - 30.7) < 0.2  # 15.35 GMAC for VGG16 features at 224^2
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 32, 32), torch.zeros(1, 3, 32, 32))
    assert set(LPIPS(use_dropout=False).state_dict().keys()) == {k.replace(".model.1.", ".model.0.") for k in sd}


def test_block_mask_generator_and_ssl_collate():
    """vtp_amd.data: block-wise iBOT masks + the collate that yields the reference's ssl_dict keys (vtp.py:365-374)."""
    import numpy as np
    import torch
    from vtp_amd.data import BlockMaskGenerator, collate_ssl_batch, collate_ssl_masks
    rng = np.random.default_rng(0)
    gen = BlockMaskGenerator((16, 16), max_num_patches=128, rng=rng)
    for n in (0, 5, 40, 128):
        m = gen(n)
        assert m.shape == (16, 16) and m.dtype == bool and int(m.sum()) <= n
        if n >= 40:
            assert int(m.sum()) >= n // 2  # blocks are added until the budget is (nearly) used
    tot = []
    for _ in range(20):
        d = collate_ssl_masks(64, (16, 16), 0.5, (0.1, 0.5), rng)
        masks = d["masks"]
        assert masks.shape == (64, 256) and int((masks.sum(1) > 0).sum()) <= 32
        nm = int(d["n_masked_patches"])
        assert nm == int(masks.sum()) == d["mask_indices_list"].numel() <= d["upperbound"]
        assert torch.equal(d["mask_indices_list"], masks.flatten().nonzero().flatten())
        per = masks.sum(1).clamp(min=1).float()
        assert torch.allclose(d["masks_weight"], (1.0 / per)[d["mask_indices_list"] // 256])
        tot.append(nm)
    assert len({collate_ssl_masks(64, (16, 16), 0.5, (0.1, 0.5), rng)["upperbound"] for _ in range(5)}) == 1  # fixed buffer size
    assert 0.2 * 64 * 256 * 0.5 < sum(tot) / len(tot) < 0.4 * 64 * 256 * 0.5 * 1.2
    B = 3
    out = collate_ssl_batch([torch.randn(B, 3, 64, 64) for _ in range(2)], [torch.randn(B, 3, 32, 32) for _ in range(4)], rng=rng)
    assert set(out) >= {"global_crops", "n_global_crops", "mask_indices_list", "n_masked_patches", "upperbound", "local_crops", "masks"}
    assert out["global_crops"].shape == (2 * B, 3, 64, 64) and out["local_crops"].shape == (4 * B, 3, 32, 32) and out["n_global_crops"] == 2
    # the SSL index plan honours the collate's upperbound (fixed-size masked-token buffers)
    from vtp_amd.ssl_engine import build_ssl_indices
    p = build_ssl_indices(out["masks"].numpy(), B, 16, 4, 4, 1.0, 1.0, upperbound=out["upperbound"])
    assert p["Tm"] == max(64, (out["upperbound"] + 63) // 64 * 64) and p["n_masked"] == int(out["n_masked_patches"])


def test_torch_library_registration():
    """torch.ops.vtp_hip.* exist with schemas (CPU: registration + schema only; the GPU tests call them)."""
    import torch
    import vtp_amd.torch_ops as t
    assert {"gemm_nt", "gemm_tn", "gemm_qkv_rope", "norm_fwd", "norm_bwd", "attn_fwd", "attn_bwd", "rope_qk", "adamw"} <= set(t.OPS)
    for name in t.OPS:
        op = getattr(torch.ops.vtp_hip, name)
        assert "vtp_hip::" + name in str(op.default._schema)
    with pytest.raises((NotImplementedError, RuntimeError)):  # no CPU kernel: loud, never a silent fallback
        torch.ops.vtp_hip.ema(torch.zeros(4), torch.zeros(4), 4, 0.5)


def test_drop_allocation_matches_reference_rule():
    """Stack.drop_allocation = get_branges_scales' allocation (block.py:27-33 single process, :44-62 under DDP)."""
    from vtp_amd.engine import Stack
    assert Stack.drop_allocation(5, 0.4) == (3, 5 / 3)
    assert Stack.drop_allocation(4, 0.99) == (1, 4.0)
    for b, r, W in ((32, 0.1, 8), (5, 0.4, 3), (7, 0.5, 4), (2, 0.9, 8)):
        gb = b * W
        gkeep = max(int(gb * (1 - r)), W)
        base, extra = gkeep // W, gkeep % W
        alloc = [min(base + (1 if i < extra else 0), b) for i in range(W)]
        for rank in range(W):
            k, sc = Stack.drop_allocation(b, r, W, rank)
            assert k == alloc[rank] and abs(sc - gb / sum(alloc)) < 1e-12


# ------------------------------------------------------------------------------------------------ f3: tokenizer host logic
def test_distributed_indices_match_torch_sampler():
    """extract_features_vtp.py:59-62 uses DistributedSampler(shuffle=False): same rank slices, wrap-around padding included"""
    from torch.utils.data.distributed import DistributedSampler
    from vtp_amd.tokenizer import distributed_indices
    for n in (1, 2, 7, 8, 13, 100):
        for world in (1, 2, 3, 8):
            for rank in range(world):
                ref = list(DistributedSampler(range(n), num_replicas=world, rank=rank, shuffle=False))
                assert distributed_indices(n, world, rank) == ref, (n, world, rank)


def test_latent_shard_writer_layout(tmp_path):
    """file names, keys, metadata and the 10000 // batch_size flush rule of extract_features_vtp.py:88-118"""
    from safetensors import safe_open
    from safetensors.torch import load_file
    from vtp_amd.tokenizer import LatentShardWriter, shard_name
    w = LatentShardWriter(str(tmp_path), rank=3, batch_size=4000)
    assert w.batches_per_shard == 2
    for i in range(3):
        w.add(torch.full((2, 4, 2, 2), float(i)), torch.full((2, 4, 2, 2), -float(i)), torch.tensor([i, i + 10]))
    assert w.saved_files == 1
    last = w.close()
    assert os.path.basename(last) == "latents_rank03_shard001.safetensors" == shard_name(3, 1)
    assert w.close() is None
    a = load_file(os.path.join(str(tmp_path), shard_name(3, 0)))
    assert a["latents"].shape == (4, 4, 2, 2) and a["labels"].tolist() == [0, 10, 1, 11]
    assert torch.equal(a["latents_flip"], -a["latents"])
    with safe_open(last, "pt") as f:
        assert f.metadata() == {"total_size": "2", "dtype": "torch.float32", "device": "cpu"}
    with pytest.raises(ValueError):
        LatentShardWriter(str(tmp_path), batch_size=20000)


def test_bench_flop_accounting_matches_survey():
    """bench.py's algorithmic-FLOP helpers against the per-image forward GFLOP table of SURVEY.md §8(d) / BASELINE.md §4"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_bench", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    from vtp_amd.config import swiglu_hidden
    hw = 256
    rows = {"S": (384, 12, 12.30, 12.24, 3.38), "B": (768, 12, 46.42, 46.23, 13.30), "L": (1024, 24, 162.35, 161.70, 47.09)}
    for name, (D, L, enc, dec, txt) in rows.items():
        H = swiglu_hidden(D)
        assert abs(b.vit_fwd_gflop(D, H, L, hw + 1, hw, True) - enc) < 0.02 * enc, name
        assert abs(b.vit_fwd_gflop(D, H, L, hw, hw, False) - dec) < 0.02 * dec, name
        assert abs(b.text_fwd_gflop(D, L, 77) - txt) < 0.02 * txt, name
    assert abs(b.vit_fwd_gflop(1024, swiglu_hidden(1024), 24, 1025, 1024, True) - 724.91) < 0.02 * 724.91
    assert abs(3 * (46.42 + 46.23) - 277.9) < 0.1  # the rec-only train-step figure the bench's step_frac is built on
    assert set(b.WORKLOADS) >= {"vtp_base_full", "vtp_small_rec", "vtp_large_full_512"} and "vtp_large_fp8_fwd" in b.FORWARD_WORKLOADS


def test_gemm_dispatch_table_of_the_step():
    """vtp_gemm_nt_config (host-only): the measured kernel choice for the GEMM shapes of the VTP-B step -- 8 = 256x256 8-phase
    kernel, 9 = 128x256 half-size kernel with two workgroups per CU (round 4), 10 = 256x256 one-wave-per-SIMD kernel with the
    hand-scheduled k loop (round 4: plain bf16 epilogue, K >= 2048, >= 192 tiles), 7 = 128x64 ring tiles, 5 / 0 = 128x128 ring; slices =
    in-launch split-K.  Pinned because every row is a measurement (profiles/r03_gemm8p_bench.log, profiles/r04_gemm8h_bench.log, profiles/r04_gemm4w_bench.log,
    tools/text_gemm_ab.py, tools/proto_gemm_ab.py) that an edit of the heuristics can silently undo."""
    from vtp_amd import _lib, ops
    lib = _lib.load()
    BF, F32, SW, GELU = ops.EPI_BF16, ops.EPI_F32, ops.EPI_SWIGLU, ops.EPI_GELU
    want = {
        # list forward / backward of the trunk (34144 rows), teacher (16448), pixel decoder (8192)
        (34144, 2304, 768, BF): (8, 0), (34144, 768, 768, F32): (8, 0), (34144, 4096, 768, SW): (8, 0), (34144, 768, 2048, F32): (8, 0),
        (34144, 768, 4096, BF): (10, 0), (34144, 768, 2304, BF): (10, 0), (16448, 768, 4096, BF): (10, 0), (34144, 768, 2048, BF): (10, 0),
        (34144, 2048, 768, BF): (8, 0), (16448, 768, 768, F32): (8, 0), (16448, 4096, 768, SW): (9, 0),
        (8192, 2304, 768, BF): (8, 0), (8192, 4096, 768, SW): (8, 0), (8192, 768, 4096, BF): (8, 2),
        (8192, 768, 2048, F32): (7, 0), (8192, 768, 768, F32): (7, 0), (8192, 768, 2304, BF): (9, 0), (8192, 768, 768, BF): (9, 0),
        (2464, 3072, 768, BF): (9, 0),
        # text tower (32 x 77 rows) and DINO head
        (2464, 768, 3072, F32): (7, 0), (2464, 768, 3072, BF): (7, 0), (2464, 2304, 768, BF): (7, 0), (2464, 3072, 768, GELU): (5, 0),
        (2816, 65536, 256, BF): (8, 0), (2816, 2048, 2048, GELU): (5, 0), (2816, 256, 2048, F32): (7, 0),
    }
    for shape, (cfg, slices) in want.items():
        got = lib.vtp_gemm_nt_config(*shape)
        assert (got & 255, got >> 8) == (cfg, slices), (shape, got & 255, got >> 8)
    assert lib.vtp_gemm_nt_config(0, 768, 768, BF) == -1


def test_bench_self_launch_builds_the_torchrun_command(monkeypatch):
    """`python bench.py --gpus N` with no launcher in the environment must re-run itself under torch.distributed.run with N ranks
    on 127.0.0.1 (the driver's 8-GPU command is the plain one); with WORLD_SIZE set it must not."""
    import importlib.util
    import subprocess
    spec = importlib.util.spec_from_file_location("_bench_sl", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("OMP_NUM_THREADS", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as ei:
        b.main()
    assert ei.value.code == 7  # the launcher's exit code is passed through
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 1024
    assert cmd[-6:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["OMP_NUM_THREADS"] == str(max(1, (os.cpu_count() or 8) // 8))
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_grouped_wgrad_kernel_choice(monkeypatch):
    """ops.wgrad_group_kernel: the one-wave-per-SIMD kernel (gemm4w_tn.hip) needs token counts that are multiples of 8 (its LDS-DMA pieces
    are 8 k rows, zero-filled as a whole beyond the slice) and K slices of >= 4096 rows (profiles/r04_wgrad_kernel_ab.log); the
    environment switch of the same-box step A/B overrides the measured rule but never the multiple-of-8 requirement"""
    from vtp_amd import ops
    monkeypatch.delenv("VTP_GEMM4W_TN", raising=False)
    assert ops.wgrad_group_kernel(108, 2, 34144) == 1 and ops.wgrad_group_kernel(196, 1, 8192) == 1 and ops.wgrad_group_kernel(108, 2, 8192) == 1
    assert ops.wgrad_group_kernel(108, 2, 2464) == 0      # 1232-row slices: the 8-phase kernel
    assert ops.wgrad_group_kernel(108, 2, 2134) == 0      # 2134 % 8 != 0
    assert ops.wgrad_group_kernel(24, 4, 2048) == 0       # 512-row slices
    monkeypatch.setenv("VTP_GEMM4W_TN", "0")
    assert ops.wgrad_group_kernel(108, 2, 34144) == 0
    monkeypatch.setenv("VTP_GEMM4W_TN", "1")
    assert ops.wgrad_group_kernel(24, 4, 2048) == 1 and ops.wgrad_group_kernel(108, 2, 2134) == 0


def test_generated_k_loops_are_up_to_date(tmp_path, monkeypatch):
    """vtp_amd/csrc/gemm4w_ktile.inc / gemm4w_tn_ktile.inc are GENERATED (tools/gen_gemm4w_ktile.py) and committed: the committed text
    must be what the generator writes (an edit of the schedule that forgets to regenerate -- or a hand edit of the .inc -- fails here)"""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen4w", os.path.join(root, "tools", "gen_gemm4w_ktile.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    monkeypatch.setattr(gen, "OUT", str(tmp_path / "gemm4w_ktile.inc"))
    gen.main()
    for name in ("gemm4w_ktile.inc", "gemm4w_tn_ktile.inc"):
        assert (tmp_path / name).read_text() == open(os.path.join(root, "vtp_amd", "csrc", name)).read(), name
    body = gen.body()
    assert sum("v_mfma" in l for l in body) == 128 and sum("global_load_lds" in l for l in body) == 32 and sum("s_barrier" in l for l in body) == 4
    tn = gen.tn_body(True)
    assert sum("v_mfma" in l for l in tn) == 128 and sum("ds_read_b64_tr_b16" in l for l in tn) == 48 + 128 and sum("v_dot2" in l for l in tn) == 128


def test_trace_gaps_union_and_gap_accounting(tmp_path, capsys, monkeypatch):
    """tools/trace_gaps.py on a hand-made kernel trace: two streams whose kernels overlap, a 30-us and a 5-us hole per step; the marker
    kernel delimits the steps, the union of the intervals (not their sum) is the busy time."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("trace_gaps", os.path.join(root, "tools", "trace_gaps.py"))
    tg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tg)
    rows = ["Start_Timestamp,End_Timestamp,Kernel_Name"]
    for s in range(5):  # 5 steps of 1000 us
        t = 1_000_000 * s
        rows += [f"{t},{t + 100_000},marker_kernel",                 # 0 .. 100
                 f"{t + 50_000},{t + 400_000},gemm_a",                # 50 .. 400 (overlaps the marker)
                 f"{t + 430_000},{t + 700_000},gemm_b",               # hole of 30 us in front
                 f"{t + 600_000},{t + 900_000},side_kernel",          # overlaps gemm_b
                 f"{t + 905_000},{t + 1_000_000},tail_kernel"]        # hole of 5 us in front
    f = tmp_path / "k.csv"
    f.write_text("\n".join(rows) + "\n")
    monkeypatch.setattr(sys, "argv", ["trace_gaps.py", str(f), "--marker", "marker_kernel", "--skip", "1"])
    detail, nsteps, wall = tg.main()
    out = capsys.readouterr().out
    assert nsteps == 3 and wall == 3_000_000
    assert sorted(x for x, *_ in detail) == [5_000] * 3 + [30_000] * 3
    assert "idle/step 0.035 ms" in out and "busy(union)/step 0.965 ms" in out
    assert "sum of kernel durations/step 1.115 ms" in out  # 100 + 350 + 270 + 300 + 95 us: more than the wall time, streams overlap
    assert "idle before gemm_b" in out


def test_shipped_kernels_spill_ratchet():
    """Register spills of the SHIPPED device code (vtp_amd/lib/*.o -> .hip_fatbin -> gfx950 code object -> AMDGPU metadata, read by
    tools/spill_report.py --built: no compile).  A ratchet, not a wish: the kernels whose k loops must stay clean are pinned to zero,
    the known epilogue-only spills to their current counts, and nothing in the library may spill more than one accumulator tile (64) --
    the 462-register spill of the 256 x 256 bf16 ring fallback is gone."""
    import importlib.util
    import __graft_entry__ as ge
    ge.build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("spill_report", os.path.join(root, "tools", "spill_report.py"))
    sr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sr)
    rows = {}
    for o in sr.built_objects():
        for name, vs, ss, scr, vg, ag in sr.report_built(o):
            rows[name] = (vs, scr, vg, ag)
    assert len(rows) > 150, f"only {len(rows)} kernels found in the built objects"
    worst = max(rows.items(), key=lambda kv: kv[1][0])
    assert worst[1][0] <= 64, f"{worst[0]} spills {worst[1][0]} VGPRs"
    # (the third template argument of the half-size / one-wave kernels: false = static tile lists, the one-GPU default; true = tiles
    # drawn from the queues, round 6)
    zero = ["vtp::gemm8h_kernel<0, 0, false>", "vtp::gemm8h_kernel<0, 1, false>", "vtp::gemm8h_kernel<0, 2, false>", "vtp::gemm8h_kernel<2, 0, false>",
            "vtp::gemm4w_kernel<0, 0, false>", "vtp::attn_bwd_fused_kernel<8>", "vtp::attn_bwd_fused_kernel<2>", "vtp::attn_fwd_res2_kernel",
            "vtp::gemm_nt_kernel<128, 64, 4, 1, 3, 0, false, false>", "vtp::gemm_nt_kernel<128, 64, 4, 1, 3, 1, false, false>",
            "vtp::adamw_ema_kernel", "vtp::prep_weights_kernel"]
    for k in zero:
        hit = [n for n in rows if k in n]
        assert hit, f"kernel {k} not in the built objects"
        for n in hit:
            assert rows[n][0] == 0, f"{n}: {rows[n][0]} spilled VGPRs (was 0)"
    # epilogue-only spills of the 8-phase family and the one-wave kernels (tools/spill_report.py shows where): may shrink, not grow
    caps = {"vtp::gemm8p_kernel<0, false, 0, 0>": 4, "vtp::gemm8p_kernel<1, false, 0, 0>": 1, "vtp::gemm8p_kernel<2, false, 0, 0>": 4,
            "vtp::gemm8p_kernel<0, false, 0, 1>": 33, "vtp::gemm8p_kernel<0, false, 0, 2>": 8, "vtp::gemm4w_grouped_tn_kernel": 48,
            "vtp::gemm4w_grouped_tn_items_kernel": 48, "vtp::gemm8p_grouped_tn_kernel": 4, "vtp::gemm4w_kernel<1, 0, false>": 18,
            # the queue-drawing variants (on beside collectives only): epilogue-side spills, none inside a k loop
            "vtp::gemm8p_dyn_kernel<0, 0, 0>": 4, "vtp::gemm8p_dyn_kernel<1, 0, 0>": 12, "vtp::gemm8p_dyn_kernel<2, 0, 0>": 4,
            "vtp::gemm8h_kernel<0, 0, true>": 4, "vtp::gemm8h_kernel<0, 1, true>": 4, "vtp::gemm8h_kernel<0, 2, true>": 8,
            "vtp::gemm4w_kernel<0, 0, true>": 4}
    for k, cap in caps.items():
        hit = [n for n in rows if k in n]
        assert hit, f"kernel {k} not in the built objects"
        for n in hit:
            assert rows[n][0] <= cap, f"{n}: {rows[n][0]} spilled VGPRs (cap {cap})"


def test_wgrad_group_item_list_partitions_every_tile():
    """ops.wgrad_group_items (the uneven cut of the grouped weight-gradient launch): every tile's items partition [0, K) at multiples of
    64, slots are numbered 0 .. nparts - 1, the bias-gradient tiles (first tile column of a problem with a colsum target) get one slice
    more than the rest while the launch still fits one round of the CUs, and never otherwise"""
    from vtp_amd.ops import wgrad_group_items, wgrad_group_splits

    def rows(D, H):  # the four problems of a ViT block as WgradGroup.add records them: w3, w12 (+bias), proj, qkv (+bias)
        out, t0 = [], 0
        for N, K, gb in ((D, H, 0), (2 * H, D, 1), (D, D, 0), (3 * D, D, 1)):
            out.append([1, 2, 3, gb, N, K, K, N, K, 0, 0, t0, 1, 0, 0, 0])
            t0 += ((N + 255) // 256) * ((K + 255) // 256)
        return out, t0

    for D, H, Ktok in ((768, 2048, 34144), (768, 2048, 8192), (1024, 2736, 34144), (384, 1024, 65792), (768, 2048, 2048)):
        rs, ntiles = rows(D, H)
        _, base = wgrad_group_splits(ntiles, Ktok)
        items, slots = wgrad_group_items(rs, Ktok, base)
        by_tile = {}
        for tile, kb, kc, n, z, *_ in items:
            assert kb % 64 == 0 and kc > 0 and 0 <= z < n <= slots
            by_tile.setdefault(tile, []).append((z, kb, kc, n))
        assert sorted(by_tile) == list(range(ntiles))
        heavy = 0
        for tile, its in by_tile.items():
            its.sort()
            assert [z for z, *_ in its] == list(range(its[0][3])) and its[0][1] == 0
            assert all(a[1] + a[2] == b[1] for a, b in zip(its, its[1:])) and its[-1][1] + its[-1][2] == Ktok
            heavy += its[0][3] > base
        ncs = sum(((r[7] + 255) // 256) for r in rs if r[3])
        fits = ntiles * base + ncs <= 256 and Ktok // (base + 1) >= 1024
        assert heavy == (ncs if fits else 0) and len(items) <= max(256, ntiles * base), (D, Ktok, heavy, ncs, len(items))
        if (D, Ktok) == (768, 34144):
            assert (base, slots, len(items)) == (2, 3, 241)
            # planned for fewer CUs (VTP_GEMM_CUS = 240 beside RCCL's channels): 241 workgroups no longer fit one round -> the uniform cut
            items2, slots2 = wgrad_group_items(rs, Ktok, base, cus=240)
            assert (slots2, len(items2)) == (2, 216)


def test_overlap_defer_issues_in_order_behind_the_next_main_kernel(monkeypatch):
    """engine.Overlap.defer: forked side work is queued and issued -- in fork order, exactly once -- by the next run_deferred() (what
    Stack.backward calls behind a block's first dgrad GEMM) or by a join; VTP_FORK_LATE=0 issues at the fork"""
    from vtp_amd import engine
    ov = engine.Overlap()
    log = []
    monkeypatch.setattr(engine, "FORK_LATE", True)
    ov.defer(lambda: log.append("lane"))
    ov.defer(lambda: log.append("wgrad"))
    assert log == []
    log.append("dgrad")          # the main stream's kernel goes first
    ov.run_deferred()
    assert log == ["dgrad", "lane", "wgrad"]
    ov.run_deferred()
    assert log == ["dgrad", "lane", "wgrad"]   # nothing is issued twice
    ov.defer(lambda: log.append("late"))
    ov.join()                    # a join may not leave a fork un-issued (no side stream yet: nothing to wait for)
    assert log[-1] == "late" and ov._deferred == []
    monkeypatch.setattr(engine, "FORK_LATE", False)
    ov.defer(lambda: log.append("now"))
    assert log[-1] == "now" and ov._deferred == []
