"""The data-parallel exchange on its REAL backend: `init_process_group("nccl")` = RCCL, one rank (the test
boxes have one GPU; two ranks on one device are rejected by RCCL), every collective forced on (`force_collectives=True`: a
one-rank all-reduce / reduce-scatter / all-gather is an identity, but it is issued, runs on RCCL's stream, and is waited for
exactly like the 8-GPU one).  Covered: bucketed async all-reduce during backward, reduce_scatter_tensor + rank-sharded AdamW +
all_gather_into_tensor (fp32 and bf16 buckets), the contrastive feature all-gather / gradient reduce-scatter, the DINO / iBOT
centre all-reduce, eager and hipGraph segments (thread-local stream capture next to the RCCL watchdog thread) -- each must
reproduce the plain single-process step (reference behaviour: torch DDP over NCCL, tools/test_reconstruction_hf.py:220,250)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data():
    g = torch.Generator().manual_seed(5)
    img = torch.randn(4, 3, 64, 64, generator=g)
    txt = torch.randint(1, 500, (4, 16), generator=g)
    txt[:, 0] = 510
    txt[torch.arange(4), torch.tensor([5, 9, 12, 15])] = 511
    return img, txt


def _run_rec_clip(dist_on, use_graphs, shard, grad_dtype):
    from oracle.ref_stubs import TINY
    from safetensors.torch import load_file
    from vtp_amd import VTPConfig, VTPModel, VTPTrainer
    g = load_file(os.path.join(ROOT, "tests", "golden", "vtp_tiny.safetensors"))
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    m = VTPModel(VTPConfig(**TINY))
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda:0")
    tr = VTPTrainer(m, lr=1e-3, weight_decay=0.01, use_graphs=use_graphs, bucket_blocks=1, shard_optimizer=shard and dist_on,
                    grad_dtype=grad_dtype if (shard and dist_on) else "fp32", force_collectives=dist_on)
    tr.time_comm = dist_on
    img, txt = _data()
    losses = []
    for i in range(3):
        r, c = tr.step((img + 0.01 * i).cuda(), txt.cuda())
        losses.append((float(r), float(c)))
    torch.cuda.synchronize()
    info = dict(bytes=tr.bucketer.comm_bytes, exposed=tr.comm_exposed_ms() if dist_on else 0.0)
    return losses, m._engine().flat_p.detach().cpu().clone(), info


def _run_ssl(dist_on, use_graphs, centering):
    from safetensors.torch import load_file
    from oracle.make_golden_ssl import SSL_CFG as C
    from vtp_amd import VTP, VTPConfig, VTPTrainer
    g = load_file(os.path.join(ROOT, "tests", "golden", "vtp_tiny_ssl.safetensors"))
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    torch.manual_seed(0)
    cfg = VTPConfig(image_size=C["R"], vision_embed_dim=C["embed_dim"], vision_depth=C["depth"], vision_num_heads=C["heads"],
                    text_embed_dim=128, text_depth=1, text_num_heads=2, text_vocab_size=64, text_context_length=8,
                    decoder_embed_dim=128, decoder_depth=1, decoder_num_heads=2)
    m = VTP(cfg, dino_out_dim=C["K"], dino_hidden_dim=C["hidden"], dino_bottleneck_dim=C["bott"])
    m.load_state_dict(sd, strict=False)
    m = m.to("cuda:0")
    tr = VTPTrainer(m, lr=5e-4, weight_decay=0.0, use_graphs=use_graphs, centering=centering, force_collectives=dist_on)
    ssl = tr.prepare_ssl(g["in.global_crops"].cuda(), g["in.local_crops"].cuda(), g["in.masks"].bool())
    img = torch.randn(C["B"], 3, C["R"], C["R"], device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    txt = torch.randint(1, 60, (C["B"], 8), device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))
    txt[:, 5] = 63
    hist = []
    for _ in range(3):
        r, c = tr.step(img, txt, ssl)
        hist.append((float(r), float(c), float(tr.ssl_loss_sum)))
    torch.cuda.synchronize()
    return hist, m._engine().flat_p.detach().cpu().clone(), tr.center_dino.cpu().clone()


def _worker(rank, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    assert dist.get_backend() == "nccl"
    res = {}
    for key in (("allreduce", False, False, "fp32"), ("allreduce", True, False, "fp32"), ("shard", False, True, "fp32"),
                ("shard", True, True, "fp32"), ("shard", True, True, "bf16")):
        _, use_graphs, shard, gd = key
        res[key] = _run_rec_clip(True, use_graphs, shard, gd)
    for key in (("ssl", False, "softmax"), ("ssl", True, "softmax"), ("ssl", True, "sinkhorn_knopp")):
        res[key] = _run_ssl(True, key[1], key[2])
    out["res"] = res
    dist.destroy_process_group()


def test_rccl_single_rank_all_paths_match_plain_step():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ref = _run_rec_clip(False, False, False, "fp32")
    ref_ssl = {c: _run_ssl(False, False, c) for c in ("softmax", "sinkhorn_knopp")}
    torch.cuda.empty_cache()
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(_free_port(), out), nprocs=1, join=True)
    res = out["res"]
    for key, (losses, p, info) in ((k, v) for k, v in res.items() if k[0] != "ssl"):
        rel = float((p - ref[1]).norm() / ref[1].norm())
        print(f"RCCL world-1 {key}: losses {losses[-1]} (plain {ref[0][-1]}), weights rel {rel:.2e}, comm bytes/3 steps {info['bytes']}, "
              f"main-stream wait for RCCL {info['exposed']:.3f} ms")
        assert info["bytes"] > 0, "no collective was issued"
        for a, b in zip(losses, ref[0]):
            assert abs(a[0] - b[0]) < 2e-3 * abs(b[0]) and abs(a[1] - b[1]) < 5e-3 * abs(b[1]) + 1e-4
        assert rel < (3e-3 if key[3] == "bf16" else 2e-4), key
    for key, (hist, p, cen) in ((k, v) for k, v in res.items() if k[0] == "ssl"):
        r_hist, r_p, r_cen = ref_ssl[key[2]]
        rel = float((p - r_p).norm() / r_p.norm())
        print(f"RCCL world-1 {key}: {hist[-1]} (plain {r_hist[-1]}), weights rel {rel:.2e}")
        for a, b in zip(hist, r_hist):
            for x, y in zip(a, b):
                assert abs(x - y) < 1e-2 * abs(y) + 2e-4
        assert rel < 1e-3
        assert float((cen - r_cen).norm() / (r_cen.norm() + 1e-30)) < 1e-3
