"""Two data-parallel ranks (two processes sharing the one GPU of the test box, gloo backend) must reproduce the
single-process step on the concatenated batch: bucketed gradient all-reduce during backward, contrastive feature
all-gather + gradient reduce-scatter, 1/world in the fused optimizer, in eager mode and with hipGraph segments."""
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


def _build(golden_sd, siglip=False):
    from oracle.ref_stubs import TINY
    from vtp_amd import VTPConfig, VTPModel
    m = VTPModel(VTPConfig(**dict(TINY, init_logit_bias=-3.0) if siglip else VTPConfig(**TINY).to_dict()))
    m.load_state_dict(golden_sd, strict=not siglip)
    return m.to("cuda:0")


def _data():
    g = torch.Generator().manual_seed(5)
    img = torch.randn(4, 3, 64, 64, generator=g)
    txt = torch.randint(1, 500, (4, 16), generator=g)
    txt[:, 0] = 510
    txt[torch.arange(4), torch.tensor([5, 9, 12, 15])] = 511
    return img, txt


def _worker(rank, world, port, use_graphs, out, shard=False, grad_dtype="fp32", siglip=False):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from safetensors.torch import load_file
    from vtp_amd import VTPTrainer
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    g = load_file(os.path.join(ROOT, "tests", "golden", "vtp_tiny.safetensors"))
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    m = _build(sd, siglip)
    tr = VTPTrainer(m, lr=1e-3, weight_decay=0.01, use_graphs=use_graphs, bucket_blocks=1, shard_optimizer=shard,
                    grad_dtype=grad_dtype)
    tr.time_comm = True
    img, txt = _data()
    sl = slice(rank * 2, rank * 2 + 2)
    losses = []
    for i in range(3):
        r, c = tr.step((img[sl] + 0.01 * i).cuda(), txt[sl].cuda())
        losses.append((float(r), float(c)))
    torch.cuda.synchronize()
    exposed = tr.comm_exposed_ms()
    assert exposed > 0 and tr.bucketer.comm_bytes > 0
    osd = tr.state_dict()  # sharded mode: moments are gathered from their owners
    mom = torch.cat([osd["exp_avg"][n].reshape(-1) for n in sorted(osd["exp_avg"])])
    out[rank] = (losses, m._engine().flat_p.detach().cpu().clone(), mom)
    dist.destroy_process_group()


@pytest.mark.parametrize("use_graphs", [False, True])
def test_two_ranks_match_single_process(golden_sd, use_graphs):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from vtp_amd import VTPTrainer
    m = _build(golden_sd)
    tr = VTPTrainer(m, lr=1e-3, weight_decay=0.01)
    img, txt = _data()
    ref_losses = []
    for i in range(3):
        r, c = tr.step((img + 0.01 * i).cuda(), txt.cuda())
        ref_losses.append((float(r), float(c)))
    ref_p = m._engine().flat_p.detach().cpu().clone()
    del tr, m
    torch.cuda.empty_cache()
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, use_graphs, out), nprocs=world, join=True)
    l0, p0, _ = out[0]
    l1, p1, _ = out[1]
    assert torch.equal(p0, p1), "ranks diverged"
    rel = float((p0 - ref_p).norm() / ref_p.norm())

    print("single:", ref_losses, "rank0:", l0, "rank1:", l1, f"weights rel diff {rel:.3e}")
    for i in range(3):
        # mean over ranks of the local losses == the single-process loss on the concatenated batch
        assert abs(0.5 * (l0[i][0] + l1[i][0]) - ref_losses[i][0]) < 2e-3 * ref_losses[i][0]
        assert abs(0.5 * (l0[i][1] + l1[i][1]) - ref_losses[i][1]) < 5e-3 * ref_losses[i][1]
    assert rel < 2e-4



@pytest.mark.parametrize("use_graphs,grad_dtype", [(False, "fp32"), (True, "fp32"), (True, "bf16")])
def test_sharded_optimizer_matches_allreduce(golden_sd, use_graphs, grad_dtype):
    """reduce-scatter + rank-sharded AdamW + parameter all-gather == all-reduce + replicated AdamW: same weights on both
    ranks, same Adam moments after gathering them from their owners (fp32 buckets: to summation order; bf16 buckets: to the
    one rounding of the local gradients)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    res = {}
    for shard in (False, True):
        world, port = 2, _free_port()
        out = mp.Manager().dict()
        mp.spawn(_worker, args=(world, port, use_graphs, out, shard, grad_dtype if shard else "fp32"), nprocs=world, join=True)
        assert torch.equal(out[0][1], out[1][1]), "ranks diverged"
        assert torch.equal(out[0][2], out[1][2]), "gathered moments differ between ranks"
        res[shard] = out[0]
    rel_p = float((res[True][1] - res[False][1]).norm() / res[False][1].norm())
    rel_m = float((res[True][2] - res[False][2]).norm() / res[False][2].norm())
    print(f"sharded vs all-reduce ({grad_dtype}, graphs={use_graphs}): weights rel {rel_p:.3e}, exp_avg rel {rel_m:.3e}")
    assert rel_p < (5e-6 if grad_dtype == "fp32" else 3e-3)  # Adam normalises: 3 steps at lr 1e-3 amplify the bf16 rounding of g
    assert rel_m < (1e-4 if grad_dtype == "fp32" else 1e-2)  # fp32: summation order (fused bias-gradient atomics) only
    for a, b in zip(res[True][0], res[False][0]):
        assert abs(a[0] - b[0]) < 1e-4 * abs(b[0]) + 1e-6


def _ssl_batch(B=4, n_local=4, R=64, r=32):
    g = torch.Generator().manual_seed(11)
    glob = torch.randn(2 * B, 3, R, R, generator=g)            # view-major: [view 0 images | view 1 images]
    loc = torch.randn(n_local * B, 3, r, r, generator=g)       # crop-major: index = crop * B + image
    masks = torch.rand(2 * B, (R // 16) ** 2, generator=g) < 0.3
    masks[1] = False                                            # an image without masked patches
    img = torch.randn(B, 3, R, R, generator=g)
    return img, glob, loc, masks


def _ssl_shard(rank, world, B, n_local, img, glob, loc, masks):
    bl = B // world
    ids = torch.arange(rank * bl, (rank + 1) * bl)
    gi = torch.cat([ids, B + ids])
    li = torch.cat([j * B + ids for j in range(n_local)])
    return img[ids], glob[gi], loc[li], masks[gi]


def _ssl_worker(rank, world, port, centering, out):
    sys.path.insert(0, ROOT)
    import importlib.util
    import torch.distributed as dist
    from safetensors.torch import load_file
    from vtp_amd import VTPTrainer
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    spec = importlib.util.spec_from_file_location("_ssl_t", os.path.join(ROOT, "tests", "test_ssl_gpu.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = load_file(os.path.join(ROOT, "tests", "golden", "vtp_tiny_ssl.safetensors"))
    torch.manual_seed(0)  # the towers the SSL golden does not cover (decoder, text) are random-initialised: same seed everywhere
    m = mod.build_vtp({k[3:]: v for k, v in g.items() if k.startswith("sd.")})
    tr = VTPTrainer(m, lr=1e-3, weight_decay=0.0, centering=centering, bucket_blocks=1, use_graphs=True)  # every rank must use the same driver: the graph driver's eager warm-up step runs the collectives too
    B, n_local = 4, 4
    img, glob, loc, masks = _ssl_shard(rank, world, B, n_local, *_ssl_batch(B, n_local))
    losses = []
    for i in range(2):
        ssl = tr.prepare_ssl(glob.cuda(), loc.cuda(), masks, upperbound=int(0.5 * masks.numel()))
        tr.step(img.cuda(), None, ssl)
        losses.append(float(tr.ssl_loss_sum))
    torch.cuda.synchronize()
    out[rank] = (losses, m._engine().flat_p.detach().cpu().clone(), tr.center_dino.cpu().clone())
    if world > 1:
        dist.destroy_process_group()


@pytest.mark.parametrize("centering", ["softmax", "sinkhorn_knopp"])
def test_two_ranks_ssl_step_matches_single_process(centering):
    """rec + DINO/iBOT step, 2 ranks x 2 images vs 1 process x 4 images: gradient buckets, the centre statistics all-reduce (softmax
    centring) or the phased Sinkhorn-Knopp all-reduces, EMA teacher -- same weights and centres afterwards"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    res = {}
    for world in (1, 2):
        out = mp.Manager().dict()
        mp.spawn(_ssl_worker, args=(world, _free_port(), centering, out), nprocs=world, join=True)
        res[world] = dict(out)
    l1, p1, c1 = res[1][0]
    (la, pa, ca), (lb, pb, cb) = res[2][0], res[2][1]
    assert torch.equal(pa, pb) or float((pa - pb).norm() / pa.norm()) < 1e-6, "ranks diverged"
    rel = float((pa - p1).norm() / p1.norm())
    print(f"[{centering}] single {l1} rank0 {la} rank1 {lb}; weights rel {rel:.3e}; centre rel {float((ca - c1).norm() / (c1.norm() + 1e-30)):.3e}")
    for i in range(2):
        assert abs(0.5 * (la[i] + lb[i]) - l1[i]) < 2e-3 * abs(l1[i])
    assert rel < 5e-4
    if centering == "softmax":
        assert float((ca - c1).norm() / c1.norm()) < 1e-3 and torch.equal(ca, cb)



def test_two_ranks_siglip_matches_single_process(golden_sd):
    """SigLIP under data parallelism: every (local image, any rank's text) pair counted once, text gradients returned by
    reduce-scatter -- 2 ranks x 2 pairs == 1 process x 4 pairs"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from vtp_amd import VTPTrainer
    m = _build(golden_sd, siglip=True)
    tr = VTPTrainer(m, lr=1e-3, weight_decay=0.01)
    img, txt = _data()
    ref_losses = [tuple(float(x) for x in tr.step((img + 0.01 * i).cuda(), txt.cuda())) for i in range(3)]
    ref_p = m._engine().flat_p.detach().cpu().clone()
    del tr, m
    torch.cuda.empty_cache()
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(2, _free_port(), False, out, False, "fp32", True), nprocs=2, join=True)
    (l0, p0, _), (l1, p1, _) = out[0], out[1]
    rel = float((p0 - ref_p).norm() / ref_p.norm())
    print("siglip single:", ref_losses, "rank0:", l0, "rank1:", l1, f"weights rel {rel:.3e}")
    assert torch.equal(p0, p1)
    for i in range(3):
        assert abs(0.5 * (l0[i][1] + l1[i][1]) - ref_losses[i][1]) < 5e-3 * abs(ref_losses[i][1])
    assert rel < 2e-4


def _resume_worker(rank, world, port, shard, tmp, out):
    """run A: 4 uninterrupted steps.  run B: 2 steps, then checkpoint (model.state_dict() + VTPTrainer.state_dict(), the latter a
    collective under shard_optimizer).  run C: a DIFFERENTLY initialised model + a new trainer load the checkpoint and take
    steps 3-4.  C must land where A did: student, EMA teacher, DINO head, SSL centres and the (rank-sharded) Adam moments."""
    sys.path.insert(0, ROOT)
    import importlib.util
    import torch.distributed as dist
    from safetensors.torch import load_file
    from vtp_amd import VTPTrainer
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    spec = importlib.util.spec_from_file_location("_ssl_t", os.path.join(ROOT, "tests", "test_ssl_gpu.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = load_file(os.path.join(ROOT, "tests", "golden", "vtp_tiny_ssl.safetensors"))
    sd0 = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    B, n_local = 4, 4
    img, glob, loc, masks = _ssl_shard(rank, world, B, n_local, *_ssl_batch(B, n_local))
    txt = torch.randint(1, 60, (B, 8), generator=torch.Generator().manual_seed(2))
    txt[:, 5] = 63
    txt = txt[rank * (B // world):(rank + 1) * (B // world)]

    def trainer(seed):
        torch.manual_seed(seed)  # decoder / text tower are not in the SSL golden: random-initialised from this seed
        m = mod.build_vtp(sd0)
        return m, VTPTrainer(m, lr=1e-3, weight_decay=0.01, teacher_momentum=0.9, bucket_blocks=1, shard_optimizer=shard)

    def steps(tr, lo, hi):
        for i in range(lo, hi):
            ssl = tr.prepare_ssl((glob + 0.01 * i).cuda(), (loc - 0.01 * i).cuda(), masks, upperbound=int(0.5 * masks.numel()))
            tr.step((img + 0.02 * i).cuda(), txt.cuda(), ssl)
        torch.cuda.synchronize()

    def snapshot(m, tr):
        osd = tr.state_dict()
        mom = torch.cat([osd[k][n].reshape(-1) for k in ("exp_avg", "exp_avg_sq") for n in sorted(osd[k])])
        return (m._engine().flat_p.detach().cpu().clone(), mom, tr.center_dino.cpu().clone(), tr.center_ibot.cpu().clone(), tr.step_no,
                {k: v.detach().cpu().clone() for k, v in m.state_dict().items() if k.startswith(("teacher_", "dino_head."))})

    mA, tA = trainer(0)
    steps(tA, 0, 4)
    ref = snapshot(mA, tA)
    del mA, tA
    mB, tB = trainer(0)
    steps(tB, 0, 2)
    tsd = tB.state_dict()  # collective in sharded mode: every rank calls it, rank 0 writes
    if rank == 0:
        torch.save({"model": {k: v.detach().cpu() for k, v in mB.state_dict().items()}, "trainer": tsd}, os.path.join(tmp, "ckpt.pt"))
    dist.barrier()
    del mB, tB
    ck = torch.load(os.path.join(tmp, "ckpt.pt"), weights_only=False)
    torch.manual_seed(1234 + rank)  # a different (and rank-dependent) initialisation: everything must come from the checkpoint
    mC = mod.build_vtp(sd0)
    with torch.no_grad():
        for p in mC.parameters():
            p.add_(0.1 * torch.randn_like(p))
    mC.load_state_dict(ck["model"], strict=True)
    tC = VTPTrainer(mC, lr=1e-3, weight_decay=0.01, teacher_momentum=0.9, bucket_blocks=1, shard_optimizer=shard)
    tC.load_state_dict(ck["trainer"])
    assert tC.step_no == 2
    steps(tC, 2, 4)
    out[rank] = (ref, snapshot(mC, tC))
    dist.destroy_process_group()


@pytest.mark.parametrize("shard", [True, False])
def test_resume_vtp_ssl_training_state_world2(shard, tmp_path):
    """VERDICT r3 item 1c / SURVEY §8 f4: resume of the class that is benchmarked -- legacy VTP (student + EMA teacher + DINO head,
    vtp/models/vtp.py:262-268,388-401) in a rec + clip + DINO/iBOT run on 2 ranks with the rank-sharded optimizer (and with the
    replicated one): uninterrupted 4 steps == 2 steps + checkpoint + fresh processes' state + 2 steps."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = mp.Manager().dict()
    mp.spawn(_resume_worker, args=(2, _free_port(), shard, str(tmp_path), out), nprocs=2, join=True)
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
    for rank in (0, 1):
        (p_a, mom_a, cd_a, ci_a, n_a, t_a), (p_c, mom_c, cd_c, ci_c, n_c, t_c) = out[rank]
        worst_t = max(rel(t_c[k], t_a[k]) for k in t_a)
        print(f"resume shard={shard} rank {rank}: weights rel {rel(p_c, p_a):.3e} moments rel {rel(mom_c, mom_a):.3e} centres "
              f"{rel(cd_c, cd_a):.3e}/{rel(ci_c, ci_a):.3e} teacher+head worst tensor rel {worst_t:.3e}")
        assert n_a == n_c == 4
        # not bit-equal: fp32 atomics in the fused bias-gradient sums reorder from run to run (same bound as the eager-vs-graph test)
        assert rel(p_c, p_a) < 1e-3 and rel(mom_c, mom_a) < 5e-3 and worst_t < 1e-3
        assert rel(cd_c, cd_a) < 1e-3 and rel(ci_c, ci_a) < 1e-3
    assert torch.equal(out[0][1][0], out[1][1][0]), "resumed ranks diverged"
