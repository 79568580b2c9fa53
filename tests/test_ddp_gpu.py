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


def _build(golden_sd):
    from oracle.ref_stubs import TINY
    from vtp_amd import VTPConfig, VTPModel
    m = VTPModel(VTPConfig(**TINY))
    m.load_state_dict(golden_sd, strict=True)
    return m.to("cuda:0")


def _data():
    g = torch.Generator().manual_seed(5)
    img = torch.randn(4, 3, 64, 64, generator=g)
    txt = torch.randint(1, 500, (4, 16), generator=g)
    txt[:, 0] = 510
    txt[torch.arange(4), torch.tensor([5, 9, 12, 15])] = 511
    return img, txt


def _worker(rank, world, port, use_graphs, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from safetensors.torch import load_file
    from vtp_amd import VTPTrainer
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    g = load_file(os.path.join(ROOT, "tests", "golden", "vtp_tiny.safetensors"))
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    m = _build(sd)
    tr = VTPTrainer(m, lr=1e-3, weight_decay=0.01, use_graphs=use_graphs, bucket_blocks=1)
    img, txt = _data()
    sl = slice(rank * 2, rank * 2 + 2)
    losses = []
    for i in range(3):
        r, c = tr.step((img[sl] + 0.01 * i).cuda(), txt[sl].cuda())
        losses.append((float(r), float(c)))
    torch.cuda.synchronize()
    out[rank] = (losses, m._engine().flat_p.detach().cpu().clone())
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
    l0, p0 = out[0]
    l1, p1 = out[1]
    assert torch.equal(p0, p1), "ranks diverged"
    rel = float((p0 - ref_p).norm() / ref_p.norm())

    print("single:", ref_losses, "rank0:", l0, "rank1:", l1, f"weights rel diff {rel:.3e}")
    for i in range(3):
        # mean over ranks of the local losses == the single-process loss on the concatenated batch
        assert abs(0.5 * (l0[i][0] + l1[i][0]) - ref_losses[i][0]) < 2e-3 * ref_losses[i][0]
        assert abs(0.5 * (l0[i][1] + l1[i][1]) - ref_losses[i][1]) < 5e-3 * ref_losses[i][1]
    assert rel < 2e-4
