"""VTPTrainer.step with the reference's training inputs (vtp/models/vtp.py:323-338: `image` AND `reconstruction_image`; per-objective
stochastic-depth rates clip_drop_rate / ssl_drop_rate / rec_drop_rate, vtp.py:205-207): a separate reconstruction input runs as its
own item of the list forward, and its gradients equal the sum of the two separate steps; hipGraph segments == eager; per-objective
rates draw per-item image subsets."""
import os

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


def _model(golden_sd):
    from oracle.ref_stubs import TINY
    from vtp_amd import VTPConfig, VTPModel
    m = VTPModel(VTPConfig(**TINY))
    m.load_state_dict(golden_sd, strict=True)
    return m.to(DEV)


def _data():
    g = torch.Generator().manual_seed(5)
    a = torch.randn(4, 3, 64, 64, generator=g)
    b = torch.randn(4, 3, 64, 64, generator=g)
    txt = torch.randint(1, 500, (4, 16), generator=g)
    txt[:, 0] = 510
    txt[torch.arange(4), torch.tensor([5, 9, 12, 15])] = 511
    return a.to(DEV), b.to(DEV), txt.to(DEV)


def test_separate_reconstruction_input_equals_sum_of_two_steps(golden_sd):
    from vtp_amd import VTPTrainer
    a, b, txt = _data()
    m = _model(golden_sd)
    st = m._engine()
    kw = dict(lr=0.0, weight_decay=0.0)
    tr = VTPTrainer(m, **kw)
    l_rec, l_clip = tr.step(a, txt, reconstruction_image=b)
    g_sep, l_sep = st.flat_g.clone(), (float(l_rec), float(l_clip))
    # the same tensor object (or None): the shared pass, as before
    tr.step(a, txt, reconstruction_image=a)
    g_shared = st.flat_g.clone()
    tr.step(a, txt)
    assert relF(st.flat_g, g_shared) < 1e-5  # (bias-gradient sums use fp32 atomics: equal to summation order)
    # reference: clip on `a` alone (rec weight 0) + rec on `b` alone
    t_clip = VTPTrainer(m, rec_weight=0.0, **kw)
    _, lc = t_clip.step(a, txt)
    g_clip = st.flat_g.clone()
    t_rec = VTPTrainer(m, **kw)
    lr_, _ = t_rec.step(b)
    g_rec = st.flat_g.clone()
    print(f"separate step: rec {l_sep[0]:.6f} (alone {float(lr_):.6f}) clip {l_sep[1]:.6f} (alone {float(lc):.6f}); "
          f"grad rel diff vs sum {relF(g_sep, g_clip + g_rec):.2e}; vs the shared-pass step {relF(g_sep, g_shared):.2e}")
    assert abs(l_sep[0] - float(lr_)) < 1e-5 * abs(float(lr_)) and abs(l_sep[1] - float(lc)) < 1e-5 * abs(float(lc)) + 1e-7
    assert relF(g_sep, g_clip + g_rec) < 2e-4
    assert relF(g_sep, g_shared) > 1e-3, "a different reconstruction input must change the gradients"
    # hipGraph segments == eager for the separate-input step
    res = []
    for use_graphs in (False, True):
        m2 = _model(golden_sd)
        t2 = VTPTrainer(m2, lr=1e-3, weight_decay=0.0, use_graphs=use_graphs)
        res.append([tuple(float(v) for v in t2.step(a + 0.01 * i, txt, reconstruction_image=b)) for i in range(3)])
    for x, y in zip(*res):
        assert abs(x[0] - y[0]) < 1e-3 * abs(x[0]) and abs(x[1] - y[1]) < 5e-3 * abs(x[1]) + 1e-4


def test_per_objective_drop_rates(golden_sd):
    """clip_drop_rate != rec_drop_rate on the SAME images: two list items with their own subsets (keep 2 of 4 and 4 of 4), loss
    finite and decreasing; equal rates keep the shared item"""
    from vtp_amd import VTPTrainer
    a, _, txt = _data()
    m = _model(golden_sd)
    tr = VTPTrainer(m, lr=1e-3, weight_decay=0.0, clip_drop_rate=0.5, rec_drop_rate=0.0, drop_seed=1)
    hist = [tuple(float(v) for v in tr.step(a, txt)) for _ in range(4)]
    p = tr.trunk.stack.last_drop_plan
    assert p["batches"] == [4, 4] and p["keeps"] == [2, 4] and abs(p["scales"][0] - 2.0) < 1e-9 and p["scales"][1] == 1.0
    assert all(torch.isfinite(torch.tensor(h)).all() for h in hist) and hist[-1][0] < hist[0][0]
    tr2 = VTPTrainer(_model(golden_sd), lr=1e-3, weight_decay=0.0, drop_rate=0.5, drop_seed=1)
    tr2.step(a, txt)
    assert tr2.trunk.stack.last_drop_plan["batches"] == [4]


def test_prepare_ssl_batches_keep_their_index_tensors():
    """a loader that prepares many batches ahead (more than the staging ring has pinned slots) must find every
    batch's device index / mask tensors intact -- they are owned allocations, not views of a recycled ring slot."""
    import numpy as np
    from vtp_amd.train import HostStager
    st = HostStager(torch.device(DEV), slots=4)
    rng = np.random.default_rng(0)
    sent, got = [], []
    for i in range(9):
        arrays = {"idx": rng.integers(0, 1000, size=257 + i).astype(np.int32), "w": rng.random(33).astype(np.float32),
                  "masks": (rng.random((4, 16)) < 0.3).astype(np.uint8)}
        sent.append(arrays)
        got.append(st.upload(arrays))
    torch.cuda.synchronize()
    for arrays, up in zip(sent, got):
        for k, a in arrays.items():
            assert up[k].shape == a.shape and np.array_equal(up[k].cpu().numpy(), a), k
