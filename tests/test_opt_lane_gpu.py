"""The per-bucket optimizer lane of VTPTrainer (round 5): fused AdamW + EMA-teacher kernel, ranged weight refresh,
and the step with the lane on (bucket updates on a side stream beside the remaining backward, gradient zeroing under the forward) against
the serial optimizer leg (one AdamW + one EMA + one refresh launch behind the last backward kernel) -- eager and hipGraph segments."""
import pytest
import torch

from test_ssl_gpu import DEV, build_vtp, relF, sslg  # noqa: F401  (fixture + helpers)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


@pytest.mark.parametrize("n,with_teacher,with_table", [(4096 * 3, True, True), (4096 * 2 + 4 * 37, True, False), (260, False, True),
                                                       (1 << 20, True, True), (4, False, False)])
def test_fused_adamw_ema_kernel_is_the_two_kernels(n, with_teacher, with_table):
    """adamw_ema_dev == adamw_dev (masked) followed by ema_dev, bit for bit, on ranges that are not multiples of a block"""
    from vtp_amd import ops
    g = torch.Generator(device=DEV).manual_seed(n)
    p, gr = torch.randn(n, device=DEV, generator=g), torch.randn(n, device=DEV, generator=g) * 1e-2
    m, v = torch.randn(n, device=DEV, generator=g) * 1e-2, torch.rand(n, device=DEV, generator=g) * 1e-4
    t = torch.randn(n, device=DEV, generator=g)
    nd = (torch.rand(n // 4, device=DEV, generator=g) < 0.3).to(torch.uint8) if with_table else None
    hyper = torch.zeros(16, device=DEV)
    hyper[:10] = torch.tensor([1e-3, 0.9, 0.95, 1e-8, 0.05, 1 - 0.9 ** 3, (1 - 0.95 ** 3) ** 0.5, 0.5, 0.0, 0.994])
    a = [x.clone() for x in (p, m, v, t)]
    ops.adamw_dev(a[0], gr, a[1], a[2], None, n, hyper, nd)
    if with_teacher:
        ops.ema_dev(a[3], a[0], n, hyper[9:10])
    b = [x.clone() for x in (p, m, v, t)]
    guard = gr.clone()
    ops.adamw_ema_dev(b[0], gr, b[1], b[2], b[3] if with_teacher else None, n, hyper, nd)
    torch.cuda.synchronize()
    for x, y, name in zip(a, b, ("p", "m", "v", "teacher")):
        assert torch.equal(x, y), f"{name}: {int((x != y).sum())} of {n} elements differ"
    assert torch.equal(gr, guard)


def test_ranged_weight_refresh_equals_full_refresh(sslg):
    """prep_runs over every run of the table == one prep(): W, W^T, SwiGLU interleave and bias interleave copies"""
    _, sd = sslg
    m = build_vtp(sd)
    st = m._engine()
    with torch.no_grad():
        st.flat_p.add_(0.01 * torch.randn_like(st.flat_p))
    st.prep()
    ref = st.flat_bf16.clone()
    keep = [t.clone() for t in st._keep]
    st.flat_bf16.zero_()
    for t in st._keep:
        t.zero_()
    n = st._ndesc
    # bucket-like runs: first third, a middle slice starting inside the table, the rest
    for run in ((0, n // 3), (n // 3, n // 3 + 1), (n // 3 + 1, n)):
        st.prep_runs([run])
    torch.cuda.synchronize()
    assert torch.equal(st.flat_bf16, ref)
    for a, b in zip(st._keep, keep):
        assert torch.equal(a, b)
    # desc_runs of the whole parameter range = the whole table
    assert st.desc_runs([(0, st.numel)]) == [(0, n)]


@pytest.mark.parametrize("use_graphs", [False, True])
def test_lane_matches_serial_optimizer_leg(sslg, use_graphs):
    """4 steps of rec + clip + DINO/iBOT with the lane on == the serial leg: same kernels' arithmetic per element, so student, teacher and
    Adam moments agree to the run-to-run noise of the step itself (fp32 atomics in the bias-gradient sums; measured between two serial
    runs by the same test)"""
    from oracle.make_golden_ssl import SSL_CFG as C
    from vtp_amd import VTPTrainer
    g, sd = sslg
    img = torch.randn(C["B"], 3, C["R"], C["R"], device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    txt = torch.randint(1, 60, (C["B"], 8), device=DEV)
    txt[:, 5] = 63
    res = {}
    for tag, lane in (("serial", False), ("serial2", False), ("lane", True)):
        torch.manual_seed(0)
        m = build_vtp(sd)
        tr = VTPTrainer(m, lr=5e-4, weight_decay=0.05, use_graphs=use_graphs, teacher_momentum=0.9)
        tr.overlap_opt = lane
        ssl = tr.prepare_ssl(g["in.global_crops"].to(DEV), g["in.local_crops"].to(DEV), g["in.masks"].bool())
        for _ in range(4):
            tr.step(img, txt, ssl)
        torch.cuda.synchronize()
        st = m._engine()
        res[tag] = (st.flat_p.clone(), tr.m.clone(), tr.v.clone(), st.flat_bf16.clone())
        if lane:
            assert tr._opt_plans, "the lane must have planned bucket updates"
            # the teacher moved (EMA fused into the bucket updates) and every bf16 copy is current
            ref = st.flat_bf16.clone()
            st.prep()
            torch.cuda.synchronize()
            assert torch.equal(st.flat_bf16, ref), "a weight copy was not refreshed by the lane"
    noise = [relF(a, b) for a, b in zip(res["serial2"], res["serial"])]
    for name, a, b, nz in zip(("params (student + teacher)", "exp_avg", "exp_avg_sq", "bf16 copies"), res["lane"], res["serial"], noise):
        e = relF(a, b)
        print(f"optimizer lane vs serial leg ({'graphs' if use_graphs else 'eager'}) {name}: rel {e:.2e} (serial vs serial {nz:.2e})")
        # floor: the run-to-run level of the eager step (fp32 atomics of the bias-gradient sums reorder with the timing: moments 4e-6,
        # a bf16 weight copy flips an ulp at 5e-6); a graph replay happens to repeat itself to 1e-8, which is not the lane's reference
        assert e <= max(3 * nz, 2e-5), name


def test_lane_rec_only_and_rec_clip_steps(sslg):
    """objective sets without a teacher / without the text tower: the lane covers exactly the step's parameter ranges (it raises otherwise)"""
    from oracle.make_golden_ssl import SSL_CFG as C
    from vtp_amd import VTPTrainer
    _, sd = sslg
    m = build_vtp(sd)
    tr = VTPTrainer(m, lr=1e-3)
    assert tr.overlap_opt
    img = torch.randn(C["B"], 3, C["R"], C["R"], device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    txt = torch.randint(1, 60, (C["B"], 8), device=DEV)
    txt[:, 5] = 63
    l0 = float(tr.step(img)[0])
    for _ in range(3):
        l1 = float(tr.step(img)[0])
    assert l1 < l0
    c0 = float(tr.step(img, txt)[1])
    for _ in range(3):
        c1 = float(tr.step(img, txt)[1])
    assert c1 < c0


def test_lane_emas_every_pair_on_a_step_without_text(sslg):
    """rec + DINO/iBOT WITHOUT captions on a model that has a CLIP head: visual_proj gets no gradient (it is outside the step's ranges),
    but update_teacher (vtp.py:392-401) still moves teacher_proj towards it -- the lane's optimizer leg EMAs the pairs its buckets did not
    touch, like the serial leg"""
    from oracle.make_golden_ssl import SSL_CFG as C
    from vtp_amd import VTPTrainer
    from vtp_amd.vtp import _range
    g, sd = sslg
    img = torch.randn(C["B"], 3, C["R"], C["R"], device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    res = {}
    for tag, lane in (("serial", False), ("lane", True)):
        torch.manual_seed(0)
        m = build_vtp(sd)
        st = m._engine()
        proj = [(t, s) for t, s in m.ema_pairs() if t == "teacher_proj."]
        if not proj:
            pytest.skip("model without a teacher_proj pair")
        (tlo, thi), (slo, shi) = _range(st, proj[0][0]), _range(st, proj[0][1])
        with torch.no_grad():  # teacher_proj != visual_proj, so that an EMA step is visible
            st.flat_p[tlo:thi].add_(0.5)
        before = st.flat_p[tlo:thi].clone()
        tr = VTPTrainer(m, lr=5e-4, weight_decay=0.05, teacher_momentum=0.9)
        tr.overlap_opt = lane
        ssl = tr.prepare_ssl(g["in.global_crops"].to(DEV), g["in.local_crops"].to(DEV), g["in.masks"].bool())
        for _ in range(2):
            tr.step(img, None, ssl)
        torch.cuda.synchronize()
        assert not torch.equal(st.flat_p[tlo:thi], before), f"{tag}: teacher_proj did not move"
        exp = 0.9 * 0.9 * before + (1 - 0.81) * st.flat_p[slo:shi]  # the student's proj is constant on these steps
        assert torch.allclose(st.flat_p[tlo:thi], exp, rtol=1e-5, atol=1e-6), tag
        res[tag] = (st.flat_p.clone(), st.flat_bf16.clone())
        ref = st.flat_bf16.clone()
        st.prep()
        torch.cuda.synchronize()
        assert torch.equal(st.flat_bf16, ref), f"{tag}: a weight copy is stale"
    for a, b, name in zip(res["lane"], res["serial"], ("params", "bf16 copies")):
        assert relF(a, b) <= 2e-5, name
