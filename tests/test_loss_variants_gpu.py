"""Loss-head variants inside the training step: SigLIP when `init_logit_bias` is set
(vtp/models/vtp.py:180-188), Sinkhorn-Knopp teacher targets and the KoLeo regulariser on the SSL branch -- the step's gradients
against the oracle's autograd (fp32 = the reference value, bf16 autocast = the noise floor)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def relF(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_siglip_step_gradients_vs_oracle_autograd(golden, golden_sd):
    from oracle import vtp_oracle as O
    from oracle.loss_oracle import siglip_loss
    from oracle.make_golden import GRAD2_KEYS
    from oracle.ref_stubs import TINY
    from vtp_amd import VTPConfig, VTPModel, VTPTrainer
    img, txt = golden["in.image"], golden["in.text"]
    bias0 = -3.0

    def oracle_grads(autocast):
        sd = {k: v.clone().requires_grad_(v.dtype == torch.float32) for k, v in golden_sd.items()}
        sd["logit_bias"] = torch.tensor(bias0, requires_grad=True)
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            l1 = O.rec_train_loss(sd, img, 2, 2)
            i = O.clip_image_feature(sd, img, 2)
            t = O.clip_text_feature(sd, txt, 2)
            ls = siglip_loss(i.float(), t.float(), sd["logit_scale"], sd["logit_bias"])
            (l1 + ls).backward()
        return sd, float(ls)

    ref, ls_ref = oracle_grads(False)
    noisy, _ = oracle_grads(True)
    m = VTPModel(VTPConfig(**dict(TINY, init_logit_bias=bias0)))
    missing, unexpected = m.load_state_dict(golden_sd, strict=False)
    assert list(missing) == ["logit_bias"] and not unexpected
    m = m.to(DEV)
    assert float(m.logit_bias) == bias0
    tr = VTPTrainer(m, lr=0.0, weight_decay=0.0)
    _, clip = tr.step(img.to(DEV), txt.to(DEV))
    torch.cuda.synchronize()
    print(f"siglip loss {float(clip):.5f} (oracle {ls_ref:.5f})")
    assert abs(float(clip) - ls_ref) < 5e-3 * abs(ls_ref)
    params = dict(m.named_parameters())
    for k in list(GRAD2_KEYS) + ["logit_bias", "logit_scale"]:
        e, e_ref = relF(params[k].grad, ref[k].grad), relF(noisy[k].grad, ref[k].grad)
        print(f"siglip grad {k}: E_ours={e:.3e} E_ref={e_ref:.3e}")
        assert e <= max(1.5 * e_ref, 2.5e-2) and e < 8e-2, k
    # graphs == eager, loss decreases, logits API adds the bias
    res = []
    for use_graphs in (False, True):
        m2 = VTPModel(VTPConfig(**dict(TINY, init_logit_bias=bias0)))
        m2.load_state_dict(golden_sd, strict=False)
        m2 = m2.to(DEV)
        tr2 = VTPTrainer(m2, lr=1e-3, weight_decay=0.0, use_graphs=use_graphs)
        res.append([float(tr2.step(img.to(DEV), txt.to(DEV))[1]) for _ in range(5)])
    print("eager", res[0], "graphs", res[1])
    assert res[0][-1] < res[0][0]
    for a, b in zip(*res):
        assert abs(a - b) < 2e-3 * abs(a)
    m.eval()
    with torch.no_grad():
        lg = m.get_clip_logits(img.to(DEV), txt.to(DEV))
    lg = lg[0] if isinstance(lg, (tuple, list)) else lg
    sdp = dict(golden_sd, logit_bias=torch.tensor(bias0))
    ref_lg = O.clip_logits(sdp, img, txt, 2, 2) + bias0
    assert relF(lg, ref_lg) < 2e-2


@pytest.mark.parametrize("centering,koleo", [("sinkhorn_knopp", 0.0), ("softmax", 0.1), ("sinkhorn_knopp", 0.1)])
def test_ssl_variant_step_gradients_vs_oracle_autograd(centering, koleo):
    from safetensors.torch import load_file
    from oracle import vtp_oracle as O
    from oracle.make_golden_ssl import SSL_CFG as C, SSL_GRAD_KEYS
    import importlib.util
    spec = importlib.util.spec_from_file_location('_ssl_t', os.path.join(ROOT, 'tests', 'test_ssl_gpu.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    build_vtp = mod.build_vtp
    from vtp_amd import VTPTrainer
    g = load_file(os.path.join(ROOT, "tests", "golden", "vtp_tiny_ssl.safetensors"))
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    masks = g["in.masks"].bool()

    def oracle_grads(autocast):
        sdr = {k: v.clone().requires_grad_(v.dtype == torch.float32 and not k.startswith("teacher_")) for k, v in sd.items()}
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            t_b, s_b = O.ssl_outputs(sdr, g["in.global_crops"], g["in.local_crops"], masks, C["heads"])
            loss = O.ssl_loss(t_b, s_b, masks, g["in.center_dino"], g["in.center_ibot"], n_local=C["n_local"], centering=centering,
                              koleo_weight=koleo)
            loss.backward()
        return sdr, float(loss)

    ref, loss_ref = oracle_grads(False)
    noisy, _ = oracle_grads(True)
    res = []
    for use_graphs in (False, True):
        m = build_vtp(sd)
        tr = VTPTrainer(m, lr=0.0, weight_decay=0.0, rec_weight=0.0, teacher_momentum=1.0, center_momentum=1.0, centering=centering, koleo_weight=koleo,
                        use_graphs=use_graphs)
        tr.center_dino.copy_(g["in.center_dino"])
        tr.center_ibot.copy_(g["in.center_ibot"])
        ssl = tr.prepare_ssl(g["in.global_crops"].to(DEV), g["in.local_crops"].to(DEV), masks)
        img = torch.randn(C["B"], 3, C["R"], C["R"], device=DEV)
        for _ in range(2 if use_graphs else 1):  # second call replays the captured segments (frozen teacher / centre)
            tr.step(img, None, ssl)
        torch.cuda.synchronize()
        loss = float(tr.ssl_loss_sum) + float(tr.koleo_loss_sum)
        print(f"[{centering} koleo={koleo} graphs={use_graphs}] loss {loss:.4f} (oracle {loss_ref:.4f})")
        assert abs(loss - loss_ref) < 1e-2 * abs(loss_ref)
        params = dict(m.named_parameters())
        for k in SSL_GRAD_KEYS:
            e, e_ref = relF(params[k].grad, ref[k].grad), relF(noisy[k].grad, ref[k].grad)
            print(f"  grad {k}: E_ours={e:.3e} E_ref={e_ref:.3e}")
            assert e <= max(1.5 * e_ref, 3e-2) and e < 0.1, k
        res.append(loss)
        if centering == "sinkhorn_knopp":  # no EMA centre in this mode
            assert torch.equal(tr.center_dino.cpu(), g["in.center_dino"])
    assert abs(res[0] - res[1]) < 2e-3 * abs(res[0])
