"""GPU: the reference's evaluation entry points (tools/test_zero_shot_hf.py, test_reconstruction_hf.py, test_linear_probing_hf.py --
their model-facing plumbing as restated and pinned in oracle/tools_oracle.py) driving `vtp_amd.VTPModel` through its public methods,
against the golden outputs of the REAL tools on the REAL reference model (tests/golden/tools_tiny.safetensors; SURVEY.md §8 f3,
VERDICT r3 item 9).  Bar: E_ours <= 1.5 E_ref per output, E_ref = the same plumbing on the oracle model under CPU bf16 autocast
(the reference algorithm's own bf16 noise; tiny model, single draws -> the slack of tests/test_model_gpu.py)."""
import os

import pytest
import torch
from safetensors.torch import load_file

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda"


def relF(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_eval_tools_plumbing_on_the_hip_model(golden_sd):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import tools_oracle as T
    from oracle.ref_stubs import TINY
    from vtp_amd import VTPConfig, VTPModel
    tg = load_file(os.path.join(ROOT, "tests", "golden", "tools_tiny.safetensors"))
    images, targets = tg["in.images"], tg["in.targets"]
    V, C = TINY["text_vocab_size"], TINY["text_context_length"]
    model = VTPModel(VTPConfig(**TINY))
    model.load_state_dict(golden_sd, strict=True)
    model = model.to(DEV).eval()
    ours = T.run_all(model, torch.device(DEV), images, targets, V, C)
    # E_ref: the MODEL calls of the oracle under bf16 autocast, the tools' own arithmetic (probe training, metrics) in fp32 -- the
    # reference tools wrap only the model call in their autocast context
    noisy = T.run_all(T.OracleModel(golden_sd, 2, 2, 2, autocast_dtype=torch.bfloat16), torch.device("cpu"), images, targets, V, C)
    for k in ("zs.classifier", "zs.logits", "rec.latents", "rec.recon_denorm", "lp.patch0", "lp.cls0", "lp.patch1", "lp.cls1",
              "lp.input_1_avg", "lp.input_2", "lp.w_after"):
        ref = tg["out." + k]
        e, e_ref = relF(ours[k], ref), relF(noisy[k], ref)
        print(f"TOOLS {k}: E_ours={e:.3e} E_ref={e_ref:.3e} ratio={e / e_ref:.2f}")
        assert ours[k].shape == ref.shape and e <= 1.5 * e_ref, (k, e, e_ref)
    # a few scalars each (PSNR in dB per image, summed cross-entropy per probe step): largest deviation against the reference's own
    # bf16 deviation, slack 2 (single draws) -- and never asked to agree beyond what the tools print (PSNR :.2f, loss :.3g)
    for k, printed in (("rec.psnr", 5e-3), ("lp.losses", 5e-4)):
        ref = tg["out." + k]
        e, e_ref = float((ours[k] - ref).abs().max()), float((noisy[k].float() - ref).abs().max())
        print(f"TOOLS {k}: max|err| ours={e:.3e} ref={e_ref:.3e}  values ours={[round(float(v), 4) for v in ours[k]]}")
        assert e <= max(2.0 * e_ref, printed), (k, e, e_ref)
    # top-1 / top-5 of 8 random images on a random-init model are ties broken by bf16 noise: reported, compared only through the logits
    print(f"TOOLS zero-shot top1/top5: ours={ours['zs.top'].tolist()} reference={tg['out.zs.top'].tolist()} oracle-bf16={noisy['zs.top'].tolist()}")
