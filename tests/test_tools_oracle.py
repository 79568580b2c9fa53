"""CPU: the restated evaluation-tool plumbing (oracle/tools_oracle.py) against (a) the committed golden fixture
written from the REAL tools/test_*_hf.py functions driving the real reference VTPModel, with the oracle model behind the method
surface, and (b) -- where /root/reference exists -- the real tool functions themselves, live."""
import os

import pytest
import torch
from safetensors.torch import load_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tg():
    return load_file(os.path.join(ROOT, "tests", "golden", "tools_tiny.safetensors"))


def test_restated_tools_on_the_oracle_model_reproduce_the_reference_tools(tg, golden_sd):
    from oracle import tools_oracle as T
    from oracle.ref_stubs import TINY
    m = T.OracleModel(golden_sd, 2, 2, 2)
    out = T.run_all(m, torch.device("cpu"), tg["in.images"], tg["in.targets"], TINY["text_vocab_size"], TINY["text_context_length"])
    assert set("out." + k for k in out) == set(k for k in tg if k.startswith("out."))
    for k, v in out.items():
        ref = tg["out." + k]
        assert v.shape == ref.shape, k
        err = float((v.float() - ref.float()).norm() / (ref.float().norm() + 1e-30))
        assert err < 2e-5, (k, err)  # fp32 vs fp32: summation order only
    assert torch.equal(out["zs.top"], tg["out.zs.top"])


def test_toy_tokenizer_layout():
    from oracle import tools_oracle as T
    tok = T.toy_tokenizer(512, 16)
    ids = tok(["a photo of a great white shark.", "x " * 40])
    assert ids.shape == (2, 16) and int(ids[0, 0]) == 510 and int(ids[0].max()) == 511 and int(ids[0].argmax()) == 8
    assert int(ids[1].argmax()) == 15 and (ids[1, 1:15] > 0).all() and (ids < 512).all()
    assert torch.equal(tok(["tench"]), tok(["tench"]))


def test_restated_tools_against_the_live_reference_tools(tg):
    from oracle.ref_stubs import reference_available
    if not reference_available():
        pytest.skip("reference tree not present (GPU box)")
    from oracle.make_golden_tools import generate
    out = generate()  # asserts restatement == real tool function for every path, on the real model
    for k, v in out.items():
        assert torch.allclose(v.float(), tg[k].float(), atol=1e-6, rtol=1e-5), k
