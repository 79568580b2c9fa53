"""Golden fixture for the evaluation-tool plumbing -- TEST INFRASTRUCTURE ONLY.

Runs in the authoring container (needs /root/reference): imports the REAL tool modules tools/test_zero_shot_hf.py,
tools/test_linear_probing_hf.py and tools/test_reconstruction_hf.py (torchvision's transforms / datasets are stubbed: only their
names are touched at import time), drives their model-facing functions with the REAL reference VTPModel on the tiny seeded weights
of tests/golden/vtp_tiny.safetensors, checks that oracle/tools_oracle.py's restatement reproduces every one of them on the same
model, and writes the restatement's outputs (fp32) to tests/golden/tools_tiny.safetensors.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_tools"""
import importlib.util
import os
import sys
import types

import torch
from safetensors.torch import load_file, save_file

from . import tools_oracle as T
from .ref_stubs import REFERENCE_ROOT, TINY, load_reference

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_tool(name: str):
    """import /root/reference/tools/<name>.py by path (this repository has a tools/ directory of its own)"""
    load_reference()
    tv = sys.modules["torchvision"]
    if not hasattr(tv, "transforms"):
        tr = types.ModuleType("torchvision.transforms")

        class _Any:  # Compose / Resize / ... : constructed inside functions the fixture never calls
            def __init__(self, *a, **k):
                pass

        for n in ("Compose", "Resize", "CenterCrop", "RandomResizedCrop", "RandomHorizontalFlip", "ToTensor", "Normalize", "Lambda",
                  "InterpolationMode"):
            setattr(tr, n, _Any)
        ds = types.ModuleType("torchvision.datasets")
        ds.ImageFolder = _Any
        tv.transforms, tv.datasets = tr, ds
        sys.modules.update({"torchvision.transforms": tr, "torchvision.datasets": ds})
    if not hasattr(tv, "models"):  # tools/test_reconstruction_hf.py imports vtp.utils.lpips -> torchvision.models
        from .ref_stubs import load_reference_lpips
        load_reference_lpips()
    spec = importlib.util.spec_from_file_location("_ref_tool_" + name, os.path.join(REFERENCE_ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def inputs():
    g = torch.Generator().manual_seed(21)
    images = torch.randn(8, 3, 64, 64, generator=g)
    targets = torch.tensor([0, 1, 2, 3, 4, 5, 6, 0])
    return images, targets


def generate():
    """the restatement's outputs on the real reference model, after checking every restated function against the real tool function"""
    ref = load_reference()
    zs, lp, rc = load_tool("test_zero_shot_hf"), load_tool("test_linear_probing_hf"), load_tool("test_reconstruction_hf")
    g = load_file(os.path.join(ROOT, "tests", "golden", "vtp_tiny.safetensors"))
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    model = ref.VTPModel(ref.VTPConfig(**TINY))
    model.load_state_dict(sd, strict=True)
    model.eval()
    dev = torch.device("cpu")
    images, targets = inputs()
    tok = T.toy_tokenizer(TINY["text_vocab_size"], TINY["text_context_length"])
    mine = T.run_all(model, dev, images, targets, TINY["text_vocab_size"], TINY["text_context_length"])

    # ---- the real tool functions on the same model
    clf = zs.build_zero_shot_classifier(model, tok, T.CLASSNAMES, T.TEMPLATES, num_classes_per_batch=3, device=dev, use_tqdm=False)
    assert torch.allclose(clf, mine["zs.classifier"], atol=1e-6), "build_zero_shot_classifier"
    t1, t5 = zs.evaluate(model, clf, [(images[:4], targets[:4]), (images[4:], targets[4:])], dev, precision="fp32")
    assert abs(t1 - float(mine["zs.top"][0])) < 1e-9 and abs(t5 - float(mine["zs.top"][1])) < 1e-9, (t1, t5, mine["zs.top"])
    assert zs.accuracy(mine["zs.logits"], targets, topk=(1, 5)) == T.accuracy(mine["zs.logits"], targets, topk=(1, 5))
    fe = lp.FeatureExtractor(model, 2, torch.float32)
    feats = fe(images)
    for i, (p, c) in enumerate(feats):
        assert torch.equal(p, mine[f"lp.patch{i}"]) and torch.equal(c, mine[f"lp.cls{i}"])
    assert torch.equal(lp.create_linear_input(feats, 1, True), mine["lp.input_1_avg"])
    assert torch.equal(lp.create_linear_input(feats, 2, False), mine["lp.input_2"])
    torch.manual_seed(0)
    D = feats[0][1].shape[-1]
    clfs = lp.AllClassifiers({"blocks_1_avgpool_True": lp.LinearClassifier(2 * D, 1, True, len(T.CLASSNAMES)),
                              "blocks_2_avgpool_False": lp.LinearClassifier(2 * D, 2, False, len(T.CLASSNAMES))})
    opt = torch.optim.SGD(clfs.parameters(), lr=0.1, momentum=0.9, weight_decay=0)

    class _NoSched:
        def step(self):
            pass

    batches = [(images[:4], targets[:4]), (images[4:], targets[4:])] * 2
    avg = lp.train_one_epoch(fe, clfs, opt, _NoSched(), torch.nn.CrossEntropyLoss(), batches, 0, 4, dev)
    assert abs(avg - float(mine["lp.losses"].mean())) < 1e-5, (avg, mine["lp.losses"])
    assert torch.allclose(clfs.classifiers_dict["blocks_1_avgpool_True"].linear.weight, mine["lp.w_after"], atol=1e-6)
    for i in range(8):  # PSNR helper on the reconstruction block's outputs
        a = T.denormalize(images)[i].clamp(0, 1) * 255.0
        assert abs(rc.calculate_psnr(a, mine["rec.recon_denorm"][i] * 255.0) - float(mine["rec.psnr"][i])) < 1e-4
    out = {"in.images": images, "in.targets": targets}
    out.update({"out." + k: v.contiguous() for k, v in mine.items()})
    return out


def main():
    out = generate()
    path = os.path.join(ROOT, "tests", "golden", "tools_tiny.safetensors")
    save_file(out, path, metadata={"source": "oracle/make_golden_tools.py: real tools/test_*_hf.py functions on the reference VTPModel (TINY, vtp_tiny.safetensors weights)"})
    print("wrote", path, {k: tuple(v.shape) for k, v in out.items()})


if __name__ == "__main__":
    main()
