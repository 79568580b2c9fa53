"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (vtp_amd/).

Imports the *real* reference implementation (MiniMax-AI/VTP, mounted read-only at
/root/reference) on CPU so that (a) the oracle restatement in ``oracle/vtp_oracle.py`` can be
validated against it and (b) golden fixtures under ``tests/golden/`` can be generated
(``oracle/make_golden.py``).  /root/reference does not exist on the GPU box, so everything that
uses this module must be skipped there (``reference_available()``).

The reference needs two packages that are not installed (omegaconf, torchvision); both are only
touched at import time (vtp/models/vtp.py:27, vtp/models/utils/text_utils.py:9), so two empty
stub modules are enough (recipe: SURVEY.md Appendix A).
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("VTP_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "vtp", "models"))


_loaded = None


def load_reference():
    """Returns a namespace with VTPConfig, VTPModel, VTP (legacy training arch) and DictConfig."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True  # reference tree is read-only
    import torch
    from transformers import PreTrainedModel, PretrainedConfig  # noqa: F401  (must precede the torchvision stub)

    if "omegaconf" not in sys.modules:
        om = types.ModuleType("omegaconf")

        class DictConfig(dict):
            pass

        class OmegaConf:
            pass

        om.DictConfig, om.OmegaConf = DictConfig, OmegaConf
        sys.modules["omegaconf"] = om
    if "torchvision" not in sys.modules:
        tv, tvo, tvm = (types.ModuleType(n) for n in ("torchvision", "torchvision.ops", "torchvision.ops.misc"))

        class FrozenBatchNorm2d(torch.nn.Module):
            pass

        tvm.FrozenBatchNorm2d = FrozenBatchNorm2d
        tv.ops = tvo
        tvo.misc = tvm
        sys.modules.update({"torchvision": tv, "torchvision.ops": tvo, "torchvision.ops.misc": tvm})
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from vtp.models.vtp_hf import VTPConfig, VTPModel
    from vtp.models.vtp import VTP

    ns = types.SimpleNamespace(VTPConfig=VTPConfig, VTPModel=VTPModel, VTP=VTP,
                               DictConfig=sys.modules["omegaconf"].DictConfig)
    _loaded = ns
    return ns


TINY = dict(  # tiny preset used for committed golden fixtures (small enough to commit weights)
    image_size=64,
    vision_embed_dim=128, vision_depth=2, vision_num_heads=2,
    text_embed_dim=128, text_depth=2, text_num_heads=2, text_vocab_size=512, text_context_length=16,
    decoder_embed_dim=128, decoder_depth=2, decoder_num_heads=2,
)
SMALL = dict(vision_embed_dim=384, vision_depth=12, vision_num_heads=6,
             text_embed_dim=384, text_depth=12, text_num_heads=6,
             decoder_embed_dim=384, decoder_depth=12, decoder_num_heads=6)
BASE = dict()  # VTPConfig() defaults are VTP-Base f16d64 (configuration_vtp.py:70-113)


def load_reference_lpips():
    """The reference's LPIPS class (vtp/utils/lpips.py) made constructible offline: a stub `torchvision.models.vgg16`
    returns a module whose `.features` has torchvision's VGG16 layer sequence (conv3x3/ReLU/MaxPool indices 0..30) with
    default-initialised weights, and the HTTP download of vgg.pth (lpips.py:76-82) is replaced by a no-op; the caller
    loads weights with load_state_dict."""
    load_reference()
    import importlib
    import torch
    from torch import nn

    tv = sys.modules["torchvision"]
    if not hasattr(tv, "models"):
        tvmod = types.ModuleType("torchvision.models")

        def vgg16(pretrained=False, **kw):
            cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
            layers, cin = [], 3
            for v in cfg:
                if v == "M":
                    layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
                else:
                    layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                    cin = v
            m = nn.Module()
            m.features = nn.Sequential(*layers)
            return m

        tvmod.vgg16 = vgg16
        tv.models = tvmod
        sys.modules["torchvision.models"] = tvmod
    for name in ("requests", "tqdm"):
        try:
            importlib.import_module(name)
        except ImportError:  # only used by the download helper
            sys.modules[name] = types.ModuleType(name)
    mod = importlib.import_module("vtp.utils.lpips")
    mod.LPIPS.load_from_pretrained = lambda self, name="vgg_lpips": None
    return mod.LPIPS
