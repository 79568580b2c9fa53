"""vtp_amd -- MI355X-native (gfx950 / CDNA4) VTP training hot path.

Public surface mirrors the reference (`from vtp.models.vtp_hf import VTPConfig, VTPModel`):
    from vtp_amd import VTPConfig, VTPModel, VTPTrainer
Everything below the Python API is hand-written HIP in libvtp_hip.so (C ABI: include/vtp_hip.h)."""
from .config import VTPConfig  # noqa: F401


def __getattr__(name):  # lazy: importing the package must not require torch.cuda
    if name == "VTPModel":
        from .model import VTPModel
        return VTPModel
    if name == "VTP":
        from .vtp import VTP
        return VTP
    if name == "LPIPS":
        from .lpips import LPIPS
        return LPIPS
    if name == "VTP_Tokenizer":
        from .tokenizer import VTP_Tokenizer
        return VTP_Tokenizer
    if name == "patch_model":
        from .patch import patch_model
        return patch_model
    if name == "VTPTrainer":
        from .train import VTPTrainer
        return VTPTrainer
    raise AttributeError(name)
