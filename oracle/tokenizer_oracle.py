"""CPU restatement of the byte-image ends of the reference tokenizer -- TEST INFRASTRUCTURE ONLY (tests/ may import it; the
product path in vtp_amd/tokenizer.py never does).

  to_tensor_normalize   torchvision ToTensor + Normalize(+ RandomHorizontalFlip(p in {0,1})) as chained in
                        VTP_Tokenizer.img_transform (generation/tokenizer/vtp_tokenizer.py:74-81); torchvision is not
                        installed here, so its documented arithmetic is restated: ToTensor = uint8 HWC -> float CHW `.div(255)`,
                        Normalize = `tensor.sub_(mean).div_(std)` with fp32 mean / std tensors, hflip = reverse the last axis
  decode_tail           VTP_Tokenizer.decode_to_images after the model call (vtp_tokenizer.py:105-111)
  latent_stats          per-channel mean / unbiased std over (N, H, W)  (layout of generation/latent_stats/*/latents_stats.pt)
"""
import numpy as np
import torch


def to_tensor_normalize(u8_nhwc: np.ndarray, mean, std, flip: bool = False) -> torch.Tensor:
    t = torch.from_numpy(np.ascontiguousarray(u8_nhwc)).permute(0, 3, 1, 2).contiguous()
    if flip:
        t = t.flip(-1)
    t = t.to(torch.float32).div(255)
    m = torch.tensor(mean, dtype=torch.float32).view(1, 3, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(1, 3, 1, 1)
    return t.sub_(m).div_(s)


def decode_tail(decoded: torch.Tensor, norm_mean, norm_std) -> np.ndarray:
    inv_mean = [-m / s for m, s in zip(norm_mean, norm_std)]
    inv_std = [1.0 / s for s in norm_std]
    m = torch.tensor(inv_mean, dtype=torch.float32).view(1, 3, 1, 1)
    s = torch.tensor(inv_std, dtype=torch.float32).view(1, 3, 1, 1)
    x = decoded.float().clone().sub_(m).div_(s)
    images = torch.clamp(x * 255, 0, 255)
    return images.permute(0, 2, 3, 1).to("cpu", dtype=torch.uint8).numpy()


def latent_stats(latents: torch.Tensor) -> dict:
    x = latents.double()
    return {"mean": x.mean(dim=[0, 2, 3], keepdim=True).float(), "std": x.std(dim=[0, 2, 3], keepdim=True).float()}
