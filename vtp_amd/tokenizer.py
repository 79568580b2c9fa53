"""The downstream tokenizer flow on the MI355X: `VTP_Tokenizer` (generation/tokenizer/vtp_tokenizer.py:15-111) and the
latent-shard extraction of generation/tools/extract_features_vtp.py:22-128, with the same names, arguments, file names and
safetensors layout, so the LightningDiT side (ImgLatentDataset) reads what this writes.

What runs where: PIL decode / `center_crop_arr` stay on the host (vtp/utils/image_utils.py:5-33, restated in
`center_crop_arr`); ToTensor + Normalize + horizontal flip, the encode / decode towers, the uint8 image packing and the
per-channel latent statistics are gfx950 kernels (tokenizer.hip, the engines).  No CPU compute path: without the HIP library
every entry point raises.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops

# timm.data.constants.IMAGENET_DEFAULT_MEAN / STD (vtp_tokenizer.py:4,11)
NORMALIZE_HALF = {"mean": [0.5, 0.5, 0.5], "std": [0.5, 0.5, 0.5]}
NORMALIZE_IMAGENET = {"mean": [0.485, 0.456, 0.406], "std": [0.229, 0.224, 0.225]}


def center_crop_arr(pil_image, image_size: int):
    """ADM centre crop (vtp/utils/image_utils.py:5-33): halve with BOX while >= 2x, BICUBIC to the short side, centre crop."""
    from PIL import Image
    while min(*pil_image.size) >= 2 * image_size:
        pil_image = pil_image.resize(tuple(x // 2 for x in pil_image.size), resample=Image.BOX)
    scale = image_size / min(*pil_image.size)
    pil_image = pil_image.resize(tuple(round(x * scale) for x in pil_image.size), resample=Image.BICUBIC)
    arr = np.array(pil_image)
    cy, cx = (arr.shape[0] - image_size) // 2, (arr.shape[1] - image_size) // 2
    return Image.fromarray(arr[cy:cy + image_size, cx:cx + image_size])


class VTP_Tokenizer:
    """Same constructor and methods as the reference class.  `hf_model_path` is a VTPModel directory (save_pretrained layout) or
    an already constructed `vtp_amd.VTPModel` (random-init benches, tests)."""

    def __init__(self, hf_model_path, img_size: int = 256, horizon_flip: float = 0.5, fp16: bool = True,
                 normalize_type: str = "imagenet", device: str = "cuda"):
        self.img_size, self.horizon_flip, self.fp16, self.normalize_type = img_size, horizon_flip, fp16, normalize_type
        self._setup_normalization(normalize_type)
        from .model import VTPModel
        if isinstance(hf_model_path, torch.nn.Module):
            self.model = hf_model_path
        else:
            self.model = VTPModel.from_pretrained(hf_model_path)
        self.model = self.model.to(device).eval()
        self.device = torch.device(device)
        config = self.model.config
        self.patch_size = config.vision_patch_size
        self.embed_dim = config.vision_feature_bottleneck
        self.downsample_ratio = self.patch_size
        self.latent_size = img_size // self.downsample_ratio

    def _setup_normalization(self, normalize_type: str):
        if normalize_type == "half":
            cfg = NORMALIZE_HALF
        elif normalize_type == "imagenet":
            cfg = NORMALIZE_IMAGENET
        else:
            raise ValueError(f"Unknown normalize_type: {normalize_type}. Use 'half' or 'imagenet'.")
        self.norm_mean, self.norm_std = cfg["mean"], cfg["std"]
        # inverse normalisation as the reference builds it (vtp_tokenizer.py:67-72): Normalize(-mean/std, 1/std)
        self.inv_mean = [-m / s for m, s in zip(self.norm_mean, self.norm_std)]
        self.inv_std = [1.0 / s for s in self.norm_std]

    def img_transform(self, p_hflip: float = 0, img_size: Optional[int] = None):
        """The reference's dataset transform (vtp_tokenizer.py:74-81: center_crop_arr -> RandomHorizontalFlip(p) -> ToTensor ->
        Normalize) as a plain callable PIL image -> f32 [3, S, S] on the host (torchvision is not a dependency here; ImageFolder
        only needs a callable).  Same fp32 op order as torchvision: x / 255, then (x - mean) / std.  Batches on the device go through
        `crop_to_u8` + `images_from_u8` (one kernel instead)."""
        size = self.img_size if img_size is None else img_size
        mean = torch.tensor(self.norm_mean, dtype=torch.float32).view(3, 1, 1)
        std = torch.tensor(self.norm_std, dtype=torch.float32).view(3, 1, 1)

        def transform(pil_image):
            u8 = torch.from_numpy(self.crop_to_u8(pil_image, size).copy())          # [S, S, 3]
            if p_hflip > 0 and float(torch.rand(1)) < p_hflip:
                u8 = u8.flip(1)
            x = u8.permute(2, 0, 1).contiguous().to(torch.float32).div(255)
            return x.sub_(mean).div_(std)
        return transform

    def transform_inv(self, x: torch.Tensor) -> torch.Tensor:
        """vtp_tokenizer.py:67-72: Normalize(-mean / std, 1 / std), i.e. x * std + mean in the reference's op order"""
        m = torch.tensor(self.inv_mean, dtype=x.dtype, device=x.device).view(-1, 1, 1)
        sd = torch.tensor(self.inv_std, dtype=x.dtype, device=x.device).view(-1, 1, 1)
        return (x - m) / sd

    # ---- host side of img_transform (vtp_tokenizer.py:74-81): crop on the host, the rest on the device
    def crop_to_u8(self, pil_image, img_size: Optional[int] = None) -> np.ndarray:
        """PIL image -> uint8 [S, S, 3] (RGB) centre crop; feed batches of these to `images_from_u8`."""
        size = self.img_size if img_size is None else img_size
        return np.asarray(center_crop_arr(pil_image.convert("RGB"), size), dtype=np.uint8)

    def images_from_u8(self, u8_nhwc, flip: bool = False) -> torch.Tensor:
        """uint8 [B, H, W, 3] (host or device) -> normalised f32 [B, 3, H, W] on the device: ToTensor + Normalize, with the
        p = 1 horizontal flip of the `latents_flip` pass (extract_features_vtp.py:55-58) folded into the same kernel."""
        t = torch.as_tensor(u8_nhwc)
        if t.dtype != torch.uint8 or t.dim() != 4 or t.shape[-1] != 3:
            raise ValueError(f"expected uint8 [B, H, W, 3], got {t.dtype} {tuple(t.shape)}")
        t = t.to(self.device, non_blocking=True).contiguous()
        out = torch.empty(t.shape[0], 3, t.shape[1], t.shape[2], dtype=torch.float32, device=self.device)
        ops.u8_to_images(t, out, self.norm_mean, self.norm_std, flip)
        return out

    # ---- the reference's two methods
    def encode_images(self, images: torch.Tensor) -> torch.Tensor:
        """normalised images [B, 3, H, W] -> latents f32 [B, C, H/16, W/16] on the CPU (vtp_tokenizer.py:83-95)"""
        with torch.no_grad():
            lat = self.encode_images_device(images)
            return lat.detach().cpu()

    def encode_images_device(self, images: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            if not images.is_cuda:
                images = images.to(self.device)
            _, _, H, W = images.shape
            self._current_img_h, self._current_img_w = H, W
            return self.model.get_reconstruction_latents(images)

    def decode_to_images(self, z: torch.Tensor) -> np.ndarray:
        """latents [B, C, h, w] -> uint8 images [B, 16h, 16w, 3] (numpy), vtp_tokenizer.py:97-111"""
        with torch.no_grad():
            if not z.is_cuda:
                z = z.to(self.device)
            _, _, hl, wl = z.shape
            self._current_img_h, self._current_img_w = hl * self.patch_size, wl * self.patch_size
            decoded = self.model.get_latents_decoded_images(z).float().contiguous()
            out = torch.empty(decoded.shape[0], decoded.shape[2], decoded.shape[3], 3, dtype=torch.uint8, device=decoded.device)
            ops.images_to_u8(decoded, out, self.inv_mean, self.inv_std)
            return out.cpu().numpy()


# ---------------------------------------------------------------------------------------------------------------------
# latent shards (extract_features_vtp.py:69-118)
# ---------------------------------------------------------------------------------------------------------------------
def shard_name(rank: int, shard: int) -> str:
    return f"latents_rank{rank:02d}_shard{shard:03d}.safetensors"


def distributed_indices(n: int, world_size: int, rank: int) -> List[int]:
    """DistributedSampler(shuffle=False, drop_last=False) order (extract_features_vtp.py:59-62): indices padded by wrapping to a
    multiple of world_size, then rank::world_size."""
    total = (n + world_size - 1) // world_size * world_size
    idx = list(range(n))
    pad = total - n
    if pad:
        idx += (idx * ((pad + n - 1) // max(n, 1) + 1))[:pad]
    return idx[rank:total:world_size]


class LatentShardWriter:
    """Accumulates (latents, latents_flip, labels) batches and writes a shard every `10000 // batch_size` batches, plus the
    remainder on close(): keys, dtypes, metadata and file names of extract_features_vtp.py:88-118."""

    def __init__(self, output_dir: str, rank: int = 0, batch_size: int = 1, batches_per_shard: Optional[int] = None):
        self.output_dir, self.rank = output_dir, rank
        self.batches_per_shard = 10000 // batch_size if batches_per_shard is None else batches_per_shard
        if self.batches_per_shard < 1:
            raise ValueError("batch_size larger than the 10000-sample shard")
        self.latents, self.latents_flip, self.labels = [], [], []
        self.saved_files = 0
        os.makedirs(output_dir, exist_ok=True)

    def add(self, latents: torch.Tensor, latents_flip: torch.Tensor, labels: torch.Tensor):
        self.latents.append(latents.detach().cpu())
        self.latents_flip.append(latents_flip.detach().cpu())
        self.labels.append(torch.as_tensor(labels).detach().cpu())
        if len(self.latents) == self.batches_per_shard:
            self.flush()

    def flush(self) -> Optional[str]:
        if not self.latents:
            return None
        from safetensors.torch import save_file
        d = {"latents": torch.cat(self.latents, dim=0).contiguous(), "latents_flip": torch.cat(self.latents_flip, dim=0).contiguous(),
             "labels": torch.cat(self.labels, dim=0).contiguous()}
        path = os.path.join(self.output_dir, shard_name(self.rank, self.saved_files))
        save_file(d, path, metadata={"total_size": f'{d["latents"].shape[0]}', "dtype": f'{d["latents"].dtype}',
                                     "device": f'{d["latents"].device}'})
        self.latents, self.latents_flip, self.labels = [], [], []
        self.saved_files += 1
        return path

    close = flush


def latent_stats(shard_paths: Sequence[str], device: str = "cuda") -> dict:
    """Per-channel mean / std over the `latents` of all shards -> {"mean": [1,C,1,1], "std": [1,C,1,1]} f32, the layout of
    generation/latent_stats/*/latents_stats.pt.  fp64 sums on the device; std is the unbiased one (torch.std default).
    (The reference delegates this to LightningDiT's ImgLatentDataset, which is not vendored: parity unpinned.)"""
    from safetensors.torch import load_file
    sums, n, C = None, 0, None
    for p in shard_paths:
        lat = load_file(p)["latents"].to(device).float().contiguous()
        if sums is None:
            C = lat.shape[1]
            sums = torch.zeros(2 * C, dtype=torch.float64, device=device)
        ops.latent_channel_stats(lat, sums)
        n += lat.shape[0] * lat[0, 0].numel()
    if sums is None:
        raise ValueError("no shards")
    s, q = sums[:C], sums[C:]
    mean = s / n
    var = (q - n * mean * mean) / max(n - 1, 1)
    return {"mean": mean.float().view(1, C, 1, 1).cpu(), "std": var.clamp_min(0).sqrt().float().view(1, C, 1, 1).cpu()}


def extract_features(tokenizer: VTP_Tokenizer, samples: Sequence[Tuple[object, int]], output_dir: str, batch_size: int,
                     rank: int = 0, world_size: int = 1, batches_per_shard: Optional[int] = None, write_stats: bool = True,
                     group=None) -> List[str]:
    """The loop of extract_features_vtp.py:55-128 over `samples` = a sequence of (uint8 [S,S,3] array or PIL image, label):
    every rank encodes its DistributedSampler slice twice (plain and horizontally flipped) and writes its own shards; rank 0
    then writes latents_stats.pt.  Independent images -> no data-path collective, only the closing barrier."""
    idx = distributed_indices(len(samples), world_size, rank)
    writer = LatentShardWriter(output_dir, rank, batch_size, batches_per_shard)
    paths = []
    for b0 in range(0, len(idx), batch_size):
        chunk = [samples[i] for i in idx[b0:b0 + batch_size]]
        u8 = np.stack([s if isinstance(s, np.ndarray) else tokenizer.crop_to_u8(s) for s, _ in chunk])
        labels = torch.tensor([int(y) for _, y in chunk], dtype=torch.int64)
        u8_dev = torch.as_tensor(u8).to(tokenizer.device)
        z = tokenizer.encode_images(tokenizer.images_from_u8(u8_dev, flip=False))
        zf = tokenizer.encode_images(tokenizer.images_from_u8(u8_dev, flip=True))
        before = writer.saved_files
        writer.add(z, zf, labels)
        if writer.saved_files != before:
            paths.append(os.path.join(output_dir, shard_name(rank, before)))
    last = writer.close()
    if last:
        paths.append(last)
    if world_size > 1:
        import torch.distributed as dist
        dist.barrier(group=group)
    if write_stats and rank == 0:
        every = sorted(os.path.join(output_dir, f) for f in os.listdir(output_dir)
                       if f.startswith("latents_rank") and f.endswith(".safetensors"))
        torch.save(latent_stats(every, str(tokenizer.device)), os.path.join(output_dir, "latents_stats.pt"))
    if world_size > 1:
        import torch.distributed as dist
        dist.barrier(group=group)
    return paths
