"""Host-side data plumbing of the self-supervised objective: the block-wise iBOT mask generator and the collate step that
produces the `ssl_dict` consumed by `VTP.forward(..., forward_type="ssl")` (reference: the keyword set of
`VTP.forward_ssl_learning`, vtp/models/vtp.py:365-374 -- global_crops, n_global_crops, mask_indices_list, n_masked_patches,
upperbound, local_crops, masks).  The reference ships neither piece; the algorithms follow the
DINOv2 / iBOT / BEiT recipe the reference credits: rectangular blocks with log-uniform aspect ratio are added until the
requested number of patches is masked; per batch, `mask_probability` of the global crops are masked at ratios stratified
over `mask_ratio` and `upperbound` = sum of the strata maxima (the fixed size of the masked-token buffers, vtp.py:432-439).

Plain numpy / torch-CPU: this runs in the data-loader processes, not on the GPU."""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch


class BlockMaskGenerator:
    """Block-wise masking of an (h, w) patch grid: __call__(n) -> bool [h, w] with ~n masked patches (never more than n)."""

    def __init__(self, grid: Tuple[int, int], max_num_patches: Optional[int] = None, min_num_patches: int = 4,
                 min_aspect: float = 0.3, max_aspect: Optional[float] = None, rng: Optional[np.random.Generator] = None):
        self.h, self.w = int(grid[0]), int(grid[1])
        self.num_patches = self.h * self.w
        self.max_num_patches = self.num_patches // 2 if max_num_patches is None else int(max_num_patches)
        self.min_num_patches = int(min_num_patches)
        max_aspect = max_aspect or 1.0 / min_aspect
        self.log_aspect = (math.log(min_aspect), math.log(max_aspect))
        self.rng = rng or np.random.default_rng()

    def _add_block(self, mask: np.ndarray, budget: int) -> int:
        """try (<= 10 times) to place one rectangle that adds between 1 and `budget` newly masked patches"""
        for _ in range(10):
            area = self.rng.uniform(self.min_num_patches, max(budget, self.min_num_patches))
            aspect = math.exp(self.rng.uniform(*self.log_aspect))
            bh, bw = int(round(math.sqrt(area * aspect))), int(round(math.sqrt(area / aspect)))
            if not (0 < bh < self.h and 0 < bw < self.w):
                continue
            top = int(self.rng.integers(0, self.h - bh + 1))
            left = int(self.rng.integers(0, self.w - bw + 1))
            blk = mask[top:top + bh, left:left + bw]
            new = bh * bw - int(blk.sum())
            if 0 < new <= budget:
                blk[...] = True
                return new
        return 0

    def __call__(self, num_masking_patches: int = 0) -> np.ndarray:
        mask = np.zeros((self.h, self.w), dtype=bool)
        count = 0
        while count < num_masking_patches:
            budget = min(num_masking_patches - count, self.max_num_patches)
            added = self._add_block(mask, budget)
            if added == 0:
                break
            count += added
        return mask


def collate_ssl_masks(n_global_crops_total: int, grid: Tuple[int, int], mask_probability: float = 0.5,
                      mask_ratio: Tuple[float, float] = (0.1, 0.5), rng: Optional[np.random.Generator] = None,
                      generator: Optional[BlockMaskGenerator] = None) -> Dict[str, object]:
    """Masks of one batch of global crops (rows = crops, view-major like the `global_crops` tensor).

    Returns the mask-related entries of the reference's ssl_dict: masks bool [n, hw], mask_indices_list int64 [n_masked]
    (indices into the flattened [n * hw] patch grid), n_masked_patches (0-d int64 tensor), upperbound (int) and masks_weight
    f32 [n_masked] (1 / masked patches of the crop -- the per-token iBOT loss weight)."""
    rng = rng or np.random.default_rng()
    gen = generator or BlockMaskGenerator(grid, max_num_patches=int(0.5 * grid[0] * grid[1]), rng=rng)
    n, hw = int(n_global_crops_total), grid[0] * grid[1]
    n_masked_crops = int(n * mask_probability)
    edges = np.linspace(mask_ratio[0], mask_ratio[1], n_masked_crops + 1)
    masks, upperbound = [], 0
    for i in range(n_masked_crops):
        masks.append(gen(int(hw * rng.uniform(edges[i], edges[i + 1]))))
        upperbound += int(hw * edges[i + 1])
    for _ in range(n_masked_crops, n):
        masks.append(np.zeros(grid, dtype=bool))
    order = rng.permutation(n)
    m = np.stack([masks[i] for i in order]).reshape(n, hw)
    idx = np.flatnonzero(m.reshape(-1))
    per_crop = np.maximum(m.sum(1), 1)
    weight = (1.0 / per_crop)[idx // hw].astype(np.float32)
    return dict(masks=torch.from_numpy(m), mask_indices_list=torch.from_numpy(idx.astype(np.int64)),
                n_masked_patches=torch.tensor(int(idx.size), dtype=torch.long), upperbound=int(upperbound),
                masks_weight=torch.from_numpy(weight))


def collate_ssl_batch(global_crops: Sequence[torch.Tensor], local_crops: Sequence[torch.Tensor], patch_size: int = 16,
                      mask_probability: float = 0.5, mask_ratio: Tuple[float, float] = (0.1, 0.5),
                      rng: Optional[np.random.Generator] = None) -> Dict[str, object]:
    """samples -> ssl_dict.  global_crops: n_global tensors [B, 3, R, R] (one per view); local_crops: n_local tensors
    [B, 3, r, r].  Crops are stacked view-major ([view 0 of every image | view 1 of every image | ...], the layout
    vtp.py:416-426 chunks and swaps).  Keys = the keyword arguments of VTP.forward_ssl_learning (vtp.py:365-374)."""
    g = torch.cat(list(global_crops), dim=0)
    loc = torch.cat(list(local_crops), dim=0) if len(local_crops) else g.new_zeros((0, 3, patch_size, patch_size))
    grid = (g.shape[-2] // patch_size, g.shape[-1] // patch_size)
    out = collate_ssl_masks(g.shape[0], grid, mask_probability, mask_ratio, rng)
    out.update(global_crops=g, local_crops=loc, n_global_crops=len(global_crops))
    return out
