"""I-JEPA block-mask sampling (host side) with the reference's RNG call sequence.

``IJEPAMaskGenerator`` mirrors mmlearn/datasets/processors/masking.py:290-415, including its
observable quirks: one block per mask shared by the whole batch, scale and
aspect ratio driven by the same uniform sample, ``allow_overlap`` / ``min_keep`` accepted but unused,
block origins drawn from the GLOBAL torch RNG with ``randint(0, H - h)``.

Besides the reference's ``{"encoder_masks", "predictor_masks"}`` (int32 0/1 tensors expanded to the
batch) the result carries ``encoder_indices`` / ``predictor_indices``: the sorted kept-patch indices
``int32[n_masks, 1, keep]`` that the HIP gather kernels consume -- built on the host from the block
rectangle, so the device never runs a ``nonzero`` (the reference's per-mask host sync).
"""

from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Any

import torch

from .registry import store


@store(group="datasets/masking", name="IJEPAMaskGeneratorHIP")
@dataclass
class IJEPAMaskGenerator:
    input_size: tuple[int, int] = (224, 224)
    patch_size: int = 16
    min_keep: int = 10
    allow_overlap: bool = False
    enc_mask_scale: tuple[float, float] = (0.85, 1.0)
    pred_mask_scale: tuple[float, float] = (0.15, 0.2)
    aspect_ratio: tuple[float, float] = (0.75, 1.5)
    nenc: int = 1
    npred: int = 4

    def __post_init__(self) -> None:
        self.height = self.input_size[0] // self.patch_size
        self.width = self.input_size[1] // self.patch_size

    def _block_size(self, u: float, scale: tuple[float, float], aspect: tuple[float, float]) -> tuple[int, int]:
        # ONE uniform sample drives both the scale and the aspect ratio (reference :340-346)
        keep = int(self.height * self.width * (scale[0] + u * (scale[1] - scale[0])))
        ar = aspect[0] + u * (aspect[1] - aspect[0])
        h = min(int(round(math.sqrt(keep * ar))), self.height - 1)
        w = min(int(round(math.sqrt(keep / ar))), self.width - 1)
        return h, w

    def _place(self, hw: tuple[int, int]) -> tuple[int, int, int, int]:
        h, w = hw
        top = int(torch.randint(0, self.height - h, (1,)).item())    # global RNG, last legal origin excluded
        left = int(torch.randint(0, self.width - w, (1,)).item())
        return top, left, h, w

    def _render(self, rect: tuple[int, int, int, int], batch_size: int):
        top, left, h, w = rect
        mask = torch.zeros((self.height, self.width), dtype=torch.int32)
        mask[top: top + h, left: left + w] = 1
        rows = torch.arange(top, top + h, dtype=torch.int32).unsqueeze(1) * self.width
        idx = (rows + torch.arange(left, left + w, dtype=torch.int32).unsqueeze(0)).reshape(1, -1)
        return mask.flatten().unsqueeze(0).expand(batch_size, -1), idx

    def __call__(self, batch_size: int = 1) -> dict[str, Any]:
        seed = int(torch.randint(0, 2**32, (1,)).item())
        g = torch.Generator().manual_seed(seed)
        p_size = self._block_size(torch.rand(1, generator=g).item(), self.pred_mask_scale, self.aspect_ratio)
        e_size = self._block_size(torch.rand(1, generator=g).item(), self.enc_mask_scale, (1.0, 1.0))
        pred = [self._render(self._place(p_size), batch_size) for _ in range(self.npred)]
        enc = [self._render(self._place(e_size), batch_size) for _ in range(self.nenc)]
        return {
            "encoder_masks": [m for m, _ in enc],
            "predictor_masks": [m for m, _ in pred],
            "encoder_indices": torch.stack([i for _, i in enc]),      # int32 [nenc, 1, keep_enc]
            "predictor_indices": torch.stack([i for _, i in pred]),   # int32 [npred, 1, keep_pred]
        }
