"""Thin layers usable as heads / postprocessors of the contrastive task
(mmlearn/modules/layers/logit_scaling.py:9-54, normalization.py:7-34)."""

from __future__ import annotations

import numpy as np
import torch

from .ops import l2_normalize
from .registry import store


@store(group="modules/layers", name="LearnableLogitScalingHIP")
class LearnableLogitScaling(torch.nn.Module):
    """``clip(exp(log_logit_scale), max=max_logit_scale) * x``.  Note: placed in
    ``heads``/``postprocessors`` of ContrastivePretraining it is cancelled by the L2 normalisation that
    follows; the effective temperature is the task's own ``log_logit_scale``."""

    def __init__(self, init_logit_scale: float = 1 / 0.07, max_logit_scale: float = 100, learnable: bool = True) -> None:
        super().__init__()
        self.max_logit_scale = max_logit_scale
        self.init_logit_scale = init_logit_scale
        self.learnable = learnable
        log_logit_scale = torch.ones([]) * np.log(self.init_logit_scale)
        if learnable:
            self.log_logit_scale = torch.nn.Parameter(log_logit_scale)
        else:
            self.register_buffer("log_logit_scale", log_logit_scale)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return torch.clip(self.log_logit_scale.exp(), max=self.max_logit_scale) * x

    def extra_repr(self) -> str:
        kind = "parameter" if self.learnable else "buffer"
        return f"init={self.init_logit_scale:g}, cap={self.max_logit_scale:g}, log-scale held as a {kind}"


@store(group="modules/layers", name="L2NormHIP")
class L2Norm(torch.nn.Module):
    """L2 normalisation along ``dim`` with the HIP row kernel (last dim; other dims are moved there)."""

    def __init__(self, dim: int) -> None:
        super().__init__()
        self.dim = dim

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.dim in (-1, x.ndim - 1):
            return l2_normalize(x)
        return l2_normalize(x.movedim(self.dim, -1)).movedim(-1, self.dim)
