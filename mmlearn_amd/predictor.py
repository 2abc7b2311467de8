"""HIP front-end of the I-JEPA predictor (mmlearn/modules/encoders/vision.py:524-569).

The transformer blocks of the predictor are stock modules and stay whatever the user built; what is
on the hot path is the *sequence assembly* around them: two ``pos_embed.repeat(B,1,1)``
materialisations, two boolean-mask gathers (host syncs), ``mask_token.repeat``, ``x.repeat`` and a
``torch.cat`` in the reference -- one gather kernel here -- and the prediction slice.

``predictor_forward`` runs any module that exposes the reference predictor's members
(``predictor_embed``, ``predictor_pos_embed``, ``mask_token``, ``predictor_blocks``,
``predictor_norm``, ``predictor_proj``) through that kernel, so the reference's own
``VisionTransformerPredictor`` (and its weights) can be used unchanged.
"""

from __future__ import annotations

import torch
from torch import nn

from . import ops

_MEMBERS = ("predictor_embed", "predictor_pos_embed", "mask_token", "predictor_blocks", "predictor_norm", "predictor_proj")


def is_compatible(predictor: nn.Module) -> bool:
    return all(hasattr(predictor, m) for m in _MEMBERS)


def predictor_forward(predictor: nn.Module, x: torch.Tensor, enc_idx: torch.Tensor, pred_idx: torch.Tensor) -> torch.Tensor:
    """x: context tokens from the encoder ``[n_enc*B, n_ctxt, D]``; enc_idx ``int32[n_enc, 1|B, n_ctxt]``;
    pred_idx ``int32[n_pred, 1|B, keep]``.  Returns predictions for the mask tokens ``[n_pred*n_enc*B, keep, D]``."""
    n_enc, _, n_ctxt = enc_idx.shape
    b = len(x) // n_enc
    x = predictor.predictor_embed(x)                                             # Linear(D -> Dp)
    seq = ops.predictor_assemble(x, predictor.predictor_pos_embed, predictor.mask_token, enc_idx, pred_idx, b)
    for blk in predictor.predictor_blocks:
        seq = blk(seq)
    seq = predictor.predictor_norm(seq)
    return predictor.predictor_proj(seq[:, n_ctxt:])                             # predictions for the mask tokens only


class HIPPredictor(nn.Module):
    """Module wrapper with the reference predictor's call signature ``(x, masks_x, masks)``."""

    def __init__(self, predictor: nn.Module):
        super().__init__()
        if not is_compatible(predictor):
            raise TypeError(f"predictor must expose {_MEMBERS}")
        self.predictor = predictor
        for name in ("num_patches", "embed_dim", "num_heads"):
            if hasattr(predictor, name):
                setattr(self, name, getattr(predictor, name))

    def forward(self, x: torch.Tensor, masks_x, masks) -> torch.Tensor:
        if not isinstance(masks_x, (list, tuple)):
            masks_x = [masks_x]
        if not isinstance(masks, (list, tuple)):
            masks = [masks]
        b = len(x) // len(masks_x)
        enc_idx = ops.masks_to_indices(masks_x, b, x.device)
        pred_idx = ops.masks_to_indices(masks, b, x.device)
        return predictor_forward(self.predictor, x, enc_idx, pred_idx)
