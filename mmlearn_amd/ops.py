"""Autograd-aware ops of the hot path, each backed by HIP kernels (no ATen math, no CPU path).

=====================  =========================================================================
op                      reference sequence it replaces
=====================  =========================================================================
``l2_normalize``        ``F.normalize(x, p=2, dim=-1)``  tasks/contrastive_pretraining.py:428-429
``masks_to_indices``    the ``nonzero`` hidden in boolean-mask indexing  processors/masking.py:264-283
``apply_masks``         ``apply_masks``  processors/masking.py:241-287
``ijepa_loss``          layer_norm + apply_masks + repeat_interleave_batch + smooth_l1  tasks/ijepa.py:232-238,250-261
``ijepa_target``        the same target path when a user loss_fn consumes ``h_masked``
``predictor_assemble``  modules/encoders/vision.py:545-560
=====================  =========================================================================
"""

from __future__ import annotations

from typing import Optional, Sequence, Union

import numpy as np
import torch

from . import kernels as K

MaskList = Union[torch.Tensor, Sequence[torch.Tensor]]


# ------------------------------------------------------------------ L2 normalise
class _L2Normalize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: torch.Tensor, twin: bool):
        if twin:
            y, inv, y16 = K.l2norm_fwd(x, twin=True)
            ctx.mark_non_differentiable(y16)
            ctx.save_for_backward(x, inv)
            return y, y16
        y, inv = K.l2norm_fwd(x)
        ctx.save_for_backward(x, inv)
        return y

    @staticmethod
    def backward(ctx, dy: torch.Tensor, *_):
        x, inv = ctx.saved_tensors
        return K.l2norm_bwd(x, dy.to(x.dtype), inv), None


def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    """``F.normalize(x, p=2, dim=-1, eps=1e-12)`` on MI355X.  Under bf16/fp16 autocast the result is
    f32, like ``F.normalize`` (an autocast-to-f32 op).  Under bf16 autocast a 2-D result also carries its own rounding to bf16
    as ``y._mmk_bf16_nograd`` = ``(copy, y._version)`` (written by the same kernel): the bf16 similarity kernels of
    :class:`ContrastiveLoss` would round these rows on every read, and the one-launch loss takes the copy instead -- same bits,
    half the bytes.  The copy carries NO gradient, hence a name of its own: ``fused.linear`` and friends pick up
    ``_mmk_bf16`` (``add_layer_norm``'s differentiable twin) and must never see this one; the version stamp lets the loss
    ignore the copy after an in-place edit of ``y``."""
    from . import compiled

    if compiled.is_compiling():   # torch.compile: one traced operator (compiled.py)
        return compiled.l2_normalize(x)
    K.require_gpu(x)
    autocast = torch.is_autocast_enabled()
    if autocast and x.dtype != torch.float32:
        x = x.float()
    if (autocast and torch.get_autocast_dtype("cuda") == torch.bfloat16 and x.dtype == torch.float32 and x.dim() == 2
            and x.shape[1] % 8 == 0):
        y, y16 = _L2Normalize.apply(x, True)
        if not y.is_inference():   # inference tensors (Lightning's validate / test / predict loops) track no version: no stamp, no twin
            y._mmk_bf16_nograd = (y16, y._version)
        return y
    return _L2Normalize.apply(x, False)


# ------------------------------------------------------------------ masks -> indices
class IndexedMasks(list):
    """A list of 0/1 patch masks (what the reference's encoders receive in ``batch["<modality>_mask"]``) that also carries
    the sorted keep-indices of those masks, ``indices`` int32 ``[n_masks, B or 1, keep]`` on the device.  A consumer that
    only knows lists of masks (the reference's ``apply_masks``, masking.py:241-287) sees exactly that; this package's
    ``apply_masks`` reads ``indices`` and never looks at the boolean masks -- no ``nonzero``, no host synchronisation
    (the reference syncs once per mask inside boolean-mask indexing, masking.py:264-283)."""

    def __init__(self, masks, indices: torch.Tensor):
        super().__init__(masks)
        if indices.dim() != 3 or indices.dtype != torch.int32 or indices.shape[0] != len(self):
            raise ValueError(f"indices must be int32 [n_masks={len(self)}, B or 1, keep], got {indices.dtype} {tuple(indices.shape)}")
        self.indices = indices


def masks_to_indices(masks: MaskList, batch_size: int, device: torch.device, keep: Optional[int] = None) -> torch.Tensor:
    """Stack a list of 0/1 masks of shape (N,), (1, N) or (B, N) into sorted keep-indices
    ``int32[n_masks, B or 1, keep]`` on ``device``.

    CPU masks (what ``IJEPAMaskGenerator`` returns) are converted on the host -- no device sync.
    Device masks are converted by a kernel; their keep count must then be given (``keep``) or is read
    back once (a host sync, as in the reference's boolean indexing).
    """
    if isinstance(masks, torch.Tensor):
        masks = [masks]
    out = []
    for m in masks:
        if m.dim() == 1:
            m = m.unsqueeze(0)
        if m.size(0) not in (1, batch_size):
            raise ValueError(f"mask batch dimension {m.size(0)} does not match batch size {batch_size}")
        if not m.is_cuda:
            mb = m.numpy().astype(bool)
            if mb.shape[0] > 1 and (mb == mb[:1]).all():
                mb = mb[:1]  # batch-shared mask (IJEPAMaskGenerator expands one block to the batch)
            counts = mb.sum(1)
            if not (counts == counts[0]).all():
                raise ValueError("all rows of a mask must keep the same number of patches")
            idx = np.stack([np.nonzero(r)[0] for r in mb]).astype(np.int32)
            out.append(torch.from_numpy(idx).to(device, non_blocking=True))
        else:
            k = keep if keep is not None else int(m[0].ne(0).sum().item())
            idx, bad = K.mask_to_index(m, k)
            if keep is None and bool(bad.item()):
                raise ValueError("all rows of a mask must keep the same number of patches")
            out.append(idx)
    keeps = {t.shape[1] for t in out}
    if len(keeps) != 1:
        raise ValueError(f"all masks of one call must keep the same number of patches, got {sorted(keeps)}")
    nb = max(t.shape[0] for t in out)
    out = [t if t.shape[0] == nb else t.expand(nb, -1) for t in out]
    return torch.stack(out).contiguous()


# ------------------------------------------------------------------ apply_masks
class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: torch.Tensor, idx: torch.Tensor):
        ctx.save_for_backward(idx)
        ctx.bn = (x.shape[0], x.shape[1])
        return K.gather_rows(x, idx)

    @staticmethod
    def backward(ctx, dout: torch.Tensor):
        (idx,) = ctx.saved_tensors
        return K.scatter_rows(dout, idx, *ctx.bn), None


def gather_patches(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """x [B, N, D], idx int32 [n_masks, B or 1, keep] (sorted) -> [n_masks*B, keep, D]."""
    K.require_gpu(x)
    return _GatherRows.apply(x, idx)


def apply_masks(x: torch.Tensor, masks: MaskList) -> torch.Tensor:
    """Drop-in for ``mmlearn.datasets.processors.masking.apply_masks``: keep the patches selected by
    each mask and concatenate along the batch dimension.  ``masks`` may be an :class:`IndexedMasks` (or any object with an
    ``indices`` attribute): the precomputed indices are used and nothing is read back from the device; plain CPU masks are
    converted on the host (no device sync either); plain device masks cost one read-back of the keep count, like the
    reference's boolean indexing."""
    idx = getattr(masks, "indices", None)
    if idx is not None:
        if idx.shape[1] not in (1, x.size(0)):
            raise ValueError(f"mask batch dimension {idx.shape[1]} does not match batch size {x.size(0)}")
        if idx.device != x.device:
            idx = idx.to(x.device, non_blocking=True)
    else:
        idx = masks_to_indices(masks, x.size(0), x.device)
    return gather_patches(x, idx)


def repeat_interleave_batch(x: torch.Tensor, b: int, repeat: int) -> torch.Tensor:
    """processors/transforms.py:55-79.  With ``repeat == 1`` (the I-JEPA case, one encoder mask) this is
    the identity and no copy is made; otherwise a view-expand + reshape."""
    if repeat == 1:
        return x
    n = len(x) // b
    return x.reshape(n, 1, b, *x.shape[1:]).expand(n, repeat, b, *x.shape[1:]).reshape(n * repeat * b, *x.shape[1:])


# ------------------------------------------------------------------ fused I-JEPA loss
_KIND = {"smooth_l1": 0, "mse": 1}


class _IJepaLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z: torch.Tensor, h: torch.Tensor, idx: torch.Tensor, kind: int, eps: float):
        loss, _ = K.ijepa_loss_fwd(z, h, idx, kind, eps, want_target=False)
        ctx.save_for_backward(z, h, idx)
        ctx.kind, ctx.eps = kind, eps
        return loss

    @staticmethod
    def backward(ctx, g: torch.Tensor):
        z, h, idx = ctx.saved_tensors
        up = g.detach().to(torch.float32).reshape(1).contiguous()
        return K.ijepa_loss_bwd(z, h, idx, ctx.kind, ctx.eps, up), None, None, None, None


def ijepa_loss(z_pred: torch.Tensor, h: torch.Tensor, pred_idx: torch.Tensor, kind: str = "smooth_l1", eps: float = 1e-5) -> torch.Tensor:
    """``loss(z_pred, layer_norm(h)[pred patches])`` in one pass over the needed rows only.

    z_pred [n_masks*B, keep, D] (requires grad), h [B, N, D] teacher output (no grad),
    pred_idx int32 [n_masks, B or 1, keep].  ``kind``: "smooth_l1" (beta = 1, the reference default,
    tasks/ijepa.py:86) or "mse".
    """
    K.require_gpu(z_pred)
    return _IJepaLoss.apply(z_pred, h.detach(), pred_idx, _KIND[kind], eps)


def ijepa_target(h: torch.Tensor, pred_idx: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """``apply_masks(F.layer_norm(h, h.shape[-1:]), pred_masks)`` computed on the gathered rows only
    (tasks/ijepa.py:232-238); used when a user-supplied ``loss_fn`` consumes the target."""
    K.require_gpu(h)
    n_masks, _, keep = pred_idx.shape
    z0 = torch.zeros((n_masks * h.shape[0], keep, h.shape[2]), dtype=h.dtype, device=h.device)
    _, target = K.ijepa_loss_fwd(z0, h.detach(), pred_idx, 1, eps, want_target=True)
    return target


# ------------------------------------------------------------------ predictor sequence assembly
class _PredAssemble(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pos, tok, enc_idx, pred_idx, b, out_dtype):
        pos32 = pos.detach().reshape(-1, pos.shape[-1]).float().contiguous()
        tok32 = tok.detach().reshape(-1).float().contiguous()
        ctx.dims = (b, enc_idx.shape[0], pred_idx.shape[0], enc_idx.shape[2], pred_idx.shape[2], x.dtype, tok.shape, tok.dtype)
        ctx.needs = (x.requires_grad, tok.requires_grad)
        return K.pred_assemble(x, pos32, tok32, enc_idx, pred_idx, b, out_dtype)

    @staticmethod
    def backward(ctx, dseq):
        b, n_enc, n_pm, n_ctxt, n_pred, x_dtype, tok_shape, tok_dtype = ctx.dims
        dx, dtok = K.pred_assemble_bwd(dseq, b, n_enc, n_pm, n_ctxt, n_pred, x_dtype, ctx.needs_input_grad[0], ctx.needs_input_grad[2])
        if dtok is not None:
            dtok = dtok.reshape(tok_shape).to(tok_dtype)
        return dx, None, dtok, None, None, None, None


def predictor_assemble(x: torch.Tensor, pos_embed: torch.Tensor, mask_token: torch.Tensor, enc_idx: torch.Tensor,
                       pred_idx: torch.Tensor, batch_size: int) -> torch.Tensor:
    """Build the predictor's input sequence (vision.py:545-560) in one kernel:

    ``cat([x + pos[enc patches]] * n_pred_masks , mask_token + pos[pred patches])`` along the token axis,
    for every (prediction mask, context row).  ``x`` is the output of ``predictor_embed``
    ``[n_enc*B, n_ctxt, Dp]``; ``pos_embed`` ``[1, N, Dp]``; ``mask_token`` ``[1, 1, Dp]``.
    Output dtype follows ``torch.cat``'s promotion of (x.dtype, mask_token.dtype).
    """
    K.require_gpu(x)
    out_dtype = torch.promote_types(x.dtype, mask_token.dtype)
    return _PredAssemble.apply(x, pos_embed, mask_token, enc_idx, pred_idx, batch_size, out_dtype)
