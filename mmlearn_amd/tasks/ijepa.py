"""``IJEPA`` task with the reference's constructor, attributes, hooks and checkpoint behaviour
(mmlearn/tasks/ijepa.py:24-295), with the path ops on HIP kernels:

* masks come with host-built patch indices (no boolean-mask ``nonzero`` syncs);
* the target path ``layer_norm -> apply_masks -> repeat_interleave_batch`` and the regression loss are ONE
  fused kernel over the predicted patches only (the reference layer-norms all 196 tokens and
  materialises the gathered target);
* the predictor's input sequence is assembled by one gather kernel (``mmlearn_amd.predictor``);
* the EMA teacher update is one multi-tensor kernel (``mmlearn_amd.ema``).
"""

from __future__ import annotations

from functools import partial
from typing import Any, Callable, Optional, Union

import torch
import torch.nn.functional as F  # noqa: N812

from .. import ops
from ..ema import ExponentialMovingAverage
from ..masking import IJEPAMaskGenerator
from ..modalities import Modalities
from ..predictor import is_compatible, predictor_forward
from ..registry import store
from .base import TrainingTask

_FUSED = {None: "smooth_l1", F.smooth_l1_loss: "smooth_l1", F.mse_loss: "mse"}


@store(group="task", name="IJEPAHIP", zen_partial=False)
class IJEPA(TrainingTask):
    """I-JEPA pretraining.  Parameters as in the reference (``encoder``, ``predictor``, ``modality``,
    ``optimizer``, ``lr_scheduler``, ``ema_decay``, ``ema_decay_end``, ``ema_anneal_end_step``, ``loss_fn``,
    ``compute_validation_loss``, ``compute_test_loss``) plus ``true_ema`` (see ``mmlearn_amd.ema``).

    ``loss_fn=None`` / ``F.smooth_l1_loss`` / ``F.mse_loss`` take the fused kernel; any other callable
    ``loss_fn(pred, target)`` receives the HIP-computed target tensor.
    """

    def __init__(
        self,
        encoder: torch.nn.Module,
        predictor: torch.nn.Module,
        modality: str = "RGB",
        optimizer: Optional[partial] = None,
        lr_scheduler: Optional[Union[dict[str, Any], partial]] = None,
        ema_decay: float = 0.996,
        ema_decay_end: float = 1.0,
        ema_anneal_end_step: int = 1000,
        loss_fn: Optional[Callable[[torch.Tensor, torch.Tensor], torch.Tensor]] = None,
        compute_validation_loss: bool = True,
        compute_test_loss: bool = True,
        true_ema: bool = False,
    ):
        super().__init__(optimizer=optimizer, lr_scheduler=lr_scheduler,
                         loss_fn=loss_fn if loss_fn is not None else F.smooth_l1_loss,
                         compute_validation_loss=compute_validation_loss, compute_test_loss=compute_test_loss)
        self._fused_kind = _FUSED.get(loss_fn)
        self.modality = Modalities.get_modality(modality)
        self.mask_generator = IJEPAMaskGenerator()  # default geometry regardless of the encoder, as in the reference (Q10)

        self.encoder = encoder
        self.predictor = predictor
        # attribute-only, like the reference (:96-98): the predictor must already be built to match
        self.predictor.num_patches = encoder.patch_embed.num_patches
        self.predictor.embed_dim = encoder.embed_dim
        self.predictor.num_heads = encoder.num_heads

        self.target_encoder = ExponentialMovingAverage(self.encoder, ema_decay, ema_decay_end, ema_anneal_end_step, true_ema=true_ema)

    def configure_model(self) -> None:
        self.target_encoder.configure_model(self.device)

    def on_before_zero_grad(self, optimizer: torch.optim.Optimizer) -> None:
        """EMA update of the target encoder (after the optimizer step, before zero_grad)."""
        if self.target_encoder is not None:
            self.target_encoder.step(self.encoder)

    def training_step(self, batch: dict[str, Any], batch_idx: int) -> torch.Tensor:
        return self._shared_step(batch, batch_idx, step_type="train")

    def validation_step(self, batch: dict[str, Any], batch_idx: int) -> Optional[torch.Tensor]:
        return self._shared_step(batch, batch_idx, step_type="val")

    def test_step(self, batch: dict[str, Any], batch_idx: int) -> Optional[torch.Tensor]:
        return self._shared_step(batch, batch_idx, step_type="test")

    def on_validation_epoch_start(self) -> None:
        self._on_eval_epoch_start("val")

    def on_validation_epoch_end(self) -> None:
        self._on_eval_epoch_end("val")

    def on_test_epoch_start(self) -> None:
        self._on_eval_epoch_start("test")

    def on_test_epoch_end(self) -> None:
        self._on_eval_epoch_end("test")

    def on_save_checkpoint(self, checkpoint: dict[str, Any]) -> None:
        if self.target_encoder is not None:
            checkpoint["ema_params"] = {"decay": self.target_encoder.decay, "num_updates": self.target_encoder.num_updates}

    def on_load_checkpoint(self, checkpoint: dict[str, Any]) -> None:
        if "ema_params" in checkpoint and self.target_encoder is not None:
            ema_params = checkpoint.pop("ema_params")
            self.target_encoder.decay = ema_params["decay"]
            self.target_encoder.num_updates = ema_params["num_updates"]
            self.target_encoder.restore(self.encoder)

    # ------------------------------------------------------------------ the step
    def _shared_step(self, batch: dict[str, Any], batch_idx: int, step_type: str) -> Optional[torch.Tensor]:
        images = batch[self.modality.name]
        batch_size = images.size(0)
        device = images.device

        mask_info = self.mask_generator(batch_size=batch_size)       # host RNG, same call sequence as the reference
        enc_idx = mask_info["encoder_indices"].to(device, non_blocking=True)      # int32 [nenc, 1, n_ctxt]
        pred_idx = mask_info["predictor_indices"].to(device, non_blocking=True)   # int32 [npred, 1, keep]
        # the masks travel as the reference's lists of 0/1 tensors, with the host-built indices attached: this package's
        # apply_masks (context encoder, vision.py:335-337) gathers by index and never syncs; a foreign consumer sees lists
        encoder_masks = ops.IndexedMasks([m.to(device, non_blocking=True) for m in mask_info["encoder_masks"]], enc_idx)
        predictor_masks = ops.IndexedMasks([m.to(device, non_blocking=True) for m in mask_info["predictor_masks"]], pred_idx)
        n_enc = len(encoder_masks)

        with torch.no_grad():  # teacher sees every patch
            h = self.target_encoder.model(batch)[0]

        batch[self.modality.mask] = encoder_masks   # the reference mutates the shared batch too (Q12)
        z = self.encoder(batch)[0]

        if is_compatible(self.predictor):
            z_pred = predictor_forward(self.predictor, z, enc_idx, pred_idx)
        else:
            z_pred = self.predictor(z, encoder_masks, predictor_masks)

        if step_type == "train":
            self.log("train/ema_decay", self.target_encoder.decay, prog_bar=True)

        if self.loss_fn is not None and (
            step_type == "train" or (step_type == "val" and self.compute_validation_loss)
            or (step_type == "test" and self.compute_test_loss)
        ):
            # target rows are ordered (pred mask, enc mask, sample): one index row per (pred, enc) pair
            idx = pred_idx if n_enc == 1 else pred_idx.repeat_interleave(n_enc, dim=0)
            if self._fused_kind is not None:
                loss = ops.ijepa_loss(z_pred, h, idx, kind=self._fused_kind)
            else:
                loss = self.loss_fn(z_pred, ops.ijepa_target(h, idx))
            self.log(f"{step_type}/loss", loss, prog_bar=True, sync_dist=True)
            return loss
        return None

    def _on_eval_epoch_start(self, step_type: str) -> None:
        if (step_type == "val" and self.compute_validation_loss) or (step_type == "test" and self.compute_test_loss):
            self.log(f"{step_type}/start", 1, prog_bar=True, sync_dist=True)

    def _on_eval_epoch_end(self, step_type: str) -> None:
        if (step_type == "val" and self.compute_validation_loss) or (step_type == "test" and self.compute_test_loss):
            self.log(f"{step_type}/end", 1, prog_bar=True, sync_dist=True)
