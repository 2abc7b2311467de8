"""``IJEPA`` task with the reference's constructor, attributes, hooks and checkpoint behaviour
(mmlearn/tasks/ijepa.py:24-295), with the path ops on HIP kernels:

* masks come with host-built patch indices (no boolean-mask ``nonzero`` syncs);
* the target path ``layer_norm -> apply_masks -> repeat_interleave_batch`` and the regression loss are ONE
  fused kernel over the predicted patches only (the reference layer-norms all 196 tokens and
  materialises the gathered target);
* the predictor's input sequence is assembled by one gather kernel (``mmlearn_amd.predictor``);
* the EMA teacher update is one multi-tensor kernel (``mmlearn_amd.ema``).

Whole-step HIP-graph capture (the reference plumbs ``torch.compile`` at mmlearn/cli/run.py:139 for the same purpose): the step's
only host work that a graph cannot hold is the mask sampling (host RNG, mmlearn/datasets/processors/masking.py:384-387) and the
upload of its results.  :meth:`IJEPA.stage_masks` draws a step's masks AHEAD of the step -- the generator's own call sequence on
the same global RNG -- into pinned memory and uploads them on a side stream; :meth:`IJEPA.commit_masks` moves them into per-shape
static device buffers on the step's stream.  ``training_step`` then reads those buffers: no RNG, no pageable copy, nothing a
capture refuses.  One graph per mask geometry ``(context patches, predicted patches)`` -- the key both calls return.
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
        self._mask_stage = None   # _MaskStage, built by the first stage_masks()

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

    # ------------------------------------------------------------------ masks ahead of the step (graph capture)
    def stage_masks(self, batch_size: int, device: Optional[torch.device] = None) -> tuple:
        """Draw the NEXT step's masks now: ``self.mask_generator(batch_size)`` on the global host RNG (the call the step would make,
        masking.py:384-387 -- G7 stays bit-exact), written into pinned memory and uploaded on a side stream.  Call it while the
        previous step is still running on the device.  Returns the geometry key ``(context patches, predicted patches)``.

        Drawing ahead moves the generator's draws before whatever else reads the global host RNG between the two steps (a new
        epoch's DataLoader seed); within an epoch the sequence is the reference's."""
        device = torch.device(device) if device is not None else self.device
        if self._mask_stage is None or self._mask_stage.device != device:
            self._mask_stage = _MaskStage(device)
        return self._mask_stage.stage(self.mask_generator(batch_size=batch_size), batch_size)

    def commit_masks(self) -> tuple:
        """Make the staged masks the ones the next ``*_step`` reads: the current stream waits for the upload and copies it into the
        static buffers of its geometry (one small device copy).  Outside any capture; required before capturing or replaying a
        step (an eager step commits by itself).  Returns the geometry key."""
        if self._mask_stage is None:
            raise RuntimeError("IJEPA.commit_masks(): nothing staged -- call stage_masks(batch_size) first")
        return self._mask_stage.commit()

    def _masks_for_step(self, batch_size: int, device: torch.device):
        st = self._mask_stage
        capturing = device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        if st is not None and st.device == device and (st.pending is not None or st.ready is not None):
            if st.pending is not None:
                if capturing:
                    raise RuntimeError("IJEPA: staged masks must be committed (commit_masks()) before the step is captured")
                st.commit()
            got = st.take(batch_size)
            if got is not None:
                return got
        if capturing:
            raise RuntimeError("IJEPA: a captured step cannot sample its masks (host RNG, pageable copies): call stage_masks(batch_size) "
                               "and commit_masks() before the capture and before every replay")
        mask_info = self.mask_generator(batch_size=batch_size)       # host RNG, same call sequence as the reference
        enc_idx = mask_info["encoder_indices"].to(device, non_blocking=True)      # int32 [nenc, 1, n_ctxt]
        pred_idx = mask_info["predictor_indices"].to(device, non_blocking=True)   # int32 [npred, 1, keep]
        return (enc_idx, pred_idx, [m.to(device, non_blocking=True) for m in mask_info["encoder_masks"]],
                [m.to(device, non_blocking=True) for m in mask_info["predictor_masks"]])

    # ------------------------------------------------------------------ the step
    def _shared_step(self, batch: dict[str, Any], batch_idx: int, step_type: str) -> Optional[torch.Tensor]:
        images = batch[self.modality.name]
        batch_size = images.size(0)
        device = images.device

        enc_idx, pred_idx, enc_m, pred_m = self._masks_for_step(batch_size, device)
        # the masks travel as the reference's lists of 0/1 tensors, with the host-built indices attached: this package's
        # apply_masks (context encoder, vision.py:335-337) gathers by index and never syncs; a foreign consumer sees lists
        encoder_masks = ops.IndexedMasks(enc_m, enc_idx)
        predictor_masks = ops.IndexedMasks(pred_m, pred_idx)
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


class _MaskStage:
    """Pinned staging + static device buffers of the masks of :class:`IJEPA` steps (one packed int32 buffer per geometry:
    context indices | predicted indices | context 0/1 masks | predicted 0/1 masks).

    ``stage``: host tensors -> one of two pinned slots -> a device staging buffer, on a side stream (event recorded).
    ``commit``: the current stream waits for that event and copies the staging buffer into the STATIC buffer of the geometry --
    the addresses a captured step has baked in.  ``take``: views of the static buffer, shaped as the step wants them."""

    def __init__(self, device: torch.device):
        self.device = device
        self.stream = torch.cuda.Stream(device=device)
        self.slots: list = [None, None]      # [pinned int32, device staging int32, upload event, event after the last commit that read the staging buffer]
        self.turn = 0
        self.pending = None                  # (slot, key, layout, batch_size) staged, not committed
        self.ready = None                    # (key, layout, batch_size) committed, not taken
        self.static: dict = {}               # key -> int32 device buffer

    @staticmethod
    def _layout(info: dict) -> tuple:
        ei, pi = info["encoder_indices"], info["predictor_indices"]
        n_patches = info["encoder_masks"][0].shape[-1]
        return (tuple(ei.shape), tuple(pi.shape), len(info["encoder_masks"]), len(info["predictor_masks"]), n_patches)

    def stage(self, info: dict, batch_size: int) -> tuple:
        layout = self._layout(info)
        key = (layout[0][-1], layout[1][-1])
        # one row per mask: every sample of the batch carries the same block (masking.py:402,409 expand it), so row 0 is the mask
        parts = [info["encoder_indices"].reshape(-1), info["predictor_indices"].reshape(-1)]
        parts += [m[0].reshape(-1) for m in info["encoder_masks"]] + [m[0].reshape(-1) for m in info["predictor_masks"]]
        flat = torch.cat([p.to(torch.int32) for p in parts])
        n = flat.numel()
        k = self.turn
        self.turn ^= 1
        slot = self.slots[k]
        if slot is None or slot[0].numel() < n:
            cap = max(2 * n, 4096)
            slot = [torch.empty(cap, dtype=torch.int32).pin_memory(), torch.empty(cap, dtype=torch.int32, device=self.device), torch.cuda.Event(), None]
            self.slots[k] = slot
        else:
            slot[2].synchronize()   # the upload that last read this pinned slot (two stagings ago) has long finished
        slot[0][:n].copy_(flat)
        with torch.cuda.stream(self.stream):
            if slot[3] is not None:   # the commit (on the step's stream) that last read this device staging buffer comes first
                self.stream.wait_event(slot[3])
            slot[1][:n].copy_(slot[0][:n], non_blocking=True)
            slot[2].record(self.stream)
        self.pending = (k, key, layout, batch_size, n)
        return key

    def commit(self) -> tuple:
        if self.pending is None:
            if self.ready is not None:
                return self.ready[0]
            raise RuntimeError("IJEPA.commit_masks(): nothing staged")
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("IJEPA.commit_masks() must run outside the capture (it waits for an upload made outside it)")
        k, key, layout, batch_size, n = self.pending
        slot = self.slots[k]
        torch.cuda.current_stream(self.device).wait_event(slot[2])
        buf = self.static.get((key, layout))
        if buf is None:
            buf = torch.empty(n, dtype=torch.int32, device=self.device)
            self.static[(key, layout)] = buf
        buf.copy_(slot[1][:n])
        if slot[3] is None:
            slot[3] = torch.cuda.Event()
        slot[3].record(torch.cuda.current_stream(self.device))
        self.pending, self.ready = None, (key, layout, batch_size)
        return key

    def take(self, batch_size: int):
        if self.ready is None:
            return None
        key, layout, staged_for = self.ready
        if staged_for != batch_size:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError(f"IJEPA: masks were staged for batch size {staged_for}, the captured step has {batch_size}")
            self.ready = None
            return None
        if not torch.cuda.is_current_stream_capturing():
            self.ready = None      # one staging serves one step; a capture leaves it for the first replay's commit to replace
        ei_shape, pi_shape, n_enc, n_pred, n_patches = layout
        buf = self.static[(key, layout)]
        o = 0
        n_ei, n_pi = ei_shape[0] * ei_shape[1] * ei_shape[2], pi_shape[0] * pi_shape[1] * pi_shape[2]
        enc_idx = buf[o:o + n_ei].view(ei_shape)
        o += n_ei
        pred_idx = buf[o:o + n_pi].view(pi_shape)
        o += n_pi
        enc_m = [buf[o + i * n_patches: o + (i + 1) * n_patches].view(1, n_patches).expand(batch_size, -1) for i in range(n_enc)]
        o += n_enc * n_patches
        pred_m = [buf[o + i * n_patches: o + (i + 1) * n_patches].view(1, n_patches).expand(batch_size, -1) for i in range(n_pred)]
        return enc_idx, pred_idx, enc_m, pred_m
