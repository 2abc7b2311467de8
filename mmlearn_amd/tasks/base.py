"""``TrainingTask``: the LightningModule surface mmlearn's tasks share (mmlearn/tasks/base.py:15-155).

When ``lightning`` is installed the base class is ``lightning.LightningModule`` and the tasks run
under ``mmlearn_run`` / ``lightning.Trainer`` unchanged.  This image has no Lightning, so a small
stand-in with the handful of members the tasks use (``log``, ``save_hyperparameters``, ``device``,
``trainer``) keeps the same classes usable from a plain training loop (``bench.py``, the tests).
"""

from __future__ import annotations

import inspect
import warnings
from functools import partial
from types import SimpleNamespace
from typing import Any, Optional, Union

import torch
from torch import nn

try:  # drop-in deployment
    import lightning as L  # type: ignore

    LightningModule = L.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # standalone (this image)
    HAVE_LIGHTNING = False

    class LightningModule(nn.Module):  # type: ignore[no-redef]
        """Minimal stand-in: records ``self.log`` calls in ``self.logged``."""

        def __init__(self, *args: Any, **kwargs: Any) -> None:
            super().__init__()
            self.logged: dict[str, Any] = {}
            self.trainer = SimpleNamespace(sanity_checking=False)
            self.hparams: dict[str, Any] = {}

        def save_hyperparameters(self, *args: Any, ignore: Optional[list] = None, **kwargs: Any) -> None:
            frame = inspect.currentframe().f_back
            init_args = {k: v for k, v in frame.f_locals.items() if k not in ("self", "__class__") and k not in (ignore or [])}
            self.hparams = {k: v for k, v in init_args.items() if isinstance(v, (int, float, str, bool, type(None)))}

        def log(self, name: str, value: Any, **kwargs: Any) -> None:
            self.logged[name] = value.detach() if isinstance(value, torch.Tensor) else value

        @property
        def device(self) -> torch.device:
            for p in self.parameters():
                return p.device
            for b in self.buffers():
                return b.device
            return torch.device("cpu")

        def configure_model(self) -> None:
            pass


def rank_zero_warn(msg: str, category: type = UserWarning, **kwargs: Any) -> None:
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()) or torch.distributed.get_rank() == 0:
        warnings.warn(msg, category=category, stacklevel=2)


class TrainingTask(LightningModule):
    """Base class for tasks that require training: optimizer / LR-scheduler plumbing with the
    ``ndim < 2 -> weight_decay = 0`` parameter split (mmlearn/tasks/base.py:72-155)."""

    def __init__(
        self,
        optimizer: Optional[partial] = None,
        lr_scheduler: Optional[Union[dict[str, Any], partial]] = None,
        loss_fn: Optional[Any] = None,
        compute_validation_loss: bool = True,
        compute_test_loss: bool = True,
    ):
        super().__init__()
        if loss_fn is None and (compute_validation_loss or compute_test_loss):
            raise ValueError("Loss function must be provided to compute validation or test loss.")
        self.optimizer = optimizer
        self.lr_scheduler = lr_scheduler
        self.loss_fn = loss_fn
        self.compute_validation_loss = compute_validation_loss
        self.compute_test_loss = compute_test_loss

    def configure_optimizers(self) -> Any:  # noqa: PLR0912
        if self.optimizer is None:
            rank_zero_warn("Optimizer not provided. Training will continue without an optimizer. LR scheduler will not be used.")
            return None

        weight_decay: Optional[float] = self.optimizer.keywords.get("weight_decay", None)
        if weight_decay is None:  # try getting the default value
            kw_param = inspect.signature(self.optimizer.func).parameters.get("weight_decay")
            if kw_param is not None and kw_param.default != inspect.Parameter.empty:
                weight_decay = kw_param.default

        parameters: Any = [p for p in self.parameters() if p.requires_grad]
        if weight_decay is not None:
            decay, no_decay = [], []
            for p in self.parameters():
                if not p.requires_grad:
                    continue
                (no_decay if p.ndim < 2 else decay).append(p)  # biases and normalisation parameters do not decay
            parameters = [
                {"params": decay, "weight_decay": weight_decay, "name": "weight_decay_params"},
                {"params": no_decay, "weight_decay": 0.0, "name": "no_weight_decay_params"},
            ]

        optimizer = self.optimizer(parameters)
        if not isinstance(optimizer, torch.optim.Optimizer):
            raise TypeError(f"Expected optimizer to be an instance of `torch.optim.Optimizer`, but got {type(optimizer)}.")

        if self.lr_scheduler is not None:
            if isinstance(self.lr_scheduler, dict):
                if "scheduler" not in self.lr_scheduler:
                    raise ValueError("Expected 'scheduler' key in the learning rate scheduler dictionary.")
                lr_scheduler = self.lr_scheduler["scheduler"](optimizer)
                if not isinstance(lr_scheduler, torch.optim.lr_scheduler.LRScheduler):
                    raise TypeError("Expected scheduler to be an instance of `torch.optim.lr_scheduler.LRScheduler`, "
                                    f"but got {type(lr_scheduler)}.")
                lr_scheduler_dict: dict[str, Any] = {"scheduler": lr_scheduler}
                if self.lr_scheduler.get("extras"):
                    lr_scheduler_dict.update(self.lr_scheduler["extras"])
                return {"optimizer": optimizer, "lr_scheduler": lr_scheduler_dict}

            lr_scheduler = self.lr_scheduler(optimizer)
            if not isinstance(lr_scheduler, torch.optim.lr_scheduler.LRScheduler):
                raise TypeError("Expected scheduler to be an instance of `torch.optim.lr_scheduler.LRScheduler`, "
                                f"but got {type(lr_scheduler)}.")
            return [optimizer], [lr_scheduler]

        return optimizer


class EvaluationHooks:
    """Hooks for evaluation tasks (mmlearn/tasks/hooks.py:9-61)."""

    def on_evaluation_epoch_start(self, pl_module: Any) -> None:
        """Prepare the evaluation loop."""

    def evaluation_step(self, pl_module: Any, batch: Any, batch_idx: int) -> Optional[dict]:
        rank_zero_warn(f"`evaluation_step` must be implemented to use {self.__class__.__name__} for evaluation.")
        return None

    def on_evaluation_epoch_end(self, pl_module: Any) -> Optional[dict]:
        """Run after the evaluation epoch."""
