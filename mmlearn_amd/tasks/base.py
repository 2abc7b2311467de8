"""``TrainingTask``: the LightningModule surface the tasks of this package sit on.

Drop-in deployment (``mmlearn`` importable): the base class IS ``mmlearn.tasks.base.TrainingTask`` and the hooks
base IS ``mmlearn.tasks.hooks.EvaluationHooks`` -- nothing of the reference's optimizer / scheduler plumbing
(mmlearn/tasks/base.py:72-155) is restated here, the tasks inherit it.

Stand-alone (this image: no ``lightning``, no ``hydra_zen``, hence no importable ``mmlearn``): a small stand-in
module with the members the tasks touch (``log``, ``save_hyperparameters``, ``device``, ``trainer``) and a minimal
``configure_optimizers`` that keeps the one behaviour the hot path depends on -- parameters with fewer than two
dimensions are exempt from weight decay -- so the same task classes run from a plain loop (``bench.py``, the tests).
"""

from __future__ import annotations

import inspect
import warnings
from types import SimpleNamespace
from typing import Any, Callable, Optional

import torch
from torch import nn


def rank_zero_warn(msg: str, category: type = UserWarning, **kwargs: Any) -> None:
    dist = torch.distributed
    if not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0:
        warnings.warn(msg, category=category, stacklevel=2)


try:  # drop-in deployment: inherit the reference's own base classes
    from mmlearn.tasks.base import TrainingTask as _ReferenceTrainingTask  # type: ignore
    from mmlearn.tasks.hooks import EvaluationHooks as _ReferenceEvaluationHooks  # type: ignore

    HAVE_MMLEARN = True
except Exception:  # stand-alone
    HAVE_MMLEARN = False

try:
    import lightning as L  # type: ignore

    HAVE_LIGHTNING = True
except Exception:
    HAVE_LIGHTNING = False


if HAVE_MMLEARN:

    class TrainingTask(_ReferenceTrainingTask):  # type: ignore[misc,valid-type]
        """mmlearn's ``TrainingTask`` itself (a ``lightning.LightningModule``); never executed in this image."""

    class EvaluationHooks(_ReferenceEvaluationHooks):  # type: ignore[misc,valid-type]
        """mmlearn's ``EvaluationHooks`` itself."""

else:

    class _ModuleStandIn(nn.Module):
        """What the tasks use of a LightningModule when there is no Lightning: ``log`` records into ``logged``."""

        def __init__(self) -> None:
            super().__init__()
            self.logged: dict[str, Any] = {}
            self.trainer = SimpleNamespace(sanity_checking=False)
            self.hparams: dict[str, Any] = {}

        def save_hyperparameters(self, *args: Any, ignore: Optional[list] = None, **kwargs: Any) -> None:
            caller = inspect.currentframe().f_back.f_locals
            skip = {"self", "__class__", *(ignore or [])}
            self.hparams = {k: v for k, v in caller.items()
                            if k not in skip and isinstance(v, (int, float, str, bool, type(None)))}

        def log(self, name: str, value: Any, **kwargs: Any) -> None:
            self.logged[name] = value.detach() if isinstance(value, torch.Tensor) else value

        @property
        def device(self) -> torch.device:
            first = next(iter(self.parameters()), None)
            if first is None:
                first = next(iter(self.buffers()), None)
            return first.device if first is not None else torch.device("cpu")

        def configure_model(self) -> None:
            pass

    _Base = L.LightningModule if HAVE_LIGHTNING else _ModuleStandIn

    def _weight_decay_of(factory: Any) -> Optional[float]:
        """The decay a ``functools.partial`` optimizer factory will run with: bound keyword, else the class default."""
        bound = getattr(factory, "keywords", {}) or {}
        if bound.get("weight_decay") is not None:
            return bound["weight_decay"]
        sig = inspect.signature(getattr(factory, "func", factory)).parameters.get("weight_decay")
        return None if sig is None or sig.default is inspect.Parameter.empty else sig.default

    def _param_groups(module: nn.Module, weight_decay: Optional[float]) -> list:
        """Trainable parameters, split so that vectors and scalars (biases, norm gains, the logit scale) do not decay."""
        trainable = [p for p in module.parameters() if p.requires_grad]
        if weight_decay is None:
            return trainable
        return [
            {"name": "weight_decay_params", "weight_decay": weight_decay, "params": [p for p in trainable if p.ndim >= 2]},
            {"name": "no_weight_decay_params", "weight_decay": 0.0, "params": [p for p in trainable if p.ndim < 2]},
        ]

    def _checked(obj: Any, kind: type, what: str) -> Any:
        if not isinstance(obj, kind):
            raise TypeError(f"{what} factory returned {type(obj)}, not a `{kind.__module__}.{kind.__qualname__}`.")
        return obj

    class TrainingTask(_Base):  # type: ignore[misc,valid-type,no-redef]
        """Stand-alone base: same constructor and the same ``configure_optimizers`` return shapes Lightning accepts."""

        def __init__(self, optimizer: Optional[Callable] = None, lr_scheduler: Any = None, loss_fn: Optional[Any] = None,
                     compute_validation_loss: bool = True, compute_test_loss: bool = True):
            super().__init__()
            if loss_fn is None and (compute_validation_loss or compute_test_loss):
                raise ValueError("Loss function must be provided to compute validation or test loss.")
            self.optimizer, self.lr_scheduler, self.loss_fn = optimizer, lr_scheduler, loss_fn
            self.compute_validation_loss, self.compute_test_loss = compute_validation_loss, compute_test_loss

        def configure_optimizers(self) -> Any:
            if self.optimizer is None:
                rank_zero_warn("No optimizer given: the task trains without one and ignores any LR scheduler.")
                return None
            opt = _checked(self.optimizer(_param_groups(self, _weight_decay_of(self.optimizer))),
                           torch.optim.Optimizer, "optimizer")
            spec = self.lr_scheduler
            if spec is None:
                return opt
            if not isinstance(spec, dict):  # a bare factory: Lightning's two-list form
                return [opt], [_checked(spec(opt), torch.optim.lr_scheduler.LRScheduler, "lr_scheduler")]
            if "scheduler" not in spec:
                raise ValueError("An lr_scheduler dictionary needs a 'scheduler' entry (the factory).")
            sched = _checked(spec["scheduler"](opt), torch.optim.lr_scheduler.LRScheduler, "lr_scheduler")
            return {"optimizer": opt, "lr_scheduler": {"scheduler": sched, **(spec.get("extras") or {})}}

    class EvaluationHooks:  # type: ignore[no-redef]
        """Stand-alone hooks base (interface of mmlearn/tasks/hooks.py:9-61): three callbacks, all optional."""

        def on_evaluation_epoch_start(self, pl_module: Any) -> None:
            return None

        def evaluation_step(self, pl_module: Any, batch: Any, batch_idx: int) -> Optional[dict]:
            rank_zero_warn(f"{type(self).__name__} defines no `evaluation_step`; it contributes nothing to evaluation.")
            return None

        def on_evaluation_epoch_end(self, pl_module: Any) -> Optional[dict]:
            return None
