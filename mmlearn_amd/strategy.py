"""``TowerDDPStrategy``: the Lightning strategy that makes the per-tower schedule reachable from ``mmlearn_run``.

The reference builds its ``lightning.Trainer`` from ``cfg.trainer`` (mmlearn/cli/run.py:52-61), so a strategy is selected in
YAML.  Stock ``DDPStrategy`` wraps the whole LightningModule in ONE ``DistributedDataParallel``; with
``task.concurrent_encoders`` (one HIP stream per tower) that single instance serialises the towers' backward passes again
(HISTORY.md 5.9).  This subclass hands the wrapping to the task -- ``ContrastivePretraining.wrap_towers_in_ddp()``: one DDP
instance per tower, built under that tower's stream, plus an all-reduce hook for the parameters outside the towers -- and
raises ``GPU_MAX_HW_QUEUES`` before the first HIP call so that the towers' streams and RCCL's do not share a hardware queue.

    trainer:
      strategy: tower_ddp            # registered in Lightning's StrategyRegistry on import of mmlearn_amd
      # or, explicitly:  strategy: {_target_: mmlearn_amd.strategy.TowerDDPStrategy, gradient_as_bucket_view: true}

Choosing this strategy IS the request for the per-tower schedule: it switches ``task.concurrent_encoders`` on (the task's
constructor also takes ``concurrent_encoders`` / ``max_side_streams`` / ``match_ahead`` from YAML); pass
``concurrent_encoders: false`` to the strategy to keep a task's own setting.  Tasks without ``wrap_towers_in_ddp`` (or with
``concurrent_encoders`` off) get stock DDP behaviour.  The DDP instances are kept outside the task's module tree, so
checkpoints keep the reference's ``encoders.<m>.*`` keys.  Lightning is not
installed in the build image: the class is import-guarded, and the part that does not depend on Lightning
(``setup_towers``) is what the tests drive against a stand-in base class.
"""

from __future__ import annotations

import os
from typing import Any

import torch

try:  # drop-in deployment
    from lightning.pytorch.strategies import DDPStrategy, StrategyRegistry  # type: ignore

    HAVE_LIGHTNING = True
except Exception:  # this image
    HAVE_LIGHTNING = False
    StrategyRegistry = None

    class DDPStrategy:  # type: ignore[no-redef]
        """The members of ``lightning.pytorch.strategies.DDPStrategy`` this module touches (stand-in, tests only)."""

        def __init__(self, **ddp_kwargs: Any) -> None:
            self._ddp_kwargs = ddp_kwargs
            self.model = None

        def determine_ddp_device_ids(self):
            return [torch.cuda.current_device()] if torch.cuda.is_available() else None

        def _setup_model(self, model: torch.nn.Module):
            from torch.nn.parallel import DistributedDataParallel

            return DistributedDataParallel(model, device_ids=self.determine_ddp_device_ids(), **self._ddp_kwargs)

        def _register_ddp_hooks(self) -> None:
            pass


def _unwrap(model: Any) -> Any:
    """The LightningModule behind Lightning's forward-redirection wrappers."""
    seen = 0
    while seen < 4 and not hasattr(model, "wrap_towers_in_ddp"):
        inner = getattr(model, "_forward_module", None) or getattr(model, "module", None)
        if inner is None:
            break
        model, seen = inner, seen + 1
    return model


class TowerDDPStrategy(DDPStrategy):
    strategy_name = "tower_ddp"

    def __init__(self, *args: Any, concurrent_encoders: bool = True, **kwargs: Any) -> None:
        self._want_towers = bool(concurrent_encoders)
        if self._want_towers:
            # must precede the first HIP call of the process (HIP multiplexes streams onto this many hardware queues; with
            # the default 4, two of {tower 1, tower 2, RCCL} can share one and the towers' overlap is lost: 216 vs 207.6
            # ms/step).  Only set when towers will actually be wrapped.
            os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
        kwargs.setdefault("gradient_as_bucket_view", True)
        super().__init__(*args, **kwargs)
        self._towers_wrapped = False

    def setup_towers(self, model: Any) -> bool:
        """Wrap the task's towers if it can and wants to; True when done (then no outer DDP must be built)."""
        task = _unwrap(model)
        if not hasattr(task, "wrap_towers_in_ddp"):
            return False
        if self._want_towers and not getattr(task, "concurrent_encoders", False):
            # choosing this strategy is the request for the per-tower schedule; say so when it overrides the task's own setting
            import warnings

            warnings.warn("TowerDDPStrategy switches task.concurrent_encoders on (one stream and one DDP instance per tower); pass "
                          "concurrent_encoders=False to the strategy to keep the task's setting", RuntimeWarning, stacklevel=2)
            task.concurrent_encoders = True
        if not getattr(task, "concurrent_encoders", False):
            return False
        kw = {k: v for k, v in dict(getattr(self, "_ddp_kwargs", {}) or {}).items() if k != "device_ids"}
        task.wrap_towers_in_ddp(**kw)
        self._towers_wrapped = True
        return True

    def _setup_model(self, model: Any) -> Any:
        if self.setup_towers(model):
            return model        # Strategy.training_step calls the LightningModule directly when self.model is the module
        return super()._setup_model(model)

    def _register_ddp_hooks(self) -> None:
        if self._towers_wrapped:
            return              # comm hooks belong to a single outer DDP instance; the towers use DDP's default all-reduce
        super()._register_ddp_hooks()


if HAVE_LIGHTNING:  # ``trainer.strategy: tower_ddp`` under the reference CLI
    try:
        StrategyRegistry.register(TowerDDPStrategy.strategy_name, TowerDDPStrategy,
                                  description="DDP with one DistributedDataParallel instance per encoder tower (mmlearn_amd)")
    except Exception:
        pass
