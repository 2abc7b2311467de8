"""``ContrastivePretraining`` task with the reference's constructor, attributes and hooks
(mmlearn/tasks/contrastive_pretraining.py:87-701), driving the HIP loss path.

Differences from the reference, all inside the hot path:

* ``encode(..., normalize=True)`` normalises with the HIP ``l2_normalize`` op;
* ``forward`` hands every finished embedding to ``loss.prefetch_gather`` (when the loss has one): the
  global-batch all-gather of modality k runs on the RCCL stream while the encoder of modality k+1 runs;
* ``loss`` is any module with the documented ``forward(embeddings, example_ids, logit_scale,
  modality_loss_pairs)`` signature, as in the reference (use :class:`~mmlearn_amd.losses.ContrastiveLoss`).
"""

from __future__ import annotations

import contextlib
import copy
import inspect

import itertools
import math
from dataclasses import dataclass
from functools import partial
from typing import Any, Literal, Mapping, Optional, Union

import numpy as np
import torch
from torch import nn

from ..compiled import is_compiling, is_opaque_tower, tower_forward
from ..losses import ContrastiveLoss, LossPairSpec
from ..modalities import Modalities
from ..ops import l2_normalize
from ..registry import store
from .base import EvaluationHooks, TrainingTask

__all__ = ["ModuleKeySpec", "LossPairSpec", "AuxiliaryTaskSpec", "EvaluationSpec", "ContrastivePretraining"]


@dataclass
class ModuleKeySpec:
    """Which encoder / head / postprocessor key serves a modality (reference :30-41)."""

    encoder_key: Optional[str] = None
    head_key: Optional[str] = None
    postprocessor_key: Optional[str] = None


@dataclass
class AuxiliaryTaskSpec:
    """An auxiliary task trained on one modality's encoder (reference :55-70): ``task`` is a
    ``functools.partial`` that receives the initialised encoder."""

    modality: str
    task: Any
    loss_weight: float = 1.0


@dataclass
class EvaluationSpec:
    """An evaluation task (``EvaluationHooks``) and when it runs (reference :73-85)."""

    task: Any
    run_on_validation: bool = True
    run_on_test: bool = True


def _unsupported(modality: str) -> ValueError:
    return ValueError(
        f"Found unsupported modality `{modality}` in the input. Supported modalities are {Modalities.list_modalities()}."
        "HINT: New modalities can be added with `Modalities.register_modality` method.")


def _as_module(entry: Union[nn.Module, Mapping[str, nn.Module]]) -> nn.Module:
    # a dict of modules becomes a Sequential over the SAME instances (shared parameters), reference :297-307
    return entry if isinstance(entry, nn.Module) else nn.Sequential(*entry.values())


@store(group="task", name="ContrastivePretrainingHIP")
class ContrastivePretraining(TrainingTask):
    """N-modality contrastive pretraining (see the reference docstring for the parameter semantics)."""

    def __init__(  # noqa: PLR0912
        self,
        encoders: dict[str, nn.Module],
        heads: Optional[dict[str, Union[nn.Module, dict[str, nn.Module]]]] = None,
        postprocessors: Optional[dict[str, Union[nn.Module, dict[str, nn.Module]]]] = None,
        modality_module_mapping: Optional[dict[str, ModuleKeySpec]] = None,
        optimizer: Optional[partial] = None,
        lr_scheduler: Optional[Union[dict[str, Any], partial]] = None,
        init_logit_scale: float = 1 / 0.07,
        max_logit_scale: float = 100,
        learnable_logit_scale: bool = True,
        loss: Optional[nn.Module] = None,
        modality_loss_pairs: Optional[list[LossPairSpec]] = None,
        auxiliary_tasks: Optional[dict[str, AuxiliaryTaskSpec]] = None,
        log_auxiliary_tasks_loss: bool = False,
        compute_validation_loss: bool = True,
        compute_test_loss: bool = True,
        evaluation_tasks: Optional[dict[str, EvaluationSpec]] = None,
        concurrent_encoders: bool = False,
        max_side_streams: int = 1,
        match_ahead: bool = True,
    ) -> None:
        """The last three parameters are additions to the reference signature (defaults = reference behaviour), settable
        from YAML under ``mmlearn_run``: ``concurrent_encoders`` runs every tower after the first on a side HIP stream
        (HISTORY.md 5.9), ``max_side_streams`` caps the number of such streams, ``match_ahead`` launches the id matcher before
        the encoders."""
        super().__init__(optimizer=optimizer, lr_scheduler=lr_scheduler, loss_fn=loss,
                         compute_validation_loss=compute_validation_loss, compute_test_loss=compute_test_loss)
        self.save_hyperparameters(ignore=["encoders", "heads", "postprocessors", "modality_module_mapping", "loss",
                                          "auxiliary_tasks", "evaluation_tasks", "modality_loss_pairs"])
        self.concurrent_encoders = bool(concurrent_encoders)
        self.max_side_streams = int(max_side_streams)
        self.match_ahead = bool(match_ahead)

        if modality_module_mapping is None:  # all module dicts are keyed by modality
            modality_module_mapping = {k: ModuleKeySpec(encoder_key=k, head_key=k, postprocessor_key=k) for k in encoders}

        enc_of: dict[str, Optional[str]] = {}
        head_of: dict[str, Optional[str]] = {}
        post_of: dict[str, Optional[str]] = {}
        for modality_key, spec in modality_module_mapping.items():
            if not Modalities.has_modality(modality_key):
                raise _unsupported(modality_key)
            enc_of[modality_key], head_of[modality_key], post_of[modality_key] = spec.encoder_key, spec.head_key, spec.postprocessor_key

        # every provided module must end up mapped to a modality
        for table, modules in ((enc_of, encoders), (head_of, heads), (post_of, postprocessors)):
            for key in modules or {}:
                if key not in table.values():
                    if not Modalities.has_modality(key):
                        raise _unsupported(key)
                    table[key] = key

        self._available_modalities = [Modalities.get_modality(k) for k in enc_of]
        assert len(self._available_modalities) >= 2, "Expected at least two modalities to be available. "

        def name(k: str) -> str:
            return Modalities.get_modality(k).name

        #: encoders keyed by modality name
        self.encoders = nn.ModuleDict({name(k): encoders[ek] for k, ek in enc_of.items() if ek is not None})
        #: projection heads keyed by modality name (or None)
        self.heads = None if heads is None else nn.ModuleDict(
            {name(k): _as_module(heads[hk]) for k, hk in head_of.items() if hk is not None and hk in heads})
        #: postprocessors keyed by modality name (or None)
        self.postprocessors = None if postprocessors is None else nn.ModuleDict(
            {name(k): _as_module(postprocessors[pk]) for k, pk in post_of.items() if pk is not None and pk in postprocessors})

        # logit scale: log-space scalar, Parameter when learnable (reference :327-337)
        log_logit_scale = torch.ones([]) * np.log(init_logit_scale)
        self.max_logit_scale = max_logit_scale
        self.learnable_logit_scale = learnable_logit_scale
        if learnable_logit_scale:
            self.log_logit_scale = nn.Parameter(log_logit_scale, requires_grad=True)
        else:
            self.register_buffer("log_logit_scale", log_logit_scale)

        if modality_loss_pairs is None:
            modality_loss_pairs = [LossPairSpec(modalities=(m1.name, m2.name))
                                   for m1, m2 in itertools.combinations(self._available_modalities, 2)]
        for pair in modality_loss_pairs:
            if not all(Modalities.get_modality(m) in self._available_modalities for m in pair.modalities):
                raise ValueError(f"Found unspecified modality in the loss pair specification {pair.modalities}. "
                                 f"Available modalities are {self._available_modalities}.")
        #: pairs of modalities (and weights) the contrastive loss is computed between
        self.modality_loss_pairs = modality_loss_pairs

        self.aux_task_specs = auxiliary_tasks or {}
        self.auxiliary_tasks = nn.ModuleDict()
        for task_name, spec in self.aux_task_specs.items():
            if not Modalities.has_modality(spec.modality):
                raise ValueError(f"Found unsupported modality `{spec.modality}` in the auxiliary tasks. "
                                 f"Available modalities are {self._available_modalities}.")
            if not isinstance(spec.task, partial):
                raise TypeError(f"Expected auxiliary task to be a partial function, but got {type(spec.task)}.")
            self.auxiliary_tasks[task_name] = spec.task(self.encoders[name(spec.modality)])
        self.log_auxiliary_tasks_loss = log_auxiliary_tasks_loss

        for spec in (evaluation_tasks or {}).values():
            if not isinstance(spec.task, EvaluationHooks) and not _is_eval_hooks(spec.task):
                raise TypeError(f"Expected {spec.task} to be an instance of `EvaluationHooks` but got {type(spec.task)}.")
        self.evaluation_tasks = evaluation_tasks

    # ------------------------------------------------------------------ model
    def configure_model(self) -> None:
        for task in self.auxiliary_tasks.values():
            task.configure_model()

    def wrap_towers_in_ddp(self, **ddp_kwargs: Any) -> None:
        """DistributedDataParallel for a task with ``concurrent_encoders``: one DDP instance PER TOWER (encoder, and that
        modality's postprocessor / head if they have parameters), each constructed under the stream its tower runs on,
        instead of one DDP around the whole task.  A module shared between towers (a common projection head) is wrapped once,
        for the first tower that uses it; the other towers call it unwrapped and autograd sums their contributions first.

        Why: autograd runs a parameter's gradient accumulation -- and DDP's bucket hooks with it -- on the stream that was
        current when the accumulator node was created, and DDP's constructor creates them all.  With a single outer DDP
        they all sit on one stream: every gradient of a side-stream tower then makes that stream wait for the tower, the
        backward passes run one after the other (measured: 152 vs 140 ms), and a gradient bucket can mix gradients written
        on two streams while its all-reduce only waits for one.  Per-tower instances keep each tower's accumulation,
        buckets and collectives on that tower's stream.  Parameters outside the towers (``log_logit_scale``) get a
        hook that all-reduces their gradient directly.  Call once, after ``.to(device)`` and before the first step.

        The DDP instances live OUTSIDE the registered module tree (``self._tower_ddp``, looked up by ``encode``): the
        ModuleDicts keep the plain modules, so ``state_dict()`` / checkpoints keep the reference's key space
        (``encoders.<m>.*``, no ``.module.`` segment) and a checkpoint written by a per-tower run loads into an unwrapped task
        with ``strict=True`` and vice versa."""
        import torch.distributed as dist
        from torch.nn.parallel import DistributedDataParallel as DDP

        world = dist.get_world_size()
        mods = list(self._available_modalities)
        side = self._encoder_streams(len(mods) - 1) if self.concurrent_encoders else []
        dev = next(self.parameters()).device
        kw: dict = dict(gradient_as_bucket_view=True)
        if dev.type == "cuda":
            kw["device_ids"] = [dev.index]
        kw.update(ddp_kwargs)
        owned: set[int] = set()   # parameters that already belong to a DDP instance (modules can be shared between towers)
        wrapped: dict = self.__dict__.setdefault("_tower_ddp", {})   # (group name, modality) -> DDP; not a registered submodule
        for k, m in enumerate(mods):
            ctx = torch.cuda.stream(side[k - 1]) if (side and k) else contextlib.nullcontext()
            with ctx:
                for gname, group in (("encoders", self.encoders), ("postprocessors", self.postprocessors), ("heads", self.heads)):
                    if not (group and m.name in group):
                        continue
                    mine = {id(p) for p in group[m.name].parameters() if p.requires_grad}
                    if not mine or mine <= owned:
                        continue   # nothing to reduce, or a module shared with an earlier tower: that tower's instance reduces it
                    if mine & owned:
                        raise NotImplementedError(f"{m.name}: module shares only part of its parameters with another tower")
                    wrapped[(gname, m.name)] = DDP(group[m.name], **kw)
                    owned |= mine
        tower_params = {id(p) for g in (self.encoders, self.postprocessors, self.heads) if g for p in g.parameters()}

        def _reduce(p: torch.Tensor) -> None:
            dist.all_reduce(p.grad)
            p.grad.div_(world)

        for p in self.parameters():
            if p.requires_grad and id(p) not in tower_params:
                p.register_post_accumulate_grad_hook(_reduce)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)   # the constructors broadcast rank 0's weights on the towers' streams

    def encode(self, inputs: dict[str, Any], modality: Any, normalize: bool = False) -> torch.Tensor:
        """encoder -> postprocessor -> head -> (optional) L2 normalisation (reference :400-431)."""
        tower = self._tower("encoders", modality.name)
        if is_compiling() and is_opaque_tower(tower):
            # an encoder patched by accelerate_encoder is ONE operator to the tracer (mmlearn_amd/compiled.py): its kernels sit behind
            # ctypes, which TorchDynamo cannot trace.  (A per-tower DDP wrapper is not such a module: it is traced as it is, graph
            # breaks and all -- its reducer hooks live on gradient accumulation, which the operator's backward bypasses.)
            output = tower_forward(tower, inputs)
        else:
            output = tower(inputs)[0]
        if self.postprocessors and modality.name in self.postprocessors:
            output = self._tower("postprocessors", modality.name)(output)
        if self.heads and modality.name in self.heads:
            output = self._tower("heads", modality.name)(output)
        if normalize:
            output = l2_normalize(output)
        return output

    def _tower(self, group: str, name: str) -> nn.Module:
        """The module ``encode`` calls for ``group[name]``: its per-tower DDP wrapper if ``wrap_towers_in_ddp`` made one -- and it still
        wraps the module that is registered now (a tower replaced after wrapping must not silently keep training the old one)."""
        module = getattr(self, group)[name]
        ddp = (self.__dict__.get("_tower_ddp") or {}).get((group, name))
        if ddp is None:
            return module
        if ddp.module is not module:
            raise RuntimeError(f"{group}[{name!r}] was replaced after wrap_towers_in_ddp(): its DistributedDataParallel wrapper still holds "
                               "the old module; call wrap_towers_in_ddp() again on a fresh task")
        if ddp.training != module.training:   # the wrappers sit outside the module tree: train() / eval() do not reach them
            ddp.train(module.training)
        return ddp

    def __deepcopy__(self, memo):
        """The per-tower DDP wrappers and side streams are bound to a process group / device queue: a copy of the task starts without
        them (call ``wrap_towers_in_ddp`` on the copy if it is to train under DDP)."""
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k in ("_tower_ddp", "_side_streams"):
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        from ..compiled import is_opaque_tower, mark_tower

        for enc in (new.__dict__.get("_modules", {}).get("encoders") or {}).values():   # accelerated towers: one operator each under
            if is_opaque_tower(enc):                                                     # torch.compile, keyed per module object
                mark_tower(enc)
        return new

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("_tower_ddp", None)
        state.pop("_side_streams", None)
        return state

    def _encoder_streams(self, n: int) -> list:
        """The side stream of tower k >= 1 is entry k - 1.  At most ``max_side_streams`` (default 1) distinct streams are
        created and towers share them round-robin: two compute streams give the overlap (the tails of one tower's kernels
        filled by the other's), more streams multiply the cross-stream waits that the runtime maps onto a handful of
        hardware queues -- a five-stream experiment stalled in the backward pass (HISTORY.md 5.9), and the three-stream
        three-tower step was the one benchmark leg that ever hung."""
        have = self.__dict__.setdefault("_side_streams", [])
        cap = max(1, int(self.max_side_streams))
        while len(have) < min(n, cap):
            have.append(torch.cuda.Stream())
        return [have[k % len(have)] for k in range(n)] if have else []

    def forward(self, inputs: dict[str, Any]) -> dict[str, torch.Tensor]:
        outputs = {}
        # under torch.compile (mmlearn/cli/run.py:139) the step is traced: the opportunistic overlaps below (gather / matcher ahead of the
        # loss, one stream per tower) are host-side scheduling that a traced graph cannot express, the loss then does that work itself
        traced = is_compiling()
        prefetch = getattr(self.loss_fn, "prefetch_gather", None) if ("example_ids" in inputs and not traced) else None
        mods = [m for m in self._available_modalities if m.name in inputs]
        # opt-in (``task.concurrent_encoders = True``): the encoders are independent until the loss, so every modality
        # after the first gets its own HIP stream; forward AND backward kernels of the towers then overlap (autograd
        # replays each node on the stream its forward ran on), which fills the tails of kernels that do not cover 256 CUs.
        side = self._encoder_streams(len(mods) - 1) if (self.concurrent_encoders and len(mods) > 1 and not traced) else None
        main = torch.cuda.current_stream() if side else None
        early_match = getattr(self.loss_fn, "prefetch_match", None) if ("example_ids" in inputs and self.loss_fn is not None and not traced) else None
        if early_match is not None and inputs.get("fully_paired") is not True and self.match_ahead:
            # matcher + status read-back overlap the encoders; on the second tower's stream when there is one
            early_match(inputs["example_ids"], self.modality_loss_pairs, **({"stream": side[0]} if side else {}))
        for k, m in enumerate(mods):
            gather = prefetch is not None and m.name in inputs["example_ids"]
            if side and k:
                side[k - 1].wait_stream(main)
                with torch.cuda.stream(side[k - 1]):
                    out = self.encode(inputs, m, normalize=True)
                    if gather:   # launched from the producing stream: the collective must wait for THIS tower
                        prefetch(m.name, out, inputs["example_ids"][m.name])
                out.record_stream(main)
                outputs[m.embedding] = out
            else:
                outputs[m.embedding] = self.encode(inputs, m, normalize=True)
                if gather:
                    # start the global-batch all-gather of this modality while the next encoder runs (RCCL stream)
                    prefetch(m.name, outputs[m.embedding], inputs["example_ids"][m.name])
        if side:
            for st in side[: len(mods) - 1]:
                main.wait_stream(st)
        dims = {o.size(-1) for o in outputs.values()}
        if len(dims) > 1:
            raise ValueError("Expected all model outputs to have the same dimension.")
        return outputs

    # ------------------------------------------------------------------ train
    def on_train_epoch_start(self) -> None:
        self.encoders.train()
        if self.heads:
            self.heads.train()
        if self.postprocessors:
            self.postprocessors.train()

    def training_step(self, batch: dict[str, Any], batch_idx: int) -> torch.Tensor:
        outputs = self(batch)
        with torch.no_grad():  # in place, after the encoders ran, before the loss reads exp() (reference :488-489)
            self.log_logit_scale.clamp_(0, math.log(self.max_logit_scale))
        loss = self._compute_loss(batch, batch_idx, outputs)
        if loss is None:
            raise ValueError("The loss function must be provided for training.")
        self.log("train/loss", loss, prog_bar=True, sync_dist=True)
        self.log("train/logit_scale", self.log_logit_scale.exp(), prog_bar=True, on_step=True, on_epoch=False)
        return loss

    def on_before_zero_grad(self, optimizer: torch.optim.Optimizer) -> None:
        for task in self.auxiliary_tasks.values():
            task.on_before_zero_grad(optimizer)

    # ------------------------------------------------------------------ eval
    def on_validation_epoch_start(self) -> None:
        self._on_eval_epoch_start("val")

    def validation_step(self, batch: dict[str, Any], batch_idx: int) -> Optional[torch.Tensor]:
        return self._shared_eval_step(batch, batch_idx, "val")

    def on_validation_epoch_end(self) -> None:
        self._on_eval_epoch_end("val")

    def on_test_epoch_start(self) -> None:
        self._on_eval_epoch_start("test")

    def test_step(self, batch: dict[str, Any], batch_idx: int) -> Optional[torch.Tensor]:
        return self._shared_eval_step(batch, batch_idx, "test")

    def on_test_epoch_end(self) -> None:
        self._on_eval_epoch_end("test")

    def on_load_checkpoint(self, checkpoint: dict[str, Any]) -> None:
        for task in self.auxiliary_tasks.values():
            task.on_load_checkpoint(checkpoint)

    def on_save_checkpoint(self, checkpoint: dict[str, Any]) -> None:
        for task in self.auxiliary_tasks.values():
            task.on_save_checkpoint(checkpoint)

    # ------------------------------------------------------------------ internals
    def _compute_loss(self, batch: dict[str, Any], batch_idx: int, outputs: dict[str, torch.Tensor]) -> Optional[torch.Tensor]:
        if self.loss_fn is None:
            return None
        # wire-format hint of mmlearn_amd.wire.DefaultDataCollator; only a loss that declares the kwarg sees it
        hint = {}
        if batch.get("fully_paired") is True:
            if getattr(self, "_loss_takes_hint", None) is None:
                self._loss_takes_hint = "fully_paired" in inspect.signature(self.loss_fn.forward).parameters
            if self._loss_takes_hint:
                hint = {"fully_paired": True}
        contrastive_loss = self.loss_fn(outputs, batch["example_ids"], self.log_logit_scale.exp(), self.modality_loss_pairs, **hint)

        aux_losses: list[torch.Tensor] = []
        for task_name, spec in self.aux_task_specs.items():
            out = self.auxiliary_tasks[task_name].training_step(batch, batch_idx)
            if isinstance(out, torch.Tensor):
                aux = out
            elif isinstance(out, Mapping):
                aux = out["loss"]
            else:
                raise ValueError(f"Expected auxiliary task output to be a tensor or a mapping containing a 'loss' key, but got {type(out)}.")
            aux *= spec.loss_weight  # in place, like the reference (Q12)
            aux_losses.append(aux)
            if self.log_auxiliary_tasks_loss:
                self.log(f"train/{task_name}_loss", aux, sync_dist=True)
        if not aux_losses:
            return contrastive_loss
        return torch.stack(aux_losses).sum() + contrastive_loss

    def _eval_specs(self, eval_type: Literal["val", "test"]):
        for spec in (self.evaluation_tasks or {}).values():
            if (eval_type == "val" and spec.run_on_validation) or (eval_type == "test" and spec.run_on_test):
                yield spec

    def _on_eval_epoch_start(self, eval_type: Literal["val", "test"]) -> None:
        self.encoders.eval()
        if self.heads:
            self.heads.eval()
        if self.postprocessors:
            self.postprocessors.eval()
        for spec in self._eval_specs(eval_type):
            spec.task.on_evaluation_epoch_start(self)

    def _shared_eval_step(self, batch: dict[str, Any], batch_idx: int, eval_type: Literal["val", "test"]) -> Optional[torch.Tensor]:
        loss: Optional[torch.Tensor] = None
        if (eval_type == "val" and self.compute_validation_loss) or (eval_type == "test" and self.compute_test_loss):
            outputs = self(batch)
            loss = self._compute_loss(batch, batch_idx, outputs)
            if loss is not None and not self.trainer.sanity_checking:
                self.log(f"{eval_type}/loss", loss, prog_bar=True, sync_dist=True)
        for spec in self._eval_specs(eval_type):
            spec.task.evaluation_step(self, batch, batch_idx)
        return loss

    def _on_eval_epoch_end(self, eval_type: Literal["val", "test"]) -> None:
        for spec in self._eval_specs(eval_type):
            spec.task.on_evaluation_epoch_end(self)


def _is_eval_hooks(obj: Any) -> bool:
    """Accept mmlearn's own EvaluationHooks subclasses when mmlearn is installed next to this package."""
    return all(hasattr(obj, m) for m in ("on_evaluation_epoch_start", "evaluation_step", "on_evaluation_epoch_end"))
