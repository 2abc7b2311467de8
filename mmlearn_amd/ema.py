"""Exponential-moving-average teacher with a single multi-tensor HIP kernel per step.

Interface of mmlearn/modules/ema.py:10-177 (``model``, ``decay``, ``num_updates``, ``step``,
``restore``, ``state_dict``, ``configure_model``, ``get_annealed_rate``).

Observable behaviour of the reference: ``_update_weights`` tests
``param.requires_grad`` on ``state_dict()`` tensors, which are always detached, so every tensor takes
the *copy* branch and the teacher equals the student after each step, while ``decay`` and
``num_updates`` still follow the annealing schedule.  ``true_ema=False`` (default) reproduces exactly
that -- as ONE kernel launch over a device-resident pointer table instead of one
``.float()/clone/load_state_dict`` round per tensor.  ``true_ema=True`` applies the evidently intended
``teacher = decay * teacher + (1 - decay) * student`` (f32 math) to floating-point parameters and
copies buffers / ``skip_keys``.

``capturable = True`` (attribute; what ``torch.optim.AdamW(capturable=True)`` is for the optimizer): the update launch reads the
decay from a device word, and ``step()`` inside a HIP-graph capture launches only -- the host schedule (``num_updates``,
``decay``) does not move at capture time, because a capture executes nothing.  After every replay of a captured step call
:meth:`advance` (one increment of ``num_updates``, the next annealed ``decay``, one ``fill_`` of the word).  Eager steps
keep the word in step with ``decay`` by themselves.
"""

from __future__ import annotations

import copy
from typing import Any, Optional, Set, Union

import torch

import warnings

from . import kernels as K


def rank_zero_warn(msg: str, category: type = UserWarning) -> None:
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()) or torch.distributed.get_rank() == 0:
        warnings.warn(msg, category=category, stacklevel=2)


class ExponentialMovingAverage:
    def __init__(self, model: torch.nn.Module, ema_decay: float, ema_end_decay: float, ema_anneal_end_step: int,
                 skip_keys: Optional[Union[list[str], Set[str]]] = None, true_ema: bool = False) -> None:
        self.model = self.deepcopy_model(model)
        self.skip_keys: Union[list[str], set[str]] = skip_keys or set()
        self.num_updates = 0
        self.decay = ema_decay  # current decay value
        self.ema_decay = ema_decay
        self.ema_end_decay = ema_end_decay
        self.ema_anneal_end_step = ema_anneal_end_step
        self.true_ema = true_ema
        self._model_configured = False
        self._tables: dict[int, Any] = {}
        self.capturable = False
        self._decay_word: Optional[torch.Tensor] = None   # device f32[1] = self.decay (capturable mode)
        self._decay_word_value: Optional[float] = None

    @staticmethod
    def deepcopy_model(model: torch.nn.Module) -> torch.nn.Module:
        try:
            return copy.deepcopy(model)
        except RuntimeError as e:
            raise RuntimeError("Unable to copy the model ", e) from e

    @staticmethod
    def get_annealed_rate(start: float, end: float, curr_step: int, total_steps: int) -> float:
        """Linear annealing from ``start`` to ``end`` over ``total_steps``."""
        return end - (end - start) * (1 - curr_step / total_steps)

    def configure_model(self, device_id: Union[int, torch.device]) -> None:
        if self._model_configured:
            return
        self.model.requires_grad_(False)
        self.model.to(device_id)
        self._model_configured = True

    def step(self, new_model: torch.nn.Module) -> None:
        if not self._model_configured:
            raise RuntimeError("Model is not configured for EMA. Call `configure_model` first.")
        if self.capturable and torch.cuda.is_current_stream_capturing():
            self._update_weights(new_model, capturing=True)   # the launch only: see advance()
            return
        self._update_weights(new_model)
        self._update_ema_decay()
        if self.capturable:
            self._sync_decay_word()

    def advance(self) -> None:
        """Host side of one update that a replayed graph has just made on the device: ``num_updates`` + 1, the next annealed
        ``decay``, and the device word the next replay reads it from."""
        if self.decay < 1:
            self.num_updates += 1
        self._update_ema_decay()
        self._sync_decay_word()

    def _sync_decay_word(self) -> None:
        dev = next(self.model.parameters()).device
        if self._decay_word is None or self._decay_word.device != dev:
            self._decay_word = torch.empty(1, dtype=torch.float32, device=dev)
            self._decay_word_value = None
        if self._decay_word_value != float(self.decay):
            self._decay_word.fill_(float(self.decay))
            self._decay_word_value = float(self.decay)

    def restore(self, model: torch.nn.Module) -> torch.nn.Module:
        """Load the teacher's weights into ``model`` (strict=False), as the reference does."""
        model.load_state_dict(self.model.state_dict(), strict=False)
        return model

    def state_dict(self) -> dict[str, Any]:
        return self.model.state_dict()

    # ------------------------------------------------------------------ internals
    def _build_tables(self, new_model: torch.nn.Module):
        teacher, student = self.model.state_dict(), new_model.state_dict()
        trainable = {k for k, p in new_model.named_parameters() if p.requires_grad}
        groups: dict[str, tuple[list, list]] = {"copy": ([], []), "ema": ([], [])}
        slow: list[tuple[torch.Tensor, torch.Tensor]] = []
        for key, s in student.items():
            t = teacher[key]
            if s.shape != t.shape:
                raise ValueError("Incompatible tensor shapes between student param and teacher param" + f"{s.shape} vs. {t.shape}")
            K.require_gpu(t, "teacher tensor")
            K.require_gpu(s, "student tensor")
            mode = "ema" if (self.true_ema and key in trainable and key not in self.skip_keys) else "copy"
            fdt = (torch.float32, torch.bfloat16, torch.float16)
            kernel_ok = t.dtype in fdt and s.dtype in fdt and t.is_contiguous() and s.is_contiguous() and t.numel() > 0
            if not kernel_ok:
                if mode == "ema":
                    raise ValueError(f"EMA of {key!r}: need contiguous f32/bf16/f16 tensors, got {t.dtype}/{s.dtype}")
                slow.append((t, s))  # integer buffers such as num_batches_tracked: a plain device copy
                continue
            groups[mode][0].append(t)
            groups[mode][1].append(s)
        tables = {m: (K.ema_table(ts, ss) if ts else None) for m, (ts, ss) in groups.items()}
        keep = [x for ts, ss in groups.values() for x in (ts, ss)]
        return tables, slow, keep

    @torch.no_grad()
    def _update_weights(self, new_model: torch.nn.Module, capturing: bool = False) -> None:
        if self.decay < 1:
            # keyed on the storages themselves: ``.to()`` / ``.half()`` / re-wrapping re-allocates parameters, and a table
            # built for the old storages would keep updating tensors nobody reads
            # (parameters() / buffers() walks, not state_dict(): no OrderedDict build, no prefix strings, no state-dict hooks
            # on the critical path of every step -- ADVICE r2)
            key = (id(new_model), tuple(t.data_ptr() for m in (self.model, new_model) for it in (m.parameters(), m.buffers()) for t in it))
            if key not in self._tables:
                if capturing:
                    raise RuntimeError("ExponentialMovingAverage: the pointer tables must exist before capture -- run one eager "
                                       "step() with the same student first")
                self._tables = {key: self._build_tables(new_model)}
            tables, slow, _ = self._tables[key]
            decay = self.decay
            if self.capturable:
                if capturing and (self._decay_word is None or self._decay_word_value != float(self.decay)):
                    raise RuntimeError("ExponentialMovingAverage(capturable): run one eager step() before capturing")
                if not capturing:
                    self._sync_decay_word()
                decay = self._decay_word
            for mode, tab in tables.items():
                if tab is not None:
                    K.ema_update(*tab, decay, mode == "ema")
            for t, s in slow:
                t.copy_(s)
            if not capturing:
                self.num_updates += 1
        else:
            rank_zero_warn("Exponential Moving Average decay is 1.0, no update is applied to the model.", category=UserWarning)

    def _update_ema_decay(self) -> None:
        if self.ema_decay != self.ema_end_decay:
            if self.num_updates >= self.ema_anneal_end_step:
                self.decay = self.ema_end_decay
            else:
                self.decay = self.get_annealed_rate(self.ema_decay, self.ema_end_decay, self.num_updates, self.ema_anneal_end_step)
