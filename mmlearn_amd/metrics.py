"""Retrieval Recall@K on MI355X: the plugin surface of mmlearn's ``RetrievalRecallAtK``
(mmlearn/modules/metrics/retrieval_recall.py:21-289 -- ``top_k`` / ``reduction`` / ``aggregation`` constructor,
``update(x, y, indexes)``, ``compute()``) over ``mmk_recall_ranks``.

The reference keeps every embedding, moves them to the CPU and, per batch of queries, materialises the ``[b, M]`` score
matrix, runs ``torch.topk`` and gathers the positive's membership.  Here ``compute()`` is two launches of one tiled f32
similarity kernel with a counting epilogue (the rank of each query's positive), nothing ``N x M`` is stored, and recall is
``rank < top_k``.  Ties are broken by the lower database index (``torch.topk`` leaves them unspecified).  There is no CPU
path.  ``reduction`` must be ``"none"`` / ``None``: with ``"sum"`` / ``"mean"`` (the reference's default!) the reference
reduces the scores to 1-D before indexing them with the positives and fails, so there is no behaviour to match.
"""

from __future__ import annotations

from typing import Any, Callable, List, Optional, Union

import torch
import torch.distributed as dist

from . import kernels as K
from .ops import l2_normalize

_AGG = {"mean": lambda v: v.mean(), "median": lambda v: v.median(), "min": lambda v: v.min(), "max": lambda v: v.max()}


class RetrievalRecallAtK(torch.nn.Module):
    is_differentiable = False
    higher_is_better = True
    full_state_update = False

    def __init__(self, top_k: int, reduction: Optional[str] = "sum", aggregation: Union[str, Callable] = "mean", **kwargs: Any) -> None:
        super().__init__()
        if top_k is not None and not (isinstance(top_k, int) and top_k > 0):
            raise ValueError("`top_k` has to be a positive integer or None")
        allowed_reduction = ("sum", "mean", "none", None)
        if reduction not in allowed_reduction:
            raise ValueError(f"Expected argument `reduction` to be one of {allowed_reduction} but got {reduction}")
        if not (aggregation in ("mean", "median", "min", "max") or callable(aggregation)):
            raise ValueError("Argument `aggregation` must be one of `mean`, `median`, `min`, `max` or a custom callable function"
                             f"which takes tensor of values, but got {aggregation}.")
        self.top_k, self.reduction, self.aggregation = top_k, reduction, aggregation
        self.process_group = kwargs.get("process_group")
        self.reset()

    def reset(self) -> None:
        self.x: List[torch.Tensor] = []
        self.y: List[torch.Tensor] = []
        self.indexes: List[torch.Tensor] = []
        self.num_samples = 0

    def _world(self) -> int:
        return dist.get_world_size(self.process_group) if dist.is_available() and dist.is_initialized() else 1

    def update(self, x: torch.Tensor, y: torch.Tensor, indexes: torch.Tensor) -> None:
        """``x [N, D]``, ``y [N, D]`` (unnormalised), ``indexes [N]``: the row of THIS call's ``y`` matching each ``x`` row."""
        if indexes is None:
            raise ValueError("Argument `indexes` cannot be None")
        if x.shape != y.shape:
            raise RuntimeError(f"Predictions and targets are expected to have the same shape, but got {x.shape} and {y.shape}.")
        if indexes.dtype != torch.long and indexes.dtype != torch.int64:
            raise ValueError("Argument `indexes` must be a tensor of long integers")
        if indexes.numel() != x.shape[0]:
            raise ValueError("`indexes` needs one entry per row of `x`")
        K.require_gpu(x)
        x, y, indexes = x.detach().float(), y.detach().float(), indexes.detach().clone()
        world = self._world()
        if world > 1:  # every rank keeps the global set, as the reference does (retrieval_recall.py:139-160)
            sizes = torch.zeros(world, dtype=torch.long, device=x.device)
            sizes[dist.get_rank(self.process_group)] = x.shape[0]
            dist.all_reduce(sizes, group=self.process_group)
            if int(sizes.min()) != int(sizes.max()):
                raise RuntimeError("mmlearn_amd.RetrievalRecallAtK needs equal per-rank batch sizes")
            gx, gy, gi = (torch.empty((world * t.shape[0],) + t.shape[1:], dtype=t.dtype, device=t.device) for t in (x, y, indexes))
            for out, inp in ((gx, x), (gy, y), (gi, indexes)):
                dist.all_gather_into_tensor(out, inp.contiguous(), group=self.process_group)
            b = x.shape[0]
            gi = gi + (torch.arange(world, device=gi.device) * b).repeat_interleave(b)   # rank r's positives sit after r batches
            x, y, indexes = gx, gy, gi
        self.x.append(x)
        self.y.append(y)
        self.indexes.append(indexes + self.num_samples)
        self.num_samples += x.shape[0]

    def ranks(self) -> torch.Tensor:
        """int32 ``[N_total]``: how many database rows score above each query's positive."""
        if not self.x:
            raise RuntimeError("no samples: call update() first")
        x, y, idx = torch.cat(self.x), torch.cat(self.y), torch.cat(self.indexes)
        return K.recall_ranks(l2_normalize(x), l2_normalize(y), idx)

    def compute(self) -> torch.Tensor:
        if self.reduction not in ("none", None):
            raise ValueError("RetrievalRecallAtK: only reduction='none' / None is computable (the reference fails for 'sum' / 'mean')")
        hits = (self.ranks() < self.top_k).float()
        return self.aggregation(hits, dim=0) if callable(self.aggregation) else _AGG[self.aggregation](hits)

    def forward(self, *args: Any, **kwargs: Any) -> Any:
        raise NotImplementedError("RetrievalRecallAtK metric does not support forward method")
