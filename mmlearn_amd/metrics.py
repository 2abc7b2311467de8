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


# ----------------------------------------------------------------------------------------------------------------------
# Zero-shot classification (mmlearn/tasks/zero_shot_classification.py:160-219): same similarity + top-k pattern, the
# database being the class prototypes and the positive the target class.


def class_prototypes(prompt_embeddings: torch.Tensor, num_templates: int) -> torch.Tensor:
    """``[C * T, D]`` prompt embeddings (class-major, as the reference builds them, :150-154) -> ``[C, D]`` prototypes:
    normalise, mean over the ``T`` templates, normalise again (:164-168)."""
    if prompt_embeddings.dim() != 2 or prompt_embeddings.shape[0] % num_templates:
        raise ValueError("prompt_embeddings must be [num_classes * num_templates, D]")
    K.require_gpu(prompt_embeddings)
    e = l2_normalize(prompt_embeddings.detach().float())
    return l2_normalize(e.view(-1, num_templates, e.shape[-1]).mean(dim=1))


def zero_shot_logits(query_embeddings: torch.Tensor, class_embeddings: torch.Tensor) -> torch.Tensor:
    """The logits ``evaluation_step`` hands to its metrics (:201-214): ``100 * q_n @ C^T``, or for two classes the
    difference of the softmax columns."""
    K.require_gpu(query_embeddings)
    s = l2_normalize(query_embeddings.detach().float()) @ class_embeddings.float().T
    if class_embeddings.shape[0] == 2:
        p = s.softmax(dim=-1)
        return p[:, 1] - p[:, 0]
    return 100.0 * s


class ZeroShotTopKAccuracy(torch.nn.Module):
    """Micro top-k accuracy of the multiclass zero-shot head for several ``k`` at once (the reference builds one
    torchmetrics ``Accuracy(task="multiclass", top_k=k, average="micro")`` per ``k``, :240-252, each re-running
    ``topk`` on the ``[B, C]`` logits).  ``update`` asks the counting kernel for the rank of the target class of every
    query and keeps only a histogram of ranks below ``max(top_k)``; the logits are never materialised.  Ties: the lower
    class index wins."""

    def __init__(self, top_k=(1,)) -> None:
        super().__init__()
        self.top_k = tuple(int(k) for k in top_k)
        if not self.top_k or min(self.top_k) < 1:
            raise ValueError("`top_k` needs positive integers")
        self.reset()

    def reset(self) -> None:
        self.hist: Optional[torch.Tensor] = None   # int64 [max_k + 1]; the last bin holds every rank >= max_k
        self.total = 0

    def update(self, query_embeddings: torch.Tensor, class_embeddings: torch.Tensor, targets: torch.Tensor) -> None:
        if targets.numel() != query_embeddings.shape[0]:
            raise ValueError("`targets` needs one class index per query")
        if not query_embeddings.shape[0]:
            return
        K.require_gpu(query_embeddings)
        r = K.recall_ranks(l2_normalize(query_embeddings.detach().float()), class_embeddings.float().contiguous(),
                           targets.to(torch.int64))
        kmax = max(self.top_k)
        h = torch.bincount(r.clamp(max=kmax).long(), minlength=kmax + 1)
        self.hist = h if self.hist is None else self.hist + h
        self.total += query_embeddings.shape[0]

    def compute(self) -> dict:
        if self.hist is None:
            raise RuntimeError("no samples: call update() first")
        c = self.hist.cumsum(0).double() / self.total
        return {k: c[k - 1].float() for k in self.top_k}
